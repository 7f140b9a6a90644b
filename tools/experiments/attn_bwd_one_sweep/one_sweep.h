/* Declarations of the one-sweep attention backward EXPERIMENT (round 4; retired from the product library in round 5: measured 12.7 % slower
 * than the shipped two sweeps).  Built by ./build.sh into its own shared object beside the product's objects; nothing in hsenet_amd/ or
 * include/hsenet_hip.h refers to it. */
#pragma once
#include "../../../include/hsenet_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* The same gradients with a caller-sized workspace (bytes from hs_attn_bwd_ws_bytes for the same flags, 256-B aligned, ZEROED before its
 * first use).  Without HS_ATTN_ONE_SWEEP — or with HS_ATTN_F32, or when the shape does not fit — this is exactly hs_attn_bwd above
 * (the workspace then only has to hold its 2 x [B][H][N] floats).
 * HS_ATTN_ONE_SWEEP (bf16, N = 256 * 2^k (+ 1), k <= 5) selects the one-sweep kernel of csrc/attn_bwd1.hip: S and dP computed once
 * (10 N^2 d executed instead of 14), dQ summed over the 256-key blocks by an ordered, bitwise reproducible hand-off between 256
 * persistent workgroups.  It is correct and tested, and on MI355X it is SLOWER than the two sweeps (one
 * wave per SIMD issues every LDS / vector-memory / VALU instruction of a slice in one stream), so nothing in the library selects it.
 * HS_ATTN_ONE_SWEEP_TILES (with HS_ATTN_ONE_SWEEP; diagnostic): every key block stores its own dQ tiles and a second kernel adds them.
 * hs_attn_bwd_ws_error reads the error word the one-sweep kernel leaves in its workspace (0 ok; 1 = a bounded hand-off wait ran out and
 * the gradients of that call are void); it synchronises the stream and is meant for tests and diagnostics. */
#define HS_ATTN_ONE_SWEEP 4
#define HS_ATTN_ONE_SWEEP_TILES 8
size_t hs_attn_bwd_ws_bytes(int B, int N, int H, int flags);
int hs_attn_bwd2(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv, void* ws, size_t ws_bytes,
                 int B, int N, int H, int dh, int flags, void* stream);
int hs_attn_bwd_ws_error(const void* ws, int B, int N, int H, int flags, void* stream);
#ifdef __cplusplus
}
#endif
