"""Development probe for the one-sweep attention backward (hs_attn_bwd2): fp32 torch reference on the same bf16 inputs,
errors broken down by section (dq | dk | dv), key half / d half / key block, so a wrong operand map shows its shape.
   python tools/dev/attn1_debug.py [B] [N] [H]          (HS_TILES=1: per-key-block dQ tiles + a second kernel instead of the ordered hand-off)"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from hsenet_amd import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N = int(sys.argv[2]) if len(sys.argv) > 2 else 513
H = int(sys.argv[3]) if len(sys.argv) > 3 else 12
mode = sys.argv[4] if len(sys.argv) > 4 else "gauss"
dev = torch.device("cuda:0")
torch.manual_seed(0)
Dm = H * 64
qkv = torch.randn(B, N, 3, H, 64, device=dev).to(torch.bfloat16)
d_o = torch.randn(B, N, H, 64, device=dev).to(torch.bfloat16)
if mode == "ones":          # dO = 1 everywhere: dV[key] = sum_q P[q][key] in every d column
    d_o = torch.ones_like(d_o)
lib = _lib.load()
if os.environ.get("HS_DBG4"):
    lib.hs_debug_set(4, int(os.environ["HS_DBG4"]))
o = torch.empty(B, N, Dm, device=dev, dtype=torch.bfloat16)
lse = torch.empty(B, H, N, device=dev, dtype=torch.float32)
st = torch.cuda.current_stream().cuda_stream
_lib.call("hs_attn_fwd", qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), B, N, H, 64, _lib.PREC_BF16, 0, st)
fl = _lib.ATTN_ONE_SWEEP | (_lib.ATTN_ONE_SWEEP_TILES if os.environ.get('HS_TILES') else 0)
wsb = lib.hs_attn_bwd_ws_bytes(B, N, H, fl)
ws = torch.zeros(wsb, device=dev, dtype=torch.uint8)
dqkv = torch.zeros(B, N, 3, H, 64, device=dev, dtype=torch.bfloat16)
_lib.call("hs_attn_bwd2", qkv.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), ws.data_ptr(), wsb, B, N, H, 64, fl, st)
torch.cuda.synchronize()
print("ws bytes", wsb, "error word", lib.hs_attn_bwd_ws_error(ws.data_ptr(), B, N, H, fl, st))
# fp32 reference on the same bf16 inputs
q = qkv[:, :, 0].float().permute(0, 2, 1, 3).requires_grad_(True)
k = qkv[:, :, 1].float().permute(0, 2, 1, 3).requires_grad_(True)
v = qkv[:, :, 2].float().permute(0, 2, 1, 3).requires_grad_(True)
s = (q @ k.transpose(-1, -2)) * 0.125
p = torch.softmax(s, -1)
out = p @ v
out.backward(d_o.float().permute(0, 2, 1, 3))
ref = torch.stack([q.grad, k.grad, v.grad], 0).permute(1, 3, 0, 2, 4)      # [B][N][3][H][64]
got = dqkv.float()
def rel(a, b): return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
for i, nm in enumerate(["dq", "dk", "dv"]):
    print(nm, "rel", f"{rel(got[:, :, i], ref[:, :, i]):.3e}", " norm got/ref", f"{(got[:, :, i].norm() / ref[:, :, i].norm()).item():.4f}")
for i, nm in ((1, "dk"), (2, "dv")):
    g, r = got[:, :, i], ref[:, :, i]                      # [B][N][H][64]
    M = (N // 256) * 256
    if M:
        gb, rb = g[:, :M].reshape(B, M // 256, 4, 2, 32, H, 2, 32), r[:, :M].reshape(B, M // 256, 4, 2, 32, H, 2, 32)
        print(nm, "by wave   ", [f"{rel(gb[:, :, w], rb[:, :, w]):.2e}" for w in range(4)])
        print(nm, "by key half", [f"{rel(gb[:, :, :, kh], rb[:, :, :, kh]):.2e}" for kh in range(2)])
        print(nm, "by d half  ", [f"{rel(gb[..., db, :], rb[..., db, :]):.2e}" for db in range(2)])
        print(nm, "by key block", [f"{rel(gb[:, m], rb[:, m]):.2e}" for m in range(M // 256)])
        print(nm, "ratio got/ref of means", f"{(gb.mean() / rb.mean()).item():.4f}", "| scale fit", f"{((gb * rb).sum() / (rb * rb).sum()).item():.4f}")
        e = (gb - rb)
        print(nm, "by key %% 32   ", " ".join(f"{(e[:, :, :, :, c_].norm() / rb[:, :, :, :, c_].norm()).item():.2f}" for c_ in range(32)))
        print(nm, "by d %% 32     ", " ".join(f"{(e[..., c_].norm() / rb[..., c_].norm()).item():.2f}" for c_ in range(32)))
    if N > M:
        print(nm, "remainder key", f"{rel(g[:, M:], r[:, M:]):.2e}")
g, r = got[:, :, 0], ref[:, :, 0]
print("dq by 32-row slice (first 6, last 2)", [f"{rel(g[:, 32 * s_:32 * s_ + 32], r[:, 32 * s_:32 * s_ + 32]):.2e}" for s_ in list(range(min(6, N // 32))) + [N // 32 - 1, N // 32][: 2 if N >= 64 else 0]])
print("dq by head", [f"{rel(g[:, :, h_], r[:, :, h_]):.2e}" for h_ in range(H)])
