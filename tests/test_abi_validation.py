"""The C ABI refuses bad arguments with a status BEFORE anything is launched (include/hsenet_hip.h: "int status, 0 ok, negative =
bad shape / argument").  A kernel launched on a shape its grid does not assume is a GPU fault, not an exception, so the host-side
checks are part of the boundary.  These tests make no compute call: every call below must return from argument validation.  They
run where there is NO GPU (the pointers are fakes that must never be dereferenced; on a GPU box a check that wrongly passed would
launch on them, so the shape cases are skipped there — the NULL cases run everywhere)."""
import ctypes as C

import pytest
import torch

from hsenet_amd import _lib as L

NO_COMPUTE = {"hs_version", "hs_build_id", "hs_device_check", "hs_prof_enable", "hs_prof_read", "hs_prof_name", "hs_prof_count", "hs_debug_set",
              "hs_debug_gemm_stamps", "hs_debug_gemm_trace"}
FAKE = 0x10000          # 16-byte aligned, never dereferenced
has_gpu = torch.cuda.is_available()
no_gpu_only = pytest.mark.skipif(has_gpu, reason="fake pointers: shape-validation cases run only where nothing can be launched")


def _zeros(name):
    vals = []
    for a in L.SIGNATURES[name]:
        if a is C.c_void_p:
            vals.append(None)
        elif a is C.c_float:
            vals.append(0.0)
        else:
            try:
                vals.append(a(0))
            except TypeError:            # POINTER(struct): NULL
                vals.append(None)
    return vals


@pytest.mark.parametrize("name", sorted(n for n in L.SIGNATURES if n not in NO_COMPUTE))
def test_null_arguments_are_refused(name):
    """every compute entry point: all pointers NULL, all sizes 0 -> negative status (or 0 bytes for the workspace queries)."""
    rc = getattr(L.load(), name)(*_zeros(name))
    if name.endswith("_workspace_bytes") or name.endswith("_ws_bytes") or name.endswith("_elems"):
        assert rc == 0
    else:
        assert rc in (L.HS_ERR_ARG, L.HS_ERR_SHAPE), (name, rc)


def _gemm(**kw):
    g = L.GemmArgs()
    g.A, g.B, g.out_f32 = FAKE, FAKE, FAKE
    g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.batch, g.split_k, g.alpha, g.layout = 256, 256, 256, 256, 256, 256, 1, 1, 1.0, L.GEMM_NT
    for k, v in kw.items():
        setattr(g, k, v)
    return g


@no_gpu_only
@pytest.mark.parametrize("kw,prec,want", [
    (dict(K=0), L.PREC_BF16, L.HS_ERR_SHAPE),
    (dict(K=252, lda=252, ldb=252), L.PREC_BF16, L.HS_ERR_SHAPE),            # bf16 rows must be 16-byte multiples
    (dict(A=FAKE + 2), L.PREC_BF16, L.HS_ERR_SHAPE),                        # operand not 16-byte aligned
    (dict(layout=7), L.PREC_BF16, L.HS_ERR_ARG),
    (dict(flags=L.EPI_BIAS), L.PREC_BF16, L.HS_ERR_ARG),                    # bias flag without a bias pointer
    (dict(flags=L.EPI_RESID), L.PREC_BF16, L.HS_ERR_ARG),
    (dict(flags=L.EPI_GELU_BWD), L.PREC_BF16, L.HS_ERR_ARG),
    (dict(split_k=4), L.PREC_BF16, L.HS_ERR_SHAPE),                         # split-K needs the atomic epilogue
    (dict(split_k=0), L.PREC_BF16, L.HS_ERR_SHAPE),
    (dict(batch=0), L.PREC_BF16, L.HS_ERR_SHAPE),
    (dict(scale_cols=6), L.PREC_BF16, L.HS_ERR_SHAPE),
    (dict(), 9, L.HS_ERR_ARG),                                              # unknown precision
    (dict(out_f32=None), L.PREC_BF16, L.HS_ERR_ARG),                        # no output at all
])
def test_gemm_argument_checks(kw, prec, want):
    assert L.load().hs_gemm(C.byref(_gemm(**kw)), prec, None) == want


@no_gpu_only
def test_kernel_shape_checks():
    lib, P, S = L.load(), FAKE, L.HS_ERR_SHAPE
    assert lib.hs_gemm_group(C.byref(_gemm()), 0, None) < 0
    assert lib.hs_gemm_group(C.byref(_gemm()), 99, None) < 0
    # patchify: the volume must tile exactly into patches
    assert lib.hs_patchify(P, P, 1, 1, 30, 256, 256, 4, 16, 16, L.PREC_BF16, None) == S
    assert lib.hs_patchify(P, P, 0, 1, 32, 256, 256, 4, 16, 16, L.PREC_BF16, None) == S
    assert lib.hs_patchify(P, P, 1, 1, 32, 256, 256, 4, 16, 16, 5, None) == L.HS_ERR_ARG
    # LayerNorm: d a multiple of 4, at most 1024
    for d in (770, 2048, 0):
        assert lib.hs_layernorm_fwd(P, None, P, P, P, None, P, P, 8, d, 1e-5, None) == S
        assert lib.hs_layernorm_bwd(P, 0, P, None, P, P, P, None, P, None, P, P, None, 8, d, None) == S
    assert lib.hs_layernorm_fwd(P, None, P, P, None, None, P, P, 8, 768, 1e-5, None) == L.HS_ERR_ARG     # no output requested
    # attention: head dim 64 only, known flags only, 16-byte aligned tensors
    assert lib.hs_attn_fwd(P, P, P, 1, 128, 12, 32, L.PREC_BF16, 0, None) == S
    assert lib.hs_attn_fwd(P, P, P, 1, 0, 12, 64, L.PREC_BF16, 0, None) == S
    assert lib.hs_attn_fwd(P + 4, P, P, 1, 128, 12, 64, L.PREC_BF16, 0, None) == S
    assert lib.hs_attn_fwd(P, P, P, 1, 128, 12, 64, L.PREC_BF16, 64, None) == L.HS_ERR_ARG
    assert lib.hs_attn_fwd(P, P, None, 1, 128, 12, 64, L.PREC_BF16, 0, None) == L.HS_ERR_ARG              # bf16 needs the lse output
    assert lib.hs_attn_bwd(P, P, P, P, P, P, 1, 128, 12, 48, 0, None) == S
    assert lib.hs_attn_bwd(P, P, P, P, P, P, 1, 128, 12, 64, 4, None) == L.HS_ERR_ARG                       # unknown flag (4 / 8 were the retired one-sweep's)
    assert lib.hs_attn_bwd(P, P, P, P, P, P, 1, 128, 12, 64, L.ATTN_F32 | L.ATTN_F16, None) == L.HS_ERR_ARG   # one operand format at a time


@no_gpu_only
def test_driver_descriptor_checks():
    """the fused drivers: descriptor geometry, workspace size and alignment."""
    lib = L.load()
    ok = L.VitDesc(1, 33, 768, 12, 3072, 2, L.PREC_BF16, 1, 0.0, None)
    need = lib.hs_vit_workspace_bytes(C.byref(ok))
    assert need > 0
    for bad in (L.VitDesc(1, 33, 768, 8, 3072, 2, L.PREC_BF16, 1, 0.0, None),        # d != 64 * heads
                L.VitDesc(0, 33, 768, 12, 3072, 2, L.PREC_BF16, 1, 0.0, None),
                L.VitDesc(1, 33, 768, 12, 3070, 2, L.PREC_BF16, 1, 0.0, None),
                L.VitDesc(1, 33, 768, 12, 3072, 2, 5, 1, 0.0, None),
                L.VitDesc(1, 33, 768, 12, 3072, 2, L.PREC_BF16, 1, -1.0, None)):
        assert lib.hs_vit_workspace_bytes(C.byref(bad)) == 0
        assert lib.hs_vit_blocks_fwd(C.byref(bad), FAKE, FAKE, FAKE, FAKE, FAKE, FAKE, 1 << 40, None) < 0
    tbl = (C.c_void_p * 64)(*([FAKE] * 64))
    ws = 1 << 20                                                     # 256-byte aligned fake workspace
    assert lib.hs_vit_blocks_fwd(C.byref(ok), FAKE, tbl, tbl, FAKE, FAKE, ws, need - 1, None) == L.HS_ERR_SHAPE      # workspace too small
    assert lib.hs_vit_blocks_fwd(C.byref(ok), FAKE, tbl, tbl, FAKE, FAKE, ws + 16, need, None) == L.HS_ERR_SHAPE     # misaligned workspace
    assert lib.hs_vit_blocks_bwd(C.byref(ok), FAKE, tbl, tbl, FAKE, FAKE, tbl, tbl, FAKE, ws, need, 3, 0, None) == L.HS_ERR_SHAPE   # l_hi > L
    assert lib.hs_vit_blocks_bwd(C.byref(ok), FAKE, tbl, tbl, FAKE, FAKE, tbl, tbl, FAKE, ws, need, 1, 1, None) == L.HS_ERR_SHAPE   # empty range
    fwd_only = L.VitDesc(1, 33, 768, 12, 3072, 2, L.PREC_BF16, 0, 0.0, None)
    assert lib.hs_vit_blocks_bwd(C.byref(fwd_only), FAKE, tbl, tbl, FAKE, FAKE, tbl, tbl, FAKE, ws, 1 << 40, 2, 0, None) == L.HS_ERR_ARG

    pk = L.PackerDesc(2, 2, 4, 4, 1, 2, 2, 768, 256, 1, 33 * 768, 2 * 8 * 256, 1, 0.0)
    pneed = lib.hs_packer_workspace_bytes(C.byref(pk))
    assert pneed > 0
    for bad in (L.PackerDesc(2, 2, 4, 4, 3, 2, 2, 768, 256, 1, 33 * 768, 2 * 8 * 256, 1, 0.0),          # kernel larger than the grid
                L.PackerDesc(2, 2, 4, 4, 1, 2, 2, 768, 256, 1, 32 * 768, 2 * 8 * 256, 1, 0.0),          # input stride shorter than the rows read
                L.PackerDesc(2, 2, 4, 4, 1, 2, 2, 768, 256, 1, 33 * 768, 8 * 256 - 8, 1, 0.0),          # output stride shorter than the rows written
                L.PackerDesc(2, 2, 4, 4, 1, 2, 2, 768, 256, 1, 33 * 768, 2 * 8 * 256, 1, 1.0)):         # dropout probability 1
        assert lib.hs_packer_workspace_bytes(C.byref(bad)) == 0
    # a PADDED input stride is fine for the forward but refused by the backward when dfeat is requested (dfeat is written dense)
    padded = L.PackerDesc(2, 2, 4, 4, 1, 2, 2, 768, 256, 1, 40 * 768, 2 * 8 * 256, 1, 0.0)
    pn = lib.hs_packer_workspace_bytes(C.byref(padded))
    assert pn > 0
    assert lib.hs_packer_bwd(C.byref(padded), FAKE, tbl, tbl, FAKE, ws, pn, None) == L.HS_ERR_SHAPE
    assert lib.hs_packer_bwd(C.byref(pk), FAKE, tbl, tbl, FAKE, ws, pneed - 1, None) == L.HS_ERR_SHAPE

    em = L.EmbedDesc(1, 32, 1024, 768, 1, 1, 32, 1, 0.0)
    assert lib.hs_embed_workspace_bytes(C.byref(em)) > 0
    assert lib.hs_embed_workspace_bytes(C.byref(L.EmbedDesc(1, 0, 1024, 768, 1, 1, 32, 1, 0.0))) == 0
    # round 5: the drivers' 16-bit operand format is a descriptor field: bf16 or fp16, nothing else (the exact-f32 mode is the module API's)
    for prec, ok_ in ((L.PREC_BF16, True), (L.PREC_F16, True), (L.PREC_F32, False), (7, False)):
        em.prec, pk.prec = prec, prec
        assert (lib.hs_embed_workspace_bytes(C.byref(em)) > 0) == ok_ and (lib.hs_packer_workspace_bytes(C.byref(pk)) > 0) == ok_
    assert lib.hs_gemm_group_prec(C.byref(_gemm()), 1, L.PREC_F32, None) == L.HS_ERR_ARG
    assert lib.hs_mse_grad_scaled(FAKE, FAKE, None, 64, 0.0, None) == L.HS_ERR_ARG                       # the loss scale must be positive
    assert lib.hs_layernorm_bwd(FAKE, 3, FAKE, None, FAKE, FAKE, FAKE, None, FAKE, None, None, None, None, 8, 64, None) == L.HS_ERR_ARG
    assert lib.hs_colsum(FAKE, 3, FAKE, 8, 64, 64, None) == L.HS_ERR_ARG


@no_gpu_only
def test_abi2_struct_size_weight_copies_and_fused_adamw_checks():
    """ABI 2 (round 6): descriptors carry the size of the struct their caller was compiled against; a mismatch is refused before
    any field is read.  Caller-owned 16-bit weight copies and the fused optimizer sweep validate their arguments."""
    lib = L.load()
    assert lib.hs_version() == L.ABI_VERSION == 2
    tbl = (C.c_void_p * 64)(*([FAKE] * 64))
    for mk, query in ((lambda: L.VitDesc(1, 33, 768, 12, 3072, 2, L.PREC_BF16, 1, 0.0, None), lib.hs_vit_workspace_bytes),
                      (lambda: L.EmbedDesc(1, 32, 1024, 768, 1, 1, 32, 1, 0.0), lib.hs_embed_workspace_bytes),
                      (lambda: L.PackerDesc(2, 2, 4, 4, 1, 2, 2, 768, 256, 1, 33 * 768, 2 * 8 * 256, 1, 0.0), lib.hs_packer_workspace_bytes)):
        d = mk()
        assert d.struct_size == C.sizeof(d) and query(C.byref(d)) > 0
        for wrong in (0, C.sizeof(d) - 4, C.sizeof(d) + 8):        # an older / newer header's struct
            d.struct_size = wrong
            assert query(C.byref(d)) == 0
    v = L.VitDesc(1, 33, 768, 12, 3072, 2, L.PREC_BF16, 1, 0.0, None)
    v.struct_size -= 16                                            # the round-5 layout (no w16 / w16t): refused, not read past
    assert lib.hs_vit_blocks_fwd(C.byref(v), FAKE, tbl, tbl, FAKE, FAKE, 1 << 20, 1 << 40, None) == L.HS_ERR_ARG
    # caller-owned weight copies: 16-bit precisions only, 16-byte aligned, transposed copies need 64-multiples; they shrink the workspace
    ok = L.VitDesc(1, 33, 768, 12, 3072, 2, L.PREC_BF16, 1, 0.0, None)
    need = lib.hs_vit_workspace_bytes(C.byref(ok))
    n = lib.hs_vit_weights_elems(C.byref(ok))
    assert n == 2 * (4 * 768 * 768 + 2 * 768 * 3072)
    ok.w16, ok.w16t = FAKE, FAKE
    assert lib.hs_vit_workspace_bytes(C.byref(ok)) == need - 2 * (2 * n)
    ok.w16 = FAKE + 8
    assert lib.hs_vit_workspace_bytes(C.byref(ok)) == 0
    f32 = L.VitDesc(1, 33, 768, 12, 3072, 2, L.PREC_F32, 1, 0.0, None)
    assert lib.hs_vit_weights_elems(C.byref(f32)) == 0
    f32.w16 = FAKE
    assert lib.hs_vit_workspace_bytes(C.byref(f32)) == 0
    odd = L.VitDesc(1, 33, 64, 1, 72, 2, L.PREC_BF16, 1, 0.0, None)                      # mlp not a multiple of 64: no transposed copies
    assert lib.hs_vit_workspace_bytes(C.byref(odd)) > 0
    assert lib.hs_vit_cast_weights(C.byref(odd), tbl, FAKE, FAKE, None) == L.HS_ERR_SHAPE
    plain = L.VitDesc(1, 33, 768, 12, 3072, 2, L.PREC_BF16, 1, 0.0, None)
    assert lib.hs_vit_cast_weights(C.byref(plain), tbl, None, None, None) == L.HS_ERR_ARG           # nothing to make
    assert lib.hs_vit_cast_weights(C.byref(plain), None, FAKE, None, None) == L.HS_ERR_ARG          # no source for the straight copy
    assert lib.hs_vit_cast_weights(C.byref(plain), None, None, FAKE, None) == L.HS_ERR_ARG
    assert lib.hs_vit_cast_weights(C.byref(f32), tbl, FAKE, None, None) < 0
    # hs_adamw_fused: whole 64-element chunks, a state block, a norm when clipping, a positive gradient scale
    P = FAKE
    args = lambda **k: [k.get("p", P), P, P, P, None, k.get("n", 128), 1e-3, 0.9, 0.999, 1e-8, 0.1, k.get("gs", 1.0), k.get("gsq", None),
                        k.get("mx", 0.0), 1, None, 0, k.get("st", P), k.get("phase", L.ADAMW_STEP), None]
    assert lib.hs_adamw_fused(*args(n=100)) == L.HS_ERR_SHAPE
    assert lib.hs_adamw_fused(*args(n=0)) == L.HS_ERR_SHAPE
    assert lib.hs_adamw_fused(*args(st=None)) == L.HS_ERR_ARG
    assert lib.hs_adamw_fused(*args(mx=1.0)) == L.HS_ERR_ARG                             # clipping without the gradient norm
    assert lib.hs_adamw_fused(*args(gs=0.0)) == L.HS_ERR_ARG
    assert lib.hs_adamw_fused(*args(p=P + 4)) == L.HS_ERR_SHAPE
    assert lib.hs_adamw_fused(*args(phase=3)) == L.HS_ERR_ARG
    assert lib.hs_adamw_fused(*args(phase=L.ADAMW_PREPARE, mx=1.0)) == L.HS_ERR_ARG      # the prologue owns the clip arguments ...
    assert lib.hs_adamw_fused(*args(phase=L.ADAMW_SWEEP, n=100)) == L.HS_ERR_SHAPE       # ... a sweep the range
    # the ranged forward: a proper sub-range of the 16-bit stack needs caller-owned weight copies
    tbl2 = (C.c_void_p * 64)(*([FAKE] * 64))
    v2 = L.VitDesc(1, 33, 768, 12, 3072, 2, L.PREC_BF16, 1, 0.0, None)
    nb = lib.hs_vit_workspace_bytes(C.byref(v2))
    assert lib.hs_vit_blocks_fwd_range(C.byref(v2), FAKE, tbl2, tbl2, FAKE, FAKE, 1 << 20, nb, 1, 2, None) == L.HS_ERR_ARG
    assert lib.hs_vit_blocks_fwd_range(C.byref(v2), FAKE, tbl2, tbl2, FAKE, FAKE, 1 << 20, nb, 0, 3, None) == L.HS_ERR_SHAPE
    assert lib.hs_vit_blocks_fwd_range(C.byref(v2), FAKE, tbl2, tbl2, FAKE, FAKE, 1 << 20, nb, 1, 1, None) == L.HS_ERR_SHAPE
    assert lib.hs_vit_blocks_fwd_range(C.byref(v2), None, tbl2, tbl2, FAKE, FAKE, 1 << 20, nb, 0, 2, None) == L.HS_ERR_ARG


def test_debug_selector_is_inert_in_the_product_build(capfd):
    """hs_debug_set: the product build accepts only the result-preserving selector (key 0, value 0 / 1); the timing-only kernel
    variants exist only in -DHS_DEBUG_HOOKS builds and a request for one is refused with a warning."""
    lib = L.load()
    lib.hs_debug_set(3, 2)          # 'one K-tile only' GEMM: would truncate every GEMM if it were honoured
    lib.hs_debug_set(0, 2)          # 'no softmax fix-up pass'
    err = capfd.readouterr().err
    assert "ignored" in err and "HS_DEBUG_HOOKS" in err
    lib.hs_debug_set(0, 1)
    lib.hs_debug_set(0, 0)
    lib.hs_debug_set(3, 0)
    assert "ignored" not in capfd.readouterr().err
