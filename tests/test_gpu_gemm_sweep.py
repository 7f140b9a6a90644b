"""Seeded randomized sweep of hs_gemm (bf16) against an fp64 statement: every layout x every epilogue flag set the product launches x
ragged and tile-aligned shapes, small-tile and 256x256-tile dispatch, batched, scaled columns, fused column sums.  The hand-picked
cases in test_gpu_kernels.py pin the shapes of the encoder layer; this sweep is for the dispatch logic around them (tile choice,
compile-time vs generic epilogues, the 16x16x32 NT kernel's four flag sets, tails handled by the buffer range check)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _case(rng, big):
    if big:        # enough 256x256 tiles to select the big tile (2 * tiles > CUs), K a multiple of 64
        M = int(rng.integers(30, 70)) * 256 + int(rng.choice([0, 0, 8, 20, 136, 255 - 7]))
        N = int(rng.choice([256, 512, 768, 1024, 776, 264]))
        K = int(rng.choice([64, 128, 192, 320]))
    else:
        M = int(rng.integers(1, 700))
        N = 8 * int(rng.integers(1, 60))
        K = 8 * int(rng.integers(1, 50))
    return M, N, K


FLAG_SETS = ["none", "bias_resid", "bias_gelu_save", "dgelu_colsum", "dgelu", "bias", "gelu_bwd_z", "scale_cols", "atomic"]


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("big", [False, True])
def test_gemm_random_sweep(dev, seed, big):
    from hsenet_amd import functional as Fn, _lib as L
    rng = np.random.default_rng(1000 + 17 * seed + int(big))
    s = torch.cuda.current_stream().cuda_stream
    for it in range(6 if big else 14):
        M, N, K = _case(rng, big)
        layout = int(rng.integers(0, 3))
        fs = FLAG_SETS[int(rng.integers(0, len(FLAG_SETS)))]
        if layout == 2:
            M = max(8, M // 8 * 8)                                   # TN: 16-byte chunks along A's contiguous axis (M)
            if fs != "atomic":
                fs = "none"
        if fs == "atomic" and layout != 2:
            fs = "none"
        g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
        a = torch.randn(M, K, generator=g).bfloat16()
        b = torch.randn(N, K, generator=g).bfloat16()
        acc = a.double() @ b.double().t()
        A = (a if layout != 2 else a.t().contiguous()).to(dev)
        Bm = (b if layout == 0 else b.t().contiguous()).to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        resid = torch.randn(M, N, generator=g).to(dev)
        aux = (0.3 * torch.randn(M, N, generator=g)).bfloat16().to(dev)
        tag = f"seed {seed} it {it}: M={M} N={N} K={K} layout={layout} {fs}"
        o32 = torch.full((M, N), float("nan"), device=dev)
        o16 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
        if fs == "none":
            Fn.gemm(A, Bm, layout, prec=L.PREC_BF16, out_bf16=o16, alpha=0.5)
            assert rel(o16.float(), 0.5 * acc) < 4e-3, tag
        elif fs == "bias":
            Fn.gemm(A, Bm, layout, prec=L.PREC_BF16, out_f32=o32, bias=bias, flags=L.EPI_BIAS)
            assert rel(o32, acc + bias.double().cpu()) < 1e-5, tag
        elif fs == "bias_resid":
            Fn.gemm(A, Bm, layout, prec=L.PREC_BF16, out_f32=o32, bias=bias, resid=resid, flags=L.EPI_BIAS | L.EPI_RESID)
            assert rel(o32, acc + bias.double().cpu() + resid.double().cpu()) < 1e-5, tag
        elif fs == "bias_gelu_save":
            g16 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
            Fn.gemm(A, Bm, layout, prec=L.PREC_BF16, out_bf16=o16, bias=bias, aux_out=g16, flags=L.EPI_BIAS | L.EPI_GELU | L.EPI_SAVE_DGELU)
            z = (acc + bias.double().cpu()).float().bfloat16().float()
            zq = z.double().requires_grad_()
            (gq,) = torch.autograd.grad(F.gelu(zq).sum(), zq)
            assert rel(o16.float(), F.gelu(z)) < 4e-3 and rel(g16.float(), gq) < 4e-3, tag
            assert not torch.isnan(o16.float()).any() and not torch.isnan(g16.float()).any(), tag
        elif fs in ("dgelu", "dgelu_colsum"):
            cs = torch.zeros(N, device=dev) if fs == "dgelu_colsum" else None
            Fn.gemm(A, Bm, layout, prec=L.PREC_BF16, out_bf16=o16, aux=aux, flags=L.EPI_GELU_BWD | L.EPI_SAVE_DGELU, colsum_out=cs)
            want = acc * aux.double().cpu()
            assert rel(o16.float(), want) < 4e-3, tag
            if cs is not None:
                assert rel(cs, want.sum(0)) < 1e-4, tag
        elif fs == "gelu_bwd_z":
            Fn.gemm(A, Bm, layout, prec=L.PREC_BF16, out_f32=o32, aux=aux, flags=L.EPI_GELU_BWD)
            zr = aux.float().double().cpu().requires_grad_()
            (gz,) = torch.autograd.grad(F.gelu(zr).sum(), zr)
            assert rel(o32, acc * gz) < 1e-4, tag
        elif fs == "scale_cols":
            sc = min(N, 8 * int(rng.integers(1, 9)))
            sc -= sc % 4
            Fn.gemm(A, Bm, layout, prec=L.PREC_BF16, out_bf16=o16, scale_cols=sc, scale_cols_alpha=0.25)
            want = acc.clone()
            want[:, :sc] *= 0.25
            assert rel(o16.float(), want) < 4e-3, tag
        elif fs == "atomic":
            o32.zero_()
            split = int(rng.integers(1, 5))
            Fn.gemm(A, Bm, layout, prec=L.PREC_BF16, out_f32=o32, flags=L.EPI_ATOMIC, split_k=min(split, (K + 63) // 64))
            assert rel(o32, acc) < 1e-5, tag
    torch.cuda.synchronize()


def test_gemm_batched_strided_bf16(dev):
    """batched NT with an output batch stride (the packers' output GEMM writes straight into its half of the concatenated tensor)."""
    from hsenet_amd import _lib as L
    import ctypes as C
    Bt, M, N, K = 3, 128, 256, 192
    g = torch.Generator().manual_seed(5)
    a = torch.randn(Bt, M, K, generator=g).bfloat16().to(dev)
    w = torch.randn(N, K, generator=g).bfloat16().to(dev)
    out = torch.full((Bt, 2, M, N), float("nan"), device=dev)             # slot 1 of 2 is written, slot 0 stays NaN
    ga = L.GemmArgs()
    ga.A, ga.B, ga.M, ga.N, ga.K, ga.lda, ga.ldb, ga.batch = a.data_ptr(), w.data_ptr(), M, N, K, K, K, Bt
    ga.strideA, ga.strideB, ga.strideC = M * K, 0, 2 * M * N
    ga.out_f32, ga.ldc, ga.alpha, ga.layout, ga.split_k = out.data_ptr() + 4 * M * N, N, 1.0, L.GEMM_NT, 1
    L.call("hs_gemm", C.byref(ga), L.PREC_BF16, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want = a.double().cpu() @ w.double().cpu().t()
    assert rel(out[:, 1], want) < 1e-5 and torch.isnan(out[:, 0]).all()


def test_dgelu_column_sums_ignore_rows_outside_the_matrix(dev):
    """the 16x16x32 NT kernel's x gelu' epilogue with fused column sums (fc2's dgrad) on a RAGGED M: lanes of rows past M load the aux
    words of row 0.  With NaN / Inf planted there the products of the real rows must still be right and the column sums finite where the
    real aux column is finite — out-of-matrix rows are dropped by a select, not by a multiply with 0 (NaN * 0 = NaN)."""
    from hsenet_amd import functional as Fn, _lib as L
    M, N, K = 33 * 256 + 20, 768, 192                                  # 34 x 3 tiles of 256x256 > CUs / 2: the big tile, ragged last row tile
    g = torch.Generator().manual_seed(77)
    a = torch.randn(M, K, generator=g).bfloat16()
    b = torch.randn(N, K, generator=g).bfloat16()
    aux = (0.3 * torch.randn(M, N, generator=g)).bfloat16()
    aux[0, :8] = float("nan")                                          # row 0, first 8 columns: what an out-of-range lane reads
    aux[0, 8:16] = float("inf")
    cs = torch.zeros(N, device=dev)
    o16 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    Fn.gemm(a.to(dev), b.to(dev), L.GEMM_NT, prec=L.PREC_BF16, out_bf16=o16, aux=aux.to(dev), flags=L.EPI_GELU_BWD | L.EPI_SAVE_DGELU,
            colsum_out=cs)
    want = (a.double() @ b.double().t()) * aux.double()
    assert rel(o16[1:].float(), want[1:]) < 4e-3
    # columns 16.. : row 0's aux is finite there -> exact column sums; columns 0..15 hold row 0's own NaN / Inf and nothing else is expected
    assert torch.isfinite(cs[16:]).all() and rel(cs[16:], want[:, 16:].sum(0)) < 1e-4


HALF_SHAPES = [(4097, 768, 768), (4097, 768, 3072), (4097, 768, 2304), (130, 136, 64), (1, 8, 64), (129, 776, 128), (640, 264, 192), (2049, 768, 768)]


@pytest.mark.parametrize("prec", ["bf16", "f16"])
@pytest.mark.parametrize("fs", ["none", "bias_resid", "bias", "dgelu_colsum", "atomic_split"])
def test_gemm_half_tile_shapes(dev, prec, fs):
    """The 128x128 tile of the NT kernel (gemm.hip half_mainloop; hs_gemm picks it when K % 64 == 0 and the 256x128 tiles would fill at
    most half of the CUs — every shape here, among them the encoder layer's N = 768 outputs at one 256³ / native volume per GPU): its
    three instantiations (plain, bias + residual, generic epilogue), ragged M / N, one K-tile, fewer K-tiles than ring stages, both
    16-bit operand formats; against an fp64 statement."""
    from hsenet_amd import functional as Fn, _lib as L
    P = L.PREC_BF16 if prec == "bf16" else L.PREC_F16
    dt = torch.bfloat16 if prec == "bf16" else torch.float16
    tol16 = 4e-3 if prec == "bf16" else 6e-4
    for i, (M, N, K) in enumerate(HALF_SHAPES):
        g = torch.Generator().manual_seed(900 + i)
        a = torch.randn(M, K, generator=g).to(dt)
        b = (torch.randn(N, K, generator=g) * (0.05 if prec == "f16" else 1.0)).to(dt)
        acc = a.double() @ b.double().t()
        A, Bm = a.to(dev), b.to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        resid = torch.randn(M, N, generator=g).to(dev)
        aux = (0.3 * torch.randn(M, N, generator=g)).to(dt).to(dev)
        tag = f"{prec} {fs}: M={M} N={N} K={K}"
        o32 = torch.full((M, N), float("nan"), device=dev)
        o16 = torch.full((M, N), float("nan"), dtype=dt, device=dev)
        if fs == "none":
            Fn.gemm(A, Bm, L.GEMM_NT, prec=P, out_bf16=o16, alpha=0.5)
            assert rel(o16.float(), 0.5 * acc) < tol16, tag
        elif fs == "bias":
            Fn.gemm(A, Bm, L.GEMM_NT, prec=P, out_f32=o32, bias=bias, flags=L.EPI_BIAS)
            assert rel(o32, acc + bias.double().cpu()) < 1e-5, tag
        elif fs == "bias_resid":
            Fn.gemm(A, Bm, L.GEMM_NT, prec=P, out_f32=o32, bias=bias, resid=resid, flags=L.EPI_BIAS | L.EPI_RESID)
            assert rel(o32, acc + bias.double().cpu() + resid.double().cpu()) < 1e-5, tag
        elif fs == "atomic_split":          # split-K slices of the 128x128 tile accumulating with float atomics (K-tile ranges kt0..kt1 of the ring)
            o32.zero_()
            Fn.gemm(A, Bm, L.GEMM_NT, prec=P, out_f32=o32, flags=L.EPI_ATOMIC, split_k=min(3, K // 64))
            assert rel(o32, acc) < 1e-5, tag
        else:
            cs = torch.zeros(N, device=dev)
            Fn.gemm(A, Bm, L.GEMM_NT, prec=P, out_bf16=o16, aux=aux, flags=L.EPI_GELU_BWD | L.EPI_SAVE_DGELU, colsum_out=cs)
            want = acc * aux.double().cpu()
            assert rel(o16.float(), want) < tol16, tag
            assert rel(cs, want.sum(0)) < 1e-4, tag
    torch.cuda.synchronize()


_TILE_PROBE = r"""
import hashlib, torch
from hsenet_amd import functional as Fn, _lib as L
dev = torch.device("cuda:0")
h = hashlib.sha256()
for i, (M, N, K) in enumerate([(4097, 768, 768), (4097, 768, 3072), (129, 776, 128)]):
    g = torch.Generator().manual_seed(4200 + i)
    a = torch.randn(M, K, generator=g).bfloat16().to(dev)
    b = torch.randn(N, K, generator=g).bfloat16().to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    resid = torch.randn(M, N, generator=g).to(dev)
    o32 = torch.empty(M, N, device=dev)
    o16 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    Fn.gemm(a, b, L.GEMM_NT, prec=L.PREC_BF16, out_f32=o32, bias=bias, resid=resid, flags=L.EPI_BIAS | L.EPI_RESID)
    Fn.gemm(a, b, L.GEMM_NT, prec=L.PREC_BF16, out_bf16=o16)
    torch.cuda.synchronize()
    h.update(o32.cpu().numpy().tobytes()); h.update(o16.view(torch.int16).cpu().numpy().tobytes())
print("TILE_PROBE", h.hexdigest())
"""


def test_half_tile_is_bit_identical_to_the_256x128_tile(dev):
    """gemm.hip claims that the tile shape never changes a result (every output element accumulates its k-steps in the same order with the
    same MFMA): the same seeded products in two child processes, one forced to the 256x128 tile (HS_GEMM_TILE=s, read once at library
    load), one to the 128x128 tile (=h), must hash equal."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = []
    for tile in ("s", "h"):
        env = dict(os.environ, HS_GEMM_TILE=tile, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        out = subprocess.run([sys.executable, "-c", _TILE_PROBE], env=env, capture_output=True, text=True, timeout=300, cwd=root)
        assert out.returncode == 0, out.stderr[-2000:]
        digests.append([ln.split()[1] for ln in out.stdout.splitlines() if ln.startswith("TILE_PROBE")][0])
    assert digests[0] == digests[1]
