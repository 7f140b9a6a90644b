"""Hook-build test of the retired one-sweep attention backward (NOT collected by the product suite: run it by path on a GPU box after
`bash tools/experiments/attn_bwd_one_sweep/build.sh`):  python -m pytest tools/experiments/attn_bwd_one_sweep/test_one_sweep.py -q
The kernel's results against the shipped two sweeps and an fp32 statement, twice on one workspace (launch epochs), bitwise reproducible."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", ".."))
LIB = os.path.join(HERE, "libhsenet_one_sweep.so")
ONE_SWEEP, TILES = 4, 8


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("B,N,H,tiles", [(2, 256, 2, False), (3, 513, 2, False), (2, 513, 12, True), (2, 2049, 12, False), (2, 4097, 12, False)])
def test_attn_bwd_one_sweep(B, N, H, tiles):
    if not torch.cuda.is_available() or not os.path.exists(LIB):
        pytest.skip("needs an MI355X and build.sh's library")
    from hsenet_amd import functional as Fn, _lib as L
    lib = C.CDLL(LIB)
    lib.hs_attn_bwd_ws_bytes.restype = C.c_size_t
    dev = torch.device("cuda:0")
    g = np.random.default_rng(140 + N)
    qkv = torch.from_numpy(g.standard_normal((B, N, 3 * H * 64), dtype=np.float32) * 1.1).to(dev).bfloat16()
    d_o = torch.from_numpy(g.standard_normal((B, N, H * 64), dtype=np.float32)).to(dev).bfloat16()
    o, lse = Fn.attn_fwd(qkv, H, L.PREC_BF16)
    two = Fn.attn_bwd(qkv, o, d_o, lse, H)
    flags = ONE_SWEEP | (TILES if tiles else 0)
    need = lib.hs_attn_bwd_ws_bytes(B, N, H, flags)
    assert need > 2 * B * H * N * 4
    ws = torch.zeros(need, dtype=torch.uint8, device=dev)
    outs = []
    for _ in range(2):
        dqkv = torch.empty_like(qkv)
        vp = C.c_void_p
        rc = lib.hs_attn_bwd2(vp(qkv.data_ptr()), vp(o.data_ptr()), vp(d_o.data_ptr()), vp(lse.data_ptr()), vp(dqkv.data_ptr()), vp(ws.data_ptr()),
                              C.c_size_t(need), B, N, H, 64, flags, vp(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        assert lib.hs_attn_bwd_ws_error(vp(ws.data_ptr()), B, N, H, flags, vp(torch.cuda.current_stream().cuda_stream)) == 0
        outs.append(dqkv)
    one, again = outs
    assert torch.equal(one, again)                                   # fixed summation order: no atomics, no run-to-run difference
    D = H * 64
    x = qkv.float().requires_grad_()
    q, k, v = (t.view(B, N, H, 64).permute(0, 2, 1, 3) for t in x.split(D, dim=-1))
    ro = (torch.softmax(q @ k.transpose(-1, -2) / 8.0, -1) @ v).permute(0, 2, 1, 3).reshape(B, N, D)
    (gr,) = torch.autograd.grad((ro * d_o.float()).sum(), x)
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        assert rel(one[..., sl].float(), gr[..., sl]) < 6e-3, name    # measured 3.1e-3 (P, dS, outputs rounded to bf16)
        assert rel(one[..., sl].float(), two[..., sl].float()) < (8e-3 if name == "dq" else 1e-3), name
