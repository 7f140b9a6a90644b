#!/usr/bin/env python3
"""time the optimizer sweep alone on the flat buffers of BASELINE configs[2] (205 M fp32 parameters incl. padding): hs_adamw_masked + the
memset + the two cast passes it used to need (round 5) against hs_adamw_fused (round 6).  python tools/bench_adamw.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hsenet_amd import _lib as L

dev = torch.device("cuda:0")
n = 64 * (205_000_000 // 64)
p, g, m, v = (torch.randn(n, device=dev) * 0.02 for _ in range(4))
v.abs_()
mask = torch.ones(n // 64, dtype=torch.uint8, device=dev)
gsq, scratch = torch.zeros(1, device=dev), torch.zeros(1024, device=dev)
state = torch.zeros(8, dtype=torch.int32, device=dev)
w16 = torch.empty(n, dtype=torch.bfloat16, device=dev)
nmap = int(0.83 * (n // 64))                                    # ~170 M of the 205 M are block weights with a 16-bit copy
table = torch.zeros(n // 64, dtype=torch.int64)
table[:nmap] = w16.data_ptr() + 128 * torch.arange(nmap)
table = table.to(dev)
s = torch.cuda.current_stream().cuda_stream


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def masked():
    L.call("hs_adamw_masked", p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), mask.data_ptr(), n, 1e-4, 0.9, 0.999, 1e-8, 0.1, 1.0,
           gsq.data_ptr(), 1.0, 5, s)


def fused(zero=1, tab=True):
    L.call("hs_adamw_fused", p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), mask.data_ptr(), n, 1e-4, 0.9, 0.999, 1e-8, 0.1, 1.0,
           gsq.data_ptr(), 1.0, zero, table.data_ptr() if tab else None, 0, state.data_ptr(), L.ADAMW_STEP, s)


L.call("hs_sumsq", g.data_ptr(), gsq.data_ptr(), scratch.data_ptr(), n, s)
print(f"n = {n / 1e6:.0f} M elements")
a = t(masked)
print(f"hs_adamw_masked                         {a:.3f} ms  ({n * 28 / a / 1e6:.0f} GB/s of 28 B/elt)")
z = t(lambda: g.zero_())
print(f"grad.zero_()                            {z:.3f} ms")
c = t(lambda: L.call("hs_cast_bf16", p.data_ptr(), w16.data_ptr(), int(0.83 * n), s))
print(f"hs_cast_bf16 of the block weights       {c:.3f} ms")
print(f"round-5 total (AdamW + memset + cast)   {a + z + c:.3f} ms")
g.normal_()
b0 = t(lambda: fused(0, False))
print(f"hs_adamw_fused (no zero, no copies)     {b0:.3f} ms")
b1 = t(lambda: fused(1, False))
print(f"hs_adamw_fused (zero_grad)              {b1:.3f} ms")
b2 = t(lambda: fused(1, True))
print(f"hs_adamw_fused (zero_grad + 16-bit)     {b2:.3f} ms  ({n * (32 + 2 * 0.83) / b2 / 1e6:.0f} GB/s of 33.7 B/elt)")
