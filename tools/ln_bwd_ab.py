"""A/B of the two LayerNorm-backward variants (4 waves x one row per trip vs. the 8-wave "few rows" variant) over the row counts of
1..8 volumes per GPU, on a hook build:

    bash tools/build_variant.sh hooks norm.hip
    HS_LIB_PATH=hsenet_amd/lib/variants/libhsenet_hooks.so python tools/ln_bwd_ab.py

Prints, per row count: time of each variant (HIP events, median of 5 x 50 launches), what the product's selection rule picks, and
whether dx of the two variants is bit-identical (it must be: the per-row arithmetic is the same, only the column-sum atomics differ
in order)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hsenet_amd import _lib as L  # noqa: E402


def run(rows, d, variant, dev, reps=50):
    lib = L.load()
    g = torch.Generator(device="cpu").manual_seed(rows)
    x = torch.randn(rows, d, generator=g).to(dev)
    dy = torch.randn(rows, d, generator=g).to(dev).bfloat16()
    din = torch.randn(rows, d, generator=g).to(dev)
    gamma = torch.randn(d, generator=g).to(dev)
    mean, rstd = x.mean(1).contiguous(), (x.var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
    dx = torch.empty_like(x)
    dg, db, ds = (torch.zeros(d, device=dev) for _ in range(3))
    s = torch.cuda.current_stream().cuda_stream
    lib.hs_debug_set(5, variant)

    def once():
        L.call("hs_layernorm_bwd", dy.data_ptr(), 1, x.data_ptr(), None, gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
               din.data_ptr(), dx.data_ptr(), None, dg.data_ptr(), db.data_ptr(), ds.data_ptr(), rows, d, s)
    for _ in range(5):
        once()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            once()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / reps)
    lib.hs_debug_set(5, 0)
    return sorted(ts)[2], dx.clone()


def product_rule(rows, d):
    """mirror of hsenet_amd/csrc/norm.hip hs_layernorm_bwd's selection (`bool wide = ...`): keep the two in step."""
    return rows // 32 >= 128 and rows <= 10 * 4100 and d <= 768


def main():
    dev = torch.device("cuda:0")
    d = 768
    print("rows      narrow_us   wide_us   wide/narrow   product picks   dx bit-identical")
    for vols in (1, 2, 3, 4, 5, 6, 7, 8, 12, 20):
        rows = 4097 * vols
        tn, dxn = run(rows, d, 1, dev)
        tw, dxw = run(rows, d, 2, dev)
        print(f"{rows:7d}  {tn:9.1f}  {tw:9.1f}  {tw / tn:9.3f}      {'wide' if product_rule(rows, d) else 'narrow':6s}          {torch.equal(dxn, dxw)}",
              flush=True)


if __name__ == "__main__":
    main()
