"""One-off consistency check of the default bench batch: the dual engine's forward output and parameter gradients for volume 0 at 53
volumes per GPU against the same volume alone (1 volume per GPU).  Row tiles of a GEMM and (batch, head) blocks of the attention are
independent of the batch size, so the forward must agree bit for bit; the gradients of the 53-volume step are sums over volumes, so the
check for the backward is linearity: grad(batch of 53 copies of one volume, loss = mean-square) == grad(1 volume) up to fp32 summation
order.  python tools/big_batch_check.py
Measured (round 3, 53 volumes): forward of volume 0 and of volume 52 bit-identical to the 1-volume run; loss 0.03957957 vs 0.03957951 (the
mean over 53x the elements); gradient relative difference 4.6e-4."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as Bn  # noqa: E402
from hsenet_amd import _lib as L, functional as Fn  # noqa: E402


def run(B, vol1, s1, dev):
    geom = Bn.GEOM["dual256"]
    eng, _, _ = Bn.build_engine("dual256", geom, B, dev)
    for m in (eng.tower, eng.p1.packer, eng.p2.packer):
        m.eval()
    vol = vol1.expand(B, -1, -1, -1, -1).contiguous()
    s2d = s1.expand(B, -1, -1).contiguous()
    out = eng.forward(vol, s2d)
    eng.loss.zero_()
    L.call("hs_mse_grad", out.data_ptr(), eng.dout.data_ptr(), eng.loss.data_ptr(), out.numel(), Fn._s())
    eng.backward(eng.dout)
    torch.cuda.synchronize()
    res = out[0].clone(), out[B - 1].clone(), eng.grad.clone(), float(eng.loss)
    del eng
    torch.cuda.empty_cache()
    return res


def main():
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(7)
    vol1 = torch.from_numpy(rng.random((1, 1, 256, 256, 256), dtype=np.float32)).to(dev)
    s1 = torch.from_numpy(rng.standard_normal((1, 32, 768), dtype=np.float32)).to(dev)
    o1, _, g1, l1 = run(1, vol1, s1, dev)
    B = int(sys.argv[1]) if len(sys.argv) > 1 else Bn.DEFAULT_BATCH
    oB0, oBl, gB, lB = run(B, vol1, s1, dev)
    print(f"forward, volume 0 of {B} == alone: {torch.equal(oB0, o1)}; last volume == alone: {torch.equal(oBl, o1)}")
    print(f"loss {lB:.8f} vs {l1:.8f}")
    # mean-square loss over B identical volumes: d/dtheta is the 1-volume gradient (each volume contributes 1/B of it)
    rel = float((gB - g1).norm() / g1.norm())
    print(f"gradient of {B} copies vs 1 volume: relative difference {rel:.3e}")
    ok = torch.equal(oB0, o1) and torch.equal(oBl, o1) and abs(lB - l1) < 1e-5 * abs(l1) and rel < 2e-3
    print("OK" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
