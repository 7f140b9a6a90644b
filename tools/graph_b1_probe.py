"""Is the small-batch step launch-bound?  Forward + backward of the dual engine (two streams, ~1000 launches) captured into ONE hipGraph and
replayed, against the eager enqueue, at 1 / 2 / 4 volumes per GPU (eval(): dropout seeds are kernel arguments and would be baked into the
graph).  Measured (round 3): 15.73 -> 15.60 ms (+0.8 %), 25.43 -> 25.31 (+0.5 %), 45.79 -> 45.54 (+0.5 %): no.  python tools/graph_b1_probe.py"""
import sys, time, torch
sys.path.insert(0, '/root/repo')
import bench as Bn
import hsenet_amd as H
from hsenet_amd import _lib as L, functional as Fn
dev = torch.device('cuda:0')
geom = Bn.GEOM['dual256']
for B in (1, 2, 4):
    eng, pe, _ = Bn.build_engine('dual256', geom, B, dev)
    for m in (eng.tower, eng.p1.packer, eng.p2.packer): m.eval()
    import numpy as np
    rng = np.random.default_rng(0)
    vol = torch.from_numpy(rng.random((B, 1) + geom['img'], dtype=np.float32)).to(dev)
    s2d = torch.from_numpy(rng.standard_normal((B, 32, 768), dtype=np.float32)).to(dev)
    def fb():
        out = eng.forward(vol, s2d)
        eng.loss.zero_()
        L.call("hs_mse_grad", out.data_ptr(), eng.dout.data_ptr(), eng.loss.data_ptr(), out.numel(), Fn._s())
        eng.backward(eng.dout)
    def timeit(fn, n=30):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    def eager(): fb(); eng.optimizer_step()
    te = timeit(eager)
    g = torch.cuda.CUDAGraph()
    fb(); torch.cuda.synchronize()
    with torch.cuda.graph(g): fb()
    def graphed(): g.replay(); eng.optimizer_step()
    tg = timeit(graphed)
    print(f"B={B}: eager {te:.3f} ms/step  graph {tg:.3f} ms/step  ({100*(te/tg-1):+.1f} %)", flush=True)
    del eng, g
    torch.cuda.empty_cache()
