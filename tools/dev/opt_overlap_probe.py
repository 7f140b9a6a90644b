"""engine step rate at small batches with the optimizer pass synchronous vs overlapped (one process, interleaved)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench_module_api as M
from hsenet_amd.engine import DualEngine
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
if os.environ.get("LOWPRI"):
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    torch.cuda.init(); torch.zeros(1, device=dev)
    least, greatest = ctypes.c_int(), ctypes.c_int()
    print("priority range rc", hip.hipDeviceGetStreamPriorityRange(ctypes.byref(least), ctypes.byref(greatest)), least.value, greatest.value)
    from hsenet_amd import fused, engine
    made = []
    def low_stream(device):
        if not made:
            # a few candidates so that one lands on a hardware queue of its own
            cands = []
            for _ in range(4):
                h = ctypes.c_void_p()
                rc = hip.hipStreamCreateWithPriority(ctypes.byref(h), 1, least.value)     # hipStreamNonBlocking
                assert rc == 0, rc
                cands.append(torch.cuda.ExternalStream(h.value, device=dev))
            import time
            def timed(others):
                torch.cuda.synchronize(); t0 = time.perf_counter(); torch.cuda._sleep(2_000_000)
                for o in others:
                    with torch.cuda.stream(o): torch.cuda._sleep(2_000_000)
                torch.cuda.synchronize(); return time.perf_counter() - t0
            timed(()); one = min(timed(()) for _ in range(2))
            side = fused.side_stream(dev)
            pick = None
            for c in cands:
                t = min(timed((side, c)) for _ in range(2))
                print("cand", t / one)
                if t < 1.5 * one and pick is None: pick = c
            made.append(pick or cands[0])
        return made[0]
    engine.optimizer_stream = low_stream
for b in [int(x) for x in sys.argv[1:]] or [1, 3]:
    vols, slices = M.synthetic_inputs(M.GEOM, b, dev)
    engs = {}
    for ov in (False, True):
        t, p1, p2 = M.build_modules(M.GEOM, dev)
        engs[ov] = DualEngine(t, p1, p2, b, lr=1e-4, weight_decay=0.1, overlap_optimizer=ov)
    steps = max(10, 40 // b)
    for rnd in range(3):
        for ov in (False, True):
            dt = M.time_steps(lambda v, s: engs[ov].train_step(v, s), vols, slices, steps, 3)
            print(f"batch {b} overlap={ov}: {b * steps / dt:.2f} volumes/s  {1e3 * dt / steps:.3f} ms", flush=True)
    del engs
    torch.cuda.empty_cache()
