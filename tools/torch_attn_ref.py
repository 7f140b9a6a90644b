"""torch SDPA (whatever flash backend this ROCm build ships) on the encoder's attention shape: a yardstick beside tools/kbench attnf/attnb."""
import sys, torch, torch.nn.functional as F
B = int(sys.argv[1]) if len(sys.argv) > 1 else 20
H, N, D = 12, 4097, 64
q, k, v = (torch.randn(B, H, N, D, device="cuda", dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
def t(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
from torch.nn.attention import sdpa_kernel, SDPBackend
for be in (SDPBackend.FLASH_ATTENTION, SDPBackend.EFFICIENT_ATTENTION):
    try:
        with sdpa_kernel(be):
            fw = t(lambda: F.scaled_dot_product_attention(q, k, v))
            o = F.scaled_dot_product_attention(q, k, v); g = torch.randn_like(o)
            bw = t(lambda: torch.autograd.grad(o, (q, k, v), g, retain_graph=True))
        fl = 4.0 * B * H * N * N * D
        print(f"{be}: fwd {fw:8.1f} us {fl/fw/1e6:7.1f} TF | bwd {bw:8.1f} us {2.5*fl/bw/1e6:7.1f} TF (10 N^2 d)", flush=True)
    except Exception as e:
        print(f"{be}: unavailable ({type(e).__name__}: {str(e)[:120]})", flush=True)
