"""hipBLASLt (torch.matmul) on the ViT-B layer's GEMM shapes: the library yardstick beside tools/kbench gemm (diagnostic only)."""
import sys, torch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 20
M = B * 4097
dev = "cuda"
def t(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for name, m, n, k, lay in [("qkv NT", M, 2304, 768, "nt"), ("proj NT", M, 768, 768, "nt"), ("fc1 NT", M, 3072, 768, "nt"), ("fc2 NT", M, 768, 3072, "nt"),
                           ("dgrad fc2 NN", M, 3072, 768, "nn"), ("dgrad fc1 NN", M, 768, 3072, "nn"), ("wgrad fc1 TN", 3072, 768, M, "tn"),
                           ("sq NT 8192", 8192, 8192, 8192, "nt")]:
    if lay == "nt":
        a = torch.randn(m, k, device=dev, dtype=torch.bfloat16); b = torch.randn(n, k, device=dev, dtype=torch.bfloat16)
        f = lambda: torch.matmul(a, b.t())
    elif lay == "nn":
        a = torch.randn(m, k, device=dev, dtype=torch.bfloat16); b = torch.randn(k, n, device=dev, dtype=torch.bfloat16)
        f = lambda: torch.matmul(a, b)
    else:
        a = torch.randn(k, m, device=dev, dtype=torch.bfloat16); b = torch.randn(k, n, device=dev, dtype=torch.bfloat16)
        f = lambda: torch.matmul(a.t(), b)
    us = t(f)
    print(f"{name:16s} M={m:6d} N={n:5d} K={k:6d}: {us:8.1f} us  {2.0*m*n*k/us/1e6:7.1f} TFLOP/s (hipBLASLt)", flush=True)
