"""micro-bench of the bf16 GEMM shapes of one ViT-B layer (B volumes); prints us and TFLOP/s per shape."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hsenet_amd import functional as Fn, _lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
only = sys.argv[3] if len(sys.argv) > 3 else ""
dev = torch.device("cuda:0")
M = B * 4097
shapes = [("qkv NT", L.GEMM_NT, M, 2304, 768), ("proj NT", L.GEMM_NT, M, 768, 768), ("fc1 NT", L.GEMM_NT, M, 3072, 768),
          ("fc2 NT", L.GEMM_NT, M, 768, 3072), ("dgrad fc2 NN", L.GEMM_NN, M, 3072, 768), ("dgrad fc1 NN", L.GEMM_NN, M, 768, 3072),
          ("wgrad fc1 TN", L.GEMM_TN, 3072, 768, M), ("wgrad qkv TN", L.GEMM_TN, 2304, 768, M), ("wgrad proj TN", L.GEMM_TN, 768, 768, M),
          ("big NT 8192^3", L.GEMM_NT, 8192, 8192, 8192)]
for name, lay, m, n, k in shapes:
    if only and only not in name:
        continue
    a = torch.randn((m, k) if lay != L.GEMM_TN else (k, m), device=dev).bfloat16()
    b = torch.randn((n, k) if lay == L.GEMM_NT else (k, n), device=dev).bfloat16()
    tn = lay == L.GEMM_TN
    out = torch.zeros(m, n, device=dev, dtype=torch.float32 if tn else torch.bfloat16)
    kw = dict(out_f32=out, flags=L.EPI_ATOMIC, split_k=Fn._wgrad_split(m, n, k)) if tn else dict(out_bf16=out)
    for _ in range(3):
        Fn.gemm(a, b, lay, prec=L.PREC_BF16, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        Fn.gemm(a, b, lay, prec=L.PREC_BF16, **kw)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"{name:16s} M={m} N={n} K={k}: {us:8.1f} us  {2.0*m*n*k/us/1e6:7.1f} TFLOP/s", flush=True)
if only == "TNvsNT":
    m, n, k = 3072, 768, 40970 // 8 * 8
    for lay, name in ((L.GEMM_NT, "NT"), (L.GEMM_NN, "NN"), (L.GEMM_TN, "TN")):
        for split in (1, 7):
            a = torch.randn((m, k) if lay != L.GEMM_TN else (k, m), device=dev).bfloat16()
            b = torch.randn((n, k) if lay == L.GEMM_NT else (k, n), device=dev).bfloat16()
            out = torch.zeros(m, n, device=dev)
            kw = dict(out_f32=out, flags=L.EPI_ATOMIC, split_k=split)
            for _ in range(3): Fn.gemm(a, b, lay, prec=L.PREC_BF16, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): Fn.gemm(a, b, lay, prec=L.PREC_BF16, **kw)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / reps
            print(f"{name} split {split}: {us:8.1f} us  {2.0*m*n*k/us/1e6:7.1f} TFLOP/s", flush=True)
