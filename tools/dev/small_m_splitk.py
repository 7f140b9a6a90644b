"""one volume per GPU: the K = 3072 / 2304 / 768, N = 768 NT GEMMs (102 tiles for 256 CUs) with split-K (atomic fp32 epilogue): what a split would buy"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hsenet_amd import functional as Fn, _lib as L
dev = torch.device("cuda:0")
for M in (4097, 8194):
    for K in (3072, 2304, 768):
        a = torch.randn(M, K, device=dev).bfloat16(); b = torch.randn(768, K, device=dev).bfloat16()
        for split in (1, 2, 3, 4):
            out = torch.zeros(M, 768, device=dev)
            kw = dict(out_f32=out, flags=L.EPI_ATOMIC if split > 1 else 0, split_k=split)
            for _ in range(3): Fn.gemm(a, b, L.GEMM_NT, prec=L.PREC_BF16, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): Fn.gemm(a, b, L.GEMM_NT, prec=L.PREC_BF16, **kw)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 50
            print(f"M={M} K={K} split {split}: {us:7.1f} us  {2.0*M*768*K/us/1e6:7.1f} TFLOP/s", flush=True)
