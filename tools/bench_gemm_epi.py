"""micro-bench of one GEMM shape under the epilogue variants the ViT driver uses (in-context flags)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hsenet_amd import functional as Fn, _lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 10
reps = 20
dev = torch.device("cuda:0")
M = B * 4097
def run(name, lay, m, n, k, **kw):
    a = torch.randn((m, k), device=dev).bfloat16()
    b = torch.randn((n, k) if lay == L.GEMM_NT else (k, n), device=dev).bfloat16()
    for _ in range(3): Fn.gemm(a, b, lay, prec=L.PREC_BF16, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): Fn.gemm(a, b, lay, prec=L.PREC_BF16, **kw)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"{name:34s} M={m} N={n} K={k}: {us:8.1f} us  {2.0*m*n*k/us/1e6:7.1f} TFLOP/s", flush=True)
f32 = lambda n: torch.empty(M, n, device=dev)
b16 = lambda n: torch.empty(M, n, device=dev, dtype=torch.bfloat16)
bias768, bias3072 = torch.randn(768, device=dev), torch.randn(3072, device=dev)
res = torch.randn(M, 768, device=dev)
run("proj plain bf16 out", L.GEMM_NT, M, 768, 768, out_bf16=b16(768))
run("proj f32 out", L.GEMM_NT, M, 768, 768, out_f32=f32(768))
run("proj f32 out + bias", L.GEMM_NT, M, 768, 768, out_f32=f32(768), bias=bias768, flags=L.EPI_BIAS)
run("proj f32 out + bias + resid", L.GEMM_NT, M, 768, 768, out_f32=f32(768), bias=bias768, resid=res, flags=L.EPI_BIAS | L.EPI_RESID)
run("fc2 f32 out + bias + resid", L.GEMM_NT, M, 768, 3072, out_f32=f32(768), bias=bias768, resid=res, flags=L.EPI_BIAS | L.EPI_RESID)
run("fc1 plain bf16", L.GEMM_NT, M, 3072, 768, out_bf16=b16(3072))
run("fc1 bias+gelu", L.GEMM_NT, M, 3072, 768, out_bf16=b16(3072), bias=bias3072, flags=L.EPI_BIAS | L.EPI_GELU)
run("fc1 bias+gelu+save dgelu", L.GEMM_NT, M, 3072, 768, out_bf16=b16(3072), bias=bias3072, aux_out=b16(3072), flags=L.EPI_BIAS | L.EPI_GELU | L.EPI_SAVE_DGELU)
z = torch.randn(M, 3072, device=dev).bfloat16()
run("dz plain bf16", L.GEMM_NN, M, 3072, 768, out_bf16=b16(3072))
run("dz gelu_bwd(dgelu) ", L.GEMM_NN, M, 3072, 768, out_bf16=b16(3072), aux=z, flags=L.EPI_GELU_BWD | L.EPI_SAVE_DGELU)
run("dz gelu_bwd(dgelu)+colsum", L.GEMM_NN, M, 3072, 768, out_bf16=b16(3072), aux=z, flags=L.EPI_GELU_BWD | L.EPI_SAVE_DGELU, colsum_out=torch.zeros(3072, device=dev))
run("qkv plain", L.GEMM_NT, M, 2304, 768, out_bf16=b16(2304))
