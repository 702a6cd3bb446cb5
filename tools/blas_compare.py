"""Context for the roofline numbers: the vendor library (torch.matmul -> hipBLASLt / rocBLAS, bf16) on the same GEMM shapes,
PLAIN (no bias, no LayerNorm, no GELU, no residual): what a non-fused design would pay per GEMM before its epilogue kernels."""
import time
import torch

dev = torch.device("cuda:0")
M = 8 * 16 * 196
for name, N, K in (("qkv", 2304, 768), ("mlp_up", 3072, 768), ("mlp_down", 768, 3072), ("out_proj", 768, 768)):
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        torch.matmul(a, w.t())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        torch.matmul(a, w.t())
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / n * 1e6
    print(f"{name:9s} [{M}x{K}]x[{K}x{N}]: {us:7.1f} us  {2.0*M*N*K/us/1e6:7.0f} TF (vendor library, plain GEMM)")
