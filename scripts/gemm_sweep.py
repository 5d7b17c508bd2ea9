import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "open-muse_amd"))
import torch
from muse import ops
dev = "cuda"
M, N = 16448, 3072
def t(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for K in (64, 128, 256, 512, 1024, 2048, 4096):
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    us = t(lambda: ops.linear(x, w, out=y))
    print(f"NN M={M} N={N} K={K}: {us:.1f} us  {2.0*M*N*K/us/1e6:.1f} TFLOP/s   per-ktile {us/(K/64):.2f} us", flush=True)
# same with torch (hipBLASLt) as a yardstick
for K in (768, 3072):
    x = torch.randn(M, K, device=dev).to(torch.bfloat16); w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    us = t(lambda: torch.matmul(x, w.t()))
    print(f"torch.matmul (hipBLASLt) M={M} N={N} K={K}: {us:.1f} us {2.0*M*N*K/us/1e6:.1f} TFLOP/s", flush=True)
    y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    us = t(lambda: ops.linear(x, w, out=y))
    print(f"muse_gemm              M={M} N={N} K={K}: {us:.1f} us {2.0*M*N*K/us/1e6:.1f} TFLOP/s", flush=True)
