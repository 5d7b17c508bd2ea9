"""Timing of the four-plane bf16x3 GEMM (muse_gemm_x3) on config-4 shapes and on one long-K product (steady-state K loop).
Usage on the GPU box: python scripts/exp/gemm_x3_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "open-muse_amd"))
from muse import ops

dev = torch.device("cuda", 0)


def run(M, N, K, la, lb, reps=10):
    a = torch.randn((M, K) if la == 0 else (K, M), device=dev)
    b = torch.randn((N, K) if lb == 0 else (K, N), device=dev)
    a2, b2 = ops._split_planes_now(a), ops._split_planes_now(b)
    c = torch.empty((M, N), device=dev)
    lda, ldb = (K if la == 0 else M), (K if lb == 0 else N)
    f = lambda: ops.gemm(a2[0], b2[0], c, M, N, K, la=la, lb=lb, lda=lda, ldb=ldb, ldc=N, x3_lo=(a.numel(), b.numel()))
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    tf = 2.0 * M * N * K / us / 1e6
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    print(f"[{M}x{K}]x[{N}x{K}] la{la} lb{lb}: {us:8.1f} us  {tf:7.1f} TFLOP/s algorithmic  {3 * tf:7.1f} issued  ({tiles} tiles, {tiles / 256:.2f} rounds, {K // 32} K-tiles)")


for shape in ((16384, 8192, 1024, 0, 0), (16384, 1024, 4096, 0, 0), (16384, 3072, 1024, 0, 0), (16384, 1024, 1024, 0, 0),
              (16384, 1024, 8192, 0, 1), (16384, 4096, 1024, 0, 1), (4096, 4096, 32768, 0, 0), (4096, 4096, 32768, 0, 1), (4096, 4096, 32768, 1, 1),
              (4096, 4096, 1024, 0, 0), (4096, 4096, 4096, 0, 0)):
    run(*shape)
