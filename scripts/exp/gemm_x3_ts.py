"""Phase timeline of the four-plane bf16x3 GEMM (g256::kernel_x3) from in-kernel s_memtime stamps (library built with -DG256_TIMESTAMPS,
scripts/exp/gemm_ts.sh): cycles per 32-wide K-tile against the 3072 cycles its 2 x 96 MFMAs per SIMD issue in, prologue / epilogue."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "open-muse_amd"))
import numpy as np
import torch
from muse import ops
from muse._hip import lib

dev = "cuda"
fn = lib().muse_debug_gemm_ts
fn.argtypes = [ctypes.c_void_p]
for M, N, K, la, lb in ((16384, 1024, 1024, 0, 0), (16384, 8192, 1024, 0, 0), (16384, 1024, 4096, 0, 0), (16384, 1024, 1024, 0, 1), (16384, 1024, 8192, 0, 1),
                        (1024, 1024, 16384, 1, 1)):
    a = torch.randn((M, K) if la == 0 else (K, M), device=dev)
    b = torch.randn((N, K) if lb == 0 else (K, N), device=dev)
    a2, b2 = ops._split_planes_now(a), ops._split_planes_now(b)
    c = torch.empty((M, N), device=dev)
    lda, ldb = (K if la == 0 else M), (K if lb == 0 else N)
    f = lambda: ops.gemm(a2[0], b2[0], c, M, N, K, la=la, lb=lb, lda=lda, ldb=ldb, ldc=N, x3_lo=(a.numel(), b.numel()))
    for _ in range(2):
        f()
    ntile, nkt = ((M + 255) // 256) * ((N + 255) // 256), (K + 31) // 32
    ts = torch.zeros(ntile * 4, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    assert fn(ts.data_ptr()) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record()
    torch.cuda.synchronize()
    fn(None)
    t = ts.cpu().numpy().reshape(ntile, 4).astype(np.float64)
    d = np.diff(t, axis=1)
    us = e0.elapsed_time(e1) * 1e3
    rounds = -(-ntile // 256)
    life = np.median(t[:, 3] - t[:, 0])
    span = t[:, 3].max() - t[:, 0].min()
    print(f"[{M}x{K}]x[{N}x{K}] la{la} lb{lb}: {us:.1f} us, {ntile} tiles ({rounds} round(s)), {nkt} K-tiles; stamp span {span:.0f} -> {span / us:.1f} ticks / us")
    print(f"    prologue {np.median(d[:, 0]):7.0f}   K loop {np.median(d[:, 1]):8.0f} = {np.median(d[:, 1]) / nkt:6.0f} ticks per K-tile"
          f"   epilogue {np.median(d[:, 2]):7.0f}   block {life:8.0f} ticks")
