"""Phase timeline of the 256^2 LDS-DMA GEMM from in-kernel s_memtime stamps (library built with -DG256_TIMESTAMPS by gemm_ts.sh):
cycles per 64-wide K-tile against 2048 cycles of pure MFMA issue, prologue / epilogue share, and the sustained clock."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "open-muse_amd"))
import numpy as np
import torch
from muse import ops
from muse._hip import lib

dev = "cuda"
T, H, I = 16448, 768, 3072
x = torch.randn(T, H, device=dev).to(torch.bfloat16)
w01 = (torch.randn(2 * I, H, device=dev) * 0.03).to(torch.bfloat16)
wo = (torch.randn(H, I, device=dev) * 0.03).to(torch.bfloat16)
hm = torch.randn(T, I, device=dev).to(torch.bfloat16)
dab = torch.randn(T, 2 * I, device=dev).to(torch.bfloat16)
x1 = torch.randn(T, H, device=dev)
fn = lib().muse_debug_gemm_ts
fn.argtypes = [ctypes.c_void_p]
cases = [("fwd FFN-in  [16448x768]x[6144x768]^T -> bf16", lambda: ops.linear(x, w01), 65 * 24, 12),
         ("fwd FFN-out [16448x3072]x[768x3072]^T -> f32 + residual", lambda: ops.linear(hm, wo, out_dtype=torch.float32, residual=x1), 65 * 3, 48),
         ("dX  FFN-in  [16448x6144]x[6144x768] -> bf16", lambda: ops.linear_dgrad(dab, w01), 65 * 3, 96)]
for name, f, ntile, nkt in cases:
    for _ in range(2):
        f()
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
    print(f"{name}: {us:.1f} us, {ntile} tiles ({rounds} round(s)), {nkt} K-tiles")
    print(f"    prologue {np.median(d[:, 0]):7.0f}   K loop {np.median(d[:, 1]):8.0f} = {np.median(d[:, 1]) / nkt:6.0f} cycles per K-tile (MFMA issue 2048)"
          f"   epilogue {np.median(d[:, 2]):7.0f}   block {life:8.0f} cycles -> clock >= {life * rounds / us / 1e3:.2f} GHz if blocks ran back to back")
