"""Phase timeline of the patch-slab convolution from in-kernel s_memtime stamps (library built with -DCDMA_TIMESTAMPS):
0 entry, 1 prologue DMAs issued, 2 first operands landed + barrier, 3 K loop done, 4 output tile staged in LDS, 5 exit.
    MUSE_HIP_LIB=.../libmuse_hip_ts.so python scripts/exp/conv_ts.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "open-muse_amd"))
import numpy as np
import torch
from muse import ops
from muse._hip import lib

NAMES = {"1": ["setup + prologue DMA issue", "first operands land", "K loop", "stage tile in LDS", "stores / GroupNorm partials"],
         "2": ["K loop", "drain look-ahead DMAs", "residual + bias loads", "two staging passes + stores", "GroupNorm partials"]}
dev = "cuda"
B, HW, Cin, Cout = 64, 128, 128, 128
ntile = B * HW * HW // 256
xh = torch.randn(B, HW, HW, Cin, device=dev).to(torch.bfloat16)
xl = (torch.randn(B, HW, HW, Cin, device=dev) * 0.004).to(torch.bfloat16)
w = torch.randn(Cout, 3, 3, Cin, device=dev) / (3 * Cin ** 0.5)
wh, wl = ops.split_bf16(w)
for res, gn in ((False, False), (True, True)):
    r = torch.randn(B, HW, HW, Cout, device=dev) if res else None
    ts = torch.zeros(ntile * 8, dtype=torch.int64, device=dev)
    fn = lib().muse_debug_conv_ts
    fn.argtypes = [ctypes.c_void_p]
    for _ in range(2):
        ops.conv2d_nhwc_split2(xh, xl, wh, wl, B, HW, HW, Cin, Cout, residual=r, gn_groups=32 if gn else 0)
    torch.cuda.synchronize()
    assert fn(ts.data_ptr()) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.conv2d_nhwc_split2(xh, xl, wh, wl, B, HW, HW, Cin, Cout, residual=r, gn_groups=32 if gn else 0)
    e1.record()
    torch.cuda.synchronize()
    fn(None)
    raw = ts.cpu().numpy().reshape(ntile, 8)[:, :6]
    ok = (raw != 0).all(axis=1)
    print(f"blocks with all six stamps: {int(ok.sum())} of {ntile}; first rows: {raw[:2].tolist()}")
    t = raw[ok].astype(np.float64)
    us = e0.elapsed_time(e1) * 1e3
    d = np.diff(t, axis=1)            # ticks; every XCD has its own counter base, only differences inside a block mean anything
    names = NAMES[os.environ.get("MUSE_CONV_SLAB", "1")]
    print(f"residual+gn={res}: launch {us:.1f} us = {us / (ntile / 256):.2f} us per tile slot; per-tile medians in s_memtime ticks (~0.5 ns):")
    for i, n in enumerate(names):
        print(f"    {n:34s} median {np.median(d[:, i]):8.0f}   p10 {np.percentile(d[:, i], 10):8.0f}   p90 {np.percentile(d[:, i], 90):8.0f}")
    print(f"    stamped span                       median {np.median(t[:, 5] - t[:, 0]):8.0f}")
    if os.environ.get("MUSE_CONV_SLAB", "1") == "2":   # persistent: tile v + 256 follows tile v in the same block
        nb = 256
        gap = raw[nb:, 0] - raw[:-nb, 5]
        per = raw[nb:, 0] - raw[:-nb, 0]
        print(f"    end of epilogue -> next K loop     median {np.median(gap):8.0f}   p10 {np.percentile(gap, 10):8.0f}   p90 {np.percentile(gap, 90):8.0f}")
        print(f"    tile period (stamp 0 to stamp 0)   median {np.median(per):8.0f}  -> {us / (ntile / 256) / np.median(per) * 1e3:.3f} ns per tick")
