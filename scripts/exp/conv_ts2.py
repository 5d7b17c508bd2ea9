"""Phase timeline of the GroupNorm-fused patch-slab convolution, launch-per-tile (MUSE_CONV_PERSIST=0) or persistent (=1), from in-kernel
s_memtime stamps (library built with -DCDMA_TIMESTAMPS), plus the plain wall time of both on the normal library.
    MUSE_HIP_LIB=.../libmuse_hip_ts.so MUSE_CONV_PERSIST=0|1 python scripts/exp/conv_ts2.py [HW]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "open-muse_amd"))
import numpy as np
import torch
from muse import ops
from muse._hip import lib

persist = os.environ.get("MUSE_CONV_PERSIST", "0") != "0"
NAMES = ["K loop", "drain look-ahead", "residual + bias loads", "staging passes + stores", "GroupNorm partials"] if persist else \
        ["setup + prologue issue", "first operands land (+ transform)", "K loop", "stage tile in LDS", "stores / GroupNorm partials"]
dev = "cuda"
B, HW, Cin, Cout = 64, int(sys.argv[1]) if len(sys.argv) > 1 else 128, 128, 128
ntile = B * HW * HW // 256
x = torch.randn(B, HW, HW, Cin, device=dev)
sc = torch.rand(B, Cin, device=dev) + 0.5
sh = torch.randn(B, Cin, device=dev) * 0.1
w = torch.randn(Cout, 3, 3, Cin, device=dev) / (3 * Cin ** 0.5)
wh, wl = ops.split_bf16(w)
has_ts = hasattr(lib(), "muse_debug_conv_ts")
for res, gn in ((False, True), (True, True)):
    r = torch.randn(B, HW, HW, Cout, device=dev) if res else None
    run = lambda: ops.conv2d_nhwc_gn_split2(x, sc, sh, wh, wl, B, HW, HW, Cin, Cout, residual=r, gn_groups=32 if gn else 0)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 5
    print(f"persist={int(persist)} {B}x{HW}x{HW} {Cin}->{Cout} residual={res}: {us:.1f} us per launch = {us / (ntile / 256):.2f} us per tile slot, "
          f"{2.0 * B * HW * HW * Cout * 9 * Cin / us / 1e6:.0f} TFLOP/s", flush=True)
    if not has_ts:
        continue
    ts = torch.zeros(ntile * 8, dtype=torch.int64, device=dev)
    fn = lib().muse_debug_conv_ts
    fn.argtypes = [ctypes.c_void_p]
    assert fn(ts.data_ptr()) == 0
    run()
    torch.cuda.synchronize()
    fn(None)
    raw = ts.cpu().numpy().reshape(ntile, 8)[:, :6]
    ok = (raw != 0).all(axis=1)
    t = raw[ok].astype(np.float64)
    d = np.diff(t, axis=1)
    print(f"    blocks with all six stamps: {int(ok.sum())} of {ntile}; per-tile medians in s_memtime ticks:")
    for i, n in enumerate(NAMES):
        print(f"    {n:36s} median {np.median(d[:, i]):8.0f}   p10 {np.percentile(d[:, i], 10):8.0f}   p90 {np.percentile(d[:, i], 90):8.0f}")
    print(f"    stamped span                         median {np.median(t[:, 5] - t[:, 0]):8.0f}")
    if persist:
        nb = 256
        gap = raw[nb:, 0] - raw[:-nb, 5]
        per = raw[nb:, 0] - raw[:-nb, 0]
        print(f"    end of epilogue -> next K loop       median {np.median(gap):8.0f}   p10 {np.percentile(gap, 10):8.0f}   p90 {np.percentile(gap, 90):8.0f}")
        print(f"    tile period (stamp 0 to stamp 0)     median {np.median(per):8.0f}  -> {us / (ntile / 256) / np.median(per) * 1e3:.3f} ns per tick")
