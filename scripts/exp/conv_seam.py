"""Where does the patch-slab convolution spend its time?  T(Cin) at fixed pixels / Cout is a line: slope = K-loop time per channel
chunk (9 taps), intercept = everything per tile that is not the K loop (prologue / epilogue / launch seam).
    MUSE_CONV_SLAB=1|2 python scripts/exp/conv_seam.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "open-muse_amd"))
import torch
from muse import ops

dev = "cuda"
reps = int(os.environ.get("REPS", "6"))
B, HW, Cout = 64, 128, 128
ntile = B * HW * HW // 256
rounds = ntile / 256.0


def timeit(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


print(f"MUSE_CONV_SLAB={os.environ.get('MUSE_CONV_SLAB', '1')}  {B}x{HW}x{HW} pixels, Cout {Cout}, {ntile} tiles = {rounds:.0f} per CU", flush=True)
for res, gn in ((False, False), (True, True)):
    ts = {}
    for Cin in (64, 128, 256, 512):
        xh = torch.randn(B, HW, HW, Cin, device=dev).to(torch.bfloat16)
        xl = (torch.randn(B, HW, HW, Cin, device=dev) * 0.004).to(torch.bfloat16)
        w = torch.randn(Cout, 3, 3, Cin, device=dev) / (3 * Cin ** 0.5)
        wh, wl = ops.split_bf16(w)
        r = torch.randn(B, HW, HW, Cout, device=dev) if res else None
        ts[Cin] = timeit(lambda: ops.conv2d_nhwc_split2(xh, xl, wh, wl, B, HW, HW, Cin, Cout, residual=r, gn_groups=32 if gn else 0))
        fl = 2.0 * B * HW * HW * Cout * 9 * Cin
        print(f"  residual+gn={res}  Cin {Cin:4d}: {ts[Cin]:8.1f} us  {fl / ts[Cin] / 1e6:6.1f} TFLOP/s algorithmic ({3 * fl / ts[Cin] / 1e6:6.1f} issued)", flush=True)
        del xh, xl, r
    slope = (ts[512] - ts[128]) / (384 / 32)          # us per 32-channel chunk (9 taps) per launch
    icpt = ts[128] - slope * 4
    print(f"  -> per tile: K loop {slope / rounds:.2f} us per chunk of 9 taps ({slope / rounds / 9 * 1e3:.0f} ns per tap; MFMA floor 1546 cyc), "
          f"seam {icpt / rounds:.2f} us", flush=True)
