import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "open-muse_amd"))
import torch
from muse import ops
B, H, W, Cpad, Cout = 64, 256, 256, 8, 128
x = torch.randn(B, H, W, Cpad, device="cuda")
w4 = torch.randn(Cout, 9, 4, device="cuda"); w4[:, :, 3] = 0
bias = torch.randn(Cout, device="cuda")
w = w4[:, :, :3].reshape(Cout, 3, 3, 3)
wp = torch.zeros(Cout, 3, 3, 8, device="cuda"); wp[..., :3] = w
w_hi, w_lo = ops.split_bf16(wp.contiguous())
big = torch.empty(B, H, W, Cout, device="cuda")
def t(name, fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / n * 1e3:.0f} us", flush=True)
t("conv_in_direct + gn partials", lambda: ops.conv_in_direct(x, w4, B, H, W, 3, Cpad, Cout, bias=bias, gn_groups=32))
t("conv_in_direct, no gn", lambda: ops.conv_in_direct(x, w4, B, H, W, 3, Cpad, Cout, bias=bias, gn_groups=0))
t("implicit-GEMM conv_in + gn partials", lambda: ops.conv2d_nhwc_split(x, w_hi, w_lo, B, H, W, 8, Cout, 3, bias=bias, gn_groups=32))
t("fill 2.1 GB (torch)", lambda: big.fill_(1.0))
t("copy 2.1 GB (torch)", lambda: big.copy_(big.view(-1).roll(0).view_as(big)) if False else big.mul_(1.0))
