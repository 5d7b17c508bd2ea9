import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "open-muse_amd"))
import torch
from muse import ops
B, H, W, C = 128, 16, 16, 1024
x, dy = torch.randn(B * H * W, C, device="cuda"), torch.randn(B * H * W, C, device="cuda")
w = torch.randn(C, 1, 3, 3, device="cuda")
def t(name, fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / n * 1e3:.0f} us", flush=True)
t("dwconv fwd", lambda: ops.dwconv3x3_nhwc(x, w, B, H, W, C))
t("dwconv bwd (dx + dw + colsum)", lambda: ops.dwconv3x3_bwd(dy, x, w, B, H, W, C))
