"""three launches of the LDS-DMA bf16x3 convolution at the VQGAN level-0 shape, for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "open-muse_amd"))
import torch
from muse import ops
B, H, C = 16, 256, 128
x = torch.randn(B, H, H, C, device="cuda")
w = torch.randn(C, 3, 3, C, device="cuda") * 0.03
wh, wl = ops.split_bf16(w)
xh, xl = ops.split_bf16(x)
res = torch.randn(B, H, H, C, device="cuda")
for _ in range(3):
    out = ops.conv2d_nhwc_split2(xh, xl, wh, wl, B, H, H, C, C, residual=res, gn_groups=32)
torch.cuda.synchronize()
print("ok", float(out[0, 0, 0, 0]))
