"""In-kernel timeline of attention2.hip's forward kernel (library built with -DATT2_TS: scripts/exp/attn2_variants.sh -> variants/libmuse_hip_ts.so).
   MUSE_HIP_LIB=.../libmuse_hip_ts.so python scripts/exp/attn2_ts.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "open-muse_amd"))
import torch
from muse import ops, _hip

B, S, nh, hd = 64, 257, 16, 48
H = nh * hd
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn(B * S, 3 * H, device=dev, generator=g).to(torch.bfloat16)
alpha = hd ** -0.5
lib = ctypes.CDLL(_hip.LIB_PATH)
ts = torch.zeros(256 * 8 * 4 * 16, dtype=torch.int64, device=dev)
for _ in range(3):
    ops.attention_fwd(qkv, B, S, nh, hd, alpha)
torch.cuda.synchronize()
assert lib.muse_dbg_attn2_ts(ctypes.c_void_p(ts.data_ptr())) == 0
ops.attention_fwd(qkv, B, S, nh, hd, alpha)
torch.cuda.synchronize()
lib.muse_dbg_attn2_ts(ctypes.c_void_p(0))
t = ts.view(256, 8, 4, 16).cpu().double()
names = ["top", "barrier passed", "dma+loads+stores issued", "QK done", "softmax done", "PV done", "packed", "shared computed", "merge barrier", "head end"]
for it in range(4):
    base = t[:, :, it, 0].min(dim=1, keepdim=True).values.unsqueeze(-1) if False else t[:, :, it, 0:1]
    print(f"head {it}: mean cycles since this wave's top (min / mean / max over 2048 waves)")
    for k in range(1, 10):
        d = (t[:, :, it, k] - t[:, :, it, 0])
        print(f"   {names[k]:26s} {d.min():9.0f} {d.mean():9.0f} {d.max():9.0f}")
    if it < 3:
        d = t[:, :, it + 1, 0] - t[:, :, it, 0]
        print(f"   next top                   {d.min():9.0f} {d.mean():9.0f} {d.max():9.0f}")
print("kernel span (cycles):", float(t[:, :, :, 9].max() - t[:, :, 0, 0].min()))
w = t[0, :, 1, :10] - t[0, 0, 1, 0]
print("block 0, head 1, per wave rows = waves, cols = stamps:")
for r in w: print("   " + " ".join(f"{x:8.0f}" for x in r))
