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
nblk = B * nh
ts = torch.zeros(nblk * 4 * 16, dtype=torch.int64, device=dev)
for _ in range(3):
    ops.attention_fwd(qkv, B, S, nh, hd, alpha)
torch.cuda.synchronize()
assert lib.muse_dbg_attn2_ts(ctypes.c_void_p(ts.data_ptr())) == 0
ops.attention_fwd(qkv, B, S, nh, hd, alpha)
torch.cuda.synchronize()
lib.muse_dbg_attn2_ts(ctypes.c_void_p(0))
t = ts.view(nblk, 4, 16).cpu().double()
order = [(1, "dma + q loads issued"), (2, "images landed, barrier"), (5, "block 0: QK done"), (6, "block 0: max done"), (7, "block 0: exp + PV done"),
         (8, "block 0: packed, stores issued"), (3, "both own blocks done"), (4, "shared block + merge done")]
print("cycles since the wave's entry (min / mean / max over %d waves)" % (nblk * 4))
for k, nm in order:
    d = t[:, :, k] - t[:, :, 0]
    print(f"   {nm:34s} {d.min():9.0f} {d.mean():9.0f} {d.max():9.0f}")
t0 = t[:, :, 0].min()
print("kernel span (cycles):", float(t[:, :, 4].max() - t0), " workgroup lifetime mean:", float((t[:, :, 4].max(dim=1).values - t[:, :, 0].min(dim=1).values).mean()))
start = (t[:, 0, 0] - t0).sort().values
print("workgroup start times (cycles) percentiles 0/25/50/75/100:", [float(start[int(q * (nblk - 1))]) for q in (0, .25, .5, .75, 1)])

# ---- backward ----
do = torch.randn(B * S, H, device=dev, generator=g).to(torch.bfloat16)
ctx, lse = ops.attention_fwd(qkv, B, S, nh, hd, alpha)
for _ in range(3):
    ops.attention_bwd(qkv, ctx, do, lse, B, S, nh, hd, alpha)
torch.cuda.synchronize()
ts.zero_()
assert lib.muse_dbg_attn2_ts(ctypes.c_void_p(ts.data_ptr())) == 0
ops.attention_bwd(qkv, ctx, do, lse, B, S, nh, hd, alpha)
torch.cuda.synchronize()
lib.muse_dbg_attn2_ts(ctypes.c_void_p(0))
t = ts.view(nblk, 4, 16).cpu().double()
order = [(1, "K, V dma issued"), (2, "K, V landed, barrier"), (3, "phase 1: first query block stored"), (4, "phase 1: both query blocks"), (5, "phase 1: shared block computed"),
         (6, "barrier (images free)"), (7, "Q, dO dma issued"), (8, "Q, dO landed, barrier"), (9, "phase 2: first key block stored"), (10, "phase 2: both key blocks"), (11, "end")]
print("BACKWARD: cycles since the wave's entry (min / mean / max over %d waves)" % (nblk * 4))
for k, nm in order:
    d = t[:, :, k] - t[:, :, 0]
    print(f"   {nm:36s} {d.min():9.0f} {d.mean():9.0f} {d.max():9.0f}")
