"""Fused-attention micro-benchmark on the shapes of the path (HIP events, per-launch average):
    python scripts/attn_bench.py [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "open-muse_amd"))
import torch
from muse import ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
only = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else None   # indices into SHAPES
dev = "cuda"
SHAPES = [  # name, B, Sq, Skv, nh, hd
    ("configB self S257 hd48", 64, 257, 257, 16, 48),
    ("configA self S257 hd64", 64, 257, 257, 8, 64),
    ("uvit self S256 hd64", 32, 256, 256, 16, 64),
    ("uvit cross S256x77 hd64", 32, 256, 77, 16, 64),
    ("uvit self S1024 hd64", 8, 1024, 1024, 16, 64),
    ("uvit cross S1024x77 hd64", 8, 1024, 77, 16, 64),
]


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


for idx, (name, B, Sq, Skv, nh, hd) in enumerate(SHAPES):
    if only is not None and idx not in only:
        continue
    H = nh * hd
    g = torch.Generator(device=dev).manual_seed(0)
    q = torch.randn(B * Sq, H, device=dev, generator=g).to(torch.bfloat16)
    kv = torch.randn(B * Skv, 2 * H, device=dev, generator=g).to(torch.bfloat16)
    do = torch.randn(B * Sq, H, device=dev, generator=g).to(torch.bfloat16)
    alpha = hd ** -0.5
    ctx, lse = ops.attention_fwd_ex(q, kv[:, :H], kv[:, H:], B, Sq, Skv, nh, hd, alpha)
    tf = timeit(lambda: ops.attention_fwd_ex(q, kv[:, :H], kv[:, H:], B, Sq, Skv, nh, hd, alpha, out=ctx))
    dq, dk, dv = ops.attention_bwd_ex(q, kv[:, :H], kv[:, H:], ctx, do, lse, B, Sq, Skv, nh, hd, alpha)
    tb = timeit(lambda: ops.attention_bwd_ex(q, kv[:, :H], kv[:, H:], ctx, do, lse, B, Sq, Skv, nh, hd, alpha, dq=dq, dk=dk, dv=dv))
    fl = 4.0 * B * nh * Sq * Skv * hd
    io_f = (2 * B * Sq * H + 2 * B * Skv * H) * 2
    io_b = (4 * B * Sq * H + 4 * B * Skv * H) * 2
    print(f"{name:28s} fwd {tf:8.1f} us {fl / tf / 1e6:7.1f} TF/s {io_f / tf / 1e6:6.2f} TB/s | "
          f"bwd {tb:8.1f} us {2.5 * fl / tb / 1e6:7.1f} TF/s {io_b / tb / 1e6:6.2f} TB/s", flush=True)
