"""one MaskGiTUViT forward at the decoding batch of the inference-latency leg (2 x bs rows of 256 tokens, bf16 compute), repeated, for
`rocprofv3 --kernel-trace --stats`: where the ~9 ms of a 512-row forward go (launch count and time per kernel family)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "open-muse_amd"))
import torch
import muse
from muse import modeling_transformer_v2 as M

dev = "cuda"
bs = int(os.environ.get("BS", "1"))
reps = int(os.environ.get("REPS", "20"))
init = M.MaskGiTUViT_v2._init_weights
M.MaskGiTUViT_v2._init_weights = lambda self: None
try:
    tr = muse.MaskGiTUViT()
finally:
    M.MaskGiTUViT_v2._init_weights = init
tr.to(dev).eval().set_compute_dtype(torch.bfloat16)
g = torch.Generator(device=dev).manual_seed(0)
with torch.no_grad():
    for n, p in tr.named_parameters():
        p.fill_(1.0) if n.endswith("norm.weight") else p.normal_(0.0, 0.02, generator=g)
tr.mark_weights_changed()
ids = torch.full((2 * bs, 256), tr.config.mask_token_id, dtype=torch.long, device=dev)
enc = torch.randn(2 * bs, 77, 768, device=dev, generator=g)
pooled = torch.randn(2 * bs, 768, device=dev, generator=g)
micro = torch.tensor([[256.0, 256.0, 0.0, 0.0, 6.0]], device=dev).repeat(2 * bs, 1)
with torch.no_grad():
    for _ in range(3):
        tr(ids, enc, pooled, micro)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        tr(ids, enc, pooled, micro)
    torch.cuda.synchronize()
print(f"forward at {2 * bs} x 256 rows: {(time.perf_counter() - t0) / reps * 1e3:.2f} ms")
