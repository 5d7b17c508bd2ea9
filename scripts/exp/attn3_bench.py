"""Timing of the fused bf16x3 attention (attention3.hip) on config 4's shapes: batch 64 x 16 heads, head_dim 64, 256 queries against
256 keys (self) and 77 (text).  MUSE_ATTN3_STAGGER etc. are read per call."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "open-muse_amd"))
from muse import ops
dev = torch.device("cuda", 0)
B, nh, hd, Sq = 64, 16, 64, 256
H = nh * hd
for Skv in (256, 77):
    q = torch.randn(B * Sq, H, device=dev); kv = torch.randn(B * Skv, 2 * H, device=dev); do = torch.randn(B * Sq, H, device=dev)
    k, v = kv[:, :H], kv[:, H:]
    ctx, lse = ops.attention_x3_fwd(q, k, v, B, Sq, Skv, nh, hd, 0.125)
    def tm(f, reps=20):
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps
    tf = tm(lambda: ops.attention_x3_fwd(q, k, v, B, Sq, Skv, nh, hd, 0.125))
    tb = tm(lambda: ops.attention_x3_bwd(q, k, v, ctx, do, lse, B, Sq, Skv, nh, hd, 0.125))
    fl = 4.0 * B * nh * Sq * Skv * hd
    print(f"S_kv {Skv}: fwd {tf:7.1f} us ({3 * fl / tf / 1e6:6.1f} TFLOP/s issued)  bwd {tb:7.1f} us ({3 * 2.5 * fl / tb / 1e6:6.1f} TFLOP/s issued)")
