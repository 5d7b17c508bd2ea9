import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "open-muse_amd"))
import torch
from muse import ops
B, S, C = 128, 256, 1024
rows = B * S
x, r, dm, dpre = (torch.randn(rows, C, device="cuda") for _ in range(4))
w = torch.rand(C, device="cuda") + 0.5
ss = torch.randn(B, 2 * C, device="cuda") * 0.3
def t(name, fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / n * 1e3:.0f} us", flush=True)
for mode in (0, 1):
    n, pre = ops.norm_res_fwd(x, w, 1e-6, mode, residual=r, want_pre=True)
    t(f"mode {mode} fwd unfused (norm_res_fwd + adaln_fwd bf16)", lambda: ops.adaln_fwd(ops.norm_res_fwd(x, w, 1e-6, mode, residual=r, want_pre=True)[0], ss, B, out_dtype=torch.bfloat16))
    t(f"mode {mode} fwd fused", lambda: ops.norm_adaln_fwd(x, w, ss, B, 1e-6, mode, residual=r, out_dtype=torch.bfloat16))
    def unf():
        dn, dss = ops.adaln_bwd(dm, n, ss, B)
        return ops.norm_res_bwd(dn, pre, w, 1e-6, mode, dpre=dpre, also_bf16=True)
    t(f"mode {mode} bwd unfused (adaln_bwd + norm_res_bwd)", unf)
    t(f"mode {mode} bwd fused", lambda: ops.norm_adaln_bwd(dm, pre, w, ss, B, 1e-6, mode, dpre=dpre, also_bf16=True))
