import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "open-muse_amd"))
import torch
from muse import ops
torch.manual_seed(0)
for rows, inter in [(33, 1000), (300, 4096)]:
    ab = (torch.randn(rows, 2 * inter) * 1.5).to(torch.bfloat16).cuda()
    dh = torch.randn(rows, inter).to(torch.bfloat16).cuda()
    h, hr = ops.glu_fwd(ab), ops.glu_fwd(ab.float()).to(torch.bfloat16)
    d, dr = ops.glu_bwd(ab, dh), ops.glu_bwd(ab.float(), dh.float()).to(torch.bfloat16)
    for name, x, y in (("fwd", h, hr), ("bwd", d, dr)):
        ne = (x != y)
        print(rows, inter, name, "mismatches", int(ne.sum()), "of", ne.numel(), "max diff", float((x.float() - y.float()).abs().max()))
        if ne.any():
            idx = ne.nonzero()[:5]
            for i in idx.tolist():
                print("   at", i, float(x[i[0], i[1]]), float(y[i[0], i[1]]), "a", float(ab[i[0], i[1] % inter]), "b", float(ab[i[0], inter + i[1] % inter]))
rows, inter = 32768, 4096
ab = torch.randn(rows, 2 * inter, device="cuda").to(torch.bfloat16)
dh = torch.randn(rows, inter, device="cuda").to(torch.bfloat16)
for name, fn in (("glu_bwd", lambda: ops.glu_bwd(ab, dh)), ("glu_fwd", lambda: ops.glu_fwd(ab))):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    print(name, rows, inter, f"{e0.elapsed_time(e1) / 10 * 1e3:.0f} us")
