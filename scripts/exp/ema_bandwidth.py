"""muse.EMAModel.step on the parameter list of config 4 (MaskGiTUViT, 728.7 M f32 parameters in ~500 tensors): HIP-event time of the
one-launch update and its HBM rate (12 bytes per parameter: read shadow, read parameter, write shadow), next to the reference's
per-tensor expression in torch on the same tensors.
    python scripts/exp/ema_bandwidth.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "open-muse_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import muse
import weights as W

torch.manual_seed(0)
with torch.device("cuda"):
    model = muse.MaskGiTUViT(**W.UVIT_CC12M)
params = list(model.parameters())
n = sum(p.numel() for p in params)
ema = muse.EMAModel(params, decay=0.9999)
ema.optimization_step = 100


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


t = timeit(lambda: ema.step(params))
print(f"tensors {len(params)}  parameters {n / 1e6:.1f} M")
print(f"muse_ema_multi (one launch): {t:.3f} ms  {12.0 * n / t / 1e9:.2f} TB/s")
shadow = [p.detach().clone() for p in params]


def ref_form():
    with torch.no_grad():
        for s, p in zip(shadow, params):
            s.sub_(1e-4 * (s - p))


t2 = timeit(ref_form, 5)
print(f"reference expression, tensor by tensor in torch ({3 * len(params)} launches): {t2:.3f} ms")
