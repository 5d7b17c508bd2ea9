"""Does the step's own stream idle at the step boundary when NO profiler is attached?  HIP events: one behind the last kernel of step N
(after optimizer.step, on the step's stream), one in front of mask_sample of step N + 1 and one behind the embedding kernel that follows it.
(A rocprofv3 kernel trace of bench.py shows 2.3 ms between the last backward kernel and the next step's first kernel, and 0.9 ms between
mask_sample and the embedding: profiles/r06_step_timeline.txt - this says whether that is the trace's own host overhead.)
    python scripts/exp/step_gap.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "open-muse_amd"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch
sys.argv = ["bench.py"]
import bench
import muse
from muse import ops, training

dev = torch.device("cuda:0")
vq, model, opt, _ = bench.build_models("B", "bf16x3", dev, seed=1234)
step = muse.TrainStep(vq, model, opt)
px, cls = bench.synthetic_batch(64, dev, seed=1000)
ev = {"end": [], "pre_mask": [], "post_mask": [], "post_embed": []}
inner_mask, inner_embed = ops.mask_sample, ops.embed_fwd


def mask_sample(*a, **k):
    e = torch.cuda.Event(enable_timing=True); e.record(); ev["pre_mask"].append(e)
    r = inner_mask(*a, **k)
    e = torch.cuda.Event(enable_timing=True); e.record(); ev["post_mask"].append(e)
    return r


def embed_fwd(*a, **k):
    r = inner_embed(*a, **k)
    e = torch.cuda.Event(enable_timing=True); e.record(); ev["post_embed"].append(e)
    return r


ops.mask_sample = training.ops.mask_sample = mask_sample
ops.embed_fwd = embed_fwd
inner_call = muse.TrainStep.__call__
K = 12
for i in range(K):
    loss, _ = step(px, cls, next_pixel_values=px)
    # the caller's stream has joined the step's stream on return: an event here sits behind the step's last kernel
    e = torch.cuda.Event(enable_timing=True); e.record(); ev["end"].append(e)
torch.cuda.synchronize()
print("step   end(N-1) -> before mask_sample   mask_sample   mask_sample -> embedding done   whole step (end to end)")
for i in range(4, K):
    a = ev["end"][i - 1].elapsed_time(ev["pre_mask"][i])
    b = ev["pre_mask"][i].elapsed_time(ev["post_mask"][i])
    c = ev["post_mask"][i].elapsed_time(ev["post_embed"][i]) if len(ev["post_embed"]) == K else float("nan")
    d = ev["end"][i - 1].elapsed_time(ev["end"][i])
    print(f"{i:3d}   {a:8.3f} ms                      {b:6.3f} ms      {c:8.3f} ms                      {d:8.3f} ms")
