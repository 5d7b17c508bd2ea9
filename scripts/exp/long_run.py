"""Does the step time drift over a long run in one process?  70 steps of the default bench configuration (prefetch on), timed in
windows of 10."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "open-muse_amd"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch
sys.argv = ["bench.py"]
import bench
import muse

dev = torch.device("cuda:0")
vq, model, opt, _ = bench.build_models("B", "bf16x3", dev, seed=1234)
step = muse.TrainStep(vq, model, opt)
px, cls = bench.synthetic_batch(64, dev, seed=1000)
for _ in range(3):
    step(px, cls, next_pixel_values=px)
out = []
for w in range(7):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        step(px, cls, next_pixel_values=px)
    torch.cuda.synchronize()
    out.append(round((time.perf_counter() - t0) * 100, 2))
print("ms per step in windows of 10 steps:", out, " reserved GiB", round(torch.cuda.memory_reserved() / 2 ** 30, 1))
