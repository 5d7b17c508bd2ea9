"""Is the step host-bound?  Times (a) the host-side enqueue of K steps (no synchronisation inside the loop) and (b) the wall time until the
GPU has finished them.  If (a) ~ (b) the Python / launch path is what limits the step, not the GPU.
    python scripts/exp/host_bound.py [tokens|full]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "open-muse_amd"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch
mode = sys.argv[1] if len(sys.argv) > 1 else "tokens"
sys.argv = ["bench.py"]
import bench
import muse

dev = torch.device("cuda:0")
vq, model, opt, _ = bench.build_models("B", "bf16x3", dev, seed=1234)
step = muse.TrainStep(vq, model, opt)
px, cls = bench.synthetic_batch(64, dev, seed=1000)
toks = vq.get_code(px)


def one():
    if mode == "tokens":
        return step(None, cls, image_tokens=toks)
    return step(px, cls, next_pixel_values=px)


for _ in range(3):
    one()
torch.cuda.synchronize()
K = 10
t0 = time.perf_counter()
for _ in range(K):
    one()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{mode}: host enqueue {1e3 * (t1 - t0) / K:.1f} ms per step, GPU done after {1e3 * (t2 - t0) / K:.1f} ms per step "
      f"(wgrad_stream={model.wgrad_stream})", flush=True)
