"""Would a captured HIP graph of the train step beat the eager launch path?  Times transformer fwd+bwd (tokens given) and the whole
tokens-given TrainStep (AdamW inside backward) eagerly and as a replayed torch.cuda.CUDAGraph.
    python scripts/exp/graph_step.py"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "open-muse_amd"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch
sys.argv = ["bench.py"]
import bench
import muse

dev = torch.device("cuda:0")
vq, model, opt, _ = bench.build_models("B", "bf16x3", dev, seed=1234)
px, cls = bench.synthetic_batch(64, dev, seed=1000)
toks = vq.get_code(px)
ids, labels, _, _ = muse.prepare_inputs_and_labels(vq, None, cls, model.config.mask_token_id, image_tokens=toks)
step = muse.TrainStep(vq, model, opt)


def fb():
    _, loss = model(input_ids=ids, labels=labels)
    loss.backward()
    return loss


def full():
    return step(None, cls, image_tokens=toks)[0]


def timed(fn, k=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / k


for name, fn in (("fwd+bwd", fb), ("tokens-given TrainStep", full)):
    try:
        eager = timed(fn)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                fn()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = fn()
        torch.cuda.synchronize()
        rep = timed(g.replay)
        print(f"{name}: eager {eager:.2f} ms, graph replay {rep:.2f} ms, loss {float(out):.5f}", flush=True)
        del g
    except Exception:
        traceback.print_exc()
        print(f"{name}: capture failed", flush=True)
    torch.cuda.synchronize()
