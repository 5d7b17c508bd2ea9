"""config-4 (U-ViT) step: host enqueue time vs GPU time, and the same step replayed as a captured HIP graph (timing experiment: the
captured AdamW keeps the bias corrections of the captured step)"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "open-muse_amd"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch
sys.argv = ["bench.py"]
import bench
import muse
from muse import modeling_transformer_v2 as M

batch, seq = 128, 256
device = torch.device("cuda:0")
init = M.MaskGiTUViT_v2._init_weights
M.MaskGiTUViT_v2._init_weights = lambda self: None
model = muse.MaskGiTUViT(**bench.UVIT_CC12M)
M.MaskGiTUViT_v2._init_weights = init
model.to(device).train().set_compute_dtype(torch.bfloat16)
g = torch.Generator(device=device).manual_seed(0)
with torch.no_grad():
    for n, p in model.named_parameters():
        p.fill_(1.0) if n.endswith("norm.weight") else p.normal_(0.0, 0.02, generator=g)
opt = muse.FusedAdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.999), weight_decay=0.01, eps=1e-8)
ids = torch.randint(0, 8256, (batch, seq), device=device, generator=g)
labels = torch.where(torch.rand(batch, seq, device=device, generator=g) < 0.5,
                     torch.randint(0, 8192, (batch, seq), device=device, generator=g), torch.full((batch, seq), -100, device=device))
enc = torch.randn(batch, 77, 768, device=device, generator=g)
cond = torch.randn(batch, 768, device=device, generator=g)
micro = torch.tensor([[256.0, 256.0, 0.0, 0.0, 6.0]], device=device).repeat(batch, 1)


def step():
    model.zero_grad(set_to_none=True)
    _, loss = model(ids, enc, cond, micro, labels=labels)
    loss.backward()
    opt.step()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
K = 5
t0 = time.perf_counter()
for _ in range(K):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"eager: host enqueue {1e3 * (t1 - t0) / K:.1f} ms per step, GPU done after {1e3 * (t2 - t0) / K:.1f} ms per step", flush=True)
for ws in (True, False):
    try:
        model.wgrad_stream = ws
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            out = step()
        torch.cuda.synchronize()
        for _ in range(2):
            gr.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            gr.replay()
        torch.cuda.synchronize()
        print(f"graph replay (wgrad_stream={ws}): {1e3 * (time.perf_counter() - t0) / K:.1f} ms per step, loss {float(out.detach()):.4f}", flush=True)
        del gr
    except Exception:
        traceback.print_exc()
        print("capture failed", flush=True)
