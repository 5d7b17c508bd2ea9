"""SURVEY.md section 8 row a12 at its real size: MaskGiTUViT_v2 of configs/cc12m_uvit_clip.yaml (+ block_num_heads 16 -> hd 64; bench.UVIT_CC12M),
256 tokens, 77 text tokens: time of forward + backward on the f32 path (random weights filled on the GPU, synthetic inputs).
    python scripts/uvit_bench.py [batch] [steps] [f32|bf16]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "open-muse_amd"))
import torch
import muse
from muse import modeling_transformer_v2 as M

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
mode = sys.argv[3] if len(sys.argv) > 3 else "f32"
S = int(sys.argv[4]) if len(sys.argv) > 4 else 256          # 256 (configs/cc12m_uvit_clip.yaml as written) or 1024 (BASELINE config 4)
with_opt = len(sys.argv) > 5 and sys.argv[5] == "adamw"
M.MaskGiTUViT_v2._init_weights = lambda self: None          # 729 M parameters: fill them on the GPU instead
t0 = time.time()
sys.path.insert(0, ROOT)
from bench import UVIT_CC12M  # noqa: E402  (the geometry the 275.10 / 1137.05 GFLOP figures below were counted on)
model = muse.MaskGiTUViT(**UVIT_CC12M)
model.to("cuda")
model.set_compute_dtype(torch.bfloat16 if mode == "bf16" else torch.float32)
g = torch.Generator(device="cuda").manual_seed(0)
with torch.no_grad():
    for n, p in model.named_parameters():
        if n.endswith("norm.weight"):
            p.fill_(1.0)
        else:
            p.normal_(0.0, 0.02, generator=g)
nparam = sum(p.numel() for p in model.parameters())
L = 77
ids = torch.randint(0, 8256, (B, S), device="cuda", generator=g)
labels = torch.where(torch.rand(B, S, device="cuda", generator=g) < 0.5, torch.randint(0, 8192, (B, S), device="cuda", generator=g),
                     torch.full((B, S), -100, device="cuda"))
enc, cond = torch.randn(B, L, 768, device="cuda", generator=g), torch.randn(B, 768, device="cuda", generator=g)
micro = torch.tensor([[256.0, 256.0, 0.0, 0.0, 6.0]], device="cuda").repeat(B, 1)
print(f"built {nparam/1e6:.1f} M params in {time.time()-t0:.1f} s", flush=True)


opt = muse.FusedAdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.999), weight_decay=0.01, eps=1e-8) if with_opt else None


def step():
    model.zero_grad(set_to_none=True)
    _, loss = model(ids, enc, cond, micro, labels=labels)
    loss.backward()
    if opt is not None:
        opt.step()
    return loss

loss = step()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(steps):
    loss = step()
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
gf = 3 * (275.10 if S == 256 else 1137.05 if S == 1024 else float("nan"))   # SURVEY.md section 8d, GFLOP per image
print(f"MaskGiTUViT_v2 {mode} fwd+bwd{'+adamw' if with_opt else ''}: batch {B}, seq {S}, {dt*1e3:.1f} ms/step, {B/dt:.1f} img/s, {gf*B/dt/1e3:.1f} TFLOP/s algorithmic, "
      f"loss {float(loss):.4f}, peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
