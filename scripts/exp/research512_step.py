"""One training step of configs/research_run_512_with_downsample.yaml's transformer on MI355X (one-off measurement, not a bench line):
MaskGiTUViT with force_down_up_sample (1024 tokens of a 512 x 512 picture -> 256 inside the blocks and the 22 layers), bf16 compute,
FusedAdamW with train_muse.py's two parameter groups, then EMAModel.step (use_ema: True in that config).  Synthetic tokens and CLIP states.
    python scripts/exp/research512_step.py [batch] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "open-muse_amd"))
import torch
import muse
from muse import modeling_transformer_v2 as M

CFG = dict(vocab_size=8256, hidden_size=1024, intermediate_size=2816, num_hidden_layers=22, num_attention_heads=16, in_channels=768,
           block_out_channels=(768,), block_has_attention=(True,), block_num_heads=12, num_res_blocks=3, res_ffn_factor=4, patch_size=1,
           encoder_hidden_size=768, add_cross_attention=True, project_encoder_hidden_states=True, codebook_size=8192, num_vq_tokens=512,
           initializer_range=0.02, norm_type="rmsnorm", layer_norm_eps=1e-6, ln_elementwise_affine=True, use_encoder_layernorm=False,
           use_bias=False, hidden_dropout=0.0, attention_dropout=0.0, use_codebook_size_for_output=True, use_empty_embeds_for_uncond=True,
           add_cond_embeds=True, cond_embed_dim=768, add_micro_cond_embeds=True, micro_cond_encode_dim=256, micro_cond_embed_dim=1280,
           force_down_up_sample=True)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = "cuda"
init = M.MaskGiTUViT_v2._init_weights
M.MaskGiTUViT_v2._init_weights = lambda self: None
try:
    model = muse.MaskGiTUViT(**CFG)
finally:
    M.MaskGiTUViT_v2._init_weights = init
model.to(dev).train().set_compute_dtype(torch.bfloat16)
g = torch.Generator(device=dev).manual_seed(0)
with torch.no_grad():
    for n, p in model.named_parameters():
        p.fill_(1.0) if n.endswith("norm.weight") else p.normal_(0.0, 0.02, generator=g)
n_params = sum(p.numel() for p in model.parameters())
opt = muse.FusedAdamW(muse.grouped_parameters(model, 0.01), lr=1e-4, betas=(0.9, 0.999), weight_decay=0.01, eps=1e-8)
ema = muse.EMAModel(model.parameters(), decay=0.9999, update_after_step=0, update_every=1)
S = 1024
ids = torch.randint(0, 8256, (batch, S), device=dev, generator=g)
labels = torch.where(torch.rand(batch, S, device=dev, generator=g) < 0.5, torch.randint(0, 8192, (batch, S), device=dev, generator=g),
                     torch.full((batch, S), -100, device=dev))
enc, cond = torch.randn(batch, 77, 768, device=dev, generator=g), torch.randn(batch, 768, device=dev, generator=g)
micro = torch.tensor([[512.0, 512.0, 0.0, 0.0, 6.0]], device=dev).repeat(batch, 1)


def step():
    model.zero_grad(set_to_none=True)
    _, loss = model(ids, enc, cond, micro, labels=labels)
    loss.backward()
    opt.step()
    ema.step(model.parameters())
    return loss


l0 = float(step()); step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"research_run_512_with_downsample transformer: {n_params / 1e6:.1f} M parameters, batch {batch} x 1024 tokens, bf16: "
      f"{dt * 1e3:.1f} ms per step (forward + backward + AdamW groups + EMA), {batch / dt:.1f} images/s, loss {l0:.4f} -> {float(loss):.4f}, "
      f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
