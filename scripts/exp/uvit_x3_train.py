"""Config 4 (MaskGiTUViT, 728.7 M parameters) in the "bf16x3" and "f16" modes: 40 optimizer steps on ONE fixed synthetic batch (batch 16, 256
tokens, 77 text states) next to the same steps in exact f32 - the loss falls the same way in all three (the operand-image caches, the
producer-written images and the GEMMs see fresh weights every step: AdamW writes them through raw pointers).  The f16 mode runs the
fp16 recipe (model.f16_update_grad_scale(): an operand overflow skips the step and halves the gradient scale) and reports what it did.
python scripts/exp/uvit_x3_train.py [modes, default bf16x3,f16,f32]"""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "open-muse_amd")); sys.path.insert(0, ROOT)
import muse
from muse import modeling_transformer_v2 as M
from bench import UVIT_CC12M

dev = torch.device("cuda", 0)
out = {}
MODES = (sys.argv[1].split(",") if len(sys.argv) > 1 else ["bf16x3", "f16", "f32"])
for mode in MODES:
    init = M.MaskGiTUViT_v2._init_weights
    M.MaskGiTUViT_v2._init_weights = lambda self: None
    try:
        model = muse.MaskGiTUViT(**UVIT_CC12M)
    finally:
        M.MaskGiTUViT_v2._init_weights = init
    model.to(dev).train().set_compute_dtype(torch.float32 if mode == "f32" else mode)
    g = torch.Generator(device=dev).manual_seed(0)
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.fill_(1.0) if n.endswith("norm.weight") else p.normal_(0.0, 0.02, generator=g)
    opt = muse.FusedAdamW(muse.grouped_parameters(model, 0.01), lr=1e-4, betas=(0.9, 0.999), weight_decay=0.01, eps=1e-8)
    B, S = 16, 256
    ids = torch.randint(0, 8256, (B, S), device=dev, generator=g)
    labels = torch.where(torch.rand(B, S, device=dev, generator=g) < 0.5, torch.randint(0, 8192, (B, S), device=dev, generator=g),
                         torch.full((B, S), -100, device=dev))
    enc, cond = torch.randn(B, 77, 768, device=dev, generator=g), torch.randn(B, 768, device=dev, generator=g)
    micro = torch.tensor([[256.0, 256.0, 0.0, 0.0, 6.0]], device=dev).repeat(B, 1)
    losses, skipped, flushed = [], 0, 0
    while len(losses) < 40:
        model.zero_grad(set_to_none=True)
        _, loss = model(ids, enc, cond, micro, labels=labels)
        loss.backward()
        if mode == "f16":
            flushed += model.f16_stats(reset=False)[1]
            if not model.f16_update_grad_scale():
                skipped += 1
                continue
        opt.step()
        losses.append(round(float(loss.detach()), 4))
    out[mode] = losses
    print(mode, "loss every 5 steps:", losses[::5], "last", losses[-1], flush=True)
    if mode == "f16":
        print("   f16: steps skipped on operand overflow", skipped, "; gradient scale", model.f16_grad_scale_for(B * S), "; operand elements rounded to zero over the run", flushed, flush=True)
    del model, opt
    torch.cuda.empty_cache()
for mode in MODES:
    if mode != "f32" and "f32" in out:
        d = max(abs(a - b) for a, b in zip(out[mode], out["f32"]))
        print(f"max |loss_{mode} - loss_f32| over 40 steps:", round(d, 4), "(first loss", out["f32"][0], ")")
    assert all(l == l for l in out[mode]) and out[mode][-1] < out[mode][0] - 1.0
