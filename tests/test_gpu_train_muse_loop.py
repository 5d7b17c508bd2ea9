"""The loop body of training/train_muse.py (:640-790) strung together from this package's pieces, on the 512-pixel research
configuration's SHAPE of problem in miniature - taming tokenizer -> mask_or_random_replace_tokens -> conditioning dropout ->
MaskGiTUViT with force_down_up_sample (loss weights, label smoothing) -> backward -> FusedAdamW with the reference's two parameter groups
-> torch LR scheduler -> EMAModel - against the same loop run on the CPU oracles with the same draws.  f32 compute: parity mode."""
import json
import os

import numpy as np
import pytest
import torch

import weights as W

pytestmark = pytest.mark.gpu
DEV = "cuda"


class _Cfg(dict):
    __getattr__ = dict.__getitem__


def test_train_muse_loop_three_steps_vs_oracles(golden_dir):
    import muse
    from muse.sampling import cosine_schedule
    from oracle import ema_oracle as E
    from oracle import maskgit_oracle as O
    from oracle import taming_oracle as T
    from oracle import uvit_oracle as U
    # ---- models: tiny taming tokenizer (32 codes, 8 x 8 tokens of a 32 x 32 picture), the force_down_up_sample U-ViT of the golden
    vcfg = dict(W.TAMING_TINY, num_embeddings=32)
    vsd = W.fill_state_dict(W.taming_shapes(vcfg), 900, "vqgan")
    g = np.load(os.path.join(golden_dir, "uvit_tiny_downup.npz"))
    ucfg = json.load(open(os.path.join(golden_dir, "config_uvit_tiny_downup.json")))
    usd = {k[len("param."):]: torch.from_numpy(g[k]).clone() for k in g.files if k.startswith("param.")}
    assert vcfg["num_embeddings"] <= ucfg["codebook_size"]
    B, steps, lr, wd, warm = 2, 3, 2e-3, 0.01, 2
    mask_id = ucfg["vocab_size"] - 1
    res = vcfg["resolution"]
    L, De, Dc = 7, ucfg["encoder_hidden_size"], ucfg["cond_embed_dim"]
    gen = torch.Generator().manual_seed(77)
    empty_enc, empty_cond = torch.randn(1, L, De, generator=gen) * 0.3, torch.randn(1, Dc, generator=gen) * 0.3
    micro = torch.tensor([[float(res), float(res), 0.0, 0.0, 6.0]]).repeat(B, 1)
    batches = []
    for s in range(steps):
        batches.append(dict(px=W.images(B, res, 910 + s), enc=torch.randn(B, L, De, generator=gen), cond=torch.randn(B, Dc, generator=gen),
                            t=W.uniforms((B,), 920 + s), nz=None, u=torch.tensor([0.05, 0.95]) if s != 1 else torch.tensor([0.6, 0.2])))
    lr_at = lambda it: lr * min(1.0, (it + 1) / warm)        # noqa: E731  constant-with-warmup, the scheduler of the research configs
    tr = _Cfg(training=_Cfg(min_masking_rate=0.1, predict_all_tokens=True), model=_Cfg(codebook_size=ucfg["codebook_size"]))
    no_decay = ("bias", "layer_norm.weight", "mlm_ln.weight", "embeddings.weight")

    # ---- the loop on the oracles (CPU, f32)
    torch.set_num_threads(min(16, os.cpu_count()))
    sd = {k: v.clone() for k, v in usd.items()}
    names = list(sd.keys())
    m1, m2 = {k: torch.zeros_like(v) for k, v in sd.items()}, {k: torch.zeros_like(v) for k, v in sd.items()}
    shadow = [sd[k].numpy().copy() for k in names]
    sched = E.Schedule(decay=0.99, update_after_step=0)
    ref_losses, ref_tokens = [], []
    for s, b in enumerate(batches):
        with torch.no_grad():
            _, _, tokens = T.encode(vsd, vcfg, b["px"])
        tokens = tokens.reshape(B, -1)
        if b["nz"] is None:
            b["nz"] = W.uniforms(tuple(tokens.shape), 930 + s)
        ids, labels, lw, _ = O.mask_or_random_replace_tokens(tokens, mask_id, 0.1, timesteps=b["t"], noise=b["nz"], all_labels=True)
        enc = O.cond_dropout(b["enc"], empty_enc, b["u"], 0.9)
        cond = O.cond_dropout(b["cond"], empty_cond, b["u"], 0.9)
        _, loss, grads = U.uvit_loss_and_grads(sd, ucfg, ids, enc, cond, micro, labels, label_smoothing=0.1, loss_weight=lw)
        for k in names:
            O.adamw_step(sd[k], grads[k], m1[k], m2[k], s + 1, lr_at(s), 0.9, 0.999, 1e-8, 0.0 if any(nd in k for nd in no_decay) else wd)
        shadow = E.ema_update(shadow, [sd[k].numpy() for k in names], [True] * len(names), sched.next())
        ref_losses.append(float(loss))
        ref_tokens.append(tokens)

    # ---- the same loop on the HIP path
    vq = muse.VQGANModel(**vcfg)
    vq.load_state_dict(vsd)
    vq.to(DEV).eval().requires_grad_(False)
    model = muse.MaskGiTUViT(**ucfg)
    model.load_state_dict(usd, strict=True)
    model.to(DEV).train().set_compute_dtype(torch.float32)
    opt = muse.FusedAdamW(muse.grouped_parameters(model, wd), lr=lr, betas=(0.9, 0.999), weight_decay=wd, eps=1e-8)
    lr_sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda it: min(1.0, (it + 1) / warm))
    ema = muse.EMAModel(model.parameters(), decay=0.99, update_after_step=0)
    assert [n for n, _ in model.named_parameters()] == names
    losses = []
    for s, b in enumerate(batches):
        assert abs(opt.param_groups[0]["lr"] - lr_at(s)) < 1e-12 and abs(opt.param_groups[1]["lr"] - lr_at(s)) < 1e-12
        tokens = vq.get_code(b["px"].to(DEV)).reshape(B, -1)
        assert torch.equal(tokens.cpu(), ref_tokens[s])                                   # bit-exact token indices
        ids, labels, lw, _ = muse.mask_or_random_replace_tokens(tokens, mask_id, tr, cosine_schedule, is_train=True,
                                                                timesteps=b["t"].to(DEV), noise=b["nz"].to(DEV))
        enc, cond = muse.cond_dropout(b["enc"].to(DEV), b["cond"].to(DEV), empty_enc.to(DEV), empty_cond.to(DEV), 0.9, uniforms=b["u"].to(DEV))
        _, loss = model(ids, enc, cond, micro.to(DEV), labels=labels, label_smoothing=0.1, loss_weight=lw)
        loss.backward()
        opt.step()
        lr_sched.step()
        opt.zero_grad(set_to_none=True)
        ema.step(model.parameters())
        losses.append(float(loss))
    for a, r in zip(losses, ref_losses):
        assert abs(a - r) < 2e-4 * abs(r), (losses, ref_losses)
    worst = (0.0, None)
    for (k, p), s_hip, s_ref in zip(model.named_parameters(), ema.shadow_params, shadow):
        moved = float((sd[k] - usd[k]).abs().max())
        err = float((p.detach().cpu() - sd[k]).abs().max())
        worst = max(worst, (err / (moved + 1e-12), k))
        assert float((s_hip.cpu() - torch.from_numpy(s_ref)).abs().max()) <= 3e-2 * moved + 1e-9, k
    # three AdamW steps move every parameter by ~lr per step; the two loops may differ by a small fraction of that movement (AdamW
    # normalises the gradient, so a 1e-4 relative gradient difference shows up as a 1e-4-class difference of the update)
    print(f"train_muse loop, 3 steps: losses {losses} (oracle {ref_losses}); worst parameter error / movement {worst[0]:.2e} ({worst[1]})")
    assert worst[0] < 3e-2, worst
    assert ema.optimization_step == steps and ema.cur_decay_value == E.get_decay(steps, decay=0.99)
