"""SURVEY.md section 8 row a12 — MaskGiTUViT_v2 (config 4) on the HIP path, forward only (round 1): the new row / elementwise
kernels against plain torch, and the whole model against the real reference's golden (tiny config) and the pinned CPU oracle
(mid config with the real text length 77).  f32 ("parity mode"): tolerances 1e-5-class on ops, 1e-4 on logits."""
import json
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ops():
    from muse import ops
    return ops


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("rows,cols", [(5, 32), (67, 768), (130, 1024), (3, 4100)])
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("with_res", [False, True])
def test_norm_with_residual_stream(rows, cols, mode, with_res):
    ops = _ops()
    x, r, w = rnd((rows, cols), 1, 2.0) + 0.3, rnd((rows, cols), 2), 1 + 0.1 * rnd((cols,), 3)
    v = (x + r if with_res else x).double()
    if mode == 0:
        ref = v * torch.rsqrt(v.pow(2).mean(-1, keepdim=True) + 1e-6) * w.double()
    else:
        ref = F.layer_norm(v, (cols,), w.double(), None, 1e-6)
    y, pre = ops.norm_res_fwd(x.to(DEV), w.to(DEV), 1e-6, mode, residual=r.to(DEV) if with_res else None, want_pre=True)
    assert rel_err(y, ref) < 2e-6
    assert torch.equal(pre.cpu(), (x + r) if with_res else x)
    y2, none = ops.norm_res_fwd(x.to(DEV), None, 1e-6, mode, residual=r.to(DEV) if with_res else None)   # no gain vector
    assert none is None and rel_err(y2, ref / w.double()) < 2e-6


def test_adaln_silu_sinusoid_weighted_mean():
    ops = _ops()
    B, S, C = 3, 16, 24
    x, ss = rnd((B * S, C), 10), rnd((B, 2 * C), 11)
    ref = x.view(B, S, C) * (1 + ss[:, None, :C]) + ss[:, None, C:]
    assert rel_err(ops.adaln_fwd(x.to(DEV), ss.to(DEV), B), ref.reshape(B * S, C)) < 1e-6
    z = rnd((1000,), 12, 3.0)
    assert rel_err(ops.silu_fwd(z.to(DEV)), F.silu(z.double())) < 1e-6
    feats = torch.tensor([256.0, 256.0, 0.0, 0.0, 6.0, 512.0, 384.0, 16.0, 8.0, 5.5])
    for dim in (16, 256, 9):
        half = dim // 2
        freq = torch.exp(torch.arange(half, dtype=torch.float32) * (-math.log(10000) / half))
        ang = feats[:, None] * freq[None, :]
        ref = torch.cat([ang.cos(), ang.sin()], dim=1)
        if dim % 2:
            ref = F.pad(ref, (0, 1))
        got = ops.sinusoidal_encode(feats.to(DEV), dim).cpu()
        assert got.shape == ref.shape and float((got - ref).abs().max()) < 2e-4   # angles up to 512 rad in f32
    v, w = rnd((777,), 13).abs(), torch.rand(777) + 0.5
    assert abs(float(ops.weighted_mean(v.to(DEV), w.to(DEV))[0]) - float((v * w).sum() / w.sum())) < 1e-6


@pytest.mark.parametrize("B,H,W,C", [(2, 4, 4, 24), (1, 16, 16, 96), (3, 5, 7, 8),
                                     (2, 16, 16, 1024), (1, 8, 8, 2048), (2, 6, 5, 128), (1, 4, 4, 10)])   # (the 4-channel-per-thread kernels and their fallbacks)
def test_depthwise_conv_and_grn(B, H, W, C):
    ops = _ops()
    x = rnd((B, H, W, C), 20)
    w = rnd((C, 1, 3, 3), 21, 0.3)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), None, padding=1, groups=C).permute(0, 2, 3, 1)
    got = ops.dwconv3x3_nhwc(x.reshape(-1, C).contiguous().to(DEV), w.to(DEV), B, H, W, C)
    assert rel_err(got.view(B, H, W, C), ref) < 1e-6
    gamma, beta = rnd((C,), 22), rnd((C,), 23)
    xd = x.double()
    gx = torch.norm(xd, p=2, dim=(1, 2), keepdim=True)
    nx = gx / (gx.mean(dim=-1, keepdim=True) + 1e-6)
    ref2 = gamma.double() * (xd * nx) + beta.double() + xd
    got2 = ops.grn_fwd(x.reshape(-1, C).contiguous().to(DEV), gamma.to(DEV), beta.to(DEV), B, H * W)
    assert rel_err(got2.view(B, H, W, C), ref2) < 2e-6


def _load_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "uvit_tiny.npz"))
    cfg = json.load(open(os.path.join(golden_dir, "config_uvit_tiny.json")))
    sd = {k[len("param."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("param.")}
    return g, cfg, sd


def test_uvit_forward_vs_reference_golden(golden_dir):
    """logits and both losses of the real reference (tests/golden/make_golden.py::golden_uvit) on the tiny configuration"""
    import muse
    g, cfg, sd = _load_golden(golden_dir)
    model = muse.MaskGiTUViT(**cfg)
    model.load_state_dict(sd, strict=True)
    model.to(DEV)
    args = [torch.from_numpy(g[k]).to(DEV) for k in ("input_ids", "encoder_hidden_states", "cond_embeds", "micro_conds")]
    labels = torch.from_numpy(g["labels"]).to(DEV)
    ref_logits = torch.from_numpy(g["logits"])
    logits, loss = model(*args, labels=labels)
    assert logits.shape == ref_logits.shape
    assert rel_err(logits, ref_logits) < 1e-4
    assert abs(float(loss) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    _, loss_w = model(*args, labels=labels, label_smoothing=float(g["label_smoothing"]),
                      loss_weight=torch.from_numpy(g["loss_weight"]).to(DEV))
    assert abs(float(loss_w) - float(g["loss_weighted"])) < 1e-4 * abs(float(g["loss_weighted"]))
    assert rel_err(model(*args), ref_logits) < 1e-4          # inference call: logits only


def test_uvit_forward_vs_oracle_text77():
    """a mid-size configuration with the real text length (77 CLIP tokens), 8 x 8 tokens, head dims 24 / 32, against the
    pinned CPU oracle"""
    import muse
    from oracle import uvit_oracle as U
    cfg = dict(hidden_size=128, cond_embed_dim=48, micro_cond_encode_dim=32, micro_cond_embed_dim=160, encoder_hidden_size=80,
               vocab_size=264, codebook_size=256, in_channels=64, block_out_channels=(96,), num_res_blocks=2,
               block_num_heads=4, num_hidden_layers=3, num_attention_heads=4, intermediate_size=192, layer_norm_eps=1e-6)
    torch.manual_seed(7)
    model = muse.MaskGiTUViT(**cfg)
    g = torch.Generator().manual_seed(8)
    with torch.no_grad():
        for _, p in model.named_parameters():
            if float(p.abs().max()) == 0.0:
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)   # the zero-initialised tensors
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    B, S, L = 2, 64, 77
    ids = torch.randint(0, cfg["vocab_size"], (B, S), generator=g)
    labels = torch.where(torch.rand(B, S, generator=g) < 0.5, torch.randint(0, 256, (B, S), generator=g), torch.full((B, S), -100))
    enc, cond = torch.randn(B, L, 80, generator=g), torch.randn(B, 48, generator=g)
    micro = torch.tensor([[256.0, 256.0, 0.0, 0.0, 6.0], [512.0, 512.0, 32.0, 64.0, 4.5]])
    ocfg = dict(model.config)
    ref_logits, ref_loss = U.uvit_forward(sd, ocfg, ids, enc, cond, micro, labels=labels)
    model.to(DEV)
    logits, loss = model(ids.to(DEV), enc.to(DEV), cond.to(DEV), micro.to(DEV), labels=labels.to(DEV))
    assert rel_err(logits, ref_logits) < 1e-4
    assert abs(float(loss) - float(ref_loss)) < 1e-4 * abs(float(ref_loss))


# ---- backward --------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,cols", [(5, 32), (67, 768), (33, 4096)])
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("with_dpre", [False, True])
def test_norm_backward(rows, cols, mode, with_dpre):
    ops = _ops()
    v = (rnd((rows, cols), 30, 2.0) + 0.3).double().requires_grad_(True)
    w = (1 + 0.1 * rnd((cols,), 31)).double().requires_grad_(True)
    dy, dpre = rnd((rows, cols), 32), rnd((rows, cols), 33)
    if mode == 0:
        y = v * torch.rsqrt(v.pow(2).mean(-1, keepdim=True) + 1e-6) * w
    else:
        y = F.layer_norm(v, (cols,), w, None, 1e-6)
    y.backward(dy.double())
    ref_dv = v.grad + (dpre.double() if with_dpre else 0)
    dv, dw = ops.norm_res_bwd(dy.to(DEV), v.detach().float().to(DEV), w.detach().float().to(DEV), 1e-6, mode,
                              dpre=dpre.to(DEV) if with_dpre else None)
    assert rel_err(dv, ref_dv) < 5e-6 and rel_err(dw, w.grad) < 5e-6


def test_adaln_silu_scale_backward():
    ops = _ops()
    B, S, C = 3, 50, 24
    x, ss, dy = rnd((B * S, C), 40), rnd((B, 2 * C), 41), rnd((B * S, C), 42)
    dx, dss = ops.adaln_bwd(dy.to(DEV), x.to(DEV), ss.to(DEV), B)
    assert rel_err(dx, (dy.view(B, S, C) * (1 + ss[:, None, :C])).reshape(B * S, C)) < 1e-6
    ref = torch.cat([(dy * x).double().view(B, S, C).sum(1), dy.double().view(B, S, C).sum(1)], dim=1)
    assert rel_err(dss, ref) < 2e-6
    z = rnd((1000,), 43, 3.0).double().requires_grad_(True)
    g = rnd((1000,), 44)
    F.silu(z).backward(g.double())
    assert rel_err(ops.silu_bwd(z.detach().float().to(DEV), g.to(DEV)), z.grad) < 2e-6
    m = rnd((37, 40), 45).to(DEV)
    w = (torch.rand(37) + 0.5).to(DEV)
    num, den = torch.tensor([5.0], device=DEV), torch.tensor([2.5], device=DEV)
    exp = m.clone()
    exp[:, :32] *= (w * 2.0)[:, None]
    assert rel_err(ops.scale_rows_(m, w, num, den, 32), exp) < 1e-6


@pytest.mark.parametrize("B,H,W,C", [(2, 4, 4, 24), (1, 16, 16, 96), (3, 5, 7, 8),
                                     (2, 16, 16, 1024), (1, 8, 8, 2048), (2, 6, 5, 128), (1, 4, 4, 10)])   # (the 4-channel-per-thread kernels and their fallbacks)
def test_depthwise_conv_and_grn_backward(B, H, W, C):
    ops = _ops()
    x = rnd((B, H, W, C), 50).double().requires_grad_(True)
    w = rnd((C, 1, 3, 3), 51, 0.3).double().requires_grad_(True)
    dy = rnd((B, H, W, C), 52)
    F.conv2d(x.permute(0, 3, 1, 2), w, None, padding=1, groups=C).permute(0, 2, 3, 1).backward(dy.double())
    dx, dw = ops.dwconv3x3_bwd(dy.reshape(-1, C).contiguous().to(DEV), x.detach().float().reshape(-1, C).contiguous().to(DEV),
                               w.detach().float().to(DEV), B, H, W, C)
    assert rel_err(dx.view(B, H, W, C), x.grad) < 2e-6 and rel_err(dw, w.grad) < 5e-6
    xg = rnd((B, H, W, C), 53).double().requires_grad_(True)
    gamma, beta = rnd((C,), 54).double().requires_grad_(True), rnd((C,), 55).double().requires_grad_(True)
    gx = torch.norm(xg, p=2, dim=(1, 2), keepdim=True)
    nx = gx / (gx.mean(dim=-1, keepdim=True) + 1e-6)
    (gamma * (xg * nx) + beta + xg).backward(dy.double())
    xr = xg.detach().float().reshape(-1, C).contiguous().to(DEV)
    _, stats = ops.grn_fwd(xr, gamma.detach().float().to(DEV), beta.detach().float().to(DEV), B, H * W, want_stats=True)
    dxg, dgam, dbet = ops.grn_bwd(dy.reshape(-1, C).contiguous().to(DEV), xr, gamma.detach().float().to(DEV), stats, B, H * W)
    assert rel_err(dxg.view(B, H, W, C), xg.grad) < 5e-6
    assert rel_err(dgam, gamma.grad) < 5e-6 and rel_err(dbet, beta.grad) < 5e-6


def _grad_check(model, ref_grads, tol):
    worst = (0.0, None)
    for name, p in model.named_parameters():
        assert p.grad is not None, name
        ref = ref_grads[name]
        e = float((p.grad.detach().double().cpu().reshape(ref.shape) - ref.double()).abs().max() / (ref.double().abs().max() + 1e-12))
        worst = max(worst, (e, name))
    assert worst[0] < tol, worst


def test_uvit_backward_vs_reference_golden(golden_dir):
    """loss.backward() through the hand-written reverse pass: every one of the 120 parameter gradients against the real
    reference (plain mean cross-entropy), then label smoothing + per-token loss weights against the pinned oracle"""
    import muse
    from oracle import uvit_oracle as U
    g, cfg, sd = _load_golden(golden_dir)
    model = muse.MaskGiTUViT(**cfg)
    model.load_state_dict(sd, strict=True)
    model.to(DEV).train()
    args = [torch.from_numpy(g[k]).to(DEV) for k in ("input_ids", "encoder_hidden_states", "cond_embeds", "micro_conds")]
    labels = torch.from_numpy(g["labels"]).to(DEV)
    logits, loss = model(*args, labels=labels)
    assert abs(float(loss) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    loss.backward()
    _grad_check(model, {k[len("grad."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("grad.")}, 5e-4)
    model.zero_grad(set_to_none=True)
    lw = torch.from_numpy(g["loss_weight"])
    ls = float(g["label_smoothing"])
    _, loss_w = model(*args, labels=labels, label_smoothing=ls, loss_weight=lw.to(DEV))
    (2.0 * loss_w).backward()                                            # an upstream gradient other than 1
    cpu_args = [torch.from_numpy(g[k]) for k in ("input_ids", "encoder_hidden_states", "cond_embeds", "micro_conds")]
    _, lo, ref = U.uvit_loss_and_grads(sd, cfg, *cpu_args, torch.from_numpy(g["labels"]), ls, lw)
    assert abs(float(loss_w) - float(lo)) < 1e-4 * abs(float(lo))
    _grad_check(model, {k: 2.0 * v for k, v in ref.items()}, 5e-4)
    with torch.no_grad():                                                # inference keeps working and records nothing
        assert model(*args, labels=labels)[1].grad_fn is None


def test_uvit_generate2(golden_dir):
    """reference :330-479 on the HIP forward: classifier-free guidance doubles the batch, every token ends up decoded (no
    mask id left), given tokens are kept, and a fixed generator reproduces the sample"""
    import muse
    g, cfg, sd = _load_golden(golden_dir)
    model = muse.MaskGiTUViT(**cfg)
    model.load_state_dict(sd, strict=True)
    model.to(DEV).eval()
    B, S, L = 2, 16, 7
    enc = torch.from_numpy(g["encoder_hidden_states"]).to(DEV)
    cond = torch.from_numpy(g["cond_embeds"]).to(DEV)
    micro = torch.tensor([[256.0, 256.0, 0.0, 0.0, 6.0]], device=DEV)                       # broadcast over the batch (:383-384)
    empty, empty_c = torch.zeros(1, L, cfg["encoder_hidden_size"], device=DEV), torch.zeros(1, cfg["cond_embed_dim"], device=DEV)
    outs = []
    for _ in range(2):
        gen = torch.Generator(device=DEV).manual_seed(123)
        ids, inter = model.generate2(enc, cond, micro, empty, empty_c, timesteps=4, guidance_scale=2.0, generator=gen, seq_len=S,
                                     return_intermediate=True)
        outs.append(ids)
        assert ids.shape == (B, S) and ids.dtype == torch.long and len(inter) == 4
        assert int(ids.min()) >= 0 and int(ids.max()) < cfg["codebook_size"]
    assert torch.equal(outs[0], outs[1])
    given = torch.full((B, S), cfg["vocab_size"] - 1, dtype=torch.long, device=DEV)
    given[:, :5] = torch.arange(5, device=DEV)
    ids = model.generate2(enc, cond, micro, empty, empty_c, input_ids=given, timesteps=3, guidance_scale=0, seq_len=S,
                          generator=torch.Generator(device=DEV).manual_seed(5))
    assert torch.equal(ids[:, :5], given[:, :5]) and int(ids.max()) < cfg["codebook_size"]


def test_uvit_train_step_with_fused_adamw(golden_dir):
    """FusedAdamW on a model without a flat parameter buffer: ONE muse_adamw_multi launch over the device table of all 120 tensors ==
    torch.optim.AdamW"""
    import muse
    g, cfg, sd = _load_golden(golden_dir)
    model = muse.MaskGiTUViT(**cfg)
    model.load_state_dict(sd, strict=True)
    model.to(DEV).train()
    opt = muse.FusedAdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.99), weight_decay=0.05, eps=1e-8)
    twins = [torch.nn.Parameter(p.detach().clone()) for p in model.parameters()]
    ref = torch.optim.AdamW(twins, lr=1e-3, betas=(0.9, 0.99), weight_decay=0.05, eps=1e-8)
    args = [torch.from_numpy(g[k]).to(DEV) for k in ("input_ids", "encoder_hidden_states", "cond_embeds", "micro_conds")]
    labels = torch.from_numpy(g["labels"]).to(DEV)
    for _ in range(2):
        model.zero_grad(set_to_none=True)
        _, loss = model(*args, labels=labels)
        loss.backward()
        for p, q in zip(model.parameters(), twins):
            q.grad = p.grad.detach().clone()
        opt.step()
        ref.step()
    for (name, p), q in zip(model.named_parameters(), twins):
        assert rel_err(p, q) < 1e-5, name


def test_bf16x3_weight_planes_refreshed_by_fused_adamw(golden_dir):
    """round 6: in the bf16x3 mode the weights' (hi, lo) operand planes are cached across steps (tape_ops._wp: one stacked tensor per
    q|k|v / k|v / wi_0|wi_1 / single Linear) and muse.FusedAdamW writes every updated parameter's planes inside its own kernel (table
    column 6 above bit 8).  Three steps on a model wide enough for the four-plane kernel: after each step every cached plane pair
    equals a fresh split of the f32 master bit for bit, the parameters equal those of the same run with MUSE_X3_WEIGHT_PLANES off
    (per-step split) bit for bit, and at least one cache hit served a product from planes the optimizer wrote."""
    import muse
    from muse import ops, tape_ops
    cfg = dict(vocab_size=520, hidden_size=256, in_channels=128, block_out_channels=(128,), encoder_hidden_size=128, cond_embed_dim=128,
               micro_cond_encode_dim=32, micro_cond_embed_dim=160, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512,
               block_num_heads=2, num_res_blocks=1, codebook_size=512, mask_token_id=519)
    B, S = 2, 256
    gen = torch.Generator().manual_seed(3)
    ids = torch.randint(0, 512, (B, S), generator=gen).to(DEV)
    labels = torch.where(torch.rand(B, S, generator=gen) < 0.5, torch.randint(0, 512, (B, S), generator=gen), torch.full((B, S), -100)).to(DEV)
    enc, cond = torch.randn(B, 77, 128, generator=gen).to(DEV), torch.randn(B, 128, generator=gen).to(DEV)
    micro = torch.tensor([[256.0, 256.0, 0.0, 0.0, 6.0]]).repeat(B, 1).to(DEV)
    results = []
    for planes_on in (True, False):
        torch.manual_seed(11)
        model = muse.MaskGiTUViT(**cfg)
        model.to(DEV).train().set_compute_dtype("bf16x3")
        opt = muse.FusedAdamW(muse.grouped_parameters(model, 0.01), lr=1e-3, betas=(0.9, 0.99), weight_decay=0.01, eps=1e-8)
        old = tape_ops.X3_WEIGHT_PLANES
        tape_ops.X3_WEIGHT_PLANES = planes_on
        try:
            for step in range(3):
                model.zero_grad(set_to_none=True)
                _, loss = model(ids, enc, cond, micro, labels=labels)
                loss.backward()
                opt.step()
                cache = model.__dict__.get("_pcache", {})
                assert (len(cache) > 0) == planes_on
                for key, (ver, wp) in cache.items():
                    ws = [p for p in model.parameters() if id(p) in key]
                    ws.sort(key=lambda p: key.index(id(p)))
                    fresh = ops._split_planes_now(torch.cat([w.detach().reshape(w.shape[0], -1) for w in ws], dim=0).contiguous())
                    assert torch.equal(wp.planes, fresh), f"stale operand planes after step {step}"
        finally:
            tape_ops.X3_WEIGHT_PLANES = old
        results.append(([p.detach().clone() for p in model.parameters()], float(loss)))
        if planes_on:
            assert any(getattr(p, "_muse_planes", None) is not None for p in model.parameters())
    for a, b in zip(results[0][0], results[1][0]):
        assert torch.equal(a, b)
    assert results[0][1] == results[1][1]


def test_f16_mode_training_steps(golden_dir):
    """round 6, the "f16" compute mode end to end on a small U-ViT: three optimizer steps with every fusion of the mode on (weights' half
    images kept across steps and refreshed inside muse.FusedAdamW's kernel, producers writing half images; a block's weight gradients
    as one grouped launch on the dW stream in both runs) give bit for bit the parameters and loss of the same run with the first two
    off (every operand through muse_cast_f32_to_f16): those fusions move bytes, not values.  After
    each step every cached weight image equals a fresh cast of the f32 master; operand overflows are handled by the fp16 recipe
    (model.f16_update_grad_scale: skip the step, halve the scale); the loss follows the exact-f32 mode's to TF32-class error."""
    import muse
    from muse import ops, tape_ops
    cfg = dict(vocab_size=520, hidden_size=256, in_channels=128, block_out_channels=(128,), encoder_hidden_size=128, cond_embed_dim=128,
               micro_cond_encode_dim=32, micro_cond_embed_dim=160, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512,
               block_num_heads=2, num_res_blocks=1, codebook_size=512, mask_token_id=519)
    B, S = 2, 256
    gen = torch.Generator().manual_seed(3)
    ids = torch.randint(0, 512, (B, S), generator=gen).to(DEV)
    labels = torch.where(torch.rand(B, S, generator=gen) < 0.5, torch.randint(0, 512, (B, S), generator=gen), torch.full((B, S), -100)).to(DEV)
    enc, cond = torch.randn(B, 77, 128, generator=gen).to(DEV), torch.randn(B, 128, generator=gen).to(DEV)
    micro = torch.tensor([[256.0, 256.0, 0.0, 0.0, 6.0]]).repeat(B, 1).to(DEV)
    results, results_skipped = [], []
    for mode, fused in (("f16", True), ("f16", False), (torch.float32, False)):
        torch.manual_seed(11)
        model = muse.MaskGiTUViT(**cfg)
        model.to(DEV).train().set_compute_dtype(mode)
        opt = muse.FusedAdamW(muse.grouped_parameters(model, 0.01), lr=1e-3, betas=(0.9, 0.99), weight_decay=0.01, eps=1e-8)
        old = (tape_ops.F16_WEIGHT_IMAGES, ops.F16_PRODUCERS)
        tape_ops.F16_WEIGHT_IMAGES, ops.F16_PRODUCERS = fused, fused
        losses = []
        try:
            skipped = 0
            while len(losses) < 3:
                model.zero_grad(set_to_none=True)
                _, loss = model(ids, enc, cond, micro, labels=labels)
                loss.backward()
                # the fp16 recipe: a gradient operand beyond half's range turns the gradients NaN, is counted, and costs a skipped step
                # and a halved scale (this toy model's gradients outgrow the default scale's headroom at its random init)
                if mode == "f16" and not model.f16_update_grad_scale():
                    skipped += 1
                    assert skipped < 24
                    continue
                step = len(losses)
                opt.step()
                losses.append(float(loss))
                assert all(bool(torch.isfinite(p).all()) for p in model.parameters())
                cache = model.__dict__.get("_wcache", {})
                assert (len(cache) > 0) == (fused and mode == "f16")
                for key, (ver, wh) in cache.items():
                    ws = [p for p in model.parameters() if id(p) in key]
                    ws.sort(key=lambda p: key.index(id(p)))
                    fresh = ops.cast_to_f16(torch.cat([w.detach().reshape(w.shape[0], -1) for w in ws], dim=0).contiguous())
                    assert wh.dtype == torch.float16 and torch.equal(wh, fresh), f"stale half weight image after step {step}"
            if mode == "f16":
                images = model.__dict__["_f16_images"]
                print("f16 mode, fusions", fused, ": steps skipped on operand overflow", skipped, "-> gradient scale", model.f16_grad_scale,
                      "; image cache hits / casts / produced", images.hits, images.misses, images.produced)
                assert (images.produced > 0) == fused
                results_skipped.append(skipped)
        finally:
            tape_ops.F16_WEIGHT_IMAGES, ops.F16_PRODUCERS = old
        results.append(([p.detach().clone() for p in model.parameters()], losses))
    # a forward without a tape (inference): the mode's images do not outlive their consumers, logits to TF32-class error of the f32 mode's
    with torch.no_grad():
        model.set_compute_dtype("f16")
        lf = model(ids, enc, cond, micro)
        images = model.__dict__["_f16_images"]
        assert not images.persist and len(images.lru) <= images.recent
        model.set_compute_dtype(torch.float32)
        l32 = model(ids, enc, cond, micro)
    assert rel_err(lf, l32) < 5e-3
    assert results_skipped[0] == results_skipped[1]            # (an overflow is a property of the values, not of who wrote the image)
    for a, b in zip(results[0][0], results[1][0]):
        assert torch.equal(a, b)
    assert results[0][1] == results[1][1]
    for lf, l32 in zip(results[0][1], results[2][1]):
        assert abs(lf - l32) < 2e-3 * abs(l32), (results[0][1], results[2][1])


def test_f16_mode_optimizer_guard(golden_dir):
    """the f16 mode with muse.FusedAdamW and NO call in the training loop: the toy model of test_f16_mode_training_steps overflows half's range
    at the default gradient scale (NaN gradients).  The optimizer kernel reads the backward pass's overflow counter on the device and
    leaves parameters and moments untouched (muse_adamw_skip_flag); the next backward pass reads that step's counters from pinned memory
    and halves the scale.  Parameters stay finite throughout, three updates are skipped, the scale ends where the explicit recipe puts
    it, and the loss of the steps that were applied falls like the exact-f32 run's."""
    import muse
    cfg = dict(vocab_size=520, hidden_size=256, in_channels=128, block_out_channels=(128,), encoder_hidden_size=128, cond_embed_dim=128,
               micro_cond_encode_dim=32, micro_cond_embed_dim=160, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512,
               block_num_heads=2, num_res_blocks=1, codebook_size=512, mask_token_id=519)
    B, S = 2, 256
    gen = torch.Generator().manual_seed(3)
    ids = torch.randint(0, 512, (B, S), generator=gen).to(DEV)
    labels = torch.where(torch.rand(B, S, generator=gen) < 0.5, torch.randint(0, 512, (B, S), generator=gen), torch.full((B, S), -100)).to(DEV)
    enc, cond = torch.randn(B, 77, 128, generator=gen).to(DEV), torch.randn(B, 128, generator=gen).to(DEV)
    micro = torch.tensor([[256.0, 256.0, 0.0, 0.0, 6.0]]).repeat(B, 1).to(DEV)
    torch.manual_seed(11)
    model = muse.MaskGiTUViT(**cfg)
    model.to(DEV).train().set_compute_dtype("f16")
    opt = muse.FusedAdamW(muse.grouped_parameters(model, 0.01), lr=1e-3, betas=(0.9, 0.99), weight_decay=0.01, eps=1e-8)
    before = [p.detach().clone() for p in model.parameters()]
    losses, moved = [], []
    for it in range(8):
        model.zero_grad(set_to_none=True)
        _, loss = model(ids, enc, cond, micro, labels=labels)
        loss.backward()
        opt.step()                                  # (no f16_update_grad_scale: the guard lives in the optimizer kernel)
        losses.append(float(loss.detach()))         # (a device read: the step's counter copy has arrived before the next backward
        torch.cuda.synchronize()                    #  pass, so the scale moves one step after each overflow - deterministic here)
        assert all(bool(torch.isfinite(p).all()) for p in model.parameters()), it
        moved.append(any(not torch.equal(p.detach(), q) for p, q in zip(model.parameters(), before)))
    model.zero_grad(set_to_none=True)
    _, loss = model(ids, enc, cond, micro, labels=labels)
    torch.cuda.synchronize()
    loss.backward()                                 # (reads the last step's counters)
    skipped = model.__dict__.get("_f16_skipped_steps", 0)
    print("f16 mode, optimizer guard: losses", [round(l, 4) for l in losses], "; updates applied after iteration", moved.index(True),
          "; skipped", skipped, "; gradient scale", model.f16_grad_scale)
    assert moved[:3] == [False, False, False] and moved[3] and skipped == 3 and model.f16_grad_scale == 2.0 ** 16
    assert losses[0] == losses[1] == losses[2] == losses[3] and losses[-1] < losses[3]
    assert model.f16_stats()[0] > 0                  # (what the three overflowed passes counted; the call resets the counters)
    model.zero_grad(set_to_none=True)
    _, loss = model(ids, enc, cond, micro, labels=labels)
    loss.backward()
    assert model.f16_stats()[0] == 0                 # at the scale the policy settled on nothing overflows


def test_uvit_fused_adamw_parameter_groups(golden_dir):
    """training/train_muse.py:425-445 on the U-ViT: two groups (no weight decay on bias / layer_norm.weight / mlm_ln.weight /
    embeddings.weight) through ONE muse_adamw_multi_groups launch == torch.optim.AdamW with the same groups"""
    import muse
    g, cfg, sd = _load_golden(golden_dir)
    model = muse.MaskGiTUViT(**cfg)
    model.load_state_dict(sd, strict=True)
    model.to(DEV).train()
    wd = 0.3
    groups = muse.training.grouped_parameters(model, wd)
    assert groups[0]["params"] and groups[1]["params"]
    opt = muse.FusedAdamW(groups, lr=1e-3, betas=(0.9, 0.99), weight_decay=wd, eps=1e-8)
    twins = {n: torch.nn.Parameter(p.detach().clone()) for n, p in model.named_parameters()}
    nd = ("bias", "layer_norm.weight", "mlm_ln.weight", "embeddings.weight")            # the literal of train_muse.py:426-436
    ref = torch.optim.AdamW([{"params": [p for n, p in twins.items() if not any(x in n for x in nd)], "weight_decay": wd},
                             {"params": [p for n, p in twins.items() if any(x in n for x in nd)], "weight_decay": 0.0}],
                            lr=1e-3, betas=(0.9, 0.99), weight_decay=wd, eps=1e-8)
    assert [len(x["params"]) for x in ref.param_groups] == [len(x["params"]) for x in groups]
    args = [torch.from_numpy(g[k]).to(DEV) for k in ("input_ids", "encoder_hidden_states", "cond_embeds", "micro_conds")]
    labels = torch.from_numpy(g["labels"]).to(DEV)
    for _ in range(2):
        model.zero_grad(set_to_none=True)
        _, loss = model(*args, labels=labels)
        loss.backward()
        for n, p in model.named_parameters():
            twins[n].grad = p.grad.detach().clone()
        opt.step()
        ref.step()
    for n, p in model.named_parameters():
        assert rel_err(p, twins[n]) < 1e-5, n


@pytest.mark.parametrize("cd", [torch.float32, torch.bfloat16])
def test_uvit_gradient_buckets_reduced_inside_backward_on_rccl(golden_dir, cd):
    """data-parallel U-ViT on the real backend (RCCL group of one rank): muse.GradReducer hangs itself on the model's
    `grad_tensors_hook`; the hand-written backward reports finished gradients block by block, buckets are packed / all-reduced /
    unpacked on the communication stream behind the compute and weight-gradient streams while backward continues
    (training/train_muse.py:753-759 gets this from DDP), finish() has nothing left.  With one rank the average is the local gradient:
    every parameter gradient must equal the run without a reducer bit for bit, f32 and bf16 compute (weight gradients on the side
    stream), several buckets per backward."""
    import torch.distributed as dist
    import muse
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(29650 + os.getpid() % 150)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    g, cfg, sd = _load_golden(golden_dir)
    args = [torch.from_numpy(g[k]).to(DEV) for k in ("input_ids", "encoder_hidden_states", "cond_embeds", "micro_conds")]
    labels = torch.from_numpy(g["labels"]).to(DEV)

    def run(with_reducer):
        model = muse.MaskGiTUViT(**cfg)
        model.load_state_dict(sd, strict=True)
        model.to(DEV).train().set_compute_dtype(cd)
        red = muse.GradReducer(model, bucket_bytes=16 * 1024) if with_reducer else None
        seen = []
        if red is not None:
            inner = red._reduce_list
            red._reduce_list = lambda bucket, side=None: (seen.append(len(bucket)), inner(bucket, side))
        _, loss = model(*args, labels=labels)
        loss.backward()
        n_inside = len(seen)
        if red is not None:
            red.finish()
            assert len(seen) == n_inside and n_inside >= 3 and red.stats["buckets"] == n_inside
        torch.cuda.synchronize()
        return {k: p.grad.detach().clone() for k, p in model.named_parameters()}, float(loss)

    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV, 0))
    try:
        g0, l0 = run(False)
        g1, l1 = run(True)
        assert l0 == l1
        for k in g0:
            assert torch.equal(g0[k], g1[k]), k
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("half", [torch.float16, torch.bfloat16])
def test_half_precision_conditioning_inside_autocast_as_the_training_script_hands_it(golden_dir, half):
    """training/train_muse.py keeps the text encoder in the mixed-precision dtype: `encoder_hidden_states` / `cond_embeds` (and the
    `micro_conds` built from their dtype, :660-663) reach the model as fp16 / bf16 tensors, and validation calls it inside
    torch.autocast (:1068).  The hand-written path ignores autocast and upcasts its inputs: same logits / loss / gradients as the
    call with those values already in f32; U-ViT and the text-conditioned MaskGitTransformer; cond_dropout keeps the dtype"""
    import muse
    import weights as W
    g, cfg, sd = _load_golden(golden_dir)
    model = muse.MaskGiTUViT(**cfg)
    model.load_state_dict(sd, strict=True)
    model.to(DEV).train().set_compute_dtype(torch.bfloat16)
    ids, enc, cond, micro = [torch.from_numpy(g[k]).to(DEV) for k in ("input_ids", "encoder_hidden_states", "cond_embeds", "micro_conds")]
    labels = torch.from_numpy(g["labels"]).to(DEV)
    eh, ch, mh = enc.to(half), cond.to(half), micro.to(half)

    def run(*a, ctx):
        model.zero_grad(set_to_none=True)
        with ctx:
            logits, loss = model(ids, *a, labels=labels)
        loss.backward()
        return logits.detach(), loss.detach(), [p.grad.clone() for p in model.parameters()]
    import contextlib
    l0, s0, g0 = run(eh.float(), ch.float(), mh.float(), ctx=contextlib.nullcontext())
    l1, s1, g1 = run(eh, ch, mh, ctx=torch.autocast("cuda", dtype=half))
    assert l1.dtype == torch.float32 and torch.equal(l0, l1) and torch.equal(s0, s1) and all(torch.equal(a, b) for a, b in zip(g0, g1))
    xcfg = dict(W.TRANSFORMER_TEXT_TINY)
    m = muse.MaskGitTransformer(**xcfg)
    m.load_state_dict(W.fill_state_dict(W.transformer_shapes(xcfg), 800, "transformer"))
    m.to(DEV).train().set_compute_dtype(torch.bfloat16)
    xi, xl, xe = (t.to(DEV) for t in W.transformer_text_inputs(xcfg, 2, 5, 52))
    with torch.autocast("cuda", dtype=half):
        a = m(input_ids=xi, encoder_hidden_states=xe.to(half), labels=xl)
    b = m(input_ids=xi, encoder_hidden_states=xe.to(half).float(), labels=xl)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    e2, c2 = muse.cond_dropout(eh, ch, torch.zeros(1, enc.shape[1], enc.shape[2], device=DEV, dtype=half), torch.zeros(1, cond.shape[1], device=DEV, dtype=half),
                               0.5, uniforms=torch.tensor([0.1, 0.9], device=DEV))
    assert e2.dtype == half and c2.dtype == half and torch.equal(e2[0], eh[0]) and float(e2[1].abs().max()) == 0.0 and float(c2[1].abs().max()) == 0.0


def test_gradient_accumulation_under_the_reducer_on_rccl(golden_dir):
    """gradient_accumulation_steps = 2 (cc12m_uvit_clip.yaml and nine more configurations; accelerate's accumulate() = DDP.no_sync on the
    first micro-batch) on an RCCL group of one rank: muse.GradReducer.no_sync() keeps the first micro-batch local, the second backward's
    gradients are summed into .grad by autograd and finish() averages the SUM.  U-ViT (tape engine) and the class-conditional
    MaskGitTransformer (flat buffer, ranges reduced from inside the second backward): gradients and the AdamW-updated parameters equal
    the run without a reducer, bit for bit"""
    import contextlib
    import torch.distributed as dist
    import weights as W
    import muse
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(29810 + os.getpid() % 150)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    g, cfg, sd = _load_golden(golden_dir)
    args = [torch.from_numpy(g[k]).to(DEV) for k in ("input_ids", "encoder_hidden_states", "cond_embeds", "micro_conds")]
    labels = torch.from_numpy(g["labels"]).to(DEV)
    labels2 = torch.where(labels >= 0, (labels + 3) % cfg["codebook_size"], labels)

    def run_uvit(with_reducer):
        model = muse.MaskGiTUViT(**cfg)
        model.load_state_dict(sd, strict=True)
        model.to(DEV).train().set_compute_dtype(torch.bfloat16)
        opt = muse.FusedAdamW(muse.grouped_parameters(model, 0.01), lr=1e-3)
        red = muse.GradReducer(model, bucket_bytes=16 * 1024) if with_reducer else None
        for i, lab in enumerate((labels, labels2)):
            with (red.no_sync() if red is not None and i == 0 else contextlib.nullcontext()):
                _, loss = model(*args, labels=lab)
                (loss / 2).backward()
            if red is not None and i == 0:
                assert red.stats["buckets"] == 0
        if red is not None:
            red.finish()
            assert red.stats["buckets"] >= 3 and red.stats["bytes"] == 4 * sum(p.numel() for p in model.parameters())
        grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
        opt.step()
        torch.cuda.synchronize()
        return grads, {k: p.detach().clone() for k, p in model.named_parameters()}

    def run_flat(with_reducer):
        tcfg = dict(W.TRANSFORMER_TINY)
        m = muse.MaskGitTransformer(**tcfg)
        m.load_state_dict(W.fill_state_dict(W.transformer_shapes(tcfg), 701, "transformer"))
        m.to(DEV).train().set_compute_dtype(torch.bfloat16)
        opt = muse.FusedAdamW(m.parameters(), lr=1e-3)
        red = muse.GradReducer(m, bucket_bytes=16 * 1024) if with_reducer else None
        batches = [W.transformer_inputs(tcfg, 3, 41), W.transformer_inputs(tcfg, 3, 42)]
        for i, (ids, lab) in enumerate(batches):
            with (red.no_sync() if red is not None and i == 0 else contextlib.nullcontext()):
                _, loss = m(input_ids=ids.to(DEV), labels=lab.to(DEV))
                (loss / 2).backward()
            if red is not None and i == 0:
                assert red.stats["buckets"] == 0
        if red is not None:
            red.finish()
            assert red.stats["buckets"] >= 2 and red.stats["bytes"] == 4 * m.flat_grads().numel()
        grads = m.flat_grads().clone()
        opt.step()
        torch.cuda.synchronize()
        return grads, m.flat_params().clone()

    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV, 0))
    try:
        (g0, p0), (g1, p1) = run_uvit(False), run_uvit(True)
        for k in g0:
            assert torch.equal(g0[k], g1[k]) and torch.equal(p0[k], p1[k]), k
        (f0, q0), (f1, q1) = run_flat(False), run_flat(True)
        assert torch.equal(f0, f1) and torch.equal(q0, q1)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["uvit_tiny_noaffine", "uvit_tiny_layernorm", "uvit_tiny_downup"])
@pytest.mark.parametrize("cd", [torch.float32, torch.bfloat16])
def test_uvit_norm_variants_vs_reference_golden(golden_dir, cd, name):
    """ln_elementwise_affine=False (reference :656-660, :705-711: norms without learnable gains), norm_type="layernorm" (:637-638) and
    force_down_up_sample=True (:510-514, :558-562: stride-2 2x2 conv / transposed conv around the blocks, 8 x 8 tokens -> 4 x 4 inside):
    same state-dict keys and parameter list as the real reference, logits / loss / every gradient against its outputs
    (tests/golden/uvit_tiny_noaffine.npz, uvit_tiny_layernorm.npz, uvit_tiny_downup.npz)"""
    import muse
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg = json.load(open(os.path.join(golden_dir, "config_" + name + ".json")))
    sd = {k[len("param."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("param.")}
    model = muse.MaskGiTUViT(**cfg)
    assert set(model.state_dict().keys()) == set(sd.keys()) and any(k.endswith("norm.weight") for k in sd) == (name != "uvit_tiny_noaffine")
    assert [n for n, _ in model.named_parameters()] == [k for k in sd if k in dict(model.named_parameters())]
    model.load_state_dict(sd, strict=True)
    model.to(DEV).train().set_compute_dtype(cd)
    args = [torch.from_numpy(g[k]).to(DEV) for k in ("input_ids", "encoder_hidden_states", "cond_embeds", "micro_conds")]
    labels = torch.from_numpy(g["labels"]).to(DEV)
    logits, loss = model(*args, labels=labels)
    loss.backward()
    f32 = cd == torch.float32
    assert rel_err(logits, torch.from_numpy(g["logits"])) < (1e-3 if f32 else 3e-2)
    assert abs(float(loss) - float(g["loss"])) < (1e-4 if f32 else 2e-3) * abs(float(g["loss"]))
    _grad_check(model, {k[len("grad."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("grad.")}, 1e-3 if f32 else 8e-2)
    opt = muse.FusedAdamW(muse.grouped_parameters(model, 0.01), lr=1e-3)
    opt.step()                                                  # (the optimizer sees exactly the reference's parameter list)


def test_uvit_bf16_mode_vs_reference_golden(golden_dir):
    """set_compute_dtype(torch.bfloat16): weight-GEMM operands rounded to bf16 (f32 accumulate / outputs), everything else f32.
    Expected from a CPU emulation of the same rounding: logits 8e-3, loss 1.3e-4, gradients <= 2.5e-2 (relative to max)."""
    import muse
    g, cfg, sd = _load_golden(golden_dir)
    model = muse.MaskGiTUViT(**cfg)
    model.load_state_dict(sd, strict=True)
    model.to(DEV).train().set_compute_dtype(torch.bfloat16)
    args = [torch.from_numpy(g[k]).to(DEV) for k in ("input_ids", "encoder_hidden_states", "cond_embeds", "micro_conds")]
    labels = torch.from_numpy(g["labels"]).to(DEV)
    logits, loss = model(*args, labels=labels)
    assert rel_err(logits, torch.from_numpy(g["logits"])) < 3e-2
    assert abs(float(loss) - float(g["loss"])) < 2e-3 * abs(float(g["loss"]))
    loss.backward()
    _grad_check(model, {k[len("grad."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("grad.")}, 8e-2)


def test_uvit_config4_vs_reference_golden(golden_dir):
    """BASELINE.json config 4 at its real size (weights.UVIT_CC12M: 728.7 M parameters, 22 layers) on a 2 x 256-token batch with 77 text
    tokens against the REAL reference's outputs (tests/golden/uvit_full*.npz, make_golden.py::golden_uvit_full): loss, sub-sampled
    logits, eighteen gradients over the network (sub-sampled values and L2 norms).  f32 mode to 1e-3; bf16 mode is reported next to the
    gap the reference itself shows between f32 and CPU-autocast-bf16 at this size (logits 1.7e-2, gradients up to 3.8e-2, loss 1.5e-4)."""
    import weights as W
    import muse
    from muse import modeling_transformer_v2 as M
    g = np.load(os.path.join(golden_dir, "uvit_full.npz"))
    gb = np.load(os.path.join(golden_dir, "uvit_full_bf16.npz"))
    init = M.MaskGiTUViT_v2._init_weights
    M.MaskGiTUViT_v2._init_weights = lambda self: None      # (every tensor is loaded below)
    try:
        model = muse.MaskGiTUViT(**W.UVIT_CC12M)
    finally:
        M.MaskGiTUViT_v2._init_weights = init
    model.load_state_dict(W.fill_by_shapes({k: tuple(v.shape) for k, v in model.state_dict().items()}, int(g["seed"])), strict=True)
    model.to(DEV).train()
    ids, enc, cond, micro, labels = (t.to(DEV) for t in W.uvit_inputs(int(g["batch"]), int(g["seq"]), int(g["text_len"]), int(g["seed"]) + 1))
    keys = W.UVIT_FULL_GRAD_KEYS
    ref_gap = max(float(np.abs(g["grad." + k] - gb["grad." + k]).max()) / float(g["absmax." + k]) for k in keys)
    # "bf16x3": f32 tensors, every GEMM as three bf16 MFMA products (TF32-class-or-tighter: the yaml's enable_tf32 regime on CDNA4) -
    # held to the f32 mode's bounds (north_star's 1e-3)
    # "f16" (round 6): f32 tensors, every weight GEMM as ONE product of IEEE-half operand images = the TF32 operand precision itself
    # (10-bit mantissa), gradient operands through a power-of-two scale - held to what that precision gives at this size (measured:
    # logits 9.6e-4, worst gradient 2.1e-3; an emulated-TF32 run of the reference would sit there too), nothing clamped to half's range
    for cd in (torch.float32, "bf16x3", "f16", torch.bfloat16):
        model.set_compute_dtype(cd)
        model.zero_grad(set_to_none=True)
        from muse import ops
        counts = {"attn": 0, "gemm_x3": 0, "gemm_f16": 0}
        inner_a, inner_g = ops.attention_x3_fwd, ops.gemm

        def count_a(*a, **k):
            counts["attn"] += 1
            return inner_a(*a, **k)

        def count_g(*a, **k):
            r = inner_g(*a, **k)
            counts["gemm_x3"] += 1 if (k.get("x3_lo") is not None and r is not None) else 0
            counts["gemm_f16"] += 1 if a[0].dtype == torch.float16 else 0
            return r
        ops.attention_x3_fwd, ops.gemm = count_a, count_g
        try:
            logits, loss = model(ids, enc, cond, micro, labels=labels)
            loss.backward()
        finally:
            ops.attention_x3_fwd, ops.gemm = inner_a, inner_g
        if cd == "bf16x3":     # the mode's own kernels ran (no silent fallback to the materialised core / the K-concatenated product)
            print("bf16x3 step:", counts["attn"], "fused attention forwards,", counts["gemm_x3"], "four-plane products")
            assert counts["attn"] >= 2 * 22 and counts["gemm_x3"] >= 3 * 6 * 22
        elif cd == "f16":
            clamped, flushed = model.f16_stats()
            print("f16 step:", counts["attn"], "fused attention forwards,", counts["gemm_f16"], "half products; operand elements overflowed / rounded to zero:",
                  clamped, "/", flushed)
            assert counts["attn"] >= 2 * 22 and counts["gemm_f16"] >= 3 * 6 * 22 and counts["gemm_x3"] == 0 and clamped == 0
        else:
            assert counts["attn"] == 0 and counts["gemm_x3"] == 0 and counts["gemm_f16"] == 0
        f32 = cd != torch.bfloat16
        assert tuple(logits.shape) == tuple(g["logits_shape"])
        el = float(np.abs(W.subsample(logits.detach().float(), 16384).cpu().numpy() - g["logits"]).max()) / float(g["logits_absmax"])
        lrel = abs(float(loss) - float(g["loss"])) / float(g["loss"])
        params = dict(model.named_parameters())
        errs, nerrs = {}, {}
        for k in keys:
            gr = params[k].grad.detach().float()
            errs[k] = float(np.abs(W.subsample(gr).cpu().numpy() - g["grad." + k]).max()) / float(g["absmax." + k])
            nerrs[k] = abs(float(gr.double().norm()) - float(g["norm." + k])) / float(g["norm." + k])
        print(cd, "config 4 vs the reference: logits", f"{el:.2e}", "loss", f"{lrel:.1e}", "worst grad", f"{max(errs.values()):.1e}",
              "worst grad norm", f"{max(nerrs.values()):.1e}", f"(reference f32 vs its own autocast: worst grad {ref_gap:.1e})")
        if cd == "f16":
            # ... and against the reference run in the regime the YAML asks for - `enable_tf32`, emulated on the CPU by rounding the operands
            # of every matmul (forward, dX, dW, attention core) to TF32's 10-bit mantissa: tests/golden/make_golden_tf32.py.  That run sits
            # 1.0e-3 (logits) / 2.1e-3 (worst gradient) from the reference's own f32 run - the f16 mode, which rounds the same operands to the
            # same mantissa, sits 9.2e-4 / 2.2e-3 from it: the same distance, i.e. the YAML's own regime.  The two runs are not each other's
            # bits (1.2e-3 / 2.0e-3 apart): the f32 parts run in another operation order, which flips individual roundings, and the products
            # the half kernels refuse - the 2-row conditioning / AdaLN Linears, whose rounding shifts every token alike - stay exact f32 here
            gt = np.load(os.path.join(golden_dir, "uvit_full_tf32emu.npz"))
            el_t = float(np.abs(W.subsample(logits.detach().float(), 16384).cpu().numpy() - gt["logits"]).max()) / float(gt["logits_absmax"])
            eg_t = max(float(np.abs(W.subsample(params[k].grad.detach().float()).cpu().numpy() - gt["grad." + k]).max()) / float(gt["absmax." + k]) for k in keys)
            ref_t = float(np.abs(gt["logits"] - g["logits"]).max()) / float(g["logits_absmax"])
            refg_t = max(float(np.abs(gt["grad." + k] - g["grad." + k]).max()) / float(g["absmax." + k]) for k in keys)
            print(f"f16 config 4 vs the reference under emulated TF32: logits {el_t:.2e}, loss {abs(float(loss) - float(gt['loss'])) / float(gt['loss']):.1e}, "
                  f"worst grad {eg_t:.1e}   (that run vs the reference's f32 run: logits {ref_t:.2e}, worst grad {refg_t:.1e})")
            assert el < 1.5 * ref_t and max(errs.values()) < 1.5 * refg_t, (el, ref_t, max(errs.values()), refg_t)      # no further from f32 than TF32 itself
            assert el_t < 2e-3 and eg_t < 4e-3, (el_t, eg_t)
        assert el < (2e-3 if cd == "f16" else 1e-3 if f32 else 5e-2), (cd, el)
        assert lrel < (1e-4 if f32 else 2e-3), (cd, lrel)
        for k in keys:
            assert errs[k] < (6e-3 if cd == "f16" else 2e-3 if f32 else 1.5e-1), (cd, k, errs[k])
            assert nerrs[k] < (1e-3 if f32 else 3e-2), (cd, k, nerrs[k])


@pytest.mark.parametrize("B,H,W,C", [(2, 8, 8, 24), (3, 6, 10, 768), (1, 32, 32, 768)])
def test_space_to_depth2_and_its_inverse(B, H, W, C):
    """muse_space_to_depth2_nhwc: full [B, H, W, C] <-> packed [B, H/2, W/2, (di, dj, c)], exact data movement in both directions;
    through it the stride-2 2x2 conv and transposed conv of force_down_up_sample equal torch's on the same operands (f32 products)"""
    from muse import ops
    torch.manual_seed(H * W + C)
    x = torch.randn(B * H * W, C, device=DEV)
    want = x.view(B, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(B * (H // 2) * (W // 2), 4 * C)
    x2 = ops.space_to_depth2(x, B, H, W, C)
    assert torch.equal(x2, want)
    assert torch.equal(ops.depth_to_space2(x2, B, H, W, C), x)
    if C <= 64:
        wd = torch.randn(C, C, 2, 2, device=DEV) * 0.1
        ref = F.conv2d(x.view(B, H, W, C).permute(0, 3, 1, 2), wd, stride=2).permute(0, 2, 3, 1).reshape(-1, C)
        got = ops.linear(x2, wd.permute(0, 2, 3, 1).reshape(C, -1).contiguous(), out_dtype=torch.float32)
        assert rel_err(got, ref) < 1e-5
        xs = torch.randn(B * (H // 2) * (W // 2), C, device=DEV)
        wu = torch.randn(C, C, 2, 2, device=DEV) * 0.1
        ref = F.conv_transpose2d(xs.view(B, H // 2, W // 2, C).permute(0, 3, 1, 2), wu, stride=2).permute(0, 2, 3, 1).reshape(-1, C)
        got = ops.depth_to_space2(ops.linear(xs, wu.permute(2, 3, 1, 0).reshape(-1, C).contiguous(), out_dtype=torch.float32), B, H, W, C)
        assert rel_err(got, ref) < 1e-5
    with pytest.raises(Exception):
        ops.space_to_depth2(torch.randn(3 * 3 * 8, 8, device=DEV), 1, 3, 3, 8)      # odd sides are refused, not mis-indexed


def test_uvit_forced_down_up_sample_modes_and_decoding(golden_dir):
    """force_down_up_sample=True beyond the golden comparison (test_uvit_norm_variants_vs_reference_golden): the bf16x3 mode against the
    reference's f32 outputs, the CPU oracle's logits on a second batch, and generate2 (eager == kept HIP graph, ids in range)"""
    import muse
    from oracle import uvit_oracle as U
    g = np.load(os.path.join(golden_dir, "uvit_tiny_downup.npz"))
    cfg = json.load(open(os.path.join(golden_dir, "config_uvit_tiny_downup.json")))
    sd = {k[len("param."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("param.")}
    model = muse.MaskGiTUViT(**cfg)
    model.load_state_dict(sd, strict=True)
    model.to(DEV).train().set_compute_dtype("bf16x3")
    args = [torch.from_numpy(g[k]).to(DEV) for k in ("input_ids", "encoder_hidden_states", "cond_embeds", "micro_conds")]
    logits, loss = model(*args, labels=torch.from_numpy(g["labels"]).to(DEV))
    loss.backward()
    assert rel_err(logits, torch.from_numpy(g["logits"])) < 1e-3 and abs(float(loss) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    _grad_check(model, {k[len("grad."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("grad.")}, 1e-3)
    # a 4 x 4 grid (2 x 2 inside the blocks), inference signature, against the oracle
    model.eval().set_compute_dtype(torch.float32)
    gen = torch.Generator().manual_seed(7)
    ids = torch.randint(0, cfg["vocab_size"], (3, 16), generator=gen)
    enc, cond = torch.randn(3, 5, cfg["encoder_hidden_size"], generator=gen), torch.randn(3, cfg["cond_embed_dim"], generator=gen)
    micro = torch.tensor([[256.0, 256.0, 0.0, 0.0, 6.0]]).repeat(3, 1)
    with torch.no_grad():
        want = U.uvit_forward(sd, cfg, ids, enc, cond, micro)
        got = model(ids.to(DEV), enc.to(DEV), cond.to(DEV), micro.to(DEV))
    assert got.shape == want.shape and rel_err(got, want) < 1e-4
    with pytest.raises(ValueError):
        model(ids[:, :9].to(DEV), enc.to(DEV), cond.to(DEV), micro.to(DEV))          # 3 x 3 tokens: no 2 x 2 blocks
    # decoding: 64 tokens, classifier-free guidance, eager and the kept graph sample the same ids from the same seed
    kw = dict(encoder_hidden_states=enc.to(DEV), cond_embeds=cond.to(DEV), micro_conds=micro.to(DEV),
              empty_embeds=torch.zeros(1, 5, cfg["encoder_hidden_size"], device=DEV), empty_cond_embeds=torch.zeros(1, cfg["cond_embed_dim"], device=DEV),
              timesteps=4, guidance_scale=2.0, seq_len=64)
    a = model.generate2(generator=torch.Generator().manual_seed(3), **kw)
    b = model.generate2(generator=torch.Generator().manual_seed(3), hip_graph=True, **kw)
    c = model.generate2(generator=torch.Generator().manual_seed(3), hip_graph=True, **kw)
    assert a.shape == (3, 64) and int(a.min()) >= 0 and int(a.max()) < cfg["codebook_size"]
    assert torch.equal(a, b) and torch.equal(b, c)


def test_cached_bf16_weights_never_go_stale_across_stackings(golden_dir):
    """the cached bf16 compute copies are validated by autograd version counters, which the raw FusedAdamW kernel does not
    bump - it refreshes ONE copy per parameter (`p._muse_shadow`).  A weight cached first in the fused q|k|v stacking and then alone
    (unfused attention at another sequence length, generate2 without eval()) must not leave a second, silently stale copy behind:
    caching it in a new stacking drops the old one, so after an optimizer step every stacking equals the cast of the f32 master."""
    import muse
    from muse import ops
    cfg = json.load(open(os.path.join(golden_dir, "config_uvit_tiny.json")))
    g = np.load(os.path.join(golden_dir, "uvit_tiny.npz"))
    m = muse.MaskGiTUViT(**cfg)
    m.load_state_dict({k[len("param."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("param.")}, strict=True)
    m.to(DEV).train().set_compute_dtype(torch.bfloat16)
    att = m.transformer_layers[0].attention
    fused = m._wb(att.query, att.key, att.value)
    assert att.query.weight._muse_shadow.data_ptr() == fused.data_ptr()
    alone = m._wb(att.query)                                   # the same weight in another stacking ...
    key_fused = tuple(id(x.weight) for x in (att.query, att.key, att.value))
    assert key_fused not in m._wcache                          # ... evicts the stacking that held it before
    assert att.query.weight._muse_shadow.data_ptr() == alone.data_ptr()
    opt = muse.FusedAdamW(m.parameters(), lr=1e-2)
    args = [torch.from_numpy(g[k]).to(DEV) for k in ("input_ids", "encoder_hidden_states", "cond_embeds", "micro_conds")]
    _, loss = m(*args, labels=torch.from_numpy(g["labels"]).to(DEV))
    loss.backward()
    opt.step()                                                  # raw kernel: no version bump, refreshes the registered shadows
    for mods in ((att.query,), (att.query, att.key, att.value), (att.key, att.value)):
        want = ops.cast_to_bf16(torch.cat([x.weight.data.reshape(x.weight.shape[0], -1) for x in mods]).contiguous())
        assert torch.equal(m._wb(*mods).view(torch.int16), want.view(torch.int16)), len(mods)


def test_bf16_operand_outputs_of_grn_and_gelu_backward_equal_a_cast_of_the_f32_results():
    """bf16 compute mode: GlobalResponseNorm and the GELU backward of a ResBlock write their result directly as the bf16 operand
    of the next weight GEMMs (muse_grn_fwd_ex, muse_gelu_bwd_f32_bf16) - the same bits a cast pass over the f32 result gives"""
    ops = _ops()
    B, S, C = 3, 64, 256
    x = rnd((B * S, C), 91).to(DEV)
    gamma, beta = rnd((C,), 92).to(DEV), rnd((C,), 93).to(DEV)
    y32, st32 = ops.grn_fwd(x, gamma, beta, B, S, want_stats=True)
    y16, st16 = ops.grn_fwd(x, gamma, beta, B, S, want_stats=True, out_dtype=torch.bfloat16)
    assert y16.dtype == torch.bfloat16 and torch.equal(y16, y32.to(torch.bfloat16)) and torch.equal(st16, st32)
    dy = rnd((B * S, C), 94).to(DEV)
    d32 = ops.gelu_bwd(x, dy)
    d16 = ops.gelu_bwd(x, dy, out_dtype=torch.bfloat16)
    assert d16.dtype == torch.bfloat16 and torch.equal(d16, d32.to(torch.bfloat16))


@pytest.mark.parametrize("B,S,C,mode,res", [(3, 16, 64, 0, True), (2, 32, 1024, 1, True), (4, 16, 768, 0, False), (2, 48, 256, 1, False)])
def test_fused_norm_adaln_matches_the_two_kernel_route(B, S, C, mode, res):
    """muse_norm_adaln_fwd / _bwd (norm + AdaLN of a MaskGiTUViT_v2 transformer layer in one pass; the norm output is never written,
    the backward recomputes it) against norm_res_fwd -> adaln_fwd and adaln_bwd -> norm_res_bwd: forward bit-identical (f32 and the
    bf16 operand copy), backward within f32 summation-order differences (the column sums of d(scale | shift) are folded per 16-row
    block, then per image)"""
    ops = _ops()
    rows = B * S
    x, r = rnd((rows, C), 61).to(DEV), (rnd((rows, C), 62).to(DEV) if res else None)
    w = (1.0 + 0.2 * rnd((C,), 63)).to(DEV)
    ss = (0.3 * rnd((B, 2 * C), 64)).to(DEV)
    assert ops.norm_adaln_ok(rows, C, B)
    n, pre = ops.norm_res_fwd(x, w, 1e-6, mode, residual=r, want_pre=True)
    m_ref = ops.adaln_fwd(n, ss, B)
    m, v = ops.norm_adaln_fwd(x, w, ss, B, 1e-6, mode, residual=r)
    assert torch.equal(v, pre) and torch.equal(m, m_ref)
    mb, _ = ops.norm_adaln_fwd(x, w, ss, B, 1e-6, mode, residual=r, out_dtype=torch.bfloat16)
    assert torch.equal(mb, ops.adaln_fwd(n, ss, B, out_dtype=torch.bfloat16))
    dm, dpre = rnd((rows, C), 65).to(DEV), rnd((rows, C), 66).to(DEV)
    dn_ref, dss_ref = ops.adaln_bwd(dm, n, ss, B)
    dv_ref, dw_ref = ops.norm_res_bwd(dn_ref, pre, w, 1e-6, mode, dpre=dpre)
    dv, dw, dss, dvb = ops.norm_adaln_bwd(dm, v, w, ss, B, 1e-6, mode, dpre=dpre, also_bf16=True)
    assert rel_err(dv, dv_ref) < 2e-6 and rel_err(dw, dw_ref) < 5e-6 and rel_err(dss, dss_ref) < 5e-6
    assert torch.equal(dvb, dv.to(torch.bfloat16))
    dv2, dw2, dss2 = ops.norm_adaln_bwd(dm, v, w, ss, B, 1e-6, mode)          # no residual-stream gradient, no bf16 copy
    dv2_ref, _ = ops.norm_res_bwd(dn_ref, pre, w, 1e-6, mode)
    assert rel_err(dv2, dv2_ref) < 2e-6 and torch.equal(dw2, dw) and torch.equal(dss2, dss)


@pytest.mark.parametrize("batch_mappers,fuse_norm", [(False, False), (False, True), (True, False)])
def test_uvit_unfused_routes_still_match_the_reference(golden_dir, batch_mappers, fuse_norm):
    """the per-site AdaLN mapper products (MUSE_ADALN_BATCH=0) and the two-kernel norm -> AdaLN route (MUSE_NORM_ADALN=0) stay
    reachable and correct: same golden check as the default (batched, fused) path"""
    import muse
    g, cfg, sd = _load_golden(golden_dir)
    model = muse.MaskGiTUViT(**cfg)
    model.load_state_dict(sd, strict=True)
    model.to(DEV).train()
    model.batch_adaln_mappers, model.fuse_norm_adaln = batch_mappers, fuse_norm
    args = [torch.from_numpy(g[k]).to(DEV) for k in ("input_ids", "encoder_hidden_states", "cond_embeds", "micro_conds")]
    logits, loss = model(*args, labels=torch.from_numpy(g["labels"]).to(DEV))
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    loss.backward()
    _grad_check(model, {k[len("grad."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("grad.")}, 5e-4)
