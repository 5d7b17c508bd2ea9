"""CPU: the oracle restatement (oracle/maskgit_oracle.py) against golden vectors produced by the real reference
(tests/golden/make_golden.py).  This is what pins the oracle."""
import os

import numpy as np
import pytest
import torch

import weights as W
from oracle import maskgit_oracle as O

torch.set_num_threads(1)


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


@pytest.mark.parametrize("name,cfg", [("transformer_tiny", W.TRANSFORMER_TINY),
                                      ("transformer_tiny_ls", W.TRANSFORMER_TINY),
                                      ("transformer_hd48", W.TRANSFORMER_HD48)])
def test_transformer_forward_backward(golden_dir, name, cfg):
    g = _load(golden_dir, name)
    sd = W.fill_state_dict(W.transformer_shapes(cfg), int(g["seed"]), "transformer")
    ids, labels = W.transformer_inputs(cfg, int(g["batch"]), int(g["seed"]) + 1)
    logits, loss, grads = O.transformer_loss_and_grads(sd, cfg, ids, labels, float(g["label_smoothing"]))
    np.testing.assert_allclose(logits.numpy(), g["logits"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(loss.numpy(), g["loss"], rtol=1e-6)
    for k, v in grads.items():
        ref = g["grad." + k]
        scale = max(float(np.abs(ref).max()), 1e-8)
        assert float(np.abs(v.numpy() - ref).max()) <= 2e-5 * scale + 1e-9, k


@pytest.mark.parametrize("name,cfg", [("transformer_text_tiny", W.TRANSFORMER_TEXT_TINY),
                                      ("transformer_text_proj_tiny", W.TRANSFORMER_TEXT_PROJ_TINY),
                                      ("transformer_plain_tiny", W.TRANSFORMER_PLAIN_TINY),
                                      ("transformer_text_bias_tiny", W.TRANSFORMER_TEXT_BIAS_TINY),      # use_bias=True
                                      ("transformer_rms_bias_tiny", W.TRANSFORMER_RMS_BIAS_TINY)])
def test_transformer_general_forward_backward(golden_dir, name, cfg):
    """the general form of MaskGitTransformer (oracle.transformer_forward_general: cross attention to text states, RMSNorm, plain
    pre-LN layers, projected text states, optional final norm / MLM head) against the REAL reference: logits, loss, every parameter
    gradient, the gradient of the text states, and the condition-dropout pass with the reference's recorded draws"""
    g = _load(golden_dir, name)
    sd = W.fill_state_dict(W.transformer_shapes(cfg), int(g["seed"]), "transformer")
    B = int(g["batch"])
    if cfg.get("add_cross_attention"):
        ids, labels, enc = W.transformer_text_inputs(cfg, B, int(g["text_len"]), int(g["seed"]) + 1)
    else:
        (ids, labels), enc = W.transformer_inputs(cfg, B, int(g["seed"]) + 1), None
    logits, loss, grads, genc = O.transformer_general_loss_and_grads(sd, cfg, ids, labels, enc)
    np.testing.assert_allclose(logits.numpy(), g["logits"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(loss.numpy(), g["loss"], rtol=1e-6)
    assert set("grad." + k for k in grads) == set(f for f in g.files if f.startswith("grad."))
    for k, v in grads.items():
        ref = g["grad." + k]
        if k.endswith(".key.bias"):
            # exactly zero in exact arithmetic (a key bias shifts every score of a query row by the same q.b: softmax does not see
            # it); both sides hold round-off only, measured against the value bias next to it
            floor = 1e-5 * float(np.abs(g["grad." + k.replace(".key.", ".value.")]).max())
            assert float(np.abs(ref).max()) <= floor and float(np.abs(v.numpy()).max()) <= floor, k
            continue
        assert float(np.abs(v.numpy() - ref).max()) <= 2e-5 * max(float(np.abs(ref).max()), 1e-8) + 1e-9, k
    if enc is not None:
        assert float(np.abs(genc.numpy() - g["grad_enc"]).max()) <= 2e-5 * float(np.abs(g["grad_enc"]).max())
        keep = torch.from_numpy(g["cd_u"]) < (1.0 - float(g["cd_p"]))       # prob_mask_like(.., 1 - p): uniform < 1 - p
        assert 0 < int(keep.sum()) < B                                       # (the recorded draws exercise both branches)
        _, loss_d, grads_d, _ = O.transformer_general_loss_and_grads(sd, cfg, ids, labels, enc, label_smoothing=0.1, cond_keep=keep)
        np.testing.assert_allclose(loss_d.numpy(), g["cd_loss"], rtol=1e-6)
        for f in g.files:
            if f.startswith("cd_grad."):
                ref = g[f]
                assert float(np.abs(grads_d[f[8:]].numpy() - ref).max()) <= 2e-5 * float(np.abs(ref).max()) + 1e-9, f


def test_transformer_text_cc12m_width_vs_reference(golden_dir):
    """two layers of configs/cc12m.yaml's transformer (hidden 1024, 16 heads, GLU 4096, T5 states 77 x 1024, 256 tokens, codebook
    8192) through the oracle against the real reference's sub-sampled outputs"""
    torch.set_num_threads(8)
    try:
        g = _load(golden_dir, "transformer_cc12m_2l")
        cfg = W.TRANSFORMER_CC12M_2L
        sd = W.fill_state_dict(W.transformer_shapes(cfg), int(g["seed"]), "transformer")
        ids, labels, enc = W.transformer_text_inputs(cfg, int(g["batch"]), int(g["text_len"]), int(g["seed"]) + 1)
        logits, loss, grads, _ = O.transformer_general_loss_and_grads(sd, cfg, ids, labels, enc)
        assert abs(float(loss) - float(g["loss"])) < 2e-6 * float(g["loss"])
        assert float(np.abs(W.subsample(logits, 16384).numpy() - g["logits"]).max()) < 2e-5 * float(g["logits_absmax"])
        for f in g.files:
            if f.startswith("grad."):
                assert float(np.abs(W.subsample(grads[f[5:]]).numpy() - g[f]).max()) < 5e-5 * float(g["absmax." + f[5:]]), f
    finally:
        torch.set_num_threads(1)


def test_generate2_text_guided(golden_dir):
    """oracle.generate2_text (classifier-free guidance over a doubled batch, with and without negative_embeds) reproduces the real
    reference's generate2 ids from its recorded generator draws"""
    g = _load(golden_dir, "generate2_text_tiny")
    cfg = W.TRANSFORMER_TEXT_TINY
    sd = W.fill_state_dict(W.transformer_shapes(cfg), int(g["seed"]), "transformer")
    _, _, enc = W.transformer_text_inputs(cfg, int(g["batch"]), int(g["text_len"]), int(g["seed"]) + 1)
    T = int(g["timesteps"])
    noise = [(torch.from_numpy(g[f"q{i}"]), torch.from_numpy(g[f"u{i}"])) for i in range(T)]
    for tag, neg in (("", None), ("_neg", torch.from_numpy(g["negative_embeds"]))):
        ids = O.generate2_text(sd, cfg, enc, T, float(g["temperature"]), noise, float(g["guidance_scale"]), neg)
        assert np.array_equal(ids.numpy(), g["ids" + tag]), tag


def test_adamw_step(golden_dir):
    cfg = W.TRANSFORMER_TINY
    g = _load(golden_dir, "transformer_tiny")
    sd = W.fill_state_dict(W.transformer_shapes(cfg), int(g["seed"]), "transformer")
    for k in ("mlm_layer.to_logits.weight", "transformer_layers.0.ffn.wo.weight", "encoder_layer_norm.weight"):
        p = sd[k].clone()
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        O.adamw_step(p, torch.from_numpy(g["grad." + k]), m, v, 1, 1e-4, 0.9, 0.999, 1e-8, 0.01)
        np.testing.assert_allclose(p.numpy(), g["adamw." + k], rtol=0, atol=2e-7)


def test_vqgan_encode_decode(golden_dir):
    cfg = W.VQGAN_TINY
    g = _load(golden_dir, "vqgan_tiny")
    sd = W.fill_state_dict(W.vqgan_shapes(cfg), int(g["seed"]), "vqgan")
    px = W.images(int(g["batch"]), cfg["resolution"], int(g["seed"]) + 1)
    z, z_q, idx = O.vqgan_encode(sd, cfg, px)
    np.testing.assert_allclose(z.numpy(), g["z"], rtol=1e-5, atol=1e-5)
    assert np.array_equal(idx.numpy(), g["indices"])  # bit-exact token indices
    np.testing.assert_allclose(z_q.numpy(), g["z_q"], rtol=0, atol=0)
    rec = O.vqgan_decode_code(sd, cfg, idx)
    np.testing.assert_allclose(rec.numpy(), g["rec"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name,cfg", [("taming_tiny", W.TAMING_TINY), ("taming_tiny_pool", W.TAMING_TINY_POOL)])
def test_taming_vqgan_encode_decode(golden_dir, name, cfg):
    """the taming `VQGANModel` restatement (oracle/taming_oracle.py) against the real reference
    (golden_taming): encoder output, quant_conv latents, bit-exact indices / z_q, both decode entry points"""
    from oracle import taming_oracle as T
    g = _load(golden_dir, name)
    sd = W.fill_state_dict(W.taming_shapes(cfg), int(g["seed"]), "vqgan")
    px = W.images(int(g["batch"]), cfg["resolution"], int(g["seed"]) + 1)
    np.testing.assert_allclose(T.encoder(sd, cfg, px).numpy(), g["enc"], rtol=1e-5, atol=1e-5)
    z, z_q, idx = T.encode(sd, cfg, px)
    np.testing.assert_allclose(z.numpy(), g["z"], rtol=1e-5, atol=1e-5)
    assert np.array_equal(idx.numpy(), g["indices"])
    np.testing.assert_allclose(z_q.numpy(), g["z_q"], rtol=0, atol=0)
    np.testing.assert_allclose(T.decode_code(sd, cfg, idx).numpy(), g["rec"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(T.decode(sd, cfg, z_q).numpy(), g["rec_decode"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name", ["mask_b64", "mask_small"])
def test_mask_sampling(golden_dir, name):
    g = _load(golden_dir, name)
    ids, labels, prob = O.prepare_inputs_and_labels(
        torch.from_numpy(g["image_tokens"]), torch.from_numpy(g["class_ids"]), torch.from_numpy(g["timesteps"]),
        torch.from_numpy(g["noise"]), int(g["mask_id"]), int(g["codebook_size"]), float(g["min_rate"]))
    assert np.array_equal(ids.numpy(), g["input_ids"])      # bit-exact mask indices
    assert np.array_equal(labels.numpy(), g["labels"])
    np.testing.assert_array_equal(prob.numpy(), g["mask_prob"])


@pytest.mark.parametrize("name", ["uvit_tiny", "uvit_tiny_noaffine", "uvit_tiny_layernorm", "uvit_tiny_downup"])
def test_uvit_oracle_vs_reference_golden(golden_dir, name):
    """SURVEY.md section 8 row a12 (MaskGiTUViT_v2, config 4): logits, plain / smoothed+weighted loss and EVERY parameter
    gradient of the CPU restatement against the real reference (tests/golden/make_golden.py::golden_uvit); `uvit_tiny_noaffine`:
    ln_elementwise_affine=False, norms without learnable gains (:656-660: their weights are absent from the state dict); `uvit_tiny_downup`:
    force_down_up_sample=True on an 8 x 8 token grid (stride-2 conv / transposed conv around the blocks, :510-514 / :558-562)"""
    import json
    from oracle import uvit_oracle as U
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg = json.load(open(os.path.join(golden_dir, "config_" + name + ".json")))
    if name == "uvit_tiny_noaffine":
        assert not any(k.endswith("norm.weight") for k in g.files)
    sd = {k[len("param."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("param.")}
    args = [torch.from_numpy(g[k]) for k in ("input_ids", "encoder_hidden_states", "cond_embeds", "micro_conds")]
    labels = torch.from_numpy(g["labels"])
    logits, loss, grads = U.uvit_loss_and_grads(sd, cfg, *args, labels)
    ref_logits = torch.from_numpy(g["logits"])
    assert float(ref_logits.abs().max()) > 0.5          # the perturbed init really exercises the network
    assert torch.allclose(logits, ref_logits, rtol=1e-5, atol=1e-5 * float(ref_logits.abs().max()))
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    checked = 0
    for k in g.files:
        if k.startswith("grad."):
            ref = torch.from_numpy(g[k])
            got = grads[k[len("grad."):]]
            assert got.shape == ref.shape
            assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-9, k
            checked += 1
    assert checked == len(sd)                           # no bias-free buffers: every state-dict entry is a parameter
    _, loss_w = U.uvit_forward(sd, cfg, *args, labels=labels, label_smoothing=float(g["label_smoothing"]),
                               loss_weight=torch.from_numpy(g["loss_weight"]))
    assert abs(float(loss_w) - float(g["loss_weighted"])) < 1e-5 * abs(float(g["loss_weighted"]))
    # inference call signature: logits only
    assert U.uvit_forward(sd, cfg, *args).shape == ref_logits.shape


# ---- "next" rows f2 / f3: decoding loop and train_muse masking, oracle vs the real reference -------------------------------
def _noise(g, steps):
    return [(torch.from_numpy(g[f"q{i}"]), torch.from_numpy(g[f"u{i}"])) for i in range(steps)]


def test_generate2_oracle_vs_reference(golden_dir):
    """MaskGitTransformer.generate2 of the real reference (seeded CPU generator) == the oracle fed with the replayed draws"""
    g = _load(golden_dir, "generate2_tiny")
    cfg = W.TRANSFORMER_TINY
    sd = W.fill_state_dict(W.transformer_shapes(cfg), int(g["seed"]), "transformer")
    T = int(g["timesteps"])
    with torch.no_grad():
        ids, fed = O.generate2(sd, cfg, torch.from_numpy(g["class_ids"]), T, float(g["temperature"]), _noise(g, T))
    assert np.array_equal(ids.numpy(), g["ids"])
    for i in range(T):
        assert np.array_equal(fed[i].numpy(), g[f"fed{i}"])
    mask_id = cfg["vocab_size"] - 1
    masked = [int((g[f"fed{i}"] == mask_id).sum()) for i in range(T)]
    assert masked[0] == g["ids"].size and all(a > b for a, b in zip(masked, masked[1:]))   # the schedule really unmasks step by step


def test_uvit_generate2_oracle_vs_reference(golden_dir):
    import json
    from oracle import uvit_oracle as U
    g = _load(golden_dir, "uvit_generate2_tiny")
    gp = np.load(os.path.join(golden_dir, "uvit_tiny.npz"))
    cfg = json.load(open(os.path.join(golden_dir, "config_uvit_tiny.json")))
    sd = {k[len("param."):]: torch.from_numpy(gp[k]) for k in gp.files if k.startswith("param.")}
    T = int(g["timesteps"])
    t = tuple(float(x) for x in g["temperature"])
    with torch.no_grad():
        ids, inter = U.generate2(sd, cfg, *(torch.from_numpy(g[k]) for k in ("encoder_hidden_states", "cond_embeds", "micro_conds",
                                                                              "empty_embeds", "empty_cond_embeds")),
                                 T, t, float(g["guidance_scale"]), _noise(g, T), int(g["seq"]))
    assert np.array_equal(ids.numpy(), g["ids"])
    for i in range(T):
        assert np.array_equal(inter[i].numpy(), g[f"raw{i}"])


@pytest.mark.parametrize("case", ["default", "predict_all", "random_replace", "region", "eval_ratios"])
def test_mask_or_random_replace_tokens_oracle(golden_dir, case):
    g = _load(golden_dir, "mask_muse")
    tokens = torch.from_numpy(g["tokens"])
    kw = dict(all_labels=case in ("predict_all", "random_replace"))
    if case == "region":
        kw.update(timesteps=torch.from_numpy(g[case + ".timesteps"]), rects=torch.from_numpy(g["region.rects"]))
    elif case == "eval_ratios":
        kw.update(mask_prob=torch.from_numpy(g[case + ".mask_prob"]), noise=torch.from_numpy(g[case + ".noise"]))
    else:
        kw.update(timesteps=torch.from_numpy(g[case + ".timesteps"]), noise=torch.from_numpy(g[case + ".noise"]))
    ids, labels, lw, mp = O.mask_or_random_replace_tokens(tokens, int(g["mask_id"]), float(g[case + ".min_rate"]), **kw)
    assert np.array_equal(ids.numpy(), g[case + ".input_ids"]) and np.array_equal(labels.numpy(), g[case + ".labels"])
    np.testing.assert_array_equal(mp.numpy(), g[case + ".mask_prob"])
    if kw["all_labels"]:
        np.testing.assert_array_equal(lw.numpy(), g[case + ".loss_weight"])
    else:
        assert lw is None and (case + ".loss_weight") not in g.files


def test_cond_dropout_oracle(golden_dir):
    g = _load(golden_dir, "mask_muse")
    u, p = torch.from_numpy(g["cd.u"]), float(g["cd.prob"])
    enc = O.cond_dropout(torch.from_numpy(g["cd.enc"]), torch.from_numpy(g["cd.empty"]), u, p)
    clip = O.cond_dropout(torch.from_numpy(g["cd.clip"]), torch.from_numpy(g["cd.empty_clip"]), u, p)
    np.testing.assert_array_equal(enc.numpy(), g["cd.enc_out"])
    np.testing.assert_array_equal(clip.numpy(), g["cd.clip_out"])
    kept = (g["cd.u"] < p)
    assert kept.any() and (~kept).any()
    assert np.array_equal(g["cd.enc_out"][1, 0, :4], g["cd.empty"][0, 0, :4])    # kept image, exact-zero elements -> empty's values


def test_oracle_at_benched_geometry_vs_reference(golden_dir):
    """The restatement against the REAL reference at the benched sizes (make_golden.py::golden_transformer_full / golden_vqgan_full):
    configs/imagenet.yaml's 24-layer transformer at batch 2, S = 257 (loss, sub-sampled logits, ten gradients over the depth), and
    the f16-256 tokenizer on one 256 x 256 image (encoder output, bit-exact token ids and codes, reconstruction)."""
    torch.set_num_threads(min(8, os.cpu_count()))
    try:
        g = _load(golden_dir, "transformer_b_full")
        cfg = dict(W.TRANSFORMER_B)
        sd = W.fill_state_dict(W.transformer_shapes(cfg), int(g["seed"]), "transformer")
        ids, labels = W.transformer_inputs(cfg, int(g["batch"]), int(g["seed"]) + 1)
        logits, loss, grads = O.transformer_loss_and_grads(sd, cfg, ids, labels, 0.0)
        np.testing.assert_allclose(loss.numpy(), g["loss"], rtol=1e-6)
        assert float(np.abs(W.subsample(logits, 16384).numpy() - g["logits"]).max()) <= 1e-5 * float(g["logits_absmax"])
        for k in [f[5:] for f in g.files if f.startswith("grad.")]:
            e = float(np.abs(W.subsample(grads[k]).numpy() - g["grad." + k]).max())
            assert e <= 2e-5 * float(g["absmax." + k]), (k, e)
            assert abs(float(grads[k].double().norm()) - float(g["norm." + k])) <= 1e-5 * float(g["norm." + k]), k
        del grads, logits, sd
        gv = _load(golden_dir, "vqgan_f16_full")
        sdv = W.fill_state_dict(W.vqgan_shapes(W.VQGAN_F16), int(gv["seed"]), "vqgan")
        px = W.images(1, 256, int(gv["seed"]) + 1)
        with torch.no_grad():
            z, z_q, idx = O.vqgan_encode(sdv, W.VQGAN_F16, px)
            rec = O.vqgan_decode_code(sdv, W.VQGAN_F16, idx)
        np.testing.assert_allclose(z.numpy(), gv["z"], rtol=0, atol=2e-5)
        assert np.array_equal(idx.numpy(), gv["indices"])                      # (smallest top-2 distance margin: 6.4e-3)
        assert np.array_equal(W.subsample(z_q).numpy(), gv["z_q"])
        assert float(np.abs(W.subsample(rec, 16384).numpy() - gv["rec"]).max()) <= 1e-4 * float(gv["rec_absmax"])
    finally:
        torch.set_num_threads(1)


def test_uvit_oracle_at_config4_vs_reference(golden_dir):
    """the U-ViT restatement against the REAL reference at BASELINE config 4's size (728.7 M parameters; make_golden.py::
    golden_uvit_full): loss, sub-sampled logits and the eighteen stored gradients of a 2 x 256-token batch (the restatement runs the
    same torch CPU operators in the same order: measured difference 0)"""
    from oracle import uvit_oracle as U
    g = _load(golden_dir, "uvit_full")
    torch.set_num_threads(min(8, os.cpu_count()))
    try:
        import muse
        from muse import modeling_transformer_v2 as M
        init = M.MaskGiTUViT_v2._init_weights
        M.MaskGiTUViT_v2._init_weights = lambda self: None
        try:
            model = muse.MaskGiTUViT(**W.UVIT_CC12M)          # (only for the state-dict template and the config dict)
        finally:
            M.MaskGiTUViT_v2._init_weights = init
        shapes, cfg = {k: tuple(v.shape) for k, v in model.state_dict().items()}, dict(model.config)
        del model
        sd = W.fill_by_shapes(shapes, int(g["seed"]))
        ids, enc, cond, micro, labels = W.uvit_inputs(int(g["batch"]), int(g["seq"]), int(g["text_len"]), int(g["seed"]) + 1)
        logits, loss, grads = U.uvit_loss_and_grads(sd, cfg, ids, enc, cond, micro, labels)
        assert abs(float(loss) - float(g["loss"])) <= 1e-6 * float(g["loss"])
        assert float(np.abs(W.subsample(logits, 16384).numpy() - g["logits"]).max()) <= 1e-5 * float(g["logits_absmax"])
        for k in W.UVIT_FULL_GRAD_KEYS:
            assert float(np.abs(W.subsample(grads[k]).numpy() - g["grad." + k]).max()) <= 2e-5 * float(g["absmax." + k]), k
    finally:
        torch.set_num_threads(1)


# ---- the weight average behind the optimizer step (muse/modeling_ema.py; training/train_muse.py:779-780) -----------------------------------
def test_ema_oracle_vs_reference_golden(golden_dir):
    """the numpy restatement of EMAModel.get_decay / step against what the REAL reference class produced (tests/golden/ema_tiny.npz,
    make_golden.py::golden_ema): two schedules (update_after_step; warmup + min_decay + update_every 2), 14 calls on changing
    parameters, a frozen tensor among them - decays equal, every shadow tensor BIT-identical after every call"""
    from oracle import ema_oracle as E
    g = np.load(os.path.join(golden_dir, "ema_tiny.npz"))
    seed, steps = int(g["seed"]), int(g["steps"])
    rg = [i != 4 for i in range(len(W.EMA_SHAPES))]
    for si, kw in enumerate(W.EMA_SCHEDULES):
        sched = E.Schedule(**kw)
        shadow = [t.numpy().copy() for t in W.ema_params(seed, 0)]
        used = 0
        for step in range(1, steps + 1):
            params = [t.numpy() for t in W.ema_params(seed, step)]
            decay = sched.next()
            if decay is None:
                assert float(g[f"s{si}.decay{step}"]) == -1.0
            else:
                assert decay == float(g[f"s{si}.decay{step}"])
                shadow = E.ema_update(shadow, params, rg, decay)
                used += 1
            for i, sh in enumerate(shadow):
                assert np.array_equal(sh, g[f"s{si}.shadow{step}.{i}"]), (si, step, i)
        assert used == (steps if kw.get("update_every", 1) == 1 else steps // 2)



def test_reference_tf32_regime_golden_sits_tf32_class_from_its_f32_run():
    """tests/golden/uvit_full_tf32emu.npz (make_golden_tf32.py): the REAL reference at config 4's full size with the operands of every
    matmul rounded to TF32's 10-bit mantissa - the `enable_tf32` regime of configs/cc12m_uvit_clip.yaml:102-103 emulated on the CPU -
    next to its plain f32 run (uvit_full.npz).  The gap between the two is what "the YAML's precision class" means in numbers; the
    f16 compute mode of this package is held to it on the GPU (test_uvit_config4_vs_reference_golden)."""
    import numpy as np
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g, t = np.load(os.path.join(here, "uvit_full.npz")), np.load(os.path.join(here, "uvit_full_tf32emu.npz"))
    assert int(g["seed"]) == int(t["seed"]) and tuple(g["logits_shape"]) == tuple(t["logits_shape"])
    keys = [k[5:] for k in g.files if k.startswith("grad.")]
    el = float(np.abs(t["logits"] - g["logits"]).max()) / float(g["logits_absmax"])
    eg = max(float(np.abs(t["grad." + k] - g["grad." + k]).max()) / float(g["absmax." + k]) for k in keys)
    lrel = abs(float(t["loss"]) - float(g["loss"])) / float(g["loss"])
    print(f"reference under emulated TF32 vs its f32 run: logits {el:.2e}, loss {lrel:.1e}, worst gradient {eg:.1e}")
    assert 3e-4 < el < 3e-3 and 5e-4 < eg < 6e-3 and lrel < 1e-4      # TF32 class: far from f32-exact (1e-6), far from bf16 (1e-2)
