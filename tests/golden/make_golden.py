"""Generate golden vectors by running the REAL reference (/root/reference, importable only in the build
container) on seeded inputs.  Output: tests/golden/*.npz (committed).  Re-run:

    python tests/golden/make_golden.py

The reference ships no golden vectors for this path, so these are the pins for oracle/.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")

import weights as W  # noqa: E402

import muse as ref_muse  # noqa: E402  (the reference package)
from muse.sampling import cosine_schedule  # noqa: E402

torch.set_num_threads(1)  # one reduction order, reproducible


def np_(t):
    return t.detach().cpu().numpy()


def golden_transformer(name, cfg, batch, seed, label_smoothing):
    model = ref_muse.MaskGitTransformer(**cfg)
    sd = W.fill_state_dict(W.transformer_shapes(cfg), seed, "transformer")
    missing = model.load_state_dict(sd, strict=True)
    model.train()  # dropout p=0 -> identity; same branch the training script takes
    input_ids, labels = W.transformer_inputs(cfg, batch, seed + 1)
    logits, loss = model(input_ids=input_ids, labels=labels, label_smoothing=label_smoothing)
    loss.backward()
    out = dict(logits=np_(logits), loss=np_(loss), label_smoothing=np.float32(label_smoothing),
               batch=np.int64(batch), seed=np.int64(seed))
    for k, p in model.named_parameters():
        out["grad." + k] = np_(p.grad)
    # one AdamW step with the target config's hyper-parameters (configs/imagenet.yaml:64-72)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.999), weight_decay=0.01, eps=1e-8)
    opt.step()
    for k in ("mlm_layer.to_logits.weight", "transformer_layers.0.ffn.wo.weight", "encoder_layer_norm.weight"):
        out["adamw." + k] = np_(dict(model.named_parameters())[k])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "loss", float(loss), "logits", logits.shape)


def golden_transformer_autocast(name, cfg, batch, seed):
    """the same forward / backward under torch.autocast("cpu", bfloat16) - the regime accelerate's mixed_precision: bf16 puts the
    reference in (training/train_maskgit_imagenet.py:152-158; SURVEY.md section 3.2 dtype flow).  It pins how far the reference's
    OWN bf16 path sits from its f32 path, which is what the HIP bf16 mode's tolerances are derived from."""
    model = ref_muse.MaskGitTransformer(**cfg)
    model.load_state_dict(W.fill_state_dict(W.transformer_shapes(cfg), seed, "transformer"), strict=True)
    model.train()
    input_ids, labels = W.transformer_inputs(cfg, batch, seed + 1)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        logits, loss = model(input_ids=input_ids, labels=labels)
    loss.float().backward()
    out = dict(logits=np_(logits.float()), loss=np_(loss.float()), batch=np.int64(batch), seed=np.int64(seed))
    for k, p in model.named_parameters():
        out["grad." + k] = np_(p.grad.float())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "loss", float(loss), "logits dtype", logits.dtype)


def golden_vqgan(name, cfg, batch, seed):
    model = ref_muse.MaskGitVQGAN(**cfg)
    sd = W.fill_state_dict(W.vqgan_shapes(cfg), seed, "vqgan")
    model.load_state_dict(sd, strict=True)
    model.eval()
    px = W.images(batch, cfg["resolution"], seed + 1)
    with torch.no_grad():
        z = model.encoder(px)
        z_q, idx = model.encode(px)
        rec = model.decode_code(idx)
        code = model.get_code(px)
        dist = model.quantize.compute_distances(z.permute(0, 2, 3, 1).contiguous())
    assert torch.equal(code, idx)
    top2 = torch.topk(dist, 2, dim=1, largest=False).values
    out = dict(z=np_(z), z_q=np_(z_q), indices=np_(idx), rec=np_(rec), margin=np_(top2[:, 1] - top2[:, 0]),
               batch=np.int64(batch), seed=np.int64(seed))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "z", z.shape, "idx", idx.shape, "rec", rec.shape, "min margin", float(out["margin"].min()))


FULL_GRAD_KEYS = ["embed.word_embeddings.weight", "embed.position_embeddings.weight", "transformer_layers.0.attention.query.weight",
                  "transformer_layers.0.attn_layer_norm.weight", "transformer_layers.11.ffn.wi_1.weight",
                  "transformer_layers.12.attention.out.weight", "transformer_layers.23.ffn.wo.weight",
                  "transformer_layers.23.post_attn_layer_norm.weight", "encoder_layer_norm.weight", "mlm_layer.to_logits.weight"]


def golden_transformer_full(name, cfg, batch, seed, autocast=False):
    """the BENCHED transformer (configs/imagenet.yaml: 24 layers, hidden 768, 16 heads of 48, vocab 2048, S = 257) run by the real
    reference, f32 (and under CPU autocast-bf16).  Outputs are large (logits 2 x 257 x 2048, 230 M gradient elements), so the fixture
    keeps the loss, per-tensor max|.| and L2 norm, and every k-th element (weights.subsample) of the logits and of ten gradients
    spread over the depth of the stack."""
    model = ref_muse.MaskGitTransformer(**cfg)
    model.load_state_dict(W.fill_state_dict(W.transformer_shapes(cfg), seed, "transformer"), strict=True)
    model.train()
    input_ids, labels = W.transformer_inputs(cfg, batch, seed + 1)
    if autocast:
        with torch.autocast("cpu", dtype=torch.bfloat16):
            logits, loss = model(input_ids=input_ids, labels=labels)
        logits, loss = logits.float(), loss.float()
    else:
        logits, loss = model(input_ids=input_ids, labels=labels)
    loss.backward()
    out = dict(loss=np_(loss), batch=np.int64(batch), seed=np.int64(seed), logits=np_(W.subsample(logits, 16384)),
               logits_absmax=np_(logits.abs().max()), logits_norm=np_(logits.double().norm()))
    params = dict(model.named_parameters())
    for k in FULL_GRAD_KEYS:
        g = params[k].grad.float()
        out["grad." + k] = np_(W.subsample(g))
        out["absmax." + k] = np_(g.abs().max())
        out["norm." + k] = np_(g.double().norm())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "loss", float(loss), "logits absmax", float(logits.abs().max()))


def golden_transformer_full_chunked(name, cfg, batch, chunk, seed):
    """the benched transformer at the BENCHED BATCH (64 images, T = 16448 rows: the weight-gradient split-K plans, the 1560-tile GEMM
    grids and the ~8 k hits on the mask token's embedding row only exist at this size).  One reference pass at batch 64 needs more
    host memory than the build container has (every layer keeps its f32 attention scores and GLU intermediates), so the real
    reference runs the batch in chunks of `chunk` images: logits are per image; its loss is the mean over the masked positions, so
    loss = sum_c n_c loss_c / N and grad = sum_c (n_c / N) grad_c, recombined here in f64 (differs from one batch-64 pass by f32
    summation order only)."""
    model = ref_muse.MaskGitTransformer(**cfg)
    model.load_state_dict(W.fill_state_dict(W.transformer_shapes(cfg), seed, "transformer"), strict=True)
    model.train()
    input_ids, labels = W.transformer_inputs(cfg, batch, seed + 1)
    n_total = int((labels != -100).sum())
    params = dict(model.named_parameters())
    acc = {k: torch.zeros(params[k].shape, dtype=torch.float64) for k in FULL_GRAD_KEYS}
    loss_sum, logit_parts, sq, amax = 0.0, [], 0.0, 0.0
    for c0 in range(0, batch, chunk):
        ids_c, lab_c = input_ids[c0:c0 + chunk], labels[c0:c0 + chunk]
        n_c = int((lab_c != -100).sum())
        model.zero_grad(set_to_none=True)
        logits, loss = model(input_ids=ids_c, labels=lab_c)
        loss.backward()
        loss_sum += float(loss.double()) * n_c
        for k in FULL_GRAD_KEYS:
            acc[k] += params[k].grad.double() * (n_c / n_total)
        logit_parts.append(logits.detach().reshape(-1)[:: 4096].clone())      # every 4096-th logit of the chunk
        sq += float(logits.double().pow(2).sum())
        amax = max(amax, float(logits.abs().max()))
        print(name, "chunk", c0 // chunk, "loss", float(loss), flush=True)
    out = dict(loss=np.float64(loss_sum / n_total), batch=np.int64(batch), chunk=np.int64(chunk), seed=np.int64(seed),
               n_masked=np.int64(n_total), logits=np_(torch.cat(logit_parts)), logits_stride=np.int64(4096),
               logits_absmax=np.float64(amax), logits_norm=np.float64(sq ** 0.5))
    for k in FULL_GRAD_KEYS:
        g = acc[k]
        out["grad." + k] = np_(W.subsample(g).float())
        out["absmax." + k] = np_(g.abs().max())
        out["norm." + k] = np_(g.norm())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "loss", loss_sum / n_total, "masked", n_total)


def golden_transformer_text(name, cfg, batch, text_len, seed, cond_p=0.5):
    """the general form of the reference's MaskGitTransformer (text states through cross attention, RMSNorm, plain pre-LN layers,
    optional projection / final norm / MLM head; muse/modeling_transformer.py:1083-1281) on seeded inputs: logits, loss, every
    parameter gradient, the gradient of the text states, and a second pass with condition dropout (prob_mask_like's uniform draws
    recorded: `uniform` of the reference module is replaced for that call)."""
    import muse.modeling_transformer as ref_mt
    model = ref_muse.MaskGitTransformer(**cfg)
    sd = W.fill_state_dict(W.transformer_shapes(cfg), seed, "transformer")
    model.load_state_dict(sd, strict=True)
    model.train()
    text = bool(cfg.get("add_cross_attention", False))
    if text:
        input_ids, labels, enc = W.transformer_text_inputs(cfg, batch, text_len, seed + 1)
        enc.requires_grad_(True)
        logits, loss = model(input_ids=input_ids, encoder_hidden_states=enc, labels=labels)
    else:
        input_ids, labels = W.transformer_inputs(cfg, batch, seed + 1)
        enc = None
        logits, loss = model(input_ids=input_ids, labels=labels)
    loss.backward()
    out = dict(logits=np_(logits), loss=np_(loss), batch=np.int64(batch), seed=np.int64(seed), text_len=np.int64(text_len))
    for k, p in model.named_parameters():
        out["grad." + k] = np_(p.grad)
    if text:
        out["grad_enc"] = np_(enc.grad)
        # condition dropout (:1243-1247) with label smoothing; the draws are the file's
        u = torch.from_numpy(np.random.default_rng(seed + 2).random((batch, 1, 1)).astype(np.float32))
        real_uniform = ref_mt.uniform
        ref_mt.uniform = lambda shape, min=0, max=1, device=None: u.clone()
        try:
            model.zero_grad()
            _, loss_d = model(input_ids=input_ids, encoder_hidden_states=enc.detach(), labels=labels, label_smoothing=0.1,
                              cond_dropout_prob=cond_p)
            loss_d.backward()
        finally:
            ref_mt.uniform = real_uniform
        out.update(cd_u=np_(u.reshape(batch)), cd_p=np.float32(cond_p), cd_loss=np_(loss_d))
        for k in ("transformer_layers.0.crossattention.key.weight", "transformer_layers.1.attention.out.weight" if cfg["num_hidden_layers"] > 1
                  else "transformer_layers.0.attention.out.weight", "embed.word_embeddings.weight"):
            out["cd_grad." + k] = np_(dict(model.named_parameters())[k].grad)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "loss", float(loss), "logits", tuple(logits.shape))


def golden_transformer_text_sub(name, cfg, batch, text_len, seed):
    """the same at the width of configs/cc12m.yaml (2 of its 24 layers, 256 tokens, 77 text states of width 1024): loss, sub-sampled
    logits and gradients with their norms, f32 and under CPU autocast-bf16"""
    input_ids, labels, enc = W.transformer_text_inputs(cfg, batch, text_len, seed + 1)
    for tag, autocast in (("", False), ("_bf16", True)):
        model = ref_muse.MaskGitTransformer(**cfg)
        model.load_state_dict(W.fill_state_dict(W.transformer_shapes(cfg), seed, "transformer"), strict=True)
        model.train()
        if autocast:
            with torch.autocast("cpu", dtype=torch.bfloat16):
                logits, loss = model(input_ids=input_ids, encoder_hidden_states=enc, labels=labels)
            logits, loss = logits.float(), loss.float()
        else:
            logits, loss = model(input_ids=input_ids, encoder_hidden_states=enc, labels=labels)
        loss.backward()
        out = dict(loss=np_(loss), batch=np.int64(batch), seed=np.int64(seed), text_len=np.int64(text_len),
                   logits=np_(W.subsample(logits, 16384)), logits_absmax=np_(logits.abs().max()))
        for k, p in model.named_parameters():
            if any(t in k for t in ("layers.0.crossattention", "layers.1.attention.query", "layers.1.ffn.wo", "layers.0.ffn.wi_0",
                                    "layers.1.crossattn_layer_norm", "encoder_layer_norm", "word_embeddings", "mlm_layer.to_logits")):
                g = p.grad.float()
                out["grad." + k], out["absmax." + k], out["norm." + k] = np_(W.subsample(g)), np_(g.abs().max()), np_(g.double().norm())
        np.savez_compressed(os.path.join(HERE, name + tag + ".npz"), **out)
        print(name + tag, "loss", float(loss))


def golden_generate2_text(name, cfg, batch, text_len, seed, timesteps, temperature, guidance_scale):
    """MaskGitTransformer.generate2 of the real reference with text states and classifier-free guidance (:1394-1416), seeded CPU
    generator; negative_embeds given for half of the cases (file `..._neg`)"""
    from oracle import maskgit_oracle as O
    model = ref_muse.MaskGitTransformer(**cfg)
    sd = W.fill_state_dict(W.transformer_shapes(cfg), seed, "transformer")
    model.load_state_dict(sd, strict=True)
    model.eval()
    _, _, enc = W.transformer_text_inputs(cfg, batch, text_len, seed + 1)
    neg = torch.from_numpy(np.random.default_rng(seed + 3).standard_normal(tuple(enc.shape)).astype(np.float32))
    S, V = cfg["num_vq_tokens"], cfg["codebook_size"]
    out = dict(batch=np.int64(batch), seed=np.int64(seed), text_len=np.int64(text_len), timesteps=np.int64(timesteps),
               temperature=np.float32(temperature), guidance_scale=np.float32(guidance_scale), negative_embeds=np_(neg))
    for tag, negative in (("", None), ("_neg", neg)):
        with torch.no_grad():
            ids = model.generate2(encoder_hidden_states=enc, negative_embeds=negative, timesteps=timesteps, temperature=temperature,
                                  guidance_scale=guidance_scale, generator=torch.Generator().manual_seed(seed + 7))
        noise = replay_decode_noise(seed + 7, timesteps, batch, S, V)
        ids_o = O.generate2_text(sd, cfg, enc, timesteps, temperature, noise, guidance_scale, negative)
        assert torch.equal(ids, ids_o), "the recorded draws do not reproduce the reference's sample"
        out["ids" + tag] = np_(ids)
        if tag == "":
            for i, (q, u) in enumerate(noise):
                out[f"q{i}"], out[f"u{i}"] = np_(q), np_(u)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "ids", out["ids"][0, :8], out["ids_neg"][0, :8])


def golden_vqgan_full(name, cfg, seed):
    """the f16-256 tokenizer (54.5 M parameters) run by the real reference on one 256 x 256 image: encoder output z (full), the 256
    token ids, the two smallest distances per token (near-tie margins), and every k-th pixel of decode_code's reconstruction"""
    model = ref_muse.MaskGitVQGAN(**cfg)
    model.load_state_dict(W.fill_state_dict(W.vqgan_shapes(cfg), seed, "vqgan"), strict=True)
    model.eval()
    px = W.images(1, cfg["resolution"], seed + 1)
    with torch.no_grad():
        z = model.encoder(px)
        z_q, idx = model.encode(px)
        rec = model.decode_code(idx)
        dist = model.quantize.compute_distances(z.permute(0, 2, 3, 1).contiguous())
    top2 = torch.topk(dist, 2, dim=1, largest=False).values
    out = dict(z=np_(z), indices=np_(idx), rec=np_(W.subsample(rec, 16384)), rec_absmax=np_(rec.abs().max()),
               z_q=np_(W.subsample(z_q)), top2=np_(top2), seed=np.int64(seed))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "z", z.shape, "idx", idx.shape, "rec", rec.shape, "min margin", float((top2[:, 1] - top2[:, 0]).min()))


def golden_uvit_full(name, batch, seq, text_len, seed, autocast=False):
    """BASELINE.json config 4 (weights.UVIT_CC12M = configs/cc12m_uvit_clip.yaml's model.transformer + block_num_heads=16, SURVEY.md D3:
    hidden 1024, 22 layers, GLU 4096, 3 + 3 ResBlock / attention stages of 1024 channels, in_channels 512, vocab 8256; 728.7 M
    parameters) run by the real reference on one seeded batch, f32 and under CPU autocast-bf16.  Stored like golden_transformer_full: loss, per-tensor max|.| and
    L2 norm, every k-th element of the logits and of eighteen gradients spread over the network."""
    from muse.modeling_transformer_v2 import MaskGiTUViT_v2
    model = MaskGiTUViT_v2(**W.UVIT_CC12M)
    assert sum(p.numel() for p in model.parameters()) == 728725504
    model.load_state_dict(W.fill_by_shapes({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed), strict=True)
    model.train()
    ids, enc, cond, micro, labels = W.uvit_inputs(batch, seq, text_len, seed + 1)
    if autocast:
        with torch.autocast("cpu", dtype=torch.bfloat16):
            logits, loss = model(ids, enc, cond, micro, labels=labels)
        logits, loss = logits.float(), loss.float()
    else:
        logits, loss = model(ids, enc, cond, micro, labels=labels)
    loss.backward()
    out = dict(loss=np_(loss), batch=np.int64(batch), seq=np.int64(seq), text_len=np.int64(text_len), seed=np.int64(seed),
               logits=np_(W.subsample(logits, 16384)), logits_absmax=np_(logits.abs().max()), logits_norm=np_(logits.double().norm()),
               logits_shape=np.array(logits.shape, dtype=np.int64))
    params = dict(model.named_parameters())
    for k in W.UVIT_FULL_GRAD_KEYS:
        g = params[k].grad.float()
        out["grad." + k] = np_(W.subsample(g))
        out["absmax." + k] = np_(g.abs().max())
        out["norm." + k] = np_(g.double().norm())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "loss", float(loss), "logits", tuple(logits.shape), "absmax", float(logits.abs().max()))


def golden_taming(name, cfg, batch, seed):
    """the taming tokenizer of the text-to-image configs (muse/modeling_taming_vqgan.py:512-585, configs/cc12m_uvit_clip.yaml
    :19-21): encoder latents, quant_conv output, indices, z_q, reconstruction, and the top-2 distance margin per token"""
    from muse.modeling_taming_vqgan import VQGANModel
    model = VQGANModel(**cfg)
    sd = W.fill_state_dict(W.taming_shapes(cfg), seed, "vqgan")
    model.load_state_dict(sd, strict=True)
    model.eval()
    px = W.images(batch, cfg["resolution"], seed + 1)
    with torch.no_grad():
        enc = model.encoder(px)
        z = model.quant_conv(enc)
        z_q, idx = model.encode(px)
        rec = model.decode_code(idx)
        rec2 = model.decode(z_q)
        code = model.get_code(px)
        dist = model.quantize.compute_distances(z.permute(0, 2, 3, 1).contiguous())
    assert torch.equal(code, idx)
    # (decode_code feeds a permuted view, decode a contiguous tensor: same math, the CPU convolution may round differently)
    print(name, "decode_code vs decode(z_q): max abs diff", float((rec - rec2).abs().max()))
    top2 = torch.topk(dist, 2, dim=1, largest=False).values
    out = dict(enc=np_(enc), z=np_(z), z_q=np_(z_q), indices=np_(idx), rec=np_(rec), rec_decode=np_(rec2),
               margin=np_(top2[:, 1] - top2[:, 0]),
               batch=np.int64(batch), seed=np.int64(seed))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    with open(os.path.join(HERE, "config_" + name + ".json"), "w") as f:
        f.write(model.to_json_string())   # what the reference's save_pretrained writes as config.json
    print(name, "z", z.shape, "idx", idx.shape, "rec", rec.shape, "min margin", float(out["margin"].min()),
          "distinct codes", len(set(idx.flatten().tolist())))


def golden_mask(name, batch, seq, seed, mask_id, codebook_size, min_rate):
    """training/train_maskgit_imagenet.py:371-394 executed line by line with supplied uniforms."""
    rng = np.random.default_rng(seed)
    image_tokens = torch.from_numpy(rng.integers(0, codebook_size, size=(batch, seq)).astype(np.int64))
    class_ids = torch.from_numpy(rng.integers(0, 1000, size=(batch,)).astype(np.int64))
    timesteps = W.uniforms((batch,), seed + 1)
    noise = W.uniforms((batch, seq), seed + 2)
    # --- reference lines (:376-393), rand draws replaced by the tensors above ---
    mask_prob = cosine_schedule(timesteps)
    mask_prob = mask_prob.clip(min_rate)
    num_token_masked = (seq * mask_prob).round().clamp(min=1)
    batch_randperm = noise.argsort(dim=-1)
    mask = batch_randperm < num_token_masked.unsqueeze(-1)
    input_ids = torch.where(mask, mask_id, image_tokens)
    labels = torch.where(mask, image_tokens, -100)
    class_ids_s = class_ids + codebook_size
    input_ids = torch.cat([class_ids_s.unsqueeze(-1), input_ids], dim=-1)
    labels_mask = torch.ones_like(class_ids_s).unsqueeze(-1).fill_(-100)
    labels = torch.cat([labels_mask, labels], dim=-1)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), image_tokens=np_(image_tokens), class_ids=np_(class_ids),
                        timesteps=np_(timesteps), noise=np_(noise), input_ids=np_(input_ids), labels=np_(labels),
                        mask_prob=np_(mask_prob), mask_id=np.int64(mask_id), codebook_size=np.int64(codebook_size),
                        min_rate=np.float32(min_rate))
    print(name, "masked per row", mask.sum(-1)[:8].tolist())


UVIT_TINY = dict(hidden_size=32, use_bias=False, hidden_dropout=0.0, cond_embed_dim=16, micro_cond_encode_dim=8,
                 micro_cond_embed_dim=40, encoder_hidden_size=24, vocab_size=40, codebook_size=32, in_channels=16,
                 block_out_channels=(24,), num_res_blocks=2, force_down_up_sample=False, block_num_heads=2,
                 num_hidden_layers=2, num_attention_heads=2, attention_dropout=0.0, intermediate_size=48,
                 norm_type="rmsnorm", layer_norm_eps=1e-6, ln_elementwise_affine=True)


def golden_uvit(name, cfg, batch, seq, text_len, seed):
    """SURVEY.md section 8 row a12 (config 4): muse/modeling_transformer_v2.py:MaskGiTUViT_v2 on a tiny configuration.
    The tensors the reference zero-initialises (AdaLN mappers, mlm_layer.conv1, GlobalResponseNorm gamma/beta - :209-223,
    :744-745) are perturbed first: with them at zero the logits are identically 0 and nothing would be tested."""
    from muse.modeling_transformer_v2 import MaskGiTUViT_v2
    torch.manual_seed(seed)
    model = MaskGiTUViT_v2(**cfg)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for k, p_ in model.named_parameters():
            if float(p_.abs().max()) == 0.0:
                p_.copy_(torch.randn(p_.shape, generator=g) * 0.05)
            elif k.endswith("norm.weight") or k.endswith("layer_norm.weight"):
                p_.add_(torch.randn(p_.shape, generator=g) * 0.1)     # norm gains away from exactly 1
    model.train()
    side = int(seq ** 0.5)
    assert side * side == seq
    V = cfg["codebook_size"]
    input_ids = torch.randint(0, V, (batch, seq), generator=g)
    masked = torch.rand(batch, seq, generator=g) < 0.5
    labels = torch.where(masked, input_ids, torch.full_like(input_ids, -100))
    input_ids = torch.where(masked, torch.full_like(input_ids, cfg["vocab_size"] - 1), input_ids)
    enc = torch.randn(batch, text_len, cfg["encoder_hidden_size"], generator=g)
    cond = torch.randn(batch, cfg["cond_embed_dim"], generator=g)
    micro = torch.tensor([[256.0, 256.0, 0.0, 0.0, 6.0], [512.0, 384.0, 16.0, 8.0, 5.5]])[:batch]
    loss_weight = torch.rand(batch, seq, generator=g) + 0.5
    out = dict(input_ids=np_(input_ids), labels=np_(labels), encoder_hidden_states=np_(enc), cond_embeds=np_(cond),
               micro_conds=np_(micro), loss_weight=np_(loss_weight), label_smoothing=np.float32(0.1))
    for k, v in model.state_dict().items():
        out["param." + k] = np_(v)
    # (1) plain mean cross-entropy, all gradients
    logits, loss = model(input_ids, enc, cond, micro, labels=labels)
    loss.backward()
    out["logits"], out["loss"] = np_(logits), np_(loss)
    for k, p_ in model.named_parameters():
        out["grad." + k] = np_(p_.grad)
    # (2) label smoothing + per-token loss weights (training/train_muse.py:742-750)
    model.zero_grad()
    _, loss_w = model(input_ids, enc, cond, micro, labels=labels, label_smoothing=0.1, loss_weight=loss_weight)
    out["loss_weighted"] = np_(loss_w)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    import json
    with open(os.path.join(HERE, "config_" + name + ".json"), "w") as f:
        json.dump({k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()}, f, indent=1, sort_keys=True)
    print(name, "loss", float(loss), "weighted", float(loss_w), "logits", tuple(logits.shape), "max|logit|", float(logits.abs().max()))


def golden_ema(name, seed, steps=14):
    """muse/modeling_ema.py:EMAModel (the real class) over `steps` calls of step() on changing parameters, two schedules: the decay each
    call used and every shadow tensor after every call"""
    from muse.modeling_ema import EMAModel
    out = dict(seed=np.int64(seed), steps=np.int64(steps))
    for si, kw in enumerate(W.EMA_SCHEDULES):
        params = [torch.nn.Parameter(t) for t in W.ema_params(seed, 0)]
        params[4].requires_grad_(False)
        ema = EMAModel(params, **kw)
        for step in range(1, steps + 1):
            with torch.no_grad():
                for p_, t in zip(params, W.ema_params(seed, step)):
                    p_.copy_(t)
            ema.step(params)
            out[f"s{si}.decay{step}"] = np.float64(-1.0 if (step - 1) % kw["update_every"] else ema.cur_decay_value)
            for i, sh in enumerate(ema.shadow_params):
                out[f"s{si}.shadow{step}.{i}"] = np_(sh).copy()        # (sub_ updates the shadow in place: a view would alias every later step)
        assert ema.optimization_step == steps
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "decays", [round(float(out[f"s0.decay{t}"]), 4) for t in range(1, steps + 1)])


def reference_configs(path="/root/reference/configs"):
    """the `model.transformer` section (+ the few dataset / training keys the loop reads) of every configs/*.yaml whose section the
    reference's own classes can construct - 16 of 25: a multi-entry block_out_channels trips MaskGiTUViT_v2's assert (:166), and
    use_conv_in_out without embedding_size trips nn.Embedding(vocab, None) - written to tests/golden/reference_configs.json for
    tests/test_gpu_reference_configs.py (the yaml files do not travel to the GPU box)"""
    import glob
    import json
    import yaml
    out = {}
    for f in sorted(glob.glob(os.path.join(path, "*.yaml"))):
        try:
            c = yaml.safe_load(open(f))
            m = c.get("model", {})
        except Exception:      # noqa: BLE001  (three files are data-shard lists, not OmegaConf trees)
            continue
        t = m.get("transformer")
        if t is None:
            continue
        bo = t.get("block_out_channels")
        if (isinstance(bo, (list, tuple)) and len(bo) != 1) or t.get("use_conv_in_out"):
            continue
        cls = ref_muse.MaskGiTUViT if m.get("architecture", "transformer") == "uvit" else ref_muse.MaskGitTransformer
        def template(kw):
            """the state-dict template of the reference's model: tensor count, parameter count, digest of the (name, shape) list"""
            import hashlib
            with torch.device("meta"):
                mod = cls(**kw)
            items = [(k, tuple(v.shape)) for k, v in mod.state_dict().items()]
            return dict(tensors=len(items), parameters=int(sum(p.numel() for p in mod.parameters())),
                        sha1=hashlib.sha1(repr(items).encode()).hexdigest())
        try:
            tmpl = template(t)                             # the reference really constructs it ...
            err = None
        except Exception as e:                             # noqa: BLE001  ... or says why not (block_num_heads 12 on 1024 channels: SURVEY.md D3)
            err = f"{type(e).__name__}: {e}"
            tmpl = template(dict(t, block_num_heads=16))   # (the override BASELINE.json's config 4 uses)
        pre = c.get("dataset", {}).get("preprocessing", {})
        out[os.path.basename(f)] = dict(
            reference_error=err, state_dict_template=tmpl,
            architecture=m.get("architecture", "transformer"), transformer=t, resolution=pre.get("resolution"),
            max_seq_length=pre.get("max_seq_length"), text_encoder=m.get("text_encoder", {}).get("type"), vq=m.get("vq_model", {}).get("type"),
            training={k: c.get("training", {}).get(k) for k in ("gradient_accumulation_steps", "batch_size", "mixed_precision", "use_ema",
                                                                "min_masking_rate", "label_smoothing", "cond_dropout_prob",
                                                                "predict_all_tokens", "noise_type")})
    with open(os.path.join(HERE, "reference_configs.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("reference_configs", len(out), "configurations")


def golden_checkpoints():
    """checkpoint directories written by the REFERENCE's save_pretrained (config.json + pytorch_model.bin, modeling_utils.py:228-285) for
    one tiny model of every class, with known parameters: tests/test_surface.py loads them with this package's from_pretrained (and, in the
    build container, hands this package's checkpoints to the reference's from_pretrained)"""
    out = os.path.join(HERE, "ckpt")
    gp = np.load(os.path.join(HERE, "uvit_tiny.npz"))
    import json
    ucfg = json.load(open(os.path.join(HERE, "config_uvit_tiny.json")))
    from muse.modeling_transformer_v2 import MaskGiTUViT_v2
    jobs = [("transformer_tiny", ref_muse.MaskGitTransformer(**W.TRANSFORMER_TINY), W.fill_state_dict(W.transformer_shapes(W.TRANSFORMER_TINY), 100, "transformer")),
            ("transformer_text_tiny", ref_muse.MaskGitTransformer(**W.TRANSFORMER_TEXT_TINY),
             W.fill_state_dict(W.transformer_shapes(W.TRANSFORMER_TEXT_TINY), 800, "transformer")),
            ("uvit_tiny", MaskGiTUViT_v2(**ucfg), {k[len("param."):]: torch.from_numpy(gp[k]) for k in gp.files if k.startswith("param.")}),
            ("vqgan_tiny", ref_muse.MaskGitVQGAN(**W.VQGAN_CKPT), W.fill_state_dict(W.vqgan_shapes(W.VQGAN_CKPT), 300, "vqgan")),
            ("taming_tiny", ref_muse.VQGANModel(**W.TAMING_CKPT), W.fill_state_dict(W.taming_shapes(W.TAMING_CKPT), 310, "vqgan"))]
    for name, model, sd in jobs:
        model.load_state_dict(sd, strict=True)
        model.save_pretrained(os.path.join(out, name))
        print("checkpoint", name, sorted(os.listdir(os.path.join(out, name))), sum(v.numel() for v in sd.values()), "parameters")


def golden_sampling_helpers(name, seed):
    """muse/sampling.py of the reference on seeded inputs: every schedule get_mask_chedule knows, log, top_k, gumbel_sample and
    mask_by_random_topk with seeded CPU generators"""
    from muse import sampling as S
    g = torch.Generator().manual_seed(seed)
    t = torch.cat([torch.tensor([0.0, 1.0, 0.5]), torch.rand(13, generator=g)])
    out = dict(t=np_(t), seed=np.int64(seed))
    for method in ("cosine", "linear", "pow0.5", "pow2", "pow3.5", "sigmoid"):
        out["schedule." + method] = np_(S.get_mask_chedule(method)(t))
    out["schedule.sigmoid_kw"] = np_(S.get_mask_chedule("sigmoid", start=-2, end=4, tau=0.7)(t))
    x = torch.rand(4, 9, generator=g) * torch.tensor([1.0, 1e-3, 1e-12, 1e-25]).view(4, 1)
    out["log.in"], out["log.out"] = np_(x), np_(S.log(x))
    logits = torch.randn(2, 5, 16, generator=g)
    out["top_k.in"] = np_(logits)
    for thres in (0.9, 0.5, 0.97):
        out[f"top_k.{thres}"] = np_(S.top_k(logits, thres))
    out["gumbel_sample.t1"] = np_(S.gumbel_sample(logits, temperature=1.0, generator=torch.Generator().manual_seed(seed + 1)))
    out["gumbel_sample.t0"] = np_(S.gumbel_sample(logits, temperature=0.0, generator=torch.Generator().manual_seed(seed + 2)))
    probs = torch.softmax(torch.randn(3, 16, generator=g), dim=-1)
    mask_len = torch.tensor([[1], [7], [15]])
    out["mask.probs"], out["mask.len"] = np_(probs), np_(mask_len)
    out["mask.out"] = np_(S.mask_by_random_topk(mask_len, probs, temperature=2.0, generator=torch.Generator().manual_seed(seed + 3)))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "masked per row", out["mask.out"].sum(-1).tolist())


LR_CASES = [("constant", {}), ("constant_with_warmup", dict(num_warmup_steps=5)), ("linear", dict(num_warmup_steps=4, num_training_steps=30)),
            ("cosine", dict(num_warmup_steps=6, num_training_steps=30)), ("cosine_with_restarts", dict(num_warmup_steps=3, num_training_steps=30, num_cycles=3)),
            ("polynomial", dict(num_warmup_steps=5, num_training_steps=30, power=2.0)), ("polynomial", dict(num_warmup_steps=0, num_training_steps=20))]


def golden_lr_schedules(name, steps=36):
    """muse/lr_schedulers.py::get_scheduler of the reference for every schedule name: the learning rates of two parameter groups (base
    1e-3 and 2.5e-4) over `steps` scheduler steps - past the end of training for the ones that have one"""
    from muse.lr_schedulers import get_scheduler
    out = dict(steps=np.int64(steps))
    for ci, (kind, kw) in enumerate(LR_CASES):
        w = [torch.nn.Parameter(torch.zeros(2)), torch.nn.Parameter(torch.zeros(3))]
        opt = torch.optim.AdamW([{"params": [w[0]]}, {"params": [w[1]], "lr": 2.5e-4}], lr=1e-3)
        sched = get_scheduler(kind, opt, **kw)
        lrs = []
        for _ in range(steps):
            lrs.append([g["lr"] for g in opt.param_groups])
            opt.step()
            sched.step()
        out[f"case{ci}"] = np.asarray(lrs, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, {c[0]: float(out[f"case{i}"][-1, 0]) for i, c in enumerate(LR_CASES)})


def golden_training_utils(name, seed):
    """muse/training_utils.py's logging diagnostics (the reference module) on a seeded batch of 12 images x 20 tokens x 16 classes whose
    masked shares cover all ten buckets but one"""
    from muse import training_utils as TU
    g = torch.Generator().manual_seed(seed)
    B, S, V, mask_id = 12, 20, 16, 15
    tokens = torch.randint(0, V - 1, (B, S), generator=g)
    counts = [1, 2, 3, 5, 7, 9, 11, 13, 16, 18, 20, 4]            # 0.05 ... 1.0 of 20 tokens: buckets 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 1
    input_ids = tokens.clone()
    for b, n in enumerate(counts):
        input_ids[b, torch.randperm(S, generator=g)[:n]] = mask_id
    labels = torch.where(input_ids == mask_id, tokens, torch.full_like(tokens, -100))
    logits = torch.randn(B, S, V, generator=g) * 2
    out = dict(input_ids=np_(input_ids), labels=np_(labels), logits=np_(logits), mask_id=np.int64(mask_id))
    out["buckets"] = np_(TU.input_ids_to_masked_buckets(input_ids, mask_id))
    out["pixel_entropy"] = np_(TU.pixel_entropy_per_percent_masked_bucket(logits.clone(), input_ids, mask_id))
    out["image_entropy"] = np_(TU.image_entropy_per_percent_masked_bucket(logits.clone(), input_ids, mask_id))
    out["cross_entropy"] = np_(TU.cross_entropy_per_percent_masked_bucket(logits.clone(), labels, input_ids, mask_id, V, 0.1))
    df = TU.token_probability_distributions_per_percent_masked_bucket(logits.clone(), input_ids, mask_id)
    out["dist.bucket"], out["dist.prob"] = df["bucket"].to_numpy().astype(np.int64), df["masked_pixel_prob"].to_numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "buckets", out["buckets"].tolist(), "rows", len(df))


def replay_decode_noise(seed, steps, rows, seq, vocab):
    """the draws a reference generate2 call makes from torch.Generator().manual_seed(seed), per step: torch.multinomial(probs
    [rows*seq, vocab], 1) fills an Exp(1) tensor of the probabilities' shape (ATen multinomial_out, one-sample fast path), then
    gumbel_noise (muse/sampling.py:13-15) fills a uniform [rows, seq] tensor"""
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(steps):
        q = torch.empty(rows * seq, vocab).exponential_(1, generator=g)
        u = torch.zeros(rows, seq).uniform_(0, 1, generator=g)
        out.append((q, u))
    return out


def golden_generate2(name, cfg, batch, seed, timesteps, temperature):
    """MaskGitTransformer.generate2 of the real reference (muse/modeling_transformer.py:1363-1456) with a seeded CPU generator;
    the file records the generator's draws (replayed, and checked against the reference's own output through the oracle)."""
    from oracle import maskgit_oracle as O
    model = ref_muse.MaskGitTransformer(**cfg)
    sd = W.fill_state_dict(W.transformer_shapes(cfg), seed, "transformer")
    model.load_state_dict(sd, strict=True)
    model.eval()
    rng = np.random.default_rng(seed + 1)
    class_ids = torch.from_numpy(rng.integers(0, cfg["num_classes"], size=(batch,)).astype(np.int64))
    with torch.no_grad():
        ids = model.generate2(class_ids=class_ids.clone(), timesteps=timesteps, temperature=temperature,
                              generator=torch.Generator().manual_seed(seed + 2))
    S, V = cfg["num_vq_tokens"], cfg["codebook_size"]
    noise = replay_decode_noise(seed + 2, timesteps, batch, S, V)
    with torch.no_grad():
        ids_o, fed = O.generate2(sd, cfg, class_ids, timesteps, temperature, noise)
    assert torch.equal(ids, ids_o), "replayed draws do not reproduce the reference's sample"
    out = dict(class_ids=np_(class_ids), ids=np_(ids), timesteps=np.int64(timesteps), temperature=np.float32(temperature),
               seed=np.int64(seed), batch=np.int64(batch))
    for i, (q, u) in enumerate(noise):
        out[f"q{i}"], out[f"u{i}"], out[f"fed{i}"] = np_(q), np_(u), np_(fed[i])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "ids", ids[0, :8].tolist(), "masked fed to last step", int((fed[-1] == cfg["vocab_size"] - 1).sum()))


def golden_uvit_generate2(name, cfg, batch, seq, text_len, seed, timesteps, temperature, guidance_scale):
    """MaskGiTUViT_v2.generate2 (muse/modeling_transformer_v2.py:330-479) of the real reference with classifier-free guidance,
    on the parameters of uvit_tiny.npz; records the generator's draws and the per-step raw samples (`intermediate`)."""
    from muse.modeling_transformer_v2 import MaskGiTUViT_v2
    from oracle import uvit_oracle as UO
    gold = np.load(os.path.join(HERE, "uvit_tiny.npz"))
    model = MaskGiTUViT_v2(**cfg)
    sd = {k[len("param."):]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("param.")}
    model.load_state_dict(sd, strict=True)
    model.eval()
    g = torch.Generator().manual_seed(seed)
    enc = torch.randn(batch, text_len, cfg["encoder_hidden_size"], generator=g)
    cond = torch.randn(batch, cfg["cond_embed_dim"], generator=g)
    empty, empty_c = torch.randn(1, text_len, cfg["encoder_hidden_size"], generator=g), torch.randn(1, cfg["cond_embed_dim"], generator=g)
    micro = torch.tensor([[256.0, 256.0, 0.0, 0.0, 6.0]])
    with torch.no_grad():
        ids, inter = model.generate2(enc, cond, micro, empty, empty_c, timesteps=timesteps, temperature=temperature,
                                     guidance_scale=guidance_scale, generator=torch.Generator().manual_seed(seed + 1),
                                     return_intermediate=True, seq_len=seq)
    noise = replay_decode_noise(seed + 1, timesteps, batch, seq, cfg["codebook_size"])
    ocfg = {k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()}
    with torch.no_grad():
        ids_o, inter_o = UO.generate2(sd, ocfg, enc, cond, micro, empty, empty_c, timesteps, temperature, guidance_scale, noise, seq)
    assert torch.equal(ids, ids_o) and all(torch.equal(a, b) for a, b in zip(inter, inter_o)), "replay != reference"
    out = dict(encoder_hidden_states=np_(enc), cond_embeds=np_(cond), empty_embeds=np_(empty), empty_cond_embeds=np_(empty_c),
               micro_conds=np_(micro), ids=np_(ids), timesteps=np.int64(timesteps), guidance_scale=np.float32(guidance_scale),
               temperature=np.array(temperature, dtype=np.float32), seq=np.int64(seq))
    for i, (q, u) in enumerate(noise):
        out[f"q{i}"], out[f"u{i}"], out[f"raw{i}"] = np_(q), np_(u), np_(inter[i])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "ids", ids[0, :8].tolist())


def _reference_function(path, *names):
    """compile selected top-level functions of a reference source file WITHOUT importing the module (training/train_muse.py
    imports wandb / omegaconf / webdataset at module level, none of which exist in this container)"""
    import ast
    import math
    import random
    src = open(path).read()
    tree = ast.parse(src)
    ns = {"torch": torch, "math": math, "random": random}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
    return ns


class _Cfg(dict):
    """attribute + .get access like the OmegaConf node the reference function receives"""
    __getattr__ = dict.__getitem__


def golden_mask_muse(name, seed, batch=6, seq=16, mask_id=47, codebook_size=32):
    """training/train_muse.py:149-226 mask_or_random_replace_tokens of the real reference (function body compiled from the
    file), its torch.rand draws replaced by recorded tensors, for every branch: default / predict_all_tokens / random_replace /
    contiguous region / eval mask ratios."""
    import random
    ns = _reference_function("/root/reference/training/train_muse.py", "mask_or_random_replace_tokens", "get_loss_weight")
    fn = ns["mask_or_random_replace_tokens"]
    rng = np.random.default_rng(seed)
    tokens = torch.from_numpy(rng.integers(0, codebook_size, size=(batch, seq)).astype(np.int64))
    out = dict(tokens=np_(tokens), mask_id=np.int64(mask_id), codebook_size=np.int64(codebook_size))
    cases = {
        "default": dict(training=_Cfg(min_masking_rate=0.1)),
        "predict_all": dict(training=_Cfg(min_masking_rate=0.0, predict_all_tokens=True)),
        "random_replace": dict(training=_Cfg(min_masking_rate=0.25, noise_type="random_replace")),
        "region": dict(training=_Cfg(min_masking_rate=0.0, mask_contiguous_region_prob=1.0)),
        "eval_ratios": dict(training=_Cfg(min_masking_rate=0.0, eval_mask_ratios=[0.2, 0.55, 0.9])),
    }
    real_rand = torch.rand
    for i, (cname, c) in enumerate(cases.items()):
        cfg = _Cfg(model=_Cfg(codebook_size=codebook_size), **c)
        draws = [W.uniforms((batch,), seed + 10 * i + 1), W.uniforms((batch, seq), seed + 10 * i + 2)]
        queue = list(draws) if cname != "eval_ratios" else [draws[1]]   # (eval_mask_ratios draws mask_prob with random.choices)
        torch.rand = lambda *a, **k: queue.pop(0)          # the function's torch.rand calls, in order (:157, :175)
        random.seed(seed + i)
        try:
            ids, labels, lw, mp = fn(tokens, mask_id, cfg, cosine_schedule, is_train=(cname != "eval_ratios"))
        finally:
            torch.rand = real_rand
        out[cname + ".input_ids"], out[cname + ".labels"], out[cname + ".mask_prob"] = np_(ids), np_(labels), np_(mp.float())
        out[cname + ".timesteps"], out[cname + ".noise"] = np_(draws[0]), np_(draws[1])
        out[cname + ".min_rate"] = np.float32(c["training"]["min_masking_rate"])
        if lw is not None:
            out[cname + ".loss_weight"] = np_(lw)
        if cname == "region":   # the rectangle each image received: the bounding box of its mask
            m = (ids == mask_id).view(batch, int(seq ** 0.5), -1)
            rects = []
            for b in range(batch):
                ys, xs = torch.nonzero(m[b].any(1)).flatten(), torch.nonzero(m[b].any(0)).flatten()
                rects.append([int(ys[0]), int(xs[0]), int(ys[-1] - ys[0] + 1), int(xs[-1] - xs[0] + 1)])
                assert int(m[b].sum()) == rects[-1][2] * rects[-1][3]
            out["region.rects"] = np.array(rects, dtype=np.int32)
        print(name, cname, "masked per image", (ids == mask_id).sum(-1).tolist())
    # conditioning dropout, training/train_muse.py:715-731: the statements themselves, executed on recorded tensors
    lines = open("/root/reference/training/train_muse.py").read().split("\n")[715:731]     # `assert` .. `cond_embeds = ...`
    import textwrap
    g = torch.Generator().manual_seed(seed + 99)
    B, L, D = 5, 3, 8
    enc = torch.randn(B, L, D, generator=g)
    enc[1, 0, :4] = 0.0                                    # exact zeros: they take the empty embedding's value even when kept
    clip = torch.randn(B, D, generator=g)
    empty, empty_clip = torch.randn(1, L, D, generator=g), torch.randn(1, D, generator=g)
    u = W.uniforms((B,), seed + 98)

    class _Z:   # torch.zeros(...).float().uniform_(0, 1) -> the recorded draws, shaped like the zeros tensor
        def __init__(self, shape): self.shape = shape
        def float(self): return self
        def uniform_(self, a, b): return u.reshape(self.shape)
    env = dict(torch=torch, config=_Cfg(training=_Cfg(cond_dropout_prob=0.6)), encoder_hidden_states=enc, clip_embeds=clip,
               empty_embeds=empty, empty_clip_embeds=empty_clip)
    real_zeros = torch.zeros
    torch.zeros = lambda shape, device=None: _Z(shape)
    try:
        exec(textwrap.dedent("\n".join(lines)), env)
    finally:
        torch.zeros = real_zeros
    out.update({"cd.enc": np_(enc), "cd.clip": np_(clip), "cd.empty": np_(empty), "cd.empty_clip": np_(empty_clip), "cd.u": np_(u),
                "cd.prob": np.float32(0.6), "cd.enc_out": np_(env["encoder_hidden_states"]), "cd.clip_out": np_(env["cond_embeds"])})
    print(name, "cond dropout kept", (u < 0.6).tolist())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


if __name__ == "__main__":
    if "--bias" in sys.argv:
        # use_bias=True (a bias on every nn.Linear and LayerNorm; no shipped config sets it, the constructor accepts it)
        golden_transformer_text("transformer_text_bias_tiny", W.TRANSFORMER_TEXT_BIAS_TINY, batch=2, text_len=5, seed=869)
        golden_transformer_text("transformer_rms_bias_tiny", W.TRANSFORMER_RMS_BIAS_TINY, batch=3, text_len=6, seed=870)
        sys.exit(0)
    if "--skip-full" not in sys.argv and "--bs64" not in sys.argv:   # the benched geometries (about two minutes on one thread)
        golden_vqgan_full("vqgan_f16_full", W.VQGAN_F16, seed=600)
        golden_transformer_full("transformer_b_full", W.TRANSFORMER_B, batch=2, seed=510)
        golden_transformer_full("transformer_b_full_bf16", W.TRANSFORMER_B, batch=2, seed=510, autocast=True)
    if "--bs64" in sys.argv or ("--skip-full" not in sys.argv and "--skip-bs64" not in sys.argv):   # (~15 minutes on one thread)
        golden_transformer_full_chunked("transformer_b_full_bs64", W.TRANSFORMER_B, batch=64, chunk=2, seed=510)
        if "--bs64" in sys.argv:
            sys.exit(0)
    if "--skip-full" not in sys.argv and "--skip-uvit-full" not in sys.argv:   # (729 M parameters: ~12 GB of host memory)
        torch.set_num_threads(8)
        golden_uvit_full("uvit_full", batch=2, seq=256, text_len=77, seed=700)
        golden_uvit_full("uvit_full_bf16", batch=2, seq=256, text_len=77, seed=700, autocast=True)
        torch.set_num_threads(1)
    if "--only-full" in sys.argv:
        sys.exit(0)
    golden_transformer("transformer_tiny", W.TRANSFORMER_TINY, batch=3, seed=100, label_smoothing=0.0)
    golden_transformer("transformer_tiny_ls", W.TRANSFORMER_TINY, batch=2, seed=110, label_smoothing=0.1)
    golden_transformer("transformer_hd48", W.TRANSFORMER_HD48, batch=2, seed=120, label_smoothing=0.0)
    golden_vqgan("vqgan_tiny", W.VQGAN_TINY, batch=2, seed=200)
    golden_mask("mask_b64", batch=64, seq=256, seed=300, mask_id=2047, codebook_size=1024, min_rate=0.0)
    golden_mask("mask_small", batch=5, seq=16, seed=310, mask_id=47, codebook_size=32, min_rate=0.3)
    golden_uvit("uvit_tiny", UVIT_TINY, batch=2, seq=16, text_len=7, seed=400)
    golden_uvit("uvit_tiny_noaffine", dict(UVIT_TINY, ln_elementwise_affine=False), batch=2, seq=16, text_len=7, seed=410)   # norms without gains (:656-660)
    golden_uvit("uvit_tiny_layernorm", dict(UVIT_TINY, norm_type="layernorm"), batch=2, seq=16, text_len=7, seed=420)        # Norm = LayerNorm (:637-638)
    golden_uvit("uvit_tiny_downup", dict(UVIT_TINY, force_down_up_sample=True), batch=2, seq=64, text_len=7, seed=430)       # stride-2 conv / transposed conv around the blocks (:510-514, :558-562)
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))   # the repo root, for `oracle`
    golden_generate2("generate2_tiny", W.TRANSFORMER_TINY, batch=3, seed=500, timesteps=6, temperature=4.5)
    golden_uvit_generate2("uvit_generate2_tiny", UVIT_TINY, batch=2, seq=16, text_len=7, seed=520, timesteps=5, temperature=(2, 0),
                          guidance_scale=3.0)
    golden_mask_muse("mask_muse", seed=540)
    golden_ema("ema_tiny", seed=560)
    golden_sampling_helpers("sampling_helpers", seed=570)
    golden_lr_schedules("lr_schedules")
    golden_training_utils("training_utils", seed=580)
    reference_configs()
    golden_checkpoints()
    golden_transformer_autocast("transformer_tiny_bf16", W.TRANSFORMER_TINY, batch=3, seed=100)
    golden_transformer_autocast("transformer_hd48_bf16", W.TRANSFORMER_HD48, batch=2, seed=120)
    golden_transformer_text("transformer_text_tiny", W.TRANSFORMER_TEXT_TINY, batch=3, text_len=7, seed=800)
    golden_transformer_text("transformer_text_proj_tiny", W.TRANSFORMER_TEXT_PROJ_TINY, batch=2, text_len=5, seed=810)
    golden_transformer_text("transformer_plain_tiny", W.TRANSFORMER_PLAIN_TINY, batch=2, text_len=0, seed=820)
    golden_transformer_text("transformer_text_bias_tiny", W.TRANSFORMER_TEXT_BIAS_TINY, batch=2, text_len=5, seed=869)
    golden_transformer_text("transformer_rms_bias_tiny", W.TRANSFORMER_RMS_BIAS_TINY, batch=3, text_len=6, seed=870)
    golden_generate2_text("generate2_text_tiny", W.TRANSFORMER_TEXT_TINY, batch=2, text_len=7, seed=830, timesteps=5, temperature=3.0,
                          guidance_scale=2.5)
    if "--skip-full" not in sys.argv:
        torch.set_num_threads(8)
        golden_transformer_text_sub("transformer_cc12m_2l", W.TRANSFORMER_CC12M_2L, batch=2, text_len=77, seed=840)
        torch.set_num_threads(1)
    golden_taming("taming_tiny", W.TAMING_TINY, batch=2, seed=600)
    golden_taming("taming_tiny_pool", W.TAMING_TINY_POOL, batch=3, seed=610)
