"""CPU oracle for the open-muse MaskGit hot path.  TEST INFRASTRUCTURE ONLY.

This module is a functional, CPU-only (torch fp32 / numpy) restatement of the algorithm of the
reference's hot path.  It is the *checker* for the HIP kernels: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it.  Nothing
under ``open-muse_amd/`` imports it, and the product path raises if the HIP library is missing
instead of falling back to this file.

Parity status: **pinned**.  ``tests/golden/make_golden.py`` imports the real reference from
``/root/reference`` (possible only in the build container), runs it on seeded inputs and writes
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` replays the same inputs through this file and
compares.  The reference itself ships no golden vectors / unit tests for this path.

Every function cites the reference file:line it restates (paths relative to /root/reference).
State dicts use the reference's parameter names and shapes, so a ``pytorch_model.bin`` written by the
reference can be fed in unchanged.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# ----------------------------------------------------------------------------------------------
# MaskGitTransformer  (muse/modeling_transformer.py)
# ----------------------------------------------------------------------------------------------
def _ln(x: Tensor, w: Tensor, eps: float, b: Optional[Tensor] = None) -> Tensor:
    # LayerNorm, weight-only unless use_bias: muse/modeling_transformer.py:124-137
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def _linear(x: Tensor, sd: SD, name: str) -> Tensor:
    """nn.Linear `name` of the state dict: x W^T (+ bias when the model was built with use_bias, muse/modeling_transformer.py:170-176)"""
    y = x @ sd[name + ".weight"].t()
    b = sd.get(name + ".bias")
    return y if b is None else y + b


def _drop(t: Tensor, keep: Optional[Tensor], p: float) -> Tensor:
    """nn.Dropout in training mode with the Bernoulli keep-mask made explicit: t * keep / (1 - p)"""
    return t if keep is None else t * keep.to(t.dtype) / (1.0 - p)


def attention(x: Tensor, sd: SD, prefix: str, num_heads: int, keep: Optional[Tensor] = None, p: float = 0.0) -> Tensor:
    """Full-visibility self attention.  muse/modeling_transformer.py:190-241 (non-xformers branch).

    scores = (q k^T) * (1/sqrt(hd)) via baddbmm alpha (:226-231), softmax over keys (:236), P@V (:238),
    heads re-assembled side by side (:240), then the ``out`` projection (:218).
    """
    B, S, H = x.shape
    hd = H // num_heads
    q = _linear(x, sd, prefix + "query")
    k = _linear(x, sd, prefix + "key")
    v = _linear(x, sd, prefix + "value")
    q = q.view(B, S, num_heads, hd).transpose(1, 2)
    k = k.view(B, S, num_heads, hd).transpose(1, 2)
    v = v.view(B, S, num_heads, hd).transpose(1, 2)
    # the reference divides by float32(sqrt(float32(hd))) through baddbmm's alpha
    alpha = 1.0 / float(torch.sqrt(torch.tensor(hd, dtype=torch.float32)))
    scores = torch.matmul(q, k.transpose(-1, -2)) * alpha
    probs = _drop(torch.softmax(scores, dim=-1), keep, p)                     # self.dropout(attn_weights) :237
    ctx = torch.matmul(probs, v).transpose(1, 2).reshape(B, S, H)
    return _linear(ctx, sd, prefix + "out")


def feed_forward(x: Tensor, sd: SD, prefix: str, eps: float, keep: Optional[Tensor] = None, p: float = 0.0) -> Tensor:
    """NormFormer GLU MLP.  muse/modeling_transformer.py:785-799.

    LN(H) -> gelu_erf(x W0^T) * (x W1^T) -> LN(I) -> Wo.
    """
    h = _ln(x, sd[prefix + "pre_mlp_layer_norm.weight"], eps)
    g = F.gelu(h @ sd[prefix + "wi_0.weight"].t())
    lin = h @ sd[prefix + "wi_1.weight"].t()
    h = _drop(_ln(g * lin, sd[prefix + "mid_mlp_layer_norm.weight"], eps), keep, p)   # self.dropout(hidden_states) :797
    return h @ sd[prefix + "wo.weight"].t()


def transformer_layer(x: Tensor, sd: SD, prefix: str, num_heads: int, eps: float, drop: Optional[dict] = None) -> Tensor:
    """Pre-LN + NormFormer block.  muse/modeling_transformer.py:875-904 (no cross attention).
    drop = {"attn": keep [B, nh, S, S], "ffn": keep [B, S, I], "p_attn", "p_hidden"} for training-mode dropout."""
    d = drop or {}
    a = attention(_ln(x, sd[prefix + "attn_layer_norm.weight"], eps), sd, prefix + "attention.", num_heads, d.get("attn"),
                  d.get("p_attn", 0.0))
    x = x + _ln(a, sd[prefix + "post_attn_layer_norm.weight"], eps)
    return x + feed_forward(x, sd, prefix + "ffn.", eps, d.get("ffn"), d.get("p_hidden", 0.0))


def transformer_forward(
    sd: SD,
    cfg: dict,
    input_ids: Tensor,
    labels: Optional[Tensor] = None,
    label_smoothing: float = 0.0,
    dropout: Optional[dict] = None,
):
    """MaskGitTransformer.forward.  muse/modeling_transformer.py:1224-1281.
    dropout (training mode, hidden_dropout / attention_dropout > 0): {"p_hidden", "p_attn", "embed": keep [B, S, H],
    "layers": [{"attn": keep, "ffn": keep}, ...]} - the Bernoulli masks nn.Dropout would draw, made explicit.

    Embed (:942-957) -> L x TransformerLayer -> encoder_layer_norm (:1268) -> MlmLayer (:979-985)
    -> F.cross_entropy(ignore_index=-100) (:1276-1279).  Dropout is the identity (p=0 / eval).
    """
    eps = float(cfg.get("layer_norm_eps", 1e-5))
    nh = int(cfg["num_attention_heads"])
    L = int(cfg["num_hidden_layers"])
    S = input_ids.shape[-1]
    x = sd["embed.word_embeddings.weight"][input_ids] + sd["embed.position_embeddings.weight"][:S][None]
    if dropout is not None:
        x = _drop(x, dropout.get("embed"), dropout.get("p_hidden", 0.0))         # Embed.dropout :956
    for i in range(L):
        dl = None if dropout is None else dict(dropout["layers"][i], p_hidden=dropout.get("p_hidden", 0.0), p_attn=dropout.get("p_attn", 0.0))
        x = transformer_layer(x, sd, f"transformer_layers.{i}.", nh, eps, dl)
    x = _ln(x, sd["encoder_layer_norm.weight"], eps)
    h = F.gelu(x @ sd["mlm_layer.mlm_dense.weight"].t())
    h = _ln(h, sd["mlm_layer.mlm_ln.weight"], eps)
    logits = h @ sd["mlm_layer.to_logits.weight"].t()
    if labels is None:
        return logits
    V = logits.shape[-1]
    loss = F.cross_entropy(logits.view(-1, V), labels.view(-1), ignore_index=-100, label_smoothing=label_smoothing)
    return logits, loss


def transformer_loss_and_grads(sd: SD, cfg: dict, input_ids: Tensor, labels: Tensor, label_smoothing: float = 0.0,
                               dropout: Optional[dict] = None):
    """loss.backward() of the forward above (train_maskgit_imagenet.py:426-433); returns (logits, loss, grads)."""
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    logits, loss = transformer_forward(leaf, cfg, input_ids, labels, label_smoothing, dropout)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaf.items()}
    return logits.detach(), loss.detach(), grads


def adamw_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, beta1: float, beta2: float,
               eps: float, weight_decay: float) -> None:
    """torch.optim.AdamW single-tensor update (decoupled weight decay), as selected at
    training/train_maskgit_imagenet.py:242-261 and stepped at :438.  In place on p, m, v."""
    p.mul_(1.0 - lr * weight_decay)
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


# ----------------------------------------------------------------------------------------------
# mask sampling of the train step  (training/train_maskgit_imagenet.py:357-394)
# ----------------------------------------------------------------------------------------------
def cosine_schedule(t: Tensor) -> Tensor:
    # muse/sampling.py:38-39
    return torch.cos(t * math.pi * 0.5)


def prepare_inputs_and_labels(
    image_tokens: Tensor,
    class_ids: Tensor,
    timesteps: Tensor,
    noise: Tensor,
    mask_id: int,
    codebook_size: int,
    min_masking_rate: float = 0.0,
):
    """training/train_maskgit_imagenet.py:371-394, with the two torch.rand draws (:375, :381) supplied by
    the caller so that CPU and GPU see identical uniforms.

    Note the reference thresholds the *argsort array itself* (:381-382): position j is masked iff the index of
    the j-th smallest noise value is < num_token_masked.
    """
    B, S = image_tokens.shape
    mask_prob = cosine_schedule(timesteps).clip(min_masking_rate)
    num_token_masked = (S * mask_prob).round().clamp(min=1)
    batch_randperm = noise.argsort(dim=-1)
    mask = batch_randperm < num_token_masked.unsqueeze(-1)
    input_ids = torch.where(mask, mask_id, image_tokens)
    labels = torch.where(mask, image_tokens, -100)
    cls = (class_ids + codebook_size).unsqueeze(-1)
    input_ids = torch.cat([cls, input_ids], dim=-1)
    labels = torch.cat([torch.full_like(cls, -100), labels], dim=-1)
    return input_ids, labels, mask_prob


# ----------------------------------------------------------------------------------------------
# MaskGitVQGAN  (muse/modeling_maskgit_vqgan.py)
# ----------------------------------------------------------------------------------------------
def _conv_same(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    # Conv2dSame, stride 1: muse/modeling_maskgit_vqgan.py:33-45 (pad (k-1)//2 left/top, rest right/bottom)
    k = w.shape[-1]
    p = k - 1
    if p > 0:
        x = F.pad(x, [p // 2, p - p // 2, p // 2, p - p // 2])
    return F.conv2d(x, w, b)


def _gn_silu(x: Tensor, sd: SD, prefix: str) -> Tensor:
    # GroupNorm(32, eps=1e-6, affine) followed by SiLU: muse/modeling_maskgit_vqgan.py:61,73-74
    return F.silu(F.group_norm(x, 32, sd[prefix + "weight"], sd[prefix + "bias"], 1e-6))


def resnet_block(x: Tensor, sd: SD, prefix: str) -> Tensor:
    """muse/modeling_maskgit_vqgan.py:71-85.  Quirk kept: when channels change the 'shortcut' is a 1x1 conv of the
    conv2 *output* (:82-83), i.e. out = h + nin(h); the block input is dropped."""
    h = _conv_same(_gn_silu(x, sd, prefix + "norm1."), sd[prefix + "conv1.weight"], None)
    h = _conv_same(_gn_silu(h, sd, prefix + "norm2."), sd[prefix + "conv2.weight"], None)
    if (prefix + "nin_shortcut.weight") in sd:
        res = _conv_same(h, sd[prefix + "nin_shortcut.weight"], None)
    else:
        res = x
    return h + res


def vqgan_encoder(sd: SD, cfg: dict, pixel_values: Tensor) -> Tensor:
    """Encoder.forward, muse/modeling_maskgit_vqgan.py:175-189 (+ DownsamplingBlock :107-114)."""
    nres = len(cfg["channel_mult"])
    nb = int(cfg["num_res_blocks"])
    h = _conv_same(pixel_values, sd["encoder.conv_in.weight"], None)
    for lvl in range(nres):
        for b in range(nb):
            h = resnet_block(h, sd, f"encoder.down.{lvl}.block.{b}.")
        if lvl != nres - 1:
            h = F.avg_pool2d(h, kernel_size=2, stride=2)
    for b in range(nb):
        h = resnet_block(h, sd, f"encoder.mid.{b}.")
    h = _gn_silu(h, sd, "encoder.norm_out.")
    return _conv_same(h, sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"])


def vq_distances(z_flat: Tensor, codebook: Tensor) -> Tensor:
    """VectorQuantizer.compute_distances, muse/modeling_maskgit_vqgan.py:303-316:
    addmm(|z|^2 + |e|^2, z, e^T, alpha=-2)."""
    zn = z_flat.pow(2.0).sum(dim=1, keepdim=True)
    en = codebook.t().pow(2.0).sum(dim=0, keepdim=True)
    return torch.addmm(zn + en, z_flat, codebook.t(), alpha=-2.0)


def vq_indices(z: Tensor, codebook: Tensor) -> Tensor:
    """VectorQuantizer.get_code, muse/modeling_maskgit_vqgan.py:342-348: NCHW -> NHWC -> argmin of distances."""
    B = z.shape[0]
    zf = z.permute(0, 2, 3, 1).contiguous().reshape(-1, codebook.shape[1])
    return torch.argmin(vq_distances(zf, codebook), dim=1).reshape(B, -1)


def vqgan_encode(sd: SD, cfg: dict, pixel_values: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """MaskGitVQGAN.encode, muse/modeling_maskgit_vqgan.py:380-386 -> (z, z_q, indices)."""
    z = vqgan_encoder(sd, cfg, pixel_values)
    cb = sd["quantize.embedding.weight"]
    idx = vq_indices(z, cb)
    B, C, Hh, Ww = z.shape
    z_q = cb[idx].view(B, Hh, Ww, C).permute(0, 3, 1, 2).contiguous()  # == one-hot @ codebook (:280-284,299)
    return z, z_q, idx


def vqgan_decoder(sd: SD, cfg: dict, z_q: Tensor) -> Tensor:
    """Decoder.forward, muse/modeling_maskgit_vqgan.py:223-240 (+ UpsamplingBlock :141-149)."""
    nres = len(cfg["channel_mult"])
    nb = int(cfg["num_res_blocks"])
    h = _conv_same(z_q, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"])
    for b in range(nb):
        h = resnet_block(h, sd, f"decoder.mid.{b}.")
    for lvl in reversed(range(nres)):
        for b in range(nb):
            h = resnet_block(h, sd, f"decoder.up.{lvl}.block.{b}.")
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv_same(h, sd[f"decoder.up.{lvl}.upsample_conv.weight"], sd[f"decoder.up.{lvl}.upsample_conv.bias"])
    h = _gn_silu(h, sd, "decoder.norm_out.")
    return _conv_same(h, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"])


def vqgan_decode_code(sd: SD, cfg: dict, indices: Tensor) -> Tensor:
    """MaskGitVQGAN.decode_code, muse/modeling_maskgit_vqgan.py:392-395 (+ get_codebook_entry :318-324)."""
    B, T = indices.shape
    side = int(math.sqrt(T))
    z_q = sd["quantize.embedding.weight"][indices].reshape(B, side, side, -1).permute(0, 3, 1, 2)
    return vqgan_decoder(sd, cfg, z_q)


# ----------------------------------------------------------------------------------------------
# One full train step, exactly as training/train_maskgit_imagenet.py:357-452 strings the pieces together
# ----------------------------------------------------------------------------------------------
def train_step(vq_sd: SD, vq_cfg: dict, tr_sd: SD, tr_cfg: dict, pixel_values: Tensor, class_ids: Tensor,
               timesteps: Tensor, noise: Tensor, label_smoothing: float = 0.0):
    with torch.no_grad():
        _, _, tokens = vqgan_encode(vq_sd, vq_cfg, pixel_values)
        input_ids, labels, mask_prob = prepare_inputs_and_labels(
            tokens, class_ids, timesteps, noise, mask_id=int(tr_cfg["vocab_size"]) - 1,
            codebook_size=int(vq_cfg["num_embeddings"]))
    logits, loss, grads = transformer_loss_and_grads(tr_sd, tr_cfg, input_ids, labels, label_smoothing)
    return dict(tokens=tokens, input_ids=input_ids, labels=labels, mask_prob=mask_prob, logits=logits, loss=loss,
                grads=grads)


# ----------------------------------------------------------------------------------------------
# MaskGit parallel decoding with explicit random draws  (muse/modeling_transformer.py:1363-1456, muse/sampling.py)
# ----------------------------------------------------------------------------------------------
def _clamp_log(t: Tensor, eps: float = 1e-20) -> Tensor:
    # muse/sampling.py:9-10
    return torch.log(t.clamp(min=eps))


def sample_step(logits: Tensor, input_ids: Tensor, mask_id: int, temperature, sched_mask_len: int, q_exp: Tensor, u: Tensor):
    """One decoding iteration on codebook-sliced logits [B, S, V] with the random draws made explicit:
    q_exp [B*S, V] = the Exp(1) draws torch.multinomial(num_samples=1) makes internally (it returns argmax(probs / q):
    ATen multinomial_out, "fast path for one sample"), u [B, S] = the uniform draws of gumbel_noise (muse/sampling.py:13-15).
    muse/modeling_transformer.py:1425-1454; mask_by_random_topk muse/sampling.py:30-35.
    -> (raw samples, sampled ids with known tokens restored, next input ids)"""
    B, S, V = logits.shape
    probs = logits.softmax(dim=-1)
    raw = torch.argmax(probs.reshape(-1, V) / q_exp.reshape(-1, V), dim=-1).view(B, S)
    unknown_map = input_ids == mask_id
    sampled = torch.where(unknown_map, raw, input_ids)
    selected = torch.gather(probs, -1, sampled.long()[..., None]).squeeze(-1)
    selected = torch.where(unknown_map, selected, torch.finfo(selected.dtype).max)
    mask_len = torch.tensor([[float(sched_mask_len)]])
    mask_len = torch.max(torch.tensor([1]), torch.min(unknown_map.sum(dim=-1, keepdim=True) - 1, mask_len))
    gumbel = -_clamp_log(-_clamp_log(u))
    confidence = _clamp_log(selected) + temperature * gumbel
    cut_off = torch.gather(torch.sort(confidence, dim=-1).values, 1, mask_len.long())
    masking = confidence < cut_off
    return raw, sampled, torch.where(masking, mask_id, sampled)


def generate2(sd: SD, cfg: dict, class_ids: Tensor, timesteps: int, temperature: float, noise, input_ids: Optional[Tensor] = None):
    """MaskGitTransformer.generate2 (class-conditional path, guidance off), muse/modeling_transformer.py:1363-1456.
    noise[step] = (q_exp [B*S, codebook_size], u [B, S]).  -> (final sampled ids, list of the input_ids fed to each step)"""
    mask_id, S, V = cfg["vocab_size"] - 1, cfg["num_vq_tokens"], cfg["codebook_size"]
    B = len(class_ids)
    cls = class_ids + V                                                      # :1388-1389
    if input_ids is None:
        input_ids = torch.full((B, S), mask_id, dtype=torch.long)            # :1392-1393
    fed, sampled = [], input_ids
    for step in range(timesteps):
        fed.append(input_ids)
        logits = transformer_forward(sd, cfg, torch.cat([cls[:, None], input_ids], dim=1))
        logits = logits[..., :V][:, 1:]                                      # :1416-1421
        ratio = 1.0 * (step + 1) / timesteps
        mask_ratio = cosine_schedule(torch.tensor(ratio))
        sched = int((S * mask_ratio).floor())
        temperature = temperature * (1.0 - ratio)                            # :1451 (compounds across steps)
        q, u = noise[step]
        _, sampled, input_ids = sample_step(logits, input_ids, mask_id, temperature, sched, q, u)
    return sampled, fed


# ----------------------------------------------------------------------------------------------
# MaskGitTransformer, general form: text conditioning, RMSNorm, plain pre-LN layers, optional final norm / MLM head
# ----------------------------------------------------------------------------------------------
def _rms(x: Tensor, w: Tensor, eps: float) -> Tensor:
    # muse/modeling_transformer.py:75-100 (the pure-torch RMSNorm the reference falls back to without apex): f32 variance, no mean
    variance = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
    return x * torch.rsqrt(variance + eps) * w


def _norm_by_type(x: Tensor, sd: SD, name: str, eps: float, cfg: dict) -> Tensor:
    # norm_cls of muse/modeling_transformer.py:1128, :833, :775, :973: LayerNorm (with a bias under use_bias) or RMSNorm (never one)
    w = sd[name + ".weight"]
    return _ln(x, w, eps, sd.get(name + ".bias")) if cfg.get("norm_type", "layernorm") == "layernorm" else _rms(x, w, eps)


def cross_attention(x: Tensor, ctx: Tensor, sd: SD, prefix: str, num_heads: int) -> Tensor:
    """Attention.forward with encoder_hidden_states (muse/modeling_transformer.py:190-241): queries from x [B, S, H], keys and
    values from the text states ctx [B, L, E]; no mask (the reference's encoder_attention_mask path raises, :214)."""
    B, S, H = x.shape
    L = ctx.shape[1]
    hd = H // num_heads
    q = _linear(x, sd, prefix + "query").view(B, S, num_heads, hd).transpose(1, 2)
    k = _linear(ctx, sd, prefix + "key").view(B, L, num_heads, hd).transpose(1, 2)
    v = _linear(ctx, sd, prefix + "value").view(B, L, num_heads, hd).transpose(1, 2)
    alpha = 1.0 / float(torch.sqrt(torch.tensor(hd, dtype=torch.float32)))
    probs = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) * alpha, dim=-1)
    out = torch.matmul(probs, v).transpose(1, 2).reshape(B, S, H)
    return _linear(out, sd, prefix + "out")


def transformer_forward_general(sd: SD, cfg: dict, input_ids: Tensor, encoder_hidden_states: Optional[Tensor] = None,
                                labels: Optional[Tensor] = None, label_smoothing: float = 0.0, cond_keep: Optional[Tensor] = None):
    """MaskGitTransformer.forward for every configuration the constructor accepts without conv embeddings (use_bias: every
    nn.Linear and every LayerNorm carries a bias, :130, :170-176, :770-778, :973-977, :1155; RMSNorm never does)
    (muse/modeling_transformer.py:1224-1281; layer :875-904; feed-forward :785-799; MLM head :979-985).
    cond_keep [B] bool = the mask prob_mask_like draws for condition dropout (:1243-1247), applied AFTER the projection."""
    eps = float(cfg.get("layer_norm_eps", 1e-5))
    nh, L = int(cfg["num_attention_heads"]), int(cfg["num_hidden_layers"])
    nf = bool(cfg.get("use_normformer", True))
    S = input_ids.shape[-1]
    x = sd["embed.word_embeddings.weight"][input_ids] + sd["embed.position_embeddings.weight"][:S][None]
    enc = encoder_hidden_states
    if enc is not None and cfg.get("project_encoder_hidden_states", False):
        enc = _norm_by_type(_linear(enc, sd, "encoder_proj"), sd, "encoder_proj_layer_norm", eps, cfg)                # :1239-1241
    if enc is not None and cond_keep is not None:
        enc = enc * cond_keep.view(-1, 1, 1).to(enc.dtype)                                                            # :1243-1247
    for i in range(L):
        p = f"transformer_layers.{i}."
        a = attention(_norm_by_type(x, sd, p + "attn_layer_norm", eps, cfg), sd, p + "attention.", nh)
        if nf:
            a = _norm_by_type(a, sd, p + "post_attn_layer_norm", eps, cfg)
        x = x + a
        if enc is not None and (p + "crossattention.query.weight") in sd:
            a = cross_attention(_norm_by_type(x, sd, p + "crossattn_layer_norm", eps, cfg), enc, sd, p + "crossattention.", nh)
            if nf:
                a = _norm_by_type(a, sd, p + "post_crossattn_layer_norm", eps, cfg)
            x = x + a
        h = _ln(x, sd[p + "ffn.pre_mlp_layer_norm.weight"], eps, sd.get(p + "ffn.pre_mlp_layer_norm.bias"))   # always a LayerNorm (:768-770)
        h = F.gelu(_linear(h, sd, p + "ffn.wi_0")) * _linear(h, sd, p + "ffn.wi_1")
        if nf:
            h = _norm_by_type(h, sd, p + "ffn.mid_mlp_layer_norm", eps, cfg)
        x = x + _linear(h, sd, p + "ffn.wo")
    if cfg.get("use_encoder_layernorm", True):
        x = _norm_by_type(x, sd, "encoder_layer_norm", eps, cfg)
    if cfg.get("use_mlm_layer", True):
        h = F.gelu(_linear(x, sd, "mlm_layer.mlm_dense"))
        if cfg.get("use_mlm_layernorm", True):
            h = _norm_by_type(h, sd, "mlm_layer.mlm_ln", eps, cfg)
        logits = _linear(h, sd, "mlm_layer.to_logits")
    else:
        logits = _linear(x, sd, "to_logits")
    if labels is None:
        return logits
    V = logits.shape[-1]
    return logits, F.cross_entropy(logits.view(-1, V), labels.view(-1), ignore_index=-100, label_smoothing=label_smoothing)


def transformer_general_loss_and_grads(sd: SD, cfg: dict, input_ids: Tensor, labels: Tensor, encoder_hidden_states: Optional[Tensor] = None,
                                       label_smoothing: float = 0.0, cond_keep: Optional[Tensor] = None):
    """-> (logits, loss, parameter grads, gradient of the text states or None)"""
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    enc = encoder_hidden_states.detach().clone().requires_grad_(True) if encoder_hidden_states is not None else None
    logits, loss = transformer_forward_general(leaf, cfg, input_ids, enc, labels, label_smoothing, cond_keep)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaf.items()}
    return logits.detach(), loss.detach(), grads, (enc.grad if enc is not None else None)


def generate2_text(sd: SD, cfg: dict, encoder_hidden_states: Tensor, timesteps: int, temperature: float, noise,
                   guidance_scale: float = 0.0, negative_embeds: Optional[Tensor] = None):
    """MaskGitTransformer.generate2 with text states and classifier-free guidance (muse/modeling_transformer.py:1394-1416):
    conditional and unconditional (zeros unless negative_embeds) halves of one doubled batch, logits cut to the codebook, mixed as
    uncond + scale * (cond - uncond).  -> final sampled ids"""
    mask_id, S, V = cfg["vocab_size"] - 1, cfg["num_vq_tokens"], cfg["codebook_size"]
    B = encoder_hidden_states.shape[0]
    input_ids = torch.full((B, S), mask_id, dtype=torch.long)
    guided = guidance_scale > 0
    if guided:
        un = torch.zeros_like(encoder_hidden_states) if negative_embeds is None else negative_embeds
        cond = torch.cat([encoder_hidden_states, un])
    sampled = input_ids
    for step in range(timesteps):
        if guided:
            lo = transformer_forward_general(sd, cfg, torch.cat([input_ids] * 2), cond)
            c, u_ = lo[:B, :, :V], lo[B:, :, :V]
            logits = u_ + guidance_scale * (c - u_)
        else:
            logits = transformer_forward_general(sd, cfg, input_ids, encoder_hidden_states)[..., :V]
        ratio = 1.0 * (step + 1) / timesteps
        sched = int((S * cosine_schedule(torch.tensor(ratio))).floor())
        temperature = temperature * (1.0 - ratio)
        q, u = noise[step]
        _, sampled, input_ids = sample_step(logits, input_ids, mask_id, temperature, sched, q, u)
    return sampled


# ----------------------------------------------------------------------------------------------
# training/train_muse.py: masking variants and conditioning dropout
# ----------------------------------------------------------------------------------------------
def mask_or_random_replace_tokens(image_tokens: Tensor, mask_id: int, min_masking_rate: float, *, timesteps: Optional[Tensor] = None,
                                  mask_prob: Optional[Tensor] = None, noise: Optional[Tensor] = None, rects: Optional[Tensor] = None,
                                  all_labels: bool = False):
    """training/train_muse.py:149-226 with the draws explicit.  timesteps -> cosine schedule clipped to min_masking_rate
    (:157-161), or mask_prob given directly (eval_mask_ratios, :152-154).  Random mask: argsort(noise) < k (:175-176);
    contiguous region: rects [B, 4] = (y0, x0, h, w) on the sqrt(S) grid (:178-199; the reference draws them with `random`).
    input_ids = mask_id where masked - the `noise_type` test at :202 is always true, so there is no random-replace branch.
    all_labels (predict_all_tokens or noise_type == "random_replace", :212-217): labels = tokens and
    loss_weight = 1 - (1 - mask) * ((1 - mask_prob) * (1 - 0.3)) (:145-146), else labels = -100 outside the mask, no weight."""
    B, S = image_tokens.shape
    if mask_prob is None:
        mask_prob = cosine_schedule(timesteps).clip(min_masking_rate)
    num_token_masked = (S * mask_prob).round().clamp(min=1)
    if rects is None:
        mask = noise.argsort(dim=-1) < num_token_masked.unsqueeze(-1)
    else:
        side = int(S ** 0.5)
        mask = torch.zeros((B, side, side))
        for b in range(B):
            y0, x0, h, w = (int(v) for v in rects[b])
            mask[b, y0:y0 + h, x0:x0 + w] = 1
        mask = mask.reshape(B, S).to(torch.bool)
    input_ids = torch.where(mask, mask_id, image_tokens)
    if all_labels:
        labels = image_tokens
        loss_weight = 1 - (1 - mask.long()) * ((1 - mask_prob) * (1 - 0.3))[:, None]
    else:
        labels = torch.where(mask, image_tokens, -100)
        loss_weight = None
    return input_ids, labels, loss_weight, mask_prob


def cond_dropout(x: Tensor, empty: Tensor, uniforms: Tensor, prob: float) -> Tensor:
    """training/train_muse.py:715-731: mask = u < p, shaped to broadcast over one image's embedding; where (x * mask) is
    non-zero keep x, else the empty embedding (so the conditioning survives with probability p, element-wise zeros excepted)."""
    B = x.shape[0]
    mask = (uniforms.reshape(B, *([1] * (x.dim() - 1))) < prob)
    return torch.where((x * mask).bool(), x, empty.expand(B, *empty.shape[-(x.dim() - 1):]))
