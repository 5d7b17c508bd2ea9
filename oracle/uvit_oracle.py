"""CPU oracle for SURVEY.md §8 row a12: ``MaskGiTUViT_v2`` (config 4, ``configs/cc12m_uvit_clip.yaml``).  TEST INFRASTRUCTURE ONLY.

Functional, CPU-only (torch fp32) restatement of ``muse/modeling_transformer_v2.py`` — forward, weighted / smoothed
cross-entropy, and (through autograd on this restatement) every parameter gradient.  Like ``maskgit_oracle.py`` it is a
checker: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` may import it; nothing under
``open-muse_amd/`` does.

Parity status: **pinned** — ``tests/golden/make_golden.py`` runs the real reference class on a tiny configuration (with the
zero-initialised AdaLN mappers / ``mlm_layer.conv1`` perturbed, SURVEY.md §8b: a freshly constructed model outputs logits ≡ 0
and any implementation would "pass") and commits ``tests/golden/uvit_tiny*.npz``; ``tests/test_oracle_golden.py`` replays them.

Status of the HIP path for this row: not built in round 1; this file and its goldens are the groundwork.

State dicts use the reference's parameter names and shapes.  All line numbers are muse/modeling_transformer_v2.py.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


def sinusoidal_encode(features: Tensor, dim: int, max_positions: int = 10000) -> Tensor:
    """:59-76 — [cos | sin] of features x exp(-ln(max_positions) * i / half), zero padded to an odd dim."""
    half = dim // 2
    freq = torch.exp(torch.arange(half, dtype=torch.float32) * (-math.log(max_positions) / half))
    ang = features.to(torch.float32)[:, None] * freq[None, :]
    emb = torch.cat([ang.cos(), ang.sin()], dim=1)
    if dim % 2 == 1:
        emb = F.pad(emb, (0, 1))
    return emb


def norm(x: Tensor, sd: SD, prefix: str, cfg: dict, residual: Optional[Tensor] = None):
    """``Norm`` = RMSNorm (:638-664, ``unfused_rms_norm`` :673-691) or LayerNorm (:694-737) with the residual stream:
    returns (normed, prenorm_residual) where prenorm_residual = x + residual."""
    if residual is not None:
        x = x + residual
    pre = x
    w = sd.get(prefix + "weight")
    if cfg.get("norm_type", "rmsnorm") == "rmsnorm":
        var = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
        y = x * torch.rsqrt(var + cfg["layer_norm_eps"])
        if w is not None:
            y = y * w
    else:
        y = F.layer_norm(x, (x.shape[-1],), w, sd.get(prefix + "bias"), cfg["layer_norm_eps"])
    return y, pre


def layer_norm(x: Tensor, sd: SD, prefix: str, cfg: dict, residual: Optional[Tensor] = None):
    """``LayerNorm`` class used directly by the feed-forward (:928), whatever ``norm_type`` says (:694-737)."""
    if residual is not None:
        x = x + residual
    return F.layer_norm(x, (x.shape[-1],), sd.get(prefix + "weight"), sd.get(prefix + "bias"), cfg["layer_norm_eps"]), x


def norm2d(x: Tensor, sd: SD, prefix: str, cfg: dict) -> Tensor:
    """``Norm2D`` :621-630 — channel norm of an NCHW tensor."""
    y, _ = norm(x.permute(0, 2, 3, 1), sd, prefix + "norm.", cfg)
    return y.permute(0, 3, 1, 2)


def linear(x: Tensor, sd: SD, prefix: str) -> Tensor:
    return F.linear(x, sd[prefix + "weight"], sd.get(prefix + "bias"))


def ada_ln(x: Tensor, cond: Tensor, sd: SD, prefix: str) -> Tensor:
    """``AdaLNModulation`` :1025-1037 — x * (1 + scale) + shift with (scale, shift) = mapper(silu(cond)).chunk(2)."""
    scale, shift = linear(F.silu(cond), sd, prefix + "mapper.").chunk(2, dim=1)
    if x.dim() > 3:
        scale, shift = scale[:, :, None, None], shift[:, :, None, None]
    else:
        scale, shift = scale[:, None], shift[:, None]
    return x * (1 + scale) + shift


def attention(x: Tensor, context: Tensor, sd: SD, prefix: str, num_heads: int) -> Tensor:
    """``Attention`` :834-915 — q from x, k/v from context, scores scaled by 1/float32(sqrt(hd)) through baddbmm's alpha
    (:898-903), softmax over keys, P·V, heads side by side, bias-free ``out``."""
    B, Sq, H = x.shape
    Skv = context.shape[1]
    hd = H // num_heads
    q = linear(x, sd, prefix + "query.").view(B, Sq, num_heads, hd).transpose(1, 2)
    k = linear(context, sd, prefix + "key.").view(B, Skv, num_heads, hd).transpose(1, 2)
    v = linear(context, sd, prefix + "value.").view(B, Skv, num_heads, hd).transpose(1, 2)
    alpha = 1.0 / float(torch.sqrt(torch.tensor(hd, dtype=torch.float32)))
    probs = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) * alpha, dim=-1)
    ctx = torch.matmul(probs, v).transpose(1, 2).reshape(B, Sq, H)
    return linear(ctx, sd, prefix + "out.")


def res_block(x: Tensor, cond: Tensor, sd: SD, prefix: str, cfg: dict) -> Tensor:
    """``ResBlock`` :586-618 — depthwise 3x3 -> channel norm -> Linear(C,4C) -> GELU -> GlobalResponseNorm (:741-751)
    -> Linear(4C,C) -> + x -> AdaLN."""
    C = x.shape[1]
    h = F.conv2d(x, sd[prefix + "depthwise.weight"], sd.get(prefix + "depthwise.bias"), padding=1, groups=C)
    h = norm2d(h, sd, prefix + "norm.", cfg).permute(0, 2, 3, 1)
    h = F.gelu(linear(h, sd, prefix + "channelwise.0."))
    gx = torch.norm(h, p=2, dim=(1, 2), keepdim=True)
    nx = gx / (gx.mean(dim=-1, keepdim=True) + 1e-6)
    h = sd[prefix + "channelwise.2.gamma"] * (h * nx) + sd[prefix + "channelwise.2.beta"] + h
    h = linear(h, sd, prefix + "channelwise.4.").permute(0, 3, 1, 2)
    return ada_ln(h + x, cond, sd, prefix + "adaLN_modulation.")


def attention_block_2d(x: Tensor, enc: Tensor, sd: SD, prefix: str, cfg: dict) -> Tensor:
    """``AttentionBlock2D`` :795-831 — two cross-attentions to the (projected) text states on the flattened pixels."""
    B, C, Hh, Ww = x.shape
    h = x.view(B, C, Hh * Ww).permute(0, 2, 1)
    if prefix + "kv_mapper.weight" in sd:
        enc = linear(F.silu(enc), sd, prefix + "kv_mapper.")
    nh = cfg["block_num_heads"]
    h, res = norm(h, sd, prefix + "attn_layer_norm.", cfg)
    h = attention(h, enc, sd, prefix + "attention.", nh)
    h, res = norm(h, sd, prefix + "crossattn_layer_norm.", cfg, residual=res)
    h = attention(h, enc, sd, prefix + "crossattention.", nh)
    h = h + res
    return h.permute(0, 2, 1).reshape(B, C, Hh, Ww)


def transformer_layer(h: Tensor, enc: Tensor, cond: Tensor, res: Optional[Tensor], sd: SD, prefix: str, cfg: dict):
    """``TransformerLayer`` :757-792 with ``GLUFeedForward`` :926-951; the residual stream is carried by the norms."""
    nh = cfg["num_attention_heads"]
    h, res = norm(h, sd, prefix + "attn_layer_norm.", cfg, residual=res)
    h = ada_ln(h, cond, sd, prefix + "self_attn_adaLN_modulation.")
    h = attention(h, h, sd, prefix + "attention.", nh)
    h, res = norm(h, sd, prefix + "crossattn_layer_norm.", cfg, residual=res)
    h = ada_ln(h, cond, sd, prefix + "cross_attn_adaLN_modulation.")
    h = attention(h, enc, sd, prefix + "crossattention.", nh)
    h, res = layer_norm(h, sd, prefix + "ffn.pre_mlp_layer_norm.", cfg, residual=res)
    h = ada_ln(h, cond, sd, prefix + "ffn.adaLN_modulation.")
    h = F.gelu(linear(h, sd, prefix + "ffn.wi_0.")) * linear(h, sd, prefix + "ffn.wi_1.")
    return linear(h, sd, prefix + "ffn.wo."), res


def uvit_forward(sd: SD, cfg: dict, input_ids: Tensor, encoder_hidden_states: Tensor, cond_embeds: Tensor,
                 micro_conds: Tensor, labels: Optional[Tensor] = None, label_smoothing: float = 0.0,
                 loss_weight: Optional[Tensor] = None):
    """``MaskGiTUViT_v2.forward`` :242-319 -> logits [B, S, codebook_size] (and the loss when labels are given)."""
    B, S = input_ids.shape
    side = int(S ** 0.5)
    enc = linear(encoder_hidden_states, sd, "encoder_proj.")                                   # :252
    enc, _ = norm(enc, sd, "encoder_proj_layer_norm.", cfg)
    micro = sinusoidal_encode(micro_conds.flatten(), cfg["micro_cond_encode_dim"]).reshape(B, -1)  # :255-256
    cond = torch.cat([cond_embeds, micro], dim=1)
    cond = linear(F.silu(linear(cond, sd, "cond_embed.0.")), sd, "cond_embed.2.")             # :258-260
    # ConvEmbed :485-500
    emb = F.embedding(input_ids.view(B, side, side), sd["embed.embeddings.weight"])
    emb, _ = norm(emb, sd, "embed.layer_norm.", cfg)
    h = F.conv2d(emb.permute(0, 3, 1, 2), sd["embed.conv.weight"], sd.get("embed.conv.bias"))
    # DownsampleBlock :506-541; force_down_up_sample (configs/research_run_512_with_downsample*.yaml): Norm2D -> 2x2 stride-2 conv (:510-514)
    down_up = bool(cfg.get("force_down_up_sample", False))
    if down_up:
        h = F.conv2d(norm2d(h, sd, "down_blocks.0.downsample.0.", cfg), sd["down_blocks.0.downsample.1.weight"],
                     sd.get("down_blocks.0.downsample.1.bias"), stride=2)
    for i in range(cfg["num_res_blocks"]):
        h = res_block(h, cond, sd, f"down_blocks.0.res_blocks.{i}.", cfg)
        h = attention_block_2d(h, enc, sd, f"down_blocks.0.attention_blocks.{i}.", cfg)
    Bc, C, Hh, Ww = h.shape
    h = h.permute(0, 2, 3, 1).reshape(B, Hh * Ww, C)
    h, _ = norm(h, sd, "project_to_hidden_norm.", cfg)
    h = linear(h, sd, "project_to_hidden.")
    res = None
    for li in range(cfg["num_hidden_layers"]):
        h, res = transformer_layer(h, enc, cond, res, sd, f"transformer_layers.{li}.", cfg)
    h = h + res                                                                               # :288
    h, _ = norm(h, sd, "project_from_hidden_norm.", cfg)
    h = linear(h, sd, "project_from_hidden.")
    h = h.reshape(B, Hh, Ww, C).permute(0, 3, 1, 2)
    for i in range(cfg["num_res_blocks"]):                                                    # UpsampleBlock :544-583
        h = res_block(h, cond, sd, f"up_blocks.0.res_blocks.{i}.", cfg)
        h = attention_block_2d(h, enc, sd, f"up_blocks.0.attention_blocks.{i}.", cfg)
    if down_up:                                                                               # Norm2D -> 2x2 stride-2 transposed conv (:558-562)
        h = F.conv_transpose2d(norm2d(h, sd, "up_blocks.0.upsample.0.", cfg), sd["up_blocks.0.upsample.1.weight"],
                               sd.get("up_blocks.0.upsample.1.bias"), stride=2)
    # ConvMlmLayer :1002-1022 (1x1 convs = per-pixel linears)
    h = F.conv2d(h, sd["mlm_layer.conv1.weight"], sd.get("mlm_layer.conv1.bias"))
    h = norm2d(h, sd, "mlm_layer.layer_norm.", cfg)
    logits = F.conv2d(h, sd["mlm_layer.conv2.weight"], sd.get("mlm_layer.conv2.bias"))
    V = cfg["codebook_size"]
    logits = logits.permute(0, 2, 3, 1).reshape(B, -1, V)
    if labels is None:
        return logits
    # :303-317
    if loss_weight is None:
        loss = F.cross_entropy(logits.view(-1, V), labels.view(-1), ignore_index=-100, label_smoothing=label_smoothing)
    else:
        per = F.cross_entropy(logits.view(-1, V), labels.view(-1), ignore_index=-100, label_smoothing=label_smoothing,
                              reduction="none")
        lw = loss_weight.view(-1)
        loss = ((per * lw).sum(dim=-1) / lw.sum(dim=-1)).mean()
    return logits, loss


def uvit_loss_and_grads(sd: SD, cfg: dict, input_ids: Tensor, encoder_hidden_states: Tensor, cond_embeds: Tensor,
                        micro_conds: Tensor, labels: Tensor, label_smoothing: float = 0.0,
                        loss_weight: Optional[Tensor] = None):
    """forward + autograd backward on the restatement: (logits, loss, {name: grad})"""
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    logits, loss = uvit_forward(leaf, cfg, input_ids, encoder_hidden_states, cond_embeds, micro_conds, labels,
                                label_smoothing, loss_weight)
    loss.backward()
    return logits.detach(), loss.detach(), {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaf.items()}


def generate2(sd: SD, cfg: dict, encoder_hidden_states: Tensor, cond_embeds: Tensor, micro_conds: Tensor, empty_embeds: Tensor,
              empty_cond_embeds: Tensor, timesteps: int, temperature, guidance_scale: float, noise, seq_len: int):
    """MaskGiTUViT_v2.generate2 (muse/modeling_transformer_v2.py:330-479), guidance_schedule None, cosine noise schedule, with
    the random draws explicit: noise[step] = (q_exp [B*S, codebook], u [B, S]) (see maskgit_oracle.sample_step).
    -> (final sampled ids, per-step raw samples = the reference's `intermediate`)"""
    from .maskgit_oracle import cosine_schedule, sample_step
    B, S, V = encoder_hidden_states.shape[0], seq_len, cfg["codebook_size"]
    mask_id = cfg["vocab_size"] - 1
    temperatures = (torch.linspace(temperature[0], temperature[1], timesteps) if isinstance(temperature, tuple)
                    else torch.linspace(temperature, 0.01, timesteps))                       # :359-362
    scales = torch.ones(timesteps) * guidance_scale                                          # :381
    input_ids = torch.ones((B, S), dtype=torch.long) * mask_id
    if micro_conds.shape[0] == 1:
        micro_conds = micro_conds.repeat(B, 1)
    if guidance_scale > 0:                                                                   # :386-412
        enc = torch.cat([encoder_hidden_states, empty_embeds.expand(B, -1, -1) if empty_embeds.shape[0] == 1 else empty_embeds])
        cond = torch.cat([cond_embeds, empty_cond_embeds.expand(B, -1) if empty_cond_embeds.shape[0] == 1 else empty_cond_embeds])
        micro = torch.cat([micro_conds, micro_conds], dim=0)
    else:
        enc, cond, micro = encoder_hidden_states, cond_embeds, micro_conds
    intermediate, sampled = [], input_ids
    for step in range(timesteps):
        model_input = torch.cat([input_ids] * 2) if guidance_scale > 0 else input_ids
        out = uvit_forward(sd, cfg, model_input, enc, cond, micro)
        out = out[0] if isinstance(out, tuple) else out
        if guidance_scale > 0:
            cond_logits, uncond_logits = out.chunk(2)
            cond_logits, uncond_logits = cond_logits[..., :V], uncond_logits[..., :V]
            logits = uncond_logits + scales[step] * (cond_logits - uncond_logits)           # :431-436
        else:
            logits = out[..., :V]
        ratio = 1.0 * (step + 1) / timesteps
        sched = int((S * cosine_schedule(torch.tensor(ratio))).floor())
        q, u = noise[step]
        raw, sampled, input_ids = sample_step(logits, input_ids, mask_id, temperatures[step], sched, q, u)
        intermediate.append(raw)
    return sampled, intermediate
