"""Deterministic weights and inputs shared by the golden generator, the oracle tests and the GPU parity tests.

numpy's PCG64 stream is stable across numpy versions and machines, so the build container (where the real
reference is importable) and the GPU box (where it is not) construct bit-identical tensors from a seed.
"""
from __future__ import annotations

import numpy as np
import torch

# ---- tiny configs used for the committed golden vectors -------------------------------------------------------
TRANSFORMER_TINY = dict(
    vocab_size=48, hidden_size=32, num_hidden_layers=2, num_attention_heads=2, intermediate_size=64,
    max_position_embeddings=24, codebook_size=32, num_vq_tokens=16, num_classes=10,
    hidden_dropout=0.0, attention_dropout=0.0, layer_norm_eps=1e-5,
)
# head_dim 48 (= configs/imagenet.yaml's 768/16), odd sequence length
TRANSFORMER_HD48 = dict(
    vocab_size=80, hidden_size=96, num_hidden_layers=1, num_attention_heads=2, intermediate_size=160,
    max_position_embeddings=40, codebook_size=64, num_vq_tokens=36, num_classes=10,
    hidden_dropout=0.0, attention_dropout=0.0, layer_norm_eps=1e-6,
)
# ---- the general form of MaskGitTransformer (text conditioning, RMSNorm, plain pre-LN layers; oracle/ and muse/maskgit_general.py) ----
# the shipped text-to-image family in small (configs/cc12m.yaml, imagenet_text2image.yaml: rmsnorm, no NormFormer, codebook-wide logits)
TRANSFORMER_TEXT_TINY = dict(
    vocab_size=48, hidden_size=32, num_hidden_layers=2, num_attention_heads=2, intermediate_size=64,
    max_position_embeddings=16, codebook_size=32, num_vq_tokens=16, add_cross_attention=True, encoder_hidden_size=24,
    project_encoder_hidden_states=False, norm_type="rmsnorm", use_normformer=False, use_codebook_size_for_output=True,
    hidden_dropout=0.0, attention_dropout=0.0, layer_norm_eps=1e-6,
)
# the other branches: LayerNorm + NormFormer post-norms around both attentions, projected text states, MLM head without its norm
TRANSFORMER_TEXT_PROJ_TINY = dict(
    vocab_size=40, hidden_size=32, num_hidden_layers=2, num_attention_heads=2, intermediate_size=48,
    max_position_embeddings=12, codebook_size=32, num_vq_tokens=12, add_cross_attention=True, encoder_hidden_size=20,
    project_encoder_hidden_states=True, norm_type="layernorm", use_normformer=True, use_mlm_layernorm=False,
    hidden_dropout=0.0, attention_dropout=0.0, layer_norm_eps=1e-5,
)
# ... RMSNorm with NormFormer, no final norm, bare `to_logits` head, no text (class token)
TRANSFORMER_PLAIN_TINY = dict(
    vocab_size=48, hidden_size=32, num_hidden_layers=1, num_attention_heads=4, intermediate_size=64,
    max_position_embeddings=20, codebook_size=32, num_vq_tokens=16, num_classes=10, norm_type="rmsnorm", use_normformer=True,
    use_encoder_layernorm=False, use_mlm_layer=False, hidden_dropout=0.0, attention_dropout=0.0, layer_norm_eps=1e-6,
)
# use_bias=True: biases on every Linear and LayerNorm - once with LayerNorm + NormFormer + projected text states + the full MLM head,
# once with RMSNorm (no norm bias except the feed-forward's always-LayerNorm pre norm) and a bare to_logits head
TRANSFORMER_TEXT_BIAS_TINY = dict(
    vocab_size=40, hidden_size=32, num_hidden_layers=2, num_attention_heads=2, intermediate_size=48,
    max_position_embeddings=12, codebook_size=32, num_vq_tokens=12, add_cross_attention=True, encoder_hidden_size=24,
    project_encoder_hidden_states=True, norm_type="layernorm", use_normformer=True, use_bias=True,
    hidden_dropout=0.0, attention_dropout=0.0, layer_norm_eps=1e-5,
)
TRANSFORMER_RMS_BIAS_TINY = dict(
    vocab_size=48, hidden_size=32, num_hidden_layers=2, num_attention_heads=4, intermediate_size=64,
    max_position_embeddings=16, codebook_size=32, num_vq_tokens=16, add_cross_attention=True, encoder_hidden_size=24,
    project_encoder_hidden_states=False, norm_type="rmsnorm", use_normformer=False, use_mlm_layer=False, use_bias=True,
    use_codebook_size_for_output=True, hidden_dropout=0.0, attention_dropout=0.0, layer_norm_eps=1e-6,
)
# configs/cc12m.yaml:28-50 `model.transformer` with 2 of its 24 layers (T5-large states of width 1024, 77 tokens in the tests)
TRANSFORMER_CC12M_2L = dict(
    vocab_size=8256, max_position_embeddings=256, hidden_size=1024, num_hidden_layers=2, num_attention_heads=16,
    intermediate_size=4096, add_cross_attention=True, encoder_hidden_size=1024, project_encoder_hidden_states=False,
    codebook_size=8192, num_vq_tokens=256, initializer_range=0.02, norm_type="rmsnorm", layer_norm_eps=1e-6, use_normformer=False,
    use_encoder_layernorm=True, use_mlm_layer=True, use_mlm_layernorm=True, use_bias=False, hidden_dropout=0.0,
    attention_dropout=0.0, use_codebook_size_for_output=True,
)

VQGAN_TINY = dict(
    resolution=16, num_channels=3, hidden_channels=32, channel_mult=(1, 2, 2), num_res_blocks=1,
    z_channels=16, num_embeddings=32, quantized_embed_dim=16,
)

# taming VQGANModel (muse/modeling_taming_vqgan.py:512-550): attention at resolution 8 (down level 2, up level 2) and in the mid
# blocks, stride-2 conv downsampling, conv upsampling, quant_conv 24 -> 16 and back
TAMING_TINY = dict(
    resolution=32, num_channels=3, hidden_channels=32, channel_mult=(1, 2, 2), num_res_blocks=2, attn_resolutions=(8,),
    no_attn_mid_block=False, z_channels=24, num_embeddings=40, quantized_embed_dim=16, resample_with_conv=True,
)
# ... and the other branches: avg-pool / plain nearest resampling, no mid attention, no level attention
TAMING_TINY_POOL = dict(
    resolution=16, num_channels=3, hidden_channels=32, channel_mult=(1, 2), num_res_blocks=1, attn_resolutions=(),
    no_attn_mid_block=True, z_channels=16, num_embeddings=24, quantized_embed_dim=16, resample_with_conv=False,
)

# ---- full-size configs (SURVEY.md section 8 legend) --------------------------------------------------------------
TRANSFORMER_A = dict(  # README.md:90-100 ("hidden=512, 8 layers" in BASELINE.json)
    vocab_size=2025, hidden_size=512, num_hidden_layers=8, num_attention_heads=8, intermediate_size=2048,
    max_position_embeddings=257, codebook_size=1024, num_vq_tokens=256, num_classes=1000,
    hidden_dropout=0.0, attention_dropout=0.0, layer_norm_eps=1e-5,
)
TRANSFORMER_B = dict(  # configs/imagenet.yaml:23-42
    vocab_size=2048, hidden_size=768, num_hidden_layers=24, num_attention_heads=16, intermediate_size=3072,
    max_position_embeddings=264, codebook_size=1024, num_vq_tokens=256, num_classes=1000,
    initializer_range=0.02, norm_type="layernorm", layer_norm_eps=1e-6, use_normformer=True,
    use_encoder_layernorm=True, use_mlm_layer=True, use_mlm_layernorm=True, use_bias=False,
    hidden_dropout=0.0, attention_dropout=0.0,
)
VQGAN_F16 = dict(  # muse/modeling_maskgit_vqgan.py:352-367 defaults
    resolution=256, num_channels=3, hidden_channels=128, channel_mult=(1, 1, 2, 2, 4), num_res_blocks=2,
    z_channels=256, num_embeddings=1024, quantized_embed_dim=256,
)


# BASELINE.json config 4: configs/cc12m_uvit_clip.yaml:29-54 `model.transformer` + block_num_heads=16 (the YAML predates
# MaskGiTUViT_v2, whose default 12 does not divide its 1024 block channels).  728 725 504 parameters; forward 275.10 GFLOP per
# sample at 256 tokens, 1137.05 at 1024.  bench.py's config-4 legs build exactly this (tests/test_surface.py checks).
UVIT_CC12M = dict(
    vocab_size=8256, hidden_size=1024, intermediate_size=4096, num_hidden_layers=22, num_attention_heads=16,
    max_position_embeddings=256, in_channels=512, block_out_channels=(1024,), num_res_blocks=3, patch_size=1,
    encoder_hidden_size=768, add_cross_attention=True, project_encoder_hidden_states=False, codebook_size=8192, num_vq_tokens=256,
    initializer_range=0.02, norm_type="rmsnorm", layer_norm_eps=1e-6, use_normformer=False, use_encoder_layernorm=True,
    use_bias=False, hidden_dropout=0.0, attention_dropout=0.0, use_codebook_size_for_output=True, block_num_heads=16,
)


def transformer_shapes(cfg: dict) -> dict:
    """state_dict template of muse.MaskGitTransformer (reference muse/modeling_transformer.py:1083-1200 for the
    optional members: cross-attention blocks, text projection, NormFormer norms, final norm, MLM head)."""
    H, I, V, P = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"], cfg["max_position_embeddings"]
    cross, nf = bool(cfg.get("add_cross_attention", False)), bool(cfg.get("use_normformer", True))
    kv = cfg.get("encoder_hidden_size", 1024)
    out = cfg["codebook_size"] if cfg.get("use_codebook_size_for_output", False) else V
    s = {"embed.word_embeddings.weight": (V, H), "embed.position_embeddings.weight": (P, H)}
    if cfg.get("project_encoder_hidden_states", False):
        s["encoder_proj.weight"] = (H, kv)
        s["encoder_proj_layer_norm.weight"] = (H,)
        kv = H
    for i in range(cfg["num_hidden_layers"]):
        p = f"transformer_layers.{i}."
        s[p + "attn_layer_norm.weight"] = (H,)
        for n in ("query", "key", "value", "out"):
            s[p + f"attention.{n}.weight"] = (H, H)
        if nf:
            s[p + "post_attn_layer_norm.weight"] = (H,)
        if cross:
            s[p + "crossattn_layer_norm.weight"] = (H,)
            for n, c in (("query", H), ("key", kv), ("value", kv), ("out", H)):
                s[p + f"crossattention.{n}.weight"] = (H, c)
            if nf:
                s[p + "post_crossattn_layer_norm.weight"] = (H,)
        s[p + "ffn.pre_mlp_layer_norm.weight"] = (H,)
        s[p + "ffn.wi_0.weight"] = (I, H)
        s[p + "ffn.wi_1.weight"] = (I, H)
        if nf:
            s[p + "ffn.mid_mlp_layer_norm.weight"] = (I,)
        s[p + "ffn.wo.weight"] = (H, I)
    if cfg.get("use_encoder_layernorm", True):
        s["encoder_layer_norm.weight"] = (H,)
    if cfg.get("use_mlm_layer", True):
        s["mlm_layer.mlm_dense.weight"] = (H, H)
        if cfg.get("use_mlm_layernorm", True):
            s["mlm_layer.mlm_ln.weight"] = (H,)
        s["mlm_layer.to_logits.weight"] = (out, H)
    else:
        s["to_logits.weight"] = (out, H)
    if cfg.get("use_bias", False):
        # every nn.Linear and every LayerNorm gets a bias (muse/modeling_transformer.py:130, :170-176, :770-778, :973-977, :1155);
        # RMSNorm has none, and the feed-forward's pre_mlp_layer_norm is a LayerNorm whatever norm_type says (:768-770)
        ln = cfg.get("norm_type", "layernorm") == "layernorm"
        for k, shp in list(s.items()):
            if k.startswith("embed."):
                continue
            if len(shp) == 2 or ln or "pre_mlp_layer_norm" in k:
                s[k[:-len("weight")] + "bias"] = (shp[0],)
    return s


def vqgan_shapes(cfg: dict) -> dict:
    """state_dict template of muse.MaskGitVQGAN."""
    hc, mult, nb = cfg["hidden_channels"], tuple(cfg["channel_mult"]), cfg["num_res_blocks"]
    nres = len(mult)
    s = {}

    def res(prefix, cin, cout):
        s[prefix + "norm1.weight"] = (cin,)
        s[prefix + "norm1.bias"] = (cin,)
        s[prefix + "conv1.weight"] = (cout, cin, 3, 3)
        s[prefix + "norm2.weight"] = (cout,)
        s[prefix + "norm2.bias"] = (cout,)
        s[prefix + "conv2.weight"] = (cout, cout, 3, 3)
        if cin != cout:
            s[prefix + "nin_shortcut.weight"] = (cout, cout, 1, 1)

    s["encoder.conv_in.weight"] = (hc, cfg["num_channels"], 3, 3)
    in_mult = (1,) + mult
    for lvl in range(nres):
        cin, cout = hc * in_mult[lvl], hc * mult[lvl]
        for b in range(nb):
            res(f"encoder.down.{lvl}.block.{b}.", cin, cout)
            cin = cout
    mid = hc * mult[-1]
    for b in range(nb):
        res(f"encoder.mid.{b}.", mid, mid)
    s["encoder.norm_out.weight"] = (mid,)
    s["encoder.norm_out.bias"] = (mid,)
    s["encoder.conv_out.weight"] = (cfg["z_channels"], mid, 1, 1)
    s["encoder.conv_out.bias"] = (cfg["z_channels"],)

    s["decoder.conv_in.weight"] = (mid, cfg["z_channels"], 3, 3)
    s["decoder.conv_in.bias"] = (mid,)
    for b in range(nb):
        res(f"decoder.mid.{b}.", mid, mid)
    for lvl in range(nres):
        cin = hc * mult[-1] if lvl == nres - 1 else hc * mult[lvl + 1]
        cout = hc * mult[lvl]
        for b in range(nb):
            res(f"decoder.up.{lvl}.block.{b}.", cin, cout)
            cin = cout
        if lvl != 0:
            s[f"decoder.up.{lvl}.upsample_conv.weight"] = (cout, cout, 3, 3)
            s[f"decoder.up.{lvl}.upsample_conv.bias"] = (cout,)
    s["decoder.norm_out.weight"] = (hc * mult[0],)
    s["decoder.norm_out.bias"] = (hc * mult[0],)
    s["decoder.conv_out.weight"] = (cfg["num_channels"], hc * mult[0], 3, 3)
    s["decoder.conv_out.bias"] = (cfg["num_channels"],)
    s["quantize.embedding.weight"] = (cfg["num_embeddings"], cfg["quantized_embed_dim"])
    return s


def taming_shapes(cfg: dict) -> dict:
    """state_dict template of the taming muse.VQGANModel (muse/modeling_taming_vqgan.py; every convolution has a bias)"""
    hc, mult, nb = cfg["hidden_channels"], tuple(cfg["channel_mult"]), cfg["num_res_blocks"]
    nres, attn_res, with_conv = len(mult), tuple(cfg.get("attn_resolutions", (16,))), cfg.get("resample_with_conv", True)
    s = {}

    def conv(prefix, cout, cin, k):
        s[prefix + "weight"] = (cout, cin, k, k)
        s[prefix + "bias"] = (cout,)

    def norm(prefix, c):
        s[prefix + "weight"] = (c,)
        s[prefix + "bias"] = (c,)

    def res(prefix, cin, cout):
        norm(prefix + "norm1.", cin)
        conv(prefix + "conv1.", cout, cin, 3)
        norm(prefix + "norm2.", cout)
        conv(prefix + "conv2.", cout, cout, 3)
        if cin != cout:
            conv(prefix + "nin_shortcut.", cout, cin, 1)

    def attn(prefix, c):
        norm(prefix + "norm.", c)
        for n in ("q", "k", "v", "proj_out"):
            conv(prefix + n + ".", c, c, 1)

    def mid(prefix, c):
        res(prefix + "block_1.", c, c)
        if not cfg.get("no_attn_mid_block", False):
            attn(prefix + "attn_1.", c)
        res(prefix + "block_2.", c, c)

    conv("encoder.conv_in.", hc, cfg["num_channels"], 3)
    in_mult = (1,) + mult
    cur = cfg["resolution"]
    for lvl in range(nres):
        cin, cout = hc * in_mult[lvl], hc * mult[lvl]
        for b in range(nb):
            res(f"encoder.down.{lvl}.block.{b}.", cin, cout)
            cin = cout
            if cur in attn_res:
                attn(f"encoder.down.{lvl}.attn.{b}.", cout)
        if lvl != nres - 1:
            if with_conv:
                conv(f"encoder.down.{lvl}.downsample.conv.", cout, cout, 3)
            cur //= 2
    m = hc * mult[-1]
    mid("encoder.mid.", m)
    norm("encoder.norm_out.", m)
    conv("encoder.conv_out.", cfg["z_channels"], m, 3)

    conv("decoder.conv_in.", m, cfg["z_channels"], 3)
    mid("decoder.mid.", m)
    cur = cfg["resolution"] // 2 ** (nres - 1)
    for lvl in reversed(range(nres)):
        cin = hc * mult[-1] if lvl == nres - 1 else hc * mult[lvl + 1]
        cout = hc * mult[lvl]
        for b in range(nb + 1):
            res(f"decoder.up.{lvl}.block.{b}.", cin, cout)
            cin = cout
            if cur in attn_res:
                attn(f"decoder.up.{lvl}.attn.{b}.", cout)
        if lvl != 0:
            if with_conv:
                conv(f"decoder.up.{lvl}.upsample.conv.", cout, cout, 3)
            cur *= 2
    norm("decoder.norm_out.", hc * mult[0])
    conv("decoder.conv_out.", cfg["num_channels"], hc * mult[0], 3)
    s["quantize.embedding.weight"] = (cfg["num_embeddings"], cfg["quantized_embed_dim"])
    conv("quant_conv.", cfg["quantized_embed_dim"], cfg["z_channels"], 1)
    conv("post_quant_conv.", cfg["z_channels"], cfg["quantized_embed_dim"], 1)
    return s


def fill_state_dict(shapes: dict, seed: int, kind: str) -> dict:
    """Seeded fp32 weights, filled in sorted-key order.

    Norm scales are 1 + 0.1 N(0,1) (so a dropped/mis-indexed scale is visible), norm biases 0.1 N(0,1),
    linear / embedding weights N(0, std) with a std that keeps activations O(1) through the depth, and the
    VQ codebook N(0,1) * 0.5 so that nearest-code margins are far above fp32 round-off.
    """
    rng = np.random.default_rng(seed)
    sd = {}
    for k in sorted(shapes):
        shp = shapes[k]
        x = rng.standard_normal(shp).astype(np.float32)
        if "norm" in k or k.endswith("_ln.weight"):
            x = (1.0 + 0.1 * x) if k.endswith("weight") else 0.1 * x
        elif k == "quantize.embedding.weight":
            x = 0.5 * x
        elif k.endswith("bias"):
            x = 0.05 * x
        elif kind == "transformer":
            fan_in = shp[-1]
            x = x * (0.08 if "embeddings" in k else 1.0 / np.sqrt(fan_in))
        else:  # conv weight (Cout, Cin, k, k)
            fan_in = shp[1] * shp[2] * shp[3]
            x = x * (1.0 / np.sqrt(fan_in))
        sd[k] = torch.from_numpy(np.ascontiguousarray(x.astype(np.float32)))
    return sd


def transformer_inputs(cfg: dict, batch: int, seed: int):
    """Seeded (input_ids, labels) shaped like the output of prepare_inputs_and_labels."""
    rng = np.random.default_rng(seed)
    S = cfg["num_vq_tokens"]
    cb, V = cfg["codebook_size"], cfg["vocab_size"]
    tokens = rng.integers(0, cb, size=(batch, S))
    mask = rng.random((batch, S)) < 0.55
    mask[:, 0] = True
    input_ids = np.where(mask, V - 1, tokens)
    labels = np.where(mask, tokens, -100)
    cls = rng.integers(0, cfg["num_classes"], size=(batch, 1)) + cb
    input_ids = np.concatenate([cls, input_ids], axis=1)
    labels = np.concatenate([np.full((batch, 1), -100), labels], axis=1)
    return torch.from_numpy(input_ids.astype(np.int64)), torch.from_numpy(labels.astype(np.int64))


def transformer_text_inputs(cfg: dict, batch: int, text_len: int, seed: int):
    """Seeded (input_ids [B, S], labels [B, S], encoder_hidden_states [B, text_len, encoder_hidden_size]) of a text-conditioned
    step: no class token (training/train_muse.py:685-750), labels only inside the codebook."""
    rng = np.random.default_rng(seed)
    S, cb, V = cfg["num_vq_tokens"], cfg["codebook_size"], cfg["vocab_size"]
    tokens = rng.integers(0, cb, size=(batch, S))
    mask = rng.random((batch, S)) < 0.55
    mask[:, 0] = True
    input_ids = np.where(mask, V - 1, tokens)
    labels = np.where(mask, tokens, -100)
    enc = rng.standard_normal((batch, text_len, cfg["encoder_hidden_size"])).astype(np.float32)
    return (torch.from_numpy(input_ids.astype(np.int64)), torch.from_numpy(labels.astype(np.int64)), torch.from_numpy(enc))


def images(batch: int, res: int, seed: int) -> torch.Tensor:
    """Seeded synthetic images in [0,1] (ToTensor range, training/data.py:117-133), smooth + noise."""
    rng = np.random.default_rng(seed)
    x = rng.random((batch, 3, res, res)).astype(np.float32)
    return torch.from_numpy(x)


def uniforms(shape, seed: int) -> torch.Tensor:
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.random(shape).astype(np.float32))


def subsample(t, n: int = 4096):
    """every k-th element of the flattened tensor (k = numel // n, at least 1): the compact form in which the full-size goldens
    (make_golden.py::golden_transformer_full / golden_vqgan_full) store large outputs.  Works on torch tensors and numpy arrays."""
    flat = t.reshape(-1)
    return flat[:: max(1, flat.shape[0] // n)]


def fill_by_shapes(shapes: dict, seed: int) -> dict:
    """Seeded fp32 weights for ANY state-dict template {name: shape} (used for the full-size MaskGiTUViT, whose template is read off
    the instantiated model), filled in sorted-key order: 1-D norm gains 1 + 0.1 N(0,1), GlobalResponseNorm gamma / beta 0.1 N(0,1),
    every matrix / convolution / embedding N(0, 1 / fan_in) - also the tensors the reference zero-initialises (AdaLN mappers,
    mlm_layer.conv1), so that the conditioning paths carry signal."""
    rng = np.random.default_rng(seed)
    sd = {}
    for k in sorted(shapes):
        shp = tuple(shapes[k])
        x = rng.standard_normal(shp, dtype=np.float32)
        if k.endswith("gamma") or k.endswith("beta"):
            x *= 0.1
        elif len(shp) == 1:
            x = 1.0 + 0.1 * x
        else:
            fan_in = int(np.prod(shp[1:]))
            x *= np.float32(1.0 / np.sqrt(fan_in))
        sd[k] = torch.from_numpy(np.ascontiguousarray(x.astype(np.float32)))
    return sd


UVIT_FULL_GRAD_KEYS = [
    "embed.embeddings.weight", "embed.conv.weight", "cond_embed.0.weight", "encoder_proj.weight",
    "down_blocks.0.res_blocks.0.depthwise.weight", "down_blocks.0.res_blocks.1.channelwise.2.gamma",
    "down_blocks.0.res_blocks.2.adaLN_modulation.mapper.weight", "down_blocks.0.attention_blocks.0.crossattention.key.weight",
    "project_to_hidden.weight", "transformer_layers.0.attention.query.weight", "transformer_layers.10.crossattention.value.weight",
    "transformer_layers.10.ffn.adaLN_modulation.mapper.weight", "transformer_layers.21.ffn.wo.weight",
    "transformer_layers.21.attn_layer_norm.weight", "up_blocks.0.res_blocks.2.channelwise.0.weight",
    "up_blocks.0.attention_blocks.2.attention.out.weight", "mlm_layer.conv1.weight", "mlm_layer.conv2.weight",
]


def uvit_inputs(batch: int, seq: int, text_len: int, seed: int, vocab_size: int = 8256, codebook_size: int = 8192,
                encoder_hidden_size: int = 768, cond_embed_dim: int = 768):
    """seeded (input_ids, encoder_hidden_states, cond_embeds, micro_conds, labels) shaped like one train_muse.py batch"""
    rng = np.random.default_rng(seed)
    tokens = rng.integers(0, codebook_size, size=(batch, seq))
    masked = rng.random((batch, seq)) < 0.5
    input_ids = np.where(masked, vocab_size - 1, tokens).astype(np.int64)
    labels = np.where(masked, tokens, -100).astype(np.int64)
    enc = rng.standard_normal((batch, text_len, encoder_hidden_size), dtype=np.float32)
    cond = rng.standard_normal((batch, cond_embed_dim), dtype=np.float32)
    micro = np.tile(np.array([[256.0, 256.0, 0.0, 0.0, 6.0]], dtype=np.float32), (batch, 1))
    micro[1:, 2] = 16.0
    return tuple(torch.from_numpy(a) for a in (input_ids, enc, cond, micro, labels))


# ---- the weight average (muse/modeling_ema.py): tracked tensors and schedules of tests/golden/ema_tiny.npz ------------------------------
EMA_SHAPES = [(7,), (33, 5), (1030,), (515,), (3, 1, 2, 2), ()]      # (index 4 is frozen: requires_grad False -> copied, :134-135)
EMA_SCHEDULES = [dict(decay=0.9999, update_after_step=2, update_every=1),
                 dict(decay=0.999, min_decay=0.3, update_every=2, use_ema_warmup=True, inv_gamma=1.0, power=2 / 3)]


def ema_params(seed, step):
    """the tracked tensors at a training step: seeded, so the tests regenerate them instead of storing them"""
    g = torch.Generator().manual_seed(seed * 1000 + step)
    return [torch.randn(sh, generator=g) * (1.0 + 0.1 * i) for i, sh in enumerate(EMA_SHAPES)]


# ---- a real (tiny) CLIP text tower and tokenizer, built offline: what PipelineMuse loads as `text_encoder` / `tokenizer` -----------------
def tiny_clip(workdir, hidden=24, pooled=16, max_len=7, seed=5):
    """(transformers.CLIPTextModelWithProjection, transformers.CLIPTokenizer): character-level byte-BPE vocabulary without merges
    (514 entries), 3 layers; `hidden` = the U-ViT's encoder_hidden_size, `pooled` = its cond_embed_dim"""
    import json
    import os
    from transformers import CLIPTextConfig, CLIPTextModelWithProjection, CLIPTokenizer
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("\xa1"), ord("\xac") + 1)) + list(range(ord("\xae"), ord("\xff") + 1))
    cs, n = bs[:], 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    chars = [chr(c) for c in cs]
    vocab = {}
    for c in chars:
        vocab[c] = len(vocab)
    for c in chars:
        vocab[c + "</w>"] = len(vocab)
    vocab["<|startoftext|>"] = len(vocab)
    vocab["<|endoftext|>"] = len(vocab)
    os.makedirs(workdir, exist_ok=True)
    with open(os.path.join(workdir, "vocab.json"), "w") as f:
        json.dump(vocab, f)
    with open(os.path.join(workdir, "merges.txt"), "w") as f:
        f.write("#version: 0.2\n")
    tok = CLIPTokenizer(os.path.join(workdir, "vocab.json"), os.path.join(workdir, "merges.txt"), model_max_length=max_len)
    torch.manual_seed(seed)
    cfg = CLIPTextConfig(vocab_size=len(vocab), hidden_size=hidden, intermediate_size=2 * hidden, num_hidden_layers=3, num_attention_heads=2,
                         max_position_embeddings=max_len, projection_dim=pooled, bos_token_id=len(vocab) - 2, eos_token_id=len(vocab) - 1,
                         pad_token_id=len(vocab) - 1)
    return CLIPTextModelWithProjection(cfg).eval(), tok


# the two tokenizers of tests/golden/ckpt/ (reference-written checkpoints): two levels, one block per level - small files
VQGAN_CKPT = dict(VQGAN_TINY, channel_mult=(1, 1), num_res_blocks=1)
TAMING_CKPT = dict(TAMING_TINY, channel_mult=(1, 1), num_res_blocks=1, attn_resolutions=(16,))


# the schedule cases of tests/golden/lr_schedules.npz (make_golden.py::golden_lr_schedules keeps the same list)
LR_CASES = [("constant", {}), ("constant_with_warmup", dict(num_warmup_steps=5)), ("linear", dict(num_warmup_steps=4, num_training_steps=30)),
            ("cosine", dict(num_warmup_steps=6, num_training_steps=30)), ("cosine_with_restarts", dict(num_warmup_steps=3, num_training_steps=30, num_cycles=3)),
            ("polynomial", dict(num_warmup_steps=5, num_training_steps=30, power=2.0)), ("polynomial", dict(num_warmup_steps=0, num_training_steps=20))]

