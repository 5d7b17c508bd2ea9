"""CPU oracle for the taming `VQGANModel` tokenizer.

TEST INFRASTRUCTURE ONLY: imported by tests/, never by the product path.  A plain-torch, functional restatement of
/root/reference/muse/modeling_taming_vqgan.py working directly on a state_dict, every function citing the lines it
follows.  PARITY PINNED: tests/test_oracle_golden.py checks it against tests/golden/taming_*.npz, which
tests/golden/make_golden.py produced by running the real reference `muse.VQGANModel` in the build container.

Differences from the MaskGit tokenizer (oracle/maskgit_oracle.py) that matter for parity: every convolution has a bias;
the residual shortcut is applied to the block INPUT (:128-133); down/up-sampling are (pad bottom/right + stride-2 conv) and
(nearest x2 + conv) when `resample_with_conv`; single-head attention over the pixels at `attn_resolutions` — applied in a
level only when that level holds MORE THAN ONE attention block (`len(self.attn) > 1`, :210,:249) — and in both mid blocks
unless `no_attn_mid_block`; 1x1 `quant_conv` / `post_quant_conv` around the quantizer.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


def _conv(x: Tensor, sd: SD, prefix: str, stride: int = 1, padding: int = 1) -> Tensor:
    return F.conv2d(x, sd[prefix + "weight"], sd[prefix + "bias"], stride=stride, padding=padding)


def _gn(x: Tensor, sd: SD, prefix: str) -> Tensor:
    return F.group_norm(x, 32, sd[prefix + "weight"], sd[prefix + "bias"], eps=1e-6)


def resnet_block(x: Tensor, sd: SD, prefix: str) -> Tensor:
    """ResnetBlock.forward, :117-135 (dropout 0; use_conv_shortcut is never set by the model)"""
    h = _conv(F.silu(_gn(x, sd, prefix + "norm1.")), sd, prefix + "conv1.")
    h = _conv(F.silu(_gn(h, sd, prefix + "norm2.")), sd, prefix + "conv2.")
    if prefix + "nin_shortcut.weight" in sd:
        x = _conv(x, sd, prefix + "nin_shortcut.", padding=0)
    return h + x


def attn_block(x: Tensor, sd: SD, prefix: str) -> Tensor:
    """AttnBlock.forward, :148-174: GroupNorm (no SiLU), 1x1 q/k/v, softmax(q k^T / sqrt(C)) over the pixels, 1x1 proj, + x"""
    h = _gn(x, sd, prefix + "norm.")
    q = _conv(h, sd, prefix + "q.", padding=0)
    k = _conv(h, sd, prefix + "k.", padding=0)
    v = _conv(h, sd, prefix + "v.", padding=0)
    b, c, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1)
    k = k.reshape(b, c, hh * ww)
    w = torch.bmm(q, k) * (int(c) ** -0.5)
    w = F.softmax(w, dim=2)
    v = v.reshape(b, c, hh * ww)
    h = torch.bmm(v, w.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return _conv(h, sd, prefix + "proj_out.", padding=0) + x


def _level(x: Tensor, sd: SD, prefix: str, nblocks: int) -> Tensor:
    nattn = sum(1 for i in range(nblocks) if f"{prefix}attn.{i}.norm.weight" in sd)
    for i in range(nblocks):
        x = resnet_block(x, sd, f"{prefix}block.{i}.")
        if nattn > 1:   # reference quirk: a lone attention block of a level is constructed but never run
            x = attn_block(x, sd, f"{prefix}attn.{i}.")
    return x


def _mid(x: Tensor, sd: SD, prefix: str) -> Tensor:
    """MidBlock.forward, :280-285"""
    x = resnet_block(x, sd, prefix + "block_1.")
    if prefix + "attn_1.norm.weight" in sd:
        x = attn_block(x, sd, prefix + "attn_1.")
    return resnet_block(x, sd, prefix + "block_2.")


def encoder(sd: SD, cfg: dict, pixel_values: Tensor) -> Tensor:
    """Encoder.forward, :326-340 (+ DownsamplingBlock :246-255, Downsample :55-62)"""
    nres, nb = len(cfg["channel_mult"]), int(cfg["num_res_blocks"])
    h = _conv(pixel_values, sd, "encoder.conv_in.")
    for lvl in range(nres):
        h = _level(h, sd, f"encoder.down.{lvl}.", nb)
        if lvl != nres - 1:
            if cfg.get("resample_with_conv", True):
                h = _conv(F.pad(h, (0, 1, 0, 1)), sd, f"encoder.down.{lvl}.downsample.conv.", stride=2, padding=0)
            else:
                h = F.avg_pool2d(h, kernel_size=2, stride=2)
    h = _mid(h, sd, "encoder.mid.")
    return _conv(F.silu(_gn(h, sd, "encoder.norm_out.")), sd, "encoder.conv_out.")


def decoder(sd: SD, cfg: dict, z: Tensor) -> Tensor:
    """Decoder.forward, :385-401 (+ UpsamplingBlock :207-216, Upsample :40-44)"""
    nres, nb = len(cfg["channel_mult"]), int(cfg["num_res_blocks"])
    h = _conv(z, sd, "decoder.conv_in.")
    h = _mid(h, sd, "decoder.mid.")
    for lvl in reversed(range(nres)):
        h = _level(h, sd, f"decoder.up.{lvl}.", nb + 1)
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            if cfg.get("resample_with_conv", True):
                h = _conv(h, sd, f"decoder.up.{lvl}.upsample.conv.")
    return _conv(F.silu(_gn(h, sd, "decoder.norm_out.")), sd, "decoder.conv_out.")


def vq_distances(z_flat: Tensor, codebook: Tensor) -> Tensor:
    """VectorQuantizer.compute_distances, :464-477: addmm(|z|^2 + |e|^2, z, e^T, alpha=-2)"""
    zn = z_flat.pow(2.0).sum(dim=1, keepdim=True)
    en = codebook.t().pow(2.0).sum(dim=0, keepdim=True)
    return torch.addmm(zn + en, z_flat, codebook.t(), alpha=-2.0)


def encode(sd: SD, cfg: dict, pixel_values: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """VQGANModel.encode, :552-559 -> (pre-quantisation latents after quant_conv, z_q, indices)"""
    z = _conv(encoder(sd, cfg, pixel_values), sd, "quant_conv.", padding=0)
    cb = sd["quantize.embedding.weight"]
    B, C, Hh, Ww = z.shape
    zf = z.permute(0, 2, 3, 1).contiguous().reshape(-1, C)
    idx = torch.argmin(vq_distances(zf, cb), dim=1).reshape(B, -1)
    z_q = cb[idx].view(B, Hh, Ww, C).permute(0, 3, 1, 2).contiguous()   # == one-hot @ codebook (:443-446,:461)
    return z, z_q, idx


def decode(sd: SD, cfg: dict, z_q: Tensor) -> Tensor:
    """VQGANModel.decode, :561-564"""
    return decoder(sd, cfg, _conv(z_q, sd, "post_quant_conv.", padding=0))


def decode_code(sd: SD, cfg: dict, indices: Tensor) -> Tensor:
    """VQGANModel.decode_code, :566-569 (+ get_codebook_entry :479-485)"""
    B, T = indices.shape
    side = int(math.sqrt(T))
    z_q = sd["quantize.embedding.weight"][indices].reshape(B, side, side, -1).permute(0, 3, 1, 2)
    return decode(sd, cfg, z_q)
