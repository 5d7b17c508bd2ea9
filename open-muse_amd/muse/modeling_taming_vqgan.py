"""muse.VQGANModel (the taming-transformers tokenizer of the text-to-image configs) for MI355X.

Reference: muse/modeling_taming_vqgan.py:512-585 (configs/cc12m_uvit_clip.yaml:19-21 `vq_model.type: "vqgan"`, 8192 codes) —
same constructor kwargs, config keys, state_dict names / shapes and methods (`encode`, `decode`, `decode_code`, `get_code`,
`get_soft_code`, `forward`).  SURVEY.md section 8 row f4.

It runs on the same engine as MaskGitVQGAN (modeling_maskgit_vqgan._ConvEngine: NHWC activations, implicit-GEMM MFMA
convolutions with bias / residual in the epilogue, GroupNorm statistics from the producing kernel, f32 | "bf16x3" | bf16
compute modes); what this architecture adds:

  * every convolution has a bias, and the residual shortcut (1x1 `nin_shortcut`) applies to the block INPUT (:128-133);
  * `Downsample` = zero-pad bottom/right + 3x3 stride-2 convolution (:55-59): the stride-2 gather mode of the conv kernels
    (`upsample=2`), no padded copy;  `Upsample` = nearest x2 folded into the next conv's gather (:40-44);
  * `AttnBlock` (:148-174): GroupNorm without SiLU, one stacked 1x1 q|k|v GEMM, single-head attention over the pixels
    (scores / softmax / PV as batched MFMA GEMMs, f32 softmax), 1x1 proj with the residual in its epilogue.  A level's
    attention blocks only run when the level has more than one of them (`len(self.attn) > 1`, :210,:249) - kept;
  * 1x1 `quant_conv` / `post_quant_conv` around the quantizer (same distance math as MaskGitVQGAN: ops.vq_nearest).

Frozen tokenizer: no backward, no CPU path.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Tuple

import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from .modeling_maskgit_vqgan import _Conv, _ConvEngine, _Norm, _Quantizer
from .modeling_utils import ConfigMixin, ModelMixin, register_to_config


class _Res(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.norm1 = _Norm(cin)
        self.conv1 = _Conv(cin, cout, 3, True)
        self.norm2 = _Norm(cout)
        self.conv2 = _Conv(cout, cout, 3, True)
        if cin != cout:
            self.nin_shortcut = _Conv(cin, cout, 1, True)


class _Attn(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.norm = _Norm(c)
        self.q, self.k, self.v, self.proj_out = (_Conv(c, c, 1, True) for _ in range(4))


class _Resample(nn.Module):
    """Downsample / Upsample: holds `.conv` only when resample_with_conv"""

    def __init__(self, c, with_conv):
        super().__init__()
        if with_conv:
            self.conv = _Conv(c, c, 3, True)


class _Level(nn.Module):
    def __init__(self, cin, cout, nblocks, attn, resample_name, with_conv):
        super().__init__()
        self.block = nn.ModuleList([_Res(cin if i == 0 else cout, cout) for i in range(nblocks)])
        self.attn = nn.ModuleList([_Attn(cout) for _ in range(nblocks)] if attn else [])
        if resample_name:
            setattr(self, resample_name, _Resample(cout, with_conv))


class _Mid(nn.Module):
    def __init__(self, c, no_attn):
        super().__init__()
        self.block_1 = _Res(c, c)
        if not no_attn:
            self.attn_1 = _Attn(c)
        self.block_2 = _Res(c, c)


class _Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        hc, mult, nb = cfg.hidden_channels, tuple(cfg.channel_mult), cfg.num_res_blocks
        nres, attn_res = len(mult), tuple(cfg.attn_resolutions)
        self.conv_in = _Conv(cfg.num_channels, hc, 3, True)
        in_mult, cur, levels = (1,) + mult, cfg.resolution, []
        for i in range(nres):
            last = i == nres - 1
            levels.append(_Level(hc * in_mult[i], hc * mult[i], nb, cur in attn_res, None if last else "downsample",
                                 cfg.resample_with_conv))
            if not last:
                cur //= 2
        self.down = nn.ModuleList(levels)
        mid = hc * mult[-1]
        self.mid = _Mid(mid, cfg.no_attn_mid_block)
        self.norm_out = _Norm(mid)
        self.conv_out = _Conv(mid, cfg.z_channels, 3, True)


class _Decoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        hc, mult, nb = cfg.hidden_channels, tuple(cfg.channel_mult), cfg.num_res_blocks
        nres, attn_res = len(mult), tuple(cfg.attn_resolutions)
        mid = hc * mult[-1]
        self.conv_in = _Conv(cfg.z_channels, mid, 3, True)
        self.mid = _Mid(mid, cfg.no_attn_mid_block)
        cur, levels = cfg.resolution // 2 ** (nres - 1), [None] * nres
        for i in reversed(range(nres)):
            cin = mid if i == nres - 1 else hc * mult[i + 1]
            levels[i] = _Level(cin, hc * mult[i], nb + 1, cur in attn_res, "upsample" if i != 0 else None, cfg.resample_with_conv)
            if i != 0:
                cur *= 2
        self.up = nn.ModuleList(levels)
        self.norm_out = _Norm(hc * mult[0])
        self.conv_out = _Conv(hc * mult[0], cfg.num_channels, 3, True)


class VQGANModel(_ConvEngine, ModelMixin, ConfigMixin):
    _cast_selects_compute_mode = True

    def _compute_mode_for(self, dtype):
        """`.half()` / `.to(dtype=fp16 | bf16)` / `from_pretrained(torch_dtype=...)` (benchmark/muse_perf.py:251-252 casts the VAE to
        fp16): the parameters stay f32 and the tokenizer takes its fast f32-CLASS mode "bf16x3" (three bf16 MFMA products per f32
        product: token indices and images stay at the parity bar, which a plain half-precision tokenizer would not); a cast to f32 / f64
        selects exact f32.  (The reference's pipeline keeps the VAE in fp32, pipeline_muse.py:62: PipelineMuse.to does not cast it.)"""
        if dtype is None or not dtype.is_floating_point:
            return False
        self.set_compute_dtype("bf16x3" if dtype in (torch.float16, torch.bfloat16) else torch.float32)
        return True

    @register_to_config
    def __init__(
        self,
        resolution: int = 256,
        num_channels: int = 3,
        hidden_channels: int = 128,
        channel_mult: Tuple = (1, 1, 2, 2, 4),
        num_res_blocks: int = 2,
        attn_resolutions: int = (16,),
        no_attn_mid_block: bool = False,
        z_channels: int = 256,
        num_embeddings: int = 1024,
        quantized_embed_dim: int = 256,
        dropout: float = 0.0,
        resample_with_conv: bool = True,
        commitment_cost: float = 0.25,
    ):
        super().__init__()
        if dropout != 0.0:
            raise NotImplementedError("dropout > 0 in the VQGAN is outside the MI355X hot-path build (frozen tokenizer)")
        self.config.num_resolutions = len(channel_mult)
        self.config.reduction_factor = 2 ** (self.config.num_resolutions - 1)
        self.config.latent_size = resolution // self.config.reduction_factor
        self.encoder = _Encoder(self.config)
        self.decoder = _Decoder(self.config)
        self.quantize = _Quantizer(num_embeddings, quantized_embed_dim)
        self.quant_conv = _Conv(z_channels, quantized_embed_dim, 1, True)
        self.post_quant_conv = _Conv(quantized_embed_dim, z_channels, 1, True)
        self._init_engine()

    @property
    def num_embeddings(self):
        return self.config.num_embeddings

    # ---- blocks ---------------------------------------------------------------------------------------------------------
    def _res(self, x, blk: _Res, B, H, W, cd):
        """ResnetBlock (:117-135): both convolutions leave the next GroupNorm's statistics; the shortcut rides conv2's epilogue"""
        h = self._conv(self._gn_for(x, blk.norm1, blk.conv1, B, H, W, cd), blk.conv1, B, H, W, cd, gn_next=True)
        sc = self._conv(x, blk.nin_shortcut, B, H, W, cd) if hasattr(blk, "nin_shortcut") else x
        return self._conv(self._gn_for(h, blk.norm2, blk.conv2, B, H, W, cd), blk.conv2, B, H, W, cd, residual=sc, gn_next=True)

    def _qkv(self, att: _Attn, cd):
        """q | k | v 1x1 convolutions as ONE stacked [3C, C] weight (packed once per compute dtype)"""
        key = (id(att), "qkv")
        hit = self._packed.get(key)
        if hit is None:   # a stand-in with the two attributes _w reads; dropped with the rest of the packed weights
            hit = SimpleNamespace(weight=torch.cat([att.q.weight.data, att.k.weight.data, att.v.weight.data], 0),
                                  bias=torch.cat([att.q.bias.data, att.k.bias.data, att.v.bias.data], 0))
            self._packed[key] = hit
        return hit

    def _attn(self, x, att: _Attn, B, H, W, cd):
        """AttnBlock (:148-174) on the NHWC rows [B*HW, C]: softmax(q k^T / sqrt(C)) v per image, then proj_out + x"""
        C, HW = att.norm.weight.shape[0], H * W
        h = self._gn(x, att.norm, B, HW, C, silu=False)
        qkv = self._conv(h, self._qkv(att, cd), B, H, W, cd).view(B * HW, 3 * C)
        scores = torch.empty((B, HW, HW), dtype=torch.float32, device=x.device)
        ops.gemm(qkv, qkv, scores, HW, HW, C, la=0, lb=0, lda=3 * C, ldb=3 * C, ldc=HW, b_off=C, alpha=float(int(C) ** -0.5),
                 batch=B, sA=(HW * 3 * C, 0), sB=(HW * 3 * C, 0), sC=(HW * HW, 0))
        ops.softmax_(scores, B * HW, HW, HW)
        probs = scores if qkv.dtype == torch.float32 else ops.cast_to_bf16(scores)
        ctx = torch.empty((B, H, W, C), dtype=qkv.dtype, device=x.device)
        ops.gemm(probs, qkv, ctx, HW, C, HW, la=0, lb=1, lda=HW, ldb=3 * C, ldc=C, b_off=2 * C, batch=B,
                 sA=(HW * HW, 0), sB=(HW * 3 * C, 0), sC=(HW * C, 0))
        return self._conv(ctx, att.proj_out, B, H, W, cd, residual=x, gn_next=True)

    def _level(self, x, lvl: _Level, B, H, W, cd):
        run_attn = len(lvl.attn) > 1   # the reference's condition (:210,:249): a lone attention block is never applied
        for i, blk in enumerate(lvl.block):
            x = self._res(x, blk, B, H, W, cd)
            if run_attn:
                x = self._attn(x, lvl.attn[i], B, H, W, cd)
        return x

    def _mid(self, x, mid: _Mid, B, H, W, cd):
        x = self._res(x, mid.block_1, B, H, W, cd)
        if hasattr(mid, "attn_1"):
            x = self._attn(x, mid.attn_1, B, H, W, cd)
        return self._res(x, mid.block_2, B, H, W, cd)

    # ---- encoder / decoder ------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def _encode_nhwc(self, pixel_values):
        """NCHW f32 pixels -> quant_conv(encoder(x)) as [B*h*w, quantized_embed_dim] f32 rows and (B, h, w)"""
        self._check(pixel_values)
        cd, enc = self.compute_dtype, self.encoder
        B, C, H, W = pixel_values.shape
        if (H | W) % (1 << (self.config.num_resolutions - 1)):
            raise ValueError("image height / width must be multiples of the reduction factor")
        x = ops.nchw_to_nhwc(pixel_values.float(), self._act_dtype(), self._cpad(C, cd))
        h = self._conv_in(x, enc.conv_in, B, H, W, cd, self._cpad(C, cd))
        for lvl in enc.down:
            h = self._level(h, lvl, B, H, W, cd)
            if hasattr(lvl, "downsample"):
                H, W = H // 2, W // 2
                if hasattr(lvl.downsample, "conv"):
                    h = self._conv(h, lvl.downsample.conv, B, H, W, cd, upsample=2, gn_next=True)
                else:
                    h = ops.avgpool2x2_nhwc(h, B, 2 * H, 2 * W, h.shape[-1], gn_groups=32 if (cd == "bf16x3" and self.fuse_gn_stats) else 0)
        h = self._mid(h, enc.mid, B, H, W, cd)
        h = self._conv(self._gn_for(h, enc.norm_out, enc.conv_out, B, H, W, cd), enc.conv_out, B, H, W, cd)
        z = self._conv(h, self.quant_conv, B, H, W, cd).view(B * H * W, -1)
        if z.dtype != torch.float32:
            z = ops.cast_to_f32(z)
        return z, (B, H, W)

    @torch.no_grad()
    def _decode_nhwc(self, zq, B, H, W):
        """zq: [B, H, W, quantized_embed_dim] in the activation dtype -> NCHW f32 image"""
        cd, dec = self.compute_dtype, self.decoder
        h = self._conv(zq, self.post_quant_conv, B, H, W, cd)
        h = self._conv(h, dec.conv_in, B, H, W, cd, gn_next=True)
        h = self._mid(h, dec.mid, B, H, W, cd)
        for lvl in reversed(dec.up):
            h = self._level(h, lvl, B, H, W, cd)
            if hasattr(lvl, "upsample"):
                H, W = H * 2, W * 2
                if hasattr(lvl.upsample, "conv"):
                    h = self._upsample_conv(h, lvl.upsample.conv, B, H, W, cd)
                else:   # plain nearest x2 (resample_with_conv = False)
                    if h.shape[-1] % (8 if h.dtype == torch.bfloat16 else 4) == 0:
                        h = ops.upsample2x(h.contiguous(), B, H // 2, W // 2, h.shape[-1])
                    else:   # (channel counts that are not whole 16-byte vectors: no shipped configuration)
                        h = F.interpolate(h.permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest").permute(0, 2, 3, 1).contiguous()
        h = self._conv_out(h, dec.norm_out, dec.conv_out, B, H, W, cd)
        return ops.nhwc_to_nchw(h, self.config.num_channels)

    def _codebook(self):
        return self.quantize.embedding.weight.data

    # ---- public surface (reference :552-585) --------------------------------------------------------------------------------
    @torch.no_grad()
    def encode(self, pixel_values, return_loss=False):
        z, (B, H, W) = self._encode_nhwc(pixel_values)
        cb = self._codebook()
        idx = ops.vq_nearest(z, cb)
        zq_rows = ops.gather_rows(cb, idx, torch.float32)                    # == one-hot @ codebook (:443-446)
        zq = ops.nhwc_to_nchw(zq_rows.view(B, H, W, -1), zq_rows.shape[1])
        out = (zq, idx.view(B, H * W))
        if return_loss:
            zc = ops.nhwc_to_nchw(z.view(B, H, W, -1), z.shape[1])
            mse = torch.mean((zq - zc) ** 2)
            out = out + (mse + self.config.commitment_cost * mse,)
        return out

    @torch.no_grad()
    def decode(self, quantized_states):
        self._check(quantized_states)
        B, C, H, W = quantized_states.shape
        zq = ops.nchw_to_nhwc(quantized_states.float(), self._act_dtype(), C)
        return self._decode_nhwc(zq, B, H, W)

    @torch.no_grad()
    def decode_code(self, codebook_indices):
        self._check(codebook_indices)
        B, T = codebook_indices.shape
        side = int(math.sqrt(T))
        zq = ops.gather_rows(self._codebook(), codebook_indices.contiguous().view(-1), self._act_dtype())
        return self._decode_nhwc(zq.view(B, side, side, -1), B, side, side)

    @torch.no_grad()
    def get_code(self, pixel_values):
        z, (B, H, W) = self._encode_nhwc(pixel_values)
        return ops.vq_nearest(z, self._codebook()).view(B, H * W)

    @torch.no_grad()
    def get_soft_code(self, pixel_values, temp=1.0, stochastic=False):
        z, (B, H, W) = self._encode_nhwc(pixel_values)
        cb = self._codebook()
        dist = torch.cdist(z, cb).pow(2)  # adjacent feature (soft targets), torch op on the GPU
        soft = F.softmax(-dist / temp, dim=-1)
        code = torch.multinomial(soft, 1) if stochastic else ops.vq_nearest(z, cb)
        return soft.view(B, H * W, -1), code.view(B, H * W)

    def forward(self, pixel_values, return_loss=False):
        enc = self.encode(pixel_values, return_loss)
        rec = self.decode(enc[0])
        out = (rec, enc[0], enc[1])
        if return_loss:
            out = out + (enc[2],)
        return out
