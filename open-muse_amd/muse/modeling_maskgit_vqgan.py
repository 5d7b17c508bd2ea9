"""muse.MaskGitVQGAN for MI355X: the reference's class surface (muse/modeling_maskgit_vqgan.py:351-414) over HIP kernels.

Same constructor kwargs, config keys, state_dict names/shapes and methods (`encode`, `decode`, `decode_code`,
`get_code`, `get_soft_code`, `forward`).  Underneath, activations are NHWC and every layer is a libmuse_hip call:
implicit-GEMM MFMA convolution (SAME padding, bias, residual add and the decoder's nearest x2 upsample folded in),
GroupNorm+SiLU (two-pass f64 statistics), 2x2 average pool, and the VQ lookup (f32 MFMA distance GEMM with the
reference's addmm rounding order + first-index argmin).

compute_dtype = torch.float32 (default) keeps the reference's precision (the training script runs the frozen VQGAN in
f32 outside autocast: training/train_maskgit_imagenet.py:306,369) on the exact-f32 MFMA; torch.bfloat16 is the fast mode
(bf16 activations/weights, f32 accumulation, f32 VQ lookup).  The VQGAN is a frozen tokenizer on this path: there is no
backward through it, and no CPU path.
"""
from __future__ import annotations

import math
import os
from typing import Tuple

import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from ._hip import MuseHipError
from .modeling_utils import ConfigMixin, ModelMixin, register_to_config


class _Conv(nn.Module):
    def __init__(self, cin, cout, k, bias):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))  # nn.Conv2d default init
        if bias:
            bound = 1.0 / math.sqrt(cin * k * k)
            nn.init.uniform_(self.bias, -bound, bound)


class _GNInput:
    """an f32 NHWC activation with the affine form of its GroupNorm, for the convolution that applies norm + SiLU itself
    (ops.conv2d_nhwc_gn_split2)"""
    __slots__ = ("x", "scale", "shift")

    def __init__(self, x, scale, shift):
        self.x, self.scale, self.shift = x, scale, shift


class _Norm(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))


class _Res(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.norm1 = _Norm(cin)
        self.conv1 = _Conv(cin, cout, 3, False)
        self.norm2 = _Norm(cout)
        self.conv2 = _Conv(cout, cout, 3, False)
        if cin != cout:
            self.nin_shortcut = _Conv(cout, cout, 1, False)


class _Down(nn.Module):
    def __init__(self, cin, cout, nb):
        super().__init__()
        self.block = nn.ModuleList([_Res(cin if i == 0 else cout, cout) for i in range(nb)])


class _Up(nn.Module):
    def __init__(self, cin, cout, nb, upsample):
        super().__init__()
        self.block = nn.ModuleList([_Res(cin if i == 0 else cout, cout) for i in range(nb)])
        if upsample:
            self.upsample_conv = _Conv(cout, cout, 3, True)


class _Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        hc, mult, nb = cfg.hidden_channels, tuple(cfg.channel_mult), cfg.num_res_blocks
        self.conv_in = _Conv(cfg.num_channels, hc, 3, False)
        in_mult = (1,) + mult
        self.down = nn.ModuleList([_Down(hc * in_mult[i], hc * mult[i], nb) for i in range(len(mult))])
        mid = hc * mult[-1]
        self.mid = nn.ModuleList([_Res(mid, mid) for _ in range(nb)])
        self.norm_out = _Norm(mid)
        self.conv_out = _Conv(mid, cfg.z_channels, 1, True)


class _Decoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        hc, mult, nb = cfg.hidden_channels, tuple(cfg.channel_mult), cfg.num_res_blocks
        nres = len(mult)
        mid = hc * mult[-1]
        self.conv_in = _Conv(cfg.z_channels, mid, 3, True)
        self.mid = nn.ModuleList([_Res(mid, mid) for _ in range(nb)])
        ups = []
        for lvl in range(nres):
            cin = hc * mult[-1] if lvl == nres - 1 else hc * mult[lvl + 1]
            ups.append(_Up(cin, hc * mult[lvl], nb, lvl != 0))
        self.up = nn.ModuleList(ups)
        self.norm_out = _Norm(hc * mult[0])
        self.conv_out = _Conv(hc * mult[0], cfg.num_channels, 3, True)


class _Quantizer(nn.Module):
    def __init__(self, n, d):
        super().__init__()
        self.embedding = nn.Embedding(n, d)
        self.embedding.weight.data.uniform_(-1.0 / n, 1.0 / n)  # reference :265


class _ConvEngine:
    """What both tokenizers (MaskGitVQGAN here, the taming VQGANModel in modeling_taming_vqgan.py) share: NHWC activations,
    packed convolution weights per compute dtype, the convolution / GroupNorm dispatch over libmuse_hip, and the rule that
    there is no CPU path."""

    def _init_engine(self):
        self.compute_dtype = torch.float32
        self.fuse_gn_stats = True   # convolution / pooling epilogues produce the next GroupNorm's statistics
        self.fuse_gn_apply = os.environ.get("MUSE_GN_FUSE", "1") != "0"   # GroupNorm + SiLU applied inside the consuming patch-slab convolution
        self.direct_conv_in = os.environ.get("MUSE_CONV_IN_DIRECT", "1") != "0"   # bf16x3 mode: conv_in as a direct exact-f32 kernel
        self.dma_conv = True        # "bf16x3" mode: 3x3 convs after GroupNorm run as the LDS-DMA kernel on pre-split planes
        # bf16x3 mode, decoder: norm_out -> swish -> conv_out (-> 3 image channels) as one direct exact-f32 kernel; the up-sampling
        # convolutions on the LDS-DMA kernel (nearest x2 written as the convolution's operand planes)
        self.direct_conv_out = os.environ.get("MUSE_CONV_OUT_DIRECT", "1") != "0"
        self.upsample_split = os.environ.get("MUSE_UPSAMPLE_SPLIT", "1") != "0"
        self._packed = {}

    def _check(self, t):
        if not t.is_cuda:
            raise MuseHipError(f"{type(self).__name__} (MI355X build) has no CPU path: move the model and inputs to the GPU")

    # ---- weight packing: [Cout,Cin,k,k] -> [Cout,k,k,Cin_pad] in the compute dtype ------------------------------------
    def set_compute_dtype(self, dtype):
        """torch.float32: exact-f32 MFMA convolutions (bit-level parity mode); "bf16x3": f32 activations, convolutions as
        3 bf16 MFMAs per product (f32-class, tighter than the TF32 the reference's cuDNN path uses by default);
        torch.bfloat16: bf16 activations and weights."""
        if dtype not in (torch.float32, torch.bfloat16, "bf16x3"):
            raise ValueError('compute dtype must be torch.float32, torch.bfloat16 or "bf16x3"')
        self.compute_dtype = dtype
        return self

    def _act_dtype(self):
        return torch.bfloat16 if self.compute_dtype == torch.bfloat16 else torch.float32

    def _apply(self, fn, recurse=True):
        self._packed = {}
        return super()._apply(fn, recurse)

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        self._packed = {}
        return super().load_state_dict(state_dict, strict=strict, assign=assign)

    def _cpad(self, c, cd):
        q = 4 if cd == torch.float32 else 8
        return (c + q - 1) // q * q

    def _w(self, conv: _Conv, cd):
        key = (id(conv), cd)
        hit = self._packed.get(key)
        if hit is None:
            w = conv.weight.data
            cout, cin, k, _ = w.shape
            cp = self._cpad(cin, cd)
            wp = torch.zeros((cout, k, k, cp), dtype=torch.float32, device=w.device)
            wp[..., :cin] = w.permute(0, 2, 3, 1)
            wp = wp.contiguous()
            if cd == torch.bfloat16:
                wp = ops.cast_to_bf16(wp)
            elif cd == "bf16x3":
                wp = ops.split_bf16(wp)
            hit = (wp, cp, cout, k, None if conv.bias is None else conv.bias.data.float().contiguous())
            self._packed[key] = hit
        return hit

    def _conv_in(self, x, conv, B, H, W, cd, cpad):
        """the image-to-features convolution: in the bf16x3 mode a direct exact-f32 kernel (3 input channels are no matrix-core
        problem; ops.conv_in_direct), otherwise the mode's implicit GEMM"""
        cout, cin, k, _ = conv.weight.shape
        if cd == "bf16x3" and self.direct_conv_in and ops.conv_in_direct_ok(cin, cout, k, cpad, W):
            key = (id(conv), "direct")
            hit = self._packed.get(key)
            if hit is None:
                w4 = torch.zeros((cout, 9, 4), dtype=torch.float32, device=conv.weight.device)
                w4[:, :, :cin] = conv.weight.data.float().permute(0, 2, 3, 1).reshape(cout, 9, cin)
                hit = (w4.contiguous(), None if conv.bias is None else conv.bias.data.float().contiguous())
                self._packed[key] = hit
            return ops.conv_in_direct(x, hit[0], B, H, W, cin, cpad, cout, bias=hit[1], gn_groups=32 if self.fuse_gn_stats else 0)
        return self._conv(x, conv, B, H, W, cd, gn_next=True)

    def _conv(self, x, conv, B, H, W, cd, residual=None, upsample=False, gn_next=False):
        wp, cp, cout, k, bias = self._w(conv, cd)
        if isinstance(x, _GNInput):   # GroupNorm + SiLU + split inside the convolution (_gn_for)
            return ops.conv2d_nhwc_gn_split2(x.x, x.scale, x.shift, wp[0], wp[1], B, H, W, cp, cout, bias=bias, residual=residual,
                                             gn_groups=32 if (gn_next and self.fuse_gn_stats) else 0)
        if isinstance(x, tuple):   # (hi, lo) planes from _gn_for: the LDS-DMA bf16x3 convolution
            return ops.conv2d_nhwc_split2(x[0], x[1], wp[0], wp[1], B, H, W, cp, cout, bias=bias, residual=residual,
                                          gn_groups=32 if (gn_next and self.fuse_gn_stats) else 0)
        if cd == "bf16x3":
            if (self.dma_conv and self.upsample_split and not upsample and k == 3 and x.dtype == torch.float32 and x.is_contiguous()
                    and cp == x.shape[-1] and ops.conv_split2_ok(B, H, W, cp, cout, k) and ops.conv_slab_ok(H, W, cp)):
                # a 3x3 layer whose input is a plain f32 tensor (Decoder.conv_in on the gathered codebook rows): split once into the
                # operand planes and run on the LDS-DMA kernel (the register-staged kernel took 1.1 ms for this 39 GFLOP layer at
                # 64 x 16 x 16: too few workgroups for its tiling)
                hi, lo = ops.split_f32(x)
                return ops.conv2d_nhwc_split2(hi, lo, wp[0], wp[1], B, H, W, cp, cout, bias=bias, residual=residual,
                                              gn_groups=32 if (gn_next and self.fuse_gn_stats) else 0)
            return ops.conv2d_nhwc_split(x, wp[0], wp[1], B, H, W, cp, cout, k, bias=bias, residual=residual, upsample=upsample,
                                         gn_groups=32 if (gn_next and self.fuse_gn_stats) else 0)
        return ops.conv2d_nhwc(x, wp, B, H, W, cp, cout, k, bias=bias, residual=residual, upsample=upsample)

    def _upsample_conv(self, h, conv: _Conv, B, H, W, cd):
        """UpsamplingBlock (:141-149): nearest x2 then 3x3 convolution; H, W = the OUTPUT size.  bf16x3 mode: the up-sampled tensor is
        written once as the (hi, lo) operand planes and the convolution runs on the LDS-DMA kernel; other modes / shapes gather the
        nearest neighbour inside the convolution."""
        cout, cin, k, _ = conv.weight.shape
        if (cd == "bf16x3" and self.dma_conv and self.upsample_split and cin % 8 == 0 and h.dtype == torch.float32
                and ops.conv_split2_ok(B, H, W, cin, cout, k)):
            planes = ops.upsample2x_split(h, B, H // 2, W // 2, cin)
            return self._conv(planes, conv, B, H, W, cd, gn_next=True)
        return self._conv(h, conv, B, H, W, cd, upsample=True, gn_next=True)

    def _conv_out(self, h, norm: _Norm, conv: _Conv, B, H, W, cd):
        """norm_out -> swish -> conv_out of a decoder (:236-240).  bf16x3 mode with few output channels (the image): one direct
        exact-f32 kernel that normalises / activates its own input from the producer's GroupNorm sums."""
        cout, cin, k, _ = conv.weight.shape
        stats = getattr(h, "_gn_stats", None)
        if (cd == "bf16x3" and self.direct_conv_out and stats is not None and h.dtype == torch.float32
                and ops.conv_out_direct_ok(H, W, cin, cout, k)):
            key = (id(conv), "direct_out")
            hit = self._packed.get(key)
            if hit is None:
                hit = (conv.weight.data.float().permute(0, 2, 3, 1).reshape(cout, 9, cin).contiguous(),
                       None if conv.bias is None else conv.bias.data.float().contiguous())
                self._packed[key] = hit
            sc, sh = ops.groupnorm_scale_shift(stats, norm.weight.data, norm.bias.data, B, H * W, cin, groups=32, eps=1e-6)
            return ops.conv_out_direct(h, sc, sh, hit[0], hit[1], B, H, W, cin, cout)
        return self._conv(self._gn_for(h, norm, conv, B, H, W, cd), conv, B, H, W, cd)

    def _gn(self, x, norm: _Norm, B, HW, C, silu=True):
        return ops.groupnorm_silu_nhwc(x, norm.weight.data, norm.bias.data, B, HW, C, groups=32, eps=1e-6, silu=silu)

    def _gn_for(self, x, norm: _Norm, conv: _Conv, B, H, W, cd):
        """GroupNorm+SiLU feeding `conv`: in "bf16x3" mode the result is written directly as the (hi, lo) bf16 operand
        planes of the LDS-DMA convolution when the layer qualifies (3x3, Cin % 32 == 0)"""
        cout, cin, k, _ = conv.weight.shape
        stats = getattr(x, "_gn_stats", None)
        if (cd == "bf16x3" and self.dma_conv and self.fuse_gn_apply and stats is not None and ops.conv_slab_ok(H, W, cin)
                and ops.conv_gn_split2_ok(B, H, W, cin, cout, k)):
            # the convolution normalises, activates and splits its own input while staging it: no apply pass, no operand planes
            sc, sh = ops.groupnorm_scale_shift(stats, norm.weight.data, norm.bias.data, B, H * W, cin, groups=32, eps=1e-6)
            return _GNInput(x, sc, sh)
        if cd == "bf16x3" and self.dma_conv and ops.conv_split2_ok(B, H, W, cin, cout, k) and (256 % (cin // 4)) == 0:
            return ops.groupnorm_silu_nhwc_split(x, norm.weight.data, norm.bias.data, B, H * W, cin, groups=32, eps=1e-6, silu=True,
                                                 stats=getattr(x, "_gn_stats", None))
        return self._gn(x, norm, B, H * W, cin)


class MaskGitVQGAN(_ConvEngine, ModelMixin, ConfigMixin):
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
        z_channels: int = 256,
        num_embeddings: int = 1024,
        quantized_embed_dim: int = 256,
        dropout: float = 0.0,
        resample_with_conv: bool = True,
        commitment_cost: float = 0.25,
    ):
        super().__init__()
        if dropout != 0.0:
            raise NotImplementedError("dropout > 0 in the VQGAN is outside the MI355X hot-path build")
        if z_channels != quantized_embed_dim:
            raise ValueError("z_channels must equal quantized_embed_dim")
        self.config.num_resolutions = len(channel_mult)
        self.config.reduction_factor = 2 ** (self.config.num_resolutions - 1)
        self.config.latent_size = resolution // self.config.reduction_factor
        self.encoder = _Encoder(self.config)
        self.decoder = _Decoder(self.config)
        self.quantize = _Quantizer(num_embeddings, quantized_embed_dim)
        self._init_engine()

    def _res(self, x, blk: _Res, B, H, W, cd):
        cin = blk.conv1.weight.shape[1]
        cout = blk.conv1.weight.shape[0]
        # gn_next: the convolution's epilogue also leaves the GroupNorm statistics of its output for the next norm layer
        h = self._conv(self._gn_for(x, blk.norm1, blk.conv1, B, H, W, cd), blk.conv1, B, H, W, cd, gn_next=True)
        if cin == cout:
            return self._conv(self._gn_for(h, blk.norm2, blk.conv2, B, H, W, cd), blk.conv2, B, H, W, cd, residual=x, gn_next=True)
        h = self._conv(self._gn_for(h, blk.norm2, blk.conv2, B, H, W, cd), blk.conv2, B, H, W, cd)
        # reference quirk (:82-85): the "shortcut" is a 1x1 conv of the conv2 output, out = h + nin(h)
        return self._conv(h, blk.nin_shortcut, B, H, W, cd, residual=h, gn_next=True)

    # ---- encoder / decoder ----------------------------------------------------------------------------------------------
    @torch.no_grad()
    def _encode_nhwc(self, pixel_values):
        """NCHW f32 pixels -> z as [B*h*w, z_channels] f32 (NHWC flattened) and (B, h, w)."""
        self._check(pixel_values)
        cd = self.compute_dtype
        enc = self.encoder
        B, C, H, W = pixel_values.shape
        x = ops.nchw_to_nhwc(pixel_values.float(), self._act_dtype(), self._cpad(C, cd))
        h = self._conv_in(x, enc.conv_in, B, H, W, cd, self._cpad(C, cd))
        nres = self.config.num_resolutions
        for lvl, down in enumerate(enc.down):
            for blk in down.block:
                h = self._res(h, blk, B, H, W, cd)
            if lvl != nres - 1:
                h = ops.avgpool2x2_nhwc(h, B, H, W, h.shape[-1], gn_groups=32 if (cd == "bf16x3" and self.fuse_gn_stats) else 0)
                H, W = H // 2, W // 2
        for blk in enc.mid:
            h = self._res(h, blk, B, H, W, cd)
        h = self._gn_for(h, enc.norm_out, enc.conv_out, B, H, W, cd)
        z = self._conv(h, enc.conv_out, B, H, W, cd)
        z = z.view(B * H * W, -1)
        if z.dtype != torch.float32:
            z = ops.cast_to_f32(z)
        return z, (B, H, W)

    @torch.no_grad()
    def _decode_nhwc(self, zq, B, H, W):
        """zq: [B, H, W, z_channels] in the compute dtype -> NCHW f32 image."""
        cd = self.compute_dtype
        dec = self.decoder
        nres = self.config.num_resolutions
        h = self._conv(zq, dec.conv_in, B, H, W, cd, gn_next=True)
        for blk in dec.mid:
            h = self._res(h, blk, B, H, W, cd)
        for lvl in reversed(range(nres)):
            up = dec.up[lvl]
            for blk in up.block:
                h = self._res(h, blk, B, H, W, cd)
            if lvl != 0:
                H, W = H * 2, W * 2
                h = self._upsample_conv(h, up.upsample_conv, B, H, W, cd)
        out = self._conv_out(h, dec.norm_out, dec.conv_out, B, H, W, cd)
        return ops.nhwc_to_nchw(out, self.config.num_channels)

    def _codebook(self):
        return self.quantize.embedding.weight.data

    # ---- public surface (reference :380-414) ---------------------------------------------------------------------------
    @torch.no_grad()
    def encode(self, pixel_values, return_loss=False):
        z, (B, H, W) = self._encode_nhwc(pixel_values)
        cb = self._codebook()
        idx = ops.vq_nearest(z, cb)
        zq_rows = ops.gather_rows(cb, idx, torch.float32)                    # == one-hot @ codebook (:280-284)
        zq = ops.nhwc_to_nchw(zq_rows.view(B, H, W, -1), zq_rows.shape[1])   # (B, C, h, w) like :299
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
        cd = self.compute_dtype
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
        soft = ops.vq_neg_distances_scaled(z, cb, 1.0 / float(temp))       # -distances / temp (:329-331) ...
        ops.softmax_(soft, soft.shape[0], soft.shape[1], soft.shape[1])     # ... softmax over the codebook, in place
        # stochastic: one categorical draw per token (torch's generator, like the reference :333); else the exact argmin of :335
        code = torch.multinomial(soft, 1) if stochastic else ops.vq_nearest(z, cb)
        return soft.view(B, H * W, -1), code.view(B, H * W)

    def forward(self, pixel_values, return_loss=False):
        enc = self.encode(pixel_values, return_loss)
        rec = self.decode(enc[0])
        out = (rec, enc[0], enc[1])
        if return_loss:
            out = out + (enc[2],)
        return out
