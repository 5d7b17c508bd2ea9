"""muse.MaskGiTUViT_v2 / muse.MaskGiTUViT for MI355X — SURVEY.md §8 row a12 (config 4, reference
muse/modeling_transformer_v2.py:150-319).

STATUS (round 1): forward only (logits, plain / label-smoothed / per-token-weighted loss), f32 ("parity mode": exact-f32 MFMA
GEMMs, materialised attention = the reference's algorithm).  No backward yet: ``loss.backward()`` is not supported and the
returned loss carries no graph.  Same constructor kwargs (filtered like ``config_from_legacy_kwargs`` :127-147: unknown keys are
dropped), config keys, state_dict names / shapes and init as the reference.

Everything computes through libmuse_hip.so on channels-last rows ``[B * S, C]``:
  linears / 1x1 convs / attention GEMMs -> muse_gemm;  embedding -> muse_gather_rows;  RMSNorm / LayerNorm with the pre-norm
  residual stream -> muse_norm_res_fwd;  AdaLN -> muse_adaln_fwd;  depthwise 3x3 -> muse_dwconv3x3_nhwc;  GlobalResponseNorm ->
  muse_grn_fwd;  GELU / GLU / softmax / cross-entropy -> the MaskGit kernels;  micro-conditioning -> muse_sinusoidal_encode.
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np
import torch
from torch import nn

from . import ops
from ._hip import MuseHipError
from .modeling_utils import ConfigMixin, ModelMixin

# reference dataclass MaskGiTUViT_v2Config :79-124 (field -> default)
_DEFAULTS = dict(
    hidden_size=1024, use_bias=False, hidden_dropout=0.0,
    cond_embed_dim=768, micro_cond_encode_dim=256, micro_cond_embed_dim=1280, encoder_hidden_size=768,
    vocab_size=8256, mask_token_id=8255, codebook_size=8192,
    in_channels=768, block_out_channels=(768,), num_res_blocks=3, force_down_up_sample=False, block_num_heads=12,
    num_hidden_layers=22, num_attention_heads=16, attention_dropout=0.0,
    intermediate_size=2816, use_fused_mlp=False,
    norm_type="rmsnorm", layer_norm_eps=1e-6, ln_elementwise_affine=True, use_fused_residual_norm=False,
    add_cond_embeds=True, add_micro_cond_embeds=True,
)


class _Lin(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin))


class _NormW(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))


class _Norm2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.norm = _NormW(c)


class _Conv(nn.Module):
    def __init__(self, cin, cout, k, groups=1):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin // groups, k, k))


class _GRN(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.gamma = nn.Parameter(torch.zeros(1, 1, 1, c))
        self.beta = nn.Parameter(torch.zeros(1, 1, 1, c))


class _AdaLN(nn.Module):
    def __init__(self, hidden, c):
        super().__init__()
        self.mapper = _Lin(hidden, 2 * c)


class _Attn(nn.Module):
    def __init__(self, c, ctx):
        super().__init__()
        self.query, self.key, self.value, self.out = _Lin(c, c), _Lin(ctx, c), _Lin(ctx, c), _Lin(c, c)


class _ResBlock(nn.Module):
    def __init__(self, c, hidden):
        super().__init__()
        self.depthwise = _Conv(c, c, 3, groups=c)
        self.norm = _Norm2D(c)
        # nn.Sequential indices of the reference (:605-611): 0 Linear, 1 GELU, 2 GRN, 3 Dropout, 4 Linear
        self.channelwise = nn.ModuleDict({"0": _Lin(c, 4 * c), "2": _GRN(4 * c), "4": _Lin(4 * c, c)})
        self.adaLN_modulation = _AdaLN(hidden, c)


class _AttnBlock2D(nn.Module):
    def __init__(self, c, hidden):
        super().__init__()
        if hidden != c:
            self.kv_mapper = _Lin(hidden, c)
        self.attn_layer_norm = _NormW(c)
        self.attention = _Attn(c, c)
        self.crossattn_layer_norm = _NormW(c)
        self.crossattention = _Attn(c, c)


class _Block(nn.Module):
    def __init__(self, c, hidden, n):
        super().__init__()
        self.res_blocks = nn.ModuleList([_ResBlock(c, hidden) for _ in range(n)])
        self.attention_blocks = nn.ModuleList([_AttnBlock2D(c, hidden) for _ in range(n)])


class _FFN(nn.Module):
    def __init__(self, h, i):
        super().__init__()
        self.pre_mlp_layer_norm = _NormW(h)
        self.adaLN_modulation = _AdaLN(h, h)
        self.wi_0, self.wi_1, self.wo = _Lin(h, i), _Lin(h, i), _Lin(i, h)


class _Layer(nn.Module):
    def __init__(self, h, i):
        super().__init__()
        self.attn_layer_norm = _NormW(h)
        self.self_attn_adaLN_modulation = _AdaLN(h, h)
        self.attention = _Attn(h, h)
        self.crossattn_layer_norm = _NormW(h)
        self.crossattention = _Attn(h, h)
        self.cross_attn_adaLN_modulation = _AdaLN(h, h)
        self.ffn = _FFN(h, i)


class _Embed(nn.Module):
    def __init__(self, vocab, cin, cout):
        super().__init__()
        self.embeddings = nn.Embedding(vocab, cin)
        self.layer_norm = _NormW(cin)
        self.conv = _Conv(cin, cout, 1)


class _Mlm(nn.Module):
    def __init__(self, c, cin, codebook):
        super().__init__()
        self.conv1 = _Conv(c, cin, 1)
        self.layer_norm = _Norm2D(cin)
        self.conv2 = _Conv(cin, codebook, 1)


class MaskGiTUViT_v2(ModelMixin, ConfigMixin):
    def __init__(self, **kwargs):
        super().__init__()
        cfg = dict(_DEFAULTS)
        bnh = kwargs.get("block_num_heads", cfg["block_num_heads"])
        if isinstance(bnh, (tuple, list)):            # legacy kwargs :128-136
            if len(bnh) != 1:
                raise ValueError("block_num_heads must be an int or a 1-tuple")
            kwargs = {**kwargs, "block_num_heads": bnh[0]}
        for k in cfg:                                  # unknown keys are silently dropped, like the reference (:140-142)
            if k in kwargs:
                cfg[k] = kwargs[k]
        cfg["block_out_channels"] = list(cfg["block_out_channels"])
        self.register_to_config(**cfg)
        self.register_to_config(mask_token_id=cfg["vocab_size"] - 1)   # :159
        c = self.config
        if len(c.block_out_channels) != 1:
            raise ValueError("block_out_channels must have exactly one entry (reference :166)")
        if c.use_bias or c.force_down_up_sample or c.use_fused_mlp or not c.ln_elementwise_affine or c.norm_type != "rmsnorm":
            raise NotImplementedError("MI355X build of MaskGiTUViT_v2: only the shipped configuration family is built "
                                      "(rmsnorm, bias-free, GLU feed-forward, no forced down/up-sampling)")
        if c.hidden_dropout != 0.0 or c.attention_dropout != 0.0:
            raise NotImplementedError("dropout > 0 is outside the MI355X hot-path build")
        H, C, cin = c.hidden_size, c.block_out_channels[0], c.in_channels
        self.output_size = c.codebook_size
        self.encoder_proj = _Lin(c.encoder_hidden_size, H)
        self.encoder_proj_layer_norm = _NormW(H)
        self.embed = _Embed(c.vocab_size, cin, C)
        self.cond_embed = nn.ModuleDict({"0": _Lin(c.micro_cond_embed_dim + c.cond_embed_dim, H), "2": _Lin(H, H)})
        self.down_blocks = nn.ModuleList([_Block(C, H, c.num_res_blocks)])
        self.project_to_hidden_norm = _NormW(C)
        self.project_to_hidden = _Lin(C, H)
        self.transformer_layers = nn.ModuleList([_Layer(H, c.intermediate_size) for _ in range(c.num_hidden_layers)])
        self.project_from_hidden_norm = _NormW(H)
        self.project_from_hidden = _Lin(H, C)
        self.up_blocks = nn.ModuleList([_Block(C, H, c.num_res_blocks)])
        self.mlm_layer = _Mlm(C, cin, c.codebook_size)
        self._init_weights()

    def _init_weights(self):
        """reference :205-237: trunc_normal(0.02) for Linear / Conv / Embedding, ones for norm gains, then the special cases"""
        for m in self.modules():
            if isinstance(m, (_Lin, _Conv)):
                nn.init.trunc_normal_(m.weight, std=0.02)
            elif isinstance(m, nn.Embedding):
                nn.init.trunc_normal_(m.weight, std=0.02)
        nn.init.xavier_uniform_(self.embed.conv.weight, 0.02)
        nn.init.normal_(self.embed.embeddings.weight, std=float(np.sqrt(1 / self.config.vocab_size)))
        nn.init.constant_(self.mlm_layer.conv1.weight, 0)
        self.mlm_layer.conv2.weight.data = self.embed.embeddings.weight.data[: self.config.codebook_size, :, None, None].clone()
        for m in self.modules():
            if isinstance(m, _AdaLN):
                nn.init.constant_(m.mapper.weight, 0)

    # ---- forward --------------------------------------------------------------------------------------------------------
    @staticmethod
    def _f(p):
        return p.data if p.dtype == torch.float32 else p.data.float()

    def _norm(self, x, mod, mode=0, residual=None, want_pre=False):
        return ops.norm_res_fwd(x, self._f(mod.weight), float(self.config.layer_norm_eps), mode, residual=residual, want_pre=want_pre)

    def _attention(self, x, ctx, att: _Attn, B, Sq, Skv, nh, residual=None):
        """reference Attention :834-915, materialised: scores = alpha q k^T (batched per head), softmax, P v, out projection"""
        Cq = x.shape[1]
        hd = Cq // nh
        q = ops.linear(x, self._f(att.query.weight))
        k = ops.linear(ctx, self._f(att.key.weight))
        v = ops.linear(ctx, self._f(att.value.weight))
        Sp = (Skv + 7) // 8 * 8
        P = torch.empty((B * nh, Sq, Sp), dtype=torch.float32, device=x.device)
        alpha = 1.0 / float(torch.sqrt(torch.tensor(hd, dtype=torch.float32)))
        ops.gemm(q, k, P, Sq, Skv, hd, la=0, lb=0, lda=Cq, ldb=Cq, ldc=Sp, alpha=alpha, batch=B * nh, zdiv=nh,
                 sA=(Sq * Cq, hd), sB=(Skv * Cq, hd), sC=(nh * Sq * Sp, Sq * Sp))
        ops.softmax_(P, B * nh * Sq, Skv, Sp)
        o = torch.empty((B * Sq, Cq), dtype=torch.float32, device=x.device)
        ops.gemm(P, v, o, Sq, hd, Skv, la=0, lb=1, lda=Sp, ldb=Cq, ldc=Cq, batch=B * nh, zdiv=nh,
                 sA=(nh * Sq * Sp, Sq * Sp), sB=(Skv * Cq, hd), sC=(Sq * Cq, hd))
        return ops.linear(o, self._f(att.out.weight), residual=residual)

    def _adaln(self, x, mod: _AdaLN, scond, B):
        return ops.adaln_fwd(x, ops.linear(scond, self._f(mod.mapper.weight)), B)

    def _res_block(self, h, blk: _ResBlock, scond, B, side):
        C = h.shape[1]
        d = ops.dwconv3x3_nhwc(h, self._f(blk.depthwise.weight).contiguous(), B, side, side, C)
        n, _ = self._norm(d, blk.norm.norm)
        a = ops.gelu_fwd(ops.linear(n, self._f(blk.channelwise["0"].weight)))
        grn = blk.channelwise["2"]
        g = ops.grn_fwd(a, self._f(grn.gamma).reshape(-1).contiguous(), self._f(grn.beta).reshape(-1).contiguous(), B, side * side)
        x = ops.linear(g, self._f(blk.channelwise["4"].weight), residual=h)          # + x_res (:616)
        return self._adaln(x, blk.adaLN_modulation, scond, B)

    def _attn_block(self, h, blk: _AttnBlock2D, enc, senc, B, S, L):
        ctx = ops.linear(senc, self._f(blk.kv_mapper.weight)) if hasattr(blk, "kv_mapper") else enc   # :815-816
        nh = self.config.block_num_heads
        n1, _ = self._norm(h, blk.attn_layer_norm)                                    # residual = h (:819)
        a1 = self._attention(n1, ctx, blk.attention, B, S, L, nh)
        n2, res = self._norm(a1, blk.crossattn_layer_norm, residual=h, want_pre=True)  # :822
        return self._attention(n2, ctx, blk.crossattention, B, S, L, nh, residual=res)  # + residual (:824)

    @torch.no_grad()
    def forward(self, input_ids, encoder_hidden_states, cond_embeds, micro_conds, labels=None, label_smoothing=0.0,
                loss_weight=None):
        c = self.config
        for t in (input_ids, encoder_hidden_states, cond_embeds, micro_conds):
            if not t.is_cuda:
                raise MuseHipError("MaskGiTUViT_v2 (MI355X build) has no CPU path: move the model and inputs to the GPU")
        B, S = input_ids.shape
        side = int(S ** 0.5)
        if side * side != S:
            raise ValueError("the token sequence must be a square grid")
        L = encoder_hidden_states.shape[1]
        H, C = c.hidden_size, c.block_out_channels[0]
        f = self._f
        # text states :252-253
        enc = ops.linear(encoder_hidden_states.reshape(B * L, -1).float().contiguous(), f(self.encoder_proj.weight))
        enc, _ = self._norm(enc, self.encoder_proj_layer_norm)
        senc = ops.silu_fwd(enc) if C != H else None
        # conditioning :255-260
        micro = ops.sinusoidal_encode(micro_conds, c.micro_cond_encode_dim).reshape(B, -1)
        cond = torch.cat([cond_embeds.float(), micro], dim=1).contiguous()
        cond = ops.linear(ops.silu_fwd(ops.linear(cond, f(self.cond_embed["0"].weight))), f(self.cond_embed["2"].weight))
        scond = ops.silu_fwd(cond)                                                    # every AdaLN sees silu(cond) (:1032)
        # ConvEmbed :485-500
        emb = ops.gather_rows(f(self.embed.embeddings.weight), input_ids.reshape(-1).contiguous(), torch.float32)
        emb, _ = self._norm(emb, self.embed.layer_norm)
        h = ops.linear(emb, f(self.embed.conv.weight).reshape(C, -1))
        blk = self.down_blocks[0]
        for i in range(c.num_res_blocks):
            h = self._res_block(h, blk.res_blocks[i], scond, B, side)
            h = self._attn_block(h, blk.attention_blocks[i], enc, senc, B, S, L)
        n, _ = self._norm(h, self.project_to_hidden_norm)
        t = ops.linear(n, f(self.project_to_hidden.weight))
        res = None
        nh = c.num_attention_heads
        for lyr in self.transformer_layers:                                           # TransformerLayer :757-792
            n, res = self._norm(t, lyr.attn_layer_norm, residual=res, want_pre=True)
            m = self._adaln(n, lyr.self_attn_adaLN_modulation, scond, B)
            a = self._attention(m, m, lyr.attention, B, S, S, nh)
            n, res = self._norm(a, lyr.crossattn_layer_norm, residual=res, want_pre=True)
            m = self._adaln(n, lyr.cross_attn_adaLN_modulation, scond, B)
            a = self._attention(m, enc, lyr.crossattention, B, S, L, nh)
            n, res = self._norm(a, lyr.ffn.pre_mlp_layer_norm, mode=1, residual=res, want_pre=True)   # LayerNorm (:928)
            m = self._adaln(n, lyr.ffn.adaLN_modulation, scond, B)
            w01 = torch.cat([f(lyr.ffn.wi_0.weight), f(lyr.ffn.wi_1.weight)], dim=0)
            t = ops.linear(ops.glu_fwd(ops.linear(m, w01)), f(lyr.ffn.wo.weight))
        n, _ = self._norm(t, self.project_from_hidden_norm, residual=res)             # (t + residual) then norm (:288-290)
        h = ops.linear(n, f(self.project_from_hidden.weight))
        blk = self.up_blocks[0]
        for i in range(c.num_res_blocks):
            h = self._res_block(h, blk.res_blocks[i], scond, B, side)
            h = self._attn_block(h, blk.attention_blocks[i], enc, senc, B, S, L)
        # ConvMlmLayer :1002-1022
        y = ops.linear(h, f(self.mlm_layer.conv1.weight).reshape(c.in_channels, C))
        y, _ = self._norm(y, self.mlm_layer.layer_norm.norm)
        V = c.codebook_size
        Vp = (V + 7) // 8 * 8
        logits_p = torch.empty((B * S, Vp), dtype=torch.float32, device=y.device)
        ops.gemm(y, f(self.mlm_layer.conv2.weight).reshape(V, -1), logits_p, B * S, V, c.in_channels, lda=c.in_channels,
                 ldb=c.in_channels, ldc=Vp)
        logits = logits_p.view(B, S, Vp) if Vp == V else logits_p[:, :V].contiguous().view(B, S, V)
        if labels is None:
            return logits
        lab = labels.reshape(-1).contiguous()
        loss_out, _, rows = ops.cross_entropy_fwd(logits_p, lab, float(label_smoothing), vocab=V, want_rows=True)
        if loss_weight is None:
            return logits, loss_out[0]
        return logits, ops.weighted_mean(rows, loss_weight.reshape(-1).float().contiguous())[0]   # :311-316

    def generate(self):
        raise AssertionError("generate() is not part of MaskGiTUViT_v2 (reference :327-328)")


MaskGiTUViT = MaskGiTUViT_v2
