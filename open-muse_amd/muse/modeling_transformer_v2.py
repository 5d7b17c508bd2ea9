"""muse.MaskGiTUViT_v2 / muse.MaskGiTUViT for MI355X — SURVEY.md §8 row a12 (config 4, reference
muse/modeling_transformer_v2.py:150-319).

STATUS (round 1): forward (logits, plain / label-smoothed / per-token-weighted loss) and the hand-written backward of the loss
(every parameter gradient), f32 ("parity mode": exact-f32 MFMA GEMMs, materialised attention = the reference's algorithm).  Same constructor kwargs (filtered like ``config_from_legacy_kwargs`` :127-147: unknown keys are
dropped), config keys, state_dict names / shapes and init as the reference.

Everything computes through libmuse_hip.so on channels-last rows ``[B * S, C]``:
  linears / 1x1 convs / attention GEMMs -> muse_gemm;  embedding -> muse_gather_rows;  RMSNorm / LayerNorm with the pre-norm
  residual stream -> muse_norm_res_fwd;  AdaLN -> muse_adaln_fwd;  depthwise 3x3 -> muse_dwconv3x3_nhwc;  GlobalResponseNorm ->
  muse_grn_fwd;  GELU / GLU / softmax / cross-entropy -> the MaskGit kernels;  micro-conditioning -> muse_sinusoidal_encode.
"""
from __future__ import annotations

import math
import os
from typing import Tuple

import numpy as np
import torch
from torch import nn

from . import ops
from ._hip import MuseHipError
from .modeling_utils import ConfigMixin, ModelMixin
from .sampling import cosine_schedule, decode_seed, scheduled_mask_len, step_noise
from .tape_ops import TapeOps, _BF16_OPERANDS

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


_NORM_AFFINE = [True]     # set by MaskGiTUViT_v2.__init__ while it builds its submodules (config.ln_elementwise_affine)


class _NormW(nn.Module):
    """Norm (reference :632-726).  ln_elementwise_affine=False: no learnable gain (`self.weight = None` in the reference, :656-660) - here a
    non-persistent buffer of ones, so that the state dict and named_parameters() match the reference's while the kernels keep one form"""

    def __init__(self, c):
        super().__init__()
        if _NORM_AFFINE[0]:
            self.weight = nn.Parameter(torch.ones(c))
        else:
            self.register_buffer("weight", torch.ones(c), persistent=False)


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


class _ConvT(nn.Module):
    """nn.ConvTranspose2d(cin, cout, k, stride=k) of the reference: weight [cin, cout, k, k]"""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cin, cout, k, k))


class _Block(nn.Module):
    """DownsampleBlock (reference :506-541) / UpsampleBlock (:544-583).  resample ("down" / "up" / None, config.force_down_up_sample):
    Sequential(Norm2D, Conv2d(c, c, 2, stride 2)) registered BEFORE the blocks, or Sequential(Norm2D, ConvTranspose2d(c, c, 2, stride 2))
    registered AFTER them - the reference's module order, so that state dicts and parameter lists line up"""

    def __init__(self, c, hidden, n, resample=None):
        super().__init__()
        if resample == "down":
            self.downsample = nn.ModuleDict({"0": _Norm2D(c), "1": _Conv(c, c, 2)})
        self.res_blocks = nn.ModuleList([_ResBlock(c, hidden) for _ in range(n)])
        self.attention_blocks = nn.ModuleList([_AttnBlock2D(c, hidden) for _ in range(n)])
        if resample == "up":
            self.upsample = nn.ModuleDict({"0": _Norm2D(c), "1": _ConvT(c, c, 2)})


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




class _UViTFn(torch.autograd.Function):
    """one autograd node for the whole network: forward records a tape, backward runs the hand-written reverse pass and hands
    the parameter gradients (state-dict order) back to autograd"""

    @staticmethod
    def forward(ctx, model, input_ids, enc, cond, micro, labels, label_smoothing, loss_weight, need_grad, *params):
        model._drop_step_caches()
        model.__dict__["_act_cache_on"] = bool(need_grad)
        try:
            with model._gemm_mode():
                logits, loss, tape = model._run_forward(input_ids, enc, cond, micro, labels, label_smoothing, loss_weight, need_grad)
        except BaseException:
            model._drop_step_caches()      # a forward that raised keeps no activation copies / operand images alive
            raise
        finally:
            model.__dict__["_act_cache_on"] = False
        ctx.model, ctx.tape = model, tape
        ctx.set_materialize_grads(False)
        if loss is None:
            return logits
        ctx.mark_non_differentiable(logits)
        return logits, loss

    @staticmethod
    def backward(ctx, g_logits, g_loss=None):
        if ctx.tape is None:
            raise MuseHipError("backward called on a forward that ran without grad")
        if g_loss is None:
            raise MuseHipError("MaskGiTUViT_v2: only the loss is differentiable (pass labels)")
        model = ctx.model
        model.__dict__["_dw_pending"] = []          # (a backward that raised may have left collected products behind)
        model.__dict__["_grads_reported"] = set()
        with model._gemm_mode(backward=True):
            G = model._run_backward(ctx.tape, g_loss)
        ctx.tape = None
        model._drop_step_caches()
        grads = tuple(G.get(name) for name, _ in model.named_parameters())
        return (None,) * 9 + grads


class MaskGiTUViT_v2(TapeOps, ModelMixin, ConfigMixin):
    _cast_selects_compute_mode = True     # model.half() / .to(dtype) / from_pretrained(torch_dtype=) pick the compute mode; masters stay f32 (ModelMixin)
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
        if c.use_bias or c.use_fused_mlp or c.norm_type not in ("rmsnorm", "layernorm"):
            raise NotImplementedError("MI355X build of MaskGiTUViT_v2: only the bias-free GLU family is built (norm_type rmsnorm or "
                                      "layernorm, with or without learnable gains, with or without the forced down/up-sampling)")
        self.__dict__["_default_norm_mode"] = 1 if c.norm_type == "layernorm" else 0     # (Norm, reference :632-641)
        if c.hidden_dropout != 0.0 or c.attention_dropout != 0.0:
            raise NotImplementedError("dropout > 0 is outside the MI355X hot-path build")
        H, C, cin = c.hidden_size, c.block_out_channels[0], c.in_channels
        for width, heads in ((C, c.block_num_heads), (H, c.num_attention_heads)):     # Attention.__init__ of the reference (:842-847), same text:
            if width % heads != 0:                                                     # configs/cc12m_uvit_clip.yaml as shipped (1024 channels, 12 heads)
                raise ValueError(f"self.hidden_size: {width} must be divisible by self.num_heads: {heads}")     # trips it
        self.output_size = c.codebook_size
        try:   # the submodules read the gain mode while they are built; whatever happens in between, the module-level default comes back
            _NORM_AFFINE[0] = bool(c.ln_elementwise_affine)
            self._build_submodules(c, H, C, cin)
        finally:
            _NORM_AFFINE[0] = True
        self._init_weights()

    def _build_submodules(self, c, H, C, cin):
        self.encoder_proj = _Lin(c.encoder_hidden_size, H)
        self.encoder_proj_layer_norm = _NormW(H)
        self.embed = _Embed(c.vocab_size, cin, C)
        self.cond_embed = nn.ModuleDict({"0": _Lin(c.micro_cond_embed_dim + c.cond_embed_dim, H), "2": _Lin(H, H)})
        self.down_blocks = nn.ModuleList([_Block(C, H, c.num_res_blocks, "down" if c.force_down_up_sample else None)])
        self.project_to_hidden_norm = _NormW(C)
        self.project_to_hidden = _Lin(C, H)
        self.transformer_layers = nn.ModuleList([_Layer(H, c.intermediate_size) for _ in range(c.num_hidden_layers)])
        self.project_from_hidden_norm = _NormW(H)
        self.project_from_hidden = _Lin(H, C)
        self.up_blocks = nn.ModuleList([_Block(C, H, c.num_res_blocks, "up" if c.force_down_up_sample else None)])
        self.mlm_layer = _Mlm(C, cin, c.codebook_size)
        self.compute_dtype = torch.float32
        self.wgrad_stream = os.environ.get("MUSE_WGRAD_STREAM", "1") != "0"   # bf16 mode: weight-gradient GEMMs on a second HIP stream
        self.fuse_norm_adaln = os.environ.get("MUSE_NORM_ADALN", "1") != "0"  # norm + AdaLN of a transformer layer as one kernel (fwd and bwd)
        self.batch_adaln_mappers = os.environ.get("MUSE_ADALN_BATCH", "1") != "0"   # every AdaLN mapper in a few batched products
        self._side_stream = None

    def _init_weights(self):
        """reference :205-237: trunc_normal(0.02) for Linear / Conv / Embedding, ones for norm gains, then the special cases"""
        for m in self.modules():
            if isinstance(m, (_Lin, _Conv)):
                nn.init.trunc_normal_(m.weight, std=0.02)
            elif isinstance(m, nn.Embedding):
                nn.init.trunc_normal_(m.weight, std=0.02)
            elif isinstance(m, _ConvT):       # not an nn.Conv2d: the reference's _init_weights (:225-231) leaves torch's default init
                nn.init.kaiming_uniform_(m.weight, a=math.sqrt(5))
        nn.init.xavier_uniform_(self.embed.conv.weight, 0.02)
        nn.init.normal_(self.embed.embeddings.weight, std=float(np.sqrt(1 / self.config.vocab_size)))
        nn.init.constant_(self.mlm_layer.conv1.weight, 0)
        self.mlm_layer.conv2.weight.data = self.embed.embeddings.weight.data[: self.config.codebook_size, :, None, None].clone()
        for m in self.modules():
            if isinstance(m, _AdaLN):
                nn.init.constant_(m.mapper.weight, 0)

    # ---- forward / backward ---------------------------------------------------------------------------------------------
    # Activations are channels-last rows [B * S, C].  Every helper returns (output, saved); the *_bwd twin consumes `saved`,
    # stores parameter gradients in `G` (name -> tensor) and returns the input gradients.  f32 throughout.
    # ---- every AdaLN mapper of the network in a few batched products -------------------------------------------------------
    # All AdaLNModulation layers read the SAME input, silu(cond) [B, cond_embed_dim] (:1032): 72 Linear layers of the config-4 model,
    # i.e. 72 forward, 72 dX and 72 dW products with M = batch rows - 16-workgroup launches of 30-50 us each (8.6 ms of a 162 ms step).
    # They are independent of everything else, so they run as one batched GEMM per weight shape: forward before the first block,
    # d(weight) and d(silu(cond)) after the last backward block has delivered its d(scale | shift).
    def _ada_sites(self):
        """[(state-dict prefix, _AdaLN module)] of every site, cached"""
        sites = self.__dict__.get("_ada_site_list")
        if sites is None:
            sites = []
            for bname, blocks in (("down_blocks.0", self.down_blocks[0]), ("up_blocks.0", self.up_blocks[0])):
                for i, rb in enumerate(blocks.res_blocks):
                    sites.append((f"{bname}.res_blocks.{i}.adaLN_modulation", rb.adaLN_modulation))
            for li, lyr in enumerate(self.transformer_layers):
                nm = f"transformer_layers.{li}"
                sites += [(nm + ".self_attn_adaLN_modulation", lyr.self_attn_adaLN_modulation),
                          (nm + ".cross_attn_adaLN_modulation", lyr.cross_attn_adaLN_modulation),
                          (nm + ".ffn.adaLN_modulation", lyr.ffn.adaLN_modulation)]
            self.__dict__["_ada_site_list"] = sites
        return sites

    def _ada_forward_all(self, scond, B, need_grad):
        """(scale | shift) of every AdaLN site: one batched product per mapper shape -> {id(module): ss [B, 2C]} (+ the groups for the
        backward)"""
        groups = {}
        for name, mod in self._ada_sites():
            groups.setdefault(tuple(mod.mapper.weight.shape), []).append((name, mod))
        ss_of, tape = {}, []
        xc_cache = {}
        for (N, K), members in groups.items():
            Z = len(members)
            w2 = self._w2(*[m.mapper for _, m in members])                    # [Z * N, K], the compute dtype (cached stacking)
            xc = xc_cache.get(w2.dtype)
            if xc is None:
                xc = xc_cache[w2.dtype] = self._pair(scond, w2)[0]
            ss = torch.empty((Z, B, N), dtype=torch.float32, device=scond.device)
            ops.gemm(xc, w2, ss, B, N, K, la=0, lb=0, lda=K, ldb=K, ldc=N, batch=Z, sA=(0, 0), sB=(N * K, 0), sC=(B * N, 0))
            dss = torch.empty((Z, B, N), dtype=torch.float32, device=scond.device) if need_grad else None
            for z, (name, mod) in enumerate(members):
                ss_of[id(mod)] = (ss[z], dss[z] if need_grad else None)
            tape.append((members, w2, dss, N, K))
        self.__dict__["_ada_ss"] = ss_of
        return tape

    def _ada_backward_all(self, tape, scond, dscond, B, G):
        """d(mapper weights) and d(silu(cond)) of every AdaLN site from the d(scale | shift) the blocks left in the group buffers"""
        for members, w2, dss, N, K in tape:
            Z = len(members)
            if N % 8 or K % 8:      # rows of the k-major bf16 operands would not be 16-byte chunks (_mm_dw's guard): f32 operands
                w2 = w2 if w2.dtype == torch.float32 else ops.cast_to_f32(w2.contiguous())
            dsc, xc = self._pair(dss.view(Z * B, N), w2)[0].view(Z, B, N), self._pair(scond, w2)[0]
            gw = torch.empty((Z * N, K), dtype=torch.float32, device=scond.device)
            # dW_z = dss_z^T scond   (both operands k-major, k = batch rows)
            ops.gemm(dsc, xc, gw, N, K, B, la=1, lb=1, lda=N, ldb=K, ldc=K, batch=Z, sA=(B * N, 0), sB=(0, 0), sC=(N * K, 0))
            for z, (name, mod) in enumerate(members):
                G[name + ".mapper.weight"] = gw[z * N:(z + 1) * N].view(mod.mapper.weight.shape)
            # d(scond) += sum_z dss_z W_z
            part = torch.empty((Z, B * K), dtype=torch.float32, device=scond.device)
            ops.gemm(dsc, w2, part, B, K, N, la=0, lb=1, lda=N, ldb=K, ldc=K, batch=Z, sA=(B * N, 0), sB=(N * K, 0), sC=(B * K, 0))
            ops.colsum(part, dscond.view(-1), accumulate=True)

    def _ada_ss_of(self, mod, scond):
        """(ss, dss slot or None) of a site: from the batched products when they ran, else the site's own Linear"""
        hit = self.__dict__.get("_ada_ss", {}).get(id(mod))
        if hit is not None:
            return hit
        return self._lin(scond, mod.mapper), None

    def _adaln(self, x, mod: _AdaLN, scond, B, gemm_operand=False):
        """gemm_operand (bf16 mode): the modulated tensor is consumed only as a GEMM operand -> written as bf16 directly"""
        ss, slot = self._ada_ss_of(mod, scond)
        od = torch.bfloat16 if (gemm_operand and self.compute_dtype == torch.bfloat16 and _BF16_OPERANDS & 1) else torch.float32
        return ops.adaln_fwd(x, ss, B, out_dtype=od), dict(x=x, ss=ss, slot=slot)

    def _ada_dss(self, dss, sv, mod, name, G, scond, dscond):
        """hand a site's d(scale | shift) on: into its slot of the batched backward, or through the site's own Linear backward"""
        if sv.get("slot") is not None:
            if dss.data_ptr() != sv["slot"].data_ptr():
                sv["slot"].copy_(dss)
            return
        d = self._lin_bwd(dss, scond, mod.mapper, name + ".mapper", G)
        dscond.add_(d)            # every AdaLN reads the same silu(cond): sum of a [B, H] tensor (plumbing)

    def _adaln_bwd(self, dy, sv, mod: _AdaLN, name, G, scond, dscond, B):
        dx, dss = ops.adaln_bwd(dy, sv["x"], sv["ss"], B, dss_out=sv.get("slot"))
        self._ada_dss(dss, sv, mod, name, G, scond, dscond)
        return dx

    def _norm_adaln(self, x, norm_mod, ada: _AdaLN, scond, B, mode=None, residual=None, gemm_only=False):
        """norm(x + residual) followed by its AdaLN modulation (TransformerLayer :757-792) as ONE kernel where the shape allows: the
        norm output itself is never written.  -> (m = GEMM operand of the next block, v = x + residual, tape entry)"""
        mode = self._nm(mode)
        ss, slot = self._ada_ss_of(ada, scond)
        if self.fuse_norm_adaln and ops.norm_adaln_ok(x.shape[0], x.shape[1], B):
            od = torch.bfloat16 if (self.compute_dtype == torch.bfloat16 and _BF16_OPERANDS & 1) else torch.float32
            # gemm_only: the caller's m is read by weight GEMMs alone (all of them products the four-plane kernel takes) - in a bf16x3 step
            # it exists as their operand planes, the f32 tensor is not written
            po = gemm_only and od == torch.float32 and x.dtype == torch.float32 and ops.planes_only_ok(x.shape[0], x.shape[1])
            m, v = ops.norm_adaln_fwd(x, self._f(norm_mod.weight), ss, B, float(self.config.layer_norm_eps), mode, residual=residual,
                                      out_dtype=od, planes_only=po)
            return m, v, dict(ss=ss, slot=slot, fused=True)
        n, v = self._norm(x, norm_mod, mode=mode, residual=residual, want_pre=True)
        od = torch.bfloat16 if (self.compute_dtype == torch.bfloat16 and _BF16_OPERANDS & 1) else torch.float32
        return ops.adaln_fwd(n, ss, B, out_dtype=od), v, dict(x=n, ss=ss, slot=slot)

    def _norm_adaln_bwd(self, dm, sv, v, norm_mod, norm_name, ada: _AdaLN, ada_name, G, scond, dscond, B, mode=None, dpre=None):
        """-> d(x) = d(residual) of _norm_adaln; the bf16 copy of it (dY of the next weight GEMMs) rides in the activation cache"""
        mode = self._nm(mode)
        if not sv.get("fused"):
            dn = self._adaln_bwd(dm, sv, ada, ada_name, G, scond, dscond, B)
            return self._norm_bwd(dn, v, norm_mod, norm_name, G, mode=mode, dpre=dpre, gemm_operand=True)
        eps = float(self.config.layer_norm_eps)
        if self.compute_dtype == torch.bfloat16 and _BF16_OPERANDS & 2:
            dv, dw, dss, dvb = ops.norm_adaln_bwd(dm, v, self._f(norm_mod.weight), sv["ss"], B, eps, mode, dpre=dpre, also_bf16=True,
                                                  dss_out=sv.get("slot"))
            self.__dict__.setdefault("_act_cache", {})[id(dv)] = (dv, dvb)
        else:
            dv, dw, dss = ops.norm_adaln_bwd(dm, v, self._f(norm_mod.weight), sv["ss"], B, eps, mode, dpre=dpre, dss_out=sv.get("slot"))
        if isinstance(norm_mod.weight, nn.Parameter):        # (ln_elementwise_affine=False: the gain is a constant, not a parameter)
            G[norm_name + ".weight"] = dw
        self._ada_dss(dss, sv, ada, ada_name, G, scond, dscond)
        return dv

    def _res_block(self, h, blk: _ResBlock, scond, B, side):
        C = h.shape[1]
        wdw = self._f(blk.depthwise.weight).contiguous()
        d = ops.dwconv3x3_nhwc(h, wdw, B, side, side, C)
        n, _ = self._norm(d, blk.norm.norm)
        a = self._lin(n, blk.channelwise["0"])
        ga = ops.gelu_fwd(a)
        grn = blk.channelwise["2"]
        gamma = self._f(grn.gamma).reshape(-1).contiguous()
        # (g is only ever a GEMM operand - forward product and the dW product of the backward: bf16 mode gets it as bf16 from the kernel)
        g, stats = ops.grn_fwd(ga, gamma, self._f(grn.beta).reshape(-1).contiguous(), B, side * side, want_stats=True,
                               out_dtype=self.compute_dtype)
        x = self._lin(g, blk.channelwise["4"], residual=h)                          # + x_res (:616)
        y, sva = self._adaln(x, blk.adaLN_modulation, scond, B)
        return y, dict(h=h, d=d, n=n, a=a, ga=ga, g=g, stats=stats, gamma=gamma, wdw=wdw, ada=sva, side=side)

    def _res_block_bwd(self, dy, sv, blk: _ResBlock, name, G, scond, dscond, B):
        C = sv["h"].shape[1]
        side = sv["side"]
        dx = self._adaln_bwd(dy, sv["ada"], blk.adaLN_modulation, name + ".adaLN_modulation", G, scond, dscond, B)
        dg = self._lin_bwd(dx, sv["g"], blk.channelwise["4"], name + ".channelwise.4", G)      # dx also flows to h (residual)
        dga, dgam, dbet = ops.grn_bwd(dg, sv["ga"], sv["gamma"], sv["stats"], B, side * side)
        grn = blk.channelwise["2"]
        G[name + ".channelwise.2.gamma"] = dgam.view(grn.gamma.shape)
        G[name + ".channelwise.2.beta"] = dbet.view(grn.beta.shape)
        da = ops.gelu_bwd(sv["a"], dga, out_dtype=self.compute_dtype)               # (only the dY of channelwise.0's dW / dX products)
        dn = self._lin_bwd(da, sv["n"], blk.channelwise["0"], name + ".channelwise.0", G)
        dd = self._norm_bwd(dn, sv["d"], blk.norm.norm, name + ".norm.norm", G)
        dh, dwdw = ops.dwconv3x3_bwd(dd, sv["h"], sv["wdw"], B, side, side, C)
        G[name + ".depthwise.weight"] = dwdw
        return dh.add_(dx)                                                           # + residual path

    def _attn_block(self, h, blk: _AttnBlock2D, enc, senc, B, S, L):
        has_map = hasattr(blk, "kv_mapper")
        ctx = self._lin(senc, blk.kv_mapper) if has_map else enc                      # :815-816
        nh = self.config.block_num_heads
        n1, _ = self._norm(h, blk.attn_layer_norm)                                    # residual = h (:819)
        a1, s1 = self._attention(n1, ctx, blk.attention, B, S, L, nh)
        n2, res = self._norm(a1, blk.crossattn_layer_norm, residual=h, want_pre=True)  # :822
        y, s2 = self._attention(n2, ctx, blk.crossattention, B, S, L, nh, residual=res)  # + residual (:824)
        return y, dict(h=h, res=res, s1=s1, s2=s2, has_map=has_map)

    def _attn_block_bwd(self, dy, sv, blk: _AttnBlock2D, name, G, senc, denc, dsenc):
        """accumulates the context gradient into denc (no kv_mapper) or dsenc (through kv_mapper); returns dh"""
        dn2, dctx2 = self._attention_bwd(dy, sv["s2"], blk.crossattention, name + ".crossattention", G)
        dv2 = self._norm_bwd(dn2, sv["res"], blk.crossattn_layer_norm, name + ".crossattn_layer_norm", G, dpre=dy)   # d(a1) = d(h)
        dn1, dctx1 = self._attention_bwd(dv2, sv["s1"], blk.attention, name + ".attention", G)
        dh = self._norm_bwd(dn1, sv["h"], blk.attn_layer_norm, name + ".attn_layer_norm", G, dpre=dv2)
        dctx = dctx1.add_(dctx2)
        if sv["has_map"]:
            dsenc.add_(self._lin_bwd(dctx, senc, blk.kv_mapper, name + ".kv_mapper", G))
        else:
            denc.add_(dctx)
        return dh

    # ---- force_down_up_sample (configs/research_run_512_with_downsample*.yaml): 2x2 stride-2 conv before the down block, 2x2 stride-2
    # transposed conv after the up block (reference :510-514, :558-562).  Both are ONE product on the GEMM kernels around a
    # space-to-depth move (ops.space_to_depth2 / depth_to_space2): conv  y[p, co] = sum_(di,dj,ci) x2[p, (di,dj,ci)] W[co, ci, di, dj],
    # transposed conv  y2[p, (di,dj,co)] = sum_ci x[p, ci] W[ci, co, di, dj] followed by the depth-to-space placement.
    def _w_resample(self, mod, transposed):
        """the 2x2 kernel as the [N_out, K_in] operand of the product: [C, (di,dj,ci)] / [(di,dj,co), C]; compute dtype (the
        re-layout of a C x C x 2 x 2 tensor is a device copy: plumbing)"""
        w = self._f(mod.weight)
        w2 = (w.permute(2, 3, 1, 0) if transposed else w.permute(0, 2, 3, 1)).reshape((-1, w.shape[0]) if transposed else (w.shape[0], -1))
        w2 = w2.contiguous()
        if self.compute_dtype == torch.bfloat16 and w2.shape[1] % 8 == 0 and w2.shape[0] % 8 == 0:
            w2 = ops.cast_to_bf16(w2)
        return w2

    def _mm_dw_now(self, dy, x, shape2):
        """dy^T x -> f32 [N, K] on the current stream (the two resampling kernels need their gradient re-laid out right away, so they
        stay out of the grouped side-stream launch)"""
        if shape2[-1] % 8 or shape2[0] % 8 or self.compute_dtype != torch.bfloat16:
            dyc = dy if dy.dtype == torch.float32 else ops.cast_to_f32(dy.contiguous())
            xc = x if x.dtype == torch.float32 else ops.cast_to_f32(x.contiguous())
        else:
            dyc, xc = self._c(dy), self._c(x)
        dw = torch.empty(shape2, dtype=torch.float32, device=x.device)
        ops.linear_wgrad(dyc, xc, dw, False)
        return dw

    def _run_forward(self, input_ids, encoder_hidden_states, cond_embeds, micro_conds, labels, label_smoothing, loss_weight,
                     need_grad):
        c = self.config
        B, St = input_ids.shape                                                       # St tokens on a side_t x side_t grid
        side_t = int(St ** 0.5)
        if side_t * side_t != St:
            raise ValueError("the token sequence must be a square grid")
        down_up = bool(c.force_down_up_sample)
        if down_up and side_t % 2:
            raise ValueError("force_down_up_sample needs an even token grid")
        S, side = (St // 4, side_t // 2) if down_up else (St, side_t)                 # what the blocks and the transformer layers see
        L = encoder_hidden_states.shape[1]
        H, C = c.hidden_size, c.block_out_channels[0]
        f = self._f
        T = {}                                                                        # the tape
        # text states :252-253
        enc_in = encoder_hidden_states.reshape(B * L, -1).float().contiguous()
        enc0 = self._lin(enc_in, self.encoder_proj)
        enc, _ = self._norm(enc0, self.encoder_proj_layer_norm)
        senc = ops.silu_fwd(enc) if C != H else None
        # conditioning :255-260
        micro = ops.sinusoidal_encode(micro_conds, c.micro_cond_encode_dim).reshape(B, -1)
        cond_in = torch.cat([cond_embeds.float(), micro], dim=1).contiguous()
        c1 = self._lin(cond_in, self.cond_embed["0"])
        sc1 = ops.silu_fwd(c1)
        cond = self._lin(sc1, self.cond_embed["2"])
        scond = ops.silu_fwd(cond)                                                    # every AdaLN sees silu(cond) (:1032)
        self.__dict__["_ada_ss"] = {}
        ada_tape = self._ada_forward_all(scond, B, need_grad) if self.batch_adaln_mappers else None
        # ConvEmbed :485-500
        ids = input_ids.reshape(-1).contiguous()
        emb0 = ops.gather_rows(f(self.embed.embeddings.weight), ids, torch.float32)
        emb, _ = self._norm(emb0, self.embed.layer_norm)
        h = self._lin(emb, self.embed.conv)
        T["down"] = []
        blk = self.down_blocks[0]
        if down_up:                                                                   # Norm2D -> Conv2d(C, C, 2, stride 2)  (:510-514)
            n0, _ = self._norm(h, blk.downsample["0"].norm)
            x2 = ops.space_to_depth2(n0, B, side_t, side_t, C)
            wd = self._w_resample(blk.downsample["1"], False)
            T["downsample"] = dict(h=h, x2=x2, wd=wd)
            h = self._mm(x2, wd)
        for i in range(c.num_res_blocks):
            h, sr = self._res_block(h, blk.res_blocks[i], scond, B, side)
            h, sa = self._attn_block(h, blk.attention_blocks[i], enc, senc, B, S, L)
            T["down"].append((sr, sa))
        hd_in = h
        n, _ = self._norm(h, self.project_to_hidden_norm)
        t = self._lin(n, self.project_to_hidden)
        T["proj_in"] = dict(h=hd_in, n=n)
        res = None
        nh = c.num_attention_heads
        T["layers"] = []
        # m1 / m2 / m3 are read by the layer's weight GEMMs only (q | k | v, q, wi_0 | wi_1: forward and dW), whatever attention route runs;
        # without biases and with every such weight at least 128 wide they need no f32 tensor in a bf16x3 step
        go = (not self.__dict__.get("_use_bias", False) and c.hidden_size >= 128 and 2 * c.intermediate_size >= 128)
        for lyr in self.transformer_layers:                                           # TransformerLayer :757-792
            m1, res1, a1s = self._norm_adaln(t, lyr.attn_layer_norm, lyr.self_attn_adaLN_modulation, scond, B, residual=res, gemm_only=go)
            a, s1 = self._attention(m1, m1, lyr.attention, B, S, S, nh)
            m2, res2, a2s = self._norm_adaln(a, lyr.crossattn_layer_norm, lyr.cross_attn_adaLN_modulation, scond, B, residual=res1, gemm_only=go)
            a2, s2 = self._attention(m2, enc, lyr.crossattention, B, S, L, nh)
            m3, res3, a3s = self._norm_adaln(a2, lyr.ffn.pre_mlp_layer_norm, lyr.ffn.adaLN_modulation, scond, B, mode=1,
                                             residual=res2, gemm_only=go)                                  # LayerNorm (:928)
            w01 = self._w2(lyr.ffn.wi_0, lyr.ffn.wi_1, rows=m3.shape[0])
            if self.compute_dtype == torch.bfloat16:
                # the reference's autocast regime: the GLU input and output live in bf16 between the two GEMMs (no f32 round trip,
                # no separate cast of the wo operand)
                ab = ops.linear(self._c(m3), w01, out_dtype=torch.bfloat16)
            else:
                ab = self._mm(m3, w01)
            # (bf16x3 step: gl only ever feeds wo's forward and dW products - it exists as their operand planes, no f32 tensor)
            po = (self.compute_dtype == torch.float32 and not self.__dict__.get("_use_bias", False) and ab.dtype == torch.float32
                  and ops.planes_only_ok(ab.shape[0], ab.shape[1] // 2) and min(lyr.ffn.wo.weight.shape) >= 128 and min(w01.shape) >= 128)
            gl = ops.glu_fwd(ab, planes_only=po)
            t = self._lin(gl, lyr.ffn.wo)
            T["layers"].append(dict(res1=res1, res2=res2, res3=res3, a1s=a1s, a2s=a2s, a3s=a3s, s1=s1, s2=s2, m3=m3, w01=w01, ab=ab,
                                    gl=gl))
            res = res3
        vlast = None
        n, vlast = self._norm(t, self.project_from_hidden_norm, residual=res, want_pre=True)   # (t + residual) then norm (:288-290)
        h = self._lin(n, self.project_from_hidden)
        T["proj_out"] = dict(v=vlast, n=n)
        T["up"] = []
        blk = self.up_blocks[0]
        for i in range(c.num_res_blocks):
            h, sr = self._res_block(h, blk.res_blocks[i], scond, B, side)
            h, sa = self._attn_block(h, blk.attention_blocks[i], enc, senc, B, S, L)
            T["up"].append((sr, sa))
        self.__dict__["_ada_ss"] = {}     # every site has run: the tape holds what backward needs, nothing pins the [Z, B, N] buffers
        if down_up:                                                                   # Norm2D -> ConvTranspose2d(C, C, 2, stride 2)  (:558-562)
            n1, _ = self._norm(h, blk.upsample["0"].norm)
            wu = self._w_resample(blk.upsample["1"], True)
            T["upsample"] = dict(h=h, n=n1, wu=wu)
            h = ops.depth_to_space2(self._mm(n1, wu), B, side_t, side_t, C)
        # ConvMlmLayer :1002-1022
        y1 = self._lin(h, self.mlm_layer.conv1)
        y2, _ = self._norm(y1, self.mlm_layer.layer_norm.norm)
        V = c.codebook_size
        Vp = (V + 7) // 8 * 8
        w2 = self._w2(self.mlm_layer.conv2)
        logits_p = torch.empty((B * St, Vp), dtype=torch.float32, device=y2.device)
        ops.gemm(self._c(y2), w2, logits_p, B * St, V, c.in_channels, lda=c.in_channels, ldb=c.in_channels, ldc=Vp)
        logits = logits_p.view(B, St, Vp) if Vp == V else logits_p[:, :V].contiguous().view(B, St, V)
        loss = None
        if labels is not None:
            lab = labels.reshape(-1).contiguous()
            self.__dict__["_loss_rows"] = lab.numel()        # ("f16" mode: bounds d(logits), tape_ops.f16_grad_scale_for)
            loss_out, lse, rows = ops.cross_entropy_fwd(logits_p, lab, float(label_smoothing), vocab=V, want_rows=True)
            lw = None
            if loss_weight is None:
                loss = loss_out[0]
            else:
                lw = loss_weight.reshape(-1).float().contiguous()
                loss = ops.weighted_mean(rows, lw)[0]                                 # :311-316
            T["ce"] = dict(lab=lab, lse=lse, loss_out=loss_out, lw=lw, ls=float(label_smoothing))
        if not need_grad:
            return logits, loss, None
        T.update(B=B, S=S, St=St, side_t=side_t, L=L, side=side, enc_in=enc_in, enc0=enc0, enc=enc, senc=senc, cond_in=cond_in, c1=c1, sc1=sc1, cond=cond,
                 scond=scond, ids=ids, emb0=emb0, emb=emb, h_mlm=h, y1=y1, y2=y2, logits_p=logits_p, V=V, Vp=Vp, ada_tape=ada_tape)
        return logits, loss, T

    def _run_backward(self, T, g_loss):
        """gradients of every parameter for d(loss) = g_loss: {state-dict name: tensor}"""
        c = self.config
        B, S, L, V, Vp = T["B"], T["S"], T["L"], T["V"], T["Vp"]
        St, side_t = T["St"], T["side_t"]
        H, C = c.hidden_size, c.block_out_channels[0]
        G = {}
        ce = T["ce"]
        dev = T["logits_p"].device
        go = g_loss.reshape(1).to(torch.float32).contiguous()
        dl = ops.cross_entropy_bwd(T["logits_p"], ce["lab"], ce["lse"], ce["loss_out"], go, ce["ls"], torch.float32, vocab=V)
        if ce["lw"] is not None:   # mean over valid rows -> weighted mean: row r scaled by w_r * n_valid / sum(w)
            ops.scale_rows_(dl, ce["lw"], ce["loss_out"][1:2], ce["lw"].sum().reshape(1), V)
        # ConvMlmLayer
        w2 = self._w2(self.mlm_layer.conv2)
        dlc = self._c(dl)
        G["mlm_layer.conv2.weight"] = self._mm_dw(dlc, T["y2"], w2.shape, M=V, lda=Vp).view(self.mlm_layer.conv2.weight.shape)
        dy2 = self._mm_dx(dlc, w2, lda=Vp)
        dy1 = self._norm_bwd(dy2, T["y1"], self.mlm_layer.layer_norm.norm, "mlm_layer.layer_norm.norm", G)
        dh = self._lin_bwd(dy1, T["h_mlm"], self.mlm_layer.conv1, "mlm_layer.conv1", G)
        scond, senc = T["scond"], T["senc"]
        dscond = torch.zeros_like(scond)
        denc = torch.zeros_like(T["enc"])
        dsenc = torch.zeros_like(senc) if senc is not None else None
        blk = self.up_blocks[0]
        if "upsample" in T:     # y = depth_to_space(n1 Wu^T): d(n1 Wu^T) = space_to_depth(dy)
            u = T["upsample"]
            dy2 = ops.space_to_depth2(dh, B, side_t, side_t, C)
            dyc = dy2 if u["wu"].dtype == torch.float32 else self._c(dy2)
            gw = self._mm_dw_now(dyc, u["n"], (4 * C, C))                              # [(di, dj, co), ci] -> [ci, co, di, dj]
            G["up_blocks.0.upsample.1.weight"] = gw.view(2, 2, C, C).permute(3, 2, 0, 1).contiguous()
            dh = self._norm_bwd(self._mm_dx(dyc, u["wu"]), u["h"], blk.upsample["0"].norm, "up_blocks.0.upsample.0.norm", G)
        for i in reversed(range(c.num_res_blocks)):
            sr, sa = T["up"][i]
            dh = self._attn_block_bwd(dh, sa, blk.attention_blocks[i], f"up_blocks.0.attention_blocks.{i}", G, senc, denc, dsenc)
            dh = self._res_block_bwd(dh, sr, blk.res_blocks[i], f"up_blocks.0.res_blocks.{i}", G, scond, dscond, B)
            self._report_grads(G)        # (data-parallel: finished gradients go to the reducer's buckets while backward continues)
        po = T["proj_out"]
        dn = self._lin_bwd(dh, po["n"], self.project_from_hidden, "project_from_hidden", G)
        dres = self._norm_bwd(dn, po["v"], self.project_from_hidden_norm, "project_from_hidden_norm", G, gemm_operand=True)   # = dt = d(residual)
        dt = dres
        for li in reversed(range(c.num_hidden_layers)):
            lyr, sv = self.transformer_layers[li], T["layers"][li]
            nm = f"transformer_layers.{li}"
            # feed-forward
            if self.compute_dtype == torch.bfloat16:   # bf16 GLU gradient chain (mirrors the forward)
                wo2, dtb = self._w2(lyr.ffn.wo), self._c(dt)
                G[nm + ".ffn.wo.weight"] = self._mm_dw(dtb, sv["gl"], wo2.shape).view(lyr.ffn.wo.weight.shape)
                dgl = ops.linear_dgrad(dtb, wo2)
            else:
                dgl = self._lin_bwd(dt, sv["gl"], lyr.ffn.wo, nm + ".ffn.wo", G)
            dab = ops.glu_bwd(sv["ab"], dgl, planes_only=isinstance(sv["gl"], ops.Planes) and ops.planes_only_ok(*sv["ab"].shape))
            gw01 = self._mm_dw(dab, sv["m3"], sv["w01"].shape)
            I = gw01.shape[0] // 2
            G[nm + ".ffn.wi_0.weight"], G[nm + ".ffn.wi_1.weight"] = gw01[:I], gw01[I:]
            dm3 = self._mm_dx(dab, sv["w01"])
            dv3 = self._norm_adaln_bwd(dm3, sv["a3s"], sv["res3"], lyr.ffn.pre_mlp_layer_norm, nm + ".ffn.pre_mlp_layer_norm",
                                       lyr.ffn.adaLN_modulation, nm + ".ffn.adaLN_modulation", G, scond, dscond, B, mode=1, dpre=dres)
            # cross attention (dv3 = d(a2) = d(res2))
            dm2, dctx = self._attention_bwd(dv3, sv["s2"], lyr.crossattention, nm + ".crossattention", G)
            denc.add_(dctx)
            dv2 = self._norm_adaln_bwd(dm2, sv["a2s"], sv["res2"], lyr.crossattn_layer_norm, nm + ".crossattn_layer_norm",
                                       lyr.cross_attn_adaLN_modulation, nm + ".cross_attn_adaLN_modulation", G, scond, dscond, B, dpre=dv3)
            # self attention (dv2 = d(a) = d(res1))
            dm1, _ = self._attention_bwd(dv2, sv["s1"], lyr.attention, nm + ".attention", G, self_attn=True)
            dv1 = self._norm_adaln_bwd(dm1, sv["a1s"], sv["res1"], lyr.attn_layer_norm, nm + ".attn_layer_norm",
                                       lyr.self_attn_adaLN_modulation, nm + ".self_attn_adaLN_modulation", G, scond, dscond, B, dpre=dv2)
            dt = dres = dv1                                                           # d(t_prev) = d(res_prev)
            self._report_grads(G)
        pi = T["proj_in"]
        dn = self._lin_bwd(dt, pi["n"], self.project_to_hidden, "project_to_hidden", G)
        dh = self._norm_bwd(dn, pi["h"], self.project_to_hidden_norm, "project_to_hidden_norm", G)
        blk = self.down_blocks[0]
        for i in reversed(range(c.num_res_blocks)):
            sr, sa = T["down"][i]
            dh = self._attn_block_bwd(dh, sa, blk.attention_blocks[i], f"down_blocks.0.attention_blocks.{i}", G, senc, denc, dsenc)
            dh = self._res_block_bwd(dh, sr, blk.res_blocks[i], f"down_blocks.0.res_blocks.{i}", G, scond, dscond, B)
            self._report_grads(G)
        if "downsample" in T:   # y = x2 Wd^T, x2 = space_to_depth(norm(h))
            d = T["downsample"]
            dyc = dh if d["wd"].dtype == torch.float32 else self._c(dh)
            gw = self._mm_dw_now(dyc, d["x2"], (C, 4 * C))                             # [co, (di, dj, ci)] -> [co, ci, di, dj]
            G["down_blocks.0.downsample.1.weight"] = gw.view(C, 2, 2, C).permute(0, 3, 1, 2).contiguous()
            dn0 = ops.depth_to_space2(self._mm_dx(dyc, d["wd"]), B, side_t, side_t, C)
            dh = self._norm_bwd(dn0, d["h"], blk.downsample["0"].norm, "down_blocks.0.downsample.0.norm", G)
        # ConvEmbed
        demb = self._lin_bwd(dh, T["emb"], self.embed.conv, "embed.conv", G)
        demb0 = self._norm_bwd(demb, T["emb0"], self.embed.layer_norm, "embed.layer_norm", G)
        gemb = torch.zeros_like(self._f(self.embed.embeddings.weight))
        dpos = torch.empty((St, gemb.shape[1]), dtype=torch.float32, device=dev)     # (position table of the shared kernel: unused here)
        ops.embed_bwd(T["ids"].view(B, St), demb0, gemb, dpos, False)
        G["embed.embeddings.weight"] = gemb
        # conditioning: scond = silu(cond), cond = W2 silu(W0 cond_in)
        if T.get("ada_tape") is not None:
            self._ada_backward_all(T["ada_tape"], scond, dscond, B, G)
        dcond = ops.silu_bwd(T["cond"], dscond)
        dsc1 = self._lin_bwd(dcond, T["sc1"], self.cond_embed["2"], "cond_embed.2", G)
        dc1 = ops.silu_bwd(T["c1"], dsc1)
        self._lin_bwd(dc1, T["cond_in"], self.cond_embed["0"], "cond_embed.0", G, need_dx=False)
        # text states
        if dsenc is not None:
            denc.add_(ops.silu_bwd(T["enc"], dsenc))
        denc0 = self._norm_bwd(denc, T["enc0"], self.encoder_proj_layer_norm, "encoder_proj_layer_norm", G)
        self._lin_bwd(denc0, T["enc_in"], self.encoder_proj, "encoder_proj", G, need_dx=False)
        self._report_grads(G, final=True)
        if self.__dict__.pop("_side_busy", False):
            torch.cuda.current_stream(dev).wait_stream(self._side_stream)   # every weight gradient is complete before autograd sees it
        return G

    def forward(self, input_ids, encoder_hidden_states, cond_embeds, micro_conds, labels=None, label_smoothing=0.0,
                loss_weight=None):
        for t in (input_ids, encoder_hidden_states, cond_embeds, micro_conds):
            if not t.is_cuda:
                raise MuseHipError("MaskGiTUViT_v2 (MI355X build) has no CPU path: move the model and inputs to the GPU")
        params = [p for _, p in self.named_parameters()]
        need_grad = torch.is_grad_enabled() and labels is not None and any(p.requires_grad for p in params)
        out = _UViTFn.apply(self, input_ids, encoder_hidden_states, cond_embeds, micro_conds, labels, label_smoothing, loss_weight,
                            need_grad, *params)
        return out

    def generate(self):
        raise AssertionError("generate() is not part of MaskGiTUViT_v2 (reference :327-328)")

    @torch.no_grad()
    def generate2(self, encoder_hidden_states, cond_embeds, micro_conds, empty_embeds, empty_cond_embeds, input_ids=None,
                  negative_embeds=None, negative_cond_embeds=None, temperature=1.0, timesteps=18, guidance_scale=0,
                  guidance_schedule=None, noise_schedule=cosine_schedule, generator=None, return_intermediate=False,
                  seq_len=None, use_tqdm=None, topk_filter_thres=None, noise_type=None, predict_all_tokens=None, noise=None,
                  hip_graph=False):
        """reference :330-479 — iterative parallel decoding with classifier-free guidance.  Per step: one forward on the HIP
        path (batch doubled under guidance) and ONE device call for the guidance mix + everything token-level
        (muse_sample_step).  `noise` (tests): per step (exponential draws [B*S, codebook], uniform draws [B, S]) replacing the
        in-kernel Philox stream.  (The reference leaves `model_input` undefined for guidance_scale == 0; here that case feeds
        input_ids.)  hip_graph=True captures the forward once on a fixed input buffer and replays it every step."""
        B = encoder_hidden_states.shape[0]
        S = 256 if seq_len is None else seq_len
        dev = encoder_hidden_states.device
        mask_id, V = self.config.mask_token_id, self.config.codebook_size
        # host-side per-step scalars, float32 like the reference's 0-dim tensors
        temperatures = (torch.linspace(temperature[0], temperature[1], timesteps) if isinstance(temperature, tuple)
                        else torch.linspace(temperature, 0.01, timesteps))
        if guidance_schedule == "linear":
            scales = torch.linspace(0, guidance_scale, timesteps)
        elif guidance_schedule == "cosine":
            scales = torch.tensor([float((cosine_schedule(torch.tensor(1 - 1.0 * (i + 1) / timesteps)) * guidance_scale).floor())
                                   for i in range(timesteps)])
        else:
            scales = torch.ones(timesteps) * guidance_scale
        if input_ids is None:
            input_ids = torch.full((B, S), mask_id, dtype=torch.long, device=dev)
        if micro_conds.shape[0] == 1:
            micro_conds = micro_conds.repeat(B, 1).to(dev)
        guided = guidance_scale > 0
        if guided:
            def pair(x, y):   # conditional batch followed by the unconditional one (a single broadcastable entry is expanded)
                return torch.cat([x, y.expand(B, *y.shape[1:]) if y.shape[0] == 1 else y])
            encoder_hidden_states = pair(encoder_hidden_states, empty_embeds if negative_embeds is None else negative_embeds)
            cond_embeds = pair(cond_embeds, empty_cond_embeds if negative_cond_embeds is None else negative_cond_embeds)
            micro_conds = torch.cat([micro_conds, micro_conds], dim=0)
        seed = decode_seed(generator) if noise is None else 0
        order = range(timesteps)
        if use_tqdm:
            from tqdm.auto import tqdm
            order = tqdm(order)
        intermediate = []
        sampled = input_ids
        graph = None
        if hip_graph:
            # The captured forward is KEPT across calls (a serving loop decodes batch after batch of one shape): it reads its inputs from
            # static buffers, so a later call with the same shapes copies its conditioning in and replays - no capture, no per-launch
            # host work (a 512-row forward is ~530 launches at ~16 us of host time each against ~7-8 ms of kernels).  The key carries
            # what the captured kernels read by address: shapes, compute dtype, and a weight generation (autograd version counters + the
            # explicit invalidations; muse.FusedAdamW updates masters and bf16 copies in place, which a replay sees).
            wgen = (sum(p._version for p in self.parameters()), self.__dict__.get("_wgen", 0), str(self.compute_dtype), self.training)
            key = (B, S, guided, tuple(encoder_hidden_states.shape), tuple(cond_embeds.shape), tuple(micro_conds.shape), str(dev), wgen)
            hit = self.__dict__.get("_gen_graph")
            if hit is not None and hit["key"] == key:
                model_in, enc_b, cond_b, micro_b, graph, out = hit["bufs"]
                enc_b.copy_(encoder_hidden_states)
                cond_b.copy_(cond_embeds)
                micro_b.copy_(micro_conds)
            else:
                self.__dict__["_gen_graph"] = None            # (free the previous graph's pool before capturing the next one)
                model_in = torch.cat([input_ids, input_ids]) if guided else input_ids.clone()
                enc_b, cond_b, micro_b = encoder_hidden_states.clone(), cond_embeds.clone(), micro_conds.clone()
                graph, out = ops.capture_graph(lambda: self(model_in, enc_b, cond_b, micro_b))
                wgen = (sum(p._version for p in self.parameters()), self.__dict__.get("_wgen", 0), str(self.compute_dtype), self.training)
                key = key[:-1] + (wgen,)                      # (the first forward may have built / re-cast cached weights)
                self.__dict__["_gen_graph"] = dict(key=key, bufs=(model_in, enc_b, cond_b, micro_b, graph, out))
        for step in order:
            if graph is None:
                out = self(torch.cat([input_ids, input_ids]) if guided else input_ids, encoder_hidden_states, cond_embeds, micro_conds)
            else:
                model_in[:B] = input_ids
                if guided:
                    model_in[B:] = input_ids
                graph.replay()
            q, u = step_noise(noise, step)
            sampled, input_ids, raw = ops.sample_step(out[:B], input_ids, mask_id, V, float(temperatures[step]),
                                                      scheduled_mask_len(S, step, timesteps, noise_schedule),
                                                      uncond_logits=out[B:] if guided else None, guidance_scale=float(scales[step]),
                                                      noise_exp=q, noise_u=u, seed=seed, step=step, want_raw=return_intermediate)
            if return_intermediate:
                intermediate.append(raw)
        return (sampled, intermediate) if return_intermediate else sampled


MaskGiTUViT = MaskGiTUViT_v2
