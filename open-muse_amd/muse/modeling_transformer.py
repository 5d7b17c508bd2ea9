"""muse.MaskGitTransformer for MI355X: the reference's class surface over hand-written HIP kernels.

Replaces muse/modeling_transformer.py:1083-1456 of the reference (constructor kwargs, config keys, state_dict names
and shapes, `forward(input_ids, labels=..., label_smoothing=...) -> logits | (logits, loss)`, `generate2`).
The computation underneath is not an nn.Module graph: one autograd node runs a hand-scheduled forward and a
hand-written backward over libmuse_hip (MFMA GEMMs, fused LayerNorm(+residual), softmax, GLU, fused cross-entropy),
with

  * all parameters living in ONE flat f32 buffer (q/k/v and wi_0/wi_1 adjacent, so the fused [3H,H] / [2I,H] GEMM
    weights are plain views) and all gradients in one flat f32 buffer written directly by the backward kernels
    (no autograd accumulation pass, bucket all-reduce without copies, single-launch AdamW);
  * a bf16 compute copy of the flat weights when compute_dtype is bfloat16 (the reference's autocast regime:
    residual stream / LayerNorm statistics / softmax / loss in f32, GEMM operands bf16, f32 MFMA accumulation).

There is no CPU path: calling forward with CPU tensors raises.
"""
from __future__ import annotations

import contextlib
import math
import weakref
from typing import Callable, List, Optional

import os

import torch
from torch import nn

from . import ops
from ._hip import MuseHipError
from .modeling_utils import ConfigMixin, ModelMixin, register_to_config
from .maskgit_general import GeneralMaskGitEngine
from .sampling import cosine_schedule, decode_seed, scheduled_mask_len, step_noise

_ALIGN = 64  # elements; keeps every parameter view 256-byte aligned inside the flat buffer


class _W(nn.Module):
    """Parameter holder named like the reference's nn.Linear / nn.Embedding / LayerNorm leaf (`.weight`)."""

    def __init__(self, *shape):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(*shape))


class _Attention(nn.Module):
    def __init__(self, H):
        super().__init__()
        self.query, self.key, self.value, self.out = _W(H, H), _W(H, H), _W(H, H), _W(H, H)


class _FeedForward(nn.Module):
    def __init__(self, H, I):
        super().__init__()
        self.pre_mlp_layer_norm = _W(H)
        self.wi_0, self.wi_1 = _W(I, H), _W(I, H)
        self.mid_mlp_layer_norm = _W(I)
        self.wo = _W(H, I)


class _Layer(nn.Module):
    def __init__(self, H, I):
        super().__init__()
        self.attn_layer_norm = _W(H)
        self.attention = _Attention(H)
        self.post_attn_layer_norm = _W(H)
        self.ffn = _FeedForward(H, I)


class _Embed(nn.Module):
    def __init__(self, V, P, H):
        super().__init__()
        self.word_embeddings = _W(V, H)
        self.position_embeddings = _W(P, H)


class _Mlm(nn.Module):
    def __init__(self, H, V):
        super().__init__()
        self.mlm_dense = _W(H, H)
        self.mlm_ln = _W(H)
        self.to_logits = _W(V, H)


class _MaskGitFn(torch.autograd.Function):
    """One autograd node for the whole network.  Parameter gradients are written straight into the model's flat grad
    buffer (p.grad are views of it) unless model.direct_grad is False, in which case they are returned to autograd."""

    @staticmethod
    def forward(ctx, model, input_ids, labels, label_smoothing, need_grad, *params):
        with model._flat_gemm_mode(False):
            logits, loss, saved = model._run_forward(input_ids, labels, label_smoothing, need_grad)
        ctx.model, ctx.saved = model, saved
        ctx.set_materialize_grads(False)
        if loss is None:
            return logits
        return logits, loss

    @staticmethod
    def backward(ctx, g_logits, g_loss=None):
        model = ctx.model
        if ctx.saved is None:
            raise MuseHipError("backward called on a forward that ran without grad")
        with model._flat_gemm_mode(True):
            grads = model._run_backward(ctx.saved, g_logits, g_loss)
        ctx.saved = None
        return (None, None, None, None, None) + tuple(grads)


class MaskGitTransformer(GeneralMaskGitEngine, ModelMixin, ConfigMixin):
    _cast_selects_compute_mode = True     # model.half() / .to(dtype) / from_pretrained(torch_dtype=) pick the compute mode; masters stay f32 (ModelMixin)
    _supports_gradient_checkpointing = True

    @register_to_config
    def __init__(
        self,
        vocab_size,
        hidden_size=768,
        embedding_size=None,
        num_hidden_layers=12,
        num_attention_heads=12,
        intermediate_size=3072,
        hidden_dropout=0.1,
        attention_dropout=0.1,
        max_position_embeddings=256,
        add_cross_attention=False,
        encoder_hidden_size=1024,
        project_encoder_hidden_states=False,
        initializer_range=0.02,
        norm_type="layernorm",
        layer_norm_eps=1e-5,
        use_normformer=True,
        use_encoder_layernorm=True,
        use_mlm_layer=True,
        use_mlm_layernorm=True,
        use_bias=False,
        codebook_size=1024,
        num_vq_tokens=256,
        num_classes=None,
        use_codebook_size_for_output=False,
        use_conv_in_out=False,
        patch_size=1,
        **kwargs,
    ):
        super().__init__()
        if use_conv_in_out:
            # (configs/cc12m_movq.yaml and imagenet_text2image_movq_conv.yaml set it without `embedding_size`, which the reference's own
            #  constructor cannot build: nn.Embedding(vocab_size, None) raises, modeling_transformer.py:1132-1141, :1007)
            raise NotImplementedError("MaskGitTransformer (MI355X build): use_conv_in_out is not built (the two reference configurations "
                                      "that set it do not construct in the reference either; ConvEmbed / ConvMlmLayer of this class are "
                                      "only reachable through it)")
        if norm_type not in ("layernorm", "rmsnorm"):
            raise ValueError(f"norm_type must be 'layernorm' or 'rmsnorm', got {norm_type}")
        # (`embedding_size` is accepted and, as in the reference, unused: the constructor hands `hidden_size` to Embed for both widths,
        #  modeling_transformer.py:1143-1152)
        # the class-conditional NormFormer family (README example, configs/imagenet.yaml) runs on the flat-buffer engine below;
        # everything else - text conditioning, RMSNorm, plain pre-LN layers, no MLM head, biases - on the tape engine (maskgit_general.py)
        self._general = bool(add_cross_attention or project_encoder_hidden_states or norm_type != "layernorm" or not use_normformer
                             or not use_encoder_layernorm or not use_mlm_layer or not use_mlm_layernorm or use_bias)
        if hidden_size % num_attention_heads:
            raise ValueError(f"embed_dim must be divisible by num_heads (got `embed_dim`: {hidden_size} and"
                             f" `num_heads`: {num_attention_heads}).")
        head_dim = hidden_size // num_attention_heads
        if hidden_size % 8 or intermediate_size % 8 or head_dim % 8:
            raise ValueError("hidden_size, intermediate_size and head_dim must be multiples of 8 (16-byte MFMA operand rows)")
        self.vocab_size = vocab_size
        self.hidden_size = hidden_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.intermediate_size = intermediate_size
        self.hidden_dropout = hidden_dropout
        self.attention_dropout = attention_dropout
        self.max_position_embeddings = max_position_embeddings
        self.initializer_range = initializer_range
        self.embedding_size = embedding_size or hidden_size
        self.register_to_config(mask_token_id=vocab_size - 1)
        self.output_size = codebook_size if use_codebook_size_for_output else vocab_size
        self.gradient_checkpointing = False
        self._flat = None
        if self._general:
            self.wgrad_stream = os.environ.get("MUSE_WGRAD_STREAM", "1") != "0"
            self._cd_request = "auto"
            self._gen_build()
            return

        H, I, V = hidden_size, intermediate_size, vocab_size
        self.embed = _Embed(V, max_position_embeddings, H)
        self.transformer_layers = nn.ModuleList([_Layer(H, I) for _ in range(num_hidden_layers)])
        self.encoder_layer_norm = _W(H)
        self.mlm_layer = _Mlm(H, self.output_size)
        self.gradient_checkpointing = False

        # engine state
        self.compute_dtype = "auto"   # "auto": bf16 under torch.autocast(bf16), else f32 (what the reference would do)
        self.direct_grad = True       # backward writes p.grad (views of the flat grad buffer) itself
        self.fused_attention = True   # bf16 compute: fused attention kernels (f32 parity mode keeps the reference algorithm)
        self.grad_ready_hook: Optional[Callable[[int, int], None]] = None  # (flat_begin, flat_end) after each segment
        self._flat = self._flat_grad = self._flat_c = self._flat_ct = None
        self._shadow_fresh = False
        self._shadow_version, self._ct_version, self._shadow_pversion = 0, -1, -1
        self.wgrad_stream = os.environ.get("MUSE_WGRAD_STREAM", "1") != "0"   # bf16 mode: weight-gradient GEMMs on a second HIP stream
        self._side_stream = None
        self.transposed_dgrad = False  # bf16 mode option: W^T copies so dgrad uses the k-contiguous GEMM kernel (measured: no net gain)
        self._build_flat()
        self._init_weights()

    # ------------------------------------------------------------------------------------------------------------
    # flat parameter storage
    # ------------------------------------------------------------------------------------------------------------
    def _param_order(self) -> List[nn.Parameter]:
        order = [self.embed.word_embeddings.weight, self.embed.position_embeddings.weight]
        for l in self.transformer_layers:
            a, f = l.attention, l.ffn
            order += [l.attn_layer_norm.weight, a.query.weight, a.key.weight, a.value.weight, a.out.weight,
                      l.post_attn_layer_norm.weight, f.pre_mlp_layer_norm.weight, f.wi_0.weight, f.wi_1.weight,
                      f.mid_mlp_layer_norm.weight, f.wo.weight]
        m = self.mlm_layer
        order += [self.encoder_layer_norm.weight, m.mlm_dense.weight, m.mlm_ln.weight, m.to_logits.weight]
        return order

    def _build_flat(self, device=None):
        """(Re)allocate the flat f32 parameter buffer on `device` and point every parameter at its slice."""
        params = self._param_order()
        device = device or params[0].device
        offs, n = [], 0
        for p in params:
            offs.append(n)
            n += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        flat = torch.zeros(n, dtype=torch.float32, device=device)
        for p, o in zip(params, offs):
            view = flat[o:o + p.numel()].view(p.shape)
            if p.data.numel() == view.numel() and p.data.device.type != "meta":
                view.copy_(p.data.to(torch.float32))
            p.data = view
            p._muse_owner = weakref.ref(self)
        self._flat, self._offsets, self._flat_n = flat, offs, n
        self._flat_grad = None
        self._flat_c = self._flat_ct = None
        self._shadow_fresh = False
        self._grad_views = None

    def _flat_ok(self) -> bool:
        f = self._flat
        if f is None:
            return False
        params = self._param_order()
        base = f.data_ptr()
        return all(p.data_ptr() == base + o * 4 and p.dtype == torch.float32 and p.device == f.device
                   for p, o in zip(params, self._offsets))

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        if self._general:
            self._wcache, self._wcache_owner = {}, {}
            return out
        if self._flat is not None:
            p0 = self._param_order()[0]
            if p0.dtype != torch.float32:
                raise MuseHipError("master parameters stay float32 (compute precision is selected with set_compute_dtype)")
            self._build_flat(p0.device)
        return out

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        out = super().load_state_dict(state_dict, strict=strict, assign=False)
        if self._general:
            self._wcache, self._wcache_owner = {}, {}
            return out
        if not self._flat_ok():
            self._build_flat()
        self._shadow_fresh = False
        return out

    def _init_weights(self):
        """trunc_normal_(std=initializer_range) for Linear / Embedding weights, ones for LayerNorm
        (reference :1203-1219)."""
        for name, p in self.named_parameters():
            if p.dim() == 1:
                p.data.fill_(1.0)
            else:
                nn.init.trunc_normal_(p.data, std=self.config.initializer_range)

    def _set_gradient_checkpointing(self, module, value=False):
        self.gradient_checkpointing = True  # accepted and ignored: activations fit (288 GB HBM), never recomputed

    def set_compute_dtype(self, dtype):
        if self._general and dtype in ("bf16x3", "f16"):      # f32 tensors; every f32 GEMM as three bf16 MFMA products / one half product
            self._cd_request = torch.float32                  # (tape_ops.set_compute_dtype)
            self.__dict__["_f32_split3"] = dtype == "bf16x3"
            self.__dict__["_f32_f16"] = dtype == "f16"
            return self
        if dtype == "f16":      # the flat engine's f32 mode with every weight GEMM as one IEEE-half product (_flat_gemm_mode)
            self.compute_dtype = torch.float32
            self.__dict__["_f32_f16"] = True
            self._shadow_fresh = False
            return self
        if dtype not in ("auto", torch.float32, torch.bfloat16):
            raise ValueError("compute dtype must be 'auto', torch.float32, torch.bfloat16 or \"f16\"")
        if self._general:
            self._cd_request = dtype          # (the tape helpers read self.compute_dtype: resolved at every forward)
            self.__dict__["_f32_split3"] = False
            self.__dict__["_f32_f16"] = False
            return self
        self.compute_dtype = dtype
        self.__dict__["_f32_f16"] = False
        self._shadow_fresh = False
        return self

    def _flat_gemm_mode(self, backward):
        """the flat (class-conditional) engine's "f16" compute mode: its exact-f32 mode - f32 tensors, kernels and saved activations -
        with every weight GEMM the half kernels take (the layers' four Linears, the head's dense layer; their dX and dW) as ONE
        IEEE-half product with f32 accumulation: TF32's operand precision (tape_ops.set_compute_dtype has the reasoning), gradient
        operands through the pass's power-of-two scale (f16_grad_scale / f16_update_grad_scale / f16_stats as for the tape engines).
        The batched per-head products of the materialised attention core (257 tokens) and the 2025-wide logits head stay exact f32.
        Operands are converted per product (no image cache: this engine rewrites its buffers in place)."""
        if self._general or not self.__dict__.get("_f32_f16", False):
            return contextlib.nullcontext()
        images = self.__dict__.get("_f16_images")
        if images is None:
            images = self.__dict__["_f16_images"] = ops.F16Images(recent=0)
        images.clear()
        images.backward, images.keep = bool(backward), False
        if backward:
            for ov, _ in images.consume_snapshots():
                if self.f16_auto_scale:
                    self._f16_scale_policy(ov, self.f16_growth_interval)
            images.set_grad_scale(self.f16_grad_scale_for(self.__dict__.get("_loss_rows", 1)))
        return ops.f32_gemms_as_f16(True, images)

    def _resolve_cd(self):
        want = self._cd_request if self._general else self.compute_dtype
        if want != "auto":
            return want
        if torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16:
            return torch.bfloat16
        return torch.float32

    # flat gradient buffer --------------------------------------------------------------------------------------
    def flat_params(self) -> torch.Tensor:
        return self._flat

    def flat_grads(self) -> torch.Tensor:
        if self._flat_grad is None or self._flat_grad.device != self._flat.device:
            self._flat_grad = torch.zeros(self._flat_n, dtype=torch.float32, device=self._flat.device)
            self._grad_views = [self._flat_grad[o:o + p.numel()].view(p.shape)
                                for p, o in zip(self._param_order(), self._offsets)]
        return self._flat_grad

    def _param_version(self) -> int:
        """sum of the parameters' autograd version counters: changes whenever anything updates a parameter in place
        (torch.optim.AdamW / Lion `p.add_()`, `p.copy_()`, ...).  Writes through `p.data` are invisible to it; those callers
        (EMA copy_to / restore) are covered by train()/eval() transitions and by mark_weights_changed()."""
        return sum(p._version for p in self._param_order())

    def mark_weights_changed(self):
        """call after writing the f32 master weights behind autograd's back (`p.data.copy_`, a raw kernel on flat_params())"""
        self._shadow_fresh = False
        self._wcache, self._wcache_owner = {}, {}

    def _note_shadow_refreshed(self, fresh: bool):
        """FusedAdamW has just rewritten the master weights and (if `fresh`) the bf16 copy in the same kernel"""
        self._shadow_fresh = bool(fresh)
        self._shadow_pversion = self._param_version()
        self._shadow_version += 1   # transposed weight copies (dgrad) are rebuilt from the refreshed shadow

    def train(self, mode: bool = True):
        if mode != self.training:
            self._shadow_fresh = False   # EMA copy_to()/restore() around evaluation write p.data (reference modeling_ema.py)
            self._wcache, self._wcache_owner = {}, {}
        return nn.Module.train(self, mode)

    def _wgrad_side_stream(self, dev):
        if self._side_stream is None or self._side_stream.device != dev:
            self._side_stream = torch.cuda.Stream(device=dev)
        return self._side_stream

    def compute_weights(self, cd) -> torch.Tensor:
        """flat weights in the compute dtype (the f32 master itself, or its bf16 shadow).  The shadow is re-cast whenever
        it is not PROVEN current: only FusedAdamW (same kernel writes both) and an unchanged parameter-version sum count."""
        if cd == torch.float32:
            return self._flat
        if self._flat_c is None or self._flat_c.device != self._flat.device:
            self._flat_c = torch.empty(self._flat_n, dtype=torch.bfloat16, device=self._flat.device)
            self._shadow_fresh = False
        pv = self._param_version()
        if not self._shadow_fresh or pv != self._shadow_pversion:
            ops.cast_to_bf16(self._flat, self._flat_c)
            self._shadow_fresh = True
            self._shadow_pversion = pv
            self._shadow_version += 1
        return self._flat_c

    def compute_weights_t(self, cd):
        """bf16 compute mode: a second flat copy holding every Linear weight TRANSPOSED ([K_in, N_out]; the fused q|k|v and
        wi_0|wi_1 blocks transposed as one [H, 3H] / [H, 2I] matrix), so that dX = dY W runs on the k-contiguous x
        k-contiguous GEMM kernel (the fastest variant) instead of the transposing-read one.  Refreshed together with the
        bf16 shadow (once per optimizer step)."""
        if cd != torch.bfloat16 or not self.transposed_dgrad:
            return None
        Wc = self.compute_weights(cd)
        if self._flat_ct is None or self._flat_ct.device != Wc.device:
            self._flat_ct = torch.empty_like(Wc)
            self._ct_version = -1
        if self._ct_version != self._shadow_version:
            H, I, V = self.hidden_size, self.intermediate_size, self.output_size
            off = self._offsets
            mats = []
            for li in range(self.num_hidden_layers):
                b0 = 2 + li * 11
                mats += [(b0 + 1, 3 * H, H), (b0 + 4, H, H), (b0 + 7, 2 * I, H), (b0 + 10, H, I)]
            t0 = 2 + self.num_hidden_layers * 11
            mats += [(t0 + 1, H, H), (t0 + 3, V, H)]
            for idx, n, k in mats:
                o = off[idx]
                ops.transpose(Wc[o:o + n * k].view(n, k), self._flat_ct[o:o + n * k].view(k, n))
            self._ct_version = self._shadow_version
        return self._flat_ct

    # ------------------------------------------------------------------------------------------------------------
    # forward / backward
    # ------------------------------------------------------------------------------------------------------------
    def forward(self, input_ids, encoder_hidden_states=None, encoder_attention_mask=None, labels=None,
                label_smoothing=0.0, cond_dropout_prob=0.0, **kwargs):
        """reference :1224-1281.  Extra kwargs are accepted and ignored like the reference's **kwargs (train_muse.py passes
        cond_embeds / loss_weight / micro_conds, :742-750); `cond_dropout_uniforms` [B] (tests) replaces the draw of
        `prob_mask_like`.  `encoder_attention_mask` raises: the reference's own mask path fails with a TypeError
        (`make_attention_mask(..., dtype=...)`, :214) and its fused-attention path refuses masks (:191-192)."""
        if self.config.add_cross_attention and encoder_hidden_states is None:
            raise ValueError("If `add_cross_attention` is True, `encoder_hidden_states` should be provided.")
        if encoder_attention_mask is not None:
            raise NotImplementedError("encoder_attention_mask is not supported (the reference's mask path is broken as well)")
        if self._general:
            cd = self._resolve_cd()
            if cd != self.compute_dtype:
                self.compute_dtype = cd
                self._wcache, self._wcache_owner = {}, {}
            if encoder_hidden_states is not None and not self.config.add_cross_attention:
                # (layers without a cross-attention block: the reference fails on the missing module, :886-889; same error as the flat engine)
                raise ValueError("this model was built without cross attention (add_cross_attention=False)")
            return self._gen_call(input_ids, encoder_hidden_states, labels, label_smoothing, cond_dropout_prob,
                                  kwargs.get("cond_dropout_uniforms"))
        if encoder_hidden_states is not None:
            raise ValueError("this model was built without cross attention (add_cross_attention=False)")
        if not input_ids.is_cuda:
            raise MuseHipError("MaskGitTransformer (MI355X build) has no CPU path: move the model and inputs to the GPU")
        if not self._flat_ok():
            self._build_flat()
        params = self._param_order()
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)  # (grad mode is off inside Function.forward)
        out = _MaskGitFn.apply(self, input_ids, labels, float(label_smoothing), need_grad, *params)
        return out

    def _run_forward(self, input_ids, labels, label_smoothing, need_grad):
        cd = self._resolve_cd()
        H, I, nh, V = self.hidden_size, self.intermediate_size, self.num_attention_heads, self.output_size
        hd = H // nh
        B, S = input_ids.shape
        T = B * S
        Sp = (S + 7) // 8 * 8
        eps = float(self.config.layer_norm_eps)
        alpha = 1.0 / float(torch.sqrt(torch.tensor(hd, dtype=torch.float32)))
        dev = input_ids.device
        Wc = self.compute_weights(cd)           # flat, compute dtype
        Wf = self._flat                         # flat, f32 (LayerNorm scales, embeddings)
        off = self._offsets
        ids = input_ids.contiguous()

        def wview(t, idx, shape):
            o = off[idx]
            n = 1
            for s in shape:
                n *= s
            return t[o:o + n].view(shape)

        x = ops.embed_fwd(ids, self.embed.word_embeddings.weight.data, self.embed.position_embeddings.weight.data)
        # nn.Dropout sites of the reference (training mode only): embeddings :956, attention probabilities :237, feed-forward
        # :797.  Masks are Philox streams (seed drawn per forward from torch's CPU generator, one offset range per site); the
        # backward regenerates them.  Attention dropout needs the probabilities, so it runs on the materialised path.
        pd_h = float(self.hidden_dropout) if self.training else 0.0
        pd_a = float(self.attention_dropout) if self.training else 0.0
        seed = int(torch.randint(0, 2 ** 62, (1,))) if (pd_h > 0.0 or pd_a > 0.0) else 0
        self._last_dropout = (seed, pd_h, pd_a)
        site = lambda li, k: ((li * 2 + k + 1) << 40)   # noqa: E731  (k = 0 attention, 1 feed-forward; 0 = embeddings)
        if pd_h > 0.0:
            ops.dropout(x, pd_h, seed, 0, out=x)
        fused = self.fused_attention and ops.attention_supported(cd, S, hd) and pd_a == 0.0
        saved = {"layers": [], "cd": cd, "ids": ids, "B": B, "S": S, "Sp": Sp, "fused": fused, "drop": (seed, pd_h, pd_a)} if need_grad else None

        for li in range(self.num_hidden_layers):
            b0 = 2 + li * 11
            w_ln1 = wview(Wf, b0 + 0, (H,))
            w_qkv = wview(Wc, b0 + 1, (3 * H, H))
            w_out = wview(Wc, b0 + 4, (H, H))
            w_post = wview(Wf, b0 + 5, (H,))
            w_pre = wview(Wf, b0 + 6, (H,))
            w_01 = wview(Wc, b0 + 7, (2 * I, H))
            w_mid = wview(Wf, b0 + 9, (I,))
            w_o2 = wview(Wc, b0 + 10, (H, I))

            ln1, mu1, rs1 = ops.layernorm_fwd(x, w_ln1, eps, cd)
            qkv = ops.linear(ln1, w_qkv)
            if fused:
                # softmax(alpha Q K^T) V in one kernel, S x S never leaves the CU (reference :206-210 / :226-240)
                ctx, P = ops.attention_fwd(qkv, B, S, nh, hd, alpha)   # P := row log-sum-exp
            else:
                P = torch.empty((B * nh, S, Sp), dtype=cd, device=dev)
                # scores[b,h] = alpha * Q K^T        (reference :226-231)
                ops.gemm(qkv, qkv, P, S, S, hd, la=0, lb=0, lda=3 * H, ldb=3 * H, ldc=Sp, a_off=0, b_off=H, alpha=alpha,
                         batch=B * nh, zdiv=nh, sA=(S * 3 * H, hd), sB=(S * 3 * H, hd), sC=(nh * S * Sp, S * Sp))
                ops.softmax_(P, B * nh * S, S, Sp)
                Pd = ops.dropout(P, pd_a, seed, site(li, 0)) if pd_a > 0.0 else P    # (reference :237)
                ctx = torch.empty((T, H), dtype=cd, device=dev)
                # ctx[b,:,h] = P V                   (reference :238-240)
                ops.gemm(Pd, qkv, ctx, S, hd, S, la=0, lb=1, lda=Sp, ldb=3 * H, ldc=H, b_off=2 * H, batch=B * nh, zdiv=nh,
                         sA=(nh * S * Sp, S * Sp), sB=(S * 3 * H, hd), sC=(S * H, hd))
            ao = ops.linear(ctx, w_out)
            if ops.layernorm_pair_ok(ao, x, cd, 1):   # both LayerNorms in one pass (bit-identical to the two calls below)
                x1, mu_p, rs_p, ln2, mu2, rs2 = ops.layernorm_pair_fwd(ao, x, w_post, w_pre, eps)
            else:
                x1, mu_p, rs_p = ops.layernorm_fwd(ao, w_post, eps, torch.float32, residual=x)   # x + LN(attn)  (:882-884)
                ln2, mu2, rs2 = ops.layernorm_fwd(x1, w_pre, eps, cd)
            ab = ops.linear(ln2, w_01)
            # gelu(a)*b and the NormFormer mid-LN in one pass (:789-797); bf16 mode: the GLU product h is not kept for backward (it is
            # recomputed there from ab with the erf the GLU backward evaluates anyway: 101 MB per layer neither written nor read)
            h, hm, mu_m, rs_m = ops.ffn_mid_fwd(ab, w_mid, eps, keep_h=(cd != torch.bfloat16))
            if pd_h > 0.0:
                ops.dropout(hm, pd_h, seed, site(li, 1), out=hm)   # (:797) in place: only the dropped tensor is needed again (dW_o)
            x2 = ops.linear(hm, w_o2, out_dtype=torch.float32, residual=x1)                  # x + FFN(x)    (:902-903)
            if need_grad:
                if not fused and pd_a > 0.0:
                    P = (P, Pd)
                saved["layers"].append(dict(x=x, mu1=mu1, rs1=rs1, ln1=ln1, qkv=qkv, P=P, ctx=ctx, ao=ao, mu_p=mu_p,
                                            rs_p=rs_p, x1=x1, mu2=mu2, rs2=rs2, ln2=ln2, ab=ab, h=h, mu_m=mu_m,
                                            rs_m=rs_m, hm=hm))
            x = x2

        t0 = 2 + self.num_hidden_layers * 11
        w_enc = wview(Wf, t0 + 0, (H,))
        w_dense = wview(Wc, t0 + 1, (H, H))
        w_mln = wview(Wf, t0 + 2, (H,))
        w_log = wview(Wc, t0 + 3, (V, H))
        xf, mu_e, rs_e = ops.layernorm_fwd(x, w_enc, eps, cd)
        d = ops.linear(xf, w_dense)
        g = ops.gelu_fwd(d)
        gl, mu_g, rs_g = ops.layernorm_fwd(g, w_mln, eps, cd)
        Vp = (V + 7) // 8 * 8   # 16-byte rows for the backward GEMMs (vocab 2025 -> 2032); pad columns never read as logits
        logits = torch.empty((T, Vp), dtype=torch.float32, device=dev)
        ops.gemm(gl, w_log, logits, T, V, H, la=0, lb=0, lda=H, ldb=H, ldc=Vp)

        loss = None
        loss_out = lse = lab = None
        if labels is not None:
            lab = labels.contiguous().view(-1)
            self.__dict__["_loss_rows"] = T          # ("f16" mode: bounds d(logits), tape_ops.f16_grad_scale_for)
            loss_out, lse = ops.cross_entropy_fwd(logits, lab, label_smoothing, vocab=V)
            loss = loss_out[0]
        if need_grad:
            saved.update(x_last=x, mu_e=mu_e, rs_e=rs_e, xf=xf, d=d, g=g, mu_g=mu_g, rs_g=rs_g, gl=gl, logits=logits,
                         loss_out=loss_out, lse=lse, labels=lab, ls=label_smoothing)
        out_logits = logits.view(B, S, V) if Vp == V else logits[:, :V].contiguous().view(B, S, V)
        return out_logits, loss, saved

    def _run_backward(self, sv, g_logits, g_loss):
        cd = sv["cd"]
        H, I, nh, V = self.hidden_size, self.intermediate_size, self.num_attention_heads, self.output_size
        hd = H // nh
        B, S, Sp = sv["B"], sv["S"], sv["Sp"]
        T = B * S
        alpha = 1.0 / float(torch.sqrt(torch.tensor(hd, dtype=torch.float32)))
        dev = sv["ids"].device
        Wc = self.compute_weights(cd)
        Wf = self._flat
        off = self._offsets
        params = self._param_order()
        G = self.flat_grads()
        # accumulate iff the parameter already has a gradient (autograd semantics: None -> assign, else add)
        acc = [p.grad is not None for p in params]
        if self.direct_grad:
            for p, gv in zip(params, self._grad_views):
                if p.grad is None or p.grad.data_ptr() != gv.data_ptr():
                    if p.grad is not None:
                        gv.copy_(p.grad)
                    p.grad = gv
            GW = G
        else:
            GW = torch.zeros_like(G)
            acc = [False] * len(params)

        def view(t, idx, shape):
            o = off[idx]
            n = 1
            for s in shape:
                n *= s
            return t[o:o + n].view(shape)

        Wt = self.compute_weights_t(cd)
        seed, pd_h, pd_a = sv["drop"]
        site = lambda li, k: ((li * 2 + k + 1) << 40)   # noqa: E731

        def dgrad(dy, w, idx):
            """dx = dy @ w  (w = compute weights [N_out, K_in] at flat index idx)"""
            if Wt is None:
                return ops.linear_dgrad(dy, w)
            n, k = w.shape
            return ops.linear(dy, view(Wt, idx, (k, n)))

        # Weight gradients are a side branch of the backward graph (nothing downstream reads them before the all-reduce /
        # optimizer), so they run on a second HIP stream: the split-K dW GEMMs and their slice reductions fill the CUs the
        # main chain leaves idle (dX GEMMs with 195 tiles on 256 CUs, kernel tails, the HBM-bound LayerNorm / GLU backward).
        main = torch.cuda.current_stream(dev)
        side = self._wgrad_side_stream(dev) if (self.wgrad_stream and cd == torch.bfloat16) else None
        # column sums of the LayerNorm / GLU weight gradients: queued, then launched on the side stream - or, without one, handed to
        # the layer's grouped reduction launch (98 launches of 6 us per serial step otherwise)
        csq = [] if (side is not None or (cd == torch.bfloat16 and ops.WGRAD_GROUP >= 1)) else None

        # Grouped form (ops.WGRAD_GROUP >= 1, bf16 mode): the weight gradients of a layer are collected and issued as ONE launch of
        # the 256^2 kernel over all their tiles + ONE reduction launch (slice sums and the layer's column sums) when the layer is
        # done (`ready`): 144 tiles of 257 K-tiles each at config B (one slice beside the backward chain, five when it runs alone:
        # ops.WGRAD_GROUP / WGRAD_GROUP_SERIAL) instead of 4 launches of 243-288 blocks with 9-64 K-tiles each (a third of a short
        # block is prologue + f32 epilogue) and 8 small reduction launches.
        group = [] if (cd == torch.bfloat16 and ops.WGRAD_GROUP >= 1) else None

        def flush_group():
            if not group:
                return
            if side is None:
                ops.linear_wgrad_group(group, csq, split=ops.WGRAD_GROUP_SERIAL)     # nothing runs beside it: slices that fill the chip
            else:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    ops.linear_wgrad_group(group, csq)
                for dy, x, *_ in group:
                    dy.record_stream(side)   # the caching allocator must not hand these blocks out while the side stream reads them
                    x.record_stream(side)
            group.clear()

        def wgrad(dy, x, dw, accumulate, **kw):
            if group is not None:
                group.append((dy, x, dw, accumulate, kw.get("M"), kw.get("lda")))
                return
            if side is None:
                return ops.linear_wgrad(dy, x, dw, accumulate, **kw)
            side.wait_stream(main)           # dy (and on the first use x) were produced on the main stream
            with torch.cuda.stream(side):
                ops.flush_colsums(csq)       # LayerNorm / GLU weight-gradient column sums queued since the last call
                ops.linear_wgrad(dy, x, dw, accumulate, **kw)
            dy.record_stream(side)           # the caching allocator must not hand these blocks out while the side stream reads them
            x.record_stream(side)

        def ready(i0, i1):
            flush_group()
            if self.grad_ready_hook is not None and self.direct_grad:
                end = off[i1] if i1 < len(off) else self._flat_n
                if side is None:
                    self.grad_ready_hook(off[i0], end)
                else:   # the range is complete once both streams have passed this point
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        ops.flush_colsums(csq)
                        self.grad_ready_hook(off[i0], end)

        # ---- loss / head ----------------------------------------------------------------------------------------
        dlog = None
        if g_loss is not None and sv["labels"] is not None:
            dlog = ops.cross_entropy_bwd(sv["logits"], sv["labels"], sv["lse"], sv["loss_out"],
                                         g_loss.reshape(1).to(torch.float32).contiguous(), sv["ls"], cd, vocab=V)
        Vp = sv["logits"].shape[1]
        if g_logits is not None:
            gl_ = torch.zeros((T, Vp), dtype=cd, device=dev)
            gl_[:, :V] = g_logits.reshape(T, V).to(cd)
            dlog = gl_ if dlog is None else dlog + gl_
        if dlog is None:
            raise MuseHipError("backward without an incoming gradient")
        t0 = 2 + self.num_hidden_layers * 11
        w_enc, w_dense = view(Wf, t0 + 0, (H,)), view(Wc, t0 + 1, (H, H))
        w_mln, w_log = view(Wf, t0 + 2, (H,)), view(Wc, t0 + 3, (V, H))
        # dW_logits[V,H] = dlog^T gl ; dgl[T,H] = dlog W_logits   (dlog rows are Vp wide, only V valid)
        wgrad(dlog, sv["gl"], view(GW, t0 + 3, (V, H)), acc[t0 + 3], M=V, lda=Vp)
        dgl = torch.empty((T, H), dtype=cd, device=dev)
        if Wt is not None and V % 8 == 0:
            ops.gemm(dlog, view(Wt, t0 + 3, (H, V)), dgl, T, H, V, la=0, lb=0, lda=Vp, ldb=V, ldc=H)
        else:
            ops.gemm(dlog, w_log, dgl, T, H, V, la=0, lb=1, lda=Vp, ldb=H, ldc=H)
        dg = ops.layernorm_bwd(dgl, sv["g"], w_mln, sv["mu_g"], sv["rs_g"], cd, view(GW, t0 + 2, (H,)), acc[t0 + 2], colsum_queue=csq)
        dd = ops.gelu_bwd(sv["d"], dg)
        wgrad(dd, sv["xf"], view(GW, t0 + 1, (H, H)), acc[t0 + 1])
        dxf = dgrad(dd, w_dense, t0 + 1)
        bf = cd == torch.bfloat16   # bf16 mode: LayerNorm backward also writes the bf16 copy of dx the next layer's GEMMs read
        dx = ops.layernorm_bwd(dxf, sv["x_last"], w_enc, sv["mu_e"], sv["rs_e"], torch.float32, view(GW, t0 + 0, (H,)),
                               acc[t0 + 0], also_bf16=bf, colsum_queue=csq)
        dx, dxc = dx if bf else (dx, dx)
        ready(t0, t0 + 4)

        # ---- layers, last to first ----------------------------------------------------------------------------------
        for li in reversed(range(self.num_hidden_layers)):
            s = sv["layers"][li]
            b0 = 2 + li * 11
            w_ln1, w_qkv, w_out = view(Wf, b0 + 0, (H,)), view(Wc, b0 + 1, (3 * H, H)), view(Wc, b0 + 4, (H, H))
            w_post, w_pre = view(Wf, b0 + 5, (H,)), view(Wf, b0 + 6, (H,))
            w_01, w_mid, w_o2 = view(Wc, b0 + 7, (2 * I, H)), view(Wf, b0 + 9, (I,)), view(Wc, b0 + 10, (H, I))
            # FFN
            wgrad(dxc, s["hm"], view(GW, b0 + 10, (H, I)), acc[b0 + 10])
            dhm = dgrad(dxc, w_o2, b0 + 10)
            if pd_h > 0.0:
                ops.dropout(dhm, pd_h, seed, site(li, 1), out=dhm)
            dab = ops.ffn_mid_bwd(dhm, s["h"], s["ab"], w_mid, s["mu_m"], s["rs_m"], view(GW, b0 + 9, (I,)), acc[b0 + 9], colsum_queue=csq)
            wgrad(dab, s["ln2"], view(GW, b0 + 7, (2 * I, H)), acc[b0 + 7])
            dln2 = dgrad(dab, w_01, b0 + 7)
            if ops.layernorm_pair_ok(s["ao"], s["x1"], cd, 2) and dln2.dtype == torch.bfloat16:
                dx1, dao = ops.layernorm_pair_bwd(dln2, s["x1"], w_pre, s["mu2"], s["rs2"], dx, s["ao"], w_post, s["mu_p"], s["rs_p"],
                                                  view(GW, b0 + 6, (H,)), acc[b0 + 6], view(GW, b0 + 5, (H,)), acc[b0 + 5], colsum_queue=csq)
            else:
                dx1 = ops.layernorm_bwd(dln2, s["x1"], w_pre, s["mu2"], s["rs2"], torch.float32, view(GW, b0 + 6, (H,)),
                                        acc[b0 + 6], dres=dx, colsum_queue=csq)
                # attention
                dao = ops.layernorm_bwd(dx1, s["ao"], w_post, s["mu_p"], s["rs_p"], cd, view(GW, b0 + 5, (H,)), acc[b0 + 5], colsum_queue=csq)
            wgrad(dao, s["ctx"], view(GW, b0 + 4, (H, H)), acc[b0 + 4])
            dctx = dgrad(dao, w_out, b0 + 4)
            qkv, P = s["qkv"], s["P"]
            if sv["fused"]:
                dqkv = ops.attention_bwd(qkv, s["ctx"], dctx, P, B, S, nh, hd, alpha)
                wgrad(dqkv, s["ln1"], view(GW, b0 + 1, (3 * H, H)), acc[b0 + 1])
                dln1 = dgrad(dqkv, w_qkv, b0 + 1)
                dx = ops.layernorm_bwd(dln1, s["x"], w_ln1, s["mu1"], s["rs1"], torch.float32, view(GW, b0 + 0, (H,)),
                                       acc[b0 + 0], dres=dx1, also_bf16=bf and li > 0, colsum_queue=csq)
                dx, dxc = dx if (bf and li > 0) else (dx, dx)
                sv["layers"][li] = None
                ready(b0, b0 + 11)
                continue
            dqkv = torch.empty((T, 3 * H), dtype=cd, device=dev)
            sQ, sP_, sX = (S * 3 * H, hd), (nh * S * Sp, S * Sp), (S * H, hd)
            P, Pd = P if isinstance(P, tuple) else (P, P)
            # dV = P^T dctx   (the dropped probabilities, as the forward used them)
            ops.gemm(Pd, dctx, dqkv, S, hd, S, la=1, lb=1, lda=Sp, ldb=H, ldc=3 * H, c_off=2 * H, batch=B * nh, zdiv=nh,
                     sA=sP_, sB=sX, sC=sQ)
            # dP = dctx V^T
            dP = torch.empty((B * nh, S, Sp), dtype=cd, device=dev)
            ops.gemm(dctx, qkv, dP, S, S, hd, la=0, lb=0, lda=H, ldb=3 * H, ldc=Sp, b_off=2 * H, batch=B * nh, zdiv=nh,
                     sA=sX, sB=sQ, sC=sP_)
            if pd_a > 0.0:
                ops.dropout(dP, pd_a, seed, site(li, 0), out=dP)
            ops.softmax_bwd_(P, dP, B * nh * S, S, Sp)   # dS in place
            # dQ = alpha dS K ; dK = alpha dS^T Q
            ops.gemm(dP, qkv, dqkv, S, hd, S, la=0, lb=1, lda=Sp, ldb=3 * H, ldc=3 * H, b_off=H, c_off=0, alpha=alpha,
                     batch=B * nh, zdiv=nh, sA=sP_, sB=sQ, sC=sQ)
            ops.gemm(dP, qkv, dqkv, S, hd, S, la=1, lb=1, lda=Sp, ldb=3 * H, ldc=3 * H, b_off=0, c_off=H, alpha=alpha,
                     batch=B * nh, zdiv=nh, sA=sP_, sB=sQ, sC=sQ)
            wgrad(dqkv, s["ln1"], view(GW, b0 + 1, (3 * H, H)), acc[b0 + 1])
            dln1 = dgrad(dqkv, w_qkv, b0 + 1)
            dx = ops.layernorm_bwd(dln1, s["x"], w_ln1, s["mu1"], s["rs1"], torch.float32, view(GW, b0 + 0, (H,)),
                                   acc[b0 + 0], dres=dx1, also_bf16=bf and li > 0, colsum_queue=csq)
            dx, dxc = dx if (bf and li > 0) else (dx, dx)
            sv["layers"][li] = None  # free activations as we go
            ready(b0, b0 + 11)

        if pd_h > 0.0:
            ops.dropout(dx, pd_h, seed, 0, out=dx)
        ops.embed_bwd(sv["ids"], dx, view(GW, 0, tuple(params[0].shape)), view(GW, 1, tuple(params[1].shape)), acc[0])
        # position rows beyond S received no gradient this step
        if not acc[1] and S < self.max_position_embeddings:
            view(GW, 1, tuple(params[1].shape))[S:].zero_()
        ready(0, 2)
        if side is not None:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                ops.flush_colsums(csq)
            main.wait_stream(side)   # the optimizer (and anything else on the main stream) sees every weight gradient
        elif csq:
            ops.flush_colsums(csq)
        if self.direct_grad:
            return [None] * len(params)
        return [view(GW, i, tuple(p.shape)) for i, p in enumerate(params)]

    # ------------------------------------------------------------------------------------------------------------
    # MaskGit parallel decoding (reference :1363-1456, "next" row f2 of SURVEY.md section 8): per step one forward on the
    # HIP kernels and ONE device call for everything token-level (libmuse_hip muse_sample_step: softmax, categorical
    # sample, Gumbel-perturbed confidence, k-th-smallest threshold, re-mask).  Nothing in the loop reads the device.
    # ------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def generate2(self, input_ids=None, class_ids=None, encoder_hidden_states=None, negative_embeds=None, temperature=1.0,
                  timesteps=18, guidance_scale=0, noise_schedule=cosine_schedule, generator=None, noise=None, hip_graph=False,
                  **kwargs):
        """-> sampled ids [B, num_vq_tokens].  `noise` (tests): per step a pair (exponential draws [B*S, codebook_size],
        uniform draws [B, S]) replacing the in-kernel Philox stream, e.g. the draws the reference's CPU generator produced.
        hip_graph=True captures the forward once (fixed input buffer) and replays it every step: one graph launch instead of
        ~12 kernel launches per layer - what small-batch decoding is bound by.
        Like the reference, `class_ids` is shifted by codebook_size IN PLACE (:1388-1389) and the temperature compounds
        across steps (:1451)."""
        cfg = self.config
        mask_id, S, V = cfg.mask_token_id, cfg.num_vq_tokens, cfg.codebook_size
        B = len(class_ids) if class_ids is not None else encoder_hidden_states.shape[0]
        dev = class_ids.device if class_ids is not None else encoder_hidden_states.device
        if class_ids is not None:
            class_ids += V
        if input_ids is None:
            input_ids = torch.full((B, S), mask_id, dtype=torch.long, device=dev)
        seed = decode_seed(generator) if noise is None else 0
        # classifier-free guidance (:1394-1401): conditional batch followed by the unconditional one (zeros unless negative_embeds)
        guided = encoder_hidden_states is not None and guidance_scale > 0
        enc = encoder_hidden_states
        if guided:
            enc = torch.cat([encoder_hidden_states, torch.zeros_like(encoder_hidden_states) if negative_embeds is None else negative_embeds])
        off = 1 if class_ids is not None else 0
        rows = 2 * B if guided else B
        model_in = torch.empty((rows, S + off), dtype=torch.long, device=dev)
        if class_ids is not None:
            model_in[:, 0] = class_ids.repeat(2) if guided else class_ids
        sampled = input_ids
        graph = None
        if hip_graph and enc is None:
            graph, model_in, logits = self._decode_graph(B, S + 1, dev)
            model_in[:, 0] = class_ids
        for step in range(timesteps):
            model_in[:B, off:] = input_ids
            if guided:
                model_in[B:, off:] = input_ids
            if graph is None:
                logits = self(model_in, encoder_hidden_states=enc)       # [rows, S + off, vocab] f32; row 0 of each image: the class token
            else:
                graph.replay()
            temperature = temperature * (1.0 - 1.0 * (step + 1) / timesteps)
            q, u = step_noise(noise, step)
            sampled, input_ids, _ = ops.sample_step(logits[:B, off:], input_ids, mask_id, V, temperature,
                                                    scheduled_mask_len(S, step, timesteps, noise_schedule),
                                                    uncond_logits=logits[B:, off:] if guided else None,
                                                    guidance_scale=float(guidance_scale) if guided else 0.0,
                                                    noise_exp=q, noise_u=u, seed=seed, step=step)
        return sampled

    def _decode_graph(self, B, S, device):
        """(graph, input buffer [B, S], logits) of the forward captured as a HIP graph, kept across generate2 calls for as long
        as the weight buffers, the compute dtype and the mode stay the same.  The graph reads the weights through their buffers:
        in-place updates are seen; a stale bf16 shadow is re-cast here (host logic the replay does not run)."""
        if self._general:
            raise MuseHipError("hip_graph decoding is built for the class-conditional model only")
        if not self._flat_ok():
            self._build_flat()
        cd = self._resolve_cd()
        key = (B, S, str(cd), self.training, str(device), self._flat.data_ptr())
        hit = getattr(self, "_decode_graphs", {}).get(key)
        if hit is None:
            model_in = torch.zeros((B, S), dtype=torch.long, device=device)
            graph, logits = ops.capture_graph(lambda: self(model_in))
            hit = (graph, model_in, logits)
            self._decode_graphs = {key: hit}      # one live graph: a new shape / dtype / weight buffer replaces it
        else:
            self.compute_weights(cd)
        return hit

    def generate(self, *args, **kwargs):
        raise NotImplementedError("use generate2 (the reference's generate() is broken: modeling_transformer.py:1307)")
