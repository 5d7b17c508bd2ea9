"""Hand-written forward / backward building blocks shared by the tape engines (muse.MaskGiTUViT_v2 and the general form of
muse.MaskGitTransformer: text conditioning, RMSNorm, non-NormFormer layers).

Activations are channels-last rows ``[B * S, C]``, f32 between blocks.  Every helper returns (output, saved); its *_bwd twin
consumes `saved`, stores parameter gradients in `G` (state-dict name -> tensor) and returns the input gradients.  The host class
provides: `compute_dtype` (torch.float32 = exact-f32 MFMA parity mode, torch.bfloat16 = bf16 GEMM operands with f32 accumulation,
the reference's autocast regime), `config.layer_norm_eps`, `wgrad_stream`, `_side_stream`.
"""
from __future__ import annotations

import os

import torch

from . import ops
from ._hip import MuseHipError

# MUSE_UVIT_BF16_OPERANDS (experiments): bit 0 = AdaLN writes the bf16 GEMM operand, bit 1 = norm backward writes the bf16 copy of dv
_X3_IMAGE_CACHE = os.environ.get("MUSE_X3_IMAGES", "1") != "0"   # bf16x3 mode: share operand images inside a step (0: split per product)
_X3_ATTENTION = os.environ.get("MUSE_X3_ATTENTION", "1") != "0"   # bf16x3 mode: fused attention (csrc/attention3.hip) where it takes the shape
_BF16_OPERANDS = int(os.environ.get("MUSE_UVIT_BF16_OPERANDS", "3"))


X3_WEIGHT_PLANES = os.environ.get("MUSE_X3_WEIGHT_PLANES", "1") != "0"   # bf16x3 mode: weight operand planes kept across steps, refreshed by FusedAdamW
F16_WEIGHT_IMAGES = os.environ.get("MUSE_F16_WEIGHT_IMAGES", "1") != "0"   # f16 mode: the weights' half images kept across steps, refreshed by FusedAdamW
F16_WGRAD_GROUP = os.environ.get("MUSE_F16_WGRAD_GROUP", "1") != "0"       # f16 mode: a block's weight gradients as one grouped launch on the dW stream


class TapeOps:
    # Data-parallel training: `grad_tensors_hook(tensors, final, side_stream)` is called from inside the hand-written backward every
    # time a block's parameter gradients are COMPLETE (muse.GradReducer hangs itself here) - the all-reduce of a bucket then runs on
    # the communication stream while backward computes the earlier blocks, like DDP's bucketed overlap which the reference gets from
    # accelerate (training/train_muse.py:753-759).  `final=True` comes with the last gradients, before autograd sees any of them.
    grad_tensors_hook = None

    def _flush_dw(self):
        """issue the collected weight-gradient products of the finished block(s): one grouped launch per <= 8 products on the
        weight-gradient stream, behind everything the main stream has enqueued so far"""
        pend = self.__dict__.get("_dw_pending")
        if not pend:
            return
        side = self._side_stream
        main = torch.cuda.current_stream(pend[0][2].device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            for i in range(0, len(pend), 8):
                ops.linear_wgrad_group(pend[i:i + 8], None)
        for dyc, xc, *_ in pend:
            dyc.record_stream(side)
            xc.record_stream(side)
        self.__dict__["_dw_pending"] = []
        self.__dict__["_side_busy"] = True

    def _report_grads(self, G, final=False):
        self._flush_dw()
        hook = self.grad_tensors_hook
        if hook is None:
            return
        seen = self.__dict__.setdefault("_grads_reported", set())
        new = [v for k, v in G.items() if k not in seen and not k.startswith("__") and v is not None]
        seen.update(G.keys())
        hook(new, final, self._side_stream if self.__dict__.get("_side_busy", False) else None)
        if final:
            self.__dict__["_grads_reported"] = set()

    @staticmethod
    def _f(p):
        return p.data if p.dtype == torch.float32 else p.data.float()

    def set_compute_dtype(self, dtype):
        """torch.float32 (default): exact-f32 MFMA everywhere (parity mode).  torch.bfloat16: the operands of every weight GEMM
        (linears, 1x1 convs, their dX / dW) are rounded to bf16 and run on the bf16 MFMA kernels with f32 accumulation and f32
        outputs - the reference's autocast regime; the residual stream, norms, AdaLN, GRN, depthwise conv, softmax / attention
        core and the loss stay f32."""
        if dtype not in (torch.float32, torch.bfloat16, "bf16x3", "f16"):
            raise ValueError('compute dtype must be torch.float32, torch.bfloat16, "bf16x3" or "f16"')
        # "bf16x3": the f32 mode's tensors and kernels, with every f32 GEMM computed as three bf16 MFMA products of hi / lo operand
        # planes (ops.f32_gemms_as_bf16x3): TF32-class-or-tighter products (2^-16 relative) at 1/3 of the bf16 matrix rate instead of
        # the 157 TFLOP/s exact-f32 MFMA - the CDNA4 counterpart of the `enable_tf32` regime of configs/cc12m_uvit_clip.yaml:102-103
        # "f16" (round 6): the f32 mode's tensors and kernels, with every weight GEMM as ONE product of IEEE-half operand images
        # (ops.f32_gemms_as_f16): half's 10-bit mantissa IS the TF32 operand format of that `enable_tf32` regime, at the bf16 matrix rate;
        # the exponent range TF32 has and half lacks is covered by power-of-two operand scales (gradients: `f16_grad_scale`).  The
        # attention core runs the bf16x3 mode's fused kernels (csrc/attention3.hip: tighter than TF32).
        self.__dict__["_f32_split3"] = dtype == "bf16x3"
        self.__dict__["_f32_f16"] = dtype == "f16"
        self.compute_dtype = torch.float32 if dtype in ("bf16x3", "f16") else dtype
        self._wcache, self._wcache_owner = {}, {}
        self.__dict__["_wgen"] = self.__dict__.get("_wgen", 0) + 1      # (a kept decoding graph reads the old weight copies: generate2 re-captures)
        return self

    def _gemm_mode(self, backward=False):
        """context manager for one forward / backward pass: f32 GEMMs as three bf16 products in "bf16x3" mode.  A training step
        (forward that records a tape, then its backward) shares the operand images of its products (ops.X3Images): an activation /
        gradient is split once, not once per product that reads it."""
        if self.__dict__.get("_f32_f16", False):
            images = self.__dict__.get("_f16_images")
            if images is None:
                images = self.__dict__["_f16_images"] = ops.F16Images()
            images.backward = bool(backward)
            images.keep = bool(backward or self.__dict__.get("_act_cache_on", False))
            if backward:
                for ov, _ in images.consume_snapshots():      # counters of guarded optimizer steps (muse.FusedAdamW) that have arrived
                    if self.f16_auto_scale:
                        self._f16_scale_policy(ov, self.f16_growth_interval)
                images.set_grad_scale(self.f16_grad_scale_for(self.__dict__.get("_loss_rows", 1)))
            return ops.f32_gemms_as_f16(True, images)
        on = self.__dict__.get("_f32_split3", False)
        images = None
        if on and _X3_IMAGE_CACHE and (backward or self.__dict__.get("_act_cache_on", False)):
            images = self.__dict__.get("_x3_images")
            if images is None:
                images = self.__dict__["_x3_images"] = ops.X3Images()
            images.backward = bool(backward)
        return ops.f32_gemms_as_bf16x3(on, images)

    f16_grad_scale = None      # "f16" mode: the power of two every gradient operand image is scaled by; None = from the loss's row count

    def f16_grad_scale_for(self, loss_rows):
        """"f16" mode: the scale of a backward pass's gradient operands.  The loss is a mean over `loss_rows` token rows, so the largest
        element of d(logits) is at most 1 / loss_rows; the scale puts that bound at 2^10, which leaves a factor 64 of headroom for
        gradients that grow on the way down (clamped elements are counted: `f16_stats`) and keeps full half precision down to 2^-24 of it."""
        if self.f16_grad_scale is not None:
            return float(self.f16_grad_scale)
        n = 1
        while n < int(loss_rows):
            n *= 2
        return float(n) * 1024.0

    def f16_stats(self, reset=True):
        """"f16" mode: (operand elements that overflowed half's range, non-zero elements rounded to zero) since the last call - a device read"""
        images = self.__dict__.get("_f16_images")
        return (0, 0) if images is None else images.stats(reset)

    def f16_update_grad_scale(self, growth_interval=2000, group=None):
        """"f16" mode, the dynamic loss-scaling recipe of fp16 training (torch.cuda.amp.GradScaler's policy) applied to the gradient OPERAND
        scale: call after backward().  An operand overflowed half's range (the gradients are NaN): the scale is halved and False comes
        back - skip the optimizer step.  Otherwise True, and after `growth_interval` good steps in a row the scale doubles.  The scale
        only moves rounding (it is undone exactly in every product's alpha), never values; one device read per call.  Under data
        parallelism (torch.distributed initialised) the overflow count is summed over `group` first, so every rank takes the same
        decision and keeps the same scale - like GradScaler's found_inf all-reduce."""
        images = self.__dict__.get("_f16_images")
        if images is not None and images._stats is not None and torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.all_reduce(images._stats, group=group)
        overflowed, _ = self.f16_stats()
        return self._f16_scale_policy(overflowed, growth_interval)

    # With muse.FusedAdamW nothing has to be called: its multi-tensor kernel skips the update on the device when the backward pass that
    # made the gradients overflowed (muse_adamw_skip_flag), and the next backward pass reads that step's counters - copied to pinned
    # memory behind the update - and moves the scale by the same policy.  f16_auto_scale = False leaves the scale to the caller.
    f16_auto_scale = True
    f16_growth_interval = 2000

    def _f16_scale_policy(self, overflowed, growth_interval):
        cur = self.f16_grad_scale_for(self.__dict__.get("_loss_rows", 1))
        if overflowed:
            self.f16_grad_scale = max(cur * 0.5, 1.0)
            self.__dict__["_f16_good_steps"] = 0
            self.__dict__["_f16_skipped_steps"] = self.__dict__.get("_f16_skipped_steps", 0) + 1
            return False
        good = self.__dict__.get("_f16_good_steps", 0) + 1
        if good >= growth_interval:
            self.f16_grad_scale, good = cur * 2.0, 0
        self.__dict__["_f16_good_steps"] = good
        return True

    def _drop_step_caches(self):
        """start of a forward / end of a backward: nothing of the previous pass (bf16 activation copies, bf16x3 / half operand images) survives"""
        self.__dict__["_act_cache"] = {}
        for name in ("_x3_images", "_f16_images"):
            images = self.__dict__.get(name)
            if images is not None:
                images.clear()

    def mark_weights_changed(self):
        """call after writing parameters behind autograd's back (`p.data.copy_`, EMA swap): drops the cached bf16 weights"""
        self._wcache, self._wcache_owner = {}, {}
        self.__dict__["_wgen"] = self.__dict__.get("_wgen", 0) + 1
        self.__dict__["_gen_graph"] = None

    def train(self, mode: bool = True):
        if mode != self.training:
            self._wcache, self._wcache_owner = {}, {}        # EMA copy_to()/restore() around evaluation write p.data
            self.__dict__["_wgen"] = self.__dict__.get("_wgen", 0) + 1
        return super().train(mode)

    def _c(self, t):
        """GEMM operand in the compute dtype"""
        if self.compute_dtype == torch.float32 or t.dtype == torch.bfloat16:
            return t
        # an activation is a GEMM operand twice: in the forward product and in the backward dW product.  The bf16 copy made
        # for the first use is kept (keyed by the tensor object, dropped when the step's backward has run) instead of casting again.
        cache = self.__dict__.setdefault("_act_cache", {})
        hit = cache.get(id(t))
        if hit is not None and hit[0] is t:
            return hit[1]
        tb = ops.cast_to_bf16(t.contiguous())
        if self.__dict__.get("_act_cache_on", False):
            cache[id(t)] = (t, tb)
        return tb

    def _wb(self, *mods):
        """bf16 compute copy of one Linear / 1x1-conv weight, or of several stacked along the output dim (q|k|v, k|v, wi_0|wi_1),
        as [N_out_total, K_in].  Cached across steps: valid while no source parameter has been updated in place (autograd version
        counters; muse.FusedAdamW refreshes the copy it is given - `p._muse_shadow`, ONE pointer per parameter - inside its own
        kernel without touching the counters).  A weight therefore lives in at most one cached stacking at a time: caching it in
        another one (fused q|k|v against single q / k / v when `attention_supported` differs between sequence lengths, or a
        generate2 call on a model left in train mode) drops the stacking that held it before, which would otherwise keep
        passing the version check with stale bytes after the next optimizer step."""
        key = tuple(id(m.weight) for m in mods)
        ver = tuple(m.weight._version for m in mods)
        cache = self.__dict__.setdefault("_wcache", {})
        owner = self.__dict__.setdefault("_wcache_owner", {})
        hit = cache.get(key)
        if hit is not None and hit[0] == ver and hit[1].device == mods[0].weight.device:
            return hit[1]
        ws = [self._f(m.weight).reshape(m.weight.shape[0], -1) for m in mods]
        w32 = (ws[0] if len(ws) == 1 else torch.cat(ws, dim=0)).contiguous()
        # ("f16" mode: the copy is the weight's IEEE-half operand image, unscaled - FusedAdamW refreshes it as such)
        wb = ops.cast_to_f16(w32) if self.__dict__.get("_f32_f16", False) else ops.cast_to_bf16(w32)
        for pid in key:
            prev = owner.get(pid)
            if prev is not None and prev != key:
                cache.pop(prev, None)          # (its other members lose their shadow too: they are re-cast on their next use)
                for q in prev:
                    if owner.get(q) == prev:
                        del owner[q]
            owner[pid] = key
        cache[key] = (ver, wb)
        off = 0
        for m in mods:   # muse.FusedAdamW writes each parameter's refreshed bf16 copy straight into its row block of the cached tensor
            n = m.weight.shape[0]
            m.weight._muse_shadow = wb[off:off + n]
            if hasattr(m.weight, "_muse_planes"):
                del m.weight._muse_planes       # (one compute copy per parameter: the bf16x3 mode's planes are stale from here on)
            off += n
        return wb

    def _wp(self, *mods):
        """bf16x3 mode: the (hi, lo) operand planes of one weight, or of several stacked along the output dim, as ops.Planes [N_out_total, K_in]
        - cached ACROSS steps like _wb's bf16 copy: muse.FusedAdamW writes each parameter's refreshed planes straight into its row block
        (`p._muse_planes` = (hi-plane rows, elements to the lo plane)) inside its own kernel, so no weight is split (and no q|k|v stacked)
        per step.  Valid while no source parameter has been updated in place by anything else (autograd version counters)."""
        key = tuple(id(m.weight) for m in mods)
        ver = tuple(m.weight._version for m in mods)
        cache = self.__dict__.setdefault("_pcache", {})
        owner = self.__dict__.setdefault("_pcache_owner", {})
        hit = cache.get(key)
        if hit is not None and hit[0] == ver and hit[1].planes.device == mods[0].weight.device:
            return hit[1]
        ws = [self._f(m.weight).reshape(m.weight.shape[0], -1) for m in mods]
        pl = ops._split_planes_now((ws[0] if len(ws) == 1 else torch.cat(ws, dim=0)).contiguous())
        for pid in key:
            prev = owner.get(pid)
            if prev is not None and prev != key:
                cache.pop(prev, None)
                for q in prev:
                    if owner.get(q) == prev:
                        del owner[q]
            owner[pid] = key
        wp = ops.Planes(pl)
        cache[key] = (ver, wp)
        off, lo = 0, pl[0].numel()
        for m in mods:
            n = m.weight.shape[0]
            m.weight._muse_planes = (pl[0][off:off + n], lo)
            if hasattr(m.weight, "_muse_shadow"):
                del m.weight._muse_shadow       # (one compute copy per parameter: the bf16 mode's is stale from here on)
            off += n
        return wp

    def _w2(self, *mods, rows=None):
        """the weight(s) as the GEMM operand of the current compute mode: cached bf16 copy, cached operand planes (bf16x3 mode, when the
        caller names the `rows` of the activation it multiplies and the products are ones the four-plane kernel takes), or the f32 master
        (stacked on the fly)"""
        k_in = mods[0].weight.numel() // mods[0].weight.shape[0]
        if self.compute_dtype == torch.bfloat16 and k_in % 8 == 0:
            return self._wb(*mods)
        if (self.__dict__.get("_f32_f16", False) and F16_WEIGHT_IMAGES and rows is not None and rows >= 128 and k_in >= 128 and k_in % 8 == 0
                and sum(m.weight.shape[0] for m in mods) >= 128 and sum(m.weight.shape[0] for m in mods) % 8 == 0):
            return self._wb(*mods)          # the half image kept across steps (every product of this Linear is one the half kernels take)
        if (rows is not None and rows >= 128 and self.__dict__.get("_f32_split3", False) and X3_WEIGHT_PLANES and k_in >= 128
                and ops.planes_only_ok(sum(m.weight.shape[0] for m in mods), k_in)):
            return self._wp(*mods)
        # (bf16 rows must be whole 16-byte chunks: a weight whose input width is not a multiple of 8 - no shipped configuration -
        #  keeps its products in exact f32, see _f32_pair)
        ws = [self._f(m.weight).reshape(m.weight.shape[0], -1) for m in mods]
        return ws[0] if len(ws) == 1 else torch.cat(ws, dim=0)

    def _pair(self, a, w2):
        """(activation, weight) as operands of one GEMM: the compute dtype, or both f32 when the weight stayed f32 (_w2)"""
        if w2.dtype == torch.float32:
            return (a if a.dtype == torch.float32 else ops.cast_to_f32(a.contiguous())), w2
        return self._c(a), w2

    def _b(self, *mods):
        """f32 bias of one Linear, or of several stacked along the output dim like their weights; None for a model without biases
        (`use_bias=False`, every shipped configuration)"""
        if not self.__dict__.get("_use_bias", False):
            return None
        bs = [self._f(m.bias) for m in mods]
        return bs[0] if len(bs) == 1 else torch.cat(bs)

    def _mm(self, x, w2, residual=None, bias=None):
        """x w2^T (+ bias) (+ residual) -> f32"""
        xx, ww = self._pair(x, w2)
        return ops.linear(xx, ww, out_dtype=torch.float32, residual=residual, bias=bias)

    def _mm_dx(self, dy, w2, lda=None, out=None, accumulate=False):
        """dy w2 -> f32 [rows, K] (optionally accumulated into `out`);  dy [rows, >= N] with row stride lda, w2 [N, K]"""
        N, K = w2.shape
        dyb, ww = self._pair(dy, w2)
        if out is None:
            out = torch.empty((dy.shape[0], K), dtype=torch.float32, device=dy.device)
        ops.gemm(dyb, ww, out, dy.shape[0], K, N, la=0, lb=1, lda=lda or dyb.stride(0), ldb=K, ldc=out.stride(0),
                 accumulate=accumulate)
        return out

    def _mm_dw(self, dy, x, shape2, M=None, lda=None):
        """dy^T x -> f32 [N, K].  bf16 mode: on a second HIP stream (weight gradients are leaves of the backward graph: the
        split-K GEMM and its slice reduction fill CUs the dX / attention / norm chain leaves idle); _run_backward joins the
        stream before it hands the gradients to autograd."""
        if shape2[-1] % 8 or (M if M is not None else shape2[0]) % 8:      # rows of the k-major bf16 operands would not be 16-byte chunks
            dyc = dy if dy.dtype == torch.float32 else ops.cast_to_f32(dy.contiguous())
            xc = x if x.dtype == torch.float32 else ops.cast_to_f32(x.contiguous())
        else:
            dyc, xc = self._c(dy), self._c(x)
        half = None
        if self.__dict__.get("_f32_f16", False) and F16_WGRAD_GROUP and self.wgrad_stream and x.is_cuda and ops.WGRAD_GROUP >= 1:
            # f16 mode: the half images of both operands (the ones the dX / forward products made: the step's image cache) go to the
            # grouped launch on the weight-gradient stream like the bf16 mode's copies
            half = ops.f16_wgrad_operands(dyc, xc, M, lda)
            if half is not None:
                dyc, xc = half
        if not (self.wgrad_stream and (self.compute_dtype == torch.bfloat16 or half is not None) and x.is_cuda):
            dw = torch.empty(shape2, dtype=torch.float32, device=x.device)
            ops.linear_wgrad(dyc, xc, dw, False, M=M, lda=lda)
            return dw
        main = torch.cuda.current_stream(x.device)
        if self._side_stream is None or self._side_stream.device != x.device:
            self._side_stream = torch.cuda.Stream(device=x.device)
        side = self._side_stream
        if ops.WGRAD_GROUP >= 1 and dyc.dtype == xc.dtype and dyc.dtype in (torch.bfloat16, torch.float16):
            # grouped form: the block's weight gradients are collected and issued as ONE launch over all their tiles when the block is
            # done (_flush_dw, called from _report_grads) - the returned tensor is filled then; nothing reads a weight gradient earlier
            with torch.cuda.stream(side):
                dw = torch.empty(shape2, dtype=torch.float32, device=x.device)
            dw.record_stream(main)
            self.__dict__.setdefault("_dw_pending", []).append((dyc, xc, dw, False, M, lda))
            return dw
        side.wait_stream(main)
        with torch.cuda.stream(side):
            dw = torch.empty(shape2, dtype=torch.float32, device=x.device)
            ops.linear_wgrad(dyc, xc, dw, False, M=M, lda=lda)
        dyc.record_stream(side)
        xc.record_stream(side)
        dw.record_stream(main)
        self.__dict__["_side_busy"] = True
        return dw

    def _lin(self, x, mod, residual=None):
        return self._mm(x, self._w2(mod, rows=x.shape[0]), residual=residual, bias=self._b(mod))

    def _lin_bwd(self, dy, x, mod, name, G, need_dx=True):
        w2 = self._w2(mod, rows=dy.shape[0])
        if self.__dict__.get("_use_bias", False):
            G[name + ".bias"] = ops.bias_grad(dy)
        dyc = dy if w2.dtype == torch.float32 else self._c(dy)            # one cast feeds both the dW and the dX product
        G[name + ".weight"] = self._mm_dw(dyc, x, w2.shape).view(mod.weight.shape)
        return self._mm_dx(dyc, w2) if need_dx else None

    def _nm(self, mode):
        """norm kind of a call that does not name one: the model's `norm_type` (0 RMSNorm, 1 LayerNorm; MaskGiTUViT sets `_default_norm_mode`)"""
        return self.__dict__.get("_default_norm_mode", 0) if mode is None else mode

    def _norm(self, x, mod, mode=None, residual=None, want_pre=False):
        mode = self._nm(mode)
        y, pre = ops.norm_res_fwd(x, self._f(mod.weight), float(self.config.layer_norm_eps), mode, residual=residual,
                                  want_pre=want_pre)
        if getattr(mod, "bias", None) is not None and self.__dict__.get("_use_bias", False):
            ops.add_rowvec_(y, self._f(mod.bias))           # LayerNorm bias (reference :130-137; RMSNorm never has one)
        return y, pre

    def _norm_bwd(self, dy, v, mod, name, G, mode=None, dpre=None, gemm_operand=False):
        """v = the tensor that was normalised (x + residual); returns d(x) = d(residual).
        gemm_operand (bf16 mode): dv is also the dY of the next weight GEMMs - the kernel writes its bf16 copy in the same pass
        and _c(dv) finds it instead of launching a cast."""
        mode = self._nm(mode)
        if getattr(mod, "bias", None) is not None and self.__dict__.get("_use_bias", False):
            G[name + ".bias"] = ops.bias_grad(dy)
        if gemm_operand and self.compute_dtype == torch.bfloat16 and _BF16_OPERANDS & 2:
            dv, dw, dvb = ops.norm_res_bwd(dy, v, self._f(mod.weight), float(self.config.layer_norm_eps), mode, dpre=dpre, also_bf16=True)
            self.__dict__.setdefault("_act_cache", {})[id(dv)] = (dv, dvb)
        else:
            dv, dw = ops.norm_res_bwd(dy, v, self._f(mod.weight), float(self.config.layer_norm_eps), mode, dpre=dpre)
        if isinstance(mod.weight, torch.nn.Parameter):       # (a norm without a learnable gain keeps a constant buffer of ones)
            G[name + ".weight"] = dw
        return dv

    def _attention(self, x, ctx, att, B, Sq, Skv, nh, residual=None, drop=None):
        """`drop` = (p, seed, offset): nn.Dropout on the attention probabilities (training mode of models with attention_dropout > 0;
        needs the probabilities, so it runs on the materialised path whatever the compute dtype).
        reference Attention :834-915.  bf16 compute mode: fused attention kernel (S x S never materialised; self- and
        cross-attention) on a packed q|k|v (self) or k|v (cross) projection.  f32 parity mode: the reference's algorithm
        materialised: scores = alpha q k^T (batched per head), softmax, P v, out projection."""
        Cq = x.shape[1]
        hd = Cq // nh
        pdrop = drop[0] if drop is not None else 0.0
        if (self.compute_dtype == torch.bfloat16 and ops.attention_supported(torch.bfloat16, Sq, hd, Skv) and pdrop == 0.0
                and att.key.weight.shape[1] % 8 == 0):
            alpha = 1.0 / float(torch.sqrt(torch.tensor(hd, dtype=torch.float32)))
            xb = self._c(x)                   # (already bf16 when it is an AdaLN output; cached otherwise)
            self_attn = ctx is x
            if self_attn:
                qkv = ops.linear(xb, self._wb(att.query, att.key, att.value), bias=self._b(att.query, att.key, att.value))  # [B*Sq, 3C] bf16
                q, k, v, cb = qkv[:, :Cq], qkv[:, Cq:2 * Cq], qkv[:, 2 * Cq:], xb
            else:
                cb = self._c(ctx)             # the text states feed every layer: one cast per step
                q = ops.linear(xb, self._wb(att.query), bias=self._b(att.query))
                qkv = ops.linear(cb, self._wb(att.key, att.value), bias=self._b(att.key, att.value))   # [B*Skv, 2C] bf16
                k, v = qkv[:, :Cq], qkv[:, Cq:]
            o, lse = ops.attention_fwd_ex(q, k, v, B, Sq, Skv, nh, hd, alpha)
            y = ops.linear(o, self._wb(att.out), out_dtype=torch.float32, residual=residual, bias=self._b(att.out))
            return y, dict(fused=True, self_attn=self_attn, xb=xb, cb=cb, q=q, qkv=qkv, o=o, lse=lse,
                           dims=(B, Sq, Skv, nh, hd, Cq, alpha))
        if ((self.__dict__.get("_f32_split3", False) or self.__dict__.get("_f32_f16", False)) and _X3_ATTENTION and pdrop == 0.0
                and ops.attention_x3_supported(Sq, Skv, hd)
                and x.dtype == torch.float32 and ctx.dtype == torch.float32 and att.key.weight.shape[1] % 8 == 0):
            # "bf16x3" mode: the same fused form on f32 tensors, every product of the core as three bf16 MFMA products like the mode's
            # GEMMs (csrc/attention3.hip); packed q|k|v (self) / k|v (cross) projections: one forward, one dX and one dW product each
            alpha = 1.0 / float(torch.sqrt(torch.tensor(hd, dtype=torch.float32)))
            self_attn = ctx is x
            if self_attn:
                w = self._w2(att.query, att.key, att.value, rows=x.shape[0])     # [3C, C] operand planes kept across steps (or f32, stacked for this step)
                qkv = self._mm(x, w, bias=self._b(att.query, att.key, att.value))
                q, k, v = qkv[:, :Cq], qkv[:, Cq:2 * Cq], qkv[:, 2 * Cq:]
            else:
                q = self._lin(x, att.query)
                w = self._w2(att.key, att.value, rows=ctx.shape[0])
                qkv = self._mm(ctx, w, bias=self._b(att.key, att.value))
                k, v = qkv[:, :Cq], qkv[:, Cq:]
            o, lse = ops.attention_x3_fwd(q, k, v, B, Sq, Skv, nh, hd, alpha)
            y = self._lin(o, att.out, residual=residual)
            return y, dict(fused_x3=True, self_attn=self_attn, x=x, ctx=ctx, q=q, qkv=qkv, w=w, o=o, lse=lse, dims=(B, Sq, Skv, nh, hd, Cq, alpha))
        q, k, v = self._lin(x, att.query), self._lin(ctx, att.key), self._lin(ctx, att.value)
        Sp = (Skv + 7) // 8 * 8
        P = torch.empty((B * nh, Sq, Sp), dtype=torch.float32, device=x.device)
        alpha = 1.0 / float(torch.sqrt(torch.tensor(hd, dtype=torch.float32)))
        sQ, sK, sP = (Sq * Cq, hd), (Skv * Cq, hd), (nh * Sq * Sp, Sq * Sp)
        ops.gemm(q, k, P, Sq, Skv, hd, la=0, lb=0, lda=Cq, ldb=Cq, ldc=Sp, alpha=alpha, batch=B * nh, zdiv=nh, sA=sQ, sB=sK, sC=sP)
        ops.softmax_(P, B * nh * Sq, Skv, Sp)
        Pd = ops.dropout(P, pdrop, drop[1], drop[2]) if pdrop > 0.0 else P
        o = torch.empty((B * Sq, Cq), dtype=torch.float32, device=x.device)
        ops.gemm(Pd, v, o, Sq, hd, Skv, la=0, lb=1, lda=Sp, ldb=Cq, ldc=Cq, batch=B * nh, zdiv=nh, sA=sP, sB=sK, sC=sQ)
        y = self._lin(o, att.out, residual=residual)
        return y, dict(x=x, ctx=ctx, q=q, k=k, v=v, P=P, Pd=Pd, drop=drop if pdrop > 0.0 else None, o=o,
                       dims=(B, Sq, Skv, nh, hd, Cq, Sp, alpha))

    def _attention_bwd(self, dy, sv, att, name, G, self_attn=False):
        """-> (dx, dctx); for self attention the two are already summed and returned as dx (dctx = None)"""
        if sv.get("fused"):
            return self._attention_bwd_fused(dy, sv, att, name, G, self_attn)
        if sv.get("fused_x3"):
            return self._attention_bwd_x3(dy, sv, att, name, G, self_attn)
        B, Sq, Skv, nh, hd, Cq, Sp, alpha = sv["dims"]
        q, k, v, P = sv["q"], sv["k"], sv["v"], sv["P"]
        sQ, sK, sP = (Sq * Cq, hd), (Skv * Cq, hd), (nh * Sq * Sp, Sq * Sp)
        do = self._lin_bwd(dy, sv["o"], att.out, name + ".out", G)
        dv = torch.empty_like(v)
        ops.gemm(sv.get("Pd", P), do, dv, Skv, hd, Sq, la=1, lb=1, lda=Sp, ldb=Cq, ldc=Cq, batch=B * nh, zdiv=nh, sA=sP, sB=sQ, sC=sK)   # dV = P^T dO
        dP = torch.empty_like(P)
        ops.gemm(do, v, dP, Sq, Skv, hd, la=0, lb=0, lda=Cq, ldb=Cq, ldc=Sp, batch=B * nh, zdiv=nh, sA=sQ, sB=sK, sC=sP)   # dP = dO V^T
        if sv.get("drop") is not None:
            ops.dropout(dP, sv["drop"][0], sv["drop"][1], sv["drop"][2], out=dP)
        ops.softmax_bwd_(P, dP, B * nh * Sq, Skv, Sp)                                                                     # dS in place
        dq = torch.empty_like(q)
        ops.gemm(dP, k, dq, Sq, hd, Skv, la=0, lb=1, lda=Sp, ldb=Cq, ldc=Cq, alpha=alpha, batch=B * nh, zdiv=nh, sA=sP, sB=sK, sC=sQ)
        dk = torch.empty_like(k)
        ops.gemm(dP, q, dk, Skv, hd, Sq, la=1, lb=1, lda=Sp, ldb=Cq, ldc=Cq, alpha=alpha, batch=B * nh, zdiv=nh, sA=sP, sB=sQ, sC=sK)
        dx = self._lin_bwd(dq, sv["x"], att.query, name + ".query", G)
        dctx = self._lin_bwd(dk, sv["ctx"], att.key, name + ".key", G)
        wv = self._w2(att.value)
        G[name + ".value.weight"] = self._mm_dw(dv, sv["ctx"], wv.shape)
        if self.__dict__.get("_use_bias", False):
            G[name + ".value.bias"] = ops.bias_grad(dv)
        # dctx += dv Wv ; for self attention query and context are the same tensor: everything lands in dx
        self._mm_dx(dv, wv, out=dctx, accumulate=True)
        if self_attn:
            return dx.add_(dctx), None
        return dx, dctx

    def _attention_bwd_x3(self, dy, sv, att, name, G, self_attn):
        """backward of the "bf16x3" fused form: f32 tensors throughout, the packed projections' dX / dW as single products"""
        B, Sq, Skv, nh, hd, Cq, alpha = sv["dims"]
        do = self._lin_bwd(dy, sv["o"], att.out, name + ".out", G)                        # f32 [B*Sq, C]
        q, qkv, w = sv["q"], sv["qkv"], sv["w"]
        ub = self.__dict__.get("_use_bias", False)
        # (block-by-block form of the long sequences: f32 gradients, no operand planes - the streaming form writes them like the one-tile one)
        blocked = ops.attention_x3_blocked(Sq, Skv) and not ops.attention_x3_streamed(Sq, Skv)
        if sv["self_attn"]:
            if not self_attn:
                raise MuseHipError("self-attention tape replayed as cross-attention")
            lo = qkv.numel()
            if not blocked and not ub and ops.planes_only_ok(qkv.shape[0], qkv.shape[1]) and min(w.shape) >= 128:
                # dqkv feeds one dW and one dX product and nothing else: it exists as their operand planes only
                pl = ops.planes_alloc(tuple(qkv.shape), qkv.device)
                ops.attention_x3_bwd(q, qkv[:, Cq:2 * Cq], qkv[:, 2 * Cq:], sv["o"], do, sv["lse"], B, Sq, Skv, nh, hd, alpha, planes_only=True,
                                     planes=((pl[0][:, :Cq], lo), (pl[0][:, Cq:2 * Cq], lo), (pl[0][:, 2 * Cq:], lo)))
                dqkv = ops.Planes(pl)
            else:
                dqkv = torch.empty_like(qkv)
                pl = None if blocked else ops.x3_new_planes(dqkv)       # ... or as f32 and planes, both out of the kernel
                ops.attention_x3_bwd(q, qkv[:, Cq:2 * Cq], qkv[:, 2 * Cq:], sv["o"], do, sv["lse"], B, Sq, Skv, nh, hd, alpha,
                                     dq=dqkv[:, :Cq], dk=dqkv[:, Cq:2 * Cq], dv=dqkv[:, 2 * Cq:],
                                     planes=None if pl is None else ((pl[0][:, :Cq], lo), (pl[0][:, Cq:2 * Cq], lo), (pl[0][:, 2 * Cq:], lo)))
                ops.x3_put_planes(dqkv, pl)
            gqkv = self._mm_dw(dqkv, sv["x"], (3 * Cq, Cq))
            G[name + ".query.weight"], G[name + ".key.weight"], G[name + ".value.weight"] = gqkv[:Cq], gqkv[Cq:2 * Cq], gqkv[2 * Cq:]
            if ub:
                gb = ops.bias_grad(dqkv)
                G[name + ".query.bias"], G[name + ".key.bias"], G[name + ".value.bias"] = gb[:Cq], gb[Cq:2 * Cq], gb[2 * Cq:]
            return self._mm_dx(dqkv, w), None                                             # d(x) through q, k and v in one product
        if (not blocked and not ub and ops.planes_only_ok(q.shape[0], q.shape[1]) and ops.planes_only_ok(qkv.shape[0], qkv.shape[1]) and min(w.shape) >= 128
                and min(att.query.weight.shape) >= 128):
            # dq and dkv feed one dW and one dX product each and nothing else: planes only
            plq = ops.planes_alloc(tuple(q.shape), q.device)
            plkv = ops.planes_alloc(tuple(qkv.shape), q.device)
            ops.attention_x3_bwd(q, qkv[:, :Cq], qkv[:, Cq:], sv["o"], do, sv["lse"], B, Sq, Skv, nh, hd, alpha, planes_only=True,
                                 planes=((plq[0], q.numel()), (plkv[0][:, :Cq], qkv.numel()), (plkv[0][:, Cq:], qkv.numel())))
            dq, dkv = ops.Planes(plq), ops.Planes(plkv)
        else:
            dq = torch.empty_like(q)
            dkv = torch.empty_like(qkv)
            plq, plkv = (None, None) if blocked else (ops.x3_new_planes(dq), ops.x3_new_planes(dkv))
            ops.attention_x3_bwd(q, qkv[:, :Cq], qkv[:, Cq:], sv["o"], do, sv["lse"], B, Sq, Skv, nh, hd, alpha, dq=dq, dk=dkv[:, :Cq], dv=dkv[:, Cq:],
                                 planes=(None if plq is None else (plq[0], dq.numel()), None if plkv is None else (plkv[0][:, :Cq], dkv.numel()),
                                         None if plkv is None else (plkv[0][:, Cq:], dkv.numel())))
            ops.x3_put_planes(dq, plq)
            ops.x3_put_planes(dkv, plkv)
        dx = self._lin_bwd(dq, sv["x"], att.query, name + ".query", G)
        Ck = att.key.weight.shape[1]
        gkv = self._mm_dw(dkv, sv["ctx"], (2 * Cq, Ck))
        G[name + ".key.weight"], G[name + ".value.weight"] = gkv[:Cq], gkv[Cq:]
        if ub:
            gb = ops.bias_grad(dkv)
            G[name + ".key.bias"], G[name + ".value.bias"] = gb[:Cq], gb[Cq:]
        dctx = self._mm_dx(dkv, w)
        if self_attn:
            return dx.add_(dctx), None
        return dx, dctx

    def _attention_bwd_fused(self, dy, sv, att, name, G, self_attn):
        B, Sq, Skv, nh, hd, Cq, alpha = sv["dims"]
        dev = dy.device
        dyb = self._c(dy)
        wo = self._wb(att.out)
        G[name + ".out.weight"] = self._mm_dw(dyb, sv["o"], att.out.weight.shape)
        ub = self.__dict__.get("_use_bias", False)
        if ub:
            G[name + ".out.bias"] = ops.bias_grad(dy)
        do = ops.linear_dgrad(dyb, wo)                                                     # bf16 [B*Sq, C]
        q, qkv = sv["q"], sv["qkv"]
        if sv["self_attn"]:
            if not self_attn:
                raise MuseHipError("self-attention tape replayed as cross-attention")
            k, v = qkv[:, Cq:2 * Cq], qkv[:, 2 * Cq:]
            dqkv = torch.empty_like(qkv)
            ops.attention_bwd_ex(q, k, v, sv["o"], do, sv["lse"], B, Sq, Skv, nh, hd, alpha, dq=dqkv[:, :Cq], dk=dqkv[:, Cq:2 * Cq],
                                 dv=dqkv[:, 2 * Cq:])
            gqkv = self._mm_dw(dqkv, sv["xb"], (3 * Cq, Cq))
            G[name + ".query.weight"], G[name + ".key.weight"], G[name + ".value.weight"] = gqkv[:Cq], gqkv[Cq:2 * Cq], gqkv[2 * Cq:]
            if ub:
                gb = ops.bias_grad(dqkv)
                G[name + ".query.bias"], G[name + ".key.bias"], G[name + ".value.bias"] = gb[:Cq], gb[Cq:2 * Cq], gb[2 * Cq:]
            dx = torch.empty((B * Sq, Cq), dtype=torch.float32, device=dev)
            ops.linear_dgrad(dqkv, self._wb(att.query, att.key, att.value), out=dx)      # d(x) through q, k and v in one GEMM
            return dx, None
        k, v = qkv[:, :Cq], qkv[:, Cq:]
        dq = torch.empty_like(q)
        dkv = torch.empty_like(qkv)
        ops.attention_bwd_ex(q, k, v, sv["o"], do, sv["lse"], B, Sq, Skv, nh, hd, alpha, dq=dq, dk=dkv[:, :Cq], dv=dkv[:, Cq:])
        G[name + ".query.weight"] = self._mm_dw(dq, sv["xb"], att.query.weight.shape)
        Ck = att.key.weight.shape[1]
        gkv = self._mm_dw(dkv, sv["cb"], (2 * Cq, Ck))
        G[name + ".key.weight"], G[name + ".value.weight"] = gkv[:Cq], gkv[Cq:]
        if ub:
            gb = ops.bias_grad(dkv)
            G[name + ".query.bias"], G[name + ".key.bias"], G[name + ".value.bias"] = ops.bias_grad(dq), gb[:Cq], gb[Cq:]
        dx = torch.empty((B * Sq, Cq), dtype=torch.float32, device=dev)
        ops.linear_dgrad(dq, self._wb(att.query), out=dx)
        dctx = torch.empty((B * Skv, Ck), dtype=torch.float32, device=dev)
        ops.linear_dgrad(dkv, self._wb(att.key, att.value), out=dctx)
        if self_attn:
            return dx.add_(dctx), None
        return dx, dctx

