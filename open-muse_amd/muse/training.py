"""Train-step pieces of the hot path that live in the reference's *training script* rather than in its package:

  * prepare_inputs_and_labels  — training/train_maskgit_imagenet.py:357-394 (VQ encode + cosine-schedule mask sampling)
  * FusedAdamW                 — the `fused_adamw` optimizer choice (:242-261, apex FusedAdam adam_w_mode), one HIP
                                 launch over the model's flat parameter buffer
  * GradReducer                — replaces accelerate/DDP (:152-158, :305, :433): bucketed all-reduce of the flat
                                 gradient buffer over RCCL (xGMI) on a side stream, overlapped with backward
  * TrainStep                  — the loop body (:405-452) strung together for benchmarks / smoke tests
"""
from __future__ import annotations

import os
import weakref
import warnings
from typing import Optional

import torch
import torch.distributed as dist

from . import ops
from ._hip import MuseHipError


@torch.no_grad()
def prepare_inputs_and_labels(vq_model, pixel_values, class_ids, mask_id, min_masking_rate: float = 0.0, timesteps=None,
                              noise=None, generator=None, image_tokens=None, codebook_size=None):
    """-> (input_ids [B,S+1], labels [B,S+1], soft_targets=None, mask_prob [B]).

    `timesteps` [B] and `noise` [B,S] are the two torch.rand draws of the reference (:375, :381); pass them in for
    bit-reproducible masks (parity tests), otherwise they are drawn on the GPU.  With pre-encoded `image_tokens` the
    tokenizer is not touched: `vq_model` may be None when `codebook_size` (the class-token offset, :388) is given."""
    if codebook_size is None:
        if vq_model is None:
            raise ValueError("prepare_inputs_and_labels: without a vq_model pass codebook_size (the class-token offset)")
        codebook_size = vq_model.num_embeddings
    if image_tokens is None:
        image_tokens = vq_model.get_code(pixel_values)      # == vq_model.encode(pixel_values)[1] (:369) without the unused z_q
    B, S = image_tokens.shape
    dev = image_tokens.device
    if timesteps is None:
        timesteps = torch.rand(B, device=dev, generator=generator)
    if noise is None:
        noise = torch.rand(B, S, device=dev, generator=generator)
    input_ids, labels, mask_prob = ops.mask_sample(image_tokens.contiguous(), class_ids.contiguous(),
                                                   timesteps.float().contiguous(), noise.float().contiguous(), int(mask_id),
                                                   int(codebook_size), float(min_masking_rate))
    return input_ids, labels, None, mask_prob


def mask_or_random_replace_tokens(image_tokens, mask_id, config, mask_schedule=None, is_train=True, *, timesteps=None, noise=None,
                                  rects=None, generator=None):
    """training/train_muse.py:149-226 on the device (muse_mask_tokens), same signature and return value
    `(input_ids, labels, loss_weight, mask_prob)`; `config` is read like the reference reads its OmegaConf node
    (`config.training.get(...)`, `config.training.min_masking_rate`).

    Host-side randomness follows the reference: `random.choices` for eval_mask_ratios, `random.random()` /
    `random.randint` for the contiguous-region coin and rectangle (which costs the reference - and this function - one
    device read of the per-image mask counts); the two device draws (`timesteps` [B], `noise` [B, S]) can be passed in for
    bit-reproducible masks.  The reference's `noise_type` test (:202) is always true, so "random_replace" also writes the mask
    id; that is kept."""
    import math
    import random
    tr = config.training
    B, S = image_tokens.shape
    dev = image_tokens.device
    mask_prob_in = None
    if not is_train and tr.get("eval_mask_ratios", None):
        mask_prob_in = torch.tensor(random.choices(tr.eval_mask_ratios, k=B), device=dev)
    elif timesteps is None:
        timesteps = torch.rand(B, device=dev, generator=generator)
    if mask_prob_in is None and mask_schedule is not None:
        from .sampling import cosine_schedule
        if mask_schedule is not cosine_schedule:       # any other schedule: evaluate it like the reference does (:160-161)
            mask_prob_in = mask_schedule(timesteps.float()).clip(tr.min_masking_rate)
    region_p = tr.get("mask_contiguous_region_prob", None)
    region = region_p is not None and random.random() < region_p
    if region and rects is None:
        # ONE mask_prob for the rectangle sizes and for the kernel (which otherwise evaluates its own double-precision cosine: the two
        # could disagree at a .5 rounding boundary): the f32 cosine of the reference (:160-161) is handed to muse_mask_tokens
        mp = mask_prob_in if mask_prob_in is not None else torch.cos(timesteps.float() * (math.pi * 0.5)).clip(tr.min_masking_rate)
        mask_prob_in = mp
        counts = (S * mp).round().clamp(min=1).tolist()
        res = int(S ** 0.5)
        boxes = []
        for n in counts:                                # :186-199, the reference's "a bit handwavy" rectangle
            n = int(n)
            h = min(random.randint(math.ceil(n / res), min(res, n)), res)
            w = min(math.ceil(n / h), res)
            boxes.append([random.randint(0, res - h), random.randint(0, res - w), h, w])
        rects = torch.tensor(boxes, dtype=torch.int32)
    if not region:
        rects = None
        if noise is None:
            noise = torch.rand(B, S, device=dev, generator=generator)
    all_labels = bool(tr.get("predict_all_tokens", False)) or tr.get("noise_type", "mask") == "random_replace"
    input_ids, labels, loss_weight, mask_prob = ops.mask_tokens(
        image_tokens, int(mask_id), timesteps=None if mask_prob_in is not None else timesteps, mask_prob=mask_prob_in, noise=noise,
        rects=rects, min_masking_rate=float(tr.min_masking_rate), all_labels=all_labels, want_weight=all_labels)
    return input_ids, labels, loss_weight, mask_prob


def cond_dropout(encoder_hidden_states, clip_embeds, empty_embeds, empty_clip_embeds, cond_dropout_prob, uniforms=None,
                 generator=None):
    """training/train_muse.py:715-731: one uniform per image decides, for the text states and the pooled embedding alike,
    whether the conditioning is kept (u < p) or replaced by the empty-prompt embedding -> (encoder_hidden_states, cond_embeds)"""
    B = encoder_hidden_states.shape[0]
    if uniforms is None:
        uniforms = torch.rand(B, device=encoder_hidden_states.device, generator=generator)
    enc = ops.cond_dropout(encoder_hidden_states, empty_embeds, uniforms, cond_dropout_prob).view(encoder_hidden_states.shape)
    cond = ops.cond_dropout(clip_embeds, empty_clip_embeds, uniforms, cond_dropout_prob).view(clip_embeds.shape)
    # (the kernel works in f32; torch.where in the reference keeps the inputs' dtype)
    return enc.to(encoder_hidden_states.dtype), cond.to(clip_embeds.dtype)


def _owner_of(params):
    """the muse.MaskGitTransformer that owns all `params` in its flat buffer, or None when the parameters are ordinary tensors
    (muse.MaskGiTUViT): those are stepped one launch per tensor with the same kernel"""
    owner = None
    if all(getattr(p, "_muse_owner", None) is None for p in params):
        return None
    for p in params:
        ref = getattr(p, "_muse_owner", None)
        m = ref() if ref is not None else None
        if m is None:
            raise MuseHipError("FusedAdamW: either all or none of the parameters may live in a flat parameter buffer")
        if owner is None:
            owner = m
        elif owner is not m:
            raise MuseHipError("FusedAdamW: parameters of several models in one optimizer are not supported")
    return owner


class FusedAdamW(torch.optim.Optimizer):
    """AdamW (decoupled weight decay, torch.optim.AdamW numerics) as ONE kernel launch over the model's flat f32
    parameter / gradient buffers; also refreshes the bf16 compute copy of the weights in the same pass.

    Drop-in for `optimizer_cls(model.parameters(), lr=..., betas=..., weight_decay=..., eps=...)`.  `state_dict()` /
    `load_state_dict()` use torch.optim.AdamW's layout ({"state": {i: {"step", "exp_avg", "exp_avg_sq"}}, "param_groups"}),
    so optimizer checkpoints written by the reference's `adamw` choice load here and vice versa."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        params = list(params)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        # Parameter groups: `params` may be the list of dicts training/train_muse.py:425-445 builds (weight decay on the matrices,
        # none on bias / LayerNorm / embedding weights) - torch.optim semantics, every group its own lr / betas / eps / weight_decay.
        # One group runs the single-set kernels (muse_adamw_flat / _multi); several run the *_groups kernels, still ONE launch.
        if len(self.param_groups) > 8:
            raise MuseHipError(f"FusedAdamW: at most 8 parameter groups (got {len(self.param_groups)})")
        every = self._all_params()
        owner = _owner_of(every)
        self._model = weakref.ref(owner) if owner is not None else None
        self._flat_gid = self._seg_dev = None
        if owner is not None:
            theirs = owner._param_order()
            if len(every) != len(theirs) or {id(q) for q in every} != {id(q) for q in theirs}:
                raise MuseHipError("FusedAdamW steps the model's whole flat parameter buffer: pass model.parameters() "
                                   "(all of them, as one list or split into groups); for a subset use torch.optim.AdamW")
            gid = {id(q): k for k, grp in enumerate(self.param_groups) for q in grp["params"]}
            self._flat_gid = [gid[id(q)] for q in theirs]           # group of each parameter, in the flat buffer's order
        self._m = self._v = None
        self._step = 0
        self._table = self._chunk_first = self._table_key = self._stage = None   # per-tensor mode: device table of muse_adamw_multi
        self._ranges_done = self._ranges_done_live = None                        # (step, [(begin, end), ...]) applied inside backward
        self._upd_stream, self._upd_used = None, False                           # stream of the per-bucket update (begin_step_in_reducer)
        self.grad_scale = 1.0   # multiplied into the gradient inside the kernel (GradReducer sets 1/world for SUM reductions)

    def _all_params(self):
        """every parameter, in torch.optim's state-dict order (group by group)"""
        return [q for grp in self.param_groups for q in grp["params"]]

    def _segments(self, model):
        """device tables of the flat buffer's parameter-group segments: (seg_end int64 - absolute element offsets -, seg_group int32).
        Neighbouring parameters of one group form one segment; a parameter's alignment padding stays with it."""
        flat = model.flat_params()
        if self._seg_dev is not None and self._seg_dev[0].device == flat.device and self._seg_dev[2] == tuple(model._offsets):
            return self._seg_dev[0], self._seg_dev[1]
        ends, gids = [], []
        offs = list(model._offsets) + [flat.numel()]
        for i, k in enumerate(self._flat_gid):
            if gids and gids[-1] == k:
                ends[-1] = offs[i + 1]
            else:
                ends.append(offs[i + 1])
                gids.append(k)
        self._seg_dev = (torch.tensor(ends, dtype=torch.int64, device=flat.device),
                         torch.tensor(gids, dtype=torch.int32, device=flat.device), tuple(model._offsets))
        return self._seg_dev[0], self._seg_dev[1]

    def _flat_grad_checked(self, model, params):
        """the flat gradient buffer, after making sure it really holds this step's gradients: every p.grad must be the
        parameter's view of it (what backward writes with model.direct_grad); a gradient that lives elsewhere (autograd-
        returned with direct_grad=False, or assigned by the user) is copied in."""
        g = model.flat_grads()
        for p, gv in zip(params, model._grad_views):
            if not p.requires_grad:
                raise MuseHipError("FusedAdamW steps the whole flat buffer: a frozen parameter (requires_grad=False) is not supported")
            if p.grad.data_ptr() != gv.data_ptr():
                if p.grad.shape != gv.shape:
                    raise MuseHipError("FusedAdamW: gradient shape differs from its parameter")
                gv.copy_(p.grad)
                p.grad = gv
        return g

    @torch.no_grad()
    def step(self, closure=None):
        self._check_not_partial()
        loss = closure() if closure is not None else None
        if self._model is None:
            return self._step_per_tensor(loss)
        model = self._model()
        if model is None:
            raise MuseHipError("FusedAdamW: the model that owned these parameters is gone")
        if not model._flat_ok():
            raise MuseHipError("FusedAdamW: the model's flat parameter buffer was rebuilt (model.to(...) / load after the "
                               "optimizer was created is fine, replacing p.data is not)")
        flat = model.flat_params()
        order = model._param_order()
        if any(p.grad is None for p in order):
            if all(p.grad is None for p in order):
                return loss  # nothing to do before the first backward (torch skips None grads)
            raise MuseHipError("FusedAdamW: some parameters have no gradient; the flat step needs all of them")
        g = self._flat_grad_checked(model, order)
        self._ensure_flat_state(flat)
        self._step += 1
        shadow = model._flat_c if model._flat_c is not None and model._flat_c.device == flat.device else None
        done = self._ranges_done
        self._ranges_done = None
        if done is not None and done[0] == self._step:
            # backward already applied this step's update range by range (begin_step_in_backward); cover what it did not report
            covered = sorted(done[1])
            pos = 0
            for b, e in covered:
                if b > pos:
                    self._apply(model, pos, b, shadow)
                pos = max(pos, e)
            if pos < flat.numel():
                self._apply(model, pos, flat.numel(), shadow)
        else:
            self._apply(model, 0, flat.numel(), shadow)
        model._note_shadow_refreshed(shadow is not None)
        return loss

    def _apply(self, model, b, e, shadow):
        flat, g = model.flat_params(), model.flat_grads()
        if len(self.param_groups) > 1:
            seg_end, seg_group = self._segments(model)
            ops.adamw_flat_groups(flat[b:e], g[b:e], self._m[b:e], self._v[b:e], None if shadow is None else shadow[b:e], b,
                                  seg_end, seg_group, self.param_groups, self._step_for_apply, grad_scale=float(self.grad_scale))
            return
        grp = self.param_groups[0]
        ops.adamw_flat(flat[b:e], g[b:e], self._m[b:e], self._v[b:e], None if shadow is None else shadow[b:e], float(grp["lr"]),
                       grp["betas"][0], grp["betas"][1], grp["eps"], grp["weight_decay"], self._step_for_apply,
                       grad_scale=float(self.grad_scale))

    @property
    def _step_for_apply(self):
        d = self._ranges_done_live
        return d[0] if d is not None else self._step

    def begin_step_in_backward(self, model):
        """Arm the optimizer to apply THIS step's update inside backward: every time backward reports a finished range of the flat
        gradient buffer (model.grad_ready_hook) the AdamW kernel runs on that slice, on the stream the report comes from (the
        weight-gradient stream) - the HBM-bound update hides behind the MFMA-bound dX chain of the earlier layers instead of
        following backward.  Element-wise identical to one launch over the whole buffer.  The following step() only advances the
        step count and covers ranges backward did not report.  Valid when nothing sits between backward and step(): no gradient
        clipping, no gradient accumulation, no all-reduce (muse.TrainStep checks this and arms it)."""
        self._check_not_partial()
        flat = model.flat_params()
        self._ensure_flat_state(flat)
        if any((o * 4) % 16 for o in model._offsets):
            return False                      # (slices must keep the kernel's 16-byte alignment)
        self._ranges_done_live = (self._step + 1, [])
        self._direct_grad_before = model.direct_grad
        shadow = model.compute_weights(torch.bfloat16) if model._resolve_cd() == torch.bfloat16 else None

        def hook(begin, end):
            live = self._ranges_done_live
            if live is None or end <= begin:
                return
            self._apply(model, begin, end, model._flat_c if shadow is not None else None)
            live[1].append((begin, end))
        model.direct_grad = True
        model.grad_ready_hook = hook
        return True

    def end_step_in_backward(self, model, failed=False):
        """`failed`: backward raised.  Ranges it had already reported carry this step's update (weights changed, moments advanced)
        while the step count has not: the step can be neither retried nor skipped, and every later call says so."""
        model.grad_ready_hook = None
        model.direct_grad = getattr(self, "_direct_grad_before", model.direct_grad)
        self._ranges_done, self._ranges_done_live = self._ranges_done_live, None
        if failed and self._ranges_done is not None and self._ranges_done[1]:
            self._partial_step = True

    def _check_not_partial(self):
        if getattr(self, "_partial_step", False):
            raise MuseHipError("FusedAdamW: an earlier step failed inside backward after AdamW had already been applied to part of "
                               "the parameters (in-backward / behind-the-all-reduce update); the optimizer state is inconsistent - "
                               "restore parameters and optimizer from a checkpoint (or set MUSE_OPT_IN_BACKWARD=0 / "
                               "MUSE_OPT_IN_REDUCER=0 to keep the update out of backward)")

    def begin_step_in_reducer(self, model, reducer):
        """The data-parallel form of begin_step_in_backward: the AdamW kernel runs on each gradient bucket right after its all-reduce
        (GradReducer.post_reduce), while backward is still computing the earlier layers and the next bucket is being filled - instead
        of one pass over all parameters after the last all-reduce.  The kernel runs on the reducer's (high-priority) stream, behind
        the bucket's collective.  MUSE_OPT_REDUCER_STREAM=own puts it on a normal-priority stream of its own instead (same results
        bit for bit, but measured much slower at one rank: 69.6 vs 61.5 ms per step, profiles/r02_ab_dp1_updstream_*.json - the
        low-priority update is starved until the end of the step and then serialises).  Same validity conditions as
        begin_step_in_backward; call end_step_in_reducer after reducer.finish()."""
        flat = model.flat_params()
        self._ensure_flat_state(flat)
        if any((o * 4) % 16 for o in model._offsets):
            return False
        self._ranges_done_live = (self._step + 1, [])
        has_shadow = model._resolve_cd() == torch.bfloat16
        if has_shadow:
            model.compute_weights(torch.bfloat16)
        # where the per-bucket update runs: "comm" (default) = the reducer's high-priority stream right behind the collective;
        # "side" = the model's weight-gradient stream, in order with the dW GEMMs still to come (the placement of the single-GPU
        # in-backward update); "own" = a normal-priority stream of its own.  One-rank launch line on MI355X, per step
        # (profiles/r03_dp1_update_placement.txt): comm 60.7 ms / side 62.2 ms with the default 4 hardware queues, 66.7 / 65.8 ms
        # with GPU_MAX_HW_QUEUES=8 (plain single-GPU step on that box: 59.9 ms) - the 8-queue penalty is not the update's placement.
        where = os.environ.get("MUSE_OPT_REDUCER_STREAM", "comm")
        own = flat.is_cuda and where in ("own", "side")
        if own and where == "side" and hasattr(model, "_wgrad_side_stream"):
            self._upd_stream = model._wgrad_side_stream(flat.device)
        elif own and (self._upd_stream is None or self._upd_stream.device != flat.device or
                      self._upd_stream is getattr(model, "_side_stream", None)):
            self._upd_stream = torch.cuda.Stream(device=flat.device)
        self._upd_used = False

        def hook(begin, end):
            live = self._ranges_done_live
            if live is None or end <= begin:
                return
            if own:
                self._upd_stream.wait_stream(torch.cuda.current_stream(flat.device))   # (the reducer's stream, behind the collective)
                with torch.cuda.stream(self._upd_stream):
                    self._apply(model, begin, end, model._flat_c if has_shadow else None)
                self._upd_used = True
            else:
                self._apply(model, begin, end, model._flat_c if has_shadow else None)
            live[1].append((begin, end))
        reducer.post_reduce = hook
        return True

    def end_step_in_reducer(self, reducer, failed=False):
        reducer.post_reduce = None
        if self._upd_used:   # the caller's stream continues behind the last bucket's update
            torch.cuda.current_stream(self._upd_stream.device).wait_stream(self._upd_stream)
            self._upd_used = False
        self._ranges_done, self._ranges_done_live = self._ranges_done_live, None
        if failed and self._ranges_done is not None and self._ranges_done[1]:
            self._partial_step = True

    def _ensure_flat_state(self, flat):
        if self._m is None:
            self._m = torch.zeros_like(flat)
            self._v = torch.zeros_like(flat)
        elif self._m.device != flat.device:       # the model moved after the state was created / loaded: follow it
            self._m, self._v = self._m.to(flat.device), self._v.to(flat.device)
        if self._m.shape != flat.shape:
            raise MuseHipError("FusedAdamW: optimizer state does not match the model's flat parameter buffer")

    def _step_per_tensor(self, loss):
        """parameters that are ordinary (contiguous f32) tensors: one muse_adamw_multi launch over all of them.  torch.optim.AdamW
        semantics: a parameter without a gradient is skipped and keeps its own state; the step count is shared (all
        parameters of these models receive a gradient every step).  With several parameter groups the table carries each
        tensor's group and the launch is muse_adamw_multi_groups."""
        params = [(p, k) for k, grp in enumerate(self.param_groups) for p in grp["params"] if p.grad is not None]
        # (operand planes of the bf16x3 mode ride in the group column of the grouped table: a one-group model with planes takes that kernel too)
        multi = len(self.param_groups) > 1 or any(getattr(p, "_muse_planes", None) is not None
                                                  or getattr(getattr(p, "_muse_shadow", None), "dtype", None) == torch.float16 for p, _ in params)
        if not params:
            return loss
        if self._m is None:
            self._m, self._v = {}, {}
        self._step += 1
        rows = []
        for p, gk in params:
            if p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                raise MuseHipError("FusedAdamW: parameters and gradients must be contiguous float32 tensors")
            k = id(p)
            if k not in self._m:
                self._m[k] = torch.zeros_like(p)
                self._v[k] = torch.zeros_like(p)
            elif self._m[k].device != p.device:
                self._m[k], self._v[k] = self._m[k].to(p.device), self._v[k].to(p.device)
            shadow = getattr(p, "_muse_shadow", None)   # the model's cached bf16 compute copy of this weight (MaskGiTUViT bf16 mode)
            if shadow is not None and (shadow.numel() != p.numel() or shadow.device != p.device or not shadow.is_contiguous()):
                shadow = None
            # ... or (bf16x3 mode, tape_ops._wp) its (hi, lo) operand planes: hi-plane rows + the element distance to the lo plane, which
            # rides in the group column above bit 8
            lo = -1 if (shadow is not None and shadow.dtype == torch.float16) else 0     # ("f16" mode: the copy is IEEE half)
            planes = getattr(p, "_muse_planes", None)
            if planes is not None and shadow is None and planes[0].numel() == p.numel() and planes[0].device == p.device \
                    and planes[0].is_contiguous() and planes[1] > 0:
                shadow, lo = planes[0], int(planes[1])
            row = (p.data.data_ptr(), p.grad.data_ptr(), self._m[k].data_ptr(), self._v[k].data_ptr(),
                   shadow.data_ptr() if shadow is not None else 0, p.numel())
            rows.append(row + (gk | (lo << 8),) if multi else row)
        # one launch for all tensors.  Gradients are fresh allocations every step, so the pointer table is rebuilt every step: ~500
        # rows staged through two alternating PINNED host buffers and copied asynchronously (a pageable copy would make the host
        # wait for the stream and lose its run-ahead into the next step)
        dev = params[0][0].device
        ncol = 7 if multi else 6
        key = tuple(rows)
        if self._table_key != key:
            nt = len(rows)
            if self._table is None or self._table.shape[0] < nt or self._table.shape[1] != ncol or self._table.device != dev:
                self._table = torch.empty((nt, ncol), dtype=torch.int64, device=dev)
                self._chunk_first = torch.empty(nt + 1, dtype=torch.int32, device=dev)
                self._stage = [(torch.empty((nt, ncol), dtype=torch.int64).pin_memory(), torch.empty(nt + 1, dtype=torch.int32).pin_memory(),
                                torch.cuda.Event()) for _ in range(2)]
                self._stage_i = 0
            ht, hf, ev = self._stage[self._stage_i]
            self._stage_i ^= 1
            ev.synchronize()      # the copy issued from this staging buffer two rebuilds ago has run
            nchunks = 0
            first = [0] * (nt + 1)
            for i, r in enumerate(rows):
                first[i] = nchunks
                nchunks += (r[5] + 4095) // 4096
            first[nt] = nchunks
            ht[:nt].copy_(torch.tensor(rows, dtype=torch.int64))
            hf[:nt + 1].copy_(torch.tensor(first, dtype=torch.int32))
            self._table[:nt].copy_(ht[:nt], non_blocking=True)
            self._chunk_first[:nt + 1].copy_(hf[:nt + 1], non_blocking=True)
            ev.record(torch.cuda.current_stream(dev))
            self._table_key, self._nchunks = key, nchunks
        # "f16" compute mode: the update is guarded ON THE DEVICE by the overflow counter of the backward pass that made these gradients
        # (an operand beyond half's range turns them NaN): the kernel leaves parameters and moments untouched when it is non-zero, the
        # model's next backward pass halves the gradient scale (tape_ops._gemm_mode) - GradScaler's policy without a host round trip
        # (the host does not know about the skip when it happens: self._step advances anyway, so the bias corrections of the following
        #  updates are those of one step later - a factor that tends to 1)
        guard = ops._F16_GUARD[0]
        if guard is not None and (guard._stats is None or guard._stats.device != dev):
            guard = None
        if guard is not None:
            if dist.is_available() and dist.is_initialized():
                dist.all_reduce(guard._stats)        # every rank skips together (GradScaler's found_inf all-reduce)
            ops.check(ops.lib().muse_adamw_skip_flag(guard._stats.data_ptr()), "muse_adamw_skip_flag")
        try:
            if multi:
                ops.adamw_multi_groups(self._table, self._chunk_first, len(rows), self._nchunks, self.param_groups, self._step,
                                       grad_scale=float(self.grad_scale))
            else:
                grp = self.param_groups[0]
                ops.adamw_multi(self._table, self._chunk_first, len(rows), self._nchunks, float(grp["lr"]), grp["betas"][0], grp["betas"][1],
                                grp["eps"], grp["weight_decay"], self._step, grad_scale=float(self.grad_scale))
        finally:
            if guard is not None:
                ops.check(ops.lib().muse_adamw_skip_flag(None), "muse_adamw_skip_flag")
                guard.after_optimizer_step()
                ops._F16_GUARD[0] = None
        return loss

    # ---- checkpointing in torch.optim.AdamW's layout ----------------------------------------------------------------
    def _flat_offsets(self, model):
        return {id(q): o for q, o in zip(model._param_order(), model._offsets)}

    def state_dict(self):
        ps = self._all_params()
        groups, n0 = [], 0
        for grp in self.param_groups:                      # torch.optim's packing: parameters numbered group by group
            g = {k: v for k, v in grp.items() if k != "params"}
            g["params"] = list(range(n0, n0 + len(grp["params"])))
            n0 += len(grp["params"])
            groups.append(g)
        state = {}
        if self._step > 0 and self._m is not None:
            stepv = torch.tensor(float(self._step))
            if self._model is None:
                for i, p in enumerate(ps):
                    if id(p) in self._m:
                        state[i] = {"step": stepv.clone(), "exp_avg": self._m[id(p)], "exp_avg_sq": self._v[id(p)]}
            else:
                offs = self._flat_offsets(self._model())
                for i, p in enumerate(ps):
                    o, n = offs[id(p)], p.numel()
                    state[i] = {"step": stepv.clone(), "exp_avg": self._m[o:o + n].view(p.shape),
                                "exp_avg_sq": self._v[o:o + n].view(p.shape)}
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        if "state" not in sd or "param_groups" not in sd:
            raise MuseHipError("FusedAdamW.load_state_dict expects torch.optim's layout {'state', 'param_groups'}")
        ps = self._all_params()
        if len(sd["param_groups"]) != len(self.param_groups):
            raise ValueError("loaded state dict has a different number of parameter groups")
        ids = []
        for mine, grp in zip(self.param_groups, sd["param_groups"]):
            gi = list(grp.get("params", range(len(ids), len(ids) + len(mine["params"]))))
            if len(gi) != len(mine["params"]):
                raise ValueError("loaded state dict contains a parameter group that doesn't match the size of optimizer's group")
            ids += gi
            mine.update({k: v for k, v in grp.items() if k != "params"})
        state = {ids.index(k) if k in ids else k: v for k, v in sd["state"].items()}
        steps = {int(float(v["step"])) for v in state.values()}
        if len(steps) > 1:
            raise MuseHipError("FusedAdamW keeps one step count for all parameters; the checkpoint has several")
        self._step = steps.pop() if steps else 0
        if not state:
            self._m = self._v = None
            return
        model = self._model() if self._model is not None else None
        if model is None:
            self._m, self._v = {}, {}
            for i, st in state.items():
                p = ps[i]
                for name, store in (("exp_avg", self._m), ("exp_avg_sq", self._v)):
                    t = st[name]
                    if tuple(t.shape) != tuple(p.shape):
                        raise ValueError(f"FusedAdamW.load_state_dict: {name} of parameter {i} has shape {tuple(t.shape)}, "
                                         f"expected {tuple(p.shape)}")
                    store[id(p)] = t.detach().to(device=p.device, dtype=torch.float32).contiguous().clone()
            return
        if len(state) != len(ps):
            raise MuseHipError("FusedAdamW (flat): the checkpoint must carry state for every parameter")
        flat = model.flat_params()
        offs = self._flat_offsets(model)
        self._m, self._v = torch.zeros_like(flat), torch.zeros_like(flat)
        for i, p in enumerate(ps):
            o, n = offs[id(p)], p.numel()
            for name, buf in (("exp_avg", self._m), ("exp_avg_sq", self._v)):
                t = state[i][name]
                if tuple(t.shape) != tuple(p.shape):
                    raise ValueError(f"FusedAdamW.load_state_dict: {name} of parameter {i} has shape {tuple(t.shape)}, "
                                     f"expected {tuple(p.shape)}")
                buf[o:o + n].view(p.shape).copy_(t.detach().to(device=flat.device, dtype=torch.float32))


def grouped_parameters(model, weight_decay, no_decay=("bias", "layer_norm.weight", "mlm_ln.weight", "embeddings.weight")):
    """The two parameter groups of training/train_muse.py:425-437 - weight decay on everything whose name carries none of the
    `no_decay` fragments, 0.0 on the rest - ready for `muse.FusedAdamW(grouped_parameters(model, wd), lr=..., weight_decay=wd)`."""
    named = list(model.named_parameters())
    return [
        {"params": [p for n, p in named if not any(nd in n for nd in no_decay)], "weight_decay": weight_decay},
        {"params": [p for n, p in named if any(nd in n for nd in no_decay)], "weight_decay": 0.0},
    ]


class GradReducer:
    """Data-parallel gradient averaging.  muse.MaskGitTransformer: the flat gradient buffer is all-reduced in large
    contiguous buckets (default 64 MiB; xGMI rings are per-link bound, so few large messages) as soon as backward has
    finished writing them, on a side stream, while backward keeps computing earlier layers.

    The mean is taken by the collective itself (`ReduceOp.AVG`, one pass) on RCCL; on gloo (CPU tests), which has no
    AVG, the bucket is pre-scaled.  `grad_dtype=torch.bfloat16` sends each bucket as bf16 (half the xGMI bytes:
    SURVEY.md section 5 sizes config B at 1.5 ms instead of 3.0 ms per step) and expands the reduced bucket back to f32.

    torch.distributed's "nccl" backend is RCCL on ROCm; on CPU tensors (gloo) the same logic runs synchronously, which
    is how the world_size-2 tests cover it."""

    def __init__(self, model, process_group=None, bucket_bytes: int = 64 << 20, broadcast_params: bool = True,
                 grad_dtype=torch.float32):
        if not dist.is_initialized():
            raise MuseHipError("GradReducer needs an initialised torch.distributed process group")
        if grad_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("grad_dtype must be torch.float32 or torch.bfloat16")
        self.model = model
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        # gloo has no ReduceOp.AVG: pre-scale and SUM (also for GPU tensors: protocol tests of the N > 1 path run two gloo ranks on one GPU)
        self._has_avg = dist.get_backend(process_group) != "gloo"
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.grad_dtype = grad_dtype
        self._hi = self._lo = None
        self._handles = []
        self._stream = None
        self.post_reduce = None   # callable(lo, hi): runs on the reduction stream right after bucket [lo, hi) has been averaged
        self.stats = {"buckets": 0, "bytes": 0}
        # bench.py's `comm` block: a list here makes every flat-buffer bucket leave (start event, end event, payload bytes) on the
        # communication stream and finish() an event at the point where backward's last kernel has been enqueued on the compute
        # stream - per bucket, the share of its all-reduce that ran before that point is what backward hid (bucket_overlap())
        self.trace = None
        self._bwd_end = None
        self._sync = True         # False inside no_sync(): gradient accumulation, no collective
        self._deferred = False    # tape engines: this backward adds to existing gradients -> the SUM is reduced in finish()
        # Models with a flat gradient buffer (MaskGitTransformer) report finished ranges during backward.  Models whose
        # parameters are ordinary tensors (MaskGiTUViT: one autograd node hands every gradient back at once) are reduced in
        # finish(): gradients packed, in reverse parameter order, into the same large buckets.
        self._flat_mode = hasattr(model, "flat_grads") and hasattr(model, "grad_ready_hook")
        if self._flat_mode:
            model.direct_grad = True
            model.grad_ready_hook = self._on_ready
            if broadcast_params:
                dist.broadcast(model.flat_params(), src=0, group=process_group)  # DDP's constructor sync (:305)
                model.mark_weights_changed()
        else:
            self._params = [p for p in model.parameters() if p.requires_grad]
            # the tape engines (MaskGiTUViT, text-conditioned MaskGitTransformer) report finished gradients from inside their
            # hand-written backward: buckets are packed and all-reduced on the communication stream while backward computes the
            # earlier blocks (DDP's overlap, training/train_muse.py:753-759); finish() then has nothing left to do.  Models without
            # the hook are reduced after backward.
            self._pending, self._pending_n, self._in_backward_done = [], 0, False
            if hasattr(model, "grad_tensors_hook"):
                model.grad_tensors_hook = self._on_tensors
            if broadcast_params:
                with torch.no_grad():
                    for bucket in self._buckets([p.data for p in self._params]):
                        flat = torch.cat([t.reshape(-1) for t in bucket])
                        dist.broadcast(flat, src=0, group=process_group)
                        o = 0
                        for t in bucket:
                            t.copy_(flat[o:o + t.numel()].view_as(t))
                            o += t.numel()
                if hasattr(model, "mark_weights_changed"):
                    model.mark_weights_changed()

    def no_sync(self):
        """Gradient accumulation: backward inside this context keeps the gradients local (no collective, no per-bucket callback), like
        `DistributedDataParallel.no_sync()` - what `accelerator.accumulate(model)` enters on all but the last micro-batch of
        `gradient_accumulation_steps` (training/train_muse.py:734; 2 in ten of the reference's configurations, cc12m_uvit_clip.yaml among
        them).  The first synchronising backward after it averages the ACCUMULATED gradients: the flat buffer already holds the sum when
        its ranges are reported; the tape engines hand over only that micro-batch's gradients, so their in-backward buckets are skipped
        for this step and finish() reduces the `.grad` tensors autograd has summed into.

            for i, batch in enumerate(micro_batches):
                with (reducer.no_sync() if i + 1 < len(micro_batches) else contextlib.nullcontext()):
                    loss(batch).backward()
            reducer.finish(); optimizer.step(); optimizer.zero_grad(set_to_none=True)"""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            prev, self._sync = self._sync, False
            try:
                yield self
            finally:
                self._sync = prev
        return ctx()

    def _buckets(self, tensors):
        """consecutive tensors grouped into buckets of >= bucket_elems elements (the last one may be smaller)"""
        out, cur, n = [], [], 0
        for t in tensors:
            cur.append(t)
            n += t.numel()
            if n >= self.bucket_elems:
                out.append(cur)
                cur, n = [], 0
        if cur:
            out.append(cur)
        return out

    def _on_tensors(self, tensors, final, side_stream=None):
        """called from inside backward (TapeOps._report_grads) with gradient tensors that are complete on the model's compute stream
        (and its weight-gradient stream `side_stream`).  They are collected into buckets of >= bucket_elems elements; a full bucket is
        packed, all-reduced and unpacked IN PLACE on the communication stream, behind both compute streams.  `final`: flush the
        rest and make the compute stream wait for every bucket - autograd receives averaged gradients."""
        if not self._sync:
            return                                   # no_sync(): this micro-batch's gradients stay local (autograd accumulates them)
        if not self._pending and not self._deferred and any(p.grad is not None for p in self._params):
            self._deferred = True                    # gradients of earlier micro-batches exist: reduce the sums after backward (finish)
            # ... also when the earlier backward of this step was itself reduced in-backward (two synchronising backwards before one
            # finish()): its averaged gradients are identical on every rank, so averaging the sums again yields avg(g1) + avg(g2),
            # what DDP's per-backward all-reduce produces.  Without this, finish() took the "already reduced" early return and every
            # rank kept avg(g1) + its LOCAL g2.
            self._in_backward_done = False
        if self._deferred:
            return
        for t in tensors:
            self._pending.append(t)
            self._pending_n += t.numel()
        if self._pending and (self._pending_n >= self.bucket_elems or final):
            self._reduce_list(self._pending, side_stream)
            self._pending, self._pending_n = [], 0
        if final:
            for h in self._handles:
                h.wait()
            self._handles = []
            if self._stream is not None and self._live:
                torch.cuda.current_stream().wait_stream(self._stream)
                self._live = False
            self._in_backward_done = True

    _live = False

    def _reduce_list(self, bucket, side_stream=None):
        on_gpu = bucket[0].is_cuda
        self.stats["buckets"] += 1
        self.stats["bytes"] += sum(t.numel() for t in bucket) * (2 if self.grad_dtype == torch.bfloat16 else 4)
        if not on_gpu:
            flat = torch.cat([g.reshape(-1) for g in bucket])
            self._reduce(flat, False)
            o = 0
            for g in bucket:
                g.copy_(flat[o:o + g.numel()].view_as(g))
                o += g.numel()
            return
        if self._stream is None:
            self._stream = torch.cuda.Stream(priority=-1)
        cur = torch.cuda.current_stream()
        self._stream.wait_stream(cur)
        if side_stream is not None:
            self._stream.wait_stream(side_stream)
        with torch.cuda.stream(self._stream):
            flat = torch.cat([g.reshape(-1) for g in bucket])        # pack on the communication stream: no work on the compute streams
            h = self._reduce(flat, True)
            if h is not None:
                h.wait()                                             # stream-side: the unpack below follows the collective
            o = 0
            for g in bucket:
                g.copy_(flat[o:o + g.numel()].view_as(g))
                o += g.numel()
        for g in bucket:
            g.record_stream(self._stream)
        self._live = True

    def _finish_tensor_list(self):
        self._deferred = False
        if not self._sync:
            return
        if self._in_backward_done:           # every gradient was reduced from inside backward (_on_tensors)
            self._in_backward_done = False
            return
        grads = [p.grad for p in reversed(self._params) if p.grad is not None]
        if not grads:
            return
        on_gpu = grads[0].is_cuda
        if on_gpu and self._stream is None:
            self._stream = torch.cuda.Stream(priority=-1)
        packed = []
        for bucket in self._buckets(grads):
            flat = torch.cat([g.reshape(-1) for g in bucket])        # pack (one pass); the collective then moves one large message
            self.stats["buckets"] += 1
            self.stats["bytes"] += flat.numel() * (2 if self.grad_dtype == torch.bfloat16 else 4)
            if on_gpu:
                self._stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self._stream):
                    h = self._reduce(flat, True)
                    if h is not None:
                        self._handles.append(h)
                flat.record_stream(self._stream)
            else:
                self._reduce(flat, False)
            packed.append((bucket, flat))
        for h in self._handles:
            h.wait()
        self._handles = []
        if on_gpu:
            torch.cuda.current_stream().wait_stream(self._stream)
        for bucket, flat in packed:                                   # unpack into the .grad tensors the optimizer reads
            o = 0
            for g in bucket:
                g.copy_(flat[o:o + g.numel()].view_as(g))
                o += g.numel()

    def _reduce(self, g, on_gpu):
        """mean over ranks of one bucket (in place)"""
        if self.grad_dtype == torch.bfloat16:
            c = ops.cast_to_bf16(g) if on_gpu else g.to(torch.bfloat16)
            if on_gpu and self._has_avg:
                dist.all_reduce(c, op=dist.ReduceOp.AVG, group=self.pg)
                ops.cast_to_f32(c, g)
            else:   # gloo: no AVG and no bf16 arithmetic guarantee: reduce the bf16-rounded values in f32
                c = c.to(torch.float32).mul_(1.0 / self.world)
                dist.all_reduce(c, op=dist.ReduceOp.SUM, group=self.pg)
                g.copy_(c)
            return None
        if on_gpu and self._has_avg:
            return dist.all_reduce(g, op=dist.ReduceOp.AVG, group=self.pg, async_op=True)
        g.mul_(1.0 / self.world)
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.pg)
        return None

    def _launch(self, lo, hi):
        g = self.model.flat_grads()[lo:hi]
        self.stats["buckets"] += 1                     # (bench.py's `comm` block: buckets and payload bytes per step)
        self.stats["bytes"] += (hi - lo) * (2 if self.grad_dtype == torch.bfloat16 else 4)
        if g.is_cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream(priority=-1)
            self._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                ev = None
                if self.trace is not None:
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[0].record()
                h = self._reduce(g, True)
                if ev is not None:
                    if h is not None:
                        h.wait()
                        h = None
                    ev[1].record()
                    self.trace.append((ev[0], ev[1], (hi - lo) * (2 if self.grad_dtype == torch.bfloat16 else 4)))
                if self.post_reduce is not None:
                    if h is not None:
                        h.wait()          # (a stream-side wait: the reduction stream continues behind the collective; the host does not block)
                    self.post_reduce(lo, hi)
                elif h is not None:
                    self._handles.append(h)
        else:
            self._reduce(g, False)
            if self.post_reduce is not None:
                self.post_reduce(lo, hi)

    def _on_ready(self, begin: int, end: int):
        """backward reports finished [begin, end) ranges of the flat grad buffer, from the end of the buffer downward"""
        if not self._sync:
            return                                   # no_sync(): the buffer keeps accumulating, nothing is sent
        if self._hi is None:
            self._hi, self._lo = end, begin
        elif end == self._lo:
            self._lo = begin
        else:  # non-contiguous report: flush what we have and start a new run
            self._launch(self._lo, self._hi)
            self._hi, self._lo = end, begin
        if (self._hi - self._lo) >= self.bucket_elems:
            self._launch(self._lo, self._hi)
            self._hi = self._lo = None

    def finish(self):
        """flush the last bucket and make the compute stream wait for every outstanding all-reduce"""
        if not self._flat_mode:
            return self._finish_tensor_list()
        if not self._sync:
            return
        if self.trace is not None and self.model.flat_grads().is_cuda:
            self._bwd_end = torch.cuda.Event(enable_timing=True)
            self._bwd_end.record()                   # backward's last kernel is on the compute stream; what follows is exposed
        if self._hi is not None:
            self._launch(self._lo, self._hi)
            self._hi = self._lo = None
        for h in self._handles:
            h.wait()
        self._handles = []
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)

    def bucket_overlap(self):
        """after a traced step and a device synchronisation: per bucket {bytes, ms, hidden_fraction} - the share of the bucket's
        all-reduce (as timed on the communication stream) that ran before backward's last kernel was enqueued.  A bucket launched
        in finish() itself (the last, ragged one) is exposed by construction."""
        out = []
        if not self.trace or self._bwd_end is None:
            return out
        for e0, e1, nbytes in self.trace:
            ms = e0.elapsed_time(e1)
            before = e0.elapsed_time(self._bwd_end)          # > 0: the bucket started that long before the end of backward
            hidden = 0.0 if ms <= 0 else max(0.0, min(1.0, before / ms))
            out.append({"bytes": int(nbytes), "ms": round(ms, 3), "hidden_fraction": round(hidden, 3)})
        return out

    def reduce_metrics(self, loss, mask_prob):
        """The two logged scalars of the reference's loop (`accelerator.gather(loss.repeat(bs)).mean()` and
        `accelerator.gather(mask_prob.repeat(bs)).mean()`, train_maskgit_imagenet.py:430-431) as ONE 2-float all-reduce
        instead of two all-gathers of bs and bs^2 floats.  -> (mean loss, mean mask rate) over all ranks."""
        v = torch.stack([loss.detach().float().reshape(()), mask_prob.detach().float().mean()])
        if v.is_cuda and self._has_avg:
            dist.all_reduce(v, op=dist.ReduceOp.AVG, group=self.pg)
        else:
            v.mul_(1.0 / self.world)
            dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self.pg)
        return v[0], v[1]


class TrainStep:
    """One optimisation step exactly as the reference's loop body strings it together (:405-452):
    VQ-encode -> mask -> forward(loss) -> backward (+ overlapped gradient all-reduce) -> AdamW -> zero_grad.

    `next_pixel_values`: the following batch's images.  The tokenizer is frozen (`vq_model.requires_grad_(False)`, :283), so its
    token ids do not depend on the optimizer state and the NEXT batch can be encoded while this one trains: the encode is
    enqueued on a second HIP stream before this step's forward (the software analogue of the reference's dataloader workers
    running ahead of the loop; the token ids are bit-identical to encoding inline).  The convolution blocks and the
    transformer's GEMM / HBM-bound kernels then share the chip: each fills the CUs and memory cycles the other leaves idle
    (tile seams, short grids).  The next call picks the tokens up when it is given the same tensor as `pixel_values`."""

    def __init__(self, vq_model, model, optimizer, reducer: Optional[GradReducer] = None, label_smoothing: float = 0.0,
                 min_masking_rate: float = 0.0):
        self.vq_model, self.model, self.optimizer, self.reducer = vq_model, model, optimizer, reducer
        self.label_smoothing, self.min_masking_rate = label_smoothing, min_masking_rate
        self._pf = None          # (pixel tensor, tokens, ready event) of the prefetched batch
        self._pf_stream = None
        self.compute_priority = -1   # stream priority of the step while a prefetch is in flight (None: stay on the caller's stream)
        self.optimizer_in_backward = os.environ.get("MUSE_OPT_IN_BACKWARD", "1") != "0"
        self.optimizer_in_reducer = os.environ.get("MUSE_OPT_IN_REDUCER", "1") != "0"
        self._hp_stream, self._in_hp = None, False

    @torch.no_grad()
    def _prefetch(self, pixel_values):
        if not pixel_values.is_cuda:
            raise MuseHipError("TrainStep: next_pixel_values must be on the GPU")
        main = torch.cuda.current_stream(pixel_values.device)
        if self._pf_stream is None or self._pf_stream.device != pixel_values.device:
            # MUSE_PF_PRIORITY: stream priority of the tokenizer's prefetch stream (default 0 = normal; 1 = low where the runtime offers it)
            self._pf_stream = torch.cuda.Stream(device=pixel_values.device, priority=int(os.environ.get("MUSE_PF_PRIORITY", "0")))
        side = self._pf_stream
        side.wait_stream(main)                      # the images (and the previous use of the tokenizer's buffers) are ready
        with torch.cuda.stream(side), ops.conv_persistent(False):
            # (beside the step the launch-per-tile convolution: a persistent workgroup would hold its CU against the step's stream)
            tokens = self.vq_model.get_code(pixel_values)
            ev = torch.cuda.Event()
            ev.record(side)
        pixel_values.record_stream(side)
        self._pf = (pixel_values, tokens, ev)

    def __call__(self, pixel_values, class_ids, timesteps=None, noise=None, image_tokens=None, next_pixel_values=None):
        """image_tokens [B, S] int64: pre-encoded VQ tokens (muse.pre_encode; the reference's scripts/pre_encode.py regime) -
        the tokenizer is then skipped and pixel_values may be None"""
        # With a tokenizer pass of the next batch in flight, the train step itself runs on a HIGH-priority stream: the hardware
        # then serves its (latency-critical, serially dependent) kernels first and fits the tokenizer's blocks into what is left
        # (62.3 -> 61.1 ms per step on MI355X; the reverse, a high-priority tokenizer stream, costs 1.5 ms).  The caller's stream is
        # joined on entry and on return, so nothing changes for code around the step.
        if next_pixel_values is not None and self.compute_priority is not None and class_ids.is_cuda and not self._in_hp:
            dev = class_ids.device
            if self._hp_stream is None or self._hp_stream.device != dev:
                self._hp_stream = torch.cuda.Stream(device=dev, priority=self.compute_priority)
            cur = torch.cuda.current_stream(dev)
            self._hp_stream.wait_stream(cur)
            self._in_hp = True
            try:
                with torch.cuda.stream(self._hp_stream):
                    loss, mask_prob = self.__call__(pixel_values, class_ids, timesteps, noise, image_tokens, next_pixel_values)
            finally:
                self._in_hp = False
            cur.wait_stream(self._hp_stream)
            loss.record_stream(cur)
            mask_prob.record_stream(cur)
            return loss, mask_prob
        if image_tokens is None and self._pf is not None and self._pf[0] is pixel_values:
            _, image_tokens, ev = self._pf
            main = torch.cuda.current_stream(image_tokens.device)
            main.wait_event(ev)
            image_tokens.record_stream(main)
        elif self._pf is not None and image_tokens is None:
            # the prefetch is keyed by tensor IDENTITY: a different tensor object here means this batch is tokenised a second time
            warnings.warn("TrainStep: the prefetched batch is discarded - `pixel_values` is not the tensor object that was passed as "
                          "`next_pixel_values` to the previous call; this step encodes its images again", RuntimeWarning, stacklevel=2)
        self._pf = None
        if next_pixel_values is not None:
            if image_tokens is None:   # first step of a run: this batch inline, then the next one on the side stream
                image_tokens = self.vq_model.get_code(pixel_values)
            self._prefetch(next_pixel_values)
        input_ids, labels, _, mask_prob = prepare_inputs_and_labels(
            self.vq_model, pixel_values, class_ids, self.model.config.mask_token_id, self.min_masking_rate, timesteps, noise,
            image_tokens=image_tokens,
            codebook_size=None if self.vq_model is not None else self.model.config.codebook_size)
        _, loss = self.model(input_ids=input_ids, labels=labels, label_smoothing=self.label_smoothing)
        # AdamW inside backward (FusedAdamW.begin_step_in_backward): only when nothing sits between the two - no reducer here
        armed = (self.optimizer_in_backward and self.reducer is None and isinstance(self.optimizer, FusedAdamW)
                 and getattr(self.model, "wgrad_stream", False) and hasattr(self.model, "grad_ready_hook")
                 and self.model._resolve_cd() == torch.bfloat16 and self.optimizer.begin_step_in_backward(self.model))
        # ... and with a reducer: AdamW on each bucket right behind its all-reduce, on the reducer's stream
        armed_r = (not armed and self.optimizer_in_reducer and self.reducer is not None and getattr(self.reducer, "_flat_mode", False)
                   and self.reducer.model is self.model and isinstance(self.optimizer, FusedAdamW) and loss.is_cuda
                   and self.model._resolve_cd() == torch.bfloat16 and self.optimizer.begin_step_in_reducer(self.model, self.reducer))
        failed = True
        try:
            loss.backward()
            if self.reducer is not None:
                self.reducer.finish()
            failed = False
        finally:
            if armed:
                self.optimizer.end_step_in_backward(self.model, failed=failed)
            if armed_r:
                self.optimizer.end_step_in_reducer(self.reducer, failed=failed)
        self.optimizer.step()
        self.optimizer.zero_grad(set_to_none=True)
        return loss.detach(), mask_prob
