"""Train-step pieces of the hot path that live in the reference's *training script* rather than in its package:

  * prepare_inputs_and_labels  — training/train_maskgit_imagenet.py:357-394 (VQ encode + cosine-schedule mask sampling)
  * FusedAdamW                 — the `fused_adamw` optimizer choice (:242-261, apex FusedAdam adam_w_mode), one HIP
                                 launch over the model's flat parameter buffer
  * GradReducer                — replaces accelerate/DDP (:152-158, :305, :433): bucketed all-reduce of the flat
                                 gradient buffer over RCCL (xGMI) on a side stream, overlapped with backward
  * TrainStep                  — the loop body (:405-452) strung together for benchmarks / smoke tests
"""
from __future__ import annotations

import weakref
from typing import Optional

import torch
import torch.distributed as dist

from . import ops
from ._hip import MuseHipError


@torch.no_grad()
def prepare_inputs_and_labels(vq_model, pixel_values, class_ids, mask_id, min_masking_rate: float = 0.0, timesteps=None,
                              noise=None, generator=None, image_tokens=None):
    """-> (input_ids [B,S+1], labels [B,S+1], soft_targets=None, mask_prob [B]).

    `timesteps` [B] and `noise` [B,S] are the two torch.rand draws of the reference (:375, :381); pass them in for
    bit-reproducible masks (parity tests), otherwise they are drawn on the GPU."""
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
                                                   int(vq_model.num_embeddings), float(min_masking_rate))
    return input_ids, labels, None, mask_prob


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

    Drop-in for `optimizer_cls(model.parameters(), lr=..., betas=..., weight_decay=..., eps=...)`."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        params = list(params)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 1:
            raise MuseHipError("FusedAdamW supports a single parameter group (the reference uses one: :257-263)")
        owner = _owner_of(self.param_groups[0]["params"])
        self._model = weakref.ref(owner) if owner is not None else None
        self._m = self._v = None
        self._step = 0

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        grp = self.param_groups[0]
        if self._model is None:
            return self._step_per_tensor(grp, loss)
        model = self._model()
        flat = model.flat_params()
        if any(p.grad is None for p in grp["params"]):
            return loss  # nothing to do before the first backward (torch skips None grads)
        g = model.flat_grads()
        if self._m is None or self._m.device != flat.device:
            self._m = torch.zeros_like(flat)
            self._v = torch.zeros_like(flat)
        self._step += 1
        lr = float(grp["lr"])
        shadow = model._flat_c if model._flat_c is not None and model._flat_c.device == flat.device else None
        ops.adamw_flat(flat, g, self._m, self._v, shadow, lr, grp["betas"][0], grp["betas"][1], grp["eps"],
                       grp["weight_decay"], self._step)
        model._shadow_fresh = shadow is not None
        model._shadow_version += 1   # transposed weight copies (dgrad) are rebuilt from the refreshed shadow
        return loss

    def _step_per_tensor(self, grp, loss):
        """parameters that are ordinary (contiguous f32) tensors: muse_adamw_flat once per tensor.  torch.optim.AdamW
        semantics: a parameter without a gradient is skipped and keeps its own state; the step count is shared (all
        parameters of these models receive a gradient every step)."""
        params = [p for p in grp["params"] if p.grad is not None]
        if not params:
            return loss
        if self._m is None:
            self._m, self._v = {}, {}
        self._step += 1
        for p in params:
            if p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                raise MuseHipError("FusedAdamW: parameters and gradients must be contiguous float32 tensors")
            k = id(p)
            if k not in self._m or self._m[k].device != p.device:
                self._m[k] = torch.zeros_like(p)
                self._v[k] = torch.zeros_like(p)
            ops.adamw_flat(p.data, p.grad, self._m[k], self._v[k], None, float(grp["lr"]), grp["betas"][0], grp["betas"][1],
                           grp["eps"], grp["weight_decay"], self._step)
        return loss

    def state_dict(self):
        groups = [{k: v for k, v in self.param_groups[0].items() if k != "params"}]
        if self._model is None:   # per-tensor mode: moments in parameter order (None for a parameter that never had a gradient)
            ps = self.param_groups[0]["params"]
            m, v = self._m or {}, self._v or {}
            return {"step": self._step, "exp_avg": [m.get(id(p)) for p in ps], "exp_avg_sq": [v.get(id(p)) for p in ps],
                    "param_groups": groups}
        return {"step": self._step, "exp_avg": self._m, "exp_avg_sq": self._v, "param_groups": groups}

    def load_state_dict(self, sd):
        self._step = int(sd["step"])
        if self._model is None:
            ps = self.param_groups[0]["params"]
            self._m = {id(p): t for p, t in zip(ps, sd["exp_avg"]) if t is not None}
            self._v = {id(p): t for p, t in zip(ps, sd["exp_avg_sq"]) if t is not None}
        else:
            self._m = sd["exp_avg"]
            self._v = sd["exp_avg_sq"]
        self.param_groups[0].update(sd["param_groups"][0])


class GradReducer:
    """Data-parallel gradient averaging for a muse.MaskGitTransformer: the flat gradient buffer is all-reduced in large
    contiguous buckets (default 64 MiB; xGMI rings are per-link bound, so few large messages) as soon as backward has
    finished writing them, on a side stream, while backward keeps computing earlier layers.

    torch.distributed's "nccl" backend is RCCL on ROCm; on CPU tensors (gloo) the same logic runs synchronously, which
    is how the world_size-2 tests cover it."""

    def __init__(self, model, process_group=None, bucket_bytes: int = 64 << 20, broadcast_params: bool = True):
        if not dist.is_initialized():
            raise MuseHipError("GradReducer needs an initialised torch.distributed process group")
        self.model = model
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.bucket_elems = max(1, bucket_bytes // 4)
        self._hi = self._lo = None
        self._handles = []
        self._stream = None
        model.direct_grad = True
        model.grad_ready_hook = self._on_ready
        if broadcast_params:
            dist.broadcast(model.flat_params(), src=0, group=process_group)  # DDP's constructor sync (:305)
            model._shadow_fresh = False

    def _launch(self, lo, hi):
        g = self.model.flat_grads()[lo:hi]
        if g.is_cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream(priority=-1)
            self._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                g.mul_(1.0 / self.world)
                self._handles.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        else:
            g.mul_(1.0 / self.world)
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.pg)

    def _on_ready(self, begin: int, end: int):
        """backward reports finished [begin, end) ranges of the flat grad buffer, from the end of the buffer downward"""
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
        if self._hi is not None:
            self._launch(self._lo, self._hi)
            self._hi = self._lo = None
        for h in self._handles:
            h.wait()
        self._handles = []
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)


class TrainStep:
    """One optimisation step exactly as the reference's loop body strings it together (:405-452):
    VQ-encode -> mask -> forward(loss) -> backward (+ overlapped gradient all-reduce) -> AdamW -> zero_grad."""

    def __init__(self, vq_model, model, optimizer, reducer: Optional[GradReducer] = None, label_smoothing: float = 0.0,
                 min_masking_rate: float = 0.0):
        self.vq_model, self.model, self.optimizer, self.reducer = vq_model, model, optimizer, reducer
        self.label_smoothing, self.min_masking_rate = label_smoothing, min_masking_rate

    def __call__(self, pixel_values, class_ids, timesteps=None, noise=None):
        input_ids, labels, _, mask_prob = prepare_inputs_and_labels(
            self.vq_model, pixel_values, class_ids, self.model.config.mask_token_id, self.min_masking_rate, timesteps, noise)
        _, loss = self.model(input_ids=input_ids, labels=labels, label_smoothing=self.label_smoothing)
        loss.backward()
        if self.reducer is not None:
            self.reducer.finish()
        self.optimizer.step()
        self.optimizer.zero_grad(set_to_none=True)
        return loss.detach(), mask_prob
