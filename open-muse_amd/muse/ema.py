"""muse.EMAModel for MI355X: the running average of the weights that the reference's text-to-image loop keeps when `training.use_ema`
is set (configs/research_run_512*.yaml; training/train_muse.py:367-386 builds it, :779-780 advances it right behind every optimizer
step, :856-936 swaps it in for validation / saving).  Interface, state-dict keys and the decay schedule follow the reference class
(muse/modeling_ema.py:9-244) so the training script needs no change.

What runs on the step is `step()`.  The reference evaluates `s.sub_(one_minus_decay * (s - p))` tensor by tensor (:131-137: three
elementwise kernels per tracked tensor, ~1500 launches for the 500 tensors of MaskGiTUViT); here every (shadow, parameter) pair sits in
a device-side pointer table and ONE launch of `muse_ema_multi` updates all of them, keeping the expression's three f32 roundings, so the
shadow weights come out bit-identical to the reference's.  `step()` has no CPU path (the shadow must have been moved with
`ema.to(device)`, as train_muse.py:536-537 does).  `copy_to` / `restore` write through `Tensor.copy_` under `no_grad`: that bumps the
parameters' version counters, which is what the models' cached bf16 weight copies are validated against.
"""
from __future__ import annotations

import copy

import torch

from . import ops
from ._hip import MuseHipError

# scalar entries of the state dict: (key, accepted types, error text of the reference's load_state_dict :207-236)
_SCALAR_STATE = (
    ("min_decay", (float,), "Invalid min_decay"),
    ("optimization_step", (int,), "Invalid optimization_step"),
    ("update_after_step", (int,), "Invalid update_after_step"),
    ("use_ema_warmup", (bool,), "Invalid use_ema_warmup"),
    ("inv_gamma", (float, int), "Invalid inv_gamma"),
    ("power", (float, int), "Invalid power"),
)


class EMAModel:
    def __init__(self, parameters, decay=0.9999, min_decay=0.0, update_after_step=0, update_every=1, use_ema_warmup=False,
                 inv_gamma=1.0, power=2 / 3, model_cls=None, model_config=None):
        self.shadow_params = [p.detach().clone() for p in parameters]
        self.temp_stored_params = None
        self.decay, self.min_decay = decay, min_decay
        self.update_after_step, self.update_every = update_after_step, update_every
        self.use_ema_warmup, self.inv_gamma, self.power = use_ema_warmup, inv_gamma, power
        self.optimization_step = 0
        self.cur_decay_value = None                     # the decay the last step() applied
        self.model_cls, self.model_config = model_cls, model_config
        self._pairs = None                              # cached device table of the (shadow, parameter) pairs

    # ---- schedule ------------------------------------------------------------------------------------------------------------------------
    def get_decay(self, optimization_step):
        """reference :90-107: 0 until `update_after_step` has passed, then (1 + n) / (10 + n) - or the inv_gamma / power warmup -
        clamped into [min_decay, decay]"""
        n = max(0, optimization_step - self.update_after_step - 1)
        if n <= 0:
            return 0.0
        value = 1 - (1 + n / self.inv_gamma) ** -self.power if self.use_ema_warmup else (1 + n) / (10 + n)
        return max(min(value, self.decay), self.min_decay)

    # ---- the per-step update ---------------------------------------------------------------------------------------------------------------
    def _device_pairs(self, params):
        key = tuple((s.data_ptr(), p.data_ptr(), p.numel(), p.requires_grad) for s, p in zip(self.shadow_params, params))
        if self._pairs is not None and self._pairs["key"] == key:
            return self._pairs
        entries, first = [], [0]
        for s, p in zip(self.shadow_params, params):
            if not (s.is_cuda and p.is_cuda):
                raise MuseHipError("EMAModel.step (MI355X build) has no CPU path: move the model and the EMA (`ema.to(device)`) to the GPU")
            if s.dtype != torch.float32 or p.dtype != torch.float32 or s.shape != p.shape or not (s.is_contiguous() and p.is_contiguous()):
                raise MuseHipError("EMAModel.step: shadow and parameter must be contiguous f32 tensors of one shape")
            if p.numel():
                entries += [s.data_ptr(), p.data_ptr(), p.numel(), 0 if p.requires_grad else 1]      # mode 1: frozen parameter, plain copy (:134-135)
                first.append(first[-1] + (p.numel() + 4095) // 4096)
        dev = self.shadow_params[0].device
        self._pairs = dict(key=key, n=len(first) - 1, chunks=first[-1],
                           table=torch.tensor(entries, dtype=torch.int64, device=dev) if entries else None,
                           first=torch.tensor(first, dtype=torch.int32, device=dev))
        return self._pairs

    @torch.no_grad()
    def step(self, parameters):
        params = list(parameters)
        self.optimization_step += 1
        if (self.optimization_step - 1) % self.update_every:
            return
        self.cur_decay_value = self.get_decay(self.optimization_step)
        if len(params) != len(self.shadow_params):
            raise MuseHipError(f"EMAModel.step: {len(params)} parameters for {len(self.shadow_params)} shadow tensors")
        if not params:
            return
        t = self._device_pairs(params)
        if t["n"]:
            ops.ema_multi(t["table"], t["first"], t["n"], t["chunks"], 1 - self.cur_decay_value)

    # ---- swapping the average in and out (validation / saving, train_muse.py:856-936) -------------------------------------------------------
    def copy_to(self, parameters):
        with torch.no_grad():
            for shadow, p in zip(self.shadow_params, list(parameters)):
                p.copy_(shadow.to(p.device))

    def store(self, parameters):
        self.temp_stored_params = [p.detach().cpu().clone() for p in parameters]

    def restore(self, parameters):
        if self.temp_stored_params is None:
            raise RuntimeError("This ExponentialMovingAverage has no `store()`ed weights to `restore()`")
        with torch.no_grad():
            for kept, p in zip(self.temp_stored_params, list(parameters)):
                p.copy_(kept)
        self.temp_stored_params = None

    def to(self, device=None, dtype=None):
        self.shadow_params = [s.to(device=device, dtype=dtype) if s.is_floating_point() else s.to(device=device) for s in self.shadow_params]
        self._pairs = None

    # ---- persistence ---------------------------------------------------------------------------------------------------------------------
    def state_dict(self):
        """what accelerate checkpoints (reference :163-177)"""
        out = {"decay": self.decay}
        out.update({key: getattr(self, key) for key, _, _ in _SCALAR_STATE})
        out["shadow_params"] = self.shadow_params
        return out

    def load_state_dict(self, state_dict):
        state = copy.deepcopy(state_dict)
        self.decay = state.get("decay", self.decay)
        if not 0.0 <= self.decay <= 1.0:
            raise ValueError("Decay must be between 0 and 1")
        for key, kinds, complaint in _SCALAR_STATE:
            setattr(self, key, state.get(key, getattr(self, key)))
            if not isinstance(getattr(self, key), kinds):
                raise ValueError(complaint)
        shadow = state.get("shadow_params")
        if shadow is not None:
            if not isinstance(shadow, list):
                raise ValueError("shadow_params must be a list")
            if any(not isinstance(s, torch.Tensor) for s in shadow):
                raise ValueError("shadow_params must all be Tensors")
            self.shadow_params, self._pairs = shadow, None

    @classmethod
    def from_pretrained(cls, path, model_cls):
        """reference :64-73.  The average comes back (it was saved as the model's weights); the scalars save_pretrained wrote into
        config.json do NOT: `load_config(..., return_unused_kwargs=True)` returns the unused *call* kwargs - none - exactly as in the
        reference, whose resumed runs therefore restart the schedule at optimization_step 0 (train_muse.py:379-380).  Kept as is."""
        _, extra = model_cls.load_config(path, return_unused_kwargs=True)
        model = model_cls.from_pretrained(path)
        ema = cls(model.parameters(), model_cls=model_cls, model_config=model.config)
        ema.load_state_dict(extra)
        return ema

    def save_pretrained(self, path):
        for need, what in ((self.model_cls, "model_cls"), (self.model_config, "model_config")):
            if need is None:
                raise ValueError(f"`save_pretrained` can only be used if `{what}` was defined at __init__.")
        model = self.model_cls.from_config(self.model_config)
        scalars = {k: v for k, v in self.state_dict().items() if k != "shadow_params"}
        model.register_to_config(**scalars)
        self.copy_to(model.parameters())
        model.save_pretrained(path)
