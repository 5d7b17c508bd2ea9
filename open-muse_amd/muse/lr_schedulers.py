"""muse.lr_schedulers - the learning-rate schedules the reference's training scripts ask for by name
(`from muse.lr_schedulers import get_scheduler`, training/train_muse.py:61,512-517; training/train_maskgit_imagenet.py:295; reference
muse/lr_schedulers.py:29-283).  Host-side scalars only: each schedule is a multiplier of the optimizer's base learning rate as a
function of the step count, handed to torch's LambdaLR - which drives `muse.FusedAdamW.param_groups` like any optimizer's.

Pinned against the real module: tests/golden/lr_schedules.npz holds the learning rates the reference's get_scheduler produced for every
schedule name over a whole run (tests/test_surface.py::test_lr_schedules_vs_reference_golden).
"""
from __future__ import annotations

import math
from enum import Enum
from typing import Optional, Union

from torch.optim import Optimizer
from torch.optim.lr_scheduler import LambdaLR


class SchedulerType(Enum):
    LINEAR = "linear"
    COSINE = "cosine"
    COSINE_WITH_RESTARTS = "cosine_with_restarts"
    POLYNOMIAL = "polynomial"
    CONSTANT = "constant"
    CONSTANT_WITH_WARMUP = "constant_with_warmup"


def _ramp(step: int, warmup: int) -> float:
    """0 -> 1 over the warm-up steps"""
    return float(step) / float(max(1, warmup))


def _progress(step: int, warmup: int, total: int) -> float:
    """fraction of the post-warm-up part of the run that lies behind `step`"""
    return float(step - warmup) / float(max(1, total - warmup))


def get_constant_schedule(optimizer: Optimizer, last_epoch: int = -1):
    return LambdaLR(optimizer, lambda _: 1, last_epoch=last_epoch)


def get_constant_schedule_with_warmup(optimizer: Optimizer, num_warmup_steps: int, last_epoch: int = -1):
    def factor(step: int):
        return float(step) / float(max(1.0, num_warmup_steps)) if step < num_warmup_steps else 1.0
    return LambdaLR(optimizer, factor, last_epoch=last_epoch)


def get_linear_schedule_with_warmup(optimizer, num_warmup_steps, num_training_steps, last_epoch=-1):
    def factor(step: int):
        if step < num_warmup_steps:
            return _ramp(step, num_warmup_steps)
        return max(0.0, float(num_training_steps - step) / float(max(1, num_training_steps - num_warmup_steps)))
    return LambdaLR(optimizer, factor, last_epoch)


def get_cosine_schedule_with_warmup(optimizer: Optimizer, num_warmup_steps: int, num_training_steps: int, num_cycles: float = 0.5,
                                    last_epoch: int = -1):
    def factor(step: int):
        if step < num_warmup_steps:
            return _ramp(step, num_warmup_steps)
        p = _progress(step, num_warmup_steps, num_training_steps)
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * p)))
    return LambdaLR(optimizer, factor, last_epoch)


def get_cosine_with_hard_restarts_schedule_with_warmup(optimizer: Optimizer, num_warmup_steps: int, num_training_steps: int,
                                                       num_cycles: int = 1, last_epoch: int = -1):
    def factor(step: int):
        if step < num_warmup_steps:
            return _ramp(step, num_warmup_steps)
        p = _progress(step, num_warmup_steps, num_training_steps)
        if p >= 1.0:
            return 0.0
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * ((float(num_cycles) * p) % 1.0))))
    return LambdaLR(optimizer, factor, last_epoch)


def get_polynomial_decay_schedule_with_warmup(optimizer, num_warmup_steps, num_training_steps, lr_end=1e-7, power=1.0, last_epoch=-1):
    lr_init = optimizer.defaults["lr"]
    if not (lr_init > lr_end):
        raise ValueError(f"lr_end ({lr_end}) must be be smaller than initial lr ({lr_init})")

    def factor(step: int):
        if step < num_warmup_steps:
            return _ramp(step, num_warmup_steps)
        if step > num_training_steps:
            return lr_end / lr_init          # (a multiplier: LambdaLR scales lr_init by it)
        remaining = 1 - (step - num_warmup_steps) / (num_training_steps - num_warmup_steps)
        return ((lr_init - lr_end) * remaining ** power + lr_end) / lr_init
    return LambdaLR(optimizer, factor, last_epoch)


TYPE_TO_SCHEDULER_FUNCTION = {
    SchedulerType.LINEAR: get_linear_schedule_with_warmup,
    SchedulerType.COSINE: get_cosine_schedule_with_warmup,
    SchedulerType.COSINE_WITH_RESTARTS: get_cosine_with_hard_restarts_schedule_with_warmup,
    SchedulerType.POLYNOMIAL: get_polynomial_decay_schedule_with_warmup,
    SchedulerType.CONSTANT: get_constant_schedule,
    SchedulerType.CONSTANT_WITH_WARMUP: get_constant_schedule_with_warmup,
}


def get_scheduler(name: Union[str, SchedulerType], optimizer: Optimizer, num_warmup_steps: Optional[int] = None,
                  num_training_steps: Optional[int] = None, num_cycles: int = 1, power: float = 1.0):
    """the schedule called `name` (reference :237-283): which arguments each one needs, and the errors for missing ones, as there"""
    name = SchedulerType(name)
    make = TYPE_TO_SCHEDULER_FUNCTION[name]
    if name == SchedulerType.CONSTANT:
        return make(optimizer)
    if num_warmup_steps is None:
        raise ValueError(f"{name} requires `num_warmup_steps`, please provide that argument.")
    if name == SchedulerType.CONSTANT_WITH_WARMUP:
        return make(optimizer, num_warmup_steps=num_warmup_steps)
    if num_training_steps is None:
        raise ValueError(f"{name} requires `num_training_steps`, please provide that argument.")
    extra = {}
    if name == SchedulerType.COSINE_WITH_RESTARTS:
        extra["num_cycles"] = num_cycles
    elif name == SchedulerType.POLYNOMIAL:
        extra["power"] = power
    return make(optimizer, num_warmup_steps=num_warmup_steps, num_training_steps=num_training_steps, **extra)
