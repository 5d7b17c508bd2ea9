"""CPU oracle for the weight average on the training step (reference muse/modeling_ema.py:90-137; training/train_muse.py:779-780).
TEST INFRASTRUCTURE ONLY: imported by tests/ and nothing under open-muse_amd/.

numpy restatement of `EMAModel.get_decay` and of one `EMAModel.step`:  s <- s - f32(1 - decay) * (s - p)  evaluated the way the
reference's tensor expression evaluates it - subtract, multiply by the f32-rounded scalar, subtract, each rounded to f32 - and a plain
copy for parameters that do not require grad (:134-135).

Parity status: **pinned** - tests/golden/ema_tiny.npz holds what the real reference class produced over 14 steps of two schedules
(make_golden.py::golden_ema); tests/test_oracle_golden.py replays it bit for bit.
"""
from __future__ import annotations

import numpy as np


def get_decay(optimization_step, decay=0.9999, min_decay=0.0, update_after_step=0, use_ema_warmup=False, inv_gamma=1.0, power=2 / 3):
    """:90-107"""
    step = max(0, optimization_step - update_after_step - 1)
    if step <= 0:
        return 0.0
    if use_ema_warmup:
        cur = 1 - (1 + step / inv_gamma) ** -power
    else:
        cur = (1 + step) / (10 + step)
    return max(min(cur, decay), min_decay)


def ema_update(shadow, params, requires_grad, decay):
    """one update of every shadow tensor (:128-137); f32 arrays in, new f32 arrays out"""
    omd = np.float32(1 - decay)
    out = []
    for s, p, rg in zip(shadow, params, requires_grad):
        if rg:
            t = (s - p).astype(np.float32)
            u = (omd * t).astype(np.float32)
            out.append((s - u).astype(np.float32))
        else:
            out.append(p.astype(np.float32).copy())
    return out


class Schedule:
    """the counters of EMAModel.step (:109-127): which calls update, and with which decay"""

    def __init__(self, update_every=1, **kw):
        self.kw, self.update_every, self.optimization_step = kw, update_every, 0

    def next(self):
        """-> decay of this call, or None when the call is skipped"""
        self.optimization_step += 1
        if (self.optimization_step - 1) % self.update_every != 0:
            return None
        return get_decay(self.optimization_step, **self.kw)
