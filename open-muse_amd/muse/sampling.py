"""Mask-ratio schedules and the host side of MaskGit parallel decoding.

The schedules are the reference's (muse/sampling.py:38-77: cosine / linear / pow<k> / sigmoid, looked up through
`get_mask_chedule` — the reference's spelling is part of its import surface).  They are host-side scalar maths: a schedule
is evaluated once per decoding step on a 0-dim CPU tensor.  Everything per token — softmax, categorical sampling, the
Gumbel-perturbed confidence, the k-th-smallest threshold and the re-masking (muse/sampling.py:30-35 `mask_by_random_topk`
and its callers) — runs in ONE device call per step (`ops.sample_step` -> libmuse_hip `muse_sample_step`).  The
reference's tensor helpers (`log`, `gumbel_noise`, `gumbel_sample`, `top_k`, `mask_by_random_topk`) stay importable from here
for downstream code (the reference's own modules do `from .sampling import ...`); they are thin torch expressions that run
on whatever device their inputs live on and are NOT on this package's decode path.
"""
from __future__ import annotations

import math
from typing import Callable, Optional, Sequence, Tuple

import torch

_HALF_PI = math.pi * 0.5
_FLOOR = 1e-6   # lower clamp of the non-cosine schedules


# ---- import-surface helpers (muse/sampling.py:8-35); not used by generate2 here -------------------------------------------------
def log(t, eps=1e-20):
    return torch.log(t.clamp(min=eps))


def gumbel_noise(t, generator=None):
    u = torch.zeros_like(t).uniform_(0, 1, generator=generator)
    return -log(-log(u))


def gumbel_sample(t, temperature=1.0, dim=-1, generator=None):
    return (t / max(temperature, 1e-10) + gumbel_noise(t, generator=generator)).argmax(dim=dim)


def top_k(logits, thres=0.9):
    keep = math.ceil((1 - thres) * logits.shape[-1])
    val, ind = logits.topk(keep, dim=-1)
    return torch.full_like(logits, float("-inf")).scatter_(2, ind, val)


def mask_by_random_topk(mask_len, probs, temperature=1.0, generator=None):
    """True where the Gumbel-perturbed log-confidence is below the mask_len-th smallest of its row"""
    confidence = log(probs) + temperature * gumbel_noise(probs, generator=generator)
    cut_off = torch.gather(torch.sort(confidence, dim=-1).values, 1, mask_len.long())
    return confidence < cut_off


def cosine_schedule(t):
    """mask ratio cos(pi/2 * t); also the schedule of the train-time mask sampler (training/train_maskgit_imagenet.py:376)"""
    return torch.cos(t * _HALF_PI)


def linear_schedule(t):
    return torch.clamp(1 - t, min=_FLOOR, max=1.0)


def _pow_schedule(t, exponent: float):
    return torch.clamp(1.0 - t ** exponent, min=_FLOOR, max=1.0)


def pow(t, method):
    """reference spelling of the power schedule: method = "pow<exponent>" (muse/sampling.py:48-52)"""
    return _pow_schedule(t, float(method.replace("pow", "")))


def sigmoid_schedule(t, start=-3, end=3, tau=1.0, clip_min=_FLOOR):
    lo, hi = (torch.sigmoid(torch.tensor(b / tau)) for b in (start, end))
    mid = torch.sigmoid((t * (end - start) + start) / tau)
    return torch.clip((hi - mid) / (hi - lo), clip_min, 1.0)


def get_mask_chedule(method: str, **schedule_kwargs) -> Callable:
    """name -> schedule ("cosine", "linear", "pow<exponent>", "sigmoid"); raises ValueError for anything else"""
    if method == "cosine":
        return cosine_schedule
    if method == "linear":
        return linear_schedule
    if "pow" in method:
        exponent = float(method.replace("pow", ""))
        return lambda t: _pow_schedule(t, exponent)
    if method == "sigmoid":
        return lambda t: sigmoid_schedule(t, **schedule_kwargs)
    raise ValueError("Unknown schedule method: {}".format(method))


def scheduled_mask_len(seq_len: int, step: int, timesteps: int, noise_schedule: Callable = cosine_schedule) -> int:
    """floor(seq_len * schedule((step + 1) / timesteps)) with the reference's float32 arithmetic (a 0-dim CPU tensor:
    muse/modeling_transformer.py:1430-1440).  Can be -1 on the last step (cos(pi/2) is slightly negative in float32); the
    device kernel then applies max(1, min(#unknown - 1, .)) per image."""
    ratio = 1.0 * (step + 1) / timesteps
    return int((seq_len * noise_schedule(torch.tensor(ratio))).floor())


def decode_seed(generator: Optional[torch.Generator]) -> int:
    """64-bit key for the in-kernel Philox stream of one generate2 call: two draws from `generator` (on its own device;
    one host read per call), so a seeded generator reproduces the sample and successive calls differ; without a generator
    the key comes from torch's global CPU generator"""
    dev = generator.device if generator is not None else "cpu"
    hi, lo = torch.randint(0, 2 ** 31, (2,), generator=generator, device=dev).tolist()
    return (int(hi) << 32) | int(lo)


def step_noise(noise: Optional[Sequence[Tuple[torch.Tensor, torch.Tensor]]], step: int):
    """(exponential draws [B*S, V], uniform draws [B, S]) of one step when the caller replays a recorded stream, else (None, None)"""
    if noise is None:
        return None, None
    q, u = noise[step]
    return q, u
