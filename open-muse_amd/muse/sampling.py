"""Mask schedules and confidence-based re-masking (reference: muse/sampling.py).  Host-side scalar maths plus a few
torch tensor ops used by generate2; cosine_schedule is also the schedule of the train-step mask sampler."""
import math
from functools import partial

import torch


def log(t, eps=1e-20):
    return torch.log(t.clamp(min=eps))


def gumbel_noise(t, generator=None):
    u = torch.zeros_like(t).uniform_(0, 1, generator=generator)
    return -log(-log(u))


def gumbel_sample(t, temperature=1.0, dim=-1, generator=None):
    return ((t / max(temperature, 1e-10)) + gumbel_noise(t, generator=generator)).argmax(dim=dim)


def top_k(logits, thres=0.9):
    k = math.ceil((1 - thres) * logits.shape[-1])
    val, ind = logits.topk(k, dim=-1)
    out = torch.full_like(logits, float("-inf"))
    out.scatter_(2, ind, val)
    return out


def mask_by_random_topk(mask_len, probs, temperature=1.0, generator=None):
    """mask the `mask_len` least confident positions (confidence = log p + T * gumbel)."""
    confidence = log(probs) + temperature * gumbel_noise(probs, generator=generator)
    cut_off = torch.gather(torch.sort(confidence, dim=-1).values, 1, mask_len.long())
    return confidence < cut_off


def cosine_schedule(t):
    return torch.cos(t * math.pi * 0.5)


def linear_schedule(t):
    return (1 - t).clamp(min=1e-6, max=1.0)


def pow(t, method):
    exponent = float(method.replace("pow", ""))
    return (1.0 - t ** exponent).clamp(min=1e-6, max=1.0)


def sigmoid_schedule(t, start=-3, end=3, tau=1.0, clip_min=1e-6):
    v_start = torch.sigmoid(torch.tensor(start / tau))
    v_end = torch.sigmoid(torch.tensor(end / tau))
    out = torch.sigmoid((t * (end - start) + start) / tau)
    return torch.clip((v_end - out) / (v_end - v_start), clip_min, 1.0)


def get_mask_chedule(method, **schedule_kwargs):
    if method == "cosine":
        return cosine_schedule
    if method == "linear":
        return linear_schedule
    if "pow" in method:
        return partial(pow, method=method)
    if method == "sigmoid":
        return partial(sigmoid_schedule, **schedule_kwargs)
    raise ValueError("Unknown schedule method: {}".format(method))
