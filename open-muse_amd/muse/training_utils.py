"""muse.training_utils - what training/train_muse.py reaches for in `muse.training_utils` (`import muse.training_utils`, :50): seeding
helpers and the four logging diagnostics it computes from a step's logits every `log_*_every` steps (:1319-1375; reference
muse/training_utils.py:27-58, :299-455).  Diagnostics, not hot path: a few torch reductions over logits the step already produced,
bucketed by the share of masked tokens per image.  Each function returns exactly what the reference's returns for the same tensors -
including where the reference's indexing is not what its comments describe (noted inline) - pinned by tests/golden/training_utils.npz.
The deep-copy `EMA` class of that file is superseded by `muse.EMAModel`, which is what the training script uses (:367-373).
"""
from __future__ import annotations

import os
import random

import numpy as np
import torch
import torch.nn.functional as F

_EDGES = (0.0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0)
_BUCKETS = 10


def set_seed(seed: int):
    """`random`, numpy and torch (all devices) from one seed"""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)


def enable_full_determinism(seed: int):
    """seed everything and ask torch for deterministic algorithms (the environment switches the reference sets are kept: harmless on ROCm)"""
    set_seed(seed)
    os.environ["CUDA_LAUNCH_BLOCKING"] = "1"
    os.environ["CUBLAS_WORKSPACE_CONFIG"] = ":16:8"
    torch.use_deterministic_algorithms(True)
    torch.backends.cudnn.deterministic = True
    torch.backends.cudnn.benchmark = False


def input_ids_to_masked_buckets(input_ids, mask_id, total_buckets=10):
    """per image: which tenth its share of masked tokens falls into - (0, 0.1] -> 0, ..., (0.9, 1] -> 9; an image with nothing masked
    also lands in 0"""
    assert total_buckets == _BUCKETS
    share = (input_ids == mask_id).sum(-1) / input_ids.shape[-1]
    bucket = torch.zeros(share.shape, dtype=torch.long, device=share.device)
    for k in range(1, _BUCKETS):
        bucket += ((_EDGES[k] < share) & (share <= _EDGES[k + 1])) * k
    return bucket


def average_by_buckets(values, masked_buckets, total_buckets):
    """mean of `values` per bucket (0 for an empty bucket).  `values` may be longer than `masked_buckets`: like the reference's
    scatter_add_, only its leading len(masked_buckets) entries take part"""
    total = torch.zeros(total_buckets, device=values.device).scatter_add_(0, masked_buckets, values)
    count = torch.bincount(masked_buckets, minlength=total_buckets).clamp(min=1)
    return total / count


def pixel_entropy_per_percent_masked_bucket(logits, input_ids, mask_id):
    """entropy of each masked token's predicted distribution, averaged per image over its masked tokens, then per bucket"""
    masked = input_ids == mask_id
    logp = F.log_softmax(logits, dim=-1)
    entropy = -(F.softmax(logits, dim=-1) * logp).sum(-1)
    entropy[~masked] = 0
    per_image = entropy.sum(-1) / masked.sum(-1)
    return average_by_buckets(per_image, input_ids_to_masked_buckets(input_ids, mask_id, _BUCKETS), _BUCKETS)


def image_entropy_per_percent_masked_bucket(logits, input_ids, mask_id):
    """entropy of the image's AVERAGE predicted distribution over its masked tokens, then per bucket"""
    masked = input_ids == mask_id
    probs = F.softmax(logits, dim=-1)
    probs[~masked] = 0
    mean_probs = probs.sum(-2) / masked.sum(-1, keepdim=True)
    per_image = -(mean_probs * mean_probs.log()).sum(-1)
    return average_by_buckets(per_image, input_ids_to_masked_buckets(input_ids, mask_id, _BUCKETS), _BUCKETS)


def cross_entropy_per_percent_masked_bucket(logits, labels, input_ids, mask_id, output_size, label_smoothing):
    """per-token cross-entropy handed to the bucket average.  (The reference passes one value per TOKEN with one bucket index per IMAGE,
    so the average runs over the first batch-size token losses - kept, see average_by_buckets.)"""
    per_token = F.cross_entropy(logits.view(-1, output_size), labels.view(-1), ignore_index=-100, label_smoothing=label_smoothing,
                                reduction="none")
    return average_by_buckets(per_token, input_ids_to_masked_buckets(input_ids, mask_id, _BUCKETS), _BUCKETS)


def token_probability_distributions_per_percent_masked_bucket(logits, input_ids, mask_id):
    """pandas frame {bucket, masked_pixel_prob}: for every bucket that occurs, the predicted distribution of the first masked token of
    ONE image.  (The reference indexes the batch with the bucket's own number - `masked_buckets[masked_buckets == b][0]` is b - kept.)"""
    import pandas as pd
    probs = F.softmax(logits, dim=-1)
    buckets = input_ids_to_masked_buckets(input_ids, mask_id, _BUCKETS)
    rows = []
    for b in range(_BUCKETS):
        if not bool((buckets == b).any()):
            continue
        first_masked = probs[b][input_ids[b] == mask_id][0]
        rows += [{"bucket": b, "masked_pixel_prob": p} for p in first_masked.cpu().numpy()]
    return pd.DataFrame(rows)
