"""Pre-encoded token shards — SURVEY.md section 8(f) row 4: take the VQ tokenizer out of the train step.

The reference's text-to-image configs never run the VQGAN inside the loop: scripts/pre_encode.py:440-511 encodes every image once
(`vae.get_code(image)`) and writes webdataset shards whose samples carry `<key>.<vae checkpoint>.pth` (the token ids) and
`<key>.<text encoder checkpoint>.pth` (the text states) next to a `<key>.json`; training/data.py:561-573 reads them back as
`image_input_ids` / `encoder_hidden_states` (the TODO at training/train_maskgit_imagenet.py:404 asks for the same for ImageNet).

This module produces and consumes that layout with the tokenizer on the HIP kernels (`MaskGitVQGAN.get_code`) and plain `tarfile`
(webdataset itself is not needed to write or read a POSIX tar of `<key>.<ext>` members): shards written here are readable by the
reference's pipeline and vice versa (member extensions are matched case-insensitively, as webdataset does).  `muse.TrainStep(...)(pixel_values=None, class_ids, image_tokens=tokens)` is the step that
consumes the tokens.
"""
from __future__ import annotations

import io
import json
import os
import tarfile
from typing import Dict, Iterable, Iterator, Optional, Sequence

import torch


def checkpoint_ext(checkpoint: str) -> str:
    """member extension of a checkpoint name: scripts/pre_encode.py:54-56 joins the path parts with '.', training/data.py:562-563
    lower-cases it and replaces '/' by '.' before matching"""
    return checkpoint.lower().replace("/", ".") + ".pth"


def _add(tar: tarfile.TarFile, name: str, payload: bytes):
    info = tarfile.TarInfo(name)
    info.size = len(payload)
    tar.addfile(info, io.BytesIO(payload))


def _pth(t: torch.Tensor) -> bytes:
    buf = io.BytesIO()
    torch.save(t.detach().cpu().clone(), buf)      # what webdataset's `pth` handler (torch_dumps) writes
    return buf.getvalue()


def write_token_shard(path: str, keys: Sequence[str], image_tokens: torch.Tensor, vae_checkpoint: str,
                      encoder_hidden_states: Optional[torch.Tensor] = None, text_encoder_checkpoint: Optional[str] = None,
                      metadata: Optional[Sequence[dict]] = None) -> None:
    """one tar shard: per sample `<key>.<vae>.pth` = int64 [num_vq_tokens] (+ `<key>.<text encoder>.pth`, `<key>.json`)"""
    if len(keys) != image_tokens.shape[0]:
        raise ValueError("one key per row of image_tokens")
    with tarfile.open(path, "w") as tar:
        for i, key in enumerate(keys):
            _add(tar, f"{key}.{checkpoint_ext(vae_checkpoint)}", _pth(image_tokens[i].to(torch.int64)))
            if encoder_hidden_states is not None:
                _add(tar, f"{key}.{checkpoint_ext(text_encoder_checkpoint)}", _pth(encoder_hidden_states[i]))
            _add(tar, f"{key}.json", json.dumps(dict(metadata[i]) if metadata is not None else {}).encode())


def read_token_shard(path: str, vae_checkpoint: str, text_encoder_checkpoint: Optional[str] = None) -> Iterator[Dict[str, object]]:
    """samples of a shard as training/data.py:561-573 presents them: {"__key__", "image_input_ids"[, "encoder_hidden_states"]}"""
    want = {checkpoint_ext(vae_checkpoint): "image_input_ids"}
    if text_encoder_checkpoint is not None:
        want[checkpoint_ext(text_encoder_checkpoint)] = "encoder_hidden_states"
    cur_key, cur = None, {}
    with tarfile.open(path, "r") as tar:
        for m in tar:
            if not m.isfile():
                continue
            # webdataset (base_plus_ext): the key is the directory part plus everything before the first dot of the BASE name, and
            # the extension is lower-cased on read - the reference writes mixed-case members (`<key>.openMUSE.vqgan-....pth`)
            dirname, base = os.path.split(m.name)
            stem, _, ext = base.partition(".")
            key, ext = (os.path.join(dirname, stem) if dirname else stem), ext.lower()
            if key != cur_key:
                if cur_key is not None and "image_input_ids" in cur:
                    yield cur
                cur_key, cur = key, {"__key__": key}
            if ext in want:
                cur[want[ext]] = torch.load(io.BytesIO(tar.extractfile(m).read()), map_location="cpu")
    if cur_key is not None and "image_input_ids" in cur:
        yield cur


@torch.no_grad()
def pre_encode(vq_model, batches: Iterable, shard_pattern: str, vae_checkpoint: str, samples_per_shard: int = 10000) -> int:
    """encode batches of (keys, pixel_values [B,3,H,W] in [0,1] on the GPU) with `vq_model.get_code` (scripts/pre_encode.py:497-498)
    and write them as shards `shard_pattern % n`; returns the number of samples written"""
    n_written, shard, keys, toks = 0, 0, [], []

    def flush():
        nonlocal shard, keys, toks
        if keys:
            write_token_shard(shard_pattern % shard, keys, torch.cat(toks), vae_checkpoint)
            shard, keys, toks = shard + 1, [], []
    for bkeys, pixel_values in batches:
        codes = vq_model.get_code(pixel_values).cpu()
        keys += list(bkeys)
        toks.append(codes)
        n_written += len(bkeys)
        if len(keys) >= samples_per_shard:
            flush()
    flush()
    return n_written


def token_batches(shards: Sequence[str], vae_checkpoint: str, batch_size: int, device="cuda") -> Iterator[torch.Tensor]:
    """[batch_size, num_vq_tokens] int64 batches on `device` from pre-encoded shards (drops the ragged tail like the reference's
    `.batched(batch_size, partial=False)`)"""
    buf = []
    for path in shards:
        for s in read_token_shard(path, vae_checkpoint):
            buf.append(s["image_input_ids"].reshape(-1))
            if len(buf) == batch_size:
                yield torch.stack(buf).to(device, non_blocking=True)
                buf = []
