"""muse.PipelineMuse — class-conditional generation wrapper (reference: muse/pipeline_muse.py:38-369).

Thin: `generate2` on the transformer then `vae.decode_code`; both run on the HIP kernels.  Text conditioning needs
a CLIP/T5 encoder, which is outside the hot-path build, so only `is_class_conditioned=True` is executable here; the
constructor / `to` / `from_pretrained` / `save_pretrained` signatures are kept.
"""
from __future__ import annotations

import os
from typing import List, Optional, Union

import numpy as np
import torch

from .modeling_maskgit_vqgan import MaskGitVQGAN
from .modeling_transformer import MaskGitTransformer


class PipelineMuse:
    def __init__(self, vae: Optional[MaskGitVQGAN] = None, transformer: Optional[MaskGitTransformer] = None,
                 is_class_conditioned: bool = False, text_encoder=None, tokenizer=None):
        self.vae = vae
        self.transformer = transformer
        self.is_class_conditioned = is_class_conditioned
        self.text_encoder = text_encoder
        self.tokenizer = tokenizer
        self.device = "cpu"

    def to(self, device="cpu", dtype=None):
        if dtype is not None and dtype != torch.float32:
            self.transformer.set_compute_dtype(dtype)
            self.vae.set_compute_dtype(dtype if dtype == torch.bfloat16 else torch.float32)
        self.vae.to(device)
        self.transformer.to(device)
        if self.text_encoder is not None:
            self.text_encoder.to(device)
        self.device = device
        return self

    @torch.no_grad()
    def __call__(self, text: Optional[Union[str, List[str]]] = None, negative_text=None,
                 class_ids: Optional[Union[int, List[int]]] = None, timesteps: int = 8, guidance_scale: float = 8.0,
                 temperature: float = 1.0, topk_filter_thres: float = 0.9, num_images_per_prompt: int = 1,
                 use_maskgit_generate: bool = True, generator: Optional[torch.Generator] = None, use_fp16: bool = False,
                 output_type: str = "pil", **kwargs):
        if text is not None or not self.is_class_conditioned:
            raise NotImplementedError("text-conditioned generation needs a text encoder (outside the MI355X hot-path build)")
        if class_ids is None:
            raise ValueError("Either `text` or `class_ids` must be provided.")
        if isinstance(class_ids, int):
            class_ids = [class_ids]
        class_ids = torch.tensor(class_ids, device=self.device, dtype=torch.long)
        class_ids = class_ids.repeat_interleave(num_images_per_prompt, dim=0)
        ids = self.transformer.generate2(class_ids=class_ids, timesteps=timesteps, temperature=temperature,
                                         guidance_scale=guidance_scale, generator=generator)
        images = self.vae.decode_code(ids)
        images = torch.clamp(images, 0.0, 1.0).permute(0, 2, 3, 1).float().cpu().numpy()
        if output_type == "np":
            return images
        from PIL import Image
        return [Image.fromarray(np.uint8(np.round(im * 255.0))) for im in images]

    def save_pretrained(self, save_directory: Union[str, os.PathLike], push_to_hub: bool = False):
        self.vae.save_pretrained(os.path.join(save_directory, "vae"))
        self.transformer.save_pretrained(os.path.join(save_directory, "transformer"))

    @classmethod
    def from_pretrained(cls, model_name_or_path: str = None, text_encoder_path: Optional[str] = None,
                        vae_path: Optional[str] = None, transformer_path: Optional[str] = None,
                        is_class_conditioned: bool = False, **kwargs):
        if model_name_or_path is not None:
            vae = MaskGitVQGAN.from_pretrained(model_name_or_path, subfolder="vae")
            transformer = MaskGitTransformer.from_pretrained(model_name_or_path, subfolder="transformer")
        else:
            vae = MaskGitVQGAN.from_pretrained(vae_path)
            transformer = MaskGitTransformer.from_pretrained(transformer_path)
        return cls(vae=vae, transformer=transformer, is_class_conditioned=is_class_conditioned)
