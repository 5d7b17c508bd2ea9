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

    def to(self, device="cpu", dtype=torch.float32):
        """reference :55-64: transformer (and text encoder) to `dtype`, the VAE always stays in float32.  Here `dtype` selects the
        transformer's COMPUTE mode (its master weights stay f32): bfloat16 and float16 both map to the bf16 MFMA path (the HIP
        kernels have no f16 variant; bf16 has the wider exponent and the same 2x MFMA rate), float32 to the exact-f32 path.  The
        VAE keeps its own f32-class mode (exact f32 or bf16x3), whatever `dtype` says."""
        self.device = device
        self.dtype = dtype
        self.transformer.set_compute_dtype(torch.bfloat16 if dtype in (torch.bfloat16, torch.float16) else torch.float32)
        if getattr(self.vae, "compute_dtype", None) == torch.bfloat16:
            self.vae.set_compute_dtype("bf16x3")
        self.vae.to(device)
        self.transformer.to(device)
        if self.text_encoder is not None:
            self.text_encoder.to(device, dtype=dtype)
        return self

    @torch.no_grad()
    def __call__(self, text: Optional[Union[str, List[str]]] = None, negative_text: Optional[Union[str, List[str]]] = "",
                 prompt_embeds: Optional[torch.Tensor] = None, pooled_embeds: Optional[torch.Tensor] = None,
                 negative_prompt_embeds: Optional[torch.Tensor] = None, negative_pooled_embeds: Optional[torch.Tensor] = None,
                 class_ids: Optional[Union[int, List[int]]] = None, timesteps: int = 16, noise_schedule: str = "cosine",
                 guidance_scale: float = 10.0, guidance_schedule=None, temperature=(2, 0), topk_filter_thres: float = 0.9,
                 num_images_per_prompt: int = 1, use_maskgit_generate: bool = True, generator: Optional[torch.Generator] = None,
                 use_fp16: bool = False, noise_type="mask", predict_all_tokens=False, orig_size=(512, 512), crop_coords=(0, 0),
                 aesthetic_score=6.0, return_intermediate: bool = False, use_tqdm=True, transformer_seq_len=None,
                 clip_skip: int = None, output_type: str = "pil", empty_embeds: Optional[torch.Tensor] = None,
                 empty_pooled_embeds: Optional[torch.Tensor] = None):
        """reference :66-243, same argument names and defaults.  Three executable paths: class-conditional MaskGitTransformer
        (`class_ids`), text-conditioned MaskGitTransformer on PRE-COMPUTED text states (`prompt_embeds`, `negative_prompt_embeds`),
        and MaskGiTUViT conditioned on PRE-COMPUTED text states (`prompt_embeds` [B, 77, D] + `pooled_embeds`
        [B, D], with `empty_embeds` / `empty_pooled_embeds` or the negative_* pair for classifier-free guidance); running a text
        encoder on `text` needs CLIP, which is outside the hot-path build.  `temperature` may be the reference's (start, end)
        tuple: MaskGitTransformer.generate2 takes a float, so the tuple's first entry is used there."""
        from .sampling import get_mask_chedule
        if text is None and class_ids is None and prompt_embeds is None:
            raise ValueError("Either text or class_ids must be provided.")
        if text is not None and class_ids is not None:
            raise ValueError("Only one of text or class_ids may be provided.")
        if text is not None:
            raise NotImplementedError("text-conditioned generation from raw text needs a text encoder (outside the MI355X hot-path "
                                      "build): pass prompt_embeds / pooled_embeds instead")
        schedule = get_mask_chedule(noise_schedule)
        if class_ids is not None:
            if isinstance(class_ids, int):
                class_ids = [class_ids]
            class_ids = torch.tensor(class_ids, device=self.device, dtype=torch.long).repeat_interleave(num_images_per_prompt, dim=0)
            t0 = float(temperature[0]) if isinstance(temperature, (tuple, list)) else float(temperature)
            ids = self.transformer.generate2(class_ids=class_ids, timesteps=timesteps, temperature=t0, guidance_scale=guidance_scale,
                                             noise_schedule=schedule, generator=generator)
            intermediate = None
        elif isinstance(self.transformer, MaskGitTransformer):
            # text-conditioned MaskGitTransformer on pre-computed text states (reference :207-243 hands the same keyword set to
            # every transformer class; MaskGitTransformer.generate2 takes the states, the negative states and a float temperature)
            n = num_images_per_prompt
            rep = lambda t: None if t is None else t.to(self.device).repeat_interleave(n, dim=0)   # noqa: E731
            t0 = float(temperature[0]) if isinstance(temperature, (tuple, list)) else float(temperature)
            # `empty_embeds` is NOT the negative prompt here: the reference's MaskGitTransformer.generate2 swallows it in **kwargs
            # (muse/modeling_transformer.py:1363-1378) and fills the unconditional half with zeros_like(encoder_hidden_states) when no
            # negative_embeds are given (:1398-1402) - so does ours
            ids = self.transformer.generate2(encoder_hidden_states=rep(prompt_embeds), negative_embeds=rep(negative_prompt_embeds), timesteps=timesteps,
                                             temperature=t0, guidance_scale=guidance_scale, noise_schedule=schedule,
                                             generator=generator)
            intermediate = None
        else:
            n = num_images_per_prompt
            rep = lambda t: None if t is None else t.to(self.device).repeat_interleave(n, dim=0)   # noqa: E731
            micro = torch.tensor([list(orig_size) + list(crop_coords) + [aesthetic_score]], device=self.device, dtype=torch.float32)
            temp = tuple(temperature) if isinstance(temperature, (tuple, list)) else temperature
            # small decoding batches are bound by per-launch host time, not by kernels: the forward is captured into a HIP graph that
            # generate2 keeps across calls of one shape (`self.hip_graph`: None = automatic for <= 4096 rows, True / False to force)
            rows = (2 if guidance_scale > 0 else 1) * prompt_embeds.shape[0] * n * (transformer_seq_len or 256)
            use_graph = (rows <= 4096 and timesteps >= 2) if getattr(self, "hip_graph", None) is None else bool(self.hip_graph)
            out = self.transformer.generate2(
                rep(prompt_embeds), rep(pooled_embeds), micro,
                None if empty_embeds is None else empty_embeds.to(self.device),
                None if empty_pooled_embeds is None else empty_pooled_embeds.to(self.device),
                negative_embeds=rep(negative_prompt_embeds), negative_cond_embeds=rep(negative_pooled_embeds), temperature=temp,
                timesteps=timesteps, guidance_scale=guidance_scale, guidance_schedule=guidance_schedule, noise_schedule=schedule,
                generator=generator, return_intermediate=return_intermediate, seq_len=transformer_seq_len,
                use_tqdm=False if use_tqdm is None else use_tqdm and False, hip_graph=use_graph)
            ids, intermediate = out if return_intermediate else (out, None)
        images = self._decode(ids, output_type)
        if intermediate is not None:
            return images, [self._decode(t, output_type) for t in intermediate]
        return images

    def _decode(self, ids, output_type):
        images = torch.clamp(self.vae.decode_code(ids), 0.0, 1.0).permute(0, 2, 3, 1).float().cpu().numpy()
        if output_type == "np":
            return images
        from PIL import Image
        return [Image.fromarray((255 * im).astype(np.uint8)).convert("RGB") for im in images]   # reference to_pil_image :245-252

    def save_pretrained(self, save_directory: Union[str, os.PathLike], push_to_hub: bool = False):
        self.vae.save_pretrained(os.path.join(save_directory, "vae"))
        self.transformer.save_pretrained(os.path.join(save_directory, "transformer"))

    @classmethod
    def from_pretrained(cls, model_name_or_path: str = None, text_encoder_path: Optional[str] = None,
                        vae_path: Optional[str] = None, transformer_path: Optional[str] = None,
                        is_class_conditioned: bool = False, **kwargs):
        def load_transformer(path, **kw):   # the class named in the checkpoint's config.json (reference :300-318)
            from .modeling_transformer_v2 import MaskGiTUViT_v2
            cfg = MaskGitTransformer.load_config(path, **kw)
            klass = MaskGiTUViT_v2 if str(cfg.get("_class_name", "")).startswith("MaskGiTUViT") else MaskGitTransformer
            return klass.from_pretrained(path, **kw)
        def load_vae(path, **kw):           # likewise for the tokenizer (reference :320-329; MoVQ / Paella are not part of this build)
            from .modeling_taming_vqgan import VQGANModel
            name = str(MaskGitVQGAN.load_config(path, **kw).get("_class_name", "MaskGitVQGAN"))
            if name not in ("MaskGitVQGAN", "VQGANModel"):
                raise ValueError(f"Unknown VAE class: {name}")
            return (VQGANModel if name == "VQGANModel" else MaskGitVQGAN).from_pretrained(path, **kw)
        if model_name_or_path is not None:
            vae = load_vae(model_name_or_path, subfolder="vae")
            transformer = load_transformer(model_name_or_path, subfolder="transformer")
        else:
            vae = load_vae(vae_path)
            transformer = load_transformer(transformer_path)
        return cls(vae=vae, transformer=transformer, is_class_conditioned=is_class_conditioned)
