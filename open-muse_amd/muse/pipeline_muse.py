"""muse.PipelineMuse / muse.PipelineMuseInpainting - generation wrappers (reference: muse/pipeline_muse.py:38-369, :372-510).

Thin: `generate2` on the transformer, then `vae.decode_code`; both run on the HIP kernels (the inpainting pipeline first tokenises the
picture with `vae.encode`).  The text encoder is the reference's own third-party dependency (a `transformers` CLIP / T5 model and its
tokenizer, handed to the constructor): when one is given, `text=` is encoded by calling it exactly as the reference does; without one the
pipelines take PRE-COMPUTED text states (`prompt_embeds`, ...).  Constructor / `to` / `from_pretrained` / `save_pretrained` signatures kept.
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
        [B, D], with `empty_embeds` / `empty_pooled_embeds` or the negative_* pair for classifier-free guidance).  `text=` is encoded
        by the pipeline's own `text_encoder` / `tokenizer` when the constructor was given them (`_encode_text`: the reference's calls).
        `temperature` may be the reference's (start, end) tuple: MaskGitTransformer.generate2 takes a float, so the tuple's first
        entry is used there."""
        if text is None and class_ids is None and prompt_embeds is None:
            raise ValueError("Either text or class_ids must be provided.")
        if text is not None and class_ids is not None:
            raise ValueError("Only one of text or class_ids may be provided.")
        if text is not None and prompt_embeds is None:
            e = self._encode_text(text, negative_text if negative_prompt_embeds is None else None, clip_skip,
                                  want_empty=negative_prompt_embeds is None)
            prompt_embeds, pooled_embeds = e["prompt_embeds"], e["pooled_embeds"]
            if e["negative_prompt_embeds"] is not None:
                negative_prompt_embeds, negative_pooled_embeds = e["negative_prompt_embeds"], e["negative_pooled_embeds"]
            # (reference :178-186: the empty prompt's states only when there are no negative states, supplied or encoded)
            if e["empty_embeds"] is not None and negative_prompt_embeds is None:
                empty_embeds, empty_pooled_embeds = e["empty_embeds"], e["empty_pooled_embeds"]
        ids, intermediate = self._generate(None, class_ids, prompt_embeds, pooled_embeds, negative_prompt_embeds, negative_pooled_embeds,
                                           empty_embeds, empty_pooled_embeds, timesteps, noise_schedule, guidance_scale, guidance_schedule,
                                           temperature, num_images_per_prompt, generator, orig_size, crop_coords, aesthetic_score,
                                           return_intermediate, transformer_seq_len)
        images = self._decode(ids, output_type)
        if intermediate is not None:
            return images, [self._decode(t, output_type) for t in intermediate]
        return images

    # ---- text states from the pipeline's own encoder (reference :107-190) -------------------------------------------------------------------
    def _encode_text(self, text, negative_text, clip_skip=None, want_empty=True):
        """tokenizer + text encoder called as the reference calls them: penultimate (or `clip_skip`) hidden state + `text_embeds` for a
        transformer with `add_cond_embeds` (CLIPTextModelWithProjection), `last_hidden_state` otherwise (T5 / plain CLIP); the negative
        prompt likewise (always the penultimate layer, :149-152); without a negative prompt the empty prompt's states for
        classifier-free guidance (:178-186).  -> dict of f32 tensors on the pipeline's device (None where the reference has None)"""
        if self.text_encoder is None or self.tokenizer is None:
            raise NotImplementedError("PipelineMuse(text=...) needs the pipeline's `text_encoder` and `tokenizer` (a transformers CLIP / T5 "
                                      "model, as in the reference); without them pass prompt_embeds / pooled_embeds")
        tok, enc = self.tokenizer, self.text_encoder
        pooled_wanted = bool(getattr(self.transformer.config, "add_cond_embeds", False))

        def ids_of(t):
            return tok(t, return_tensors="pt", padding="max_length", truncation=True, max_length=tok.model_max_length).input_ids.to(self.device)

        def states(t, layer):
            if pooled_wanted:
                out = enc(ids_of(t), return_dict=True, output_hidden_states=True)
                return out.hidden_states[layer], out.text_embeds
            return enc(ids_of(t)).last_hidden_state, None

        text = [text] if isinstance(text, str) else list(text)
        f32 = lambda t: None if t is None else t.float()   # noqa: E731
        hidden, pooled = states(text, -(clip_skip + 1) if clip_skip is not None else -2)
        out = dict(prompt_embeds=f32(hidden), pooled_embeds=f32(pooled), negative_prompt_embeds=None, negative_pooled_embeds=None,
                   empty_embeds=None, empty_pooled_embeds=None)
        if negative_text is not None:
            negative_text = [negative_text] * len(text) if isinstance(negative_text, str) else list(negative_text)
            nh, npool = states(negative_text, -2)
            out.update(negative_prompt_embeds=f32(nh), negative_pooled_embeds=f32(npool))
        elif want_empty:   # (not when the caller brought pre-computed negative states: no extra encoder pass, no second conditioning source)
            empty = tok("", padding="max_length", return_tensors="pt").input_ids.to(self.device)
            o = enc(empty, output_hidden_states=True)
            out.update(empty_embeds=f32(o.hidden_states[-2]), empty_pooled_embeds=f32(o[0]))
        return out

    # ---- tokens from conditioning (shared by the inpainting pipeline, which starts from a partly masked token grid) --------------------------
    def _generate(self, input_ids, class_ids, prompt_embeds, pooled_embeds, negative_prompt_embeds, negative_pooled_embeds, empty_embeds,
                  empty_pooled_embeds, timesteps, noise_schedule, guidance_scale, guidance_schedule, temperature, num_images_per_prompt,
                  generator, orig_size, crop_coords, aesthetic_score, return_intermediate=False, transformer_seq_len=None):
        from .sampling import get_mask_chedule
        schedule = get_mask_chedule(noise_schedule) if isinstance(noise_schedule, str) else noise_schedule
        start = {} if input_ids is None else {"input_ids": input_ids}
        if class_ids is not None:
            if isinstance(class_ids, int):
                class_ids = [class_ids]
            class_ids = torch.as_tensor(class_ids, device=self.device, dtype=torch.long).repeat_interleave(num_images_per_prompt, dim=0)
            t0 = float(temperature[0]) if isinstance(temperature, (tuple, list)) else float(temperature)
            ids = self.transformer.generate2(class_ids=class_ids, timesteps=timesteps, temperature=t0, guidance_scale=guidance_scale,
                                             noise_schedule=schedule, generator=generator, **start)
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
                                             generator=generator, **start)
            intermediate = None
        else:
            n = num_images_per_prompt
            rep = lambda t: None if t is None else t.to(self.device).repeat_interleave(n, dim=0)   # noqa: E731
            micro = torch.tensor([list(orig_size) + list(crop_coords) + [aesthetic_score]], device=self.device, dtype=torch.float32)
            temp = tuple(temperature) if isinstance(temperature, (tuple, list)) else temperature
            # small decoding batches are bound by per-launch host time, not by kernels: the forward is captured into a HIP graph that
            # generate2 keeps across calls of one shape (`self.hip_graph`: None = automatic for <= 4096 rows, True / False to force)
            seq = input_ids.shape[1] if input_ids is not None else (transformer_seq_len or 256)
            rows = (2 if guidance_scale > 0 else 1) * prompt_embeds.shape[0] * n * seq
            use_graph = (rows <= 4096 and timesteps >= 2) if getattr(self, "hip_graph", None) is None else bool(self.hip_graph)
            out = self.transformer.generate2(
                rep(prompt_embeds), rep(pooled_embeds), micro,
                None if empty_embeds is None else empty_embeds.to(self.device),
                None if empty_pooled_embeds is None else empty_pooled_embeds.to(self.device),
                negative_embeds=rep(negative_prompt_embeds), negative_cond_embeds=rep(negative_pooled_embeds), temperature=temp,
                timesteps=timesteps, guidance_scale=guidance_scale, guidance_schedule=guidance_schedule, noise_schedule=schedule,
                generator=generator, return_intermediate=return_intermediate, seq_len=seq,
                use_tqdm=False, hip_graph=use_graph, **start)
            ids, intermediate = out if return_intermediate else (out, None)
        return ids, intermediate

    def _decode(self, ids, output_type):
        images = torch.clamp(self.vae.decode_code(ids), 0.0, 1.0).permute(0, 2, 3, 1).float().cpu().numpy()
        if output_type == "np":
            return images
        from PIL import Image
        return [Image.fromarray((255 * im).astype(np.uint8)).convert("RGB") for im in images]   # reference to_pil_image :245-252

    def save_pretrained(self, save_directory: Union[str, os.PathLike], push_to_hub: bool = False):
        """reference :357-369: text encoder + tokenizer under `text_encoder/` (when the pipeline has them), `vae/`, `transformer/`"""
        if not self.is_class_conditioned and self.text_encoder is not None and self.tokenizer is not None:
            self.text_encoder.save_pretrained(os.path.join(save_directory, "text_encoder"))
            self.tokenizer.save_pretrained(os.path.join(save_directory, "text_encoder"))
        self.vae.save_pretrained(os.path.join(save_directory, "vae"))
        self.transformer.save_pretrained(os.path.join(save_directory, "transformer"))

    @classmethod
    def from_pretrained(cls, model_name_or_path: str = None, text_encoder_path: Optional[str] = None,
                        vae_path: Optional[str] = None, transformer_path: Optional[str] = None, vae=None, text_encoder=None,
                        transformer=None, is_class_conditioned: bool = False, **kwargs):
        """reference :254-355, same arguments.  The text encoder and tokenizer of a text-conditioned pipeline are the reference's own
        `transformers` classes, loaded the way the reference loads them (`CLIPTextModelWithProjection`, `AutoTokenizer`) - from
        `<model>/text_encoder` or `text_encoder_path`; a LOCAL checkpoint directory without a `text_encoder/` folder gives a pipeline
        that takes pre-computed text states instead (this build's addition)."""
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

        if model_name_or_path is None and (vae_path is None or transformer_path is None or
                                           (text_encoder_path is None and not is_class_conditioned and text_encoder is None)):
            raise ValueError("If model_name_or_path is None, then text_encoder_path, vae_path, and transformer_path must be provided.")
        sub = {} if model_name_or_path is None else {"subfolder": "text_encoder"}
        te_path = text_encoder_path if model_name_or_path is None else model_name_or_path
        tokenizer = None
        if not is_class_conditioned and te_path is not None:
            local_without = os.path.isdir(str(te_path)) and not os.path.isdir(os.path.join(str(te_path), sub.get("subfolder", "")))
            if not local_without:
                from transformers import AutoTokenizer, CLIPTextModelWithProjection
                if text_encoder is None:
                    text_encoder = CLIPTextModelWithProjection.from_pretrained(te_path, **sub)
                tokenizer = AutoTokenizer.from_pretrained(te_path, **sub)
        if model_name_or_path is not None:
            vae = vae if vae is not None else load_vae(model_name_or_path, subfolder="vae")
            transformer = transformer if transformer is not None else load_transformer(model_name_or_path, subfolder="transformer")
        else:
            vae = vae if vae is not None else load_vae(vae_path)
            transformer = transformer if transformer is not None else load_transformer(transformer_path)
        if is_class_conditioned:
            return cls(vae=vae, transformer=transformer, is_class_conditioned=True)
        return cls(vae=vae, transformer=transformer, text_encoder=text_encoder, tokenizer=tokenizer, is_class_conditioned=False)


def _center_square(image, size):
    """torchvision's Resize(size, BILINEAR) -> CenterCrop(size) -> ToTensor() of the reference (:399-405) on a PIL image, without
    torchvision: the shorter side to `size` (PIL bilinear), the centred size x size window, uint8 / 255 as [3, size, size] f32"""
    from PIL import Image
    w, h = image.size
    nw, nh = (size, int(size * h / w)) if w <= h else (int(size * w / h), size)
    image = image.convert("RGB").resize((nw, nh), Image.BILINEAR)
    left, top = int(round((nw - size) / 2.0)), int(round((nh - size) / 2.0))
    arr = np.asarray(image.crop((left, top, left + size, top + size)), dtype=np.uint8)
    return torch.from_numpy(arr.copy()).permute(2, 0, 1).float().div(255.0)


class PipelineMuseInpainting(PipelineMuse):
    """reference :372-510: tokenise the picture (`vae.encode`), put the mask token where `mask` is set, let `generate2` fill those
    positions (it keeps every token that is not the mask token), decode."""

    @torch.no_grad()
    def __call__(self, image, mask: torch.Tensor, text: Optional[Union[str, List[str]]] = None,
                 negative_text: Optional[Union[str, List[str]]] = None, class_ids=None, timesteps: int = 8, guidance_scale: float = 8.0,
                 guidance_schedule=None, temperature=1.0, topk_filter_thres: float = 0.9, num_images_per_prompt: int = 1,
                 use_maskgit_generate: bool = True, generator: Optional[torch.Generator] = None, use_fp16: bool = False,
                 image_size: int = 256, orig_size=(256, 256), crop_coords=(0, 0), aesthetic_score=6.0,
                 prompt_embeds: Optional[torch.Tensor] = None, pooled_embeds: Optional[torch.Tensor] = None,
                 negative_prompt_embeds: Optional[torch.Tensor] = None, negative_pooled_embeds: Optional[torch.Tensor] = None,
                 empty_embeds: Optional[torch.Tensor] = None, empty_pooled_embeds: Optional[torch.Tensor] = None, output_type: str = "pil"):
        """same arguments as the reference; in addition the pre-computed text states PipelineMuse takes, and `image` may already be a
        [3, H, W] tensor in [0, 1].  `mask`: bool [tokens] (or [h, w]) over the token grid, True = repaint"""
        assert use_maskgit_generate
        if text is None and class_ids is None and prompt_embeds is None:
            raise ValueError("Either text or class_ids must be provided.")
        if text is not None and class_ids is not None:
            raise ValueError("Only one of text or class_ids may be provided.")
        pixels = image if isinstance(image, torch.Tensor) else _center_square(image, image_size)
        _, tokens = self.vae.encode(pixels.unsqueeze(0).to(self.device).float())
        tokens = tokens.reshape(1, -1).clone()
        mask = torch.as_tensor(mask, dtype=torch.bool, device=tokens.device).reshape(-1)
        if mask.numel() != tokens.shape[1]:
            raise ValueError(f"mask has {mask.numel()} entries for {tokens.shape[1]} image tokens")
        tokens[mask[None]] = self.transformer.config.mask_token_id
        if text is not None and prompt_embeds is None:
            e = self._encode_text(text, negative_text, None)
            prompt_embeds, pooled_embeds = e["prompt_embeds"], e["pooled_embeds"]
            negative_prompt_embeds, negative_pooled_embeds = e["negative_prompt_embeds"], e["negative_pooled_embeds"]
            empty_embeds, empty_pooled_embeds = e["empty_embeds"], e["empty_pooled_embeds"]
        batch = (len(class_ids) if isinstance(class_ids, (list, tuple)) else 1) if class_ids is not None else prompt_embeds.shape[0]
        tokens = tokens.repeat(batch * num_images_per_prompt, 1)
        ids, _ = self._generate(tokens, class_ids, prompt_embeds, pooled_embeds, negative_prompt_embeds, negative_pooled_embeds,
                                empty_embeds, empty_pooled_embeds, timesteps, "cosine", guidance_scale, guidance_schedule, temperature,
                                num_images_per_prompt, generator, orig_size, crop_coords, aesthetic_score)
        return self._decode(ids, output_type)

