"""GPU (-m gpu): SURVEY.md section 8 "next" rows f2 / f3 on the HIP path:
  * muse_sample_step (one MaskGit decoding iteration) against the CPU oracle with the same random draws, bit-exact ids;
  * MaskGitTransformer.generate2 / MaskGiTUViT.generate2 against the REAL reference's output (tests/golden: the reference ran
    with a seeded CPU generator, the golden records the generator's draws), bit-exact ids;
  * muse_mask_tokens / muse_cond_dropout against the real reference's mask_or_random_replace_tokens / cond-dropout statements
    (training/train_muse.py:149-226, :715-731), bit-exact.
"""
import json
import os
import random

import numpy as np
import pytest
import torch

import weights as W

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ops():
    from muse import ops
    return ops


def _noise(g, steps):
    return [(torch.from_numpy(g[f"q{i}"]), torch.from_numpy(g[f"u{i}"])) for i in range(steps)]


@pytest.mark.parametrize("B,S,V,ld,guided", [(3, 16, 32, 48, False), (4, 256, 1024, 2048, False), (2, 256, 8192, 8192, True),
                                             (2, 1024, 8192, 8192, True)])
def test_sample_step_vs_oracle(B, S, V, ld, guided):
    """softmax -> categorical (exponential race) -> confidence + Gumbel -> k-th smallest -> re-mask, same draws as the oracle"""
    from oracle import maskgit_oracle as O
    ops = _ops()
    g = torch.Generator().manual_seed(1234 + S + V)
    cond = torch.randn(B, S, ld, generator=g) * 2.0
    unc = torch.randn(B, S, ld, generator=g) * 2.0 if guided else None
    scale = 3.0
    mask_id = V + 7
    ids = torch.where(torch.rand(B, S, generator=g) < 0.7, torch.full((B, S), mask_id), torch.randint(0, V, (B, S), generator=g))
    ids[0, :] = mask_id                                   # one image fully masked
    q = torch.empty(B * S, V).exponential_(1, generator=g)
    u = torch.rand(B, S, generator=g)
    for temperature, sched in ((4.5, S // 2), (0.0, -1), (1.3, 3)):
        logits = (unc + scale * (cond - unc)) if guided else cond
        raw_o, samp_o, next_o = O.sample_step(logits[..., :V], ids, mask_id, temperature, sched, q, u)
        samp, nxt, raw = ops.sample_step(cond.to(DEV), ids.to(DEV), mask_id, V, temperature, sched,
                                         uncond_logits=unc.to(DEV) if guided else None, guidance_scale=scale,
                                         noise_exp=q.to(DEV), noise_u=u.to(DEV), want_raw=True)
        assert torch.equal(raw.cpu(), raw_o) and torch.equal(samp.cpu(), samp_o) and torch.equal(nxt.cpu(), next_o), (temperature, sched)


def test_sample_step_device_rng():
    """without supplied draws the kernel's Philox stream is used: reproducible for a seed, different per step / seed, and the
    categorical frequencies follow the softmax"""
    ops = _ops()
    B, S, V = 64, 256, 16
    logits = torch.log(torch.tensor([0.4, 0.2, 0.1, 0.1] + [0.2 / 12] * 12)).repeat(B, S, 1).contiguous().to(DEV)
    ids = torch.full((B, S), 99, dtype=torch.long, device=DEV)
    a, na, _ = ops.sample_step(logits, ids, 99, V, 1.0, S // 2, seed=42, step=3)
    b, nb, _ = ops.sample_step(logits, ids, 99, V, 1.0, S // 2, seed=42, step=3)
    c, _, _ = ops.sample_step(logits, ids, 99, V, 1.0, S // 2, seed=42, step=4)
    d, _, _ = ops.sample_step(logits, ids, 99, V, 1.0, S // 2, seed=43, step=3)
    assert torch.equal(a, b) and torch.equal(na, nb)
    assert not torch.equal(a, c) and not torch.equal(a, d)
    freq = torch.bincount(a.flatten().cpu(), minlength=V).double() / (B * S)
    assert abs(float(freq[0]) - 0.4) < 0.02 and abs(float(freq[1]) - 0.2) < 0.02 and abs(float(freq[4:].sum()) - 0.2) < 0.02
    assert int((na == 99).sum(-1).min()) == S // 2 and int((na == 99).sum(-1).max()) == S // 2   # exactly mask_len re-masked, no ties


def test_generate2_vs_reference_golden(golden_dir):
    """the reference's MaskGitTransformer.generate2 sample (6 steps, temperature 4.5, seeded generator) reproduced id for id"""
    import muse
    g = np.load(os.path.join(golden_dir, "generate2_tiny.npz"))
    cfg = W.TRANSFORMER_TINY
    m = muse.MaskGitTransformer(**cfg)
    m.load_state_dict(W.fill_state_dict(W.transformer_shapes(cfg), int(g["seed"]), "transformer"))
    m.to(DEV).eval().set_compute_dtype(torch.float32)
    T = int(g["timesteps"])
    cls = torch.from_numpy(g["class_ids"]).to(DEV)
    ids = m.generate2(class_ids=cls, timesteps=T, temperature=float(g["temperature"]), noise=_noise(g, T))
    assert torch.equal(ids.cpu(), torch.from_numpy(g["ids"]))
    assert torch.equal(cls.cpu(), torch.from_numpy(g["class_ids"]) + cfg["codebook_size"])   # shifted in place like the reference
    # production path (device RNG): valid ids, reproducible with a seeded generator, every token decoded
    outs = [m.generate2(class_ids=torch.from_numpy(g["class_ids"]).to(DEV), timesteps=T, temperature=2.0,
                        generator=torch.Generator(device=DEV).manual_seed(7)) for _ in range(2)]
    assert torch.equal(outs[0], outs[1]) and int(outs[0].max()) < cfg["codebook_size"] and int(outs[0].min()) >= 0


def test_generate2_text_guided_vs_reference_golden(golden_dir):
    """text-conditioned MaskGitTransformer.generate2 with classifier-free guidance 2.5 (doubled batch; zeros or negative_embeds as
    the unconditional half, reference :1394-1416): the real reference's ids from its recorded generator draws"""
    import muse
    g = np.load(os.path.join(golden_dir, "generate2_text_tiny.npz"))
    cfg = W.TRANSFORMER_TEXT_TINY
    m = muse.MaskGitTransformer(**cfg)
    m.load_state_dict(W.fill_state_dict(W.transformer_shapes(cfg), int(g["seed"]), "transformer"))
    m.to(DEV).eval().set_compute_dtype(torch.float32)
    _, _, enc = W.transformer_text_inputs(cfg, int(g["batch"]), int(g["text_len"]), int(g["seed"]) + 1)
    T = int(g["timesteps"])
    for tag, neg in (("", None), ("_neg", torch.from_numpy(g["negative_embeds"]).to(DEV))):
        ids = m.generate2(encoder_hidden_states=enc.to(DEV), negative_embeds=neg, timesteps=T, temperature=float(g["temperature"]),
                          guidance_scale=float(g["guidance_scale"]), noise=_noise(g, T))
        assert torch.equal(ids.cpu(), torch.from_numpy(g["ids" + tag])), tag
    # unguided + device RNG: valid ids, reproducible
    outs = [m.generate2(encoder_hidden_states=enc.to(DEV), timesteps=T, temperature=2.0, generator=torch.Generator(device=DEV).manual_seed(3))
            for _ in range(2)]
    assert torch.equal(outs[0], outs[1]) and int(outs[0].max()) < cfg["codebook_size"]


def test_pipeline_text_conditioned_two_prompts_guided(golden_dir):
    """muse.PipelineMuse on a text-conditioned MaskGitTransformer with TWO prompts, two images per prompt and guidance > 0: the
    unconditional half is zeros unless `negative_prompt_embeds` is given - `empty_embeds` (a [1, L, D] tensor the reference's
    MaskGitTransformer.generate2 swallows in **kwargs, muse/modeling_transformer.py:1363-1402) must not become the negative prompt
    (it would neither match the reference nor the doubled batch's row count).  The pipeline's token ids == generate2 called the
    way the reference's pipeline calls it, with the same generator."""
    import muse
    g = np.load(os.path.join(golden_dir, "generate2_text_tiny.npz"))
    cfg = W.TRANSFORMER_TEXT_TINY
    m = muse.MaskGitTransformer(**cfg)
    m.load_state_dict(W.fill_state_dict(W.transformer_shapes(cfg), int(g["seed"]), "transformer"))
    v = muse.MaskGitVQGAN(**W.VQGAN_TINY)
    pipe = muse.PipelineMuse(vae=v, transformer=m).to(DEV)
    m.eval().set_compute_dtype(torch.float32)
    L = int(g["text_len"])
    _, _, enc = W.transformer_text_inputs(cfg, 2, L, int(g["seed"]) + 1)
    empty = torch.full((1, L, enc.shape[-1]), 0.37)
    neg = torch.from_numpy(g["negative_embeds"])[:2]
    seen = {}
    orig = m.generate2
    def spy(*a, **k):
        seen["neg"] = k.get("negative_embeds")
        seen["ids"] = orig(*a, **k)
        return seen["ids"]
    m.generate2 = spy
    try:
        gen = lambda: torch.Generator(device=DEV).manual_seed(11)   # noqa: E731
        imgs = pipe(prompt_embeds=enc, empty_embeds=empty, timesteps=3, guidance_scale=2.5, num_images_per_prompt=2, output_type="np",
                    generator=gen())
        assert imgs.shape[0] == 4 and np.isfinite(imgs).all() and seen["neg"] is None
        want = orig(encoder_hidden_states=enc.to(DEV).repeat_interleave(2, dim=0), negative_embeds=None, timesteps=3, temperature=2.0,
                    guidance_scale=2.5, generator=gen())
        assert torch.equal(seen["ids"], want)
        pipe(prompt_embeds=enc, negative_prompt_embeds=neg, empty_embeds=empty, timesteps=3, guidance_scale=2.5, num_images_per_prompt=2,
             output_type="np", generator=gen())
        assert seen["neg"].shape[0] == 4
        want = orig(encoder_hidden_states=enc.to(DEV).repeat_interleave(2, dim=0), negative_embeds=neg.to(DEV).repeat_interleave(2, dim=0),
                    timesteps=3, temperature=2.0, guidance_scale=2.5, generator=gen())
        assert torch.equal(seen["ids"], want)
    finally:
        m.generate2 = orig


class _StubTokenizer:
    """stands in for a transformers tokenizer: fixed-length ids from the characters of the prompt"""
    model_max_length = 7

    def __call__(self, text, return_tensors="pt", padding=None, truncation=None, max_length=None):
        text = [text] if isinstance(text, str) else list(text)
        ids = torch.tensor([[(sum(map(ord, t)) + 7 * i * (len(t) + 1)) % 50 for i in range(self.model_max_length)] for t in text])
        return type("Enc", (), {"input_ids": ids})()


class _StubTextOut:
    def __init__(self, hidden_states, text_embeds):
        self.hidden_states, self.text_embeds, self.last_hidden_state = hidden_states, text_embeds, hidden_states[-1]

    def __getitem__(self, i):            # CLIPTextModelWithProjection output: [0] = text_embeds (reference :184-186 reads outputs[0])
        return (self.text_embeds, self.last_hidden_state)[i]


class _StubTextEncoder(torch.nn.Module):
    """stands in for CLIPTextModelWithProjection: three "layers" of hidden states and a pooled projection"""

    def __init__(self, width, pooled):
        super().__init__()
        torch.manual_seed(5)
        self.emb, self.proj = torch.nn.Embedding(50, width), torch.nn.Linear(width, pooled)

    def forward(self, input_ids, return_dict=None, output_hidden_states=None):
        h0 = self.emb(input_ids)
        h1 = torch.tanh(h0 * 2 + 0.5)
        h2 = h1 * h1 - 0.25
        return _StubTextOut((h0, h1, h2), self.proj(h2.mean(dim=1)))


def test_pipeline_encodes_text_with_its_own_encoder(golden_dir):
    """PipelineMuse(text=...) with a text encoder and tokenizer handed to the constructor (the reference's own third-party dependency:
    here two stubs with the transformers call signatures): the pipeline calls them as the reference does (:107-190) - penultimate hidden
    state (or `clip_skip`) + `text_embeds`, the negative prompt on the penultimate layer, the empty prompt's states when there is no
    negative prompt - and produces the images the pre-computed-states entry produces from the same tensors; without an encoder the
    refusal is loud"""
    import muse
    gp = np.load(os.path.join(golden_dir, "uvit_tiny.npz"))
    cfg = json.load(open(os.path.join(golden_dir, "config_uvit_tiny.json")))
    u = muse.MaskGiTUViT(**cfg)
    u.load_state_dict({k[len("param."):]: torch.from_numpy(gp[k]) for k in gp.files if k.startswith("param.")}, strict=True)
    v = muse.MaskGitVQGAN(**W.VQGAN_TINY)
    tok, enc = _StubTokenizer(), _StubTextEncoder(cfg["encoder_hidden_size"], cfg["cond_embed_dim"])
    pipe = muse.PipelineMuse(vae=v, transformer=u, text_encoder=enc, tokenizer=tok).to(DEV)
    u.eval()
    prompts = ["a red fox", "two cats"]
    gen = lambda: torch.Generator(device=DEV).manual_seed(21)   # noqa: E731
    kw = dict(timesteps=3, guidance_scale=2.0, num_images_per_prompt=2, output_type="np", transformer_seq_len=16)
    with torch.no_grad():
        o = enc(tok(prompts).input_ids.to(DEV))
        on = enc(tok([""] * 2).input_ids.to(DEV))
        oe = enc(tok("").input_ids.to(DEV))
    # (1) default negative_text "" -> the negative prompt's penultimate states and pooled embedding
    a = pipe(text=prompts, generator=gen(), **kw)
    b = pipe(prompt_embeds=o.hidden_states[-2], pooled_embeds=o.text_embeds, negative_prompt_embeds=on.hidden_states[-2],
             negative_pooled_embeds=on.text_embeds, generator=gen(), **kw)
    assert a.shape == (4, 16, 16, 3) and np.array_equal(a, b)
    # (2) no negative prompt -> the empty prompt's states; clip_skip picks the layer
    a = pipe(text=prompts, negative_text=None, clip_skip=2, generator=gen(), **kw)
    b = pipe(prompt_embeds=o.hidden_states[-3], pooled_embeds=o.text_embeds, empty_embeds=oe.hidden_states[-2], empty_pooled_embeds=oe.text_embeds,
             generator=gen(), **kw)
    c = pipe(prompt_embeds=o.hidden_states[-2], pooled_embeds=o.text_embeds, empty_embeds=oe.hidden_states[-2], empty_pooled_embeds=oe.text_embeds,
             generator=gen(), **kw)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    with pytest.raises(NotImplementedError):
        muse.PipelineMuse(vae=v, transformer=u).to(DEV)(text="a red fox")


def test_pipeline_from_pretrained_text_to_image_with_a_real_clip_tower(golden_dir, tmp_path):
    """the text-to-image entry of the reference end to end: a checkpoint directory laid out as PipelineMuse.save_pretrained writes it
    (text_encoder/ = a real transformers CLIPTextModelWithProjection + CLIPTokenizer, tiny and built offline; vae/; transformer/),
    PipelineMuse.from_pretrained(dir).to("cuda"), pipe("a red fox").  The images equal what the pre-computed-states entry gives for
    the states computed here from the same tower (penultimate layer + text_embeds, negative prompt "", :107-190); bf16 transformer"""
    import muse
    gp = np.load(os.path.join(golden_dir, "uvit_tiny.npz"))
    cfg = json.load(open(os.path.join(golden_dir, "config_uvit_tiny.json")))
    u = muse.MaskGiTUViT(**cfg)
    u.load_state_dict({k[len("param."):]: torch.from_numpy(gp[k]) for k in gp.files if k.startswith("param.")}, strict=True)
    enc, tok = W.tiny_clip(str(tmp_path / "clip_src"), hidden=cfg["encoder_hidden_size"], pooled=cfg["cond_embed_dim"])
    muse.PipelineMuse(vae=muse.MaskGitVQGAN(**W.VQGAN_TINY), transformer=u, text_encoder=enc, tokenizer=tok).save_pretrained(str(tmp_path / "ckpt"))
    pipe = muse.PipelineMuse.from_pretrained(str(tmp_path / "ckpt")).to(DEV, dtype=torch.float32)
    assert type(pipe.text_encoder).__name__ == "CLIPTextModelWithProjection" and next(pipe.text_encoder.parameters()).is_cuda
    prompts = ["a red fox", "two cats on a sofa"]
    gen = lambda: torch.Generator(device=DEV).manual_seed(5)   # noqa: E731
    kw = dict(timesteps=3, guidance_scale=1.5, output_type="np", transformer_seq_len=16)
    imgs = pipe(prompts, generator=gen(), **kw)
    assert imgs.shape == (2, 16, 16, 3) and np.isfinite(imgs).all()

    def states(texts):
        ids = pipe.tokenizer(texts, return_tensors="pt", padding="max_length", truncation=True, max_length=pipe.tokenizer.model_max_length).input_ids
        with torch.no_grad():
            o = pipe.text_encoder(ids.to(DEV), return_dict=True, output_hidden_states=True)
        return o.hidden_states[-2].float(), o.text_embeds.float()
    (h, pooled), (nh, npooled) = states(prompts), states(["", ""])
    want = pipe(prompt_embeds=h, pooled_embeds=pooled, negative_prompt_embeds=nh, negative_pooled_embeds=npooled, generator=gen(), **kw)
    assert np.array_equal(imgs, want)
    assert not np.array_equal(imgs, pipe(["a blue whale", "two cats on a sofa"], generator=gen(), **kw))       # the prompt matters


@pytest.mark.parametrize("resolution", [256, 512])
def test_the_reference_latency_benchmark_flow_runs_unchanged(golden_dir, tmp_path, resolution):
    """benchmark/muse_perf.py::muse_benchmark (the source of the reference's published latency table), statement for statement with tiny
    stand-ins for the hub checkpoint: AutoTokenizer / CLIPTextModelWithProjection from `<model>/text_encoder` cast to fp16,
    VQGANModel.from_pretrained(model, subfolder="vae").to(device, dtype=fp16), MaskGiTUViT(use_fused_mlp=False,
    use_fused_residual_norm=..., force_down_up_sample=resolution == 512).to(device, dtype=fp16).eval(), the xformers switch,
    PipelineMuse(tokenizer=, text_encoder=, vae=, transformer=) with `.device` / `.dtype` set by hand, pipe(prompt,
    num_images_per_prompt=, timesteps=, transformer_seq_len=) inside a torch.utils.benchmark Timer"""
    from torch.utils.benchmark import Timer
    from transformers import AutoTokenizer, CLIPTextModelWithProjection
    from muse import MaskGiTUViT, PipelineMuse, VQGANModel
    ucfg = json.load(open(os.path.join(golden_dir, "config_uvit_tiny.json")))
    enc, tok = W.tiny_clip(str(tmp_path / "clip_src"), hidden=ucfg["encoder_hidden_size"], pooled=ucfg["cond_embed_dim"])
    model = str(tmp_path / "hub_model")
    enc.save_pretrained(os.path.join(model, "text_encoder"))
    tok.save_pretrained(os.path.join(model, "text_encoder"))
    VQGANModel(**dict(W.TAMING_TINY, num_embeddings=ucfg["codebook_size"])).save_pretrained(os.path.join(model, "vae"))
    device, dtype = "cuda", torch.float16
    tokenizer = AutoTokenizer.from_pretrained(model, subfolder="text_encoder")
    text_encoder = CLIPTextModelWithProjection.from_pretrained(model, subfolder="text_encoder")
    text_encoder.to(device=device, dtype=dtype)
    vae = VQGANModel.from_pretrained(model, subfolder="vae")
    vae.to(device=device, dtype=dtype)
    tiny = {k: v for k, v in ucfg.items() if k not in ("force_down_up_sample", "use_fused_mlp", "use_fused_residual_norm")}
    transformer = MaskGiTUViT(use_fused_mlp=False, use_fused_residual_norm=True, force_down_up_sample=resolution == 512, **tiny)
    transformer = transformer.to(device=device, dtype=dtype)
    transformer.eval()
    transformer.enable_xformers_memory_efficient_attention()
    pipe = PipelineMuse(tokenizer=tokenizer, text_encoder=text_encoder, vae=vae, transformer=transformer)
    pipe.device = device
    pipe.dtype = dtype
    assert vae.compute_dtype == "bf16x3" and transformer.compute_dtype == torch.bfloat16 and next(transformer.parameters()).dtype == torch.float32
    seq_len = (resolution // 32) ** 2          # (tiny stand-in: 8 x 8 / 16 x 16 tokens where the real run has 16 x 16 / 32 x 32)
    prompt = "A high tech solarpunk utopia in the Amazon rainforest"
    out = pipe(prompt, num_images_per_prompt=2, timesteps=2, transformer_seq_len=seq_len)
    assert len(out) == 2 and out[0].size == (int(seq_len ** 0.5) * 4,) * 2          # PIL images, 4 pixels per token side for this tiny tokenizer
    t = Timer(stmt="benchmark_fn()", globals={"benchmark_fn": lambda: pipe(prompt, num_images_per_prompt=2, timesteps=3, transformer_seq_len=seq_len)}).timeit(2)
    assert t.mean > 0


def test_the_inpainting_log_script_flow(golden_dir, tmp_path):
    """scripts/log_inpainting_images.py of the reference: PipelineMuseInpainting.from_pretrained(model_name_or_path=,
    is_class_conditioned=).to(device=), the xformers switch on pipe.transformer, a numpy-built token mask moved to the device as bool, a
    PIL image resized to image_size, pipe(image=, mask=, class_ids= | text=, timesteps=, guidance_scale=, temperature=,
    use_maskgit_generate=, num_images_per_prompt=, image_size=) -> PIL images; class-conditional and text (real tiny CLIP tower)"""
    from PIL import Image
    import muse
    from muse import PipelineMuseInpainting
    rng = np.random.default_rng(3)
    picture = Image.fromarray((rng.random((40, 56, 3)) * 255).astype(np.uint8))
    # class-conditional checkpoint
    d = str(tmp_path / "cls")
    muse.PipelineMuse(vae=muse.MaskGitVQGAN(**W.VQGAN_TINY), transformer=muse.MaskGitTransformer(**W.TRANSFORMER_TINY),
                      is_class_conditioned=True).save_pretrained(d)
    pipe = PipelineMuseInpainting.from_pretrained(model_name_or_path=d, is_class_conditioned=True).to(device=DEV)
    pipe.transformer.enable_xformers_memory_efficient_attention()
    image_size, vae_scaling_factor = 16, 4
    class_ids = torch.tensor([7]).to(device=DEV, dtype=torch.long)
    mask = np.zeros((image_size // vae_scaling_factor, image_size // vae_scaling_factor))
    mask[1:3, 0:2] = 1
    mask = torch.tensor(mask.reshape(-1)).to(DEV, dtype=torch.bool)
    image = picture.resize((image_size, image_size))
    images = pipe(image=image, mask=mask, class_ids=class_ids, timesteps=3, guidance_scale=2.0, temperature=1.0, use_maskgit_generate=True,
                  num_images_per_prompt=2, image_size=image_size)
    assert len(images) == 2 and all(im.size == (16, 16) and im.mode == "RGB" for im in images)
    # text-conditioned checkpoint with its text encoder
    ucfg = json.load(open(os.path.join(golden_dir, "config_uvit_tiny.json")))
    enc, tok = W.tiny_clip(str(tmp_path / "clip_src"), hidden=ucfg["encoder_hidden_size"], pooled=ucfg["cond_embed_dim"])
    d = str(tmp_path / "txt")
    muse.PipelineMuse(vae=muse.MaskGitVQGAN(**W.VQGAN_TINY), transformer=muse.MaskGiTUViT(**ucfg), text_encoder=enc, tokenizer=tok).save_pretrained(d)
    pipe = PipelineMuseInpainting.from_pretrained(model_name_or_path=d, is_class_conditioned=False).to(device=DEV)
    pipe.transformer.enable_xformers_memory_efficient_attention()
    images = pipe(image=image, mask=mask, text="a small boat", timesteps=3, guidance_scale=2.0, temperature=1.0, use_maskgit_generate=True,
                  num_images_per_prompt=2, image_size=image_size)
    assert len(images) == 2 and all(im.size == (16, 16) for im in images)


def test_the_reference_benchmark_script_flow_runs_unchanged(golden_dir):
    """scripts/benchmark_models.py of the reference, statement for statement, on a reference-written config: load_config(path) ->
    from_config(config).to(device) -> eval() -> generate2(encoder_hidden_states=fp32 states) -> half() on the states and the model ->
    generate2 again -> enable_xformers_memory_efficient_attention() -> generate2; torch.utils.benchmark around one of them"""
    import torch.utils.benchmark as benchmark
    from muse import MaskGitTransformer
    config = MaskGitTransformer.load_config(os.path.join(golden_dir, "ckpt", "transformer_text_tiny"))
    model = MaskGitTransformer.from_config(config).to(DEV)
    model.eval()
    encoder_hidden_states = torch.randn(3, 9, model.config.encoder_hidden_size, device=DEV, dtype=torch.float32)
    ids = model.generate2(encoder_hidden_states=encoder_hidden_states, timesteps=4)
    assert ids.shape == (3, model.config.num_vq_tokens) and int(ids.min()) >= 0 and int(ids.max()) < model.config.codebook_size
    encoder_hidden_states = encoder_hidden_states.half()
    model = model.half()
    f = lambda: model.generate2(encoder_hidden_states=encoder_hidden_states, timesteps=4)      # noqa: E731
    ids16 = f()
    assert ids16.shape == ids.shape and int(ids16.max()) < model.config.codebook_size
    model.enable_xformers_memory_efficient_attention()
    t = benchmark.Timer(stmt="f()", globals={"f": f}).timeit(2)
    assert t.mean > 0 and f().shape == ids.shape


def test_inpainting_pipeline_repaints_only_the_masked_tokens(golden_dir):
    """muse.PipelineMuseInpainting (reference :372-510): the picture is tokenised by vae.encode, the masked positions get the mask token,
    generate2 fills them - every other token of every returned sample is the picture's own; class-conditional MaskGitTransformer (PIL
    input through the resize / centre-crop of the reference) and MaskGiTUViT on pre-computed text states (tensor input)"""
    import muse
    from PIL import Image
    from muse.pipeline_muse import _center_square
    v = muse.MaskGitVQGAN(**W.VQGAN_TINY)
    torch.manual_seed(2)
    m = muse.MaskGitTransformer(**W.TRANSFORMER_TINY)
    pipe = muse.PipelineMuseInpainting(vae=v, transformer=m, is_class_conditioned=True).to(DEV)
    m.eval()
    rng = np.random.default_rng(0)
    img = Image.fromarray((rng.random((24, 40, 3)) * 255).astype(np.uint8))
    px = _center_square(img, 16)
    assert px.shape == (3, 16, 16) and 0.0 <= float(px.min()) and float(px.max()) <= 1.0
    mask = torch.zeros(16, dtype=torch.bool)
    mask[[1, 5, 6, 11]] = True
    own = v.encode(px[None].to(DEV))[1].reshape(-1)
    seen = {}

    def spy_on(model):
        orig = model.generate2
        def spy(*a, **k):
            seen["in"] = k["input_ids"].clone()
            seen["out"] = orig(*a, **k)
            return seen["out"]
        model.generate2 = spy
    spy_on(m)
    out = pipe(img, mask, class_ids=3, timesteps=4, num_images_per_prompt=3, image_size=16, output_type="np",
               generator=torch.Generator(device=DEV).manual_seed(4))
    mask_id = m.config.mask_token_id
    assert out.shape == (3, 16, 16, 3) and seen["in"].shape == (3, 16)
    assert bool((seen["in"][:, mask] == mask_id).all()) and bool((seen["in"][:, ~mask] == own[~mask]).all())
    ids = seen["out"]
    assert bool((ids[:, ~mask.to(DEV)] == own[~mask.to(DEV)]).all()) and int(ids.min()) >= 0 and int(ids.max()) < W.VQGAN_TINY["num_embeddings"]
    assert np.array_equal(out, torch.clamp(v.decode_code(ids), 0, 1).permute(0, 2, 3, 1).cpu().numpy())
    with pytest.raises(ValueError):
        pipe(img, mask[:9], class_ids=3, image_size=16)
    # MaskGiTUViT, text states given, [h, w] mask, tensor image
    gp = np.load(os.path.join(golden_dir, "uvit_tiny.npz"))
    cfg = json.load(open(os.path.join(golden_dir, "config_uvit_tiny.json")))
    u = muse.MaskGiTUViT(**cfg)
    u.load_state_dict({k[len("param."):]: torch.from_numpy(gp[k]) for k in gp.files if k.startswith("param.")}, strict=True)
    pipe_u = muse.PipelineMuseInpainting(vae=v, transformer=u).to(DEV)
    u.eval()
    spy_on(u)
    g = torch.Generator().manual_seed(9)
    enc, pooled = torch.randn(2, 7, cfg["encoder_hidden_size"], generator=g), torch.randn(2, cfg["cond_embed_dim"], generator=g)
    out = pipe_u(px, mask.view(4, 4), prompt_embeds=enc, pooled_embeds=pooled, empty_embeds=torch.zeros(1, 7, cfg["encoder_hidden_size"]),
                 empty_pooled_embeds=torch.zeros(1, cfg["cond_embed_dim"]), timesteps=3, guidance_scale=1.5, output_type="np",
                 generator=torch.Generator(device=DEV).manual_seed(4))
    ids = seen["out"]
    assert out.shape == (2, 16, 16, 3) and ids.shape == (2, 16) and bool((ids[:, ~mask.to(DEV)] == own[~mask.to(DEV)]).all())
    assert int(ids.max()) < cfg["codebook_size"]


def test_uvit_generate2_vs_reference_golden(golden_dir):
    """MaskGiTUViT_v2.generate2 of the reference with classifier-free guidance 3.0, temperature (2, 0), 5 steps: final ids and the
    per-step raw samples (`intermediate`)"""
    import muse
    g = np.load(os.path.join(golden_dir, "uvit_generate2_tiny.npz"))
    gp = np.load(os.path.join(golden_dir, "uvit_tiny.npz"))
    cfg = json.load(open(os.path.join(golden_dir, "config_uvit_tiny.json")))
    m = muse.MaskGiTUViT(**cfg)
    m.load_state_dict({k[len("param."):]: torch.from_numpy(gp[k]) for k in gp.files if k.startswith("param.")}, strict=True)
    m.to(DEV).eval()
    T = int(g["timesteps"])
    args = [torch.from_numpy(g[k]).to(DEV) for k in ("encoder_hidden_states", "cond_embeds", "micro_conds", "empty_embeds",
                                                     "empty_cond_embeds")]
    ids, inter = m.generate2(*args, timesteps=T, temperature=tuple(float(x) for x in g["temperature"]),
                             guidance_scale=float(g["guidance_scale"]), return_intermediate=True, seq_len=int(g["seq"]),
                             noise=_noise(g, T))
    assert torch.equal(ids.cpu(), torch.from_numpy(g["ids"]))
    for i in range(T):
        assert torch.equal(inter[i].cpu(), torch.from_numpy(g[f"raw{i}"])), i


def test_uvit_decoding_graph_is_kept_and_reused_across_calls(golden_dir):
    """MaskGiTUViT.generate2(hip_graph=True) keeps the captured forward: a second call with the same shapes and OTHER conditioning copies
    its inputs into the graph's static buffers and replays - same ids as the eager loop with the same seed; a weight update that
    bumps autograd's version counters (torch optimizer, load_state_dict) or mark_weights_changed() re-captures; muse.PipelineMuse
    turns the graph on by itself for small decoding batches"""
    import muse
    gp = np.load(os.path.join(golden_dir, "uvit_tiny.npz"))
    gu = np.load(os.path.join(golden_dir, "uvit_generate2_tiny.npz"))
    u = muse.MaskGiTUViT(**json.load(open(os.path.join(golden_dir, "config_uvit_tiny.json"))))
    u.load_state_dict({k[len("param."):]: torch.from_numpy(gp[k]) for k in gp.files if k.startswith("param.")}, strict=True)
    u.to(DEV).eval()
    base = [torch.from_numpy(gu[k]).to(DEV) for k in ("encoder_hidden_states", "cond_embeds", "micro_conds", "empty_embeds", "empty_cond_embeds")]
    kw = dict(timesteps=4, temperature=(2.0, 0.0), guidance_scale=3.0, seq_len=int(gu["seq"]))

    def both(args, seed):
        a = u.generate2(*args, generator=torch.Generator(device=DEV).manual_seed(seed), hip_graph=False, **kw)
        b = u.generate2(*args, generator=torch.Generator(device=DEV).manual_seed(seed), hip_graph=True, **kw)
        assert torch.equal(a, b)
        return u.__dict__["_gen_graph"]["bufs"][4]

    g1 = both(base, 1)
    other = [base[0] * 0.5 + 0.1, base[1] * -1.0, base[2], base[3], base[4]]
    g2 = both(other, 2)
    assert g2 is g1                                   # replayed, not re-captured
    with torch.no_grad():
        u.mlm_layer.conv2.weight.mul_(1.5)            # an in-place update autograd sees (what a torch optimizer does)
    g3 = both(base, 3)
    assert g3 is not g1
    u.mark_weights_changed()
    assert u.__dict__.get("_gen_graph") is None


def test_generate2_hip_graph_matches_eager(golden_dir):
    """hip_graph=True (forward captured once, replayed per step) gives the reference-golden ids of the eager loop for both
    models (f32 compute: recorded draws); config B at batch 2 in bf16: both loops reproducible from a seed"""
    import time
    import muse
    g = np.load(os.path.join(golden_dir, "generate2_tiny.npz"))
    cfg = W.TRANSFORMER_TINY
    m = muse.MaskGitTransformer(**cfg)
    m.load_state_dict(W.fill_state_dict(W.transformer_shapes(cfg), int(g["seed"]), "transformer"))
    m.to(DEV).eval().set_compute_dtype(torch.float32)
    T = int(g["timesteps"])
    ids = m.generate2(class_ids=torch.from_numpy(g["class_ids"]).to(DEV), timesteps=T, temperature=float(g["temperature"]),
                      noise=_noise(g, T), hip_graph=True)
    assert torch.equal(ids.cpu(), torch.from_numpy(g["ids"]))
    gu = np.load(os.path.join(golden_dir, "uvit_generate2_tiny.npz"))
    gp = np.load(os.path.join(golden_dir, "uvit_tiny.npz"))
    u = muse.MaskGiTUViT(**json.load(open(os.path.join(golden_dir, "config_uvit_tiny.json"))))
    u.load_state_dict({k[len("param."):]: torch.from_numpy(gp[k]) for k in gp.files if k.startswith("param.")}, strict=True)
    u.to(DEV).eval()
    Tu = int(gu["timesteps"])
    args = [torch.from_numpy(gu[k]).to(DEV) for k in ("encoder_hidden_states", "cond_embeds", "micro_conds", "empty_embeds",
                                                      "empty_cond_embeds")]
    ids_u = u.generate2(*args, timesteps=Tu, temperature=tuple(float(x) for x in gu["temperature"]),
                        guidance_scale=float(gu["guidance_scale"]), seq_len=int(gu["seq"]), noise=_noise(gu, Tu), hip_graph=True)
    assert torch.equal(ids_u.cpu(), torch.from_numpy(gu["ids"]))
    # the benched transformer (config B), bf16, batch 2, 18 steps: each loop reproduces its own ids from the same seed; wall time of both.
    # (Since round 4 the CAPTURED forward cuts its small-batch Linears along K - ops.gemm's decoding path, taken only inside graph capture
    #  because it trades kernel time for launches - so in bf16 the graph loop and the eager loop sum those products in different (fixed)
    #  orders and their samples are two valid draws, not the same one; in f32 compute the two loops stay bit-identical: the goldens above.)
    big = muse.MaskGitTransformer(**W.TRANSFORMER_B)
    big.to(DEV).eval().set_compute_dtype(torch.bfloat16)
    res = {}
    for mode in (False, True):
        outs = []
        for rep in range(3):   # last repetition timed (the first pays one-time packing and builds the graph, kept by the model)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = big.generate2(class_ids=torch.tensor([3, 700], device=DEV), timesteps=18, temperature=2.0,
                                generator=torch.Generator(device=DEV).manual_seed(11), hip_graph=mode)
            torch.cuda.synchronize()
            outs.append(out)
            res[mode] = (out, time.perf_counter() - t0)
        assert torch.equal(outs[1], outs[2]) and torch.equal(outs[0], outs[1])
        assert int(out.min()) >= 0 and int(out.max()) < W.TRANSFORMER_B["codebook_size"]
    print(f"generate2 config B bs 2, 18 steps: eager {res[False][1] * 1e3:.1f} ms, hip_graph {res[True][1] * 1e3:.1f} ms")


class _Cfg(dict):
    __getattr__ = dict.__getitem__


@pytest.mark.parametrize("case", ["default", "predict_all", "random_replace", "region", "eval_ratios"])
def test_mask_or_random_replace_tokens_vs_reference(golden_dir, case):
    """muse.mask_or_random_replace_tokens (device kernel + the reference's host-side `random` calls) == the reference function"""
    import muse
    g = np.load(os.path.join(golden_dir, "mask_muse.npz"))
    tokens = torch.from_numpy(g["tokens"]).to(DEV)
    tr = {"default": dict(min_masking_rate=0.1), "predict_all": dict(min_masking_rate=0.0, predict_all_tokens=True),
          "random_replace": dict(min_masking_rate=0.25, noise_type="random_replace"),
          "region": dict(min_masking_rate=0.0, mask_contiguous_region_prob=1.0),
          "eval_ratios": dict(min_masking_rate=0.0, eval_mask_ratios=[0.2, 0.55, 0.9])}[case]
    cfg = _Cfg(training=_Cfg(tr), model=_Cfg(codebook_size=int(g["codebook_size"])))
    i = ["default", "predict_all", "random_replace", "region", "eval_ratios"].index(case)
    random.seed(540 + i)                                  # the seed make_golden.py gave Python's `random` for this case
    from muse.sampling import cosine_schedule
    ids, labels, lw, mp = muse.mask_or_random_replace_tokens(
        tokens, int(g["mask_id"]), cfg, cosine_schedule, is_train=case != "eval_ratios",
        timesteps=torch.from_numpy(g[case + ".timesteps"]).to(DEV), noise=torch.from_numpy(g[case + ".noise"]).to(DEV))
    assert torch.equal(ids.cpu(), torch.from_numpy(g[case + ".input_ids"])) and torch.equal(labels.cpu(), torch.from_numpy(g[case + ".labels"]))
    # mask_prob is a logged float: the device evaluates cos() correctly rounded (f64), torch's CPU cosf may differ in the last ulp
    assert torch.allclose(mp.cpu(), torch.from_numpy(g[case + ".mask_prob"]), rtol=3e-7, atol=0)
    if case in ("predict_all", "random_replace"):
        assert torch.allclose(lw.cpu(), torch.from_numpy(g[case + ".loss_weight"]), rtol=3e-7, atol=0)
    else:
        assert lw is None


def test_mask_tokens_full_size_properties():
    """seq 1024 (the 512^2 text-to-image regime), batch 64: every image has exactly round(S * cos(pi/2 t)) masked positions"""
    ops = _ops()
    B, S = 64, 1024
    g = torch.Generator().manual_seed(3)
    tok = torch.randint(0, 8192, (B, S), generator=g)
    t, nz = torch.rand(B, generator=g), torch.rand(B, S, generator=g)
    ids, labels, lw, mp = ops.mask_tokens(tok.to(DEV), 8255, timesteps=t.to(DEV), noise=nz.to(DEV), min_masking_rate=0.05,
                                          all_labels=True, want_weight=True)
    k = (S * torch.cos(t * (np.pi * 0.5)).clip(0.05)).round().clamp(min=1).long()
    assert torch.equal((ids.cpu() == 8255).sum(-1), k) and torch.equal(labels.cpu(), tok)
    assert torch.equal(ids.cpu() == 8255, lw.cpu() == 1.0)
    assert float(lw.min()) >= 0.3 - 1e-6


def test_cond_dropout_vs_reference(golden_dir):
    import muse
    g = np.load(os.path.join(golden_dir, "mask_muse.npz"))
    t = lambda k: torch.from_numpy(g[k]).to(DEV)   # noqa: E731
    enc, clip = muse.cond_dropout(t("cd.enc"), t("cd.clip"), t("cd.empty"), t("cd.empty_clip"), float(g["cd.prob"]), uniforms=t("cd.u"))
    assert torch.equal(enc.cpu(), torch.from_numpy(g["cd.enc_out"])) and torch.equal(clip.cpu(), torch.from_numpy(g["cd.clip_out"]))
