#!/usr/bin/env python
"""bench.py — the reference's headline metric on MI355X: images/sec/node of one MaskGit train step
(training/train_maskgit_imagenet.py:405-452: VQGAN encode -> cosine-schedule mask -> MaskGitTransformer fwd/bwd ->
AdamW), 256x256 synthetic images, bs=64 per GPU, random-init weights, 1..8 GPUs data-parallel over RCCL.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0.  Default workload = BASELINE.json configs[1]: configs/imagenet.yaml transformer
(hidden 768, 24 layers, 16 heads, vocab 2048 — SURVEY.md D1 "B"), bf16 compute for the transformer (the reference's
autocast regime) and the frozen VQGAN tokenizer in its f32-class "bf16x3" mode: f32 activations, every product computed
as 3 bf16 MFMAs with f32 accumulation (error <= 2^-16 per product).  The reference keeps the VQGAN in f32 tensors, and
its GPU path runs those convolutions in TF32 (torch.backends.cudnn.allow_tf32 defaults to True, configs/imagenet.yaml:86
enable_tf32); gfx950 has no xf32 MFMA, bf16x3 is the tighter CDNA4 counterpart (token indices equal to the f32 oracle up
to f32 near-ties: tests/test_gpu_models.py::test_vq_indices_over_bench_batch_vs_oracle; the count over this run's CPU-baseline
images is in cpu_baseline.vq_index_mismatches).

The line carries
  roofline      the dominant kernel (live HIP-event timing of every launch of one instrumented step) against its roof; `traffic`
                = HBM bytes per launch of that kernel from the committed PMC passes (profiles/r06_traffic.json, FETCH_SIZE x 2
                + WRITE_SIZE as /opt/skills/guides/MI355X_MICROARCH.md prescribes), null when no such profile is in the tree
  cpu_baseline  the CPU oracle (port of the reference path) on the node's host cores: warm-up + median of 3, per phase
  extra         the north_star's own targets: transformer_mfma_frac (3 x forward GFLOP / transformer fwd+bwd time / 2.5 PF),
                vqgan_hbm_frac (GroupNorm / pooling kernels vs 8 TB/s), the tokens-given step (pre-encoded tokens,
                scripts/pre_encode.py regime), config 4 (MaskGiTUViT, seq 256 and 1024), config 5 (VQGAN encode -> decode), config A
"""
import argparse
import gc
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "open-muse_amd"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK = {"bf16": 2500.0, "f32": 157.3}  # dense MFMA TFLOP/s, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK = 8000.0                      # GB/s (spec; ~6300 achievable per the same guide)
# algorithmic work per image, GFLOP
GF_ENCODE = 128.63
GF_FWD = {"A": 19.00, "B": 122.40}
GF_UVIT_FWD = {256: 275.10, 1024: 1137.05}
# BASELINE.json config 4: configs/cc12m_uvit_clip.yaml:29-54 `model.transformer` + block_num_heads=16;
# 728 725 504 parameters - GF_UVIT_FWD above is THIS geometry's forward work (tests/test_surface.py keeps it equal to the golden config)
UVIT_CC12M = dict(
    vocab_size=8256, hidden_size=1024, intermediate_size=4096, num_hidden_layers=22, num_attention_heads=16,
    max_position_embeddings=256, in_channels=512, block_out_channels=(1024,), num_res_blocks=3, patch_size=1,
    encoder_hidden_size=768, add_cross_attention=True, project_encoder_hidden_states=False, codebook_size=8192, num_vq_tokens=256,
    initializer_range=0.02, norm_type="rmsnorm", layer_norm_eps=1e-6, use_normformer=False, use_encoder_layernorm=True,
    use_bias=False, hidden_dropout=0.0, attention_dropout=0.0, use_codebook_size_for_output=True, block_num_heads=16,
)
TRAFFIC_JSON = next((f for f in (os.path.join(ROOT, "profiles", n) for n in ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json")) if os.path.exists(f)),
                    os.path.join(ROOT, "profiles", "r06_traffic.json"))
# rocprof kernel name fragment of each instrumented kernel family (to look its counters up in TRAFFIC_JSON)
KERNEL_OF = {"conv_bf16x3_dma": "cslab::conv_slab_kernel<true>", "gemm_bf16_NN": "g256p::kernel<unsigned short, 0, 0", "gemm_bf16_NT": "g256p::kernel<unsigned short, 0, 1",
             "gemm_bf16_TT": "g256::kernel_group<float, 1, 1", "conv_bf16x3": "conv_split_kernel", "attn_fwd_bf16": "attn2::fwd_kernel"}


def build_models(cfg_name, vq_dtype, device, seed, transformer_dtype=torch.bfloat16):
    import muse
    import weights as W
    tcfg = dict(W.TRANSFORMER_A if cfg_name == "A" else W.TRANSFORMER_B)
    torch.manual_seed(seed)  # identical init on every rank (the reducer also broadcasts rank 0's weights)
    vq = muse.MaskGitVQGAN(**W.VQGAN_F16)
    # random init; conv weights scaled so activations stay O(1) through 28 convs (keeps f32/bf16 comparable)
    vq.load_state_dict(W.fill_state_dict(W.vqgan_shapes(W.VQGAN_F16), seed, "vqgan"))
    vq.requires_grad_(False)
    vq.to(device).eval().set_compute_dtype({"f32": torch.float32, "bf16": torch.bfloat16, "bf16x3": "bf16x3"}[vq_dtype])
    model = muse.MaskGitTransformer(**tcfg)
    model.to(device).train().set_compute_dtype(transformer_dtype)
    opt = muse.FusedAdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.999), weight_decay=0.01, eps=1e-8)
    return vq, model, opt, tcfg


def synthetic_batch(bs, device, seed):
    g = torch.Generator().manual_seed(seed)
    px = torch.rand(bs, 3, 256, 256, generator=g)
    cls = torch.randint(0, 1000, (bs,), generator=g)
    return px.to(device), cls.to(device)


def vqgan_roundtrip(device, bs):
    """BASELINE.json config 5: MaskGitVQGAN f16-256 encode -> decode_code throughput (315.3 GFLOP / image, algorithmic HBM
    bytes 990.6 MB / image in f32: BASELINE.md section 3) for the f32-class (bf16x3) and exact-f32 tokenizer modes"""
    import muse
    import weights as W
    out = {}
    vq = muse.MaskGitVQGAN(**W.VQGAN_F16)
    vq.load_state_dict(W.fill_state_dict(W.vqgan_shapes(W.VQGAN_F16), 1234, "vqgan"))
    vq.to(device).eval()
    px, _ = synthetic_batch(bs, device, seed=77)
    for mode, name in (("bf16x3", "bf16x3"), (torch.float32, "f32")):
        vq.set_compute_dtype(mode)
        for _ in range(2):
            vq.decode_code(vq.get_code(px))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 3
        for _ in range(n):
            vq.decode_code(vq.get_code(px))
        torch.cuda.synchronize()
        ips = bs * n / (time.perf_counter() - t0)
        out[f"vqgan_encode_decode_images_per_s_{name}"] = round(ips, 1)
        out[f"vqgan_encode_decode_tflops_{name}"] = round(ips * 315.3 / 1e3, 1)
        out[f"vqgan_encode_decode_algorithmic_GBps_{name}"] = round(ips * 990.6 / 1e3, 1)
    del vq
    torch.cuda.empty_cache()
    return out


def leg_isolated(spec, fallback):
    """One secondary leg in a process of its own (`bench.py --leg SPEC` prints a JSON dict).  Late in this long-lived process the same
    legs measure 15-25 % slower than in a fresh one (tokens-given step: 46 ms here, 38.4 ms fresh - scripts/exp/host_bound.py); a
    training job is a fresh process, so that is what each leg gets.  `fallback()` computes it in-process if the child fails."""
    import subprocess
    try:
        torch.cuda.empty_cache()
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
        cmd = [sys.executable, os.path.abspath(__file__), "--leg", spec] + (["--no-prefetch"] if "--no-prefetch" in sys.argv else [])
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
        return json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    except Exception as e:   # noqa: BLE001  (a leg of `extra` must never take the bench line down)
        print(f"bench: leg {spec!r} fell back to this process ({type(e).__name__}: {e})", file=sys.stderr)
        return fallback()


def uvit_leg_isolated(device, batch, seq, steps, f32=False, x3=False, f16=False):
    """uvit_leg in a fresh process.  The U-ViT step is ~4500 small launches; at the end of this long-lived process (allocator state,
    Python heap of all the earlier legs) the same leg measured 205 ms per step against 163 ms in a process of its own, which is what a
    training job is - so it gets one (GPU memory of this process has been released by then)."""
    import subprocess
    try:
        torch.cuda.empty_cache()
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--uvit-leg", f"{batch},{seq},{steps}" + (",f32" if f32 else "") + (",x3" if x3 else "") + (",f16" if f16 else "")], capture_output=True, text=True,
                           timeout=600, env=env)
        line = [l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1]
        return json.loads(line)
    except Exception as e:   # noqa: BLE001  (a leg of `extra` must never take the bench line down)
        print(f"bench: config-4 leg {batch},{seq} did not run in a subprocess ({type(e).__name__}); running it in-process", file=sys.stderr)
        try:
            out = uvit_leg(device, batch, seq, steps, f32, x3, f16)
            out["note"] = f"in-process (subprocess failed: {type(e).__name__})"
        except Exception as e2:   # noqa: BLE001
            torch.cuda.empty_cache()
            out = {"error": f"{type(e2).__name__}: {str(e2)[:200]}", "batch": batch, "seq_len": seq}
        return out


def taming_leg(device, bs):
    """the tokenizer of the text-to-image configs (configs/cc12m_uvit_clip.yaml:19-21: taming VQGANModel f16, 8192 codes):
    get_code throughput (what scripts/pre_encode.py:440-511 runs per batch) and encode -> decode_code, bf16x3 mode"""
    import muse
    import weights as W
    cfg = dict(W.VQGAN_F16, num_embeddings=8192, attn_resolutions=(16,), no_attn_mid_block=False, resample_with_conv=True)
    vq = muse.VQGANModel(**cfg)
    vq.load_state_dict(W.fill_state_dict(W.taming_shapes(cfg), 4321, "vqgan"))
    vq.to(device).eval().set_compute_dtype("bf16x3")
    px, _ = synthetic_batch(bs, device, seed=78)
    out = {}
    for name, fn in (("encode", lambda: vq.get_code(px)), ("encode_decode", lambda: vq.decode_code(vq.get_code(px)))):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 3
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        out[f"taming_vqgan_f16_8192_{name}_images_per_s_bf16x3"] = round(bs * n / (time.perf_counter() - t0), 1)
    del vq
    torch.cuda.empty_cache()
    return out


def latency_leg(device, timesteps=12):
    """the metric the reference PUBLISHES (benchmark/muse_perf.py:241-293, benchmark/artifacts/all.csv:5,39: 507.6 / 756.2 ms on an
    A100, fp16): end-to-end latency of muse.PipelineMuse with a default-constructed MaskGiTUViT (random init, 603.5 M parameters),
    12 decoding steps at 256 x 256 (256 tokens), classifier-free guidance at the pipeline's default scale 10 (every forward runs a
    doubled batch), then the taming f16 / 8192-code VQGANModel decoder; batch 1 and 8.  Differences, stated: the text states are
    pre-computed synthetic CLIP states (77 x 768 + pooled 768; the reference's figure includes its CLIP text encoder forward, a few
    ms), compute is bf16 instead of fp16 (the HIP kernels have no f16 variant), the decoder runs its f32-class bf16x3 mode.  Also the
    time of ONE transformer forward at the decoding batch (2 x bs rows of 256 tokens): 12 of them are the loop."""
    import muse
    import weights as W
    vcfg = dict(W.VQGAN_F16, num_embeddings=8192, attn_resolutions=(16,), no_attn_mid_block=False, resample_with_conv=True)
    vae = muse.VQGANModel(**vcfg)
    vae.load_state_dict(W.fill_state_dict(W.taming_shapes(vcfg), 4321, "vqgan"))
    vae.eval().set_compute_dtype("bf16x3")     # f32 tensors, f32-class products: the mode this leg's description names (until round 4 the leg left
    from muse import modeling_transformer_v2 as M   # the decoder in its exact-f32 default: 16.4 / 32 ms per decode instead of 4.4 / 8.1)
    init = M.MaskGiTUViT_v2._init_weights
    M.MaskGiTUViT_v2._init_weights = lambda self: None     # filled on the GPU below instead of on one CPU core
    try:
        tr = muse.MaskGiTUViT()
    finally:
        M.MaskGiTUViT_v2._init_weights = init
    pipe = muse.PipelineMuse(vae=vae, transformer=tr)
    pipe.to(device, dtype=torch.bfloat16)
    tr.eval()
    g = torch.Generator(device=device).manual_seed(0)
    with torch.no_grad():
        for n, p in tr.named_parameters():
            p.fill_(1.0) if n.endswith("norm.weight") else p.normal_(0.0, 0.02, generator=g)
    tr.mark_weights_changed()
    out = {"model": f"MaskGiTUViT() defaults, {sum(p.numel() for p in tr.parameters()) / 1e6:.1f} M parameters, bf16 compute; taming VQGAN "
                    f"f16-8192 decoder bf16x3; {timesteps} steps, guidance 10 (doubled batch), 256 x 256, synthetic text states; decoding forward "
                    f"captured into a HIP graph kept across calls (PipelineMuse default for <= 4096 rows)",
           "reference_published_ms": {"bs1_a100_fp16": 507.6, "bs8_a100_fp16": 756.2, "source": "benchmark/artifacts/all.csv:5,39"}}
    for bs in (1, 8):
        enc = torch.randn(1, 77, 768, device=device, generator=g)
        pooled = torch.randn(1, 768, device=device, generator=g)
        empty, empty_p = torch.randn(1, 77, 768, device=device, generator=g), torch.randn(1, 768, device=device, generator=g)

        def call(steps):
            return pipe(prompt_embeds=enc, pooled_embeds=pooled, empty_embeds=empty, empty_pooled_embeds=empty_p, num_images_per_prompt=bs,
                        timesteps=steps, transformer_seq_len=256, orig_size=(256, 256), output_type="np", use_tqdm=False,
                        generator=torch.Generator(device=device).manual_seed(1))
        call(2)
        call(timesteps)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            call(timesteps)                      # (returns host numpy images: the call is synchronous like the reference's PIL output)
            ts.append((time.perf_counter() - t0) * 1e3)
        out[f"pipeline_ms_bs{bs}"] = round(statistics.median(ts), 1)
        # one forward at the decoding batch
        ids = torch.full((2 * bs, 256), tr.config.mask_token_id, dtype=torch.long, device=device)
        e2, p2 = enc.expand(2 * bs, -1, -1).contiguous(), pooled.expand(2 * bs, -1).contiguous()
        micro = torch.tensor([[256.0, 256.0, 0.0, 0.0, 6.0]], device=device).repeat(2 * bs, 1)
        with torch.no_grad():
            for _ in range(3):
                tr(ids, e2, p2, micro)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                tr(ids, e2, p2, micro)
            torch.cuda.synchronize()
        out[f"forward_ms_rows{2 * bs}x256"] = round((time.perf_counter() - t0) * 100, 2)
        toks = torch.randint(0, 8192, (bs, 256), device=device, generator=g)
        with torch.no_grad():
            vae.decode_code(toks)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                vae.decode_code(toks)
            torch.cuda.synchronize()
        out[f"decode_ms_bs{bs}"] = round((time.perf_counter() - t0) / 3 * 1e3, 2)
    # the 512 x 512 rows of the same table (benchmark/artifacts/all.csv:7,41: 572.2 / 1172.4 ms): muse_perf.py builds the transformer
    # with force_down_up_sample=True for them (:257) - 1024 tokens, 256 inside the blocks and layers - and decodes a 32 x 32 token grid
    del pipe, tr
    torch.cuda.empty_cache()
    try:
        M.MaskGiTUViT_v2._init_weights = lambda self: None
        try:
            tr = muse.MaskGiTUViT(force_down_up_sample=True)
        finally:
            M.MaskGiTUViT_v2._init_weights = init
        pipe = muse.PipelineMuse(vae=vae, transformer=tr)
        pipe.to(device, dtype=torch.bfloat16)
        tr.eval()
        with torch.no_grad():
            for n, p in tr.named_parameters():
                p.fill_(1.0) if n.endswith("norm.weight") else p.normal_(0.0, 0.02, generator=g)
        tr.mark_weights_changed()
        out["reference_published_ms"].update({"bs1_512_a100_fp16": 572.2, "bs8_512_a100_fp16": 1172.4, "source_512": "benchmark/artifacts/all.csv:7,41"})
        out["model_512"] = (f"MaskGiTUViT(force_down_up_sample=True), {sum(p.numel() for p in tr.parameters()) / 1e6:.1f} M parameters, 1024 tokens "
                            f"(256 inside), 512 x 512 decode; otherwise as above")
        for bs in (1, 8):
            enc = torch.randn(1, 77, 768, device=device, generator=g)
            pooled = torch.randn(1, 768, device=device, generator=g)
            empty, empty_p = torch.randn(1, 77, 768, device=device, generator=g), torch.randn(1, 768, device=device, generator=g)

            def call512(steps):
                return pipe(prompt_embeds=enc, pooled_embeds=pooled, empty_embeds=empty, empty_pooled_embeds=empty_p, num_images_per_prompt=bs,
                            timesteps=steps, transformer_seq_len=1024, orig_size=(512, 512), output_type="np", use_tqdm=False,
                            generator=torch.Generator(device=device).manual_seed(1))
            call512(2)
            call512(timesteps)
            torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                call512(timesteps)
                ts.append((time.perf_counter() - t0) * 1e3)
            out[f"pipeline_ms_bs{bs}_512"] = round(statistics.median(ts), 1)
    except Exception as e:   # noqa: BLE001  (a leg of `extra` must never take the bench line down)
        out["error_512"] = f"{type(e).__name__}: {str(e)[:200]}"
    del pipe, tr, vae
    torch.cuda.empty_cache()
    return out


def uvit_leg(device, batch, seq, steps=3, f32=False, x3=False, f16=False):
    """BASELINE.json config 4: configs/cc12m_uvit_clip.yaml MaskGiTUViT (UVIT_CC12M: 728.7 M parameters, 22 layers, hidden 1024, GLU 4096,
    1024-channel ResBlock / attention stages; block_num_heads 16 per SURVEY.md D3), synthetic CLIP states (77 x 768), tokens given,
    bf16 compute (fused self / cross attention, bf16 weight copies refreshed inside the AdamW kernel): forward + backward + FusedAdamW"""
    import muse
    from muse import modeling_transformer_v2 as M
    init = M.MaskGiTUViT_v2._init_weights
    M.MaskGiTUViT_v2._init_weights = lambda self: None     # 729 M parameters: filled on the GPU below instead of on one CPU core
    try:
        model = muse.MaskGiTUViT(**UVIT_CC12M)
    finally:
        M.MaskGiTUViT_v2._init_weights = init
    n_params = sum(p.numel() for p in model.parameters())
    assert n_params == 728725504, n_params          # the geometry GF_UVIT_FWD was counted on
    model.to(device).train().set_compute_dtype("f16" if f16 else "bf16x3" if x3 else (torch.float32 if f32 else torch.bfloat16))
    g = torch.Generator(device=device).manual_seed(0)
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.fill_(1.0) if n.endswith("norm.weight") else p.normal_(0.0, 0.02, generator=g)
    # the optimizer as training/train_muse.py:425-445 builds it: two parameter groups, no weight decay on bias / layer_norm.weight /
    # mlm_ln.weight / embeddings.weight (one muse_adamw_multi_groups launch)
    opt = muse.FusedAdamW(muse.grouped_parameters(model, 0.01), lr=1e-4, betas=(0.9, 0.999), weight_decay=0.01, eps=1e-8)
    ids = torch.randint(0, 8256, (batch, seq), device=device, generator=g)
    labels = torch.where(torch.rand(batch, seq, device=device, generator=g) < 0.5,
                         torch.randint(0, 8192, (batch, seq), device=device, generator=g), torch.full((batch, seq), -100, device=device))
    enc = torch.randn(batch, 77, 768, device=device, generator=g)
    cond = torch.randn(batch, 768, device=device, generator=g)
    micro = torch.tensor([[256.0, 256.0, 0.0, 0.0, 6.0]], device=device).repeat(batch, 1)

    def step():
        model.zero_grad(set_to_none=True)
        _, loss = model(ids, enc, cond, micro, labels=labels)
        loss.backward()
        opt.step()
        return loss
    step()
    step()   # two warm-up steps: the first touch of tens of GB of fresh HBM (page tables, allocator growth) must not land in the timed ones
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    tf = 3 * GF_UVIT_FWD[seq] * batch / dt / 1e3
    out = {"images_per_s": round(batch / dt, 1), "ms_per_step": round(dt * 1e3, 1), "batch": batch, "seq_len": seq,
           # against configs/cc12m_uvit_clip.yaml:102-103 (mixed_precision "no" + enable_tf32: 10-bit-mantissa products, f32 accumulate).
           # "narrower" legs are engineering figures, NOT config 4's number
           "precision_vs_yaml": "wider" if f32 else ("class-equal" if (x3 or f16) else "narrower"),
           "tflops": round(tf, 1), "mfma_frac": round(tf / (PEAK["bf16"] / 3 if x3 else PEAK["f32" if f32 else "bf16"]), 4), "loss": round(float(loss), 4), "parameters": n_params,
           "dtype": ("f16: f32 tensors; every weight GEMM (linears, dX, dW) as ONE v_mfma_f32_16x16x32_f16 product of IEEE-half operand images with f32 "
                     "accumulation - half's 10-bit mantissa is the TF32 operand format the yaml's enable_tf32 multiplies in (gfx950 has no xf32 MFMA); "
                     "(against the real reference's f32 run at full size: logits 9.2e-4, worst gradient 2.2e-3; the reference under emulated TF32: 1.0e-3 / 2.1e-3 - "
                     "tests/golden/make_golden_tf32.py); TF32's exponent range is covered by power-of-two operand scales (gradient operands x 2^10 x tokens, undone in alpha; "
                     "overflowed / flushed elements counted: f16_operand_stats); the attention core (muse_attention_x3_*, block by block at 1024 tokens) "
                     "with one half plane per operand and one half MFMA per K step; f32 softmax, norms, GLU, residual stream, loss, AdamW; mfma_frac against the 2500 TFLOP/s half / bf16 peak") if f16 else ("bf16x3: f32 tensors, every product (linears, dX, dW: muse_gemm_x3 on four operand planes; the attention core: "
                     + ("muse_attention_x3_*" if seq == 256 else "muse_attention_x3_*_stream - 256 queries / keys per workgroup, the other side's 256-row blocks streamed "
                                                               "through LDS, online softmax (the 77 text states: one-tile kernels per query block)")
                     + ") as three bf16 MFMA products of hi / lo operand planes with f32 accumulation (<= 2^-16 relative per "
                     "product: at or above the yaml's mixed_precision: no + enable_tf32, 10-bit mantissa products); f32 softmax, norms, GLU, "
                     "residual stream, loss, AdamW; mfma_frac against the 833 TFLOP/s roof of that scheme (2500 / 3)") if x3 else ("exact f32 everywhere (f32-input MFMA, 157 TFLOP/s peak): at or above the precision of the yaml's mixed_precision: no + "
                     "enable_tf32 (10-bit mantissa products); gfx950 has no xf32 MFMA") if f32 else
                    "bf16 weight-GEMM / attention operands, f32 accumulate, residual stream, norms, loss (the yaml itself sets mixed_precision: no)",
           "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}
    if f16:
        sat, flushed = model.f16_stats()
        out["f16_operand_stats"] = {"overflowed": sat, "rounded_to_zero": flushed, "over": f"the {steps + 2} steps of this leg"}
    del model, opt
    torch.cuda.empty_cache()
    return out


def cpu_baseline(cfg_name, device, bs=4, reps=3, bench_batch=64):
    """the CPU oracle (port of the reference path: BASELINE.json configs[0], bs = 4) on this node's host cores: one warm-up step,
    then `reps` timed steps with per-phase times; median reported.  Each repetition tokenizes different images; the oracle's token
    indices are compared with the HIP tokenizer's (the bench's bf16x3 mode) on the same images: vq_index_mismatches."""
    import muse
    import weights as W
    from oracle import maskgit_oracle as O
    cores = min(os.cpu_count(), 32)  # torch CPU ops stop scaling (and oversubscribe) far below the 256 hw threads of the node
    cpu_model = "unknown"
    try:
        cpu_model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except Exception:   # noqa: BLE001
        pass
    torch.set_num_threads(cores)
    tcfg = dict(W.TRANSFORMER_A if cfg_name == "A" else W.TRANSFORMER_B)
    vsd = W.fill_state_dict(W.vqgan_shapes(W.VQGAN_F16), 1, "vqgan")
    tsd = W.fill_state_dict(W.transformer_shapes(tcfg), 2, "transformer")
    mom = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in tsd.items()}
    vq = muse.MaskGitVQGAN(**W.VQGAN_F16)
    vq.load_state_dict(vsd)
    vq.to(device).eval().set_compute_dtype("bf16x3")
    phases, mism, ntok = [], 0, 0
    for r in range(reps + 1):
        px, cls = W.images(bs, 256, 30 + r), torch.from_numpy(np.random.default_rng(40 + r).integers(0, 1000, size=bs))
        t, nz = W.uniforms((bs,), 50 + r), W.uniforms((bs, 256), 60 + r)
        t0 = time.perf_counter()
        with torch.no_grad():
            _, _, tokens = O.vqgan_encode(vsd, W.VQGAN_F16, px)
            ids, labels, _ = O.prepare_inputs_and_labels(tokens, cls, t, nz, int(tcfg["vocab_size"]) - 1, 1024)
        t1 = time.perf_counter()
        leaf = {k: v.detach().clone().requires_grad_(True) for k, v in tsd.items()}
        _, loss = O.transformer_forward(leaf, tcfg, ids, labels, 0.0)
        t2 = time.perf_counter()
        loss.backward()
        t3 = time.perf_counter()
        with torch.no_grad():
            for k, p in tsd.items():
                O.adamw_step(p, leaf[k].grad, mom[k][0], mom[k][1], r + 1, 1e-4, 0.9, 0.999, 1e-8, 0.01)
        t4 = time.perf_counter()
        if r:   # (r == 0 is the warm-up)
            phases.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
        hip = vq.get_code(px.to(device)).cpu()
        mism += int((hip != tokens).sum())
        ntok += tokens.numel()
    med = [statistics.median(p[i] for p in phases) for i in range(4)]
    total = statistics.median(sum(p) for p in phases)
    # north_star "bit-exact VQ token indices", stated as measured on THE BENCHED BATCH (the 64 images rank 0 trains on): the oracle's
    # tokenizer on them (8 at a time, untimed) against the HIP tokenizer in the benched mode; every disagreeing token with the oracle's
    # own f32 top-2 margin and the gap between the two candidates, in f32 ulps of the distance (oracle/vq_parity.py; the full
    # float64 accounting of each disagreement is tests/test_gpu_models.py::test_vq_indices_over_bench_batch_vs_oracle)
    bench_vq = None
    try:
        from oracle import vq_parity as VP
        pxb, _ = synthetic_batch(bench_batch, torch.device("cpu"), seed=1000)         # rank 0's images of the timed run ...
        vsd = W.fill_state_dict(W.vqgan_shapes(W.VQGAN_F16), 1234, "vqgan")            # ... and its tokenizer weights (build_models)
        vq.load_state_dict(vsd)
        cb = vsd["quantize.embedding.weight"]
        io, do = [], []
        with torch.no_grad():
            for i in range(0, bench_batch, 8):
                z = O.vqgan_encoder(vsd, W.VQGAN_F16, pxb[i:i + 8])
                d = O.vq_distances(z.permute(0, 2, 3, 1).reshape(-1, cb.shape[1]).contiguous(), cb)
                do.append(d); io.append(torch.argmin(d, dim=1))
        io, do = torch.cat(io), torch.cat(do)
        ih = vq.get_code(pxb.to(device)).cpu().reshape(-1)
        mm = VP.oracle_margins(do, io, ih)
        bench_vq = {"tokens": int(io.numel()), "images": bench_batch, "disagreements": len(mm),
                    "each": [{k: (round(v, 2) if isinstance(v, float) else v) for k, v in m.items()} for m in mm[:16]],
                    "unit": "f32 ulps of the distance (an ulp at d ~ 30 is 1.9e-6)"}
    except Exception as e:   # noqa: BLE001
        bench_vq = {"error": f"{type(e).__name__}: {str(e)[:160]}"}
    return {"value": round(bs / total, 4), "unit": "images/s", "cores": cores, "kind": "port", "cpu_model": cpu_model,
            "sample": f"config {cfg_name} train step at bs={bs} (BASELINE.json configs[0]), f32, oracle/maskgit_oracle.py on {cores} of "
                      f"{os.cpu_count()} host threads (capped at 32 because torch's CPU ops stop scaling and oversubscribe well below the node's "
                      f"hardware thread count): 1 warm-up + median of {reps} steps ({total:.1f} s per step)",
            "phase_s": {"vq_encode+mask": round(med[0], 2), "forward": round(med[1], 2), "backward": round(med[2], 2), "adamw": round(med[3], 2)},
            "vq_index_mismatches": f"{mism} of {ntok} tokens (HIP bf16x3 tokenizer vs the f32 oracle, {reps + 1} x {bs} images)",
            "vq_index_mismatches_bench_batch": bench_vq}


def comm_block(info, world, dp_ms, plain_ms, grad_dtype, rccl_log):
    """What the data-parallel line did on the wire, so that a multi-GPU run explains itself (DESIGN.md section 6 predicts 7.6-7.8x at
    N = 8 when RCCL spreads the buckets over all seven xGMI links, ~6.4x on a single f32 ring): ranks RCCL saw, gradient bytes and
    buckets per step, the all-reduce alone (algorithm / bus bandwidth: busbw = algbw x 2 (N - 1) / N, the per-link figure to hold
    against ~77 GB/s per direction of one xGMI link), the communication the step could not hide (data-parallel step minus the same
    step without the reducer on the same GPUs), and RCCL's own description of the communicator (NCCL_DEBUG=INFO INIT / GRAPH lines:
    channels, rings / trees, transports)."""
    import glob
    import re
    out = {"backend": "nccl (RCCL)" if dist.get_backend() != "gloo" else "gloo (protocol test)", "ranks": dist.get_world_size() if dist.is_initialized() else world, "grad_allreduce_dtype": grad_dtype,
           "bytes_per_step": int(info.get("bytes_per_step", 0)), "buckets_per_step": info.get("buckets_per_step"),
           "dp_step_ms": round(dp_ms, 3), "same_gpus_step_without_reducer_ms": None if plain_ms is None else round(plain_ms, 3),
           "exposed_comm_ms": None if plain_ms is None else round(dp_ms - plain_ms, 3)}
    out["bucket_bytes"] = info.get("bucket_bytes")
    for k in ("allreduce_alone_ms", "allreduce_alone_algbw_GBps", "allreduce_alone_busbw_GBps", "allreduce_alone_error", "per_bucket",
              "per_bucket_error"):
        if k in info:
            out[k] = info[k]
    if world > 1 and "allreduce_alone_ms" in info and plain_ms is not None:
        out["hidden_fraction_of_allreduce"] = round(max(0.0, 1.0 - max(0.0, dp_ms - plain_ms) / max(info["allreduce_alone_ms"], 1e-9)), 3)
    lines = []
    if rccl_log:
        try:
            pat = re.compile(r"(nranks|nRanks|Channel|channel|Ring|Tree|ring|tree|Connected|Using network|NET/|P2P|xGMI|XGMI|algorithm|protocol|Pattern|comm 0x)")
            seen = set()
            files = sorted(glob.glob(rccl_log + "*"))
            out["rccl_log_files"] = len(files)
            for f in files:
                for l in open(f, errors="replace"):
                    l = l.strip()
                    key = re.sub(r"0x[0-9a-f]+|\d+", "#", l)          # one line per KIND of message (Tree 0 .. Tree 63 -> one)
                    if pat.search(l) and key not in seen and len(lines) < 24:
                        seen.add(key)
                        lines.append(l[-200:])
                break      # rank 0's file (sorted first) says what every rank built
        except Exception as e:   # noqa: BLE001
            lines.append(f"(RCCL log not read: {type(e).__name__})")
    out["rccl_info"] = lines
    return out


def traffic_of(kernel_family):
    """HBM bytes per launch of a kernel family from the committed PMC passes (scripts/gpu.sh final -> profiles/r06_traffic.json)"""
    if not os.path.exists(TRAFFIC_JSON) or kernel_family not in KERNEL_OF:
        return None, None
    t = json.load(open(TRAFFIC_JSON))
    for name, v in t.get("kernels", {}).items():
        if KERNEL_OF[kernel_family] in name:
            return v["hbm_bytes_per_launch"], t.get("note")
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="B", choices=["A", "B"])
    ap.add_argument("--vq-dtype", default="bf16x3", choices=["f32", "bf16x3", "bf16"])
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--grad-dtype", default="f32", choices=["f32", "bf16"], help="gradient all-reduce payload (N > 1)")
    ap.add_argument("--no-prefetch", action="store_true", help="encode each batch inline instead of one step ahead on a second stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary legs (config A / 4 / 5, tokenizer variants)")
    ap.add_argument("--uvit-leg", default=None, help="internal: run one config-4 leg 'batch,seq,steps' and print its JSON")
    ap.add_argument("--device", type=int, default=None, help="internal: GPU index of a secondary leg (default: LOCAL_RANK or 0)")
    ap.add_argument("--leg", default=None, help="internal: run one secondary leg ('run,<cfg>,<vq>,<mode>,<steps>,<batch>' | 'vqgan,<batch>' | "
                                                "'taming,<batch>') and print its JSON")
    args = ap.parse_args()
    if args.uvit_leg:
        parts = args.uvit_leg.split(",")
        b, sq, st = (int(x) for x in parts[:3])
        torch.cuda.set_device(0)
        print(json.dumps(uvit_leg(torch.device("cuda", 0), b, sq, st, f32=len(parts) > 3 and parts[3] == "f32", x3=len(parts) > 3 and parts[3] == "x3",
                                  f16=len(parts) > 3 and parts[3] == "f16")))
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0")) if args.device is None else args.device
    if os.environ.get("MUSE_BENCH_DEVICE") is not None:      # (protocol tests of the N > 1 path on a box with fewer GPUs than ranks)
        local = int(os.environ["MUSE_BENCH_DEVICE"])
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    distributed = "RANK" in os.environ and "WORLD_SIZE" in os.environ   # launched by torch.distributed.run (also with one rank)
    rccl_log = None
    if distributed:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL prints to the C-level stdout (its version banner whenever NCCL_DEBUG is set - the GPU boxes export NCCL_DEBUG=VERSION -, its
        # INFO lines unless NCCL_DEBUG_FILE is honoured).  So that no such line can ever land next to the ONE JSON line this script owes
        # its caller, file descriptor 1 is pointed at a log file for the rest of the process and Python's sys.stdout is re-opened on
        # the real stdout.  With more than one rank (or MUSE_BENCH_RCCL_DEBUG=1) RCCL is also asked for its INIT / GRAPH lines: its own
        # account of what it built (rings / trees, channels, transports) goes into the `comm` block.
        rccl_log = f"/tmp/muse_rccl_{os.getpid()}_r{rank}.log"
        if (world > 1 or os.environ.get("MUSE_BENCH_RCCL_DEBUG") == "1") and os.environ.get("MUSE_BENCH_RCCL_DEBUG") != "0" \
                and os.environ.get("NCCL_DEBUG", "VERSION").upper() in ("VERSION", "WARN"):
            os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,GRAPH,ENV")
        try:
            sys.stdout.flush()
            real = os.dup(1)
            sink = os.open(rccl_log, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
            os.dup2(sink, 1)
            os.close(sink)
            sys.stdout = os.fdopen(real, "w", buffering=1)
        except OSError:
            rccl_log = None
        if os.environ.get("MUSE_BENCH_BACKEND", "nccl") == "gloo":   # protocol test of the N > 1 path on a box with fewer GPUs than ranks
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)  # "nccl" is RCCL on ROCm

    import muse
    from muse import ops

    prof_bytes = {}
    comm_info, parity = {}, {}

    def run(cfg_name, vq_dtype, steps, warmup, profile=False, tokens_given=False, inline_tokenizer=False, use_reducer=True,
            transformer_dtype=torch.bfloat16):
        vq, model, opt, tcfg = build_models(cfg_name, vq_dtype, device, seed=1234, transformer_dtype=transformer_dtype)
        reducer = (muse.GradReducer(model, grad_dtype=torch.bfloat16 if args.grad_dtype == "bf16" else torch.float32)
                   if distributed and use_reducer else None)
        step = muse.TrainStep(vq, model, opt, reducer)
        px, cls = synthetic_batch(args.batch, device, seed=1000 + rank)  # different data per rank (weak scaling)
        toks = vq.get_code(px) if tokens_given else None

        prefetch = not args.no_prefetch and not tokens_given and not inline_tokenizer

        def one():
            # prefetch: the NEXT batch's tokenizer pass is enqueued on a second stream before this batch's transformer step
            # (TrainStep docstring); every timed step still runs one encode + one fwd/bwd + AdamW
            return step(None if tokens_given else px, cls, image_tokens=toks, next_pixel_values=px if prefetch else None)
        loss = None
        for _ in range(warmup):
            loss, _ = one()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        if reducer is not None:
            reducer.stats = {"buckets": 0, "bytes": 0}
        t0 = time.perf_counter()
        for _ in range(steps):
            loss, mask_prob = one()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        el = time.perf_counter() - t0
        if distributed:
            t = torch.tensor([el], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t)
            if reducer is not None:
                loss, _ = reducer.reduce_metrics(loss, mask_prob)     # the two logged scalars of the loop as one 2-float all-reduce
                comm_info.update(buckets_per_step=reducer.stats["buckets"] / steps, bytes_per_step=reducer.stats["bytes"] / steps,
                                 bucket_bytes=int(reducer.bucket_elems * (2 if args.grad_dtype == "bf16" else 4)),
                                 note="per_bucket comes from ONE extra traced step after the timed region (it also steps the optimizer)")
                try:    # one more step with per-bucket events: which share of each bucket's all-reduce backward hid
                    reducer.trace = []
                    one()
                    torch.cuda.synchronize()
                    comm_info["per_bucket"] = reducer.bucket_overlap()
                except Exception as e:   # noqa: BLE001
                    comm_info["per_bucket_error"] = f"{type(e).__name__}: {str(e)[:160]}"
                finally:
                    reducer.trace = None
                try:    # the gradient all-reduce ALONE (no compute beside it), in the reducer's own buckets: what the links deliver
                    g = model.flat_grads()
                    be = reducer.bucket_elems
                    cuts = list(range(g.numel(), 0, -be))
                    def allreduce_all():
                        for hi in cuts:
                            reducer._reduce(g[max(0, hi - be):hi], True)
                    allreduce_all()
                    torch.cuda.synchronize()
                    dist.barrier()
                    ta = time.perf_counter()
                    for _ in range(3):
                        allreduce_all()
                    torch.cuda.synchronize()
                    tb = torch.tensor([(time.perf_counter() - ta) / 3], device=device, dtype=torch.float64)
                    dist.all_reduce(tb, op=dist.ReduceOp.MAX)
                    nbytes = g.numel() * (2 if args.grad_dtype == "bf16" else 4)
                    comm_info.update(allreduce_alone_ms=round(float(tb) * 1e3, 3),
                                     allreduce_alone_algbw_GBps=round(nbytes / float(tb) / 1e9, 1),
                                     allreduce_alone_busbw_GBps=round(nbytes / float(tb) / 1e9 * 2 * (world - 1) / max(world, 1), 1))
                except Exception as e:   # noqa: BLE001   (diagnostics must never take the bench line down)
                    comm_info["allreduce_alone_error"] = f"{type(e).__name__}: {str(e)[:160]}"
        prof = tr_ms = None
        if profile:
            # per-kernel timing needs the kernels one at a time: no token prefetch, weight gradients on the main stream
            step._pf = None
            ws, model.wgrad_stream = model.wgrad_stream, False
            # (the instrumented step times the kernels the TIMED step ran: beside a train step the tokenizer's fused convolutions are the
            #  launch-per-tile ones - muse.TrainStep switches the persistent form off around its prefetch -, alone they are the persistent ones)
            with ops.conv_persistent(not prefetch):
                step(px, cls)              # one un-timed step in this serial configuration first: the tokenizer's buffers now come from the
                torch.cuda.synchronize()   # main stream's allocator pool (they lived on the prefetch stream), clocks and caches are warm
                ops.profile_start()
                step(px, cls)
                prof = ops.profile_stop(with_kind=True)
            model.wgrad_stream = ws
            prof_bytes.update(ops.PROF_BYTES)
            # the tokenizer once more with the GroupNorm applied by its own kernel (the two-kernel route the fused convolution replaced):
            # the bare convolution's rate next to the fused kernel's, from the same process
            if getattr(vq, "fuse_gn_apply", False):
                vq.fuse_gn_apply = False
                vq.get_code(px)
                torch.cuda.synchronize()
                ops.profile_start()
                vq.get_code(px)
                unf = ops.profile_stop(with_kind=True)
                vq.fuse_gn_apply = True
                cv = [(w, t) for n_, w, t, k in unf if n_ == "conv_bf16x3_dma"]
                gn = [t for n_, w, t, k in unf if n_ == "groupnorm_silu"]
                if cv:
                    prof_bytes["__unfused__"] = {"conv_tflops": sum(w for w, _ in cv) / (sum(t for _, t in cv) * 1e-3) / 1e12,
                                                 "conv_ms": sum(t for _, t in cv), "groupnorm_ms": sum(gn), "launches": len(cv)}
            # transformer forward + backward alone (tokens given, no optimizer): the north_star's "MaskGitTransformer step"
            ids, labels, _, _ = muse.prepare_inputs_and_labels(vq, None, cls, model.config.mask_token_id, image_tokens=vq.get_code(px))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 3
            for i in range(n + 1):
                if i == 1:
                    e0.record()
                model.zero_grad(set_to_none=True)
                _, l2 = model(input_ids=ids, labels=labels)
                l2.backward()
                if reducer is not None:
                    reducer.finish()
            e1.record()
            torch.cuda.synchronize()
            tr_ms = e0.elapsed_time(e1) / n
            # MEASURED parity of the benched mode on the benched batch (the line's `dtype` quotes these, not constants): the same
            # weights and inputs through the bf16 mode and through the exact-f32 parity mode (which meets north_star's 1e-3 against
            # the reference: tests/test_gpu_models.py::test_transformer_config_b_benched_batch_vs_reference_golden)
            try:
                with torch.no_grad():
                    lg_b, ls_b = model(input_ids=ids, labels=labels)
                    lg_b, ls_b = lg_b.float().clone(), float(ls_b)
                    model.set_compute_dtype(torch.float32)
                    lg_f, ls_f = model(input_ids=ids, labels=labels)
                    model.set_compute_dtype(torch.bfloat16)
                    parity.update(logits_max_abs_diff_over_max_logit=float((lg_b - lg_f).abs().max() / lg_f.abs().max()),
                                  loss_rel_diff=abs(ls_b - float(ls_f)) / abs(float(ls_f)), loss_bf16=ls_b, loss_f32=float(ls_f),
                                  rows=int(ids.numel()))
                    del lg_b, lg_f
            except Exception as e:   # noqa: BLE001
                parity["error"] = f"{type(e).__name__}: {str(e)[:160]}"
        lossv = float(loss)
        del step, vq, model, opt, reducer
        gc.collect()               # (tapes and parameter views of the finished leg: later legs must not pay for a growing Python heap)
        torch.cuda.empty_cache()
        return el, lossv, prof, tr_ms

    def run_leg(cfgn, vqd, mode, n, batch):
        e2, _, _, _ = run(cfgn, vqd, n, 2, tokens_given=(mode == "tokens"), inline_tokenizer=(mode == "inline"),
                          transformer_dtype=torch.float32 if mode == "transformer_f32" else ("f16" if mode == "transformer_f16" else torch.bfloat16))
        return {"images_per_s": round(batch * n / e2, 1)}

    if args.leg:
        f = args.leg.split(",")
        if f[0] == "run":
            args.batch = int(f[5])
            print(json.dumps(run_leg(f[1], f[2], f[3], int(f[4]), args.batch)))
        elif f[0] == "vqgan":
            print(json.dumps(vqgan_roundtrip(device, int(f[1]))))
        elif f[0] == "taming":
            print(json.dumps(taming_leg(device, int(f[1]))))
        elif f[0] == "latency":
            print(json.dumps(latency_leg(device)))
        return

    plain_ms = None
    if distributed and os.environ.get("MUSE_BENCH_COMM_PLAIN", "1") != "0":
        # the same step WITHOUT the reducer on the same GPUs first (every rank its own replica, max over ranks): the data-parallel
        # step minus this is the communication the step could not hide.  In a process of its own per rank: a second leg inside this
        # process measures 10-15 % slower than a fresh one (same effect as leg_isolated's note), and the timed run must be the fresh one.
        try:
            import subprocess
            n_plain = max(4, args.steps // 2)
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "NCCL_DEBUG",
                                                                    "NCCL_DEBUG_FILE", "NCCL_DEBUG_SUBSYS", "TORCHELASTIC_RUN_ID")}
            cmd = [sys.executable, os.path.abspath(__file__), "--leg", f"run,{args.config},{args.vq_dtype},plain,{n_plain},{args.batch}",
                   "--device", str(local)] + (["--no-prefetch"] if args.no_prefetch else [])
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
            ips = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])["images_per_s"]
            t = torch.tensor([args.batch / ips * 1e3], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            plain_ms = float(t)
        except Exception as e:   # noqa: BLE001   (diagnostics must never take the bench line down)
            print(f"bench: the no-reducer comparison leg did not run ({type(e).__name__}: {e})", file=sys.stderr)
            try:
                t = torch.tensor([0.0], device=device, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)      # keep the ranks' collectives paired
            except Exception:   # noqa: BLE001
                pass
    el, lossv, prof, tr_ms = run(args.config, args.vq_dtype, args.steps, args.warmup, profile=True)
    ms = el / args.steps * 1e3
    value = args.batch * world * args.steps / el

    # per-kernel roofline from live HIP-event timings of one instrumented step
    agg = {}
    for name, work, t, kind in prof:
        a = agg.setdefault((name, kind), [0.0, 0.0, 0])
        a[0] += work; a[1] += t; a[2] += 1
    mfma = {k[0]: v for k, v in agg.items() if k[1] == "flop"}
    hbm = {k[0]: v for k, v in agg.items() if k[1] == "byte"}
    kinds = {k: {"launches": v[2], "ms_total": round(v[1], 3), "avg_us": round(v[1] / v[2] * 1e3, 1),
                 "tflops": round(v[0] / (v[1] * 1e-3) / 1e12, 1)} for k, v in sorted(mfma.items(), key=lambda kv: -kv[1][1])}
    hbm_kinds = {k: {"launches": v[2], "ms_total": round(v[1], 3), "GBps": round(v[0] / (v[1] * 1e-3) / 1e9, 1),
                     "frac_of_8TBps": round(v[0] / (v[1] * 1e-3) / 1e9 / HBM_PEAK, 3)} for k, v in sorted(hbm.items(), key=lambda kv: -kv[1][1])}
    dname, (dfl, dms, dn) = max(mfma.items(), key=lambda kv: kv[1][1])
    x3 = dname.startswith("conv_bf16x3")
    peak = round(PEAK["bf16"] / 3.0, 1) if x3 else PEAK["bf16" if "bf16" in dname else "f32"]
    ach = dfl / (dms * 1e-3) / 1e12
    traffic, tnote = traffic_of(dname)
    roofline = {"bound": "mfma", "kernel": dname, "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s",
                "frac": round(ach / peak, 4), "traffic": traffic, "launches_per_step": dn, "avg_launch_us": round(dms / dn * 1e3, 1),
                "peak_basis": ("issued bf16 MFMA / 3: 2500 dense bf16 TFLOP/s divided by the three bf16 MFMAs the bf16x3 algorithm issues "
                               "per f32 product; `achieved` counts ALGORITHMIC flops (2*B*H*W*Cout*9*Cin), executed_mfma_tflops the issued "
                               "ones; against the plain 2500 peak frac would be frac/3") if x3 else "dense MFMA peak of the operand dtype",
                "executed_mfma_tflops": round(ach * (3 if x3 else 1), 1),
                "two_kernel_route": ({"conv_achieved": round(prof_bytes["__unfused__"]["conv_tflops"], 1),
                                      "conv_frac": round(prof_bytes["__unfused__"]["conv_tflops"] / peak, 4),
                                      "conv_ms_per_step": round(prof_bytes["__unfused__"]["conv_ms"], 3),
                                      "groupnorm_apply_ms_per_step": round(prof_bytes["__unfused__"]["groupnorm_ms"], 3),
                                      "what": "the same tokenizer pass with MUSE_GN_FUSE=0: bare patch-slab convolution + separate GroupNorm apply"}
                                     if (dname == "conv_bf16x3_dma" and "__unfused__" in prof_bytes) else None),
                "note": ("since round 3 this kernel also normalises its input (GroupNorm + SiLU + bf16x3 split inside the convolution, "
                         "muse_conv2d_nhwc_gn_split2): `achieved` still counts the convolution's flops only, over a launch that now contains "
                         "the former 5.3 ms / step GroupNorm apply pass - the fraction fell 0.56 -> ~0.50 while the step got 2.6 ms shorter "
                         "(DESIGN.md section 5)") if dname == "conv_bf16x3_dma" else None,
                "algorithmic_bytes_per_launch": round(prof_bytes[dname] / dn) if dname in prof_bytes else None,
                "traffic_source": tnote,
                "per_kernel": kinds, "hbm_bound_kernels": hbm_kinds}
    gf_img = GF_ENCODE + 3 * GF_FWD[args.config]
    tr_tf = 3 * GF_FWD[args.config] * args.batch / tr_ms   # GFLOP / ms = TFLOP/s
    vq_hbm = [v for k, v in hbm.items() if k in ("groupnorm_silu", "avgpool2x2")]
    vq_bytes, vq_ms = sum(v[0] for v in vq_hbm), sum(v[1] for v in vq_hbm)
    extra = {"loss": round(lossv, 4), "algorithmic_gflop_per_image": round(gf_img, 2),
             "step_tflops_per_gpu": round(gf_img * args.batch / ms, 1),
             "transformer_fwd_bwd_ms": round(tr_ms, 2), "transformer_tflops": round(tr_tf, 1),
             "transformer_mfma_frac": round(tr_tf / PEAK["bf16"], 4),
             "vqgan_hbm_GBps": round(vq_bytes / (vq_ms * 1e-3) / 1e9, 1) if vq_ms else None,
             "vqgan_hbm_frac": round(vq_bytes / (vq_ms * 1e-3) / 1e9 / HBM_PEAK, 4) if vq_ms else None,
             "vqgan_hbm_note": "GroupNorm+SiLU and 2x2 pooling kernels of the encoder: algorithmic bytes (each operand once) / their time / 8 TB/s",
             "mfma_ms_in_instrumented_step": round(sum(v[1] for v in mfma.values()), 2),
             "hbm_kernel_ms_in_instrumented_step": round(sum(v[1] for v in hbm.values()), 2),
             "measured_parity_bf16_vs_f32_mode": {k: (round(v, 8) if isinstance(v, float) else v) for k, v in parity.items()}}
    if args.config == "B" and args.batch == 64:
        # per family the time its BINDING resource implies (derivation and sources: profiles/r06_ceiling.md) next to this
        # run's serial time; a ratio above 1.15 has its measured decomposition there
        def _ms(*names):
            return round(sum(kinds[n]["ms_total"] for n in names if n in kinds) + sum(hbm_kinds[n]["ms_total"] for n in names if n in hbm_kinds), 3)
        ceil_rows = {
            "conv_bf16x3_dma": ("MFMA issue at the clock the kernel holds (1.72 GHz, power-limited): 3 x 8.155 TFLOP / (2.5 PF x 1.72 / 2.4)", 13.7,
                                _ms("conv_bf16x3_dma")),
            "gemm_fwd_dx": ("K loop bound by the operand stream from L2 into LDS (2500 cycles per K-tile against 2048 of MFMA: 26-28 B/clk/CU delivered, 32 needed) + "
                            "whole 256^2 tiles on 256 CUs + 82 us tile change per layer "
                            "(46 us register epilogue + 36 us stores, ablation)", 13.1, _ms("gemm_bf16_NN", "gemm_bf16_NT")),
            "gemm_dw": ("the same K loop at K = 16448, five slices alone on the chip", 5.9, _ms("gemm_bf16_TT")),
            "attention": ("HBM floor of q, k, v, o (+ dO, dq, dk, dv) at 6.3 TB/s", 1.15, _ms("attn_fwd_bf16", "attn_bwd_bf16")),
            "row_kernels": ("HBM streaming at the 6.0 TB/s a three-operand stream reaches on this chip", 8.9,
                            round(sum(v["ms_total"] for v in hbm_kinds.values()), 3)),
        }
        extra["ceiling"] = {"doc": "profiles/r06_ceiling.md",
                            "families": {k: {"binding": b, "implied_ms": i, "measured_ms": m, "ratio": round(m / i, 2) if m else None}
                                         for k, (b, i, m) in ceil_rows.items()}}
    if world == 1 and not args.no_extra:
        n2 = max(3, args.steps // 2)

        def leg(cfgn, vqd, mode):
            return leg_isolated(f"run,{cfgn},{vqd},{mode},{n2},{args.batch}", lambda: run_leg(cfgn, vqd, mode, n2, args.batch))["images_per_s"]
        extra["images_per_s_tokens_given"] = leg(args.config, args.vq_dtype, "tokens")   # pre-encoded tokens (scripts/pre_encode.py regime)
        if not args.no_prefetch:   # the same step with each batch encoded inline, in the reference loop's order (no second stream)
            extra["images_per_s_inline_tokenizer"] = leg(args.config, args.vq_dtype, "inline")
        other = "A" if args.config == "B" else "B"
        for cfgn, vqd in [(args.config, d) for d in ("f32", "bf16") if d != args.vq_dtype] + [(other, args.vq_dtype)]:
            extra[f"images_per_s_config{cfgn}_vq{vqd}"] = leg(cfgn, vqd, "plain")
        # the same step with the transformer in its exact-f32 mode: the mode that meets north_star's literal "logits within 1e-3 rel" against
        # the reference (2e-6); the headline's bf16 mode is the yaml's own mixed_precision regime, its measured gap is printed in `dtype`
        extra[f"images_per_s_config{args.config}_transformer_f32"] = leg(args.config, args.vq_dtype, "transformer_f32")
        extra.update(leg_isolated(f"vqgan,{args.batch}", lambda: vqgan_roundtrip(device, args.batch)))
        extra.update(leg_isolated(f"taming,{args.batch}", lambda: taming_leg(device, args.batch)))
        # config 4 at batch sizes that use the 288 GB (cc12m_uvit_clip.yaml trains 64 per GPU x 2 accumulation steps): the fixed
        # per-step cost (AdamW over 729 M parameters, ~500 small launches) is amortised over more tokens
        extra["config4_uvit_seq256"] = uvit_leg_isolated(device, 128, 256, 3)
        extra["config4_uvit_seq1024"] = uvit_leg_isolated(device, 48, 1024, 2)
        # ... and at the YAML's own precision class (cc12m_uvit_clip.yaml:102-103 mixed_precision "no" + TF32): exact f32 here
        extra["config4_uvit_seq256_f32"] = uvit_leg_isolated(device, 32, 256, 2, f32=True)
        # ... and the same precision class on the bf16 matrix pipes: f32 tensors, three bf16 MFMA products per GEMM
        extra["config4_uvit_seq256_bf16x3"] = uvit_leg_isolated(device, 64, 256, 2, x3=True)             # the yaml's 64 per GPU
        extra["config4_uvit_seq256_bf16x3_b128"] = uvit_leg_isolated(device, 128, 256, 2, x3=True)       # ... and a batch that uses the HBM
        # ... and BASELINE.json's sequence length in that precision class (round 6): every weight GEMM as bf16x3 products; the attention core
        # of a 1024-token sequence runs attention3.hip's streaming kernels (online-softmax forward, dQ and dK / dV passes over 256-row blocks;
        # one-tile kernels against the 77 text states): 52.3 images/s with the materialised exact-f32 core round 5 had, 81.5 with block pairs +
        # merge (profiles/r06_c4_seq1024_x3*.txt), 92 streaming
        extra["config4_uvit_seq1024_bf16x3"] = uvit_leg_isolated(device, 32, 1024, 2, x3=True)
        # ... and that precision class at its natural cost on this chip (round 6): IEEE half has TF32's 10-bit mantissa and its MFMA runs at the
        # bf16 rate - every weight GEMM as ONE half product (muse_gemm dtype MUSE_F16), gradient operands through a power-of-two scale
        # (the "f16" compute mode; equal to an emulated TF32 product to 1e-7: profiles/r06_f16_mode_parity.txt)
        extra["config4_uvit_seq256_f16"] = uvit_leg_isolated(device, 64, 256, 2, f16=True)
        extra["config4_uvit_seq256_f16_b128"] = uvit_leg_isolated(device, 128, 256, 2, f16=True)
        extra["config4_uvit_seq1024_f16"] = uvit_leg_isolated(device, 32, 1024, 2, f16=True)
        # the reference's PUBLISHED metric (its only published numbers): text-to-image pipeline latency, 12 steps, 256 x 256
        extra["inference_latency"] = leg_isolated("latency", lambda: latency_leg(device))

    out = {
        "metric": "images/sec/node (MaskGit train step, 256^2, bs=64/GPU)", "value": round(value, 2), "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": f"bf16 (transformer GEMM/attention operands; f32 accumulate, residual, norms, loss) + {args.vq_dtype} tokenizer"
                 + (" (f32 activations, f32-class products as 3 bf16 MFMAs)" if args.vq_dtype == "bf16x3" else "")
                 + ((f"; MEASURED in this run on the benched batch ({parity['rows']} rows), this mode vs the exact-f32 parity mode of the same "
                     f"weights: loss {parity['loss_rel_diff']:.1e} rel, logits {parity['logits_max_abs_diff_over_max_logit']:.1e} of max|logit| "
                     "(the reference's own autocast-bf16 gap at this geometry is 1.2e-2, tests/golden/transformer_b_full_bf16.npz; the f32 "
                     "mode meets north_star's 1e-3 against the reference: logits 2e-6)") if "loss_rel_diff" in parity else
                    "; parity of this mode was not measured in this run (" + parity.get("error", "profile leg skipped") + ")"),
        "data": "synthetic",
        "config": {"workload": f"MaskGit train step: MaskGitVQGAN f16-256 encode ({args.vq_dtype}) + cosine mask + "
                               f"MaskGitTransformer config {args.config} "
                               f"({'configs/imagenet.yaml: hidden 768, 24 layers, 16 heads, vocab 2048' if args.config == 'B' else 'README: hidden 512, 8 layers, 8 heads, vocab 2025'}"
                               f", seq 257) fwd+bwd (bf16 MFMA, f32 accum/residual) + AdamW",
                   "global_batch": args.batch * world, "per_gpu_batch": args.batch, "resolution": 256, "seq_len": 257,
                   "parallelism": f"dp{world}", "vqgan_dtype": args.vq_dtype, "random_init": True,
                   "grad_allreduce_dtype": args.grad_dtype if world > 1 else None,
                   "tokenizer_prefetch": not args.no_prefetch, "wgrad_stream": os.environ.get("MUSE_WGRAD_STREAM", "1") != "0"},
        "roofline": roofline,
        "extra": extra,
    }
    if distributed:
        out["comm"] = comm_block(comm_info, world, ms, plain_ms, args.grad_dtype, rccl_log)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.config, device, bench_batch=args.batch)
    if rank == 0:
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
