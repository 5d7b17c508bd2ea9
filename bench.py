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
to f32 near-ties: tests/test_gpu_models.py::test_vqgan_f16_256_vs_oracle).  At N=1 the line also carries, in `extra`, the
same step with the exact-f32 MFMA tokenizer (`--vq-dtype f32`) and the pure-bf16 one; `--config A` = README-tiny model.
"""
import argparse
import json
import os
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
# algorithmic work per image, GFLOP
GF_ENCODE = 128.63
GF_FWD = {"A": 19.00, "B": 122.40}


def build_models(cfg_name, vq_dtype, device, seed):
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
    model.to(device).train().set_compute_dtype(torch.bfloat16)
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


def cpu_baseline(cfg_name, bs=16):
    """the CPU oracle (port of the reference path) on this node's host cores, one full train step at bs=16 (~10-20 s)"""
    import weights as W
    from oracle import maskgit_oracle as O
    cores = min(os.cpu_count(), 32)  # torch CPU ops stop scaling (and oversubscribe) far below the 256 hw threads of the node
    torch.set_num_threads(cores)
    tcfg = dict(W.TRANSFORMER_A if cfg_name == "A" else W.TRANSFORMER_B)
    vsd = W.fill_state_dict(W.vqgan_shapes(W.VQGAN_F16), 1, "vqgan")
    tsd = W.fill_state_dict(W.transformer_shapes(tcfg), 2, "transformer")
    px, cls = W.images(bs, 256, 3), torch.from_numpy(np.random.default_rng(4).integers(0, 1000, size=bs))
    t, nz = W.uniforms((bs,), 5), W.uniforms((bs, 256), 6)
    t0 = time.time()
    out = O.train_step(vsd, W.VQGAN_F16, tsd, tcfg, px, cls, t, nz)
    k = "mlm_layer.to_logits.weight"
    O.adamw_step(tsd[k], out["grads"][k], torch.zeros_like(tsd[k]), torch.zeros_like(tsd[k]), 1, 1e-4, 0.9, 0.999, 1e-8, 0.01)
    dt = time.time() - t0
    return {"value": round(bs / dt, 4), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"1 train step, config {cfg_name}, bs={bs}, f32, oracle/maskgit_oracle.py on {cores} of {os.cpu_count()} host threads ({dt:.1f} s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="B", choices=["A", "B"])
    ap.add_argument("--vq-dtype", default="bf16x3", choices=["f32", "bf16x3", "bf16"])
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--extra", action="store_true", help="also time config A and the bf16 tokenizer (N=1 only)")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary bf16-tokenizer timing")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=device)  # "nccl" is RCCL on ROCm

    import muse
    from muse import ops

    def run(cfg_name, vq_dtype, steps, warmup, profile):
        vq, model, opt, tcfg = build_models(cfg_name, vq_dtype, device, seed=1234)
        reducer = muse.GradReducer(model) if world > 1 else None
        step = muse.TrainStep(vq, model, opt, reducer)
        px, cls = synthetic_batch(args.batch, device, seed=1000 + rank)  # different data per rank (weak scaling)
        loss = None
        for _ in range(warmup):
            loss, _ = step(px, cls)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss, _ = step(px, cls)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t)
        prof = None
        if profile:
            ops.profile_start()
            step(px, cls)
            prof = ops.profile_stop()
        lossv = float(loss)
        del step, vq, model, opt, reducer
        torch.cuda.empty_cache()
        return el, lossv, prof

    el, lossv, prof = run(args.config, args.vq_dtype, args.steps, args.warmup, profile=True)
    ms = el / args.steps * 1e3
    value = args.batch * world * args.steps / el

    # roofline of the dominant MFMA kernel, from live HIP-event timings of one instrumented step
    agg = {}
    for name, fl, t in prof:
        a = agg.setdefault(name, [0.0, 0.0, 0])
        a[0] += fl; a[1] += t; a[2] += 1
    dom = max(agg.items(), key=lambda kv: kv[1][1])
    kinds = {k: {"launches": v[2], "ms_total": round(v[1], 3), "avg_us": round(v[1] / v[2] * 1e3, 1),
                 "tflops": round(v[0] / (v[1] * 1e-3) / 1e12, 1)} for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}
    dname, (dfl, dms, dn) = dom
    peak = PEAK["bf16" if "bf16" in dname else "f32"]
    x3 = dname.startswith("conv_bf16x3")
    if x3:
        peak = round(PEAK["bf16"] / 3.0, 1)  # three bf16 MFMAs per algorithmic f32 product
    ach = dfl / (dms * 1e-3) / 1e12
    roofline = {"bound": "mfma", "kernel": dname, "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s",
                "frac": round(ach / peak, 4), "traffic": None, "launches_per_step": dn, "avg_launch_us": round(dms / dn * 1e3, 1),
                "peak_note": ("2500 dense bf16 MFMA TFLOP/s / 3: the bf16x3 algorithm issues three bf16 MFMAs per f32 product; "
                              "achieved counts algorithmic flops (x3 = executed MFMA rate)") if x3 else "dense MFMA peak of the dtype",
                "executed_mfma_tflops": round(ach * (3 if x3 else 1), 1),
                "per_kernel": kinds}
    gf_img = GF_ENCODE + 3 * GF_FWD[args.config]
    extra = {"loss": round(lossv, 4), "algorithmic_gflop_per_image": gf_img,
             "step_tflops_per_gpu": round(gf_img * args.batch / ms, 1),
             "mfma_ms_in_instrumented_step": round(sum(v[1] for v in agg.values()), 2)}
    if world == 1 and not args.no_extra:
        other = "A" if args.config == "B" else "B"
        variants = [(args.config, d) for d in ("f32", "bf16x3", "bf16") if d != args.vq_dtype] + [(other, args.vq_dtype)]
        if args.extra:
            variants += [(other, d) for d in ("f32", "bf16x3", "bf16") if d != args.vq_dtype]
        for cfgn, vqd in variants:
            if (cfgn, vqd) == (args.config, args.vq_dtype):
                continue
            e2, _, _ = run(cfgn, vqd, max(3, args.steps // 2), 2, profile=False)
            extra[f"images_per_s_config{cfgn}_vq{vqd}"] = round(args.batch * max(3, args.steps // 2) / e2, 1)

        extra.update(vqgan_roundtrip(device, args.batch))

    out = {
        "metric": "images/sec/node (MaskGit train step, 256^2, bs=64/GPU)", "value": round(value, 2), "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"MaskGit train step: MaskGitVQGAN f16-256 encode ({args.vq_dtype}) + cosine mask + "
                               f"MaskGitTransformer config {args.config} "
                               f"({'configs/imagenet.yaml: hidden 768, 24 layers, 16 heads, vocab 2048' if args.config == 'B' else 'README: hidden 512, 8 layers, 8 heads, vocab 2025'}"
                               f", seq 257) fwd+bwd (bf16 MFMA, f32 accum/residual) + AdamW",
                   "global_batch": args.batch * world, "per_gpu_batch": args.batch, "resolution": 256, "seq_len": 257,
                   "parallelism": f"dp{world}", "vqgan_dtype": args.vq_dtype, "random_init": True},
        "roofline": roofline,
        "extra": extra,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.config)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
