"""Next-round experiment (written after the round-2 GPU budget was spent; NOT yet run on hardware).

Why: profiles/r02_ab_dp1_hwqueues{8,4}_bench.json and r02_ab_dp1_updstream_* show that the train step's time depends by 13 % on which
HIP streams end up sharing a hardware queue (GPU_MAX_HW_QUEUES=8: 69.7 ms, default 4: 61.5 ms).  The step's streams:
    caller's (default)  ->  high-priority step stream (TrainStep._hp_stream)
    tokenizer prefetch  (TrainStep._pf_stream, normal priority)
    weight gradients    (MaskGitTransformer._side_stream, normal priority)
This script times the default step (bench.py's timed loop, config B, bs 64, prefetch on) under explicit placements:
    base            what the package creates by itself
    order:<perm>    the three streams created up front in the given order (p = prefetch, s = side / dW, h = high-priority step), so that
                    the runtime's round-robin stream -> queue assignment changes
    prio:<p>,<s>    priorities of the prefetch and dW streams (0 normal, -1 high)
    mask:<n>:<who>  `who` (p or s) on a stream restricted to n of the CUs (hipExtStreamCreateWithCUMask through
                    libstream_placement.so); layout `x`: the first n/32 XCD-interleaved words, layout `c`: contiguous low bits
Run each GPU_MAX_HW_QUEUES value in its own process (the runtime reads it once):
    for q in 2 4 6 8; do GPU_MAX_HW_QUEUES=$q python scripts/exp/stream_placement.py base order:psh order:hps prio:-1,0 ...; done
Every variant must give the same losses (scheduling only); the script asserts it against `base`."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402  (build_models / synthetic_batch; puts the package and tests/golden on sys.path)

DEV = torch.device("cuda", 0)
N_CU = 256


def masked_stream(n_cu, layout):
    lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libstream_placement.so"))
    words = (ctypes.c_uint32 * (N_CU // 32))()
    if layout == "c":                       # bits 0 .. n-1
        for b in range(n_cu):
            words[b // 32] |= 1 << (b % 32)
    else:                                   # every (N_CU / n)-th bit: the same share of every XCD if the bits interleave the XCDs
        stride = N_CU // n_cu
        for b in range(0, N_CU, stride):
            words[b // 32] |= 1 << (b % 32)
    out = ctypes.c_void_p()
    rc = lib.sp_stream_create(words, N_CU // 32, ctypes.byref(out))
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed: {rc}")
    got = (ctypes.c_uint32 * (N_CU // 32))()
    lib.sp_stream_get_mask(out, got, N_CU // 32)
    print("   CU mask granted:", " ".join(f"{w:08x}" for w in got), flush=True)
    return torch.cuda.ExternalStream(out.value, device=DEV)


def run(variant, steps=10, warmup=3):
    import muse
    vq, model, opt, _ = bench.build_models("B", "bf16x3", DEV, seed=1234)
    step = muse.TrainStep(vq, model, opt, None)
    kind, _, arg = variant.partition(":")
    if kind == "order":
        made = {}
        for ch in arg:
            made[ch] = torch.cuda.Stream(device=DEV, priority=-1 if ch == "h" else 0)
        step._pf_stream, model._side_stream, step._hp_stream = made["p"], made["s"], made["h"]
    elif kind == "prio":
        pp, ps = (int(x) for x in arg.split(","))
        step._pf_stream = torch.cuda.Stream(device=DEV, priority=pp)
        model._side_stream = torch.cuda.Stream(device=DEV, priority=ps)
    elif kind == "mask":
        n, who = arg.split(":")
        layout = "x"
        if n[-1] in "xc":
            n, layout = n[:-1], n[-1]
        s = masked_stream(int(n), layout)
        if who == "p":
            step._pf_stream = s
        else:
            model._side_stream = s
    elif kind != "base":
        raise SystemExit(f"unknown variant {variant}")
    px, cls = bench.synthetic_batch(64, DEV, seed=1000)
    losses = []
    for _ in range(warmup):
        step(px, cls, next_pixel_values=px)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        losses.append(step(px, cls, next_pixel_values=px)[0])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    out = torch.stack(losses).cpu()
    del step, vq, model, opt
    torch.cuda.empty_cache()
    return ms, out


if __name__ == "__main__":
    variants = sys.argv[1:] or ["base"]
    if "base" not in variants:
        variants = ["base"] + variants
    torch.cuda.set_device(0)
    ref = None
    print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES", "(default 4)"), flush=True)
    for v in variants:
        ms, losses = run(v)
        if ref is None:
            ref = losses
        same = bool(torch.equal(losses, ref))
        print(f"{v:16s} {ms:7.2f} ms/step  {64 / ms * 1e3:7.1f} img/s   losses identical to base: {same}", flush=True)
        assert same or v.startswith("mask"), v      # (a masked stream must not change results either; reported, not fatal, for triage)
