"""Round 6: the "f16" compute mode of the tape engines (one IEEE-half MFMA product per weight GEMM = TF32's operand precision) on the GPU.
  A  the product itself against an EMULATED TF32 product (operands rounded to 10 mantissa bits, float64 accumulation) and against float64,
     next to the bf16x3 and bf16 products - every operand layout, the gradient-scale path, half subnormals through the MFMA
  B  config 4 at full size against the real reference's golden outputs (tests/golden/uvit_full.npz) in the f16 mode, with the mode's
     clamp / flush counters
  C  the config-4 legs of bench.py in f16 / bf16x3 / bf16 (fresh processes)
python scripts/exp/f16_mode.py [A] [B] [C]
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "open-muse_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, ROOT)
import numpy as np
import torch

DEV = torch.device("cuda", 0)


def tf32(x):
    """round to nearest even at 10 mantissa bits (the TF32 operand format), f32 in / f32 out"""
    i = x.contiguous().view(torch.int32)
    r = (i + 0xFFF + ((i >> 13) & 1)) & ~0x1FFF
    return r.view(torch.float32)


def part_a():
    from muse import ops
    torch.manual_seed(0)
    print("== A: one product, [2048 x 1536] x [3072 x 1536]^T and its k-major forms; errors relative to sum |a||b| (max over the output)")
    M, N, K = 2048, 3072, 1536
    a = torch.randn(M, K, device=DEV)
    b = torch.randn(N, K, device=DEV) * 0.05
    mag = (a.abs().double() @ b.abs().double().t())
    exact = a.double() @ b.double().t()
    emu = tf32(a).double() @ tf32(b).double().t()

    def run(mode, la, lb, scale=None):
        A_ = a if la == 0 else a.t().contiguous()
        B_ = b if lb == 0 else b.t().contiguous()
        c = torch.empty(M, N, device=DEV)
        kw = dict(la=la, lb=lb, lda=A_.stride(0), ldb=B_.stride(0), ldc=N)
        if mode == "f16":
            im = ops.F16Images()
            if scale is not None:
                im.backward = True
                im.set_grad_scale(scale)
            with ops.f32_gemms_as_f16(True, im):
                ops.gemm(A_, B_, c, M, N, K, **kw)
            st = im.stats()
        elif mode == "x3":
            with ops.f32_gemms_as_bf16x3(True, None):
                ops.gemm(A_, B_, c, M, N, K, **kw)
            st = None
        elif mode == "bf16":
            ops.gemm(A_.bfloat16(), B_.bfloat16(), c, M, N, K, **kw)
            st = None
        else:
            ops.gemm(A_, B_, c, M, N, K, **kw)
            st = None
        return c, st

    for la, lb in ((0, 0), (0, 1), (1, 1), (1, 0)):
        row = []
        for mode in ("f16", "x3", "bf16", "f32"):
            c, st = run(mode, la, lb)
            e_exact = float(((c.double() - exact).abs() / mag).max())
            e_emu = float(((c.double() - emu).abs() / mag).max())
            row.append(f"{mode}: vs float64 {e_exact:.2e}" + (f", vs emulated TF32 {e_emu:.2e}, overflowed / flushed {st}" if mode == "f16" else ""))
        e_tf = float(((emu - exact).abs() / mag).max())
        print(f"  la{la} lb{lb}  " + " | ".join(row) + f" | emulated TF32 vs float64 {e_tf:.2e}")
    # gradient-scale path: A is a "gradient" of magnitude 1e-7 (far below half's normal range)
    a_small = a * 1e-7
    exact_s = a_small.double() @ b.double().t()
    mag_s = mag * 1e-7
    for scale in (None, 2.0 ** 24):
        a_keep = a
        a = a_small
        c, st = run("f16", 0, 0, scale)
        a = a_keep
        print(f"  A x 1e-7, gradient scale {scale}: vs float64 {float(((c.double() - exact_s).abs() / mag_s).max()):.2e}, overflowed / flushed {st}")
    # half subnormals through the MFMA: operands exactly representable as half subnormals
    sub = (torch.randint(-512, 512, (M, K), device=DEV).float() * 2.0 ** -24)
    bb = torch.randint(-8, 8, (N, K), device=DEV).float()
    a_keep, b_keep = a, b
    a, b = sub, bb
    c, st = run("f16", 0, 0)
    a, b = a_keep, b_keep
    ref = sub.double() @ bb.double().t()
    print(f"  half-subnormal operands (exact in half): max |c - exact| / max |exact| = {float((c.double() - ref).abs().max() / ref.abs().max()):.2e}"
          f" (0 = the MFMA keeps subnormal inputs), flushed {st}")


def part_b():
    import weights as W
    import muse
    from muse import modeling_transformer_v2 as Mv
    gd = os.path.join(ROOT, "tests", "golden")
    g = np.load(os.path.join(gd, "uvit_full.npz"))
    init = Mv.MaskGiTUViT_v2._init_weights
    Mv.MaskGiTUViT_v2._init_weights = lambda self: None
    try:
        model = muse.MaskGiTUViT(**W.UVIT_CC12M)
    finally:
        Mv.MaskGiTUViT_v2._init_weights = init
    model.load_state_dict(W.fill_by_shapes({k: tuple(v.shape) for k, v in model.state_dict().items()}, int(g["seed"])), strict=True)
    model.to(DEV).train()
    ids, enc, cond, micro, labels = (t.to(DEV) for t in W.uvit_inputs(int(g["batch"]), int(g["seq"]), int(g["text_len"]), int(g["seed"]) + 1))
    keys = W.UVIT_FULL_GRAD_KEYS
    print("== B: config 4 (728.7 M parameters) against the real reference's f32 outputs")
    for cd in ("f16", "bf16x3", torch.bfloat16, torch.float32):
        model.set_compute_dtype(cd)
        model.zero_grad(set_to_none=True)
        logits, loss = model(ids, enc, cond, micro, labels=labels)
        loss.backward()
        el = float(np.abs(W.subsample(logits.detach().float(), 16384).cpu().numpy() - g["logits"]).max()) / float(g["logits_absmax"])
        lrel = abs(float(loss) - float(g["loss"])) / float(g["loss"])
        params = dict(model.named_parameters())
        errs = {k: float(np.abs(W.subsample(params[k].grad.detach().float()).cpu().numpy() - g["grad." + k]).max()) / float(g["absmax." + k]) for k in keys}
        nerrs = {k: abs(float(params[k].grad.double().norm()) - float(g["norm." + k])) / float(g["norm." + k]) for k in keys}
        print(f"  {cd}: logits {el:.2e}  loss {lrel:.1e}  worst grad {max(errs.values()):.1e}  worst grad norm {max(nerrs.values()):.1e}"
              + (f"  overflowed / flushed {model.f16_stats()}  grad scale {model.f16_grad_scale_for(model.__dict__['_loss_rows'])}" if cd == "f16" else ""))


def part_c():
    print("== C: config-4 legs (bench.py --uvit-leg, fresh processes)")
    for spec in ("64,256,2,f16", "64,256,2,x3", "128,256,2,f16", "128,256,3", "32,1024,2,f16", "32,1024,2,x3"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--uvit-leg", spec], capture_output=True, text=True, timeout=900)
        try:
            d = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
            d.pop("dtype", None)
            print(" ", spec, d)
        except Exception:
            print(" ", spec, "FAILED", r.stdout[-500:], r.stderr[-1500:])


if __name__ == "__main__":
    parts = [p for p in sys.argv[1:] if p in "ABC"] or ["A", "B", "C"]
    for p in parts:
        {"A": part_a, "B": part_b, "C": part_c}[p]()
        sys.stdout.flush()
