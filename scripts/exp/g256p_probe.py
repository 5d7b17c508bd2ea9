"""Bring-up / A-B probe of the persistent 256^2 GEMM (csrc/gemm256p.h) against the launch-per-tile kernel (gemm256.h).
One process = one MUSE_G256P_EPI mode (read once by the library); MUSE_G256P=0/1 is re-read through a second library handle
is NOT possible, so the old kernel is reached by giving the GEMM a bias of zeros?  No: by MUSE_G256P=0 in a child process.
Usage: python g256p_probe.py check|time   (env: MUSE_G256P, MUSE_G256P_EPI)"""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "open-muse_amd"))
import numpy as np
import torch
from muse import ops

dev = "cuda"
what = sys.argv[1] if len(sys.argv) > 1 else "check"
tag = ""


def rnd(shape, seed, scale=1.0):
    rng = np.random.default_rng(seed)
    return torch.from_numpy((rng.standard_normal(shape) * scale).astype(np.float32))


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


if what == "check":
    import hashlib
    for (M, N, K, la, lb) in [(520, 264, 200, 0, 0), (1000, 520, 712, 0, 1), (1028, 2304, 768, 0, 0), (16448, 768, 768, 0, 0),
                              (16448, 6144, 768, 0, 0), (16448, 768, 6144, 0, 1), (4352, 3072, 768, 0, 1), (776, 264, 328, 1, 1),
                              (520, 520, 264, 1, 0)]:
        A, B = rnd((M, K), 1).to(torch.bfloat16), rnd((N, K), 2).to(torch.bfloat16)
        ref = A.double() @ B.double().t()
        Ad = (A if la == 0 else A.t().contiguous()).to(dev)
        Bd = (B if lb == 0 else B.t().contiguous()).to(dev)
        lda, ldb = (K if la == 0 else M), (K if lb == 0 else N)
        for od in (torch.bfloat16, torch.float32):
            C = torch.full((M, N), float("nan"), dtype=od, device=dev)
            ops.gemm(Ad, Bd, C, M, N, K, la=la, lb=lb, lda=lda, ldb=ldb, ldc=N)
            e = rel_err(C.float(), ref)
            h = hashlib.sha1(C.cpu().view(torch.uint8).numpy().tobytes()).hexdigest()[:12]
            same = True
            for _ in range(8):   # race screen: bit-identical reruns
                C2 = torch.full((M, N), float("nan"), dtype=od, device=dev)
                ops.gemm(Ad, Bd, C2, M, N, K, la=la, lb=lb, lda=lda, ldb=ldb, ldc=N)
                same = same and torch.equal(C.view(torch.int16 if od == torch.bfloat16 else torch.int32), C2.view(torch.int16 if od == torch.bfloat16 else torch.int32))
            print(f"[ {M}x{N}x{K} la{la} lb{lb} {str(od)[6:]}: rel {e:.2e} sha {h} rerun_identical {same}", flush=True)
        # f32 + residual, accumulate
        res = rnd((M, N), 3).to(dev)
        C = torch.empty((M, N), dtype=torch.float32, device=dev)
        ops.gemm(Ad, Bd, C, M, N, K, la=la, lb=lb, lda=lda, ldb=ldb, ldc=N, residual=res, ldr=N)
        e1 = rel_err(C, ref + res.cpu().double())
        C3 = torch.ones((M, N), dtype=torch.float32, device=dev)
        ops.gemm(Ad, Bd, C3, M, N, K, la=la, lb=lb, lda=lda, ldb=ldb, ldc=N, accumulate=True)
        e2 = rel_err(C3, ref + 1.0)
        C4 = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        ops.gemm(Ad, Bd, C4, M, N, K, la=la, lb=lb, lda=lda, ldb=ldb, ldc=N, alpha=0.5)
        e3 = rel_err(C4.float(), 0.5 * ref)
        print(f"[    +residual {e1:.2e}  accumulate {e2:.2e}  alpha {e3:.2e}", flush=True)
    c = ops.lib()
    print(f"[ check done")
else:
    T, H, I, V = 16448, 768, 3072, 2048
    reps = int(os.environ.get("REPS", "20"))
    x = torch.randn(T, H, device=dev).to(torch.bfloat16)
    xi = torch.randn(T, I, device=dev).to(torch.bfloat16)
    x3 = torch.randn(T, 3 * H, device=dev).to(torch.bfloat16)
    x6 = torch.randn(T, 2 * I, device=dev).to(torch.bfloat16)
    wqkv = (torch.randn(3 * H, H, device=dev) * 0.03).to(torch.bfloat16)
    wo = (torch.randn(H, H, device=dev) * 0.03).to(torch.bfloat16)
    w01 = (torch.randn(2 * I, H, device=dev) * 0.03).to(torch.bfloat16)
    w2 = (torch.randn(H, I, device=dev) * 0.03).to(torch.bfloat16)
    wv = (torch.randn(V, H, device=dev) * 0.03).to(torch.bfloat16)
    res = torch.randn(T, H, device=dev)
    o3, o1, o6 = torch.empty_like(x3), torch.empty_like(x), torch.empty_like(x6)
    oi = torch.empty_like(xi)
    of = torch.empty(T, H, device=dev)
    ov = torch.empty(T, V, device=dev, dtype=torch.bfloat16)
    cases = [
        ("fwd QKV    [T,768]x[2304,768]^T", lambda: ops.linear(x, wqkv, out=o3), 2.0 * T * H * 3 * H),
        ("fwd out    [T,768]x[768,768]^T", lambda: ops.linear(x, wo, out=o1), 2.0 * T * H * H),
        ("fwd FFN-in [T,768]x[6144,768]^T", lambda: ops.linear(x, w01, out=o6), 2.0 * T * H * 2 * I),
        ("fwd FFN-out[T,3072]x[768,3072]^T f32+res", lambda: ops.linear(xi, w2, out=of, residual=res), 2.0 * T * H * I),
        ("fwd logits [T,768]x[2048,768]^T", lambda: ops.linear(x, wv, out=ov), 2.0 * T * H * V),
        ("dX QKV     [T,2304]x[2304,768]", lambda: ops.linear_dgrad(x3, wqkv, out=o1), 2.0 * T * H * 3 * H),
        ("dX out     [T,768]x[768,768]", lambda: ops.linear_dgrad(x, wo, out=o1), 2.0 * T * H * H),
        ("dX FFN-in  [T,6144]x[6144,768]", lambda: ops.linear_dgrad(x6, w01, out=o1), 2.0 * T * H * 2 * I),
        ("dX FFN-out [T,768]x[768,3072]", lambda: ops.linear_dgrad(x, w2, out=oi), 2.0 * T * H * I),
    ]
    tot = 0.0
    for name, fn, fl in cases:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for r in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / reps)
        ms = sorted(ts)[1]
        tot += ms
        print(f"[ {name}: {ms*1e3:7.1f} us  {fl/ms/1e9:7.1f} TFLOP/s", flush=True)
    print(f"[ layer fwd+dX total {tot*1e3:.1f} us")
