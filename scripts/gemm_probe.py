"""micro-benchmark of the MFMA GEMM variants on the shapes of the config-B train step (for rocprofv3 --pmc runs)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "open-muse_amd"))
import torch
from muse import ops

dev = "cuda"
T, H, I = 16448, 768, 3072
reps = int(os.environ.get("REPS", "10"))
which = os.environ.get("WHICH", "nn,nt,tt,conv").split(",")
x = torch.randn(T, H, device=dev).to(torch.bfloat16)
w01 = (torch.randn(2 * I, H, device=dev) * 0.03).to(torch.bfloat16)
ab = torch.empty(T, 2 * I, dtype=torch.bfloat16, device=dev)
dab = torch.randn(T, 2 * I, device=dev).to(torch.bfloat16)
dx = torch.empty(T, H, dtype=torch.bfloat16, device=dev)
dw = torch.zeros(2 * I, H, device=dev)


def timeit(name, fn, flops):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name}: {ms*1e3:.1f} us  {flops/ms/1e9:.1f} TFLOP/s", flush=True)


if "nn" in which:
    timeit("linear fwd  [16448x768]x[6144x768]^T", lambda: ops.linear(x, w01, out=ab), 2.0 * T * H * 2 * I)
if "nt" in which:
    timeit("linear dgrad [16448x6144]x[6144x768]", lambda: ops.linear_dgrad(dab, w01, out=dx), 2.0 * T * H * 2 * I)
if "tt" in which:
    timeit("linear wgrad [6144x16448]x[16448x768]", lambda: ops.linear_wgrad(dab, x, dw, True), 2.0 * T * H * 2 * I)
if "grp" in which:
    # the four weight gradients of a config-B layer as ONE grouped launch (muse_gemm_group), alone on the chip: 1 and 5 K slices
    shapes = [(2 * I, H), (H, I), (3 * H, H), (H, H)]
    items = []
    for n_out, k_in in shapes:
        items.append((torch.randn(T, n_out, device=dev).to(torch.bfloat16), torch.randn(T, k_in, device=dev).to(torch.bfloat16),
                      torch.zeros(n_out, k_in, device=dev), False, None, None))
    fl = sum(2.0 * T * a * b for a, b in shapes)
    for sk in (1, 5):
        timeit(f"grouped layer dW, {sk} slice(s) [144 tiles x 257 K-tiles]", lambda: ops.linear_wgrad_group(items, None, split=sk), fl)
if "conv" in which:
    B, Hh, Ww, C = 64, 128, 128, 128
    xi = torch.randn(B, Hh, Ww, C, device=dev)
    w = torch.randn(C, 3, 3, C, device=dev) * 0.03
    wh, wl = ops.split_bf16(w)
    fl = 2.0 * B * Hh * Ww * C * 9 * C
    timeit("conv bf16x3 64x128x128x128 3x3", lambda: ops.conv2d_nhwc_split(xi, wh, wl, B, Hh, Ww, C, C, 3), fl)
    xh, xl = ops.split_bf16(xi)
    timeit("conv bf16x3 DMA 64x128x128x128 3x3", lambda: ops.conv2d_nhwc_split2(xh, xl, wh, wl, B, Hh, Ww, C, C), fl)
    # the fused form (GroupNorm + SiLU + split of the input inside the convolution): statistics as a producer leaves them
    y0 = ops.conv2d_nhwc_split2(xh, xl, wh, wl, B, Hh, Ww, C, C, gn_groups=32)
    gam, bet = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    sc, sh = ops.groupnorm_scale_shift(y0._gn_stats, gam, bet, B, Hh * Ww, C)
    timeit("conv bf16x3 GN-fused 64x128x128x128 3x3", lambda: ops.conv2d_nhwc_gn_split2(y0, sc, sh, wh, wl, B, Hh, Ww, C, C), fl)
    for (b2, h2, c2) in ((16, 256, 128), (64, 64, 256), (64, 16, 512)):
        x2 = torch.randn(b2, h2, h2, c2, device=dev)
        w2 = torch.randn(c2, 3, 3, c2, device=dev) * 0.03
        w2h, w2l = ops.split_bf16(w2)
        x2h, x2l = ops.split_bf16(x2)
        fl2 = 2.0 * b2 * h2 * h2 * c2 * 9 * c2
        timeit(f"conv bf16x3     {b2}x{h2}x{h2}x{c2}", lambda: ops.conv2d_nhwc_split(x2, w2h, w2l, b2, h2, h2, c2, c2, 3), fl2)
        timeit(f"conv bf16x3 DMA {b2}x{h2}x{h2}x{c2}", lambda: ops.conv2d_nhwc_split2(x2h, x2l, w2h, w2l, b2, h2, h2, c2, c2), fl2)
    timeit("conv f32    64x128x128x128 3x3", lambda: ops.conv2d_nhwc(xi, w, B, Hh, Ww, C, C, 3), fl)
    xb, wb = xi.to(torch.bfloat16), w.to(torch.bfloat16)
    timeit("conv bf16   64x128x128x128 3x3", lambda: ops.conv2d_nhwc(xb, wb, B, Hh, Ww, C, C, 3), fl)
