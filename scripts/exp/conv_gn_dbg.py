import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "open-muse_amd"))
import torch
from muse import ops
from muse._hip import lib
dev = "cuda"
torch.manual_seed(0)
for (B, H, W, Cin, Cout) in [(1, 16, 16, 64, 128), (1, 16, 16, 128, 128), (2, 32, 32, 128, 128)]:
    x = torch.randn((B, H, W, Cin), device=dev)
    gamma, beta = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.1
    w = torch.randn((Cout, 3, 3, Cin), device=dev) / (9 * Cin) ** 0.5
    w_hi, w_lo = ops.split_bf16(w)
    nchunk = lib().muse_groupnorm_nchunk(H * W)
    part = torch.empty(B * nchunk * 32 * 2, dtype=torch.float64, device=dev)
    hi = torch.empty(x.shape, dtype=torch.bfloat16, device=dev); lo = torch.empty_like(hi)
    ops.check(lib().muse_groupnorm_silu_nhwc_split(x.data_ptr(), hi.data_ptr(), lo.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                                   part.data_ptr(), 0, B, H * W, Cin, 32, 1e-6, 1, ops.stream()), "gn")
    ref = ops.conv2d_nhwc_split2(hi, lo, w_hi, w_lo, B, H, W, Cin, Cout)
    sc, sh = ops.groupnorm_scale_shift((part, nchunk), gamma, beta, B, H * W, Cin)
    got = ops.conv2d_nhwc_gn_split2(x, sc, sh, w_hi, w_lo, B, H, W, Cin, Cout)
    d = (got - ref).abs()
    print((B, H, W, Cin, Cout), "max diff", float(d.max()), "ref max", float(ref.abs().max()), "frac nonzero", float((d > 0).float().mean()))
    # which pixels / channels differ
    bad = (d > 1e-3 * ref.abs().max())
    print("  badly wrong:", int(bad.sum()), "of", bad.numel(), "pixels (y,x) with any bad:", bad.any(-1)[0].nonzero()[:12].tolist())
    # check scale/shift against torch
    xn = torch.nn.functional.group_norm(x.permute(0, 3, 1, 2), 32, gamma, beta, 1e-6).permute(0, 2, 3, 1)
    xa = x * sc[:, None, None, :] + sh[:, None, None, :]
    print("  scale/shift vs torch group_norm:", float((xa - xn).abs().max()))
