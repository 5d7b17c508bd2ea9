import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "open-muse_amd"))
import torch
from muse import ops
from muse._hip import lib
dev = "cuda"
torch.manual_seed(0)
B, H, W, Cin = 1, 32, 32, 128
Cout = Cin
x = torch.randn((B, H, W, Cin), device=dev)
gamma, beta = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.1
nchunk = lib().muse_groupnorm_nchunk(H * W)
part = torch.empty(B * nchunk * 32 * 2, dtype=torch.float64, device=dev)
hi = torch.empty(x.shape, dtype=torch.bfloat16, device=dev); lo = torch.empty_like(hi)
ops.check(lib().muse_groupnorm_silu_nhwc_split(x.data_ptr(), hi.data_ptr(), lo.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                               part.data_ptr(), 0, B, H * W, Cin, 32, 1e-6, 1, ops.stream()), "gn")
sc, sh = ops.groupnorm_scale_shift((part, nchunk), gamma, beta, B, H * W, Cin)
for (ky, kx) in [(1, 1), (0, 0), (2, 1)]:
    w = torch.zeros((Cout, 3, 3, Cin), device=dev)
    for o in range(Cout):
        w[o, ky, kx, o] = 1.0
    w_hi, w_lo = ops.split_bf16(w)
    ref = ops.conv2d_nhwc_split2(hi, lo, w_hi, w_lo, B, H, W, Cin, Cout)
    got = ops.conv2d_nhwc_gn_split2(x, sc, sh, w_hi, w_lo, B, H, W, Cin, Cout)
    d = (got - ref).abs()[0]
    print("tap", (ky, kx), "max diff", float(d.max()), "nonzero frac", float((d > 0).float().mean()))
    nz = (d > 0)
    print("   per-channel count of differing pixels:", nz.sum((0, 1)).tolist())
    ys, xs, cs = nz.nonzero(as_tuple=True)
    for i in range(min(6, len(ys))):
        y, xx, c = int(ys[i]), int(xs[i]), int(cs[i])
        print("   ", (y, xx, c), "got", float(got[0, y, xx, c]), "ref", float(ref[0, y, xx, c]), "hi+lo", float(hi[0, y + ky - 1, xx + kx - 1, c].float() + lo[0, y + ky - 1, xx + kx - 1, c].float()) if 0 <= y + ky - 1 < H and 0 <= xx + kx - 1 < W else None)
