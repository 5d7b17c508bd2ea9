"""VERDICT r5 item 1(b), the numerics half of its go / no-go gate, on the CPU: would Winograd F(2x2, 3x3) on the two 128 -> 128 levels of the
f16 encoder (76 % of the patch-slab kernel's flops), with the input / weight transforms in f32 and the bf16x3 hi / lo split AFTER the
transform, keep the VQ token indices of the bench batch?

The HIP kernel's arithmetic is emulated in torch: a bf16x3 product of f32 operands a, b is  a_lo*b_hi + a_hi*b_lo + a_hi*b_hi  with
hi = bf16(x), lo = bf16(x - hi), products exact in f32 (8 x 8 bit significands), f32 accumulation.  Three encoders over the 64 images
of bench.py's batch (seed 1000, tokenizer weights seed 1234 - the batch `cpu_baseline.vq_index_mismatches_bench_batch` is quoted on):
    f32      the oracle (oracle/maskgit_oracle.py), the reference's arithmetic
    direct   every 3x3 convolution with Cin % 64 == 0 as a direct bf16x3 product (what conv_dma.hip computes)      -> calibrates the emulation
    wino     the same, except Cin = Cout = 128 at 256^2 / 128^2: Winograd F(2x2, 3x3), transforms in f32, bf16x3 products per position
Prints the index disagreements of `direct` and `wino` against f32 out of 16384 tokens, and max |delta| of one level-0 convolution
against its float64 evaluation for both.   python scripts/exp/winograd_gate.py [n_images]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import torch.nn.functional as F
import weights as W
from oracle import maskgit_oracle as O

torch.set_num_threads(int(os.environ.get("THREADS", "8")))
N_IMG = int(sys.argv[1]) if len(sys.argv) > 1 else 64


def split(x):
    hi = x.to(torch.bfloat16).float()
    lo = (x - hi).to(torch.bfloat16).float()
    return hi, lo


def mm_x3(a, b):
    """[.., M, K] x [K, N] as three bf16 products with f32 accumulation"""
    ah, al = split(a)
    bh, bl = split(b)
    return (al @ bh + ah @ bl) + ah @ bh


def conv_direct_x3(x, w):
    """3x3 SAME, NCHW, via im2col: [B*H*W, 9*Cin] x [9*Cin, Cout]"""
    B, C, H, Wd = x.shape
    cols = F.unfold(x, 3, padding=1)                      # [B, C*9, H*W]
    out = mm_x3(cols.transpose(1, 2), w.reshape(w.shape[0], -1).t().contiguous())   # [B, HW, Cout]
    return out.transpose(1, 2).reshape(B, w.shape[0], H, Wd)


# Winograd F(2x2, 3x3) (Lavin & Gray 2016): Y = A^T [ (G g G^T) (.) (B^T d B) ] A
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def conv_wino_x3(x, w, dtype=torch.float32, x3=True):
    B, C, H, Wd = x.shape
    x = x.to(dtype)
    w = w.to(dtype)
    bt, g, at = BT.to(dtype), G.to(dtype), AT.to(dtype)
    xp = F.pad(x, [1, 1, 1, 1])
    t = xp.unfold(2, 4, 2).unfold(3, 4, 2)                # [B, C, H/2, W/2, 4, 4]
    V = bt @ t @ bt.t()                                    # input transform (f32)
    U = g @ w @ g.t()                                      # [Cout, Cin, 4, 4] weight transform (f32)
    V = V.permute(4, 5, 0, 2, 3, 1)                        # [4, 4, B, th, tw, C]
    U = U.permute(2, 3, 1, 0)                              # [4, 4, Cin, Cout]
    M = torch.empty(4, 4, B, H // 2, Wd // 2, w.shape[0], dtype=dtype)
    for i in range(4):
        for j in range(4):
            M[i, j] = mm_x3(V[i, j], U[i, j]) if x3 else V[i, j] @ U[i, j]
    M = M.permute(2, 5, 3, 4, 0, 1)                        # [B, Cout, th, tw, 4, 4]
    Y = at @ M @ at.t()                                    # [B, Cout, th, tw, 2, 2]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(B, w.shape[0], H, Wd)


MODE = ["f32"]
orig_conv = O._conv_same


def conv_hook(x, w, b):
    k = w.shape[-1]
    if MODE[0] != "f32" and k == 3 and w.shape[1] % 64 == 0:
        if MODE[0] == "wino" and w.shape[0] == 128 and w.shape[1] == 128 and x.shape[-1] >= 128:
            y = conv_wino_x3(x, w)
        else:
            y = conv_direct_x3(x, w)
        return y if b is None else y + b.view(1, -1, 1, 1)
    return orig_conv(x, w, b)


O._conv_same = conv_hook

if __name__ == "__main__":
    # one level-0 layer against float64
    g = torch.Generator().manual_seed(5)
    x = F.silu(torch.randn(2, 128, 64, 64, generator=g))
    w = torch.randn(128, 128, 3, 3, generator=g) / (9 * 128) ** 0.5
    ref = F.conv2d(x.double(), w.double(), padding=1)
    sc = float(ref.abs().max())
    for name, y in (("f32 direct (torch)", F.conv2d(x, w, padding=1)), ("bf16x3 direct", conv_direct_x3(x, w)),
                    ("bf16x3 Winograd F(2x2,3x3)", conv_wino_x3(x, w)), ("f32 Winograd F(2x2,3x3)", conv_wino_x3(x, w, x3=False))):
        d = (y.double() - ref).abs()
        print(f"one 128->128 layer, {name:28s}: max|delta| {float(d.max()):.3e} ({float(d.max()) / sc:.2e} of max|y|), rms {float(d.pow(2).mean().sqrt()):.3e}", flush=True)

    vsd = W.fill_state_dict(W.vqgan_shapes(W.VQGAN_F16), 1234, "vqgan")
    cb = vsd["quantize.embedding.weight"]
    gpx = torch.Generator().manual_seed(1000)
    px = torch.rand(64, 3, 256, 256, generator=gpx)[:N_IMG]     # bench.synthetic_batch(64, seed=1000)
    idx = {}
    for mode in ("f32", "direct", "wino"):
        MODE[0] = mode
        t0 = time.time()
        out = []
        with torch.no_grad():
            for i in range(0, N_IMG, 4):
                z = O.vqgan_encoder(vsd, W.VQGAN_F16, px[i:i + 4])
                out.append(O.vq_indices(z, cb).reshape(-1))
        idx[mode] = torch.cat(out)
        print(f"{mode}: {N_IMG} images in {time.time() - t0:.0f} s", flush=True)
    n = idx["f32"].numel()
    for mode in ("direct", "wino"):
        print(f"VQ index disagreements vs the f32 oracle, {mode:6s}: {int((idx[mode] != idx['f32']).sum())} of {n}")
    print(f"                      wino vs direct          : {int((idx['wino'] != idx['direct']).sum())} of {n}")
