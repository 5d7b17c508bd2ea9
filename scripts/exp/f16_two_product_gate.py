"""Would a TWO-product fp16 convolution keep the VQ indices of the bench batch?  (a_hi + a_lo) x w_h with a in two f16 planes (22 bits)
and the weight in ONE f16 plane (11 bits), or the other way round; products exact in f32, f32 accumulation.  The single-plane operand's
rounding is the whole error, so the emulation is the f32 oracle with that operand rounded to f16 in every 3x3 convolution the patch-slab
kernel runs (Cin % 64 == 0).  Companion of winograd_gate.py (same batch, same oracle).   python scripts/exp/f16_two_product_gate.py [n_images]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import weights as W
from oracle import maskgit_oracle as O

torch.set_num_threads(int(os.environ.get("THREADS", "8")))
N_IMG = int(sys.argv[1]) if len(sys.argv) > 1 else 64
MODE = ["f32"]
orig_conv = O._conv_same


def r16(x):
    return x.to(torch.float16).float()


def conv_hook(x, w, b):
    if MODE[0] != "f32" and w.shape[-1] == 3 and w.shape[1] % 64 == 0:
        if MODE[0] == "w_f16":
            w = r16(w)
        elif MODE[0] == "a_f16":
            x = r16(x)
        elif MODE[0] == "w_bf16":
            w = w.to(torch.bfloat16).float()
    return orig_conv(x, w, b)


O._conv_same = conv_hook

if __name__ == "__main__":
    vsd = W.fill_state_dict(W.vqgan_shapes(W.VQGAN_F16), 1234, "vqgan")
    cb = vsd["quantize.embedding.weight"]
    px = torch.rand(64, 3, 256, 256, generator=torch.Generator().manual_seed(1000))[:N_IMG]
    idx = {}
    for mode in ("f32", "w_f16", "a_f16", "w_bf16"):
        MODE[0] = mode
        t0 = time.time()
        out = []
        with torch.no_grad():
            for i in range(0, N_IMG, 4):
                out.append(O.vq_indices(O.vqgan_encoder(vsd, W.VQGAN_F16, px[i:i + 4]), cb).reshape(-1))
        idx[mode] = torch.cat(out)
        print(f"{mode}: {N_IMG} images in {time.time() - t0:.0f} s", flush=True)
    n = idx["f32"].numel()
    for mode in ("w_f16", "a_f16", "w_bf16"):
        print(f"VQ index disagreements vs the f32 oracle, {mode:6s}: {int((idx[mode] != idx['f32']).sum())} of {n}")
