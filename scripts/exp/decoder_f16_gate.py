"""Would the f16-256 DECODER keep north_star's 1e-3 if its 3x3 convolutions ran as ONE half product (operands rounded to IEEE half, f32
accumulation - the TF32-class arithmetic the reference's GPU path uses by default) instead of bf16x3?  CPU emulation through the oracle.
Measured: 1.9e-3 .. 2.3e-3 of max|image| over two weight seeds - outside the tolerance; rejected (bf16x3 decode: 2e-4).
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests","golden"))
import torch, weights as W
from oracle import maskgit_oracle as O
torch.set_num_threads(8)
MODE=["f32"]
orig=O._conv_same
def r16(x): return x.to(torch.float16).float()
def hook(x,w,b):
    if MODE[0]=="f16" and w.shape[-1]==3 and w.shape[1]%64==0:
        x, w = r16(x), r16(w)
    return orig(x,w,b)
O._conv_same=hook
cfg=W.VQGAN_F16
for seed in (600, 1234):
    sd=W.fill_state_dict(W.vqgan_shapes(cfg), seed, "vqgan")
    px=W.images(2,256,seed+1)
    with torch.no_grad():
        MODE[0]="f32"
        z,zq,idx=O.vqgan_encode(sd,cfg,px)
        rec=O.vqgan_decode_code(sd,cfg,idx)
        MODE[0]="f16"
        rec16=O.vqgan_decode_code(sd,cfg,idx)
    print(seed, "decode f16-operand convs vs f32: max|d|/max|ref| =", float((rec16-rec).abs().max()/rec.abs().max()), "rms rel", float((rec16-rec).pow(2).mean().sqrt()/rec.pow(2).mean().sqrt()))
