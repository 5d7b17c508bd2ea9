"""Golden vectors of the REAL reference (/root/reference, importable only in the build container) run in its `enable_tf32` regime, EMULATED
on the CPU: configs/cc12m_uvit_clip.yaml:102-103 trains f32 tensors with `torch.backends.cuda.matmul.allow_tf32 = True`
(training/train_muse.py:255-256) - on the GPUs the reference targets every matmul (nn.Linear forward / dX / dW, baddbmm / matmul of the
attention core, 1x1 / 2x2 convolutions through cuDNN) rounds its operands to TF32's 10-bit mantissa and accumulates in f32.  Here the
same model, weights and inputs as golden_uvit_full (make_golden.py) with exactly that operand rounding applied by patching
torch.nn.functional.linear / conv2d, torch.baddbmm and torch.matmul with autograd Functions that round both operands of the forward product
AND of the two backward products (round to nearest even at 10 mantissa bits); depthwise convolutions, norms, softmax, GELU, loss stay f32
as on the GPU.  Output: tests/golden/uvit_full_tf32emu.npz (same keys as uvit_full.npz).  Re-run:

    python tests/golden/make_golden_tf32.py

What it pins: the "f16" compute mode of this package (one IEEE-half MFMA product per matmul: the same 10-bit mantissa) against the
arithmetic the YAML actually asks for, not only against the f32 run (tests/test_gpu_uvit.py::test_uvit_config4_vs_reference_golden).
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")

import weights as W  # noqa: E402

torch.set_num_threads(int(os.environ.get("THREADS", "8")))


def tf32(x):
    """round to nearest even at 10 mantissa bits (f32 in, f32 out; strides / memory format kept: the reference views conv outputs)"""
    i = x.view(torch.int32)
    return ((i + 0xFFF + ((i >> 13) & 1)) & ~0x1FFF).view(torch.float32)


_linear, _conv2d, _matmul, _baddbmm = F.linear, F.conv2d, torch.matmul, torch.baddbmm


class _Lin(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return _linear(tf32(x), tf32(w))

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g2, x2 = tf32(g).reshape(-1, g.shape[-1]), tf32(x).reshape(-1, x.shape[-1])
        return _matmul(tf32(g), tf32(w)), _matmul(g2.t(), x2)


class _MM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return _matmul(tf32(a), tf32(b))

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        return _matmul(tf32(g), tf32(b).transpose(-1, -2)), _matmul(tf32(a).transpose(-1, -2), tf32(g))


class _Conv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, stride, padding):
        ctx.save_for_backward(x, w)
        ctx.sp = (stride, padding)
        return _conv2d(tf32(x), tf32(w), None, stride, padding)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        stride, padding = ctx.sp
        gx = torch.nn.grad.conv2d_input(x.shape, tf32(w), tf32(g), stride, padding)
        gw = torch.nn.grad.conv2d_weight(tf32(x), w.shape, tf32(g), stride, padding)
        return gx, gw, None, None


def linear_tf32(x, w, b=None):
    y = _Lin.apply(x, w)
    return y if b is None else y + b


def conv2d_tf32(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    if groups != 1:                       # depthwise 3x3: no tensor-core path, f32 on the GPU too
        return _conv2d(x, w, b, stride, padding, dilation, groups)
    y = _Conv.apply(x, w, stride, padding)
    return y if b is None else y + b.view(1, -1, 1, 1)


def matmul_tf32(a, b):
    return _MM.apply(a, b)


def baddbmm_tf32(input, batch1, batch2, *, beta=1, alpha=1):      # noqa: A002  (torch.baddbmm's own parameter names: the reference calls it by keyword)
    return beta * input + alpha * _MM.apply(batch1, batch2)


def np_(t):
    return t.detach().cpu().numpy()


def golden_uvit_full_tf32(name, batch, seq, text_len, seed):
    from muse.modeling_transformer_v2 import MaskGiTUViT_v2
    model = MaskGiTUViT_v2(**W.UVIT_CC12M)
    assert sum(p.numel() for p in model.parameters()) == 728725504
    model.load_state_dict(W.fill_by_shapes({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed), strict=True)
    model.train()
    ids, enc, cond, micro, labels = W.uvit_inputs(batch, seq, text_len, seed + 1)
    F.linear, F.conv2d, torch.matmul, torch.baddbmm = linear_tf32, conv2d_tf32, matmul_tf32, baddbmm_tf32
    torch.nn.functional.linear = linear_tf32
    try:
        logits, loss = model(ids, enc, cond, micro, labels=labels)
        loss.backward()
    finally:
        F.linear, F.conv2d, torch.matmul, torch.baddbmm = _linear, _conv2d, _matmul, _baddbmm
    out = dict(loss=np_(loss), batch=np.int64(batch), seq=np.int64(seq), text_len=np.int64(text_len), seed=np.int64(seed),
               logits=np_(W.subsample(logits, 16384)), logits_absmax=np_(logits.abs().max()), logits_norm=np_(logits.double().norm()),
               logits_shape=np.array(logits.shape, dtype=np.int64))
    params = dict(model.named_parameters())
    for k in W.UVIT_FULL_GRAD_KEYS:
        g = params[k].grad.float()
        out["grad." + k] = np_(W.subsample(g))
        out["absmax." + k] = np_(g.abs().max())
        out["norm." + k] = np_(g.double().norm())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    ref = np.load(os.path.join(HERE, "uvit_full.npz"))
    el = float(np.abs(out["logits"] - ref["logits"]).max()) / float(ref["logits_absmax"])
    eg = max(float(np.abs(out["grad." + k] - ref["grad." + k]).max()) / float(ref["absmax." + k]) for k in W.UVIT_FULL_GRAD_KEYS)
    print(name, "loss", float(loss), "| against the reference's f32 run (uvit_full.npz): logits", f"{el:.2e}", "loss",
          f"{abs(float(loss) - float(ref['loss'])) / float(ref['loss']):.1e}", "worst gradient", f"{eg:.1e}")


if __name__ == "__main__":
    g = np.load(os.path.join(HERE, "uvit_full.npz"))
    golden_uvit_full_tf32("uvit_full_tf32emu", int(g["batch"]), int(g["seq"]), int(g["text_len"]), int(g["seed"]))
