"""Thin Python wrappers over the libmuse_hip C-ABI (one function per kernel entry point, no autograd here).

Tensors are allocated by PyTorch; every wrapper enqueues on torch's current HIP stream.  All of them raise
MuseHipError if the tensors are not on the GPU or the native library is missing: there is no eager fallback.
"""
from __future__ import annotations

import ctypes as C
import math
import os

import torch

from . import _hip
from ._hip import BF16, F16, F32, GemmDesc, check, dt, lib, ptr, require_gpu, stream

# MUSE_GEMM_TR=0 routes k-major GEMM operands through an explicit transpose + the k-contiguous path instead of the
# ds_read_b64_tr_b16 path (debug / bring-up switch; both run on the GPU through libmuse_hip).
USE_TR = os.environ.get("MUSE_GEMM_TR", "1") != "0"


def _esz(t):
    return t.element_size()


# ---- optional per-launch timing of the MFMA kernels (bench.py's roofline leg): HIP events on the launch stream ----------
_PROF = None


PROF_BYTES = {}   # kernel family -> algorithmic HBM bytes of the instrumented launches (each operand read / written once)


def profile_start():
    global _PROF
    _PROF = []
    PROF_BYTES.clear()


def profile_stop(with_kind=False):
    """-> list of (kernel name, algorithmic work, milliseconds) per launch; work = flops for the MFMA kernels.  with_kind adds
    the HBM-bound kernels (work = algorithmic bytes: every operand read or written once) and a 4th field "flop" | "byte"."""
    global _PROF
    rec, _PROF = _PROF, None
    torch.cuda.synchronize()
    out = [(n, f, e0.elapsed_time(e1), k) for n, f, e0, e1, k in rec]
    return out if with_kind else [(n, f, t) for n, f, t, k in out if k == "flop"]


def _prof_begin():
    if _PROF is None:
        return None
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    return e0


def _prof_end(e0, name, work, kind="flop"):
    if e0 is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        _PROF.append((name, work, e0, e1, kind))


def _touched(t):
    """a HIP kernel of this package just rewrote t through its raw pointer: bump autograd's version counter, the way a torch in-place
    op would - X3Images keys a tensor's cached operand planes by (storage, shape, version), so a rewritten tensor never finds the planes
    of its previous contents"""
    torch.autograd.graph.increment_version(t)
    return t


def _nbytes(*tensors):
    return float(sum(t.numel() * t.element_size() for t in tensors if t is not None))


def capture_graph(fn):
    """Run fn() once eagerly (one-time packing / first-use work stays out of the graph), then capture a second run into a
    HIP graph: every libmuse_hip launch inside goes to torch's capture stream (`stream()` is torch's current stream) and the
    tensors fn allocates come from the graph's private pool.  -> (graph, fn's captured return value); graph.replay() re-runs
    it on the inputs' current contents."""
    fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = fn()
    return graph, out


SKINNY = int(os.environ.get("MUSE_GEMM_SKINNY", "1"))   # split-K + fused reduction for forward products of <= 2048 rows: 1 = inside graph capture


def gemm(A, B, C_, M, N, K, *, la=0, lb=0, lda, ldb, ldc, a_off=0, b_off=0, c_off=0, alpha=1.0, bias=None, rowvec=None,
         residual=None, ldr=0, batch=1, zdiv=1, sA=(0, 0), sB=(0, 0), sC=(0, 0), accumulate=False, act=0, split_k=1,
         split_stride=0, x3_lo=None):
    """C[z] = epilogue(alpha * A[z] @ B[z]^T).  A/B/C_ are tensors, *_off element offsets of the (0,0) entry.

    la/lb = 0: operand(r,k) at base + r*ld + k;  1: at base + k*ld + r (see include/muse_hip.h).
    x3_lo = (a_lo, b_lo): A / B are the bf16 hi planes of a bf16x3 product, their lo planes that many elements behind them
    (muse_gemm_x3); returns None when that kernel does not take the product.
    """
    require_gpu(A, B, C_)
    if (_F32_AS_F16[0] and C_.dtype == torch.float32 and (A.dtype, B.dtype) in _F16_MODE_PAIRS and act == 0 and split_k == 1 and batch == 1
            and _gemm_f16(A, B, C_, M, N, K, la, lb, lda, ldb, ldc, c_off, alpha, bias, rowvec, residual, ldr, accumulate, a_off, b_off)):
        return C_
    if A.dtype != B.dtype:
        raise _hip.MuseHipError("gemm operands must share a dtype")
    if _F32_AS_BF16X3[0] and A.dtype == torch.float32 and C_.dtype == torch.float32 and act == 0 and split_k == 1:
        done = _gemm_bf16x3(A, B, C_, M, N, K, la, lb, lda, ldb, ldc, a_off, b_off, c_off, alpha, bias, rowvec, residual, ldr, batch,
                            zdiv, sA, sB, sC, accumulate)
        if done:
            return C_
    if isinstance(A, Planes) or isinstance(B, Planes):
        # a planes-only operand has no f32 bytes behind data_ptr(): every path below would read its bf16 planes as something else
        raise _hip.MuseHipError("a planes-only operand reached a product the four-plane kernel does not take")
    if (SKINNY and x3_lo is None and A.dtype == torch.bfloat16 and batch == 1 and la == 0 and lb == 0 and act == 0 and rowvec is None and not accumulate
            and split_k == 1 and M <= 2048 and K >= 512 and N % 4 == 0 and ldc % 4 == 0 and (residual is None or ldr % 4 == 0)
            and (SKINNY == 2 or torch.cuda.is_current_stream_capturing())):
        # small-batch decoding: a forward Linear of a few hundred rows has too few 128^2 tiles for 256 CUs ([512 x 1024] x [1024 x 1024]^T:
        # 32 tiles, 36 us of a 16-K-tile loop on an eighth of the chip; 77 % of a 512-row U-ViT forward was such launches,
        # profiles/r04_decode_kernel_stats_before.csv).  K is cut so that tiles x slices fill the chip (>= 2 K-tiles per slice), the
        # slices go to an f32 workspace and ONE row kernel sums them in fixed order and applies the Linear's epilogue (bias, residual,
        # output dtype).  Only while the forward is being CAPTURED into a HIP graph (generate2(hip_graph=True), PipelineMuse): the path
        # trades kernel time for launches (+36 % launches), and an eager small-batch forward is bound by per-launch host time - measured
        # on MI355X, PipelineMuse at batch 1 (profiles/r04_latency_ab.txt): eager 123.5 ms, eager + split-K 152.3, graph 112.2,
        # graph + split-K 98.9.  (MUSE_GEMM_SKINNY=2 forces it everywhere, 0 turns it off.)
        t128 = ((M + 127) // 128) * ((N + 127) // 128)
        sk = min(256 // t128, K // 128, 16) if t128 <= 96 else 1
        if sk >= 2:
            # the kernels cut K on 64-wide tile boundaries, ceil(nk / sk) tiles per slice: a slice that starts at or beyond nk returns
            # without writing its part of the workspace, so only the slices that hold work may be summed (M=256, K=2816: 16 -> 15)
            nk = (K + 63) // 64
            per = (nk + sk - 1) // sk
            sk = (nk + per - 1) // per
        if sk >= 2:
            ws = torch.empty((sk, M, N), dtype=torch.float32, device=A.device)
            gemm(A, B, ws, M, N, K, la=0, lb=0, lda=lda, ldb=ldb, ldc=N, a_off=a_off, b_off=b_off, alpha=alpha, split_k=sk, split_stride=M * N)
            check(lib().muse_sum_slices_epilogue(ws.data_ptr(), sk, M * N, ptr(bias), ptr(residual), ldr, C_.data_ptr() + c_off * _esz(C_),
                                                 dt(C_), ldc, M, N, stream()), "muse_sum_slices_epilogue")
            return _touched(C_)
    # (Measured and dropped, round 5: the ragged last row tile of M = 16448 as a launch of its own wherever it saves a round of the
    #  256-CU chip - N = 6144 / 3072 / 2048: the step came out 0.45 ms SLOWER, 53.08 against 52.64 ms same box, transformer alone 33.34
    #  against 32.85: the extra launches cost more than the rounds, and in the step other streams fill the tail anyway.)
    if not USE_TR and A.dtype == torch.bfloat16 and (la == 1 or lb == 1):
        if x3_lo is not None:
            return None        # (the bring-up transpose route would drop the lo planes: the caller falls back to its three-product form)
        return _gemm_via_transpose(A, B, C_, M, N, K, la, lb, lda, ldb, ldc, a_off, b_off, c_off, alpha, bias, rowvec,
                                   residual, ldr, batch, zdiv, sA, sB, sC, accumulate, act)
    d = GemmDesc()
    d.A = A.data_ptr() + a_off * _esz(A)
    d.B = B.data_ptr() + b_off * _esz(B)
    d.C = C_.data_ptr() + c_off * _esz(C_)
    d.bias = ptr(bias)
    d.rowvec = ptr(rowvec)
    d.residual = ptr(residual)
    d.dtype = F16 if A.dtype == torch.float16 else dt(A)
    d.out_dtype = dt(C_)
    d.layout_a, d.layout_b = la, lb
    d.M, d.N, d.K = M, N, K
    d.batch, d.zdiv = batch, zdiv
    d.lda, d.ldb, d.ldc, d.ldr = lda, ldb, ldc, ldr
    d.sA0, d.sA1 = sA
    d.sB0, d.sB1 = sB
    d.sC0, d.sC1 = sC
    # (half operand images carry the power of two they were scaled by: the product is handed back unscaled)
    d.alpha = alpha / (getattr(A, "_muse_scale", 1.0) * getattr(B, "_muse_scale", 1.0)) if d.dtype == F16 else alpha
    d.accumulate = 1 if accumulate else 0
    d.act = act
    d.split_k = split_k
    d.split_stride = split_stride
    e0 = _prof_begin()
    if x3_lo is not None:
        rc = lib().muse_gemm_x3(C.byref(d), int(x3_lo[0]), int(x3_lo[1]), stream())
        if rc == -3:         # MUSE_ERR_UNSUPPORTED: a shape the four-plane kernel does not take
            return None
        check(rc, "muse_gemm_x3")
        _prof_end(e0, f"gemm_bf16x3_{'NT'[la]}{'NT'[lb]}", 2.0 * M * N * K * batch)
        return _touched(C_)
    check(lib().muse_gemm(C.byref(d), stream()), "muse_gemm")
    _prof_end(e0, f"gemm_{ {BF16: 'bf16', F16: 'f16'}.get(d.dtype, 'f32')}_{'NT'[la]}{'NT'[lb]}", 2.0 * M * N * K * batch)
    return _touched(C_)


# ---- "f16": every f32 weight GEMM as ONE product of IEEE-half operand images ----------------------------------------------------------
# configs/cc12m_uvit_clip.yaml:102-103 trains f32 tensors with `enable_tf32`: the products round their operands to TF32 (10-bit mantissa,
# 8-bit exponent) and accumulate in f32.  gfx950 has no xf32 MFMA; IEEE half has the SAME 10-bit mantissa, and its MFMA
# (v_mfma_f32_16x16x32_f16, f32 accumulation) runs at the bf16 rate - a third of the "bf16x3" mode's cost for the operand precision the
# YAML asks for.  What half lacks is TF32's exponent range, which is the host's job here: an operand image is half(x * s) with s a power
# of two (exact) and the product's alpha carries 1 / (s_a s_b).  Forward operands (normalised activations, weights) use s = 1; every
# GRADIENT operand of a backward pass uses the pass's `grad_scale` (per-token loss gradients are ~1 / tokens: far below half's normal
# range unscaled).  A finite element beyond +-65504 / s becomes inf - the product and the step's gradients turn NaN rather than silently
# wrong - and is counted, by the cast kernel and by every producer kernel; non-zero elements a cast rounds to zero are counted too
# (F16Images.stats()).  The fp16-training recipe applies on top: TapeOps.f16_update_grad_scale() halves the scale and has the caller
# skip the step.  Products the 256^2 half kernels refuse stay in exact f32.
_F32_AS_F16 = [False]
_F16_IMAGES = [None]
_F16_GUARD = [None]     # the F16Images of the last f16 BACKWARD pass: muse.FusedAdamW guards its update with that pass's overflow counter
# operand dtypes of a product the mode converts: f32 tensors, or an f32 tensor against an operand that already IS a half image (a
# weight's copy kept across steps: tape_ops._wb)
_F16_MODE_PAIRS = {(torch.float32, torch.float32), (torch.float32, torch.float16), (torch.float16, torch.float32)}


class F16Images:
    """The half operand images of ONE training step (the role X3Images plays for the bf16x3 mode): a tensor is converted once per step,
    whichever products read it - keyed by (storage, shape, strides, version, scale); images made during the backward live in a short LRU.
    `backward`: the pass running is a backward pass - the A operand of its products and the dy of its weight gradients are gradients and
    take `grad_scale`.  `keep` (forward passes): images stay until clear() because the backward pass reads them again; a forward that
    records no tape (inference) sets it False and its images share the short LRU, so no activation outlives its consumers."""

    def __init__(self, grad_scale=1.0, recent=12):
        self.persist, self.lru, self.recent, self.backward, self.keep = {}, {}, int(recent), False, True
        self.grad_scale = float(grad_scale)
        self.hits = self.misses = self.produced = 0
        self._stats = None
        self._snaps, self._free, self.totals = [], [], [0, 0]      # counter copies in flight (pinned buffer, event), free ones, running totals

    def clear(self):
        self.persist.clear()
        self.lru.clear()

    def set_grad_scale(self, s):
        if s <= 0 or math.frexp(float(s))[0] != 0.5:
            raise _hip.MuseHipError(f"f16 mode: the gradient scale must be a power of two, got {s}")
        self.grad_scale = float(s)

    def _ensure_stats(self, device=None):
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        if self._stats is None or self._stats.device != device:
            self._stats = torch.zeros(2, dtype=torch.int32, device=device)
        return self._stats

    def stats(self, reset=True):
        """(operand elements / 4-element groups that overflowed half's range, non-zero elements rounded to zero by a cast) since the
        last reset - one device read (plus what optimizer steps have already taken off the device counters: after_optimizer_step)"""
        self.consume_snapshots(block=True)
        v = [0, 0] if self._stats is None else self._stats.tolist()
        out = (int(v[0]) + self.totals[0], int(v[1]) + self.totals[1])
        if reset:
            if self._stats is not None:
                self._stats.zero_()
            self.totals = [0, 0]
        return out

    def after_optimizer_step(self):
        """muse.FusedAdamW has launched its update guarded by this pass's overflow counter: take the counters off the device - copied
        to pinned memory behind the update, zeroed behind the copy (all stream-ordered, no host wait) - for the next backward pass,
        which reads the copies that have arrived (consume_snapshots) to move the gradient scale"""
        if self._stats is None:
            return
        buf, ev = self._free.pop() if self._free else (torch.zeros(2, dtype=torch.int32).pin_memory(), torch.cuda.Event())
        buf.copy_(self._stats, non_blocking=True)
        ev.record(torch.cuda.current_stream(self._stats.device))
        self._stats.zero_()
        self._snaps.append((buf, ev))

    def consume_snapshots(self, block=False):
        """-> [(overflowed, rounded to zero), ...] of the guarded optimizer steps whose counter copies have ARRIVED, oldest first.  Never
        waits unless `block`: the host enqueues a step or more ahead of the GPU, and waiting here for the previous step's copy would
        cost that run-ahead (measured: 648 -> 596 images/s on the config-4 leg) - a copy still in flight is read by a later pass."""
        out = []
        while self._snaps:
            buf, ev = self._snaps[0]
            if block:
                ev.synchronize()
            elif not ev.query():
                break
            self._snaps.pop(0)
            v = buf.tolist()
            self._free.append((buf, ev))
            self.totals[0] += int(v[0])
            self.totals[1] += int(v[1])
            out.append((int(v[0]), int(v[1])))
        return out

    def image(self, t, scale):
        if isinstance(t, Planes):          # a producer's result that exists as its half image only
            if not t.half:
                raise _hip.MuseHipError("a bf16x3 planes-only operand reached a half product")
            return t.half_image()
        if t.dtype == torch.float16:       # already an image (a weight's half copy): its own scale rides on it
            return t
        key = (t.data_ptr(), tuple(t.shape), tuple(t.stride()), t._version, float(scale))
        hit = self.persist.get(key)
        if hit is None:
            hit = self.lru.get(key)
        if hit is not None:
            self.hits += 1
            return hit[1]
        self.misses += 1
        out = cast_to_f16(t, scale, self._ensure_stats(t.device))
        if self.backward or not self.keep:
            self.lru[key] = (t, out)
            while len(self.lru) > self.recent:
                self.lru.pop(next(iter(self.lru)))
        else:
            self.persist[key] = (t, out)
        return out

    def put_planes(self, t, planes):
        """a producer kernel wrote t's half image itself ([1, *t.shape], the bits muse_cast_f32_to_f16 makes of t with the scale the
        kernel was given: the pass's gradient scale in a backward pass, 1 in a forward one): no cast pass when a product reads t"""
        scale = self.grad_scale if self.backward else 1.0
        img = planes[0]
        img._muse_scale = scale
        key = (t.data_ptr(), tuple(t.shape), tuple(t.stride()), t._version, float(scale))
        self.produced += 1
        if self.backward or not self.keep:
            self.lru[key] = (t, img)
            while len(self.lru) > self.recent:
                self.lru.pop(next(iter(self.lru)))
        else:
            self.persist[key] = (t, img)


class f32_gemms_as_f16:
    """`images`: the F16Images of the step (None: a private one - every product converts its operands)"""
    def __init__(self, on=True, images=None):
        self.on = bool(on)
        self.images = images if images is not None else (F16Images() if on else None)

    @staticmethod
    def _tell_kernels(on, images):
        # the producer kernels (*_x3 entry points) write half images with this pass's gradient scale while the mode is on
        live = on and images is not None
        check(lib().muse_operand_images(1 if on else 0, float(images.grad_scale) if live else 1.0, images._ensure_stats().data_ptr() if live else None),
              "muse_operand_images")

    def __enter__(self):
        self.prev = (_F32_AS_F16[0], _F16_IMAGES[0])
        _F32_AS_F16[0] = self.on
        if self.on:
            _F16_IMAGES[0] = self.images
        if self.on or self.prev[0]:
            self._tell_kernels(self.on, self.images)
        return self

    def __exit__(self, *exc):
        _F32_AS_F16[0], _F16_IMAGES[0] = self.prev
        if self.on or self.prev[0]:
            self._tell_kernels(self.prev[0], self.prev[1])
        if self.on and self.images is not None and self.images.backward:
            _F16_GUARD[0] = self.images          # (the optimizer step that follows skips its update if this pass overflowed)
        return False


def cast_to_f16(t, scale=1.0, stats=None):
    """half(t * scale) of a contiguous f32 tensor (muse_cast_f32_to_f16); the image remembers its scale (`_muse_scale`: gemm divides alpha
    by it)"""
    require_gpu(t)
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise _hip.MuseHipError("cast_to_f16: contiguous f32 tensor")
    out = torch.empty(t.shape, dtype=torch.float16, device=t.device)
    check(lib().muse_cast_f32_to_f16(t.data_ptr(), out.data_ptr(), t.numel(), float(scale), ptr(stats), stream()), "muse_cast_f32_to_f16")
    out._muse_scale = float(scale)
    return out


def _f16_operands_ok(A, B, M, N, K, la, lb, lda, ldb):
    """whole contiguous 2-D f32 tensors read as [rows, K] / [K, rows] operands with 16-byte half rows (what an image can stand for)"""
    return (A.dim() == 2 and B.dim() == 2 and A.is_contiguous() and B.is_contiguous() and K % 8 == 0 and lda % 8 == 0 and ldb % 8 == 0
            and A.stride(0) == lda and B.stride(0) == ldb and A.shape[1 if la == 0 else 0] == K and B.shape[1 if lb == 0 else 0] == K
            and A.shape[0 if la == 0 else 1] >= M and B.shape[0 if lb == 0 else 1] >= N and M >= 128 and N >= 128 and K >= 64
            and A.numel() * 2 < (1 << 31) and B.numel() * 2 < (1 << 31))


def _f16_kernel_takes(A, B, c_ptr, M, N, K, la, lb, lda, ldb, ldc, residual=None, ldr=0):
    d = GemmDesc()
    d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), c_ptr           # (the f32 tensors' addresses stand in for their images': alignment probe)
    d.dtype, d.out_dtype, d.layout_a, d.layout_b = F16, F32, la, lb
    d.M, d.N, d.K, d.batch, d.zdiv = M, N, K, 1, 1
    d.lda, d.ldb, d.ldc, d.ldr = lda, ldb, ldc, ldr
    d.residual = ptr(residual)
    d.alpha = 1.0
    return lib().muse_gemm_tile(C.byref(d)) == 256


def _gemm_f16(A, B, C_, M, N, K, la, lb, lda, ldb, ldc, c_off, alpha, bias, rowvec, residual, ldr, accumulate, a_off, b_off):
    """-> True when the product ran on half operand images, False when the half kernels do not take it (it then runs in exact f32)"""
    if a_off or b_off or not _f16_operands_ok(A, B, M, N, K, la, lb, lda, ldb):
        return False
    if not _f16_kernel_takes(A, B, C_.data_ptr() + c_off * 4, M, N, K, la, lb, lda, ldb, ldc, residual, ldr):
        return False
    im = _F16_IMAGES[0]
    a16 = im.image(A, im.grad_scale if im.backward else 1.0)
    b16 = im.image(B, 1.0)
    gemm(a16, b16, C_, M, N, K, la=la, lb=lb, lda=lda, ldb=ldb, ldc=ldc, c_off=c_off, alpha=alpha, bias=bias, rowvec=rowvec,
         residual=residual, ldr=ldr, accumulate=accumulate)
    return True


def f16_wgrad_operands(dy, x, M=None, lda=None):
    """the half images (dy with the pass's gradient scale, x unscaled) of a weight-gradient product dy^T x the half kernels take, through
    the running step's image cache - or None (no f16 step running / a shape they refuse: the caller keeps the f32 tensors)"""
    if not (_F32_AS_F16[0] and dy.dtype == torch.float32 and x.dtype == torch.float32 and dy.dim() == 2 and x.dim() == 2
            and (lda is None or lda == dy.stride(0)) and dy.shape[0] == x.shape[0]):
        return None
    N = M if M is not None else dy.shape[1]
    if not (_f16_operands_ok(dy, x, N, x.shape[1], dy.shape[0], 1, 1, dy.stride(0), x.stride(0))
            and _f16_kernel_takes(dy, x, dy.data_ptr(), N, x.shape[1], dy.shape[0], 1, 1, dy.stride(0), x.stride(0), x.shape[1])):
        return None
    im = _F16_IMAGES[0]
    return im.image(dy, im.grad_scale), im.image(x, 1.0)


# ---- "bf16x3": every f32 GEMM as three bf16 MFMA products ---------------------------------------------------------------------------
# The exact-f32 parity mode of the tape engines (MaskGiTUViT, text-conditioned MaskGitTransformer) runs its GEMMs on the f32-input MFMA
# (157 TFLOP/s peak).  configs/cc12m_uvit_clip.yaml:102-103 trains in f32 tensors with TF32 products (10-bit mantissa); gfx950 has no
# xf32 MFMA, the CDNA4 counterpart at or above that precision is the three-product scheme the tokenizer already uses: operands split
# into hi = bf16(x), lo = bf16(x - hi), C = hi*hi + hi*lo + lo*hi with f32 accumulation (the dropped lo*lo term is 2^-16 relative) on
# the bf16 kernels at 1/3 of their rate.  Scoped by a context manager the models enter for "bf16x3" compute; a product the bf16 kernels
# cannot take (operand rows that are not whole 16-byte chunks in bf16) silently stays on the exact-f32 kernel.
_F32_AS_BF16X3 = [False]
_X3_IMAGES = [None]      # the operand-image cache of the pass that is running (X3Images), or None


class X3Images:
    """The bf16x3 operand images of ONE training step, so that a tensor is split once per step instead of once per product.  An
    activation x is the A operand of its Linear's forward product and the B operand of the weight-gradient product; a gradient dY is
    the A operand of the dX product and the A operand of dW; a weight is the B operand of the forward and of the dX product.
    Two image forms:
      * planes [2, rows, cols] (hi, lo) for the four-plane kernel (muse_gemm_x3): ONE image per tensor whatever the product reads it as;
        a producer kernel may hand them in itself (put_planes);
      * K-concatenated images for the plain kernel (shapes muse_gemm_x3 refuses): with the patterns chosen in `_gemm_bf16x3` /
        `linear_wgrad` the row-concatenated image [T, 3K] of the forward / dX product, viewed as [3T, K], IS the token-stacked image of
        the dW product (token t's three thirds are rows 3t .. 3t+2 - a contraction does not care about the order of its slots).
    Keyed by (storage address, shape, strides, autograd version, form); an entry keeps its source tensor alive, so an address cannot
    come back with other contents while the entry exists, and the wrappers of kernels that rewrite a tensor in place bump its version
    (`_touched`).  The owner clears the cache at the start of a forward and at the end of a backward.  Images made during the backward
    (dY) are dead two products later: they live in a short LRU."""

    def __init__(self, recent=12):
        self.persist, self.lru, self.recent, self.backward = {}, {}, int(recent), False
        self.hits = self.misses = self.produced = 0

    def clear(self):
        self.persist.clear()
        self.lru.clear()

    @staticmethod
    def _key(t, layout, lo_pos):
        return (t.data_ptr(), tuple(t.shape), tuple(t.stride()), t._version, layout, lo_pos)

    def get(self, t, layout, K, lo_pos, last_use=False):
        key = self._key(t, layout, lo_pos)
        hit = self.persist.pop(key, None) if last_use else self.persist.get(key)
        if hit is not None and last_use:      # (the weight-gradient product is an activation's last reader - bar siblings that share it, q / k / v)
            self._recent(key, hit)
        if hit is None:
            hit = self.lru.get(key)
        if hit is not None:
            self.hits += 1
            return hit[1], hit[2]
        self.misses += 1
        out, ld = _split_cat3_now(t, layout, K, lo_pos)
        if self.backward or last_use:
            self._recent(key, (t, out, ld))
        else:
            self.persist[key] = (t, out, ld)
        return out, ld

    def planes(self, t):
        """(hi, lo) planes [2, *t.shape] of a contiguous f32 tensor (the operand form of muse_gemm_x3): ONE image per tensor whatever
        the product reads it as (forward A, dX A, dW A / B, a weight in the forward and in dX)"""
        key = self._key(t, 9, 0)
        hit = self.persist.get(key)
        if hit is None:
            hit = self.lru.get(key)
        if hit is not None:
            self.hits += 1
            return hit[1]
        self.misses += 1
        out = _split_planes_now(t)
        if self.backward:
            self._recent(key, (t, out, 0))
        else:
            self.persist[key] = (t, out, 0)
        return out

    def put_planes(self, t, planes):
        """a producer kernel wrote t's planes itself (same bits as the split of t): no split pass when a product reads t"""
        key = self._key(t, 9, 0)
        self.produced += 1
        if self.backward:
            self._recent(key, (t, planes, 0))
        else:
            self.persist[key] = (t, planes, 0)

    def _recent(self, key, entry):
        self.lru[key] = entry
        while len(self.lru) > self.recent:
            self.lru.pop(next(iter(self.lru)))


class Planes:
    """An f32-class GEMM operand that exists ONLY as its (hi, lo) bf16 planes [2, rows, cols] - the result of a producer kernel that nothing
    but weight GEMMs reads (the GLU output and its input gradient inside an MLP, planes_only=True): the f32 tensor is never written.
    Quacks like the contiguous f32 [rows, cols] tensor it stands for as far as the GEMM entry points look (shape, dtype, strides); the
    products that take it are the four-plane ones (muse_gemm_x3) - anything else raises instead of reading bytes that are not there."""
    dtype = torch.float32
    is_cuda = True
    _version = 0

    def __init__(self, planes):
        self.planes = planes
        self.shape = torch.Size(planes.shape[1:])
        self.device = planes.device
        # "f16" mode: ONE IEEE-half image [1, rows, cols] = half(x * scale); the scale is the one the producer kernel applied - the
        # running pass's gradient scale for a backward result, 1 for a forward one (muse_operand_images)
        self.half = planes.dtype == torch.float16
        im = _F16_IMAGES[0]
        self.scale = (im.grad_scale if (im is not None and im.backward) else 1.0) if self.half else 1.0

    def half_image(self):
        t = self.planes[0]
        t._muse_scale = self.scale
        return t

    def dim(self):
        return 2

    def numel(self):
        return self.shape[0] * self.shape[1]

    def stride(self, i=None):
        st = (self.shape[1], 1)
        return st if i is None else st[i]

    def is_contiguous(self):
        return True

    def data_ptr(self):
        return self.planes.data_ptr()      # (only ever probed for alignment)

    def record_stream(self, s):
        self.planes.record_stream(s)


class f32_gemms_as_bf16x3:
    """`images`: an X3Images to share operand images between the products of a step (None: every product splits its operands)"""
    def __init__(self, on=True, images=None):
        self.on = bool(on)
        self.images = images

    def __enter__(self):
        self.prev = (_F32_AS_BF16X3[0], _X3_IMAGES[0])
        _F32_AS_BF16X3[0] = self.on
        if self.on:
            _X3_IMAGES[0] = self.images
        return self

    def __exit__(self, *exc):
        _F32_AS_BF16X3[0], _X3_IMAGES[0] = self.prev
        return False


def split_f32(t):
    """f32 tensor -> (hi, lo) bf16 tensors of the same shape AND strides (a GEMM's offsets / leading dimensions / batch strides
    stay valid); contiguous storage goes through muse_split_f32_to_bf16x2"""
    require_gpu(t)
    if t.is_contiguous():
        hi = torch.empty(t.shape, dtype=torch.bfloat16, device=t.device)
        lo = torch.empty_like(hi)
        check(lib().muse_split_f32_to_bf16x2(t.data_ptr(), hi.data_ptr(), lo.data_ptr(), t.numel(), stream()), "muse_split_f32_to_bf16x2")
        return hi, lo
    hi = torch.empty_strided(t.shape, t.stride(), dtype=torch.bfloat16, device=t.device)
    lo = torch.empty_strided(t.shape, t.stride(), dtype=torch.bfloat16, device=t.device)
    hi.copy_(t)
    lo.copy_(t - hi.float())
    return hi, lo


X3_NATIVE = os.environ.get("MUSE_X3_NATIVE", "1") != "0"   # bf16x3 GEMM mode: the four-plane kernel (muse_gemm_x3) where it takes the product
X3_CAT = os.environ.get("MUSE_X3_CAT", "1") != "0"    # bf16x3 GEMM mode: one launch over a 3K-long concatenated operand pair (0: three launches)


def _split_planes_now(t):
    out = torch.empty((2,) + tuple(t.shape), dtype=torch.bfloat16, device=t.device)
    check(lib().muse_split_f32_to_bf16x2(t.data_ptr(), out[0].data_ptr(), out[1].data_ptr(), t.numel(), stream()), "muse_split_f32_to_bf16x2")
    return out


def split_planes(t):
    """[2, *t.shape] bf16: hi = bf16(t), lo = bf16(t - hi), through the running step's image cache when there is one"""
    if isinstance(t, Planes):
        return t.planes
    require_gpu(t)
    im = _X3_IMAGES[0]
    return _split_planes_now(t) if im is None else im.planes(t)


def split_cat3(t, layout, K, lo_pos, last_use=False):
    """_split_cat3_now through the running step's image cache (X3Images), when there is one"""
    im = _X3_IMAGES[0]
    if im is None:
        return _split_cat3_now(t, layout, K, lo_pos)
    return im.get(t, layout, K, lo_pos, last_use)


def _split_cat3_now(t, layout, K, lo_pos):
    """the bf16x3 operand of a 2-D contiguous f32 tensor for a product over its K dimension: layout 0 (k-contiguous [R, K]) ->
    ([R, 3K], ld 3K), layout 1 (k-major [K, R]) -> ([3K, R], ld R); thirds (hi | hi | lo) for lo_pos 2, (hi | lo | hi) for lo_pos 1"""
    require_gpu(t)
    rows, cols = t.shape
    if layout == 0:
        out = torch.empty((rows, 3 * cols), dtype=torch.bfloat16, device=t.device)
        ld_out = 3 * cols
    else:
        out = torch.empty((3 * rows, cols), dtype=torch.bfloat16, device=t.device)
        ld_out = cols
    check(lib().muse_split_f32_to_bf16_cat3(t.data_ptr(), out.data_ptr(), rows, cols, t.stride(0), ld_out, layout, lo_pos, stream()),
          "muse_split_f32_to_bf16_cat3")
    return out, ld_out


def _gemm_bf16x3(A, B, C_, M, N, K, la, lb, lda, ldb, ldc, a_off, b_off, c_off, alpha, bias, rowvec, residual, ldr, batch, zdiv,
                 sA, sB, sC, accumulate):
    """-> True when the product ran as three bf16 GEMMs accumulating into the f32 output, False when the bf16 kernels cannot take it"""
    d = GemmDesc()
    d.A, d.B, d.C = A.data_ptr() + a_off * 2, B.data_ptr() + b_off * 2, C_.data_ptr() + c_off * 4     # (alignment probe: bf16 offsets)
    d.dtype, d.out_dtype, d.layout_a, d.layout_b = BF16, F32, la, lb
    d.M, d.N, d.K, d.batch, d.zdiv = M, N, K, batch, zdiv
    d.lda, d.ldb, d.ldc, d.ldr = lda, ldb, ldc, ldr
    d.sA0, d.sA1 = sA
    d.sB0, d.sB1 = sB
    d.sC0, d.sC1 = sC
    d.alpha = alpha
    if batch > 1:
        # the batched products of the materialised attention core (per-head Q K^T, P V and their gradients: K = head_dim or the
        # sequence length, a few hundred at most) stay on the exact-f32 MFMA: they are ~4 % of a layer's flops, and three launches of
        # the 128^2 bf16 kernel plus the operand splits cost MORE than one exact product at these sizes (rocprof of the config-4
        # bf16x3 leg, profiles/r04_config4_bf16x3_kernel_stats_before.csv: 4452 + 4464 launches of ~45 us)
        return False
    if (a_off % 8) or (b_off % 8) or lib().muse_gemm_tile(C.byref(d)) < 0:
        return False
    full2d = (a_off == 0 and b_off == 0 and A.dim() == 2 and B.dim() == 2 and A.is_contiguous() and B.is_contiguous() and K % 8 == 0
              and A.stride(0) == lda and B.stride(0) == ldb and A.shape[1 if la == 0 else 0] == K and B.shape[1 if lb == 0 else 0] == K
              and A.shape[0 if la == 0 else 1] >= M and B.shape[0 if lb == 0 else 1] >= N and A.numel() * 2 < (1 << 31) and B.numel() * 2 < (1 << 31))
    if X3_NATIVE and full2d and lda % 8 == 0 and ldb % 8 == 0 and A.numel() % 8 == 0 and B.numel() % 8 == 0:
        # ONE kernel on the four planes (hi, lo of each operand; csrc/gemm256.h PipeX3): every plane goes through the LDS-DMA path once
        # per K-tile and feeds three MFMA products; one image per tensor serves every product that reads it
        with f32_gemms_as_bf16x3(False):
            a2, b2 = split_planes(A), split_planes(B)
            if gemm(a2[0], b2[0], C_, M, N, K, la=la, lb=lb, lda=lda, ldb=ldb, ldc=ldc, c_off=c_off, alpha=alpha, bias=bias, rowvec=rowvec,
                    residual=residual, ldr=ldr, accumulate=accumulate, x3_lo=(A.numel(), B.numel())) is not None:
                return True
    if isinstance(A, Planes) or isinstance(B, Planes):
        raise _hip.MuseHipError("a planes-only operand reached a product the four-plane kernel does not take")
    if (X3_CAT and a_off == 0 and b_off == 0 and A.dim() == 2 and B.dim() == 2 and A.is_contiguous() and B.is_contiguous() and K % 8 == 0
            and A.stride(0) == lda and B.stride(0) == ldb and A.shape[1 if la == 0 else 0] == K and B.shape[1 if lb == 0 else 0] == K
            and A.shape[0 if la == 0 else 1] >= M and B.shape[0 if lb == 0 else 1] >= N and 3 * A.numel() * 2 < (1 << 31)
            and 3 * B.numel() * 2 < (1 << 31)):
        # ONE launch: A' = (hi | hi | lo), B' = (hi | lo | hi) along a three times longer K (muse_split_f32_to_bf16_cat3) - one epilogue,
        # no read-modify-write of C between the terms, the persistent kernel where the plain product would take it
        # which operand carries (hi | hi | lo) and which (hi | lo | hi) is free as long as the two differ; chosen so that the images of
        # a Linear's forward (x W^T: lb = 0) and dX (dY W: lb = 1) products are the ones its dW product wants (X3Images)
        pa, pb = (2, 1) if lb == 0 else (1, 2)
        with f32_gemms_as_bf16x3(False):
            a3, lda3 = split_cat3(A, la, K, pa)
            b3, ldb3 = split_cat3(B, lb, K, pb)
            gemm(a3, b3, C_, M, N, 3 * K, la=la, lb=lb, lda=lda3, ldb=ldb3, ldc=ldc, c_off=c_off, alpha=alpha, bias=bias, rowvec=rowvec,
                 residual=residual, ldr=ldr, accumulate=accumulate)
        return True
    with f32_gemms_as_bf16x3(False):
        ah, al = split_f32(A)
        bh, bl = split_f32(B)
        kw = dict(la=la, lb=lb, lda=lda, ldb=ldb, ldc=ldc, a_off=a_off, b_off=b_off, c_off=c_off, alpha=alpha, batch=batch, zdiv=zdiv,
                  sA=sA, sB=sB, sC=sC)
        gemm(ah, bh, C_, M, N, K, bias=bias, rowvec=rowvec, residual=residual, ldr=ldr, accumulate=accumulate, **kw)
        gemm(ah, bl, C_, M, N, K, accumulate=True, **kw)
        gemm(al, bh, C_, M, N, K, accumulate=True, **kw)
    return True


def _gemm_via_transpose(A, B, C_, M, N, K, la, lb, lda, ldb, ldc, a_off, b_off, c_off, alpha, bias, rowvec, residual,
                        ldr, batch, zdiv, sA, sB, sC, accumulate, act):
    """Bring-up fallback: materialise k-contiguous copies of k-major operands with muse_transpose."""
    Kp = (K + 7) // 8 * 8

    def fix(X, layout, R, ld, off, s):
        if layout == 0:
            return X, ld, off, s
        # X(r,k) at off + k*ld + r  ->  T[z][r][k], zero padded to Kp
        T = torch.zeros((batch, R, Kp), dtype=X.dtype, device=X.device)
        for z in range(batch):  # batches are few in this debug path
            zo = off + (z // zdiv) * s[0] + (z % zdiv) * s[1]
            check(lib().muse_transpose(X.data_ptr() + zo * _esz(X), T[z].data_ptr(), dt(X), K, R, ld, Kp, 1, 0, 0,
                                       stream()), "muse_transpose")
        return T, Kp, 0, (zdiv * R * Kp, R * Kp)

    A2, lda2, a_off2, sA2 = fix(A, la, M, lda, a_off, sA)
    B2, ldb2, b_off2, sB2 = fix(B, lb, N, ldb, b_off, sB)
    return gemm(A2, B2, C_, M, N, K, la=0, lb=0, lda=lda2, ldb=ldb2, ldc=ldc, a_off=a_off2, b_off=b_off2, c_off=c_off,
                alpha=alpha, bias=bias, rowvec=rowvec, residual=residual, ldr=ldr, batch=batch, zdiv=zdiv, sA=sA2,
                sB=sB2, sC=sC, accumulate=accumulate, act=act)


def linear(x, w, out=None, *, out_dtype=None, residual=None, act=0, bias=None):
    """y[T,N] = x[T,K] @ w[N,K]^T (+bias) (gelu) (+residual)   — nn.Linear forward."""
    T_, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((T_, N), dtype=out_dtype or x.dtype, device=x.device)
    return gemm(x, w, out, T_, N, K, la=0, lb=0, lda=x.stride(0), ldb=w.stride(0), ldc=out.stride(0), residual=residual,
                ldr=residual.stride(0) if residual is not None else 0, act=act, bias=bias)


def linear_dgrad(dy, w, out=None):
    """dx[T,K] = dy[T,N] @ w[N,K]    (w consumed as a k-major operand)."""
    T_, N = dy.shape
    K = w.shape[1]
    if out is None:
        out = torch.empty((T_, K), dtype=dy.dtype, device=dy.device)
    return gemm(dy, w, out, T_, K, N, la=0, lb=1, lda=dy.stride(0), ldb=w.stride(0), ldc=out.stride(0))


SPLIT_K = os.environ.get("MUSE_SPLIT_K", "1") != "0"   # MUSE_SPLIT_K=0: deterministic (no f32 atomics) weight gradients


def wgrad_splits(M, N, K, dtype, slots=512, tile=128, ktile_us=1.8):
    """K slices for a weight-gradient GEMM.  The [N_out, K_in] output has few tiles while K = tokens is long, so the K
    loop is cut.  128^2 kernel (2 blocks per CU = 512 slots): the slice count that fills whole rounds of resident blocks
    best, with at least 4 K-tiles per slice.  256^2 LDS-DMA kernel (1 block per CU = 256 slots): the slice count with the
    smallest modelled time = rounds x (K-tiles per slice x 1.8 us + 6 us) + workspace traffic of the slice reduction
    ((slices + 1) x 4 M N bytes at ~4 TB/s)."""
    if not SPLIT_K:
        return 1
    tiles = ((M + tile - 1) // tile) * ((N + tile - 1) // tile)
    nk = (K + 63) // 64
    best, best_eff, best_cost = 1, 0.0, float("inf")
    for s in range(1, max(1, min(nk // 4, 64)) + 1):
        per = (nk + s - 1) // s
        s_eff = (nk + per - 1) // per          # every slice must own at least one K-tile
        blocks = tiles * s_eff
        if tile == 256:
            cost = ((blocks + slots - 1) // slots) * (per * ktile_us + 6.0) + ((s_eff + 1) * 4.0 * M * N / 4e6 if s_eff > 1 else 0.0)
            if cost < best_cost - 1e-9:
                best, best_cost = s_eff, cost
        else:
            eff = blocks / (((blocks + slots - 1) // slots) * slots)
            if eff > best_eff + 0.02:
                best, best_eff = s_eff, eff
    return best


_WGRAD_PLAN = {}


def _wgrad_plan(dy, x, N, K, T_, lda, ldb):
    """(split_k) for dw[N,K] = dy^T x: sized for the 256^2 kernel when muse_gemm would take it (muse_gemm_tile), else for
    the 128^2 one."""
    key = (N, K, T_, dy.dtype, lda, ldb)
    plan = _WGRAD_PLAN.get(key)
    if plan is None:
        plan = wgrad_splits(N, K, T_, dy.dtype, slots=512, tile=128)
        if dy.dtype in (torch.bfloat16, torch.float16):
            sk = wgrad_splits(N, K, T_, dy.dtype, slots=256, tile=256)
            d = GemmDesc()
            d.A, d.B, d.C = dy.data_ptr(), x.data_ptr(), dy.data_ptr()   # (pointers only checked for alignment)
            d.dtype, d.out_dtype, d.layout_a, d.layout_b = (F16 if dy.dtype == torch.float16 else BF16), F32, 1, 1
            d.M, d.N, d.K, d.batch, d.zdiv = N, K, T_, 1, 1
            d.lda, d.ldb, d.ldc = lda, ldb, K
            d.alpha = 1.0
            d.split_k, d.split_stride = sk, (N * K if sk > 1 else 0)
            if lib().muse_gemm_tile(C.byref(d)) == 256:
                plan = sk
        _WGRAD_PLAN[key] = plan
    return plan


_WGRAD_X3_PLAN = {}
X3_WGRAD_KTILE_US = float(os.environ.get("MUSE_X3_WGRAD_KTILE_US", "4.7"))   # cost-model time of a 64-wide K chunk of kernel_x3 (plain kernel: 1.8)


def _wgrad_x3_native(dy, x, dw, accumulate, M):
    """dw (+)= dy^T x as the four-plane bf16x3 product (muse_gemm_x3, both operands k-major) with the K split of the 256-tile kernel;
    False when that kernel does not take the shape"""
    T_, ncols = dy.shape
    N = M if M is not None else ncols
    K = x.shape[1]
    if N < 128 or K < 128 or T_ < 64 or dy.numel() % 8 or x.numel() % 8 or dy.numel() * 2 >= (1 << 31) or x.numel() * 2 >= (1 << 31):
        return False
    dy2, x2 = split_planes(dy), split_planes(x)
    lo = (dy.numel(), x.numel())
    splittable = dw.dtype == torch.float32 and dw.is_contiguous() and (N * K) % 4 == 0
    sk = 1
    if splittable:
        sk = _WGRAD_X3_PLAN.get((N, K, T_))
        if sk is None:      # a 64-wide K-tile of this kernel is three products: ~2.6 x the plain kernel's time per tile
            sk = _WGRAD_X3_PLAN[(N, K, T_)] = wgrad_splits(N, K, T_, torch.bfloat16, slots=256, tile=256, ktile_us=X3_WGRAD_KTILE_US)
    if sk <= 1:
        return gemm(dy2[0], x2[0], dw, N, K, T_, la=1, lb=1, lda=ncols, ldb=K, ldc=dw.stride(0), accumulate=accumulate, x3_lo=lo) is not None
    ws = torch.empty((sk, N, K), dtype=torch.float32, device=dw.device)
    if gemm(dy2[0], x2[0], ws, N, K, T_, la=1, lb=1, lda=ncols, ldb=K, ldc=K, split_k=sk, split_stride=N * K, x3_lo=lo) is None:
        return False
    check(lib().muse_sum_slices(ws.data_ptr(), dw.data_ptr(), sk, N * K, N * K, 1 if accumulate else 0, stream()), "muse_sum_slices")
    return True


def planes_only_ok(rows, cols):
    """may a producer hand its [rows, cols] result to the weight GEMMs as planes only?  (a bf16x3 step is running, the four-plane kernel
    is on and takes products with this operand: >= 128 rows / columns, whole 16-byte rows)"""
    on = ((_X3_IMAGES[0] is not None and _F32_AS_BF16X3[0] and X3_NATIVE and X3_PRODUCERS)
          or (_F16_IMAGES[0] is not None and _F32_AS_F16[0] and F16_PRODUCERS))       # ("f16" mode: the result as its half image only)
    return (on and X3_PLANES_ONLY and rows >= 128 and cols >= 128 and rows % 8 == 0 and cols % 8 == 0 and rows * cols * 2 < (1 << 31))


def linear_wgrad(dy, x, dw, accumulate, M=None, lda=None):
    """dw[N,K] (+)= dy[T,N]^T @ x[T,K]   (both operands k-major, f32 output into the flat grad buffer).
    Split-K slices write partial tiles to a workspace that muse_sum_slices folds into dw in a fixed order."""
    if _F32_AS_F16[0] and dw.dtype == torch.float32 and dw.is_contiguous() and dw.data_ptr() % 16 == 0:
        pair = f16_wgrad_operands(dy, x, M, lda)      # half operand images: dy is a gradient (the pass's scale), x a saved activation
        if pair is not None:
            return linear_wgrad(pair[0], pair[1], dw, accumulate, M=M, lda=lda)
    x3 = _F32_AS_BF16X3[0] and dy.dtype == torch.float32 and x.dtype == torch.float32 and dw.dtype == torch.float32 \
        and dy.stride(0) % 8 == 0 and x.stride(0) % 8 == 0 and x.shape[1] % 8 == 0 and (M if M is not None else dy.shape[1]) % 8 == 0
    if not x3 and (isinstance(dy, Planes) or isinstance(x, Planes)):
        raise _hip.MuseHipError("a planes-only operand reached a weight-gradient product outside the bf16x3 mode's preconditions")
    if x3:
        with f32_gemms_as_bf16x3(False):
            if (X3_NATIVE and dy.dim() == 2 and x.dim() == 2 and dy.is_contiguous() and x.is_contiguous() and (lda is None or lda == dy.stride(0))
                    and dy.shape[0] == x.shape[0] and _wgrad_x3_native(dy, x, dw, accumulate, M)):
                return dw
            if isinstance(dy, Planes) or isinstance(x, Planes):
                raise _hip.MuseHipError("a planes-only operand reached a weight-gradient product the four-plane kernel does not take")
            if (X3_CAT and dy.dim() == 2 and x.dim() == 2 and dy.is_contiguous() and x.is_contiguous() and (lda is None or lda == dy.stride(0))
                    and dy.shape[0] == x.shape[0] and 3 * dy.numel() * 2 < (1 << 32) - 64 and 3 * x.numel() * 2 < (1 << 32) - 64):
                # one product over 3 T slots: the ROW-concatenated images dY' [T, 3N] = (hi | lo | hi), X' [T, 3K] = (hi | hi | lo) - the ones
                # the dX product of dY and the forward product of x made (X3Images) - read as [3T, N] / [3T, K]: slot 3t + s holds third s
                # of token t in both, so the contraction over slots is sum_t hi hi + lo hi + hi lo
                T_ = dy.shape[0]
                dy3, _ = split_cat3(dy, 0, dy.shape[1], 1)
                x3, _ = split_cat3(x, 0, x.shape[1], 2, last_use=True)
                linear_wgrad(dy3.view(3 * T_, dy.shape[1]), x3.view(3 * T_, x.shape[1]), dw, accumulate, M=M, lda=lda)
            else:                             # three bf16 products through the bf16 split-K machinery, accumulated in a fixed order
                dyh, dyl = split_f32(dy)
                xh, xl = split_f32(x)
                linear_wgrad(dyh, xh, dw, accumulate, M=M, lda=lda)
                linear_wgrad(dyh, xl, dw, True, M=M, lda=lda)
                linear_wgrad(dyl, xh, dw, True, M=M, lda=lda)
        return dw
    T_, N = dy.shape
    if M is not None:
        N = M
    K = x.shape[1]
    lda = lda or dy.stride(0)
    splittable = dw.dtype == torch.float32 and dw.is_contiguous() and (N * K) % 4 == 0
    sk = _wgrad_plan(dy, x, N, K, T_, lda, x.stride(0)) if splittable else 1
    if sk <= 1:
        return gemm(dy, x, dw, N, K, T_, la=1, lb=1, lda=lda, ldb=x.stride(0), ldc=dw.stride(0), accumulate=accumulate)
    ws = torch.empty((sk, N, K), dtype=torch.float32, device=dw.device)
    gemm(dy, x, ws, N, K, T_, la=1, lb=1, lda=lda, ldb=x.stride(0), ldc=K, split_k=sk, split_stride=N * K)
    check(lib().muse_sum_slices(ws.data_ptr(), dw.data_ptr(), sk, N * K, N * K, 1 if accumulate else 0, stream()),
          "muse_sum_slices")
    return dw


# ---- grouped weight gradients -------------------------------------------------------------------------------------------------------
# MUSE_WGRAD_GROUP = K slices per product of a grouped dW launch when it runs BESIDE the backward chain on the weight-gradient stream
# (0: off - one split-K launch + slice sum per weight, the round-3 path).  Default 1: 144 blocks of 257 K-tiles at config B - no
# workspace, no slice sum, one f32 epilogue per tile; the launch fills 144 of 256 CUs and the concurrent dX / row kernels take the
# rest.  Same box, default bench loop (profiles/r04_wgrad_group_ab.txt): off 1155 images/s, 1 slice 1190, 2: 1185, 3: 1182, 5: 1172, 7: 1163.
# MUSE_WGRAD_GROUP_SERIAL = the slice count when nothing runs beside it (weight gradients on the main stream: wgrad_stream off, the
# instrumented serial step of bench.py): 5 slices = 720 blocks = 2.8 rounds of the chip (1114 TFLOP/s alone; 1 slice: 798).
WGRAD_GROUP = int(os.environ.get("MUSE_WGRAD_GROUP", "1"))
WGRAD_GROUP_SERIAL = int(os.environ.get("MUSE_WGRAD_GROUP_SERIAL", "5"))
WGRAD_SEQ_SLICES = int(os.environ.get("MUSE_WGRAD_SEQ_SLICES", "1"))   # (experiment: see linear_wgrad_group)


def sum_multi(jobs):
    """jobs: list of (kind, ws, out, nslices, n, stride, accumulate) with kind 0 = slice sum (muse_sum_slices), 1 = column sum
    (muse_colsum) - ONE launch, each job bit-identical to its single-job kernel"""
    for i in range(0, len(jobs), 16):
        part = jobs[i:i + 16]
        n = len(part)
        vp, i32, i64 = C.c_void_p * n, C.c_int32 * n, C.c_int64 * n
        check(lib().muse_sum_multi(vp(*[j[1].data_ptr() for j in part]), vp(*[j[2].data_ptr() for j in part]),
                                   i32(*[int(j[3]) for j in part]), i64(*[int(j[4]) for j in part]), i64(*[int(j[5]) for j in part]),
                                   i32(*[1 if j[6] else 0 for j in part]), i32(*[int(j[0]) for j in part]), n, stream()), "muse_sum_multi")


def linear_wgrad_group(items, colsums=None, split=None):
    """The weight gradients of one transformer layer in ONE launch + one reduction launch.
    items: list of (dy, x, dw, accumulate, M, lda) as linear_wgrad takes them (M / lda may be None); colsums: a queue of pending
    column sums (see _colsum_or_defer) folded into the same reduction launch.  Falls back to linear_wgrad per item whenever the
    256^2 kernel does not take every product (f32 mode, tiny shapes)."""
    split = WGRAD_GROUP if split is None else split
    descs, meta = [], []
    dt16 = items[0][0].dtype if items else None
    ok = split >= 1 and 1 <= len(items) <= 8 and dt16 in (torch.bfloat16, torch.float16) and all(it[0].dtype == dt16 and it[1].dtype == dt16 for it in items)
    if ok:
        arr = (GemmDesc * len(items))()
        for d, (dy, x, dw, accumulate, M, lda) in zip(arr, items):
            T_, N = dy.shape
            N = M if M is not None else N
            K = x.shape[1]
            ok = ok and dw.dtype == torch.float32 and dw.is_contiguous() and (N * K) % 4 == 0
            ok = ok and (split == 1 or dw.data_ptr() % 16 == 0)   # muse_sum_multi's slice sums store 16 bytes at a time
            d.A, d.B = dy.data_ptr(), x.data_ptr()
            d.dtype, d.out_dtype, d.layout_a, d.layout_b = (F16 if dt16 == torch.float16 else BF16), F32, 1, 1
            d.M, d.N, d.K, d.batch, d.zdiv = N, K, T_, 1, 1
            d.lda, d.ldb, d.ldc = (lda or dy.stride(0)), x.stride(0), K
            d.alpha = 1.0 / (getattr(dy, "_muse_scale", 1.0) * getattr(x, "_muse_scale", 1.0))      # (half images carry their power-of-two scale)
            meta.append((N, K, T_))
        if ok:
            ws = []
            for d, (dy, x, dw, accumulate, M, lda), (N, K, T_) in zip(arr, items, meta):
                if split > 1:
                    w = torch.empty((split, N, K), dtype=torch.float32, device=dw.device)
                    ws.append(w)
                    d.C, d.split_k, d.split_stride, d.accumulate = w.data_ptr(), split, N * K, 0
                else:
                    d.C, d.split_k, d.split_stride, d.accumulate = dw.data_ptr(), 1, 0, (1 if accumulate else 0)
            ok = lib().muse_gemm_group_ok(arr, len(items), split) == 0
    if not ok:
        for dy, x, dw, accumulate, M, lda in items:
            linear_wgrad(dy, x, dw, accumulate, M=M, lda=lda)
        if colsums:
            flush_colsums(colsums)
        return
    e0 = _prof_begin()
    seq = WGRAD_SEQ_SLICES if split == 1 else 1
    if seq > 1:
        # experiment (round 6, MUSE_WGRAD_SEQ_SLICES): the token dimension cut into `seq` launches that accumulate into dw one after the other -
        # no workspace, no reduction, the same tiles, but a workgroup of the dW stream holds its CU a `seq`-th as long (the step's
        # critical path is the main stream's latency: profiles/r06_ceiling.md)
        base = [(d.A, d.B, d.K, d.accumulate) for d in arr]
        for i in range(seq):
            for d, (a0, b0, T_, acc0) in zip(arr, base):
                nk = (T_ + 63) // 64
                k0 = (nk * i // seq) * 64
                k1 = min(T_, (nk * (i + 1) // seq) * 64) if i + 1 < seq else T_
                d.A, d.B, d.K = a0 + k0 * d.lda * 2, b0 + k0 * d.ldb * 2, k1 - k0
                d.accumulate = acc0 if i == 0 else 1
            check(lib().muse_gemm_group(arr, len(items), 1, stream()), "muse_gemm_group")
    else:
        check(lib().muse_gemm_group(arr, len(items), split, stream()), "muse_gemm_group")
    _prof_end(e0, "gemm_f16_TT" if dt16 == torch.float16 else "gemm_bf16_TT", sum(2.0 * N * K * T_ for N, K, T_ in meta))
    jobs = []
    if split > 1:
        # (the slice count the kernel really cuts: ceil(nk / ceil(nk / split)) slices are non-empty; the others leave their workspace
        #  slice untouched, so only the written ones are summed)
        for w, (dy, x, dw, accumulate, M, lda), (N, K, T_) in zip(ws, items, meta):
            nk = (T_ + 63) // 64
            per = (nk + split - 1) // split
            jobs.append((0, w, dw, (nk + per - 1) // per, N * K, N * K, accumulate))
    if colsums:
        cur = torch.cuda.current_stream(colsums[0][0].device)
        for part, dw, nblk, cols, accumulate in colsums:
            jobs.append((1, part, dw, nblk, cols, cols, accumulate))
            part.record_stream(cur)
        colsums.clear()
    if jobs:
        sum_multi(jobs)


def transpose(src, dst):
    """dst[c, r] = src[r, c] for contiguous 2-D tensors of the same dtype"""
    require_gpu(src, dst)
    r, c = src.shape
    check(lib().muse_transpose(src.data_ptr(), dst.data_ptr(), dt(src), r, c, src.stride(0), dst.stride(0), 1, 0, 0, stream()),
          "muse_transpose")
    return dst


def layernorm_fwd(x, w, eps, out_dtype, residual=None):
    require_gpu(x, w)
    rows, cols = x.shape
    y = torch.empty((rows, cols), dtype=out_dtype, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    e0 = _prof_begin()
    check(lib().muse_layernorm_fwd(x.data_ptr(), dt(x), w.data_ptr(), ptr(residual), y.data_ptr(), dt(y), mean.data_ptr(),
                                   rstd.data_ptr(), rows, cols, eps, stream()), "muse_layernorm_fwd")
    _prof_end(e0, "layernorm_fwd", _nbytes(x, residual, y), "byte")
    return y, mean, rstd


# The column sums that finish a LayerNorm weight gradient (per-block partials -> dw) are leaves of the backward graph.  A caller that
# runs weight gradients on a second stream passes `colsum_queue` (a list it owns): the sum is queued instead of launched, and
# flush_colsums(queue) launches the queue on whatever stream is current - the weight-gradient stream, off the dX chain.
def _colsum_or_defer(part, dw, nblk, cols, accumulate, queue):
    if queue is not None:
        queue.append((part, dw, nblk, cols, accumulate))
    else:
        check(lib().muse_colsum(part.data_ptr(), dw.data_ptr(), nblk, cols, 1 if accumulate else 0, stream()), "muse_colsum")


def flush_colsums(queue):
    """launch the queued column sums on the current stream (the caller has ordered it behind the kernels that wrote the partials)"""
    if not queue:
        return
    cur = torch.cuda.current_stream(queue[0][0].device)
    for part, dw, nblk, cols, accumulate in queue:
        check(lib().muse_colsum(part.data_ptr(), dw.data_ptr(), nblk, cols, 1 if accumulate else 0, stream()), "muse_colsum")
        part.record_stream(cur)
    queue.clear()


def layernorm_bwd(dy, x, w, mean, rstd, dx_dtype, dw, accumulate, dres=None, also_bf16=False, colsum_queue=None):
    """returns dx (= LN'(dy) + dres); dw (+)= column sums of dy * xhat.  also_bf16: returns (dx, bf16 copy of dx) written in the
    same pass."""
    require_gpu(dy, x, w)
    rows, cols = x.shape
    dx = torch.empty((rows, cols), dtype=dx_dtype, device=x.device)
    dx2 = torch.empty((rows, cols), dtype=torch.bfloat16, device=x.device) if also_bf16 else None
    nblk = lib().muse_layernorm_bwd_nblk(rows)
    part = torch.empty((nblk, cols), dtype=torch.float32, device=x.device)
    e0 = _prof_begin()
    check(lib().muse_layernorm_bwd(dy.data_ptr(), dt(dy), x.data_ptr(), dt(x), w.data_ptr(), mean.data_ptr(),
                                   rstd.data_ptr(), ptr(dres), dx.data_ptr(), dt(dx), ptr(dx2), part.data_ptr(), nblk, rows, cols,
                                   stream()), "muse_layernorm_bwd")
    _colsum_or_defer(part, dw, nblk, cols, accumulate, colsum_queue)
    _prof_end(e0, "layernorm_bwd", _nbytes(dy, x, dres, dx, dx2), "byte")
    return (dx, dx2) if also_bf16 else dx


# bit 0: fused forward pair (ln_fwd 1.74 -> 1.30 ms per step), bit 1: fused backward pair (measured SLOWER: 128 VGPRs / 4 waves per
# SIMD against 78 / 6 of ln_bwd_kernel - 2.96 -> 3.41 ms; kept for experiments, off by default)
LN_PAIR = int(os.environ.get("MUSE_LN_PAIR", "1"))


def layernorm_pair_ok(ao, x, cd, direction):
    return bool(LN_PAIR & direction) and cd == torch.bfloat16 and ao.dtype == torch.bfloat16 and x.dtype == torch.float32 and \
        x.shape[1] % 4 == 0 and x.shape[1] <= (1024 if direction == 1 else 768)


def layernorm_pair_fwd(ao, x, w_post, w_pre, eps):
    """x1 = x + LN(ao) * w_post ; ln2 = LN(x1) * w_pre  ->  (x1 f32, mean_post, rstd_post, ln2 bf16, mean_pre, rstd_pre)"""
    require_gpu(ao, x, w_post, w_pre)
    rows, cols = x.shape
    x1 = torch.empty_like(x)
    ln2 = torch.empty((rows, cols), dtype=torch.bfloat16, device=x.device)
    st = torch.empty((4, rows), dtype=torch.float32, device=x.device)
    e0 = _prof_begin()
    check(lib().muse_layernorm_pair_fwd(ao.data_ptr(), x.data_ptr(), w_post.data_ptr(), w_pre.data_ptr(), x1.data_ptr(), ln2.data_ptr(),
                                        st[0].data_ptr(), st[1].data_ptr(), st[2].data_ptr(), st[3].data_ptr(), rows, cols, eps, stream()),
          "muse_layernorm_pair_fwd")
    _prof_end(e0, "layernorm_fwd", _nbytes(ao, x, x1, ln2), "byte")
    return x1, st[0], st[1], ln2, st[2], st[3]


def layernorm_pair_bwd(dln2, x1, w_pre, mean_pre, rstd_pre, dres, ao, w_post, mean_post, rstd_post, dw_pre, acc_pre, dw_post, acc_post,
                       colsum_queue=None):
    """dx1 = LN_pre'(dln2) + dres ; dao = LN_post'(dx1)  ->  (dx1 f32, dao bf16); dw_pre / dw_post (+)= their weight gradients"""
    require_gpu(dln2, x1, ao)
    rows, cols = x1.shape
    dx1 = torch.empty_like(x1)
    dao = torch.empty((rows, cols), dtype=torch.bfloat16, device=x1.device)
    nblk = lib().muse_layernorm_bwd_nblk(rows)
    part = torch.empty((2, nblk, cols), dtype=torch.float32, device=x1.device)
    e0 = _prof_begin()
    check(lib().muse_layernorm_pair_bwd(dln2.data_ptr(), x1.data_ptr(), w_pre.data_ptr(), mean_pre.data_ptr(), rstd_pre.data_ptr(), ptr(dres),
                                        ao.data_ptr(), w_post.data_ptr(), mean_post.data_ptr(), rstd_post.data_ptr(), dx1.data_ptr(),
                                        dao.data_ptr(), part[0].data_ptr(), part[1].data_ptr(), nblk, rows, cols, stream()),
          "muse_layernorm_pair_bwd")
    _colsum_or_defer(part[0], dw_pre, nblk, cols, acc_pre, colsum_queue)
    _colsum_or_defer(part[1], dw_post, nblk, cols, acc_post, colsum_queue)
    _prof_end(e0, "layernorm_bwd", _nbytes(dln2, x1, dres, ao, dx1, dao), "byte")
    return dx1, dao


def softmax_(x, rows, cols, ld):
    require_gpu(x)
    check(lib().muse_softmax_fwd(x.data_ptr(), x.data_ptr(), dt(x), rows, cols, ld, stream()), "muse_softmax_fwd")
    return _touched(x)


def softmax_bwd_(p, dp, rows, cols, ld):
    """in place on dp: ds = p * (dp - sum(p*dp))"""
    require_gpu(p, dp)
    check(lib().muse_softmax_bwd(p.data_ptr(), dp.data_ptr(), dp.data_ptr(), dt(p), rows, cols, ld, stream()), "muse_softmax_bwd")
    return _touched(dp)


def attention_supported(dtype, seq, head_dim, seq_kv=None):
    """the fused attention kernels take bf16, head_dim 16/32/48/64 and any query / key length"""
    return dtype == torch.bfloat16 and seq > 0 and (seq_kv is None or seq_kv > 0) and head_dim in (16, 32, 48, 64)


def _row_view(t, heads_dim):
    """(base tensor pointer, elements between tokens) of a [tokens, >= heads_dim] view with unit column stride"""
    if t.dim() != 2 or t.stride(1) != 1 or t.shape[1] < heads_dim:
        raise _hip.MuseHipError("attention operands are [tokens, features] views with contiguous features")
    return t.data_ptr(), t.stride(0)


def _attn_desc(q, k, v, o, B, Sq, Skv, nh, hd, alpha):
    d = _hip.AttnDesc()
    H = nh * hd
    (d.q, d.ldq), (d.k, d.ldk), (d.v, d.ldv), (d.o, d.ldo) = _row_view(q, H), _row_view(k, H), _row_view(v, H), _row_view(o, H)
    d.bsq, d.bsk, d.bsv, d.bso = Sq * d.ldq, Skv * d.ldk, Skv * d.ldv, Sq * d.ldo
    d.batch, d.heads, d.head_dim, d.seq_q, d.seq_kv, d.alpha = B, nh, hd, Sq, Skv, alpha
    return d


def attention_fwd_ex(q, k, v, B, Sq, Skv, nh, hd, alpha, out=None):
    """fused softmax(alpha q k^T) v for separate q [B*Sq, H], k / v [B*Skv, H] views (row strides free: slices of a packed
    projection are fine); -> (ctx [B*Sq, H] bf16, lse [B*nh, seq_pad(Sq)] f32).  Self- and cross-attention."""
    require_gpu(q, k, v)
    H = nh * hd
    ctx = out if out is not None else torch.empty((B * Sq, H), dtype=q.dtype, device=q.device)
    lse = torch.empty((B * nh, lib().muse_attention_seq_pad(Sq)), dtype=torch.float32, device=q.device)
    d = _attn_desc(q, k, v, ctx, B, Sq, Skv, nh, hd, alpha)
    e0 = _prof_begin()
    check(lib().muse_attention_fwd_ex(C.byref(d), lse.data_ptr(), stream()), "muse_attention_fwd_ex")
    _prof_end(e0, "attn_fwd_bf16", 4.0 * B * nh * Sq * Skv * hd)
    return ctx, lse


def attention_bwd_ex(q, k, v, ctx, dctx, lse, B, Sq, Skv, nh, hd, alpha, dq=None, dk=None, dv=None):
    """-> (dq [B*Sq, H], dk, dv [B*Skv, H]) bf16; dq / dk / dv may be views (e.g. the three column blocks of a packed gradient)"""
    require_gpu(q, k, v, ctx, dctx, lse)
    H = nh * hd
    dq = dq if dq is not None else torch.empty((B * Sq, H), dtype=q.dtype, device=q.device)
    dk = dk if dk is not None else torch.empty((B * Skv, H), dtype=q.dtype, device=q.device)
    dv = dv if dv is not None else torch.empty((B * Skv, H), dtype=q.dtype, device=q.device)
    dsum = torch.empty_like(lse)
    d = _attn_desc(q, k, v, ctx, B, Sq, Skv, nh, hd, alpha)
    (pdo, lddo), (pdq, lddq), (pdk, lddk), (pdv, lddv) = _row_view(dctx, H), _row_view(dq, H), _row_view(dk, H), _row_view(dv, H)
    e0 = _prof_begin()
    check(lib().muse_attention_bwd_ex(C.byref(d), pdo, lddo, Sq * lddo, lse.data_ptr(), dsum.data_ptr(), pdq, lddq, Sq * lddq,
                                      pdk, lddk, Skv * lddk, pdv, lddv, Skv * lddv, stream()), "muse_attention_bwd_ex")
    _prof_end(e0, "attn_bwd_bf16", 10.0 * B * nh * Sq * Skv * hd)
    return dq, dk, dv


X3_STREAM = os.environ.get("MUSE_X3_STREAM", "1") != "0"   # sequences of several 256-key blocks: the streaming kernels (0: block pairs + merge)


def _x3_one_tile_keys(Skv):
    return 224 < Skv <= 256 or 64 < Skv <= 96


def attention_x3_blocked(Sq, Skv):
    """longer sequences (round 6: BASELINE config 4's 1024 tokens) run attention3.hip's one-tile kernels block by block: q rows in
    blocks of 256, keys in blocks of 256 (or one block of <= 96 text states), the key blocks' partial results merged by their log-sum-exps"""
    return Sq > 256 or Skv > 256


def attention_x3_streamed(Sq, Skv):
    """whole 256-row blocks on both sides and more than one key block: the streaming kernels (online-softmax forward, dQ and dK / dV
    passes; csrc/attention3.hip) - one writer per result, so operand images / planes-only gradients work as in the one-tile form"""
    return X3_STREAM and Sq % 256 == 0 and Skv % 256 == 0 and Skv > 256


def attention_x3_supported(Sq, Skv, hd):
    """shapes of the fused bf16x3 attention (csrc/attention3.hip): config 4's 16 x 16 grid against itself or its 77 text states in one tile
    per head, and whole multiples of 256 query rows / 256 keys block by block (attention_x3_blocked)"""
    return hd == 64 and Sq >= 256 and Sq % 256 == 0 and (_x3_one_tile_keys(Skv) or (Skv >= 256 and Skv % 256 == 0))


def _x3_block_desc(q, k, v, o, B, Sq, Skv, nh, hd, alpha, qi, kj, kb):
    """descriptor of query block qi (256 rows) against key block kj (kb keys) of every image: full-sequence batch strides, block pointers"""
    d = _attn_desc(q, k, v, o, B, Sq, Skv, nh, hd, alpha)
    d.q += qi * 256 * d.ldq * 4
    d.o += qi * 256 * d.ldo * 4
    d.k += kj * 256 * d.ldk * 4
    d.v += kj * 256 * d.ldv * 4
    d.seq_q, d.seq_kv = 256, kb
    return d


def _attention_x3_fwd_blocks(q, k, v, B, Sq, Skv, nh, hd, alpha):
    """-> (ctx [B*Sq, H] f32, lse [Sq // 256, B*nh, 256] f32).  softmax over all keys = the key blocks' softmaxes re-weighted by
    exp(lse_block - lse): exact in exact arithmetic, f32 here (the products inside each block are the kernels' bf16x3 ones)."""
    H = nh * hd
    nq = Sq // 256
    nk, kb = (Skv // 256, 256) if Skv > 256 else (1, Skv)
    dev = q.device
    ctx = torch.empty((B * Sq, H), dtype=torch.float32, device=dev)
    lse = torch.empty((nq, B * nh, 256), dtype=torch.float32, device=dev)
    e0 = _prof_begin()
    if nk > 1 and X3_STREAM:
        # whole 256-row blocks on both sides: a workgroup keeps its 256 queries and streams the key blocks through LDS with an online
        # softmax (csrc/attention3.hip fwd_stream_kernel) - q read once, the context and its log-sum-exp written once, no partials
        planes = x3_new_planes(ctx)
        d = _attn_desc(q, k, v, ctx, B, Sq, Skv, nh, hd, alpha)
        check(lib().muse_attention_x3_fwd_stream(C.byref(d), lse.data_ptr(), ptr(planes), ctx.numel() if planes is not None else 0, stream()),
              "muse_attention_x3_fwd_stream")
        x3_put_planes(ctx, planes)
    elif nk == 1:
        planes = x3_new_planes(ctx)          # ctx feeds the output projection: every block writes its rows of the operand planes too
        for qi in range(nq):
            d = _x3_block_desc(q, k, v, ctx, B, Sq, Skv, nh, hd, alpha, qi, 0, kb)
            pp = None if planes is None else planes.data_ptr() + qi * 256 * H * 2
            check(lib().muse_attention_x3_fwd(C.byref(d), lse[qi].data_ptr(), pp, ctx.numel() if planes is not None else 0, stream()), "muse_attention_x3_fwd")
        x3_put_planes(ctx, planes)
    else:
        part = torch.empty((nk, B * Sq, H), dtype=torch.float32, device=dev)
        lp = torch.empty((nk, nq, B * nh, 256), dtype=torch.float32, device=dev)
        for kj in range(nk):
            for qi in range(nq):
                d = _x3_block_desc(q, k, v, part[kj], B, Sq, Skv, nh, hd, alpha, qi, kj, kb)
                check(lib().muse_attention_x3_fwd(C.byref(d), lp[kj, qi].data_ptr(), None, 0, stream()), "muse_attention_x3_fwd")
        planes = x3_new_planes(ctx)          # ctx feeds the output projection: its operand planes come out of the merge kernel
        check(lib().muse_attention_x3_merge(part.data_ptr(), part[0].numel(), lp.data_ptr(), lp[0].numel(), nk, ctx.data_ptr(), lse.data_ptr(),
                                            ptr(planes), ctx.numel(), B, Sq, nh, stream()), "muse_attention_x3_merge")
        x3_put_planes(ctx, planes)
    _prof_end(e0, "attn_fwd_bf16x3", 4.0 * B * nh * Sq * Skv * hd)
    return ctx, lse


def _attention_x3_bwd_blocks(q, k, v, ctx, dctx, lse, B, Sq, Skv, nh, hd, alpha, dq, dk, dv):
    """block-pair backward: every (query block, key block) pair with the query block's GLOBAL log-sum-exp and the final context (so the
    probabilities and the row sums dO.O are the full-sequence ones); dq summed over key blocks, dk / dv over query blocks"""
    H = nh * hd
    nq = Sq // 256
    nk, kb = (Skv // 256, 256) if Skv > 256 else (1, Skv)
    dev = q.device
    pdo, lddo = _row_view(dctx, H)
    # partial gradients: dq of key block kj, dk / dv of query block qi - one contiguous stack each, folded by ONE row kernel into the
    # caller's (possibly strided) views; a single block writes the view directly
    dqs = None if nk == 1 else torch.empty((nk, B * Sq, H), dtype=torch.float32, device=dev)
    dks = None if nq == 1 else torch.empty((nq, B * Skv, H), dtype=torch.float32, device=dev)
    dvs = None if nq == 1 else torch.empty((nq, B * Skv, H), dtype=torch.float32, device=dev)
    dqp = [dq] if dqs is None else list(dqs)
    dkp = [dk] if dks is None else list(dks)
    dvp = [dv] if dvs is None else list(dvs)
    e0 = _prof_begin()
    for qi in range(nq):
        for kj in range(nk):
            d = _x3_block_desc(q, k, v, ctx, B, Sq, Skv, nh, hd, alpha, qi, kj, kb)
            (pq, ldq_), (pk, ldk_), (pv, ldv_) = _row_view(dqp[kj], H), _row_view(dkp[qi], H), _row_view(dvp[qi], H)
            check(lib().muse_attention_x3_bwd(C.byref(d), pdo + qi * 256 * lddo * 4, lddo, Sq * lddo, lse[qi].data_ptr(),
                                              pq + qi * 256 * ldq_ * 4, ldq_, Sq * ldq_, pk + kj * 256 * ldk_ * 4, ldk_, Skv * ldk_,
                                              pv + kj * 256 * ldv_ * 4, ldv_, Skv * ldv_, None, 0, None, 0, None, 0, stream()), "muse_attention_x3_bwd")
    for stack, g, rows in ((dqs, dq, B * Sq), (dks, dk, B * Skv), (dvs, dv, B * Skv)):
        if stack is not None:
            pg, ldg = _row_view(g, H)
            check(lib().muse_sum_parts_strided(stack.data_ptr(), stack[0].numel(), stack.shape[0], rows, H, pg, ldg, 0, stream()), "muse_sum_parts_strided")
    for t in (dq, dk, dv):
        _touched(t)            # (written through raw pointers: cached operand planes of their previous contents must not be found)
    _prof_end(e0, "attn_bwd_bf16x3", 10.0 * B * nh * Sq * Skv * hd)
    return dq, dk, dv


def attention_x3_fwd(q, k, v, B, Sq, Skv, nh, hd, alpha):
    """fused softmax(alpha q k^T) v on f32 tensors with bf16x3 products (TF32-class: <= 2^-16 relative per product); q [B*Sq, H],
    k / v [B*Skv, H] f32 views (slices of a packed projection are fine) -> (ctx [B*Sq, H] f32, lse [B*nh, 256] f32)"""
    require_gpu(q, k, v)
    if q.dtype != torch.float32 or k.dtype != torch.float32 or v.dtype != torch.float32:
        raise _hip.MuseHipError("attention_x3: f32 operands")
    if not attention_x3_supported(Sq, Skv, hd):
        raise _hip.MuseHipError(f"attention_x3: shape ({Sq} x {Skv}, head_dim {hd}) outside attention_x3_supported")
    if attention_x3_blocked(Sq, Skv):
        return _attention_x3_fwd_blocks(q, k, v, B, Sq, Skv, nh, hd, alpha)
    ctx = torch.empty((B * Sq, nh * hd), dtype=torch.float32, device=q.device)
    lse = torch.empty((B * nh, Sq), dtype=torch.float32, device=q.device)
    d = _attn_desc(q, k, v, ctx, B, Sq, Skv, nh, hd, alpha)
    planes = x3_new_planes(ctx)          # ctx feeds the output projection: its operand planes come out of the kernel
    e0 = _prof_begin()
    check(lib().muse_attention_x3_fwd(C.byref(d), lse.data_ptr(), ptr(planes), ctx.numel(), stream()), "muse_attention_x3_fwd")
    _prof_end(e0, "attn_fwd_bf16x3", 4.0 * B * nh * Sq * Skv * hd)
    x3_put_planes(ctx, planes)
    return ctx, lse


def attention_x3_bwd(q, k, v, ctx, dctx, lse, B, Sq, Skv, nh, hd, alpha, dq=None, dk=None, dv=None, planes=None, planes_only=False):
    """-> (dq [B*Sq, H], dk, dv [B*Skv, H]) f32; dq / dk / dv may be views (the column blocks of a packed gradient).
    planes = ((dq_hi, dq_lo_off), (dk_hi, ..), (dv_hi, ..)): hi-plane views shaped / strided like dq / dk / dv and the element distance
    to their lo planes (entries may be None) - the gradients also come out as operand planes (x3_new_planes of the packed tensor).
    planes_only: no f32 gradients at all (dq / dk / dv stay None; the caller wraps the plane tensors in ops.Planes)"""
    require_gpu(q, k, v, ctx, dctx, lse)
    H = nh * hd
    planes = planes or (None, None, None)
    if not planes_only:
        dq = dq if dq is not None else torch.empty((B * Sq, H), dtype=torch.float32, device=q.device)
        dk = dk if dk is not None else torch.empty((B * Skv, H), dtype=torch.float32, device=q.device)
        dv = dv if dv is not None else torch.empty((B * Skv, H), dtype=torch.float32, device=q.device)
    elif dq is not None or dk is not None or dv is not None or any(e is None or e[0] is None for e in planes):
        raise _hip.MuseHipError("attention_x3_bwd(planes_only=True): three plane views and no f32 gradient tensors")
    for t in (ctx, dctx, dq, dk, dv):
        if t is not None and t.dtype != torch.float32:
            raise _hip.MuseHipError("attention_x3: f32 operands")
    streamed = attention_x3_streamed(Sq, Skv)
    if attention_x3_blocked(Sq, Skv) and not streamed:
        if planes_only or any(e is not None and e[0] is not None for e in planes):
            raise _hip.MuseHipError("attention_x3_bwd: the block-by-block form (more than 256 query rows or keys) writes f32 gradients, no operand planes")
        return _attention_x3_bwd_blocks(q, k, v, ctx, dctx, lse, B, Sq, Skv, nh, hd, alpha, dq, dk, dv)
    d = _attn_desc(q, k, v, ctx, B, Sq, Skv, nh, hd, alpha)
    pdo, lddo = _row_view(dctx, H)
    e0 = _prof_begin()
    args = []
    pl = []
    for ent, g, rows in zip(planes, (dq, dk, dv), (Sq, Skv, Skv)):
        ref = g if g is not None else ent[0]               # (strides: the f32 view's, or - planes only - the hi-plane view's own)
        pg, ldg = _row_view(ref, H)
        args += [pg if g is not None else None, ldg, rows * ldg]
        if ent is None or ent[0] is None:
            pl += [None, 0]
        else:
            if ent[0].stride() != ref.stride() or ent[0].shape != ref.shape or ent[0].dtype != (torch.float16 if _F32_AS_F16[0] else torch.bfloat16):
                raise _hip.MuseHipError("attention_x3_bwd: a hi-plane view must mirror its gradient view")
            pl += [ent[0].data_ptr(), int(ent[1])]
    if streamed:
        # dQ per query block over the streamed key blocks (it also leaves dO . O per query in dsum), then dK / dV per key block over the
        # streamed query blocks: every gradient written once (bwd_dq_stream_kernel / bwd_dkv_stream_kernel)
        dsum = torch.empty_like(lse)
        check(lib().muse_attention_x3_bwd_stream(C.byref(d), pdo, lddo, Sq * lddo, lse.data_ptr(), dsum.data_ptr(), *args, *pl, stream()),
              "muse_attention_x3_bwd_stream")
        for t in (dq, dk, dv):
            if t is not None:
                _touched(t)
    else:
        check(lib().muse_attention_x3_bwd(C.byref(d), pdo, lddo, Sq * lddo, lse.data_ptr(), *args, *pl, stream()), "muse_attention_x3_bwd")
    _prof_end(e0, "attn_bwd_bf16x3", 10.0 * B * nh * Sq * Skv * hd)
    return dq, dk, dv


def attention_fwd(qkv, B, S, nh, hd, alpha):
    """fused softmax(alpha q k^T) v; qkv [B*S, 3H] bf16 -> (ctx [B*S, H] bf16, lse [B*nh, seq_pad] f32)"""
    require_gpu(qkv)
    H = nh * hd
    ctx = torch.empty((B * S, H), dtype=qkv.dtype, device=qkv.device)
    sp = lib().muse_attention_seq_pad(S)
    lse = torch.empty((B * nh, sp), dtype=torch.float32, device=qkv.device)
    e0 = _prof_begin()
    check(lib().muse_attention_fwd(qkv.data_ptr(), ctx.data_ptr(), lse.data_ptr(), B, S, nh, hd, alpha, stream()), "muse_attention_fwd")
    _prof_end(e0, "attn_fwd_bf16", 4.0 * B * nh * S * S * hd)
    return ctx, lse


def attention_bwd(qkv, ctx, dctx, lse, B, S, nh, hd, alpha):
    """-> dqkv [B*S, 3H] bf16"""
    require_gpu(qkv, ctx, dctx, lse)
    dqkv = torch.empty_like(qkv)
    dsum = torch.empty_like(lse)
    e0 = _prof_begin()
    check(lib().muse_attention_bwd(qkv.data_ptr(), ctx.data_ptr(), dctx.data_ptr(), lse.data_ptr(), dsum.data_ptr(),
                                   dqkv.data_ptr(), B, S, nh, hd, alpha, stream()), "muse_attention_bwd")
    _prof_end(e0, "attn_bwd_bf16", 10.0 * B * nh * S * S * hd)
    return dqkv


X3_PLANES_ONLY = os.environ.get("MUSE_X3_PLANES_ONLY", "1") != "0"   # ... and skip the f32 result where only weight GEMMs read it (ops.Planes)
X3_PRODUCERS = os.environ.get("MUSE_X3_PRODUCERS", "1") != "0"   # bf16x3 mode: kernels whose f32 result feeds a product write its operand planes too
F16_PRODUCERS = os.environ.get("MUSE_F16_PRODUCERS", "1") != "0"  # f16 mode: ... write its half image too (0: every operand through muse_cast_f32_to_f16)


def _x3_producing(t):
    """the running step's image cache when a producer of `t`-like f32 results should write operand planes next to them"""
    if _F32_AS_F16[0]:
        im = _F16_IMAGES[0]
        return im if (im is not None and F16_PRODUCERS and t.dtype == torch.float32 and t.is_contiguous()) else None
    im = _X3_IMAGES[0]
    return im if (im is not None and X3_NATIVE and X3_PRODUCERS and t.dtype == torch.float32 and t.is_contiguous()) else None


def x3_new_planes(t):
    """[2, *t.shape] bf16 for a producer that writes t's operand planes next to t - None when nothing would read them (no bf16x3 step
    running, the four-plane kernel off, or t no contiguous f32 tensor with whole 16-byte bf16 rows)"""
    im = _x3_producing(t)
    if im is None or t.dim() != 2 or t.shape[1] % 8:
        return None
    return planes_alloc(tuple(t.shape), t.device)


def planes_alloc(shape, device):
    """the operand-image tensor a producer kernel of the running step writes for a [rows, cols] result: [2, rows, cols] bf16 (hi, lo
    planes, "bf16x3" mode) or [1, rows, cols] IEEE half ("f16" mode: muse_operand_images tells the kernels which)"""
    if _F32_AS_F16[0]:
        return torch.empty((1,) + tuple(shape), dtype=torch.float16, device=device)
    return torch.empty((2,) + tuple(shape), dtype=torch.bfloat16, device=device)


def x3_put_planes(t, planes):
    if planes is not None:
        (_F16_IMAGES[0] if _F32_AS_F16[0] else _X3_IMAGES[0]).put_planes(t, planes)


def glu_fwd(ab, planes_only=False):
    """planes_only (a bf16x3 step, see planes_only_ok): -> ops.Planes, the result as GEMM operand planes without its f32 tensor"""
    require_gpu(ab)
    rows, two_i = ab.shape
    if planes_only:
        if not (planes_only_ok(rows, two_i // 2) and ab.dtype == torch.float32 and ab.is_contiguous()):
            raise _hip.MuseHipError("glu_fwd(planes_only=True) outside planes_only_ok")
        planes = planes_alloc((rows, two_i // 2), ab.device)
        check(lib().muse_glu_fwd_x3(ab.data_ptr(), None, planes.data_ptr(), rows, two_i // 2, stream()), "muse_glu_fwd_x3")
        return Planes(planes)
    h = torch.empty((rows, two_i // 2), dtype=ab.dtype, device=ab.device)
    im = _x3_producing(ab)
    if im is not None and (two_i // 2) % 8 == 0:
        planes = planes_alloc((rows, two_i // 2), ab.device)
        check(lib().muse_glu_fwd_x3(ab.data_ptr(), h.data_ptr(), planes.data_ptr(), rows, two_i // 2, stream()), "muse_glu_fwd_x3")
        im.put_planes(h, planes)
        return h
    check(lib().muse_glu_fwd(ab.data_ptr(), h.data_ptr(), dt(ab), rows, two_i // 2, stream()), "muse_glu_fwd")
    return h


def glu_bwd(ab, dh, planes_only=False):
    require_gpu(ab, dh)
    rows, two_i = ab.shape
    if planes_only:
        if not (planes_only_ok(rows, two_i) and ab.dtype == torch.float32 and dh.dtype == torch.float32 and ab.is_contiguous() and dh.is_contiguous()):
            raise _hip.MuseHipError("glu_bwd(planes_only=True) outside planes_only_ok")
        planes = planes_alloc((rows, two_i), ab.device)
        check(lib().muse_glu_bwd_x3(ab.data_ptr(), dh.data_ptr(), None, planes.data_ptr(), rows, two_i // 2, stream()), "muse_glu_bwd_x3")
        return Planes(planes)
    dab = torch.empty_like(ab)
    im = _x3_producing(ab)
    if im is not None and dh.dtype == torch.float32 and dh.is_contiguous() and two_i % 16 == 0:
        planes = planes_alloc((rows, two_i), ab.device)
        check(lib().muse_glu_bwd_x3(ab.data_ptr(), dh.data_ptr(), dab.data_ptr(), planes.data_ptr(), rows, two_i // 2, stream()), "muse_glu_bwd_x3")
        im.put_planes(dab, planes)
        return dab
    check(lib().muse_glu_bwd(ab.data_ptr(), dh.data_ptr(), dab.data_ptr(), dt(ab), rows, two_i // 2, stream()), "muse_glu_bwd")
    return dab


def ffn_mid_fwd(ab, w, eps, keep_h=True):
    """fused GLU + mid LayerNorm: ab [rows, 2I] -> (h, hm, mean, rstd).  keep_h=False: h = gelu(a) * b is not written (h is None);
    ffn_mid_bwd then recomputes it from ab, bit for bit."""
    require_gpu(ab, w)
    rows, two_i = ab.shape
    inter = two_i // 2
    h = torch.empty((rows, inter), dtype=ab.dtype, device=ab.device) if keep_h else None
    hm = torch.empty((rows, inter), dtype=ab.dtype, device=ab.device)
    mean = torch.empty(rows, dtype=torch.float32, device=ab.device)
    rstd = torch.empty_like(mean)
    e0 = _prof_begin()
    check(lib().muse_ffn_mid_fwd(ab.data_ptr(), w.data_ptr(), ptr(h), hm.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                 dt(ab), rows, inter, eps, stream()), "muse_ffn_mid_fwd")
    _prof_end(e0, "ffn_mid_fwd", _nbytes(ab, h, hm), "byte")
    return h, hm, mean, rstd


def ffn_mid_bwd(dhm, h, ab, w, mean, rstd, dw, accumulate, colsum_queue=None):
    """fused mid-LayerNorm backward + GLU backward: -> dab [rows, 2I]; dw (+)= column sums of dhm * xhat.  h may be None (see
    ffn_mid_fwd keep_h=False): it is then recomputed from ab."""
    require_gpu(dhm, ab, w)
    rows, inter = dhm.shape
    dab = torch.empty_like(ab)
    rpb = lib().muse_ffn_mid_rows_per_block()
    nblk = (rows + rpb - 1) // rpb
    part = torch.empty((nblk, inter), dtype=torch.float32, device=dhm.device)
    e0 = _prof_begin()
    check(lib().muse_ffn_mid_bwd(dhm.data_ptr(), ptr(h), ab.data_ptr(), w.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                 dab.data_ptr(), part.data_ptr(), dt(dhm), rows, inter, stream()), "muse_ffn_mid_bwd")
    _colsum_or_defer(part, dw, nblk, inter, accumulate, colsum_queue)
    _prof_end(e0, "ffn_mid_bwd", _nbytes(dhm, h, ab, dab), "byte")
    return dab


def gelu_fwd(x):
    require_gpu(x)
    y = torch.empty_like(x)
    check(lib().muse_gelu_fwd(x.data_ptr(), y.data_ptr(), dt(x), x.numel(), stream()), "muse_gelu_fwd")
    return y


def gelu_bwd(x, dy, out_dtype=None):
    """out_dtype=torch.bfloat16 with f32 x / dy: the result written directly as the bf16 operand of the next weight GEMMs"""
    require_gpu(x, dy)
    if out_dtype == torch.bfloat16 and x.dtype == torch.float32 and dy.dtype == torch.float32:
        dx = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
        check(lib().muse_gelu_bwd_f32_bf16(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), stream()), "muse_gelu_bwd_f32_bf16")
        return dx
    dx = torch.empty_like(x)
    check(lib().muse_gelu_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), dt(x), x.numel(), stream()), "muse_gelu_bwd")
    return dx


def embed_fwd(ids, word, pos):
    require_gpu(ids, word, pos)
    B, S = ids.shape
    V, H = word.shape
    out = torch.empty((B * S, H), dtype=torch.float32, device=ids.device)
    check(lib().muse_embed_fwd(ids.data_ptr(), word.data_ptr(), pos.data_ptr(), out.data_ptr(), B, S, H, V, stream()), "muse_embed_fwd")
    return out


EMBED_BWD_SORT = os.environ.get("MUSE_EMBED_BWD_SORT", "1") != "0"   # 0: the scan-per-row kernel (muse_embed_bwd)


def embed_bwd(ids, dout, dword, dpos, accumulate):
    """dword[v] (+)= sum of dout rows whose token id is v (position order), dpos[s] (+)= sum over the batch"""
    require_gpu(ids, dout, dword, dpos)
    B, S = ids.shape
    V, H = dword.shape
    if EMBED_BWD_SORT:
        nb = lib().muse_embed_bwd2_scratch_bytes(B, S, H, V)
        if nb < 0:
            raise _hip.MuseHipError("muse_embed_bwd2_scratch_bytes failed")
        scratch = torch.empty(nb, dtype=torch.uint8, device=ids.device)   # (the caching allocator hands out 512-byte aligned blocks)
        check(lib().muse_embed_bwd2(ids.data_ptr(), dout.data_ptr(), dword.data_ptr(), dpos.data_ptr(), scratch.data_ptr(), nb, B, S, H,
                                    V, 1 if accumulate else 0, stream()), "muse_embed_bwd2")
        return
    n = lib().muse_embed_bwd_scratch_floats(H, V)
    scratch = torch.empty(n, dtype=torch.float32, device=ids.device)
    check(lib().muse_embed_bwd(ids.data_ptr(), dout.data_ptr(), dword.data_ptr(), dpos.data_ptr(), scratch.data_ptr(), B, S, H,
                               V, 1 if accumulate else 0, stream()), "muse_embed_bwd")


def cross_entropy_fwd(logits, labels, label_smoothing, vocab=None, want_rows=False):
    """logits [rows, ld] with the first `vocab` columns valid; returns (loss_out[2] = (mean loss, n_valid), lse[rows])"""
    require_gpu(logits, labels)
    rows = logits.shape[0]
    V = vocab or logits.shape[1]
    row_loss = torch.empty(rows, dtype=torch.float32, device=logits.device)
    lse = torch.empty(rows, dtype=torch.float32, device=logits.device)
    loss_out = torch.empty(2, dtype=torch.float32, device=logits.device)
    check(lib().muse_cross_entropy_fwd(logits.data_ptr(), dt(logits), labels.data_ptr(), row_loss.data_ptr(), lse.data_ptr(),
                                       loss_out.data_ptr(), rows, V, logits.stride(0), label_smoothing, stream()),
          "muse_cross_entropy_fwd")
    if want_rows:
        return loss_out, lse, row_loss   # row_loss[r] = 0 for ignored rows (reduction="none" semantics)
    return loss_out, lse


def cross_entropy_bwd(logits, labels, lse, loss_out, grad_out, label_smoothing, out_dtype, vocab=None):
    """dlogits has the same (padded) row stride as logits; pad columns are written as zeros."""
    require_gpu(logits, labels, grad_out)
    rows = logits.shape[0]
    V = vocab or logits.shape[1]
    dl = torch.empty((rows, logits.shape[1]), dtype=out_dtype, device=logits.device)
    check(lib().muse_cross_entropy_bwd(logits.data_ptr(), dt(logits), labels.data_ptr(), lse.data_ptr(), loss_out.data_ptr(),
                                       grad_out.data_ptr(), dl.data_ptr(), dt(dl), rows, V, logits.stride(0), label_smoothing,
                                       stream()), "muse_cross_entropy_bwd")
    return dl


def adamw_flat(p, g, m, v, p_bf16, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    require_gpu(p, g, m, v)
    e0 = _prof_begin()
    check(lib().muse_adamw_flat(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), ptr(p_bf16), p.numel(), lr, beta1,
                                beta2, eps, weight_decay, step, grad_scale, stream()), "muse_adamw_flat")
    _prof_end(e0, "adamw", 28.0 * p.numel() + (2.0 * p.numel() if p_bf16 is not None else 0.0), "byte")


def adamw_multi(table, chunk_first, num_tensors, num_chunks, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    """one AdamW launch over a device-side table of tensors (muse_adamw_multi; training.FusedAdamW builds the table)"""
    require_gpu(table, chunk_first)
    check(lib().muse_adamw_multi(table.data_ptr(), chunk_first.data_ptr(), int(num_tensors), int(num_chunks), lr, beta1, beta2, eps,
                                 weight_decay, step, grad_scale, stream()), "muse_adamw_multi")


def ema_multi(table, chunk_first, num_tensors, num_chunks, one_minus_decay):
    """shadow -= (1 - decay) * (shadow - param) over a device-side table of tensors in one launch (muse_ema_multi; muse.EMAModel builds
    the table)"""
    require_gpu(table, chunk_first)
    check(lib().muse_ema_multi(table.data_ptr(), chunk_first.data_ptr(), int(num_tensors), int(num_chunks), float(one_minus_decay), stream()),
          "muse_ema_multi")


def _group_hyper(groups):
    """host array of {lr, beta1, beta2, eps, weight_decay} rows for muse_adamw_*_groups (kept alive by the caller for the call)"""
    import ctypes
    if not 1 <= len(groups) <= 8:
        raise _hip.MuseHipError(f"FusedAdamW: 1..8 parameter groups are supported, got {len(groups)}")
    flat = []
    for g in groups:
        flat += [float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"])]
    return (ctypes.c_float * len(flat))(*flat)


def adamw_flat_groups(p, g, m, v, p_bf16, base, seg_end, seg_group, groups, step, grad_scale=1.0):
    """AdamW on elements [base, base + p.numel()) of a flat buffer whose segments belong to different parameter groups
    (muse_adamw_flat_groups); p, g, m, v, p_bf16 are the slices starting at `base`; `groups`: torch param_groups-like dicts"""
    require_gpu(p, g, m, v, seg_end, seg_group)
    import ctypes
    hy = _group_hyper(groups)
    e0 = _prof_begin()
    check(lib().muse_adamw_flat_groups(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), ptr(p_bf16), p.numel(), int(base),
                                       seg_end.data_ptr(), seg_group.data_ptr(), int(seg_end.numel()),
                                       ctypes.cast(hy, ctypes.c_void_p), len(groups), int(step), float(grad_scale), stream()),
          "muse_adamw_flat_groups")
    _prof_end(e0, "adamw", 28.0 * p.numel() + (2.0 * p.numel() if p_bf16 is not None else 0.0), "byte")


def adamw_multi_groups(table, chunk_first, num_tensors, num_chunks, groups, step, grad_scale=1.0):
    """muse_adamw_multi with a group column in the table (7 x int64 per tensor)"""
    require_gpu(table, chunk_first)
    import ctypes
    hy = _group_hyper(groups)
    check(lib().muse_adamw_multi_groups(table.data_ptr(), chunk_first.data_ptr(), int(num_tensors), int(num_chunks),
                                        ctypes.cast(hy, ctypes.c_void_p), len(groups), int(step), float(grad_scale), stream()),
          "muse_adamw_multi_groups")


def cast_to_bf16(src, dst=None):
    require_gpu(src)
    if dst is None:
        dst = torch.empty(src.shape, dtype=torch.bfloat16, device=src.device)
    check(lib().muse_cast_f32_to_bf16(src.data_ptr(), dst.data_ptr(), src.numel(), stream()), "muse_cast_f32_to_bf16")
    return dst


def cast_to_f32(src, dst=None):
    require_gpu(src)
    given = dst is not None
    if dst is None:
        dst = torch.empty(src.shape, dtype=torch.float32, device=src.device)
    check(lib().muse_cast_bf16_to_f32(src.data_ptr(), dst.data_ptr(), src.numel(), stream()), "muse_cast_bf16_to_f32")
    return _touched(dst) if given else dst


def mask_sample(tokens, class_ids, timesteps, noise, mask_id, codebook_size, min_masking_rate=0.0):
    require_gpu(tokens, class_ids, timesteps, noise)
    B, S = tokens.shape
    input_ids = torch.empty((B, S + 1), dtype=torch.int64, device=tokens.device)
    labels = torch.empty((B, S + 1), dtype=torch.int64, device=tokens.device)
    mask_prob = torch.empty(B, dtype=torch.float32, device=tokens.device)
    check(lib().muse_mask_sample(tokens.data_ptr(), class_ids.data_ptr(), timesteps.data_ptr(), noise.data_ptr(),
                                 input_ids.data_ptr(), labels.data_ptr(), mask_prob.data_ptr(), B, S, mask_id, codebook_size,
                                 min_masking_rate, stream()), "muse_mask_sample")
    return input_ids, labels, mask_prob


def sample_step(cond_logits, input_ids, mask_id, vocab, temperature, sched_mask_len, *, uncond_logits=None, guidance_scale=0.0,
                noise_exp=None, noise_u=None, seed=0, step=0, want_raw=False):
    """one MaskGit decoding iteration (muse_sample_step): cond_logits [B, S, >= vocab] f32 view (last dim contiguous; a slice
    that skips the class-token row is fine) -> (sampled [B,S], next_input_ids [B,S], raw samples or None)"""
    require_gpu(cond_logits, input_ids)
    B, S = input_ids.shape

    def layout(t):
        if t.dim() != 3 or t.shape[0] != B or t.shape[1] != S or t.stride(2) != 1 or t.dtype != torch.float32 or t.shape[2] < vocab:
            raise _hip.MuseHipError("sample_step: logits must be a float32 [B, S, >= vocab] view with a contiguous last dim")
        return t.stride(0), t.stride(1)

    lay = layout(cond_logits)
    if uncond_logits is not None and layout(uncond_logits) != lay:
        raise _hip.MuseHipError("sample_step: cond / uncond logits must share a layout")
    ids = input_ids.contiguous()
    sampled, nxt = torch.empty_like(ids), torch.empty_like(ids)
    raw = torch.empty_like(ids) if want_raw else None
    conf = torch.empty(B * S, dtype=torch.float32, device=ids.device)
    if noise_exp is not None:
        noise_exp = noise_exp.reshape(B * S, vocab).to(device=ids.device, dtype=torch.float32).contiguous()
    if noise_u is not None:
        noise_u = noise_u.reshape(B * S).to(device=ids.device, dtype=torch.float32).contiguous()
    check(lib().muse_sample_step(cond_logits.data_ptr(), ptr(uncond_logits), float(guidance_scale), lay[0], lay[1], int(vocab),
                                 ids.data_ptr(), int(mask_id), ptr(noise_exp), ptr(noise_u), int(seed) & (2 ** 64 - 1), int(step),
                                 float(temperature), int(sched_mask_len), B, S, ptr(raw), sampled.data_ptr(), nxt.data_ptr(),
                                 conf.data_ptr(), stream()), "muse_sample_step")
    return sampled, nxt, raw


def mask_tokens(tokens, mask_id, *, timesteps=None, mask_prob=None, noise=None, rects=None, min_masking_rate=0.0, all_labels=False,
                want_weight=False, weight_min=0.3):
    """training/train_muse.py:149-226 on the device -> (input_ids, labels, loss_weight or None, mask_prob)"""
    require_gpu(tokens)
    B, S = tokens.shape
    dev = tokens.device
    tokens = tokens.contiguous()
    input_ids, labels = torch.empty_like(tokens), torch.empty_like(tokens)
    lw = torch.empty((B, S), dtype=torch.float32, device=dev) if want_weight else None
    mp = torch.empty(B, dtype=torch.float32, device=dev)
    f = lambda t: None if t is None else t.to(device=dev, dtype=torch.float32).contiguous()   # noqa: E731
    timesteps, mask_prob, noise = f(timesteps), f(mask_prob), f(noise)
    if rects is not None:
        rects = rects.to(device=dev, dtype=torch.int32).contiguous()
    check(lib().muse_mask_tokens(tokens.data_ptr(), ptr(timesteps), ptr(mask_prob), ptr(noise), ptr(rects), input_ids.data_ptr(),
                                 labels.data_ptr(), ptr(lw), mp.data_ptr(), B, S, int(mask_id), float(min_masking_rate),
                                 1 if all_labels else 0, float(weight_min), stream()), "muse_mask_tokens")
    return input_ids, labels, lw, mp


def dropout(x, p, seed, offset, out=None):
    """y = x * keep / (1 - p) with the Philox keep-mask of (seed, offset); the same call on a gradient is the backward.
    `out` may be x (in place)."""
    require_gpu(x)
    if not x.is_contiguous():
        raise _hip.MuseHipError("dropout expects a contiguous tensor")
    y = out if out is not None else torch.empty_like(x)
    check(lib().muse_dropout(x.data_ptr(), y.data_ptr(), dt(x), x.numel(), float(p), int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1),
                             stream()), "muse_dropout")
    return _touched(y) if out is not None else y


def cond_dropout(x, empty, uniforms, prob):
    """training/train_muse.py:715-731: x [B, ...] f32, empty [...] (one image's worth), uniforms [B]"""
    require_gpu(x, empty, uniforms)
    B = x.shape[0]
    xc = x.float().contiguous()
    per = xc.numel() // B
    e = empty.float().contiguous()
    if e.numel() != per:
        raise _hip.MuseHipError("cond_dropout: the empty embedding must have the shape of one batch element")
    out = torch.empty_like(xc)
    check(lib().muse_cond_dropout(xc.data_ptr(), e.data_ptr(), uniforms.float().contiguous().data_ptr(), out.data_ptr(), B, per,
                                  float(prob), stream()), "muse_cond_dropout")
    return out


# ---- VQGAN (NHWC) ----------------------------------------------------------------------------------------------
def conv2d_nhwc(x, w, B, H, W, Cin, Cout, KS, bias=None, residual=None, upsample=False):
    """x: [B, Hin, Win, Cin], w: [Cout, KS, KS, Cin]; returns [B, H, W, Cout].  upsample: False / 0 = stride 1 (Hin = H);
    True / 1 = nearest x2 upsample folded in (Hin = H/2);  2 = stride-2 3x3 over the bottom/right zero-padded input (Hin = 2H)."""
    require_gpu(x, w)
    out = torch.empty((B, H, W, Cout), dtype=x.dtype, device=x.device)
    e0 = _prof_begin()
    check(lib().muse_conv2d_nhwc(x.data_ptr(), w.data_ptr(), ptr(bias), ptr(residual), out.data_ptr(), dt(x), B, H, W, Cin,
                                 Cout, KS, int(upsample), stream()), "muse_conv2d_nhwc")
    _prof_end(e0, f"conv_{'bf16' if x.dtype == torch.bfloat16 else 'f32'}", 2.0 * B * H * W * Cout * KS * KS * Cin)
    return out


def split_bf16(w):
    """f32 -> (hi, lo) bf16 with w ~= hi + lo (weights of the bf16x3 convolution; one-time packing)"""
    hi = cast_to_bf16(w.contiguous())
    lo = cast_to_bf16((w - cast_to_f32(hi)).contiguous())
    return hi, lo


def conv2d_nhwc_split(x, w_hi, w_lo, B, H, W, Cin, Cout, KS, bias=None, residual=None, upsample=False, gn_groups=0):
    """f32 NHWC convolution as 3 bf16 MFMAs per product (see muse_conv2d_nhwc_split).  gn_groups > 0: the epilogue also
    leaves the GroupNorm statistics of the output on the returned tensor (`out._gn_stats = (partial, nchunk)`)."""
    require_gpu(x, w_hi, w_lo)
    out = torch.empty((B, H, W, Cout), dtype=torch.float32, device=x.device)
    part, nchunk = None, 0
    cpg = Cout // gn_groups if gn_groups else 0
    if gn_groups and Cout % gn_groups == 0 and cpg in (4, 8, 16, 32) and (H * W) % 128 == 0:
        nchunk = (H * W) // 128
        part = torch.empty(B * nchunk * gn_groups * 2, dtype=torch.float64, device=out.device)
    e0 = _prof_begin()
    check(lib().muse_conv2d_nhwc_split(x.data_ptr(), w_hi.data_ptr(), w_lo.data_ptr(), ptr(bias), ptr(residual), out.data_ptr(),
                                       ptr(part), gn_groups if part is not None else 0, B, H, W, Cin, Cout, KS,
                                       int(upsample), stream()), "muse_conv2d_nhwc_split")
    _prof_end(e0, "conv_bf16x3", 2.0 * B * H * W * Cout * KS * KS * Cin)
    if part is not None:
        out._gn_stats = (part, nchunk)
    return out


def conv2d_nhwc_split2(x_hi, x_lo, w_hi, w_lo, B, H, W, Cin, Cout, bias=None, residual=None, gn_groups=0):
    """3x3 bf16x3 convolution on pre-split activation planes, operands by LDS-DMA (see muse_conv2d_nhwc_split2).
    gn_groups > 0: the epilogue also produces the GroupNorm statistics of the output; they ride on the returned tensor as
    `out._gn_stats = (partial, nchunk)` for groupnorm_silu_nhwc_split."""
    require_gpu(x_hi, x_lo, w_hi, w_lo)
    out = torch.empty((B, H, W, Cout), dtype=torch.float32, device=x_hi.device)
    part, nchunk = None, 0
    if gn_groups and conv_gn_stats_ok(H, W, Cout, gn_groups):
        nchunk = (H * W) // 256
        part = torch.empty(B * nchunk * gn_groups * 2, dtype=torch.float64, device=out.device)
    e0 = _prof_begin()
    check(lib().muse_conv2d_nhwc_split2(x_hi.data_ptr(), x_lo.data_ptr(), w_hi.data_ptr(), w_lo.data_ptr(), ptr(bias),
                                        ptr(residual), out.data_ptr(), ptr(part), gn_groups if part is not None else 0,
                                        B, H, W, Cin, Cout, 3, stream()), "muse_conv2d_nhwc_split2")
    _prof_end(e0, "conv_bf16x3_dma", 2.0 * B * H * W * Cout * 9 * Cin)
    if e0 is not None:
        PROF_BYTES["conv_bf16x3_dma"] = PROF_BYTES.get("conv_bf16x3_dma", 0.0) + _nbytes(x_hi, x_lo, w_hi, w_lo, residual, out)
    if part is not None:
        out._gn_stats = (part, nchunk)
    return out


def conv_in_direct_ok(Cin, Cout, KS, Cpad, W=256):
    """shapes muse_conv_in_direct takes: the image-to-features 3x3 convolution of an encoder (image rows of up to 1022 pixels: three
    of them are staged in LDS)"""
    return (KS == 3 and Cin <= 4 and Cpad % 4 == 0 and Cpad >= 4 and Cout % 4 == 0 and Cout // 4 <= 256 and 256 % (Cout // 4) == 0
            and W <= 1022)


def conv_in_direct(x, w4, B, H, W, Cin, Cpad, Cout, bias=None, gn_groups=0):
    """conv_in as a direct exact-f32 convolution (muse_conv_in_direct): x [B, H, W, Cpad] f32, w4 [Cout, 9, 4] f32.
    gn_groups * 4 == Cout: the GroupNorm statistics of the output ride on it (`out._gn_stats = (partial, H)`)."""
    require_gpu(x, w4)
    out = torch.empty((B, H, W, Cout), dtype=torch.float32, device=x.device)
    part = None
    if gn_groups and gn_groups * 4 == Cout:
        part = torch.empty(B * H * gn_groups * 2, dtype=torch.float64, device=x.device)
    e0 = _prof_begin()
    check(lib().muse_conv_in_direct(x.data_ptr(), w4.data_ptr(), ptr(bias), out.data_ptr(), ptr(part), gn_groups if part is not None else 0,
                                    B, H, W, Cin, Cpad, Cout, stream()), "muse_conv_in_direct")
    _prof_end(e0, "conv_in_direct", _nbytes(x, out), "byte")
    if part is not None:
        out._gn_stats = (part, H)
    return out


def conv_out_direct_ok(H, W, Cin, Cout, KS):
    """shapes muse_conv_out_direct takes: the features-to-image 3x3 convolution of a decoder"""
    return KS == 3 and Cout <= 4 and Cin % 32 == 0 and Cin >= 32 and H % 16 == 0 and W % 16 == 0


def conv_out_direct(x, scale, shift, w, bias, B, H, W, Cin, Cout):
    """norm_out -> swish -> conv_out as one direct exact-f32 convolution (muse_conv_out_direct): x [B, H, W, Cin] f32, scale / shift
    [B, Cin] (groupnorm_scale_shift), w [Cout, 9, Cin] f32 -> [B, H, W, Cout] f32"""
    require_gpu(x, scale, shift, w)
    out = torch.empty((B, H, W, Cout), dtype=torch.float32, device=x.device)
    e0 = _prof_begin()
    check(lib().muse_conv_out_direct(x.data_ptr(), scale.data_ptr(), shift.data_ptr(), w.data_ptr(), ptr(bias), out.data_ptr(),
                                     B, H, W, Cin, Cout, stream()), "muse_conv_out_direct")
    _prof_end(e0, "conv_out_direct", _nbytes(x, out), "byte")
    return out


def upsample2x(x, B, H, W, C):
    """nearest x2 of an NHWC tensor [B, H, W, C] (f32 / bf16) -> [B, 2H, 2W, C] (muse_upsample2x_nhwc)"""
    require_gpu(x)
    y = torch.empty((B, 2 * H, 2 * W, C), dtype=x.dtype, device=x.device)
    check(lib().muse_upsample2x_nhwc(x.data_ptr(), y.data_ptr(), dt(x), B, H, W, C, stream()), "muse_upsample2x_nhwc")
    return y


def upsample2x_split(x, B, H, W, C):
    """nearest x2 of x [B, H, W, C] f32 as the (hi, lo) bf16 operand planes [B, 2H, 2W, C] of conv2d_nhwc_split2"""
    require_gpu(x)
    hi = torch.empty((B, 2 * H, 2 * W, C), dtype=torch.bfloat16, device=x.device)
    lo = torch.empty_like(hi)
    e0 = _prof_begin()
    check(lib().muse_upsample2x_split_nhwc(x.data_ptr(), hi.data_ptr(), lo.data_ptr(), B, H, W, C, stream()), "muse_upsample2x_split_nhwc")
    _prof_end(e0, "upsample2x_split", _nbytes(x, hi, lo), "byte")
    return hi, lo


class conv_persistent:
    """with ops.conv_persistent(False): ... - the fused convolutions launched inside run on the launch-per-tile kernel (muse_conv_persistent):
    what a tokenizer pass enqueued BESIDE a train step wants (muse.TrainStep); restored on exit"""

    def __init__(self, on):
        self.on = 1 if on else 0

    def __enter__(self):
        self.prev = lib().muse_conv_persistent(-1)
        lib().muse_conv_persistent(self.on)
        return self

    def __exit__(self, *exc):
        lib().muse_conv_persistent(self.prev)
        return False


def conv_gn_split2_ok(B, H, W, Cin, Cout, KS):
    """shapes muse_conv2d_nhwc_gn_split2 takes (GroupNorm + SiLU + split fused into the patch-slab convolution)"""
    return bool(lib().muse_conv2d_nhwc_gn_split2_ok(B, H, W, Cin, Cout, KS))


def groupnorm_scale_shift(stats, gamma, beta, B, HW, C, groups=32, eps=1e-6):
    """(scale, shift) [B, C] f32 with GroupNorm(x)[b, :, c] = x * scale[b, c] + shift[b, c], from a producer's partial sums
    `stats` = (partial, nchunk)"""
    part, nchunk = stats
    require_gpu(part, gamma, beta)
    sc = torch.empty((B, C), dtype=torch.float32, device=part.device)
    sh = torch.empty((B, C), dtype=torch.float32, device=part.device)
    check(lib().muse_groupnorm_scale_shift(part.data_ptr(), nchunk, gamma.data_ptr(), beta.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                                           B, HW, C, groups, eps, stream()), "muse_groupnorm_scale_shift")
    return sc, sh


def conv2d_nhwc_gn_split2(x, scale, shift, w_hi, w_lo, B, H, W, Cin, Cout, bias=None, residual=None, gn_groups=0):
    """conv3x3(silu(GroupNorm(x))) in one kernel: x f32 NHWC, scale / shift from groupnorm_scale_shift (see
    muse_conv2d_nhwc_gn_split2); otherwise as conv2d_nhwc_split2"""
    require_gpu(x, scale, shift, w_hi, w_lo)
    out = torch.empty((B, H, W, Cout), dtype=torch.float32, device=x.device)
    part, nchunk = None, 0
    if gn_groups and conv_gn_stats_ok(H, W, Cout, gn_groups):
        nchunk = (H * W) // 256
        part = torch.empty(B * nchunk * gn_groups * 2, dtype=torch.float64, device=out.device)
    e0 = _prof_begin()
    check(lib().muse_conv2d_nhwc_gn_split2(x.data_ptr(), scale.data_ptr(), shift.data_ptr(), w_hi.data_ptr(), w_lo.data_ptr(), ptr(bias),
                                           ptr(residual), out.data_ptr(), ptr(part), gn_groups if part is not None else 0,
                                           B, H, W, Cin, Cout, 3, stream()), "muse_conv2d_nhwc_gn_split2")
    _prof_end(e0, "conv_bf16x3_dma", 2.0 * B * H * W * Cout * 9 * Cin)
    if e0 is not None:
        PROF_BYTES["conv_bf16x3_dma"] = PROF_BYTES.get("conv_bf16x3_dma", 0.0) + _nbytes(x, w_hi, w_lo, residual, out)
    if part is not None:
        out._gn_stats = (part, nchunk)
    return out


def conv_split2_ok(B, H, W, Cin, Cout, KS):
    """shapes muse_conv2d_nhwc_split2 takes (the rest stay on muse_conv2d_nhwc_split)"""
    M = B * H * W
    return KS == 3 and Cin % 32 == 0 and Cout % 4 == 0 and M * Cin * 2 < (1 << 32) - 64 and M < (1 << 31) - 256


def conv_slab_ok(H, W, Cin):
    """shapes muse_conv2d_nhwc_split2 runs on its patch-slab kernel (16 x 16 pixel patches, K order chunk / tap / channel) unless
    MUSE_CONV_SLAB=0; the others run on the tap-major kernel"""
    import os
    return H % 16 == 0 and W % 16 == 0 and Cin % 64 == 0 and os.environ.get("MUSE_CONV_SLAB", "1") != "0"


def conv_gn_stats_ok(H, W, Cout, groups):
    cpg = Cout // groups if groups else 0
    return groups in (32, 64) and Cout % groups == 0 and (H * W) % 256 == 0 and 4 <= cpg <= 128 and (cpg & (cpg - 1)) == 0


def groupnorm_silu_nhwc_split(x, gamma, beta, B, HW, C, groups=32, eps=1e-6, silu=True, stats=None):
    """GroupNorm + SiLU of an f32 NHWC tensor, returned as the (hi, lo) bf16 planes of the result.
    stats = (partial, nchunk) from the producing convolution's epilogue skips the statistics pass."""
    require_gpu(x, gamma, beta)
    hi = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    lo = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    if stats is not None:
        part, snc = stats
    else:
        nchunk = lib().muse_groupnorm_nchunk(HW)
        part, snc = torch.empty(B * nchunk * groups * 2, dtype=torch.float64, device=x.device), 0
    e0 = _prof_begin()
    check(lib().muse_groupnorm_silu_nhwc_split(x.data_ptr(), hi.data_ptr(), lo.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                               part.data_ptr(), snc, B, HW, C, groups, eps, 1 if silu else 0, stream()),
          "muse_groupnorm_silu_nhwc_split")
    _prof_end(e0, "groupnorm_silu", _nbytes(x, hi, lo), "byte")
    return hi, lo


def groupnorm_silu_nhwc(x, gamma, beta, B, HW, C, groups=32, eps=1e-6, silu=True):
    require_gpu(x, gamma, beta)
    y = torch.empty_like(x)
    nchunk = lib().muse_groupnorm_nchunk(HW)
    part = torch.empty(B * nchunk * groups * 2, dtype=torch.float64, device=x.device)
    e0 = _prof_begin()
    check(lib().muse_groupnorm_silu_nhwc(x.data_ptr(), y.data_ptr(), dt(x), gamma.data_ptr(), beta.data_ptr(), part.data_ptr(),
                                         B, HW, C, groups, eps, 1 if silu else 0, stream()), "muse_groupnorm_silu_nhwc")
    _prof_end(e0, "groupnorm_silu", _nbytes(x, y), "byte")
    return y


def avgpool2x2_nhwc(x, B, H, W, C_, gn_groups=0):
    """F.avg_pool2d(2, 2) on NHWC.  gn_groups > 0 (f32): the same pass also leaves the GroupNorm statistics of the pooled
    tensor on it (`y._gn_stats`), so the next level's first norm does not re-read it."""
    require_gpu(x)
    y = torch.empty((B, H // 2, W // 2, C_), dtype=x.dtype, device=x.device)
    e0 = _prof_begin()
    vpp = C_ // 4
    if gn_groups and x.dtype == torch.float32 and C_ % 4 == 0 and C_ % gn_groups == 0 and vpp <= 256 and 256 % vpp == 0:
        nchunk = lib().muse_groupnorm_nchunk((H // 2) * (W // 2))
        part = torch.empty(B * nchunk * gn_groups * 2, dtype=torch.float64, device=x.device)
        check(lib().muse_avgpool2x2_nhwc_stats(x.data_ptr(), y.data_ptr(), part.data_ptr(), gn_groups, B, H, W, C_, stream()),
              "muse_avgpool2x2_nhwc_stats")
        y._gn_stats = (part, nchunk)
    else:
        check(lib().muse_avgpool2x2_nhwc(x.data_ptr(), y.data_ptr(), dt(x), B, H, W, C_, stream()), "muse_avgpool2x2_nhwc")
    _prof_end(e0, "avgpool2x2", _nbytes(x, y), "byte")
    return y


def nchw_to_nhwc(x, out_dtype, cpad):
    require_gpu(x)
    B, C_, H, W = x.shape
    x = x.contiguous()
    out = torch.empty((B, H, W, cpad), dtype=out_dtype, device=x.device)
    check(lib().muse_nchw_to_nhwc(x.data_ptr(), out.data_ptr(), dt(out), B, C_, H * W, cpad, stream()), "muse_nchw_to_nhwc")
    return out


def nhwc_to_nchw(x, C_):
    require_gpu(x)
    B, H, W, cpad = x.shape
    out = torch.empty((B, C_, H, W), dtype=torch.float32, device=x.device)
    check(lib().muse_nhwc_to_nchw(x.data_ptr(), dt(x), out.data_ptr(), B, C_, H * W, cpad, stream()), "muse_nhwc_to_nchw")
    return out


def vq_nearest(z_flat, codebook, en=None, return_dist=False):
    """argmin_j |z - e_j|^2 computed as the reference does: addmm(|z|^2 + |e|^2, z, e^T, alpha=-2) then argmin.

    z_flat [N, D] f32, codebook [Kc, D] f32 -> int64 [N] (with return_dist also the f32 distance rows [N, Kc] the argmin ran over:
    the parity tests' near-tie accounting reads them)."""
    require_gpu(z_flat, codebook)
    N, D = z_flat.shape
    Kc = codebook.shape[0]
    zn = torch.empty(N, dtype=torch.float32, device=z_flat.device)
    check(lib().muse_row_sumsq(z_flat.data_ptr(), zn.data_ptr(), N, D, z_flat.stride(0), stream()), "muse_row_sumsq")
    if en is None:
        en = torch.empty(Kc, dtype=torch.float32, device=z_flat.device)
        check(lib().muse_row_sumsq(codebook.data_ptr(), en.data_ptr(), Kc, D, codebook.stride(0), stream()), "muse_row_sumsq")
    dist = torch.empty((N, Kc), dtype=torch.float32, device=z_flat.device)
    gemm(z_flat, codebook, dist, N, Kc, D, la=0, lb=0, lda=z_flat.stride(0), ldb=codebook.stride(0), ldc=Kc, alpha=-2.0,
         bias=en, rowvec=zn)
    idx = torch.empty(N, dtype=torch.int64, device=z_flat.device)
    check(lib().muse_argmin_rows(dist.data_ptr(), idx.data_ptr(), N, Kc, Kc, stream()), "muse_argmin_rows")
    return (idx, dist) if return_dist else idx


def vq_neg_distances_scaled(z_flat, codebook, inv_temp):
    """-(|z|^2 + |e|^2 - 2 z.e) * inv_temp as one GEMM with its row / column vectors pre-scaled: the logits of
    VectorQuantizer.get_soft_code (muse/modeling_maskgit_vqgan.py:327-331) -> f32 [N, Kc]"""
    require_gpu(z_flat, codebook)
    N, D = z_flat.shape
    Kc = codebook.shape[0]
    zn = torch.empty(N, dtype=torch.float32, device=z_flat.device)
    en = torch.empty(Kc, dtype=torch.float32, device=z_flat.device)
    check(lib().muse_row_sumsq(z_flat.data_ptr(), zn.data_ptr(), N, D, z_flat.stride(0), stream()), "muse_row_sumsq")
    check(lib().muse_row_sumsq(codebook.data_ptr(), en.data_ptr(), Kc, D, codebook.stride(0), stream()), "muse_row_sumsq")
    out = torch.empty((N, Kc), dtype=torch.float32, device=z_flat.device)
    gemm(z_flat, codebook, out, N, Kc, D, la=0, lb=0, lda=z_flat.stride(0), ldb=codebook.stride(0), ldc=Kc, alpha=2.0 * inv_temp,
         bias=en.mul_(-inv_temp), rowvec=zn.mul_(-inv_temp))
    return out


def gather_rows(table, idx, out_dtype):
    require_gpu(table, idx)
    rows = idx.numel()
    cols = table.shape[1]
    out = torch.empty((rows, cols), dtype=out_dtype, device=table.device)
    check(lib().muse_gather_rows(table.data_ptr(), idx.data_ptr(), out.data_ptr(), dt(out), rows, cols, stream()), "muse_gather_rows")
    return out


# ---- MaskGiTUViT_v2 : f32 forward ops --------------------------------------------------------
def norm_res_fwd(x, w, eps, mode, residual=None, want_pre=False):
    """v = x (+ residual); y = RMSNorm(v) * w (mode 0) / LayerNorm(v) * w (mode 1); returns (y, v or None)"""
    require_gpu(x)
    rows, cols = x.shape
    y = torch.empty_like(x)
    pre = torch.empty_like(x) if want_pre else None
    check(lib().muse_norm_res_fwd(x.data_ptr(), ptr(residual), ptr(w), y.data_ptr(), ptr(pre), rows, cols, eps, mode, stream()),
          "muse_norm_res_fwd")
    return y, pre


def adaln_fwd(x, ss, batch, out_dtype=torch.float32):
    """AdaLN modulation; out_dtype=torch.bfloat16 writes the result directly as the bf16 GEMM operand (no f32 copy)"""
    require_gpu(x, ss)
    rows, C_ = x.shape
    y = torch.empty((rows, C_), dtype=out_dtype, device=x.device)
    f32 = out_dtype == torch.float32
    check(lib().muse_adaln_fwd_ex(x.data_ptr(), ss.data_ptr(), y.data_ptr() if f32 else None, None if f32 else y.data_ptr(), batch,
                                  rows // batch, C_, stream()), "muse_adaln_fwd_ex")
    return y


def silu_fwd(x):
    require_gpu(x)
    y = torch.empty_like(x)
    check(lib().muse_silu_fwd(x.data_ptr(), y.data_ptr(), x.numel(), stream()), "muse_silu_fwd")
    return y


def dwconv3x3_nhwc(x, w, B, H, W, C_):
    """x [B*H*W, C] (NHWC rows), w [C, 1, 3, 3] contiguous"""
    require_gpu(x, w)
    y = torch.empty_like(x)
    check(lib().muse_dwconv3x3_nhwc(x.data_ptr(), w.data_ptr(), y.data_ptr(), B, H, W, C_, stream()), "muse_dwconv3x3_nhwc")
    return y


def space_to_depth2(x, B, H, W, C_):
    """x [B*H*W, C] f32 (NHWC rows) -> [B*(H/2)*(W/2), 4C], channel order (di, dj, c): the operand of the stride-2 2x2 convolution as
    a product (DownsampleBlock, reference modeling_transformer_v2.py:510-514); also the backward of depth_to_space2"""
    require_gpu(x)
    y = torch.empty((B * (H // 2) * (W // 2), 4 * C_), dtype=torch.float32, device=x.device)
    check(lib().muse_space_to_depth2_nhwc(x.data_ptr(), y.data_ptr(), B, H, W, C_, 0, stream()), "muse_space_to_depth2_nhwc")
    return y


def depth_to_space2(x, B, H, W, C_):
    """x [B*(H/2)*(W/2), 4C] f32, channel order (di, dj, c) -> [B*H*W, C]: the stride-2 2x2 transposed convolution's output placement
    (UpsampleBlock, reference :558-562); H, W are the full-resolution sides"""
    require_gpu(x)
    y = torch.empty((B * H * W, C_), dtype=torch.float32, device=x.device)
    check(lib().muse_space_to_depth2_nhwc(x.data_ptr(), y.data_ptr(), B, H, W, C_, 1, stream()), "muse_space_to_depth2_nhwc")
    return y


def grn_fwd(x, gamma, beta, B, S, want_stats=False, out_dtype=torch.float32):
    """GlobalResponseNorm over the S pixels of each image; x [B*S, C] f32.  out_dtype=torch.bfloat16: the result is written
    only as the bf16 GEMM operand of the bf16 compute mode (no f32 copy, no cast pass)"""
    require_gpu(x, gamma, beta)
    C_ = x.shape[1]
    bf = out_dtype == torch.bfloat16 and C_ % 4 == 0
    y = torch.empty(x.shape, dtype=torch.bfloat16 if bf else torch.float32, device=x.device)
    stats = torch.empty(2 * B * C_, dtype=torch.float32, device=x.device)   # [G | N], kept for grn_bwd
    check(lib().muse_grn_fwd_ex(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None if bf else y.data_ptr(), y.data_ptr() if bf else None,
                                stats.data_ptr(), B, S, C_, stream()), "muse_grn_fwd_ex")
    return (y, stats) if want_stats else y


def sinusoidal_encode(features, dim, max_positions=10000.0):
    require_gpu(features)
    f = features.reshape(-1).float().contiguous()
    out = torch.empty((f.numel(), dim), dtype=torch.float32, device=f.device)
    check(lib().muse_sinusoidal_encode(f.data_ptr(), out.data_ptr(), f.numel(), dim, max_positions, stream()), "muse_sinusoidal_encode")
    return out


def weighted_mean(v, w):
    require_gpu(v, w)
    out = torch.empty(1, dtype=torch.float32, device=v.device)
    check(lib().muse_weighted_mean(v.data_ptr(), w.data_ptr(), out.data_ptr(), v.numel(), stream()), "muse_weighted_mean")
    return out


def colsum(part, out, accumulate=False):
    """out[c] (+)= sum_r part[r, c]  (fixed order)"""
    require_gpu(part, out)
    rows, cols = part.shape
    check(lib().muse_colsum(part.data_ptr(), out.data_ptr(), rows, cols, 1 if accumulate else 0, stream()), "muse_colsum")
    return out


def add_rowvec_(x, b):
    """x[r, :] += b, f32 in place (LayerNorm bias of a use_bias model)"""
    require_gpu(x, b)
    if x.dtype != torch.float32 or b.dtype != torch.float32 or not x.is_contiguous():
        raise _hip.MuseHipError("add_rowvec_: contiguous f32 rows and an f32 vector")
    check(lib().muse_add_rowvec(x.data_ptr(), b.data_ptr(), x.numel() // x.shape[-1], x.shape[-1], stream()), "muse_add_rowvec")
    return _touched(x)


def bias_grad(dy, cols=None):
    """d(bias)[c] = sum_r dy[r, c] for the first `cols` columns of dy [rows, >= cols] (f32 or bf16, any row stride): two fixed-order
    stages (per-chunk partial rows, then muse_colsum)"""
    require_gpu(dy)
    rows = dy.shape[0]
    cols = dy.shape[1] if cols is None else cols
    if dy.stride(1) != 1:
        raise _hip.MuseHipError("bias_grad: unit column stride")
    R = lib().muse_bias_grad_rows_per_block()
    part = torch.empty(((rows + R - 1) // R, cols), dtype=torch.float32, device=dy.device)
    check(lib().muse_bias_grad_partial(dy.data_ptr(), dt(dy), part.data_ptr(), rows, cols, dy.stride(0), stream()), "muse_bias_grad_partial")
    return colsum(part, torch.empty(cols, dtype=torch.float32, device=dy.device))


def norm_res_bwd(dy, v, w, eps, mode, dpre=None, want_dw=True, also_bf16=False):
    """backward of norm_res_fwd: returns (dv = dx = dres, dw or None[, bf16 copy of dv]).  v = the forward's pre-norm sum."""
    require_gpu(dy, v)
    rows, cols = v.shape
    dv = torch.empty_like(v)
    dvb = torch.empty(v.shape, dtype=torch.bfloat16, device=v.device) if also_bf16 else None
    nblk = lib().muse_norm_res_bwd_nblk(rows)
    part = torch.empty((nblk, cols), dtype=torch.float32, device=v.device)
    check(lib().muse_norm_res_bwd_ex(dy.data_ptr(), ptr(dpre), v.data_ptr(), ptr(w), dv.data_ptr(), ptr(dvb), part.data_ptr(), rows,
                                     cols, eps, mode, stream()), "muse_norm_res_bwd_ex")
    dw = colsum(part, torch.empty(cols, dtype=torch.float32, device=v.device)) if want_dw else None
    return (dv, dw, dvb) if also_bf16 else (dv, dw)


def norm_adaln_ok(rows, cols, batch):
    """shapes the fused norm + AdaLN kernels take (muse_norm_adaln_fwd / _bwd)"""
    return cols % 4 == 0 and cols <= 1024 and rows % batch == 0 and (rows // batch) % 16 == 0


def norm_adaln_fwd(x, w, ss, batch, eps, mode, residual=None, out_dtype=torch.float32, planes_only=False):
    """v = x (+ residual); m = Norm(v) * w * (1 + scale) + shift with (scale | shift) = ss [batch, 2C]  ->  (m in out_dtype, v).
    planes_only (a bf16x3 step, planes_only_ok): m comes back as ops.Planes - operand planes without the f32 tensor"""
    require_gpu(x, ss)
    rows, cols = x.shape
    pre = torch.empty_like(x)
    if planes_only:
        if not (planes_only_ok(rows, cols) and x.dtype == torch.float32 and x.is_contiguous()):
            raise _hip.MuseHipError("norm_adaln_fwd(planes_only=True) outside planes_only_ok")
        planes = planes_alloc((rows, cols), x.device)
        check(lib().muse_norm_adaln_fwd_x3(x.data_ptr(), ptr(residual), ptr(w), ss.data_ptr(), pre.data_ptr(), None, planes.data_ptr(),
                                           batch, rows // batch, cols, eps, mode, stream()), "muse_norm_adaln_fwd_x3")
        return Planes(planes), pre
    m = torch.empty((rows, cols), dtype=out_dtype, device=x.device)
    f32 = out_dtype == torch.float32
    im = _x3_producing(x) if f32 else None
    if im is not None and cols % 8 == 0:        # "bf16x3" mode: m feeds weight GEMMs - its operand planes come out of this kernel
        planes = planes_alloc((rows, cols), x.device)
        check(lib().muse_norm_adaln_fwd_x3(x.data_ptr(), ptr(residual), ptr(w), ss.data_ptr(), pre.data_ptr(), m.data_ptr(), planes.data_ptr(),
                                           batch, rows // batch, cols, eps, mode, stream()), "muse_norm_adaln_fwd_x3")
        im.put_planes(m, planes)
        return m, pre
    check(lib().muse_norm_adaln_fwd(x.data_ptr(), ptr(residual), ptr(w), ss.data_ptr(), pre.data_ptr(), m.data_ptr() if f32 else None,
                                    None if f32 else m.data_ptr(), batch, rows // batch, cols, eps, mode, stream()), "muse_norm_adaln_fwd")
    return m, pre


def norm_adaln_bwd(dm, v, w, ss, batch, eps, mode, dpre=None, also_bf16=False, dss_out=None):
    """backward of norm_adaln_fwd -> (dv = dx = dres, dw, dss [batch, 2C][, bf16 copy of dv])"""
    require_gpu(dm, v, ss)
    rows, cols = v.shape
    dv = torch.empty_like(v)
    dvb = torch.empty(v.shape, dtype=torch.bfloat16, device=v.device) if also_bf16 else None
    nblk = lib().muse_norm_res_bwd_nblk(rows)
    part = torch.empty((nblk, cols), dtype=torch.float32, device=v.device)
    spart = torch.empty((nblk, 2 * cols), dtype=torch.float32, device=v.device)
    im = _x3_producing(v) if not also_bf16 else None
    if im is not None and cols % 8 == 0:        # "bf16x3" mode: dv is the dY of the weight GEMMs below - planes from this kernel
        planes = planes_alloc((rows, cols), v.device)
        check(lib().muse_norm_adaln_bwd_x3(dm.data_ptr(), ptr(dpre), v.data_ptr(), ptr(w), ss.data_ptr(), dv.data_ptr(), planes.data_ptr(),
                                           part.data_ptr(), spart.data_ptr(), batch, rows // batch, cols, eps, mode, stream()), "muse_norm_adaln_bwd_x3")
        im.put_planes(dv, planes)
    else:
        check(lib().muse_norm_adaln_bwd(dm.data_ptr(), ptr(dpre), v.data_ptr(), ptr(w), ss.data_ptr(), dv.data_ptr(), ptr(dvb), part.data_ptr(),
                                        spart.data_ptr(), batch, rows // batch, cols, eps, mode, stream()), "muse_norm_adaln_bwd")
    dw = colsum(part, torch.empty(cols, dtype=torch.float32, device=v.device))
    dss = dss_out if dss_out is not None else torch.empty((batch, 2 * cols), dtype=torch.float32, device=v.device)
    check(lib().muse_colsum_segments(spart.data_ptr(), dss.data_ptr(), batch, nblk // batch, 2 * cols, stream()), "muse_colsum_segments")
    return (dv, dw, dss, dvb) if also_bf16 else (dv, dw, dss)


def adaln_bwd(dy, x, ss, batch, dss_out=None):
    """-> (dx, dss [batch, 2C]); dss_out: a contiguous [batch, 2C] f32 buffer to write dss into"""
    require_gpu(dy, x, ss)
    rows, C_ = x.shape
    dx = torch.empty_like(x)
    dss = dss_out if dss_out is not None else torch.empty(ss.shape, dtype=torch.float32, device=ss.device)
    check(lib().muse_adaln_bwd(dy.data_ptr(), x.data_ptr(), ss.data_ptr(), dx.data_ptr(), dss.data_ptr(), batch, rows // batch, C_,
                               stream()), "muse_adaln_bwd")
    return dx, dss


def silu_bwd(x, dy):
    require_gpu(x, dy)
    dx = torch.empty_like(x)
    check(lib().muse_silu_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), stream()), "muse_silu_bwd")
    return dx


def dwconv3x3_bwd(dy, x, w, B, H, W, C_):
    """-> (dx, dw [C, 1, 3, 3])"""
    require_gpu(dy, x, w)
    dx = torch.empty_like(x)
    nchunk = lib().muse_dwconv3x3_bwd_nchunk(B * H * W)
    part = torch.empty((nchunk, C_ * 9), dtype=torch.float32, device=x.device)
    check(lib().muse_dwconv3x3_bwd(dy.data_ptr(), x.data_ptr(), w.data_ptr(), dx.data_ptr(), part.data_ptr(), B, H, W, C_, stream()),
          "muse_dwconv3x3_bwd")
    dw = colsum(part, torch.empty(C_ * 9, dtype=torch.float32, device=x.device)).view(C_, 1, 3, 3)
    return dx, dw


def grn_bwd(dy, x, gamma, stats, B, S):
    """-> (dx, dgamma [C], dbeta [C])"""
    require_gpu(dy, x, gamma, stats)
    C_ = x.shape[1]
    dx = torch.empty_like(x)
    work = torch.empty(4 * B * C_, dtype=torch.float32, device=x.device)
    check(lib().muse_grn_bwd(dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), stats.data_ptr(), dx.data_ptr(), work.data_ptr(), B, S, C_,
                             stream()), "muse_grn_bwd")
    dbeta = colsum(work[: B * C_].view(B, C_), torch.empty(C_, dtype=torch.float32, device=x.device))
    dgamma = colsum(work[B * C_: 2 * B * C_].view(B, C_), torch.empty(C_, dtype=torch.float32, device=x.device))
    return dx, dgamma, dbeta


def scale_rows_(x, w, num, den, cols):
    """x[r, :cols] *= w[r] * num[0] / den[0]  (in place; num / den are 1-element device tensors)"""
    require_gpu(x, w, num, den)
    check(lib().muse_scale_rows(x.data_ptr(), w.data_ptr(), num.data_ptr(), den.data_ptr(), x.shape[0], cols, x.stride(0), stream()),
          "muse_scale_rows")
    return _touched(x)


def probe_tr16(addr):
    require_gpu(addr)
    out = torch.empty(256, dtype=torch.int32, device=addr.device)
    check(lib().muse_probe_tr16(addr.data_ptr(), out.data_ptr(), stream()), "muse_probe_tr16")
    return out
