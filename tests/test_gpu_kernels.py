"""GPU (-m gpu): every libmuse_hip kernel against a CPU reference on the same seeded inputs, through the C-ABI.

Integer / index outputs are bit-exact; floating point tolerances are written next to each check.
"""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from muse import ops
    return ops


def rnd(shape, seed, scale=1.0):
    rng = np.random.default_rng(seed)
    return torch.from_numpy((rng.standard_normal(shape) * scale).astype(np.float32))


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def test_loaded_native_library():
    from muse import _hip
    assert os.path.exists(_hip.LIB_PATH)
    assert _hip.lib().muse_version() == 1
    assert torch.cuda.is_available()


def test_tr16_lane_mapping():
    """ds_read_b64_tr_b16 semantics the k-major GEMM path relies on: within each 16-lane group the lanes supply a 4x16
    matrix (lane p -> row p/4, cols 4*(p%4)..+3) and lane i receives column i."""
    ops = _ops()
    addr = torch.arange(64, dtype=torch.int32) * 8
    out = ops.probe_tr16(addr.to(DEV)).cpu().view(64, 4)
    exp = torch.empty(64, 4, dtype=torch.int32)
    for l in range(64):
        for j in range(4):
            exp[l, j] = (l & 15) + j * 16 + (l >> 4) * 64
    assert torch.equal(out, exp), out[:20]
    # free row stride: lane p reads row p/4 at stride 144 elements
    addr2 = torch.tensor([((l >> 4) * 4 + ((l & 15) >> 2)) * 288 + (l & 3) * 8 for l in range(64)], dtype=torch.int32)
    out2 = ops.probe_tr16(addr2.to(DEV)).cpu().view(64, 4)
    for l in range(64):
        for j in range(4):
            assert int(out2[l, j]) == ((l >> 4) * 4 + j) * 144 + (l & 15)


GEMM_CASES = [
    # M, N, K, la, lb
    (128, 128, 64, 0, 0), (257, 130, 72, 0, 0), (300, 257, 264, 0, 1), (257, 64, 257, 0, 1), (200, 136, 257, 1, 1),
    (64, 48, 515, 1, 1), (130, 257, 48, 0, 0), (1000, 96, 40, 1, 0),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K,la,lb", GEMM_CASES)
def test_gemm_layouts(dtype, M, N, K, la, lb):
    ops = _ops()
    pad = lambda n: (n + 7) // 8 * 8
    # build operands in their storage layout with zero padding up to a 16-byte chunk
    if la == 0:
        A = torch.zeros(M, pad(K)); A[:, :K] = rnd((M, K), 1)
        Am = A[:, :K]
    else:
        A = torch.zeros(K, pad(M)); A[:, :M] = rnd((K, M), 1)
        Am = A[:, :M].t()
    if lb == 0:
        B = torch.zeros(N, pad(K)); B[:, :K] = rnd((N, K), 2)
        Bm = B[:, :K]
    else:
        B = torch.zeros(K, pad(N)); B[:, :N] = rnd((K, N), 2)
        Bm = B[:, :N].t()
    Ad, Bd = A.to(DEV, dtype), B.to(DEV, dtype)
    ref = (Ad.cpu().double()[:, :K] if la == 0 else Ad.cpu().double()[:, :M].t()) @ \
          (Bd.cpu().double()[:, :K] if lb == 0 else Bd.cpu().double()[:, :N].t()).t()
    C = torch.full((M, N + 3), 7.0, dtype=torch.float32, device=DEV)
    ops.gemm(Ad, Bd, C, M, N, K, la=la, lb=lb, lda=A.shape[1], ldb=B.shape[1], ldc=N + 3, alpha=0.5)
    out = C.cpu()
    assert torch.all(out[:, N:] == 7.0), "wrote outside N"
    tol = 2e-6 if dtype == torch.float32 else 2e-5  # bf16 inputs are exact in the reference; only f32 accumulation order differs
    err = rel_err(out[:, :N], 0.5 * ref)
    assert err < tol * math.sqrt(K), (err, M, N, K, la, lb)


def test_gemm_epilogue_batch_bf16_out():
    ops = _ops()
    Bt, nh, S, hd = 3, 2, 37, 16
    H = nh * hd
    qkv = rnd((Bt * S, 3 * H), 3).to(DEV, torch.bfloat16)
    Sp = 40
    P = torch.zeros((Bt * nh, S, Sp), dtype=torch.bfloat16, device=DEV)
    ops.gemm(qkv, qkv, P, S, S, hd, la=0, lb=0, lda=3 * H, ldb=3 * H, ldc=Sp, a_off=0, b_off=H, alpha=0.25, batch=Bt * nh,
             zdiv=nh, sA=(S * 3 * H, hd), sB=(S * 3 * H, hd), sC=(nh * S * Sp, S * Sp))
    q = qkv.float().cpu().view(Bt, S, 3, nh, hd)
    ref = torch.einsum("bqhd,bkhd->bhqk", q[:, :, 0], q[:, :, 1]) * 0.25
    got = P.float().cpu().view(Bt, nh, S, Sp)[..., :S]
    assert rel_err(got, ref) < 1e-2  # bf16 output rounding
    assert torch.all(P.float().cpu().view(Bt, nh, S, Sp)[..., S:] == 0)
    # bias + gelu + residual + accumulate, f32
    x, w = rnd((70, 24), 4).to(DEV), rnd((50, 24), 5).to(DEV)
    bias, res = rnd((50,), 6).to(DEV), rnd((70, 50), 7).to(DEV)
    out = torch.ones((70, 50), device=DEV)
    ops.gemm(x, w, out, 70, 50, 24, lda=24, ldb=24, ldc=50, bias=bias, residual=res, ldr=50, act=1, accumulate=True)
    ref = F.gelu(x.cpu().double() @ w.cpu().double().t() + bias.cpu().double()) + res.cpu().double() + 1.0
    assert rel_err(out, ref) < 1e-5


@pytest.mark.parametrize("rows,cols", [(5, 32), (130, 768), (67, 3072), (33, 160)])
@pytest.mark.parametrize("din,dout", [(torch.float32, torch.float32), (torch.float32, torch.bfloat16),
                                       (torch.bfloat16, torch.float32), (torch.bfloat16, torch.bfloat16)])
def test_layernorm_fwd_bwd(rows, cols, din, dout):
    ops = _ops()
    x = rnd((rows, cols), 10, 2.0).to(din)
    w = 1.0 + 0.1 * rnd((cols,), 11)
    res = rnd((rows, cols), 12)
    eps = 1e-5
    y, mean, rstd = ops.layernorm_fwd(x.to(DEV), w.to(DEV), eps, dout, residual=res.to(DEV) if dout == torch.float32 else None)
    xr = x.double().requires_grad_(True)
    wr = w.double().requires_grad_(True)
    yr = F.layer_norm(xr, (cols,), wr, None, eps)
    ref = yr + (res.double() if dout == torch.float32 else 0)
    tol = 1e-5 if dout == torch.float32 else 1e-2
    assert rel_err(y.float(), ref.detach()) < tol
    dy = rnd((rows, cols), 13).to(dout)
    dres = rnd((rows, cols), 14)
    dw = torch.empty(cols, device=DEV)
    dx = ops.layernorm_bwd(dy.to(DEV), x.to(DEV), w.to(DEV), mean, rstd, din if din == torch.bfloat16 else torch.float32, dw,
                           False, dres=dres.to(DEV))
    yr.backward(dy.double())
    tol = 2e-5 if din == torch.float32 else 1e-2
    assert rel_err(dx.float(), xr.grad + dres.double()) < tol
    assert rel_err(dw, wr.grad) < 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_softmax_glu_gelu(dtype):
    ops = _ops()
    rows, cols, ld = 300, 257, 264
    x = torch.zeros(rows, ld)
    x[:, :cols] = rnd((rows, cols), 20, 3.0)
    x[:, cols:] = float("nan") if dtype == torch.float32 else 1e30  # pad garbage must not leak
    xd = x.to(DEV, dtype)
    ref = torch.softmax(xd.cpu().double()[:, :cols], dim=-1)
    p = ops.softmax_(xd.clone(), rows, cols, ld)
    tol = 1e-6 if dtype == torch.float32 else 1e-2
    assert rel_err(p[:, :cols].float(), ref) < tol
    assert torch.all(p[:, cols:].float() == 0)
    dp = torch.zeros(rows, ld)
    dp[:, :cols] = rnd((rows, cols), 21)
    dpd = dp.to(DEV, dtype)
    pr = p.cpu().double()[:, :cols]
    dsr = pr * (dpd.cpu().double()[:, :cols] - (pr * dpd.cpu().double()[:, :cols]).sum(-1, keepdim=True))
    ds = ops.softmax_bwd_(p, dpd.clone(), rows, cols, ld)
    assert rel_err(ds[:, :cols].float(), dsr) < (1e-5 if dtype == torch.float32 else 2e-2)
    # GLU
    ab = rnd((50, 2 * 96), 22).to(dtype)
    a, b = ab.double()[:, :96].requires_grad_(True), ab.double()[:, 96:].requires_grad_(True)
    hr = F.gelu(a) * b
    h = ops.glu_fwd(ab.to(DEV))
    assert rel_err(h.float(), hr.detach()) < (1e-6 if dtype == torch.float32 else 1e-2)
    dh = rnd((50, 96), 23).to(dtype)
    hr.backward(dh.double())
    dab = ops.glu_bwd(ab.to(DEV), dh.to(DEV))
    assert rel_err(dab.float(), torch.cat([a.grad, b.grad], 1)) < (2e-6 if dtype == torch.float32 else 1e-2)
    # GELU
    xg = rnd((64, 40), 24, 2.0).to(dtype)
    xr = xg.double().requires_grad_(True)
    yr = F.gelu(xr)
    assert rel_err(ops.gelu_fwd(xg.to(DEV)).float(), yr.detach()) < (1e-6 if dtype == torch.float32 else 1e-2)
    yr.backward(dh.double()[:, :40].repeat(2, 1)[:64])
    dxg = ops.gelu_bwd(xg.to(DEV), dh[:, :40].repeat(2, 1)[:64].contiguous().to(DEV))
    assert rel_err(dxg.float(), xr.grad) < (2e-6 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("ls", [0.0, 0.1])
@pytest.mark.parametrize("V,ld", [(48, 48), (2025, 2032)])
def test_cross_entropy(ls, V, ld):
    ops = _ops()
    rows = 203
    logits = torch.zeros(rows, ld)
    logits[:, :V] = rnd((rows, V), 30, 2.0)
    rng = np.random.default_rng(31)
    labels = torch.from_numpy(rng.integers(0, V, size=rows))
    labels[rng.random(rows) < 0.4] = -100
    lr = logits[:, :V].double().requires_grad_(True)
    ref = F.cross_entropy(lr, labels, ignore_index=-100, label_smoothing=ls)
    ref.backward(torch.tensor(2.0, dtype=torch.float64))
    ld_ = logits.to(DEV)
    loss_out, lse = ops.cross_entropy_fwd(ld_, labels.to(DEV), ls, vocab=V)
    assert abs(float(loss_out[0]) - float(ref)) < 1e-5 * abs(float(ref))  # f32 loss, rel 1e-5
    assert int(loss_out[1]) == int((labels >= 0).sum())
    dl = ops.cross_entropy_bwd(ld_, labels.to(DEV), lse, loss_out, torch.tensor([2.0], device=DEV), ls, torch.float32, vocab=V)
    assert rel_err(dl[:, :V], lr.grad) < 1e-5
    assert torch.all(dl[:, V:] == 0)


@pytest.mark.parametrize("B,S,H,V,heavy", [(6, 17, 32, 48, 6), (16, 257, 768, 3049, 120), (3, 5, 20, 7, 0)])
def test_embedding_fwd_bwd_deterministic(B, S, H, V, heavy):
    """embedding forward (bit-exact) and backward: the sort-based segmented sum (csrc/embed.hip) against an f64 index_add, incl. a row
    hit by ~half of all tokens (several 64-row segments), rows without hits, accumulation and run-to-run determinism"""
    ops = _ops()
    rng = np.random.default_rng(40)
    ids = torch.from_numpy(rng.integers(0, V, size=(B, S)))
    ids[:, 3:3 + heavy] = V - 1  # heavy hitter (mask token)
    P = S + 7
    word, pos = rnd((V, H), 41), rnd((P, H), 42)
    out = ops.embed_fwd(ids.to(DEV), word.to(DEV), pos.to(DEV))
    ref = word[ids] + pos[:S][None]
    assert torch.equal(out.cpu().view(B, S, H), ref)  # one f32 add: bit-exact
    dout = rnd((B * S, H), 43)
    dword = torch.full((V, H), 5.0, device=DEV)
    dpos = torch.full((P, H), 5.0, device=DEV)
    ops.embed_bwd(ids.to(DEV), dout.to(DEV), dword, dpos, False)
    rw = torch.zeros(V, H, dtype=torch.float64).index_add_(0, ids.view(-1), dout.double())
    rp = dout.double().view(B, S, H).sum(0)
    assert rel_err(dword, rw) < 1e-6
    assert rel_err(dpos[:S], rp) < 1e-6
    d2 = torch.empty_like(dword)
    p2 = torch.empty_like(dpos)
    ops.embed_bwd(ids.to(DEV), dout.to(DEV), d2, p2, False)
    assert torch.equal(d2, dword)  # run-to-run deterministic (no float atomics)
    ops.embed_bwd(ids.to(DEV), dout.to(DEV), d2, p2, True)   # accumulate: 2 x the gradient (rows without hits stay 0)
    assert rel_err(d2, 2 * rw) < 1e-6 and rel_err(p2[:S], 2 * rp) < 1e-6


def test_adamw_multi_matches_flat():
    """muse_adamw_multi (one launch over a device table of tensors) == muse_adamw_flat per tensor, bit for bit: sizes that are not
    multiples of 4 or of the 4096-element chunk, a 4-byte-aligned (not 16-byte) view, with and without the bf16 shadow"""
    ops = _ops()
    sizes = [1, 3, 4096, 4097, 10007, 8192 * 3 + 2, 5]
    rng_seed = 70
    tensors = []
    for i, n in enumerate(sizes):
        off = 1 if i == 4 else 0     # one tensor starts 4 bytes into its allocation
        def mk(seed, scale=1.0, positive=False):
            t = torch.zeros(n + off, device=DEV)
            v = rnd((n,), seed, scale).to(DEV)
            t[off:] = v.abs() if positive else v
            return t[off:]
        p, g = mk(rng_seed + 4 * i), mk(rng_seed + 4 * i + 1, 0.1)
        m, v = mk(rng_seed + 4 * i + 2, 0.01), mk(rng_seed + 4 * i + 3, 1e-4, positive=True)
        sh = torch.zeros(n, dtype=torch.bfloat16, device=DEV) if i % 2 == 0 else None
        tensors.append((p, g, m, v, sh))
    ref = [tuple(None if t is None else t.clone() for t in tt) for tt in tensors]
    hp = dict(lr=1e-3, beta1=0.9, beta2=0.99, eps=1e-8, weight_decay=0.05, step=3)
    for p, g, m, v, sh in ref:
        ops.adamw_flat(p, g, m, v, sh, hp["lr"], hp["beta1"], hp["beta2"], hp["eps"], hp["weight_decay"], hp["step"], grad_scale=0.5)
    rows, first, nch = [], [], 0
    for p, g, m, v, sh in tensors:
        rows.append((p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), sh.data_ptr() if sh is not None else 0, p.numel()))
        first.append(nch)
        nch += (p.numel() + 4095) // 4096
    first.append(nch)
    table = torch.tensor(rows, dtype=torch.int64).to(DEV)
    cf = torch.tensor(first, dtype=torch.int32).to(DEV)
    ops.adamw_multi(table, cf, len(rows), nch, hp["lr"], hp["beta1"], hp["beta2"], hp["eps"], hp["weight_decay"], hp["step"], grad_scale=0.5)
    for a, b in zip(tensors, ref):
        for x, y in zip(a, b):
            if x is not None and x.dtype != torch.bfloat16:
                assert torch.equal(x, y)
            elif x is not None:
                assert torch.equal(x.view(torch.int16), y.view(torch.int16))


@pytest.mark.parametrize("split", [1, 2, 3])
def test_grouped_weight_gradients_match_per_product_launches(split, monkeypatch):
    """muse_gemm_group (the four dW = dY^T X of a transformer layer as ONE launch over the concatenated tile lists, gemm256.h:
    kernel_group) + muse_sum_multi: every product bit-identical to its own muse_gemm launch with the same K split + muse_sum_slices
    (same kernel body, same slices, same summation order), and right against float64; ragged token counts, a product whose dY rows
    are wider than its valid columns (logits head: M = V, lda = Vp), accumulate into an existing gradient."""
    ops = _ops()
    monkeypatch.setenv("MUSE_GEMM256", "1")                 # the per-product reference launches take the 256^2 kernel too (cost model off)
    T = 64 * 5 + 40                                         # K = tokens: not a multiple of the 64-wide K-tile
    shapes = [(512, 256), (256, 768), (768, 256), (256, 256)]   # (N_out, K_in): 2 + 3 + 3 + 1 tiles
    items, refs = [], []
    for i, (N, K) in enumerate(shapes):
        ld = N + 8 if i == 3 else N
        dy = rnd((T, ld), 300 + i, 0.5).to(DEV).to(torch.bfloat16)
        x = rnd((T, K), 310 + i, 0.5).to(DEV).to(torch.bfloat16)
        acc = i == 1
        dw0 = rnd((N, K), 320 + i).to(DEV) if acc else torch.full((N, K), float("nan"), device=DEV)
        items.append((dy, x, dw0.clone(), acc, N if i == 3 else None, ld if i == 3 else None))
        want = dw0.clone()
        if split > 1:
            ws = torch.empty((split, N, K), dtype=torch.float32, device=DEV)
            ops.gemm(dy, x, ws, N, K, T, la=1, lb=1, lda=ld, ldb=K, ldc=K, split_k=split, split_stride=N * K)
            nk = (T + 63) // 64
            per = (nk + split - 1) // split
            from muse._hip import check, lib, stream
            check(lib().muse_sum_slices(ws.data_ptr(), want.data_ptr(), (nk + per - 1) // per, N * K, N * K, 1 if acc else 0, stream()), "sum")
        else:
            ops.gemm(dy, x, want, N, K, T, la=1, lb=1, lda=ld, ldb=K, ldc=K, accumulate=acc)
        refs.append((want, (dw0.double() if acc else 0) + dy[:, :N].double().t() @ x.double()))
    ops.linear_wgrad_group(items, None, split=split)
    for (dy, x, dw, acc, M, lda), (want, f64) in zip(items, refs):
        assert torch.equal(dw, want), float((dw - want).abs().max())
        assert rel_err(dw, f64) < 5e-6
    from muse._hip import GemmDesc, lib
    assert lib().muse_gemm_group_ok((GemmDesc * 1)(), 1, 1) != 0          # an empty descriptor is refused, not launched


def test_sum_multi_is_bit_identical_to_the_single_job_kernels():
    """muse_sum_multi: several slice sums (muse_sum_slices) and column sums (muse_colsum) in one launch, each bit-identical to its
    single-job kernel; with and without accumulation, sizes that do not fill the last work item"""
    ops = _ops()
    from muse._hip import check, lib, stream
    jobs, want = [], []
    for i, (ns, n) in enumerate([(2, 4096 * 3 + 8), (7, 1000), (3, 4096)]):
        ws = rnd((ns, n), 400 + i).to(DEV)
        acc = i == 1
        out = rnd((n,), 410 + i).to(DEV) if acc else torch.full((n,), float("nan"), device=DEV)
        ref = out.clone()
        check(lib().muse_sum_slices(ws.data_ptr(), ref.data_ptr(), ns, n, n, 1 if acc else 0, stream()), "sum")
        jobs.append((0, ws, out, ns, n, n, acc)); want.append(ref)
    for i, (rows, cols) in enumerate([(514, 768), (129, 3072), (5, 20)]):
        part = rnd((rows, cols), 420 + i).to(DEV)
        acc = i == 2
        out = rnd((cols,), 430 + i).to(DEV) if acc else torch.full((cols,), float("nan"), device=DEV)
        ref = out.clone()
        check(lib().muse_colsum(part.data_ptr(), ref.data_ptr(), rows, cols, 1 if acc else 0, stream()), "colsum")
        jobs.append((1, part, out, rows, cols, cols, acc)); want.append(ref)
    ops.sum_multi(jobs)
    for j, w in zip(jobs, want):
        assert torch.equal(j[2], w), (j[0], j[3], j[4])
    assert rel_err(jobs[3][2], jobs[3][1].double().sum(0)) < 1e-6


def test_adamw_flat_groups_matches_torch_and_flat():
    """muse_adamw_flat_groups: a flat buffer whose segments belong to different parameter groups (training/train_muse.py:425-445:
    weight decay on the matrices, none on bias / LayerNorm / embedding weights; here also a third group with its own lr / betas / eps).
    (1) one group: bit-identical to muse_adamw_flat; (2) segment boundaries that are not multiples of 4 or of the 4096-element chunk,
    range-wise calls (base > 0) == one call over the whole buffer, bit for bit; (3) == torch.optim.AdamW with the same groups."""
    ops = _ops()
    sizes = [5000, 768, 4096, 3, 10001, 8, 4100, 12288]            # parameters, back to back (offsets not aligned to anything)
    gid = [0, 1, 0, 2, 0, 1, 1, 0]
    groups = [dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05), dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0),
              dict(lr=3e-4, betas=(0.8, 0.95), eps=1e-6, weight_decay=0.1)]
    n = sum(sizes)
    npad = (n + 3) // 4 * 4
    p0, g0 = rnd((npad,), 90), rnd((npad,), 91, 0.1)
    def fresh():
        return (p0.to(DEV).clone(), g0.to(DEV).clone(), torch.zeros(npad, device=DEV), torch.zeros(npad, device=DEV),
                torch.zeros(npad, dtype=torch.bfloat16, device=DEV))
    ends, gids, o = [], [], 0
    for sz, k in zip(sizes, gid):
        o += sz
        if gids and gids[-1] == k:
            ends[-1] = o
        else:
            ends.append(o); gids.append(k)
    ends[-1] = npad
    seg_end = torch.tensor(ends, dtype=torch.int64, device=DEV)
    seg_group = torch.tensor(gids, dtype=torch.int32, device=DEV)
    # (1) one group == muse_adamw_flat
    a, b = fresh(), fresh()
    one_end = torch.tensor([npad], dtype=torch.int64, device=DEV)
    one_grp = torch.zeros(1, dtype=torch.int32, device=DEV)
    for step in (1, 2):
        ops.adamw_flat(a[0], a[1], a[2], a[3], a[4], 1e-3, 0.9, 0.999, 1e-8, 0.05, step, grad_scale=0.5)
        ops.adamw_flat_groups(b[0], b[1], b[2], b[3], b[4], 0, one_end, one_grp, groups[:1], step, grad_scale=0.5)
    for x, y in zip(a, b):
        assert torch.equal(x.view(torch.int16) if x.dtype == torch.bfloat16 else x, y.view(torch.int16) if y.dtype == torch.bfloat16 else y)
    # (2) whole buffer == three ranges
    a, b = fresh(), fresh()
    cuts = [0, 5768, 5768 + 4096 + 4, npad]      # 16-byte aligned range starts, inside and between segments
    for step in (1, 2, 3):
        ops.adamw_flat_groups(a[0], a[1], a[2], a[3], a[4], 0, seg_end, seg_group, groups, step)
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            ops.adamw_flat_groups(b[0][lo:hi], b[1][lo:hi], b[2][lo:hi], b[3][lo:hi], b[4][lo:hi], lo, seg_end, seg_group, groups, step)
    for x, y in zip(a, b):
        assert torch.equal(x.view(torch.int16) if x.dtype == torch.bfloat16 else x, y.view(torch.int16) if y.dtype == torch.bfloat16 else y)
    assert torch.equal(a[4][:n].cpu(), a[0][:n].cpu().to(torch.bfloat16))
    # (3) torch.optim.AdamW with the same groups
    twins, o = [], 0
    for sz in sizes:
        twins.append(torch.nn.Parameter(p0[o:o + sz].clone())); o += sz
    ref = torch.optim.AdamW([dict(params=[t for t, k in zip(twins, gid) if k == j], **groups[j]) for j in range(3)])
    for step in (1, 2, 3):
        o = 0
        for t, sz in zip(twins, sizes):
            t.grad = g0[o:o + sz].clone(); o += sz
        ref.step()
    o = 0
    for t, sz, k in zip(twins, sizes, gid):
        assert float((a[0][o:o + sz].cpu() - t.data).abs().max()) < 2e-6, (o, k)
        o += sz


def test_adamw_multi_groups_matches_multi_and_torch():
    """muse_adamw_multi_groups (table with a group column): one group == muse_adamw_multi bit for bit; two groups == torch.optim.AdamW"""
    ops = _ops()
    sizes = [1, 3, 4096, 4097, 10007, 5]
    gid = [0, 1, 0, 0, 1, 1]
    groups = [dict(lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.05), dict(lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.0)]
    def build():
        ts = []
        for i, sz in enumerate(sizes):
            ts.append((rnd((sz,), 120 + 2 * i).to(DEV), rnd((sz,), 121 + 2 * i, 0.1).to(DEV), torch.zeros(sz, device=DEV),
                       torch.zeros(sz, device=DEV)))
        return ts
    def table(ts, with_group, gids):
        rows, first, nch = [], [], 0
        for (p, g, m, v), k in zip(ts, gids):
            r = (p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), 0, p.numel())
            rows.append(r + (k,) if with_group else r)
            first.append(nch); nch += (p.numel() + 4095) // 4096
        first.append(nch)
        return torch.tensor(rows, dtype=torch.int64).to(DEV), torch.tensor(first, dtype=torch.int32).to(DEV), nch
    a, b = build(), build()
    ta, fa, nch = table(a, False, gid)
    tb, fb, _ = table(b, True, [0] * len(sizes))
    for step in (1, 2):
        ops.adamw_multi(ta, fa, len(sizes), nch, 1e-3, 0.9, 0.99, 1e-8, 0.05, step)
        ops.adamw_multi_groups(tb, fb, len(sizes), nch, groups[:1], step)
    for x, y in zip(a, b):
        for u, w in zip(x, y):
            assert torch.equal(u, w)
    c = build()
    tc, fc, _ = table(c, True, gid)
    twins = [torch.nn.Parameter(t[0].cpu().clone()) for t in c]
    ref = torch.optim.AdamW([dict(params=[t for t, k in zip(twins, gid) if k == j], **groups[j]) for j in range(2)])
    for step in (1, 2, 3):
        for t, x in zip(twins, c):
            t.grad = x[1].cpu().clone()
        ref.step()
        ops.adamw_multi_groups(tc, fc, len(sizes), nch, groups, step)
    for t, x in zip(twins, c):
        assert float((x[0].cpu() - t.data).abs().max()) < 2e-6


def test_adamw_matches_torch():
    ops = _ops()
    n = 4099
    p0, g = rnd((n,), 50), rnd((n,), 51, 0.1)
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([p], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    pd = p0.to(DEV).clone()
    pad = torch.zeros(4100, device=DEV); pad[:n] = pd
    gd = torch.zeros(4100, device=DEV); gd[:n] = g.to(DEV)
    m, v = torch.zeros(4100, device=DEV), torch.zeros(4100, device=DEV)
    shadow = torch.zeros(4100, dtype=torch.bfloat16, device=DEV)
    for step in range(1, 4):
        p.grad = g.clone()
        opt.step()
        ops.adamw_flat(pad, gd, m, v, shadow, 1e-3, 0.9, 0.999, 1e-8, 0.01, step)
    assert float((pad[:n].cpu() - p.data).abs().max()) < 1e-6  # a few f32 ulp after 3 steps
    assert torch.equal(shadow[:n].cpu(), pad[:n].cpu().to(torch.bfloat16))


@pytest.mark.parametrize("name", ["mask_b64", "mask_small"])
def test_mask_sampling_bit_exact(golden_dir, name):
    """HIP mask sampler vs the reference's own output (golden) and the oracle: bit-exact ids / labels."""
    ops = _ops()
    from oracle import maskgit_oracle as O
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    tok, cls = torch.from_numpy(g["image_tokens"]), torch.from_numpy(g["class_ids"])
    t, nz = torch.from_numpy(g["timesteps"]), torch.from_numpy(g["noise"])
    ids, labels, prob = ops.mask_sample(tok.to(DEV), cls.to(DEV), t.to(DEV), nz.to(DEV), int(g["mask_id"]),
                                        int(g["codebook_size"]), float(g["min_rate"]))
    assert np.array_equal(ids.cpu().numpy(), g["input_ids"])
    assert np.array_equal(labels.cpu().numpy(), g["labels"])
    np.testing.assert_allclose(prob.cpu().numpy(), g["mask_prob"], rtol=2e-7, atol=1e-7)  # <= 1 ulp (f32 cos)
    o_ids, o_lab, _ = O.prepare_inputs_and_labels(tok, cls, t, nz, int(g["mask_id"]), int(g["codebook_size"]), float(g["min_rate"]))
    assert torch.equal(ids.cpu(), o_ids) and torch.equal(labels.cpu(), o_lab)


def test_mask_sampling_full_size_properties():
    ops = _ops()
    B, S = 64, 256
    rng = np.random.default_rng(60)
    tok = torch.from_numpy(rng.integers(0, 1024, size=(B, S))).to(DEV)
    cls = torch.from_numpy(rng.integers(0, 1000, size=(B,))).to(DEV)
    t = torch.from_numpy(rng.random(B).astype(np.float32)).to(DEV)
    nz = torch.from_numpy(rng.random((B, S)).astype(np.float32)).to(DEV)
    ids, labels, prob = ops.mask_sample(tok, cls, t, nz, 2047, 1024)
    masked = ids[:, 1:] == 2047
    k = torch.clamp(torch.round(S * prob), min=1).long()
    assert torch.equal(masked.sum(-1), k)                         # exactly k masked per row
    assert torch.equal(labels[:, 1:][masked], tok[masked])        # labels carry the hidden tokens
    assert torch.all(labels[:, 1:][~masked] == -100) and torch.all(labels[:, 0] == -100)
    assert torch.equal(ids[:, 1:][~masked], tok[~masked]) and torch.equal(ids[:, 0], cls + 1024)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Cin,Cout,KS,ups,bias,res", [(32, 64, 3, False, False, False), (64, 32, 1, False, False, True),
                                                       (8, 40, 3, False, True, False), (32, 32, 3, True, True, False),
                                                       (128, 3, 3, False, True, False)])
def test_conv2d_nhwc(dtype, Cin, Cout, KS, ups, bias, res):
    ops = _ops()
    B, H, W = 2, 12, 10
    ih, iw = (H // 2, W // 2) if ups else (H, W)
    x = rnd((B, Cin, ih, iw), 70).to(dtype)
    w = (rnd((Cout, Cin, KS, KS), 71) / math.sqrt(Cin * KS * KS)).to(dtype)
    bvec = rnd((Cout,), 72) if bias else None
    rr = rnd((B, Cout, H, W), 73).to(dtype) if res else None
    xr = x.double()
    if ups:
        xr = F.interpolate(xr, scale_factor=2.0, mode="nearest")
    p = KS - 1
    ref = F.conv2d(F.pad(xr, [p // 2, p - p // 2, p // 2, p - p // 2]), w.double(), bvec.double() if bias else None)
    if res:
        ref = ref + rr.double()
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wn = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    out = ops.conv2d_nhwc(xn, wn, B, H, W, Cin, Cout, KS, bias=bvec.to(DEV) if bias else None,
                          residual=rr.permute(0, 2, 3, 1).contiguous().to(DEV) if res else None, upsample=ups)
    got = out.float().cpu().permute(0, 3, 1, 2)
    assert rel_err(got, ref) < (1e-5 if dtype == torch.float32 else 1e-2), (Cin, Cout, KS, ups)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,HW", [(32, 64), (64, 1500), (128, 4096), (512, 256)])
def test_groupnorm_silu_and_pool(dtype, C, HW):
    ops = _ops()
    B = 2
    x = (rnd((B, C, HW), 80, 1.5) + 0.7).to(dtype)
    gam, bet = 1 + 0.1 * rnd((C,), 81), 0.1 * rnd((C,), 82)
    ref = F.silu(F.group_norm(x.double(), 32, gam.double(), bet.double(), 1e-6))
    xn = x.permute(0, 2, 1).contiguous().to(DEV)
    y = ops.groupnorm_silu_nhwc(xn, gam.to(DEV), bet.to(DEV), B, HW, C)
    assert rel_err(y.float().cpu().permute(0, 2, 1), ref) < (2e-6 if dtype == torch.float32 else 1e-2)
    if HW == 4096:
        xi = x.view(B, C, 64, 64)
        pr = F.avg_pool2d(xi.double(), 2, 2)
        yp = ops.avgpool2x2_nhwc(xi.permute(0, 2, 3, 1).contiguous().to(DEV), B, 64, 64, C)
        assert rel_err(yp.float().cpu().permute(0, 3, 1, 2), pr) < (1e-6 if dtype == torch.float32 else 1e-2)


def test_layout_and_vq_lookup():
    ops = _ops()
    from oracle import maskgit_oracle as O
    x = rnd((2, 3, 8, 8), 90)
    n = ops.nchw_to_nhwc(x.to(DEV), torch.float32, 4)
    assert torch.equal(n[..., :3].cpu(), x.permute(0, 2, 3, 1)) and torch.all(n[..., 3] == 0)
    assert torch.equal(ops.nhwc_to_nchw(n, 3).cpu(), x)
    # VQ nearest neighbour: bit-exact indices vs the oracle on a well separated codebook and on the U(+-1/K) init
    z = rnd((4096, 256), 91)
    for cb in (0.5 * rnd((1024, 256), 92), (torch.from_numpy(np.random.default_rng(93).random((1024, 256)).astype(np.float32)) * 2 - 1) / 1024):
        zf = z if cb.abs().max() > 0.1 else 0.05 * z
        idx = ops.vq_nearest(zf.to(DEV), cb.to(DEV)).cpu()
        dist = O.vq_distances(zf, cb)
        ref = dist.argmin(1)
        mism = (idx != ref).nonzero().flatten()
        # any disagreement must be an f32 near-tie of the reference's own distances (|d_a - d_b| <= 4 ulp)
        for r in mism.tolist():
            da, db = float(dist[r, idx[r]]), float(dist[r, ref[r]])
            assert abs(da - db) <= 4 * np.spacing(np.float32(abs(db))), (r, da, db)
        assert len(mism) <= 4, len(mism)
    g = ops.gather_rows(cb.to(DEV), idx.to(DEV), torch.float32)
    assert torch.equal(g.cpu(), cb[idx])


@pytest.mark.parametrize("B,S,nh,hd", [(2, 17, 2, 16), (2, 37, 2, 48), (3, 257, 4, 64), (2, 257, 3, 48), (1, 64, 2, 32)])
def test_fused_attention_fwd_bwd(B, S, nh, hd):
    """fused attention vs softmax(alpha q k^T) v evaluated in f64 on the same bf16 inputs (and its autograd backward)"""
    ops = _ops()
    H = nh * hd
    alpha = 1.0 / float(torch.sqrt(torch.tensor(hd, dtype=torch.float32)))
    qkv = (rnd((B * S, 3 * H), 100, 1.0)).to(torch.bfloat16)
    dctx = rnd((B * S, H), 101).to(torch.bfloat16)
    x = qkv.double().view(B, S, 3, nh, hd).requires_grad_(True)
    q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
    p = torch.softmax(q @ k.transpose(-1, -2) * alpha, dim=-1)
    ref = (p @ v).transpose(1, 2).reshape(B * S, H)
    ref.backward(dctx.double())
    ctx, lse = ops.attention_fwd(qkv.to(DEV), B, S, nh, hd, alpha)
    assert rel_err(ctx.float(), ref.detach()) < 1.5e-2          # bf16 P and bf16 output
    lse_ref = torch.logsumexp(q @ k.transpose(-1, -2) * alpha, dim=-1).reshape(B * nh, S)
    assert rel_err(lse[:, :S], lse_ref.detach()) < 1e-3   # the row sums are sums of the bf16-rounded P (they come out of an MFMA)
    dqkv = ops.attention_bwd(qkv.to(DEV), ctx, dctx.to(DEV), lse, B, S, nh, hd, alpha)
    g = x.grad.reshape(B * S, 3 * H)
    for i, nm in enumerate("qkv"):
        e = rel_err(dqkv[:, i * H:(i + 1) * H].float(), g[:, i * H:(i + 1) * H])
        assert e < 3e-2, (nm, e)


@pytest.mark.parametrize("B,S,nh", [(5, 257, 4), (3, 257, 16), (3, 256, 4), (2, 240, 3), (3, 258, 4), (2, 260, 3), (2, 225, 2)])
def test_one_tile_attention_blocks32(B, S, nh, monkeypatch):
    """attention2.hip (32 x 32 MFMA blocks, exact softmax, operands by LDS-DMA, ONE fused two-phase backward kernel) on
    the shapes it takes (self-attention, head_dim 48, 225 <= S <= 260): against float64 on the same bf16 inputs, and against the
    general kernels of attention.hip (MUSE_ATTN2=0) - both round P to bf16 once, so they agree far inside the f64 tolerance.
    One head per 4-wave workgroup; several heads per image and several images so that the XCD remap of the grid is exercised;
    S = 256 / 240 / 225 (no 9th block, keys masked inside the 8th), 258 / 260 (two / four real rows in the 9th block)."""
    ops = _ops()
    hd = 48
    H = nh * hd
    alpha = 1.0 / float(torch.sqrt(torch.tensor(hd, dtype=torch.float32)))
    qkv = (rnd((B * S, 3 * H), 140, 1.0)).to(torch.bfloat16)
    dctx = rnd((B * S, H), 141).to(torch.bfloat16)
    x = qkv.double().view(B, S, 3, nh, hd).requires_grad_(True)
    q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
    sc = q @ k.transpose(-1, -2) * alpha
    ref = (torch.softmax(sc, dim=-1) @ v).transpose(1, 2).reshape(B * S, H)
    ref.backward(dctx.double())
    g = x.grad.reshape(B * S, 3 * H)
    monkeypatch.setenv("MUSE_ATTN2", "1")
    ctx, lse = ops.attention_fwd(qkv.to(DEV), B, S, nh, hd, alpha)
    dqkv = ops.attention_bwd(qkv.to(DEV), ctx, dctx.to(DEV), lse, B, S, nh, hd, alpha)
    monkeypatch.setenv("MUSE_ATTN2", "0")
    ctx0, lse0 = ops.attention_fwd(qkv.to(DEV), B, S, nh, hd, alpha)
    dqkv0 = ops.attention_bwd(qkv.to(DEV), ctx0, dctx.to(DEV), lse0, B, S, nh, hd, alpha)
    assert torch.isfinite(ctx.float()).all() and torch.isfinite(dqkv.float()).all()
    assert rel_err(ctx.float(), ref.detach()) < 1.5e-2
    assert rel_err(lse[:, :S], torch.logsumexp(sc, dim=-1).reshape(B * nh, S).detach()) < 1e-3
    for i, nm in enumerate("qkv"):
        e = rel_err(dqkv[:, i * H:(i + 1) * H].float(), g[:, i * H:(i + 1) * H])
        assert e < 3e-2, (nm, e)
    # the two kernel families: same contract, different summation orders
    assert rel_err(ctx.float(), ctx0.double()) < 8e-3 and rel_err(lse[:, :S], lse0[:, :S].double()) < 1e-4
    assert rel_err(dqkv.float(), dqkv0.double()) < 1.5e-2
    # every row of every head written (no stale rows from a skipped block): per-row agreement, not just a norm
    rowdiff = (ctx.float() - ref.detach().float().to(DEV)).abs().amax(dim=1)
    assert float(rowdiff.max()) < 0.1, float(rowdiff.max())
    growdiff = (dqkv.float() - g.float().to(DEV)).abs().amax(dim=1) / g.abs().max().float()
    assert float(growdiff.max()) < 0.1, float(growdiff.max())


@pytest.mark.parametrize("B,Skv,nh,packed", [(3, 256, 4, True), (2, 256, 16, True), (3, 77, 4, False), (2, 96, 3, False), (2, 240, 2, False),
                                              (2, 65, 2, False)])
def test_fused_attention_bf16x3(B, Skv, nh, packed):
    """attention3.hip: the attention core of the "bf16x3" mode (f32 tensors, every product as three bf16 MFMA products of hi / lo planes
    the kernel splits itself) - head_dim 64, 256 queries, self-attention on a packed q|k|v projection and cross-attention against 77
    (65 .. 96) keys with the last key block masked.  Against float64: forward <= 2e-5, gradients <= 5e-5 of their scale (a bf16 core
    is 1.5e-2 / 3e-2), i.e. the precision class of the mode's GEMMs; every row of every head written; and against the materialised
    route the tape engines used before (exact-f32 batched products + softmax kernels)."""
    ops = _ops()
    Sq, hd = 256, 64
    H = nh * hd
    alpha = 1.0 / float(torch.sqrt(torch.tensor(hd, dtype=torch.float32)))
    if packed:
        qkv = rnd((B * Sq, 3 * H), 160, 1.0)
        qc, kc, vc = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
        qkv_d = qkv.to(DEV)
        qd, kd, vd = qkv_d[:, :H], qkv_d[:, H:2 * H], qkv_d[:, 2 * H:]
    else:
        qc, kv = rnd((B * Sq, H), 161, 1.0), rnd((B * Skv, 2 * H), 162, 1.0)
        kc, vc = kv[:, :H], kv[:, H:]
        qd, kv_d = qc.to(DEV), kv.to(DEV)
        kd, vd = kv_d[:, :H], kv_d[:, H:]
    dctx = rnd((B * Sq, H), 163)
    q = qc.double().reshape(B, Sq, nh, hd).transpose(1, 2).detach().requires_grad_(True)
    k = kc.double().reshape(B, Skv, nh, hd).transpose(1, 2).detach().requires_grad_(True)
    v = vc.double().reshape(B, Skv, nh, hd).transpose(1, 2).detach().requires_grad_(True)
    sc = q @ k.transpose(-1, -2) * alpha
    ref = (torch.softmax(sc, dim=-1) @ v).transpose(1, 2).reshape(B * Sq, H)
    ref.backward(dctx.double())
    gq, gk, gv = (t.grad.transpose(1, 2).reshape(-1, H) for t in (q, k, v))
    assert ops.attention_x3_supported(Sq, Skv, hd)
    ctx, lse = ops.attention_x3_fwd(qd, kd, vd, B, Sq, Skv, nh, hd, alpha)
    dq, dk, dv = ops.attention_x3_bwd(qd, kd, vd, ctx, dctx.to(DEV), lse, B, Sq, Skv, nh, hd, alpha)
    assert ctx.dtype == torch.float32 and torch.isfinite(ctx).all() and all(torch.isfinite(t).all() for t in (dq, dk, dv))
    ef = rel_err(ctx, ref.detach())
    el = rel_err(lse, torch.logsumexp(sc, dim=-1).reshape(B * nh, Sq).detach())
    eg = [rel_err(a, b) for a, b in ((dq, gq), (dk, gk), (dv, gv))]
    print(f"bf16x3 attention S_kv {Skv}: ctx {ef:.1e}, lse {el:.1e}, dq / dk / dv {eg[0]:.1e} / {eg[1]:.1e} / {eg[2]:.1e} of float64")
    assert ef < 2e-5 and el < 2e-6 and max(eg) < 5e-5
    rowdiff = (ctx.double().cpu() - ref.detach()).abs().amax(dim=1)
    assert float(rowdiff.max()) < 1e-4, float(rowdiff.max())
    for a, b in ((dq, gq), (dk, gk), (dv, gv)):
        assert float((a.double().cpu() - b).abs().amax(dim=1).max()) < 1e-4 * float(b.abs().max())
    # the materialised route on the same inputs (exact-f32 MFMA products, softmax kernels): both sit inside the f64 tolerance
    Sp = (Skv + 7) // 8 * 8
    P = torch.empty((B * nh, Sq, Sp), dtype=torch.float32, device=DEV)
    ldq, ldk = qd.stride(0), kd.stride(0)
    ops.gemm(qd, kd, P, Sq, Skv, hd, la=0, lb=0, lda=ldq, ldb=ldk, ldc=Sp, alpha=alpha, batch=B * nh, zdiv=nh, sA=(Sq * ldq, hd), sB=(Skv * ldk, hd),
             sC=(nh * Sq * Sp, Sq * Sp))
    ops.softmax_(P, B * nh * Sq, Skv, Sp)
    o = torch.empty((B * Sq, H), dtype=torch.float32, device=DEV)
    ops.gemm(P, vd, o, Sq, hd, Skv, la=0, lb=1, lda=Sp, ldb=ldk, ldc=H, batch=B * nh, zdiv=nh, sA=(nh * Sq * Sp, Sq * Sp), sB=(Skv * ldk, hd),
             sC=(Sq * H, hd))
    assert rel_err(ctx, o.double()) < 3e-5
    # shapes the kernel does not take are refused, not mis-computed
    assert not ops.attention_x3_supported(257, 257, 64) and not ops.attention_x3_supported(256, 128, 64) and not ops.attention_x3_supported(256, 256, 48)
    assert not ops.attention_x3_supported(384, 384, 64) and not ops.attention_x3_supported(1024, 300, 64)


@pytest.mark.parametrize("B,Sq,Skv,nh,packed", [(2, 1024, 1024, 2, True), (1, 512, 512, 3, True), (2, 1024, 77, 2, False), (1, 512, 96, 2, False)])
@pytest.mark.parametrize("stream", [True, False])
def test_fused_attention_bf16x3_block_by_block(B, Sq, Skv, nh, packed, stream, monkeypatch):
    """round 6 (BASELINE config 4's 1024-token sequences in the bf16x3 mode): query rows in blocks of 256 against key blocks of 256 (or the
    <= 96 text states).  stream (the default for several key blocks): a workgroup keeps its 256 queries / keys and streams the other
    side's blocks through LDS - online softmax forward, dQ and dK / dV passes with the global log-sum-exp, every result written once.
    Otherwise (and for the text states): attention3.hip's one-tile kernels per block pair, the key blocks merged by their log-sum-exps,
    backward per block pair with the query block's GLOBAL log-sum-exp and the final context.  Same tolerances against float64 as the
    one-tile form either way."""
    ops = _ops()
    monkeypatch.setattr(ops, "X3_STREAM", stream)
    hd = 64
    H = nh * hd
    alpha = 1.0 / float(torch.sqrt(torch.tensor(hd, dtype=torch.float32)))
    assert ops.attention_x3_supported(Sq, Skv, hd) and ops.attention_x3_blocked(Sq, Skv)
    if packed:
        qkv = rnd((B * Sq, 3 * H), 170, 1.0)
        qc, kc, vc = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
        qkv_d = qkv.to(DEV)
        qd, kd, vd = qkv_d[:, :H], qkv_d[:, H:2 * H], qkv_d[:, 2 * H:]
    else:
        qc, kv = rnd((B * Sq, H), 171, 1.0), rnd((B * Skv, 2 * H), 172, 1.0)
        kc, vc = kv[:, :H], kv[:, H:]
        qd, kv_d = qc.to(DEV), kv.to(DEV)
        kd, vd = kv_d[:, :H], kv_d[:, H:]
    dctx = rnd((B * Sq, H), 173)
    q = qc.double().reshape(B, Sq, nh, hd).transpose(1, 2).detach().requires_grad_(True)
    k = kc.double().reshape(B, Skv, nh, hd).transpose(1, 2).detach().requires_grad_(True)
    v = vc.double().reshape(B, Skv, nh, hd).transpose(1, 2).detach().requires_grad_(True)
    sc = q @ k.transpose(-1, -2) * alpha
    ref = (torch.softmax(sc, dim=-1) @ v).transpose(1, 2).reshape(B * Sq, H)
    ref.backward(dctx.double())
    gq, gk, gv = (t.grad.transpose(1, 2).reshape(-1, H) for t in (q, k, v))
    ctx, lse = ops.attention_x3_fwd(qd, kd, vd, B, Sq, Skv, nh, hd, alpha)
    if packed:
        dqkv = torch.full_like(qkv_d, float("nan"))
        dq, dk, dv = ops.attention_x3_bwd(qd, kd, vd, ctx, dctx.to(DEV), lse, B, Sq, Skv, nh, hd, alpha, dq=dqkv[:, :H], dk=dqkv[:, H:2 * H], dv=dqkv[:, 2 * H:])
    else:
        dq, dk, dv = ops.attention_x3_bwd(qd, kd, vd, ctx, dctx.to(DEV), lse, B, Sq, Skv, nh, hd, alpha)
    assert ctx.dtype == torch.float32 and torch.isfinite(ctx).all() and all(torch.isfinite(t).all() for t in (dq, dk, dv))
    ef = rel_err(ctx, ref.detach())
    lref = torch.logsumexp(sc, dim=-1).reshape(B * nh, Sq // 256, 256).transpose(0, 1).detach()       # [query block, B*nh, 256]
    el = rel_err(lse, lref)
    eg = [rel_err(a, b) for a, b in ((dq, gq), (dk, gk), (dv, gv))]
    print(f"bf16x3 attention, blocks, {Sq} x {Skv}: ctx {ef:.1e}, lse {el:.1e}, dq / dk / dv {eg[0]:.1e} / {eg[1]:.1e} / {eg[2]:.1e} of float64")
    assert ef < 2e-5 and el < 2e-6 and max(eg) < 5e-5
    assert float((ctx.double().cpu() - ref.detach()).abs().amax(dim=1).max()) < 1e-4
    for a, b in ((dq, gq), (dk, gk), (dv, gv)):
        assert float((a.double().cpu() - b).abs().amax(dim=1).max()) < 1e-4 * float(b.abs().max())
    with pytest.raises(Exception):     # operand planes are the one-tile form's
        ops.attention_x3_bwd(qd, kd, vd, ctx, dctx.to(DEV), lse, B, Sq, Skv, nh, hd, alpha, planes_only=True, planes=((None, 0), (None, 0), (None, 0)))


def test_one_tile_attention_under_graph_capture():
    """the 32 x 32-block attention kernels inside a captured HIP graph (the decoding loop of generate2 captures its forward; a
    training step can be captured too): replay == eager, bit for bit, forward and backward"""
    ops = _ops()
    B, S, nh, hd = 2, 257, 4, 48
    H = nh * hd
    alpha = 1.0 / float(torch.sqrt(torch.tensor(hd, dtype=torch.float32)))
    qkv = rnd((B * S, 3 * H), 150, 1.0).to(torch.bfloat16).to(DEV)
    dctx = rnd((B * S, H), 151).to(torch.bfloat16).to(DEV)
    ctx0, lse0 = ops.attention_fwd(qkv, B, S, nh, hd, alpha)
    dqkv0 = ops.attention_bwd(qkv, ctx0, dctx, lse0, B, S, nh, hd, alpha)

    def run():
        c, l = ops.attention_fwd(qkv, B, S, nh, hd, alpha)
        return c, l, ops.attention_bwd(qkv, c, dctx, l, B, S, nh, hd, alpha)
    graph, (c1, l1, d1) = ops.capture_graph(run)
    c1.zero_(); d1.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(c1, ctx0) and torch.equal(l1[:, :S], lse0[:, :S]) and torch.equal(d1, dqkv0)


@pytest.mark.parametrize("B,Sq,Skv,nh,hd", [(2, 256, 77, 2, 64), (1, 1024, 77, 2, 64), (2, 40, 130, 2, 48), (1, 1024, 1024, 2, 64),
                                             (1, 600, 300, 1, 48), (2, 256, 256, 3, 32), (1, 1025, 1025, 1, 16), (2, 50, 7, 2, 64)])
def test_fused_attention_general_lengths(B, Sq, Skv, nh, hd):
    """separate q / k / v with their own row strides (cross-attention against 77 text tokens, seq 1024 with K/V streamed
    through LDS in 256-key tiles and an online softmax, lengths that end inside a tile) vs the f64 evaluation"""
    ops = _ops()
    H = nh * hd
    alpha = 1.0 / float(torch.sqrt(torch.tensor(hd, dtype=torch.float32)))
    qp = rnd((B * Sq, H + 8), 120, 1.0).to(torch.bfloat16)            # q rows padded: stride H + 8
    kvp = rnd((B * Skv, 2 * H), 121, 1.0).to(torch.bfloat16)          # k | v packed
    dctx = rnd((B * Sq, H), 122).to(torch.bfloat16)
    qd = qp[:, :H].double().view(B, Sq, nh, hd).transpose(1, 2).requires_grad_(True)
    kd = kvp[:, :H].double().view(B, Skv, nh, hd).transpose(1, 2).requires_grad_(True)
    vd = kvp[:, H:].double().view(B, Skv, nh, hd).transpose(1, 2).requires_grad_(True)
    sc = qd @ kd.transpose(-1, -2) * alpha
    ref = (torch.softmax(sc, dim=-1) @ vd).transpose(1, 2).reshape(B * Sq, H)
    ref.backward(dctx.double())
    qg, kvg = qp.to(DEV), kvp.to(DEV)
    ctx, lse = ops.attention_fwd_ex(qg[:, :H], kvg[:, :H], kvg[:, H:], B, Sq, Skv, nh, hd, alpha)
    assert rel_err(ctx.float(), ref.detach()) < 1.5e-2
    assert rel_err(lse[:, :Sq], torch.logsumexp(sc, dim=-1).reshape(B * nh, Sq).detach()) < 1e-3
    dkv = torch.full((B * Skv, 2 * H), 7.0, dtype=torch.bfloat16, device=DEV)
    dq, dk, dv = ops.attention_bwd_ex(qg[:, :H], kvg[:, :H], kvg[:, H:], ctx, dctx.to(DEV), lse, B, Sq, Skv, nh, hd, alpha,
                                      dk=dkv[:, :H], dv=dkv[:, H:])
    for got, want, nm in ((dq, qd.grad, "q"), (dkv[:, :H], kd.grad, "k"), (dkv[:, H:], vd.grad, "v")):
        want = want.transpose(1, 2).reshape(got.shape)
        e = rel_err(got.float(), want)
        assert e < 3e-2, (nm, e)


def test_fused_attention_extreme_scores():
    """rows whose scores are all very negative / whose maximum sits in a late K/V tile (the online-softmax rescale path), and a
    head with one dominant key: no NaN / inf, results match f64"""
    ops = _ops()
    B, S, nh, hd = 1, 600, 1, 64
    alpha = 0.125
    q = rnd((S, hd), 130, 1.0)
    k = rnd((S, hd), 131, 1.0)
    v = rnd((S, hd), 132, 1.0)
    u = torch.ones(hd) / math.sqrt(hd)
    k += 4.0 * u                              # every key shares a component along u ...
    q[5] = -200.0 * u                         # ... so row 5's scaled scores are all about -100 +- 25
    k[550] = 6.0 * q[7].sign()                # key 550 (third K/V tile) dominates row 7
    k[3] = 20.0 * q[9].sign()                 # key 3 dominates row 9 by a huge margin
    qb, kb, vb = q.to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16)
    sc = qb.double() @ kb.double().t() * alpha
    ref = torch.softmax(sc, -1) @ vb.double()
    ctx, lse = ops.attention_fwd_ex(qb.to(DEV), kb.to(DEV), vb.to(DEV), B, S, S, nh, hd, alpha)
    assert torch.isfinite(ctx.float()).all() and torch.isfinite(lse[:, :S]).all()
    assert rel_err(ctx.float(), ref) < 1.5e-2
    assert rel_err(lse[:, :S], torch.logsumexp(sc, -1).view(1, S)) < 1e-3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_split_k_wgrad(dtype):
    """weight-gradient shape: short M,N, long K, k-major operands, K cut in slices summed with f32 atomics"""
    ops = _ops()
    T, N, K = 4100, 136, 200
    dy, x = rnd((T, N), 110).to(dtype), rnd((T, K), 111).to(dtype)
    ref = dy.double().t() @ x.double()
    dw = torch.full((N, K), 3.0, device=DEV)
    ops.gemm(dy.to(DEV), x.to(DEV), dw, N, K, T, la=1, lb=1, lda=N, ldb=K, ldc=K, accumulate=True, split_k=7)
    assert rel_err(dw, ref + 3.0) < 1e-4
    dw2 = torch.empty((N, K), device=DEV)
    ops.linear_wgrad(dy.to(DEV), x.to(DEV), dw2, False)
    assert rel_err(dw2, ref) < 1e-4
    dw3 = torch.full((N, K), 3.0, device=DEV)   # workspace split-K + deterministic slice reduction, accumulating
    ops.linear_wgrad(dy.to(DEV), x.to(DEV), dw3, True)
    assert rel_err(dw3, ref + 3.0) < 1e-4
    dw4 = torch.empty((N, K), device=DEV)
    ops.linear_wgrad(dy.to(DEV), x.to(DEV), dw4, False)
    assert torch.equal(dw4, dw2)                 # run-to-run identical
    assert ops.wgrad_splits(768, 768, 16448, torch.bfloat16) > 1


@pytest.mark.parametrize("Cin,Cout,KS,ups,bias,res", [(32, 64, 3, False, False, False), (64, 32, 1, False, False, True),
                                                       (8, 40, 3, False, True, False), (32, 32, 3, True, True, False),
                                                       (128, 130, 3, False, True, True)])
def test_conv2d_split_bf16x3(Cin, Cout, KS, ups, bias, res):
    """f32 convolution as 3 bf16 MFMAs per product: error bound 2^-16 per product -> 3e-5 relative on the output"""
    ops = _ops()
    B, H, W = 2, 12, 10
    ih, iw = (H // 2, W // 2) if ups else (H, W)
    x = rnd((B, Cin, ih, iw), 170)
    w = rnd((Cout, Cin, KS, KS), 171) / math.sqrt(Cin * KS * KS)
    bvec = rnd((Cout,), 172) if bias else None
    rr = rnd((B, Cout, H, W), 173) if res else None
    xr = x.double()
    if ups:
        xr = F.interpolate(xr, scale_factor=2.0, mode="nearest")
    p = KS - 1
    ref = F.conv2d(F.pad(xr, [p // 2, p - p // 2, p // 2, p - p // 2]), w.double(), bvec.double() if bias else None)
    if res:
        ref = ref + rr.double()
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    w_hi, w_lo = ops.split_bf16(w.permute(0, 2, 3, 1).contiguous().to(DEV))
    assert rel_err(w_hi.float() + w_lo.float(), w.permute(0, 2, 3, 1)) < 2e-5
    out = ops.conv2d_nhwc_split(xn, w_hi, w_lo, B, H, W, Cin, Cout, KS, bias=bvec.to(DEV) if bias else None,
                                residual=rr.permute(0, 2, 3, 1).contiguous().to(DEV) if res else None, upsample=ups)
    assert rel_err(out.cpu().permute(0, 3, 1, 2), ref) < 3e-5, (Cin, Cout, KS, ups)


@pytest.mark.parametrize("B,H,W,Cin,Cout,bias,res", [
    (2, 12, 10, 32, 64, True, False),      # one partial pixel tile, one partial channel tile
    (1, 16, 16, 64, 36, False, True),      # exactly one pixel tile, ragged Cout
    (3, 20, 24, 128, 128, True, True),     # several pixel tiles (last ragged), Cin = 128 -> 36 K-tiles (6 groups)
    (2, 8, 8, 96, 260, True, False),       # Cin not a power of two, 27 K-tiles (padded to 30), three channel tiles
    (1, 32, 32, 256, 256, False, True),    # fused GroupNorm statistics: 8 channels per group, 4 pixel tiles
    (2, 16, 16, 128, 128, True, True),     # ... 4 channels per group, one tile per image
    (1, 16, 32, 64, 512, True, False),     # ... 16 channels per group, 4 channel tiles
    (2, 48, 32, 128, 128, True, True),     # patch-slab kernel: 6 patches per image (interior + every border kind), 4 channel chunks
    (1, 64, 64, 192, 256, False, True),    # ... 16 patches, 6 channel chunks (3 loop iterations), two channel tiles
    (3, 16, 16, 512, 64, True, False),     # ... one patch per image (all halo rows are padding), 16 channel chunks, ragged Cout tile
])
def test_conv2d_split2_dma_matches_split(B, H, W, Cin, Cout, bias, res):
    """the LDS-DMA bf16x3 convolution on pre-split planes computes the same products as the register-staged bf16x3 kernel on
    the f32 tensor the planes came from: BIT-identical where the K order is the same (tap-major kernel), within f32
    summation-order noise for the patch-slab kernel (K order chunk, tap, channel; ops.conv_slab_ok); and the planes GroupNorm
    writes are the split of the tensor GroupNorm writes up to the SiLU reciprocal's last bit"""
    ops = _ops()
    x = rnd((B, H, W, Cin), 180).to(DEV)
    w = (rnd((Cout, 3, 3, Cin), 181) / math.sqrt(9 * Cin)).to(DEV)
    w_hi, w_lo = ops.split_bf16(w)
    x_hi, x_lo = ops.split_bf16(x)
    bvec = rnd((Cout,), 182).to(DEV) if bias else None
    rr = rnd((B, H, W, Cout), 183).to(DEV) if res else None
    assert ops.conv_split2_ok(B, H, W, Cin, Cout, 3)
    ref = ops.conv2d_nhwc_split(x, w_hi, w_lo, B, H, W, Cin, Cout, 3, bias=bvec, residual=rr)
    got = ops.conv2d_nhwc_split2(x_hi, x_lo, w_hi, w_lo, B, H, W, Cin, Cout, bias=bvec, residual=rr)
    slab = ops.conv_slab_ok(H, W, Cin)
    if slab:
        assert rel_err(got, ref) < 2e-6, rel_err(got, ref)   # same 3 x 9 x Cin products per output, f32 accumulation in another order
    else:
        assert torch.equal(got, ref), float((got - ref).abs().max())
    ref64 = F.conv2d(x.cpu().double().permute(0, 3, 1, 2), w.cpu().double().permute(0, 3, 1, 2), bvec.cpu().double() if bias else None, padding=1)
    if res:
        ref64 = ref64 + rr.cpu().double().permute(0, 3, 1, 2)
    assert rel_err(got.cpu().permute(0, 3, 1, 2), ref64) < 3e-5
    # GroupNorm statistics fused into the convolution epilogue: [B, HW/256, 32, 2] f64 sums of the (bias + residual) output
    if ops.conv_gn_stats_ok(H, W, Cout, 32):
        got2 = ops.conv2d_nhwc_split2(x_hi, x_lo, w_hi, w_lo, B, H, W, Cin, Cout, bias=bvec, residual=rr, gn_groups=32)
        assert torch.equal(got2, got)
        part, nchunk = got2._gn_stats
        assert nchunk == H * W // 256
        st = part.view(B, nchunk, 32, 2).sum(1).cpu()
        o = got.cpu().double().view(B, H * W, 32, Cout // 32)
        assert rel_err(st[..., 0], o.sum((1, 3))) < 1e-12 and rel_err(st[..., 1], (o * o).sum((1, 3))) < 1e-12
        gam2, bet2 = (1 + 0.1 * rnd((Cout,), 186)).to(DEV), (0.1 * rnd((Cout,), 187)).to(DEV)
        if 256 % (Cout // 4) == 0:
            a_hi, a_lo = ops.groupnorm_silu_nhwc_split(got, gam2, bet2, B, H * W, Cout)
            b_hi, b_lo = ops.groupnorm_silu_nhwc_split(got2, gam2, bet2, B, H * W, Cout, stats=got2._gn_stats)
            ya, yb = a_hi.float() + a_lo.float(), b_hi.float() + b_lo.float()
            assert rel_err(yb, ya) < 1e-6   # same statistics up to f64 summation order
    # GroupNorm + SiLU with split output == split of the f32 output
    if 256 % (Cin // 4):
        return   # (channel counts the GroupNorm kernel does not take)
    gam, bet = (1 + 0.1 * rnd((Cin,), 184)).to(DEV), (0.1 * rnd((Cin,), 185)).to(DEV)
    y = ops.groupnorm_silu_nhwc(x, gam, bet, B, H * W, Cin)
    y_hi, y_lo = ops.groupnorm_silu_nhwc_split(x, gam, bet, B, H * W, Cin)
    # (the planes carry the hardware-reciprocal SiLU of the fused convolution - bit-identical to that route, see
    #  test_conv_with_fused_groupnorm_input_is_bit_identical - while a tensor OUTPUT keeps the correctly rounded division of the
    #  exact-f32 parity mode: the two agree to an f32 ulp before the split, so hi is the same bf16 except at a rounding boundary and
    #  hi + lo reproduces the tensor to the split's 2^-16)
    e_hi, e_lo = ops.split_bf16(y)
    hi_diff = float((y_hi.view(torch.int16) != e_hi.view(torch.int16)).float().mean())
    assert hi_diff < 5e-3, hi_diff
    assert float(((y_hi.float() + y_lo.float()) - y).abs().max()) <= 2.0 ** -15 * float(y.abs().max())
    ulps = ((y_hi.float() + y_lo.float()) - (e_hi.float() + e_lo.float())).abs() / (y.abs() * 2.0 ** -16 + 1e-30)
    assert float(ulps.max()) <= 2.0, float(ulps.max())      # the two reconstructions: within two units of the planes' last place


@pytest.mark.parametrize("mode", ["f32", "bf16x3", "bf16"])
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 8, 8, 32, 32), (1, 5, 7, 64, 40), (3, 16, 12, 128, 128), (1, 32, 32, 24, 256)])
def test_conv2d_stride2_downsample(mode, B, H, W, Cin, Cout):
    """upsample = 2: the taming Downsample (muse/modeling_taming_vqgan.py:55-59) = F.pad(x, (0,1,0,1)) + Conv2d(3, stride 2,
    padding 0), gathered straight from the unpadded [2H, 2W] input (bottom / right taps of the last row / column read zeros);
    H, W are the OUTPUT dims; odd / ragged tile shapes, bias and residual included"""
    ops = _ops()
    x = rnd((B, Cin, 2 * H, 2 * W), 210)
    w = rnd((Cout, Cin, 3, 3), 211) / math.sqrt(9 * Cin)
    bvec, rr = rnd((Cout,), 212), rnd((B, Cout, H, W), 213)
    ref = F.conv2d(F.pad(x.double(), (0, 1, 0, 1)), w.double(), bvec.double(), stride=2) + rr.double()
    xn, wn = x.permute(0, 2, 3, 1).contiguous().to(DEV), w.permute(0, 2, 3, 1).contiguous().to(DEV)
    rn = rr.permute(0, 2, 3, 1).contiguous().to(DEV)
    if mode == "bf16x3":
        w_hi, w_lo = ops.split_bf16(wn)
        out = ops.conv2d_nhwc_split(xn, w_hi, w_lo, B, H, W, Cin, Cout, 3, bias=bvec.to(DEV), residual=rn, upsample=2)
        tol = 3e-5
    elif mode == "f32":
        out = ops.conv2d_nhwc(xn, wn, B, H, W, Cin, Cout, 3, bias=bvec.to(DEV), residual=rn, upsample=2)
        tol = 2e-6
    else:
        out = ops.conv2d_nhwc(ops.cast_to_bf16(xn), ops.cast_to_bf16(wn), B, H, W, Cin, Cout, 3, bias=bvec.to(DEV),
                              residual=ops.cast_to_bf16(rn), upsample=2).float()
        tol = 2e-2
    assert tuple(out.shape) == (B, H, W, Cout)
    assert rel_err(out.cpu().permute(0, 3, 1, 2), ref) < tol, (mode, B, H, W, Cin, Cout)


@pytest.mark.parametrize("B,H,W,Cin,Cout,KS,res,ups", [
    (2, 16, 16, 8, 128, 3, False, False),    # conv_in shape: 3 (padded to 8) -> 128, 4 channels per group
    (3, 16, 8, 128, 256, 1, True, False),    # nin_shortcut: 1x1, residual, 8 channels per group, one tile per image
    (1, 32, 16, 64, 512, 3, False, True),    # upsample conv: 16 channels per group, 4 pixel x 4 channel tiles
    (2, 16, 32, 32, 1024, 1, True, False),   # 32 channels per group
])
def test_conv2d_split_fused_groupnorm_stats(B, H, W, Cin, Cout, KS, res, ups):
    """muse_conv2d_nhwc_split with gn_partial: same output bits, and [B, HW/128, 32, 2] f64 sums of that output"""
    ops = _ops()
    hin, win = (H // 2, W // 2) if ups else (H, W)
    x = rnd((B, hin, win, Cin), 190).to(DEV)
    w_hi, w_lo = ops.split_bf16((rnd((Cout, KS, KS, Cin), 191) / math.sqrt(KS * KS * Cin)).to(DEV))
    bvec = rnd((Cout,), 192).to(DEV)
    rr = rnd((B, H, W, Cout), 193).to(DEV) if res else None
    ref = ops.conv2d_nhwc_split(x, w_hi, w_lo, B, H, W, Cin, Cout, KS, bias=bvec, residual=rr, upsample=ups)
    got = ops.conv2d_nhwc_split(x, w_hi, w_lo, B, H, W, Cin, Cout, KS, bias=bvec, residual=rr, upsample=ups, gn_groups=32)
    assert torch.equal(got, ref)
    part, nchunk = got._gn_stats
    assert nchunk == H * W // 128
    st = part.view(B, nchunk, 32, 2).sum(1).cpu()
    o = ref.cpu().double().view(B, H * W, 32, Cout // 32)
    assert rel_err(st[..., 0], o.sum((1, 3))) < 1e-12 and rel_err(st[..., 1], (o * o).sum((1, 3))) < 1e-12
    if 256 % (Cout // 4) == 0:
        gam, bet = (1 + 0.1 * rnd((Cout,), 194)).to(DEV), (0.1 * rnd((Cout,), 195)).to(DEV)
        a_hi, a_lo = ops.groupnorm_silu_nhwc_split(ref, gam, bet, B, H * W, Cout)
        b_hi, b_lo = ops.groupnorm_silu_nhwc_split(got, gam, bet, B, H * W, Cout, stats=got._gn_stats)
        assert rel_err(b_hi.float() + b_lo.float(), a_hi.float() + a_lo.float()) < 1e-6
    # shapes the fused statistics do not take fall back to a plain call (no _gn_stats attribute)
    odd = ops.conv2d_nhwc_split(x[:, :6].contiguous(), w_hi, w_lo, B, 12 if ups else 6, W, Cin, Cout, KS, bias=bvec, upsample=ups, gn_groups=32)
    assert not hasattr(odd, "_gn_stats")


@pytest.mark.parametrize("B,H,W,C", [(2, 64, 64, 128), (3, 16, 24, 256), (1, 8, 8, 512), (2, 90, 46, 128)])
def test_avgpool_fused_groupnorm_stats(B, H, W, C):
    """muse_avgpool2x2_nhwc_stats: the pooled tensor is bit-identical to muse_avgpool2x2_nhwc and the partials are the f64
    group sums of it (several 1024-pixel chunks per image in the first case, a ragged last chunk in the last)"""
    ops = _ops()
    x = rnd((B, H, W, C), 196).to(DEV)
    ref = ops.avgpool2x2_nhwc(x, B, H, W, C)
    got = ops.avgpool2x2_nhwc(x, B, H, W, C, gn_groups=32)
    assert torch.equal(got, ref)
    part, nchunk = got._gn_stats
    ohw = (H // 2) * (W // 2)
    assert nchunk == (ohw + 1023) // 1024
    st = part.view(B, nchunk, 32, 2).sum(1).cpu()
    o = ref.cpu().double().view(B, ohw, 32, C // 32)
    assert rel_err(st[..., 0], o.sum((1, 3))) < 1e-12 and rel_err(st[..., 1], (o * o).sum((1, 3))) < 1e-12
    gam, bet = (1 + 0.1 * rnd((C,), 197)).to(DEV), (0.1 * rnd((C,), 198)).to(DEV)
    a_hi, a_lo = ops.groupnorm_silu_nhwc_split(ref, gam, bet, B, ohw, C)
    b_hi, b_lo = ops.groupnorm_silu_nhwc_split(got, gam, bet, B, ohw, C, stats=got._gn_stats)
    assert rel_err(b_hi.float() + b_lo.float(), a_hi.float() + a_lo.float()) < 1e-6


@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("la,lb", [(0, 0), (0, 1), (1, 1), (1, 0)])
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (520, 264, 200), (1000, 520, 712), (264, 776, 64)])
def test_gemm_256_tile_kernel(monkeypatch, out_dtype, la, lb, M, N, K):
    """the 256x256 LDS-DMA kernel (gemm256.h), forced with MUSE_GEMM256=1: every operand layout, ragged M / N / K (tail
    K-tile, odd tile counts), every epilogue option it handles (alpha, bias, rowvec, residual, accumulate); GELU falls back
    to the 128x128 kernel and must agree too"""
    ops = _ops()
    monkeypatch.setenv("MUSE_GEMM256", "1")
    A, B = rnd((M, K), 201).to(torch.bfloat16), rnd((N, K), 202).to(torch.bfloat16)
    ref = A.double() @ B.double().t()
    Ad = (A if la == 0 else A.t().contiguous()).to(DEV)
    Bd = (B if lb == 0 else B.t().contiguous()).to(DEV)
    lda, ldb = (K if la == 0 else M), (K if lb == 0 else N)
    d = ops.GemmDesc()
    d.A, d.B, d.C = Ad.data_ptr(), Bd.data_ptr(), Ad.data_ptr()
    d.dtype, d.out_dtype, d.layout_a, d.layout_b = 1, (1 if out_dtype == torch.bfloat16 else 0), la, lb
    d.M, d.N, d.K, d.batch, d.zdiv, d.lda, d.ldb, d.ldc, d.alpha = M, N, K, 1, 1, lda, ldb, N, 1.0
    assert ops.lib().muse_gemm_tile(ops.C.byref(d)) == 256
    C = torch.empty((M, N), dtype=out_dtype, device=DEV)
    ops.gemm(Ad, Bd, C, M, N, K, la=la, lb=lb, lda=lda, ldb=ldb, ldc=N, alpha=0.5)
    tol = 1e-2 if out_dtype == torch.bfloat16 else 2e-5 * math.sqrt(K)
    assert rel_err(C.float(), 0.5 * ref) < tol
    bias, rowvec = rnd((N,), 203).to(DEV), rnd((M,), 205).to(DEV)
    res = rnd((M, N), 204).to(DEV, out_dtype)
    C2 = torch.ones((M, N), dtype=out_dtype, device=DEV)
    ops.gemm(Ad, Bd, C2, M, N, K, la=la, lb=lb, lda=lda, ldb=ldb, ldc=N, bias=bias, rowvec=rowvec, residual=res, ldr=N,
             accumulate=True)
    ref2 = ref + bias.cpu().double() + rowvec.cpu().double()[:, None] + res.cpu().double() + 1.0
    assert rel_err(C2.float(), ref2) < (2e-2 if out_dtype == torch.bfloat16 else 1e-4)
    C3 = torch.empty((M, N), dtype=out_dtype, device=DEV)
    ops.gemm(Ad, Bd, C3, M, N, K, la=la, lb=lb, lda=lda, ldb=ldb, ldc=N, bias=bias, act=1)
    assert rel_err(C3.float(), F.gelu(ref + bias.cpu().double())) < (2e-2 if out_dtype == torch.bfloat16 else 1e-4)
    if out_dtype == torch.float32:   # split-K through the workspace, slices reduced on the host
        ws = torch.full((3, M, N), float("nan"), device=DEV)
        ops.gemm(Ad, Bd, ws, M, N, K, la=la, lb=lb, lda=lda, ldb=ldb, ldc=N, split_k=3, split_stride=M * N)
        nk = (K + 63) // 64
        per = (nk + 2) // 3
        used = (nk + per - 1) // per
        assert rel_err(ws[:used].sum(0), ref) < tol


def test_gemm_256_matches_128_on_model_shapes(monkeypatch):
    """forward / dX / dW products of one transformer layer at T = 1028 tokens: the 128x128 kernel and both pipelines of the
    256x256 kernel (MUSE_G256_BK = 64: two stages of 64-wide K-tiles, 32: five stages of 32-wide ones) agree to bf16 rounding"""
    ops = _ops()
    T_, H, I = 1028, 768, 3072
    x = rnd((T_, H), 210).to(DEV, torch.bfloat16)
    w = (0.05 * rnd((2 * I, H), 211)).to(DEV, torch.bfloat16)
    dy = rnd((T_, 2 * I), 212).to(DEV, torch.bfloat16)
    outs = {}
    for mode in ("0", "1", "1/32"):
        monkeypatch.setenv("MUSE_GEMM256", mode[0])
        monkeypatch.setenv("MUSE_G256_BK", "32" if mode.endswith("/32") else "64")
        ops._WGRAD_PLAN.clear()
        y = ops.linear(x, w)
        dx = ops.linear_dgrad(dy, w)
        dw = torch.zeros((2 * I, H), device=DEV)
        ops.linear_wgrad(dy, x, dw, False)
        outs[mode] = (y.float(), dx.float(), dw)
    ops._WGRAD_PLAN.clear()
    for other in ("1", "1/32"):
        for a, b in zip(outs["0"], outs[other]):
            assert rel_err(a, b) < 1e-2
    for a, b in zip(outs["1"], outs["1/32"]):   # same products, same accumulation order per output element
        assert torch.equal(a, b)


@pytest.mark.parametrize("la,lb", [(0, 0), (0, 1), (1, 1), (1, 0)])
@pytest.mark.parametrize("M,N,K", [(520, 264, 200), (4360, 4104, 264), (16448, 776, 328)])
def test_gemm_256_persistent_matches_launch_per_tile(monkeypatch, la, lb, M, N, K):
    """the persistent tile-walking 256x256 kernel (csrc/gemm256p.h; MUSE_G256P=1, the default) against the launch-per-tile kernel
    (MUSE_G256P=0): same products in the same order, so bf16 outputs must be BIT-identical - on a single-round grid, on a grid with
    more tiles than CUs (18 x 17 = 306 tiles: per-XCD ticket queues, tile changes inside the K pipeline, the register epilogue
    with its lane exchanges) and on the train step's ragged M = 64 x 257 (row strip with skipped MFMA groups); f32 outputs that
    start the accumulators from a residual / the old C differ by the order of one addition; repeated launches are bit-identical
    (race screen), and the f64 reference is met"""
    ops = _ops()
    monkeypatch.setenv("MUSE_GEMM256", "1")
    A, B = rnd((M, K), 301).to(torch.bfloat16), rnd((N, K), 302).to(torch.bfloat16)
    Ad = (A if la == 0 else A.t().contiguous()).to(DEV)
    Bd = (B if lb == 0 else B.t().contiguous()).to(DEV)
    lda, ldb = (K if la == 0 else M), (K if lb == 0 else N)
    res = rnd((M, N), 303).to(DEV)
    ref = (Ad.double() if la == 0 else Ad.double().t()) @ (Bd.double().t() if lb == 0 else Bd.double())

    def run(persistent, out_dtype, **kw):
        monkeypatch.setenv("MUSE_G256P", "1" if persistent else "0")
        C = torch.full((M, N), 7.0, dtype=out_dtype, device=DEV)
        ops.gemm(Ad, Bd, C, M, N, K, la=la, lb=lb, lda=lda, ldb=ldb, ldc=N, **kw)
        return C
    old_b, new_b = run(False, torch.bfloat16), run(True, torch.bfloat16)
    assert torch.equal(old_b.view(torch.int16), new_b.view(torch.int16))
    assert rel_err(new_b.float(), ref) < 1e-2
    for _ in range(6):
        assert torch.equal(run(True, torch.bfloat16).view(torch.int16), new_b.view(torch.int16))
    old_s, new_s = run(False, torch.bfloat16, alpha=0.25), run(True, torch.bfloat16, alpha=0.25)
    assert torch.equal(old_s.view(torch.int16), new_s.view(torch.int16))
    old_f, new_f = run(False, torch.float32), run(True, torch.float32)
    assert torch.equal(old_f, new_f)
    tol = 2e-5 * math.sqrt(K)
    new_r = run(True, torch.float32, residual=res, ldr=N)
    assert rel_err(new_r, ref + res.double()) < tol and rel_err(new_r, run(False, torch.float32, residual=res, ldr=N)) < 1e-6
    new_a = run(True, torch.float32, accumulate=True)
    assert rel_err(new_a, ref + 7.0) < tol
    for _ in range(3):
        assert torch.equal(run(True, torch.float32, residual=res, ldr=N), new_r)


@pytest.mark.parametrize("rows,cols", [(37, 768), (130, 512), (5, 64), (64, 1024)])
def test_layernorm_pair_matches_two_calls(rows, cols):
    """the fused NormFormer LayerNorm pair (x1 = x + LN(ao) w_post ; ln2 = LN(x1) w_pre, and its backward) against the two
    muse_layernorm_fwd / _bwd calls it replaces: same operations in the same order, results equal up to fma contraction"""
    ops = _ops()
    eps = 1e-5
    ao = rnd((rows, cols), 320).to(torch.bfloat16).to(DEV)
    x = rnd((rows, cols), 321).to(DEV)
    w_post, w_pre = (1 + 0.1 * rnd((cols,), 322)).to(DEV), (1 + 0.1 * rnd((cols,), 323)).to(DEV)
    x1r, mu_p, rs_p = ops.layernorm_fwd(ao, w_post, eps, torch.float32, residual=x)
    ln2r, mu2, rs2 = ops.layernorm_fwd(x1r, w_pre, eps, torch.bfloat16)
    x1, mp, rp, ln2, m2, r2 = ops.layernorm_pair_fwd(ao, x, w_post, w_pre, eps)
    assert torch.equal(mp, mu_p) and torch.equal(rp, rs_p), "statistics of the first LayerNorm"
    d1 = float((x1 - x1r).abs().max())
    assert d1 <= 1e-6 * float(x1r.abs().max()), d1           # same operations; at most the last bit (fma contraction) apart
    assert rel_err(m2, mu2) < 1e-6 and rel_err(r2, rs2) < 1e-6
    assert rel_err(ln2.float(), ln2r.float()) < 8e-3          # bf16 outputs: at most one rounding step apart
    print(f"ln pair fwd {rows}x{cols}: max |x1 - x1_ref| = {d1:.3e}, x1 bit-equal {torch.equal(x1, x1r)}, ln2 bit-equal {torch.equal(ln2.view(torch.int16), ln2r.view(torch.int16))}")
    if cols > 768:
        return   # (the backward pair is built for hidden sizes up to 768)
    dln2 = rnd((rows, cols), 324).to(torch.bfloat16).to(DEV)
    dres = rnd((rows, cols), 325).to(DEV)
    dw_pre_r, dw_post_r = torch.zeros(cols, device=DEV), torch.zeros(cols, device=DEV)
    dx1r = ops.layernorm_bwd(dln2, x1r, w_pre, mu2, rs2, torch.float32, dw_pre_r, False, dres=dres)
    daor = ops.layernorm_bwd(dx1r, ao, w_post, mu_p, rs_p, torch.bfloat16, dw_post_r, False)
    dw_pre, dw_post = torch.full((cols,), 7.0, device=DEV), torch.full((cols,), 7.0, device=DEV)
    dx1, dao = ops.layernorm_pair_bwd(dln2, x1, w_pre, m2, r2, dres, ao, w_post, mp, rp, dw_pre, False, dw_post, False)
    assert rel_err(dx1, dx1r) < 1e-6 and rel_err(dao.float(), daor.float()) < 1e-2     # (fma contraction may differ in the last bit)
    assert rel_err(dw_pre, dw_pre_r) < 1e-5 and rel_err(dw_post, dw_post_r) < 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,inter", [(37, 64), (130, 3072), (50, 160), (20, 2048), (7, 1024), (33, 4096)])
def test_ffn_mid_fused(dtype, rows, inter):
    """fused GLU + mid-LayerNorm forward/backward vs F.gelu(a)*b -> F.layer_norm in f64"""
    ops = _ops()
    ab = rnd((rows, 2 * inter), 300).to(dtype)
    w = 1 + 0.1 * rnd((inter,), 301)
    eps = 1e-6
    a = ab.double()[:, :inter].requires_grad_(True)
    b = ab.double()[:, inter:].requires_grad_(True)
    wr = w.double().requires_grad_(True)
    h_ref = F.gelu(a) * b
    hm_ref = F.layer_norm(h_ref, (inter,), wr, None, eps)
    h, hm, mean, rstd = ops.ffn_mid_fwd(ab.to(DEV), w.to(DEV), eps)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert rel_err(h.float(), h_ref.detach()) < (1e-6 if dtype == torch.float32 else 1e-2)
    assert rel_err(hm.float(), hm_ref.detach()) < tol
    dhm = rnd((rows, inter), 302).to(dtype)
    hm_ref.backward(dhm.double())
    dw = torch.empty(inter, device=DEV)
    dab = ops.ffn_mid_bwd(dhm.to(DEV), h, ab.to(DEV), w.to(DEV), mean, rstd, dw, False)
    assert rel_err(dab.float(), torch.cat([a.grad, b.grad], 1)) < (5e-5 if dtype == torch.float32 else 3e-2)
    assert rel_err(dw, wr.grad) < (1e-4 if dtype == torch.float32 else 3e-2)
    # h not kept: the backward recomputes it from ab
    h0, hm0, mean0, rstd0 = ops.ffn_mid_fwd(ab.to(DEV), w.to(DEV), eps, keep_h=False)
    assert h0 is None and torch.equal(hm0, hm) and torch.equal(mean0, mean) and torch.equal(rstd0, rstd)
    dw0 = torch.empty(inter, device=DEV)
    dab0 = ops.ffn_mid_bwd(dhm.to(DEV), None, ab.to(DEV), w.to(DEV), mean, rstd, dw0, False)
    if dtype == torch.bfloat16:   # (the mode the model uses it in: h is rounded to bf16 on both paths -> the same bits)
        assert torch.equal(dab0, dab) and torch.equal(dw0, dw)
    else:                         # f32: without the store the compiler fuses g * b - mean into one fma: last-bit differences
        assert rel_err(dab0, dab) < 1e-5 and rel_err(dw0, dw) < 1e-5


@pytest.mark.parametrize("rows,cols,ld", [(1, 8, 8), (130, 40, 48), (16448, 768, 768), (4112, 2304, 2304), (257, 33, 40)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bias_grad_and_add_rowvec(rows, cols, ld, dtype):
    """the two kernels behind `use_bias` (muse/modeling_transformer.py:130, :170-176): d(bias) = sum over rows (two fixed-order
    stages; f32 and bf16 dy, padded row stride, ragged last chunk) and the in-place LayerNorm bias add.  f32 sums of <= 16448 terms
    against float64: 1e-5 relative to the largest column sum of |dy|."""
    ops = _ops()
    dy = rnd((rows, ld), 77 + rows, 1.0).to(dtype)
    d = dy.to(DEV)
    got = ops.bias_grad(d, cols=cols)
    ref = dy.double()[:, :cols].sum(0)
    scale = float(dy.double()[:, :cols].abs().sum(0).max())
    assert got.dtype == torch.float32 and got.shape == (cols,)
    assert float((got.double().cpu() - ref).abs().max()) <= 1e-5 * scale
    assert torch.equal(got, ops.bias_grad(d, cols=cols))                      # fixed summation order
    if dtype == torch.float32 and ld == cols:
        b = rnd((cols,), 5)
        x = dy.clone().to(DEV)
        ops.add_rowvec_(x, b.to(DEV))
        assert torch.equal(x.cpu(), dy + b)                                  # one f32 add per element: exact


@pytest.mark.parametrize("B,H,W,Cin,Cout,res,gn", [(2, 32, 32, 128, 128, True, True), (1, 16, 16, 512, 512, False, True),
                                                   (3, 16, 48, 64, 132, True, False), (1, 64, 32, 256, 128, False, True),
                                                   (2, 16, 16, 128, 256, False, False)])
def test_conv_with_fused_groupnorm_input_is_bit_identical(B, H, W, Cin, Cout, res, gn):
    """muse_conv2d_nhwc_gn_split2 (GroupNorm + SiLU + hi/lo split applied while the patch-slab convolution stages its input)
    against the two-kernel route it replaces (muse_groupnorm_silu_nhwc_split -> muse_conv2d_nhwc_split2): same bits in the output
    and in the GroupNorm partial sums of the output, with image-border patches (zero padding AFTER the activation), several
    channel chunks, a ragged Cout tile, residual and bias"""
    ops = _ops()
    x = rnd((B, H, W, Cin), 1, 1.5).to(DEV)
    gamma, beta = (1.0 + 0.2 * rnd((Cin,), 2)).to(DEV), (0.3 * rnd((Cin,), 3)).to(DEV)
    w = rnd((Cout, 3, 3, Cin), 4, 1.0 / math.sqrt(9 * Cin)).to(DEV)
    w_hi, w_lo = ops.split_bf16(w.contiguous())
    bias = rnd((Cout,), 5, 0.1).to(DEV)
    resid = rnd((B, H, W, Cout), 6).to(DEV) if res else None
    assert ops.conv_gn_split2_ok(B, H, W, Cin, Cout, 3)
    # statistics of x the way a producer leaves them: run the apply pass once without stats and keep its partial sums
    nchunk = _hip_lib().muse_groupnorm_nchunk(H * W)
    part = torch.empty(B * nchunk * 32 * 2, dtype=torch.float64, device=DEV)
    hi = torch.empty(x.shape, dtype=torch.bfloat16, device=DEV)
    lo = torch.empty_like(hi)
    from muse.ops import check, stream
    check(_hip_lib().muse_groupnorm_silu_nhwc_split(x.data_ptr(), hi.data_ptr(), lo.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                                    part.data_ptr(), 0, B, H * W, Cin, 32, 1e-6, 1, stream()), "gn split")
    ref = ops.conv2d_nhwc_split2(hi, lo, w_hi, w_lo, B, H, W, Cin, Cout, bias=bias, residual=resid, gn_groups=32 if gn else 0)
    sc, sh = ops.groupnorm_scale_shift((part, nchunk), gamma, beta, B, H * W, Cin)
    got = ops.conv2d_nhwc_gn_split2(x, sc, sh, w_hi, w_lo, B, H, W, Cin, Cout, bias=bias, residual=resid, gn_groups=32 if gn else 0)
    assert torch.equal(got, ref)
    if gn and hasattr(ref, "_gn_stats"):
        assert torch.equal(got._gn_stats[0], ref._gn_stats[0]) and got._gn_stats[1] == ref._gn_stats[1]
    # and against the plain definition (f32 GroupNorm + SiLU + conv on the CPU; bf16x3 products: 2e-5 of the output scale)
    xn = F.silu(F.group_norm(x.cpu().permute(0, 3, 1, 2), 32, gamma.cpu(), beta.cpu(), 1e-6))
    y = F.conv2d(xn, w.cpu().permute(0, 3, 1, 2), bias.cpu(), padding=1).permute(0, 2, 3, 1)
    if res:
        y = y + resid.cpu()
    assert rel_err(got, y) < 5e-5


@pytest.mark.parametrize("env", [{"MUSE_CONV_PERSIST_GRID": "3"}, {"MUSE_CONV_PERSIST_GRID": "8"}, {"MUSE_CONV_PERSIST_TILES": "2", "MUSE_CONV_PERSIST_MIN": "0"}])
def test_persistent_fused_convolution_is_bit_identical(env):
    """conv_slab_persist_kernel<true> (round 6; the default whenever a tokenizer / decoder pass has the chip to itself and >= 2 tiles per CU - muse.TrainStep switches it off around the pass it enqueues beside a step): the parity cases of
    test_conv_with_fused_groupnorm_input_is_bit_identical with the persistent kernel taking every shape it can - workgroups that walk several
    tiles across image and N-tile changes (grid 3), one tile each (grid 8), k-tile workgroups.  The library reads its switches once per
    process, so the cases run in a child interpreter."""
    import subprocess
    import sys
    e = dict(os.environ, MUSE_CONV_PERSIST="1", MUSE_CONV_PERSIST_MIN="0")
    e.update(env)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-p", "no:cacheprovider", "-k",
                        "test_conv_with_fused_groupnorm_input_is_bit_identical"], capture_output=True, text=True, timeout=600, env=e)
    assert r.returncode == 0 and "5 passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def _hip_lib():
    from muse import _hip
    return _hip.lib()


@pytest.mark.parametrize("rows,inter", [(7, 8), (33, 1000), (300, 4096), (1030, 3072)])
def test_glu_bf16_wide_kernels_match_the_f32_kernels_rounded(rows, inter):
    """the 16-byte-per-thread bf16 GLU kernels (inter % 8 == 0) against the generic f32 kernels on the same bf16-valued inputs,
    rounded to bf16.  The bf16 kernels evaluate erf with the branch-free form of common.h (|error| <= 5e-7 in f32, four orders
    below a bf16 ulp), the f32 kernels with the library erff: the outputs agree except where that difference straddles a bf16
    rounding boundary - never more than one bf16 ulp, well under one element in a thousand"""
    ops = _ops()
    ab = rnd((rows, 2 * inter), 31, 1.5).to(torch.bfloat16).to(DEV)
    dh = rnd((rows, inter), 32).to(torch.bfloat16).to(DEV)
    for got, ref in ((ops.glu_fwd(ab), ops.glu_fwd(ab.float())), (ops.glu_bwd(ab, dh), ops.glu_bwd(ab.float(), dh.float()))):
        assert got.dtype == torch.bfloat16 and got.shape == ref.shape
        refb = ref.to(torch.bfloat16)
        ne = got != refb
        assert float(ne.float().mean()) <= 2e-3
        # never more than one bf16 ulp, plus the approximation's absolute error (5e-7 |a|, visible only where gelu(a) itself is ~0)
        assert bool(((got.float() - ref).abs() <= ref.abs() * 2.0 ** -7 + 4e-6 * float(ref.abs().max())).all())


@pytest.mark.parametrize("B,H,W,Cin,Cpad,Cout", [(2, 16, 16, 3, 8, 128), (1, 5, 7, 3, 4, 32), (3, 8, 4, 1, 8, 64), (1, 32, 32, 4, 4, 256),
                                                 (1, 6, 300, 2, 4, 128)])
def test_conv_in_direct(B, H, W, Cin, Cpad, Cout):
    """the direct exact-f32 image-to-features convolution (muse_conv_in_direct) against F.conv2d in float64 (f32 fma chains of <= 36
    terms: 2e-6), with the GroupNorm partial sums of its output (one chunk per image row; f64 sums of the f32 outputs) when the
    groups are four channels wide"""
    ops = _ops()
    x = torch.full((B, H, W, Cpad), 7.0)                      # (channels past Cin are never used)
    x[..., :Cin] = rnd((B, H, W, Cin), 41)
    w = rnd((Cout, Cin, 3, 3), 42, 0.3)
    bias = rnd((Cout,), 43, 0.1)
    w4 = torch.zeros(Cout, 9, 4)
    w4[:, :, :Cin] = w.permute(0, 2, 3, 1).reshape(Cout, 9, Cin)
    assert ops.conv_in_direct_ok(Cin, Cout, 3, Cpad)
    groups = Cout // 4 if Cout // 4 in (32, 64) else 0
    got = ops.conv_in_direct(x.to(DEV), w4.to(DEV), B, H, W, Cin, Cpad, Cout, bias=bias.to(DEV), gn_groups=groups)
    ref = F.conv2d(x[..., :Cin].permute(0, 3, 1, 2).double(), w.double(), bias.double(), padding=1).permute(0, 2, 3, 1)
    assert rel_err(got, ref) < 2e-6
    if groups:
        part, nchunk = got._gn_stats
        assert nchunk == H
        p = part.view(B, H, groups, 2).cpu()
        g = got.cpu().double().view(B, H, W, groups, 4)
        assert float((p[..., 0] - g.sum((2, 4))).abs().max()) < 1e-9 and float((p[..., 1] - (g * g).sum((2, 4))).abs().max()) < 1e-9
    else:
        assert not hasattr(got, "_gn_stats")


@pytest.mark.parametrize("B,H,W,C,Cout", [(2, 32, 48, 64, 3), (1, 16, 16, 128, 3), (3, 16, 32, 32, 4), (1, 48, 16, 96, 1)])
def test_conv_out_direct(B, H, W, C, Cout):
    """the direct exact-f32 features-to-image convolution with its GroupNorm + SiLU applied on the way in (muse_conv_out_direct:
    Decoder.norm_out -> swish -> conv_out, muse/modeling_maskgit_vqgan.py:236-240) against silu(x * scale + shift) -> F.conv2d in
    float64: f32 fma chains of 9 * C terms; zero padding AFTER the activation (a border pixel must not see silu(shift))"""
    ops = _ops()
    x = rnd((B, H, W, C), 500)
    sc, sh = rnd((B, C), 501, 0.5) + 1.0, rnd((B, C), 502, 0.5)
    w = rnd((Cout, 3, 3, C), 503, 0.1)
    bias = rnd((Cout,), 504)
    assert ops.conv_out_direct_ok(H, W, C, Cout, 3)
    got = ops.conv_out_direct(x.to(DEV), sc.to(DEV), sh.to(DEV), w.reshape(Cout, 9, C).contiguous().to(DEV), bias.to(DEV), B, H, W, C, Cout)
    t = x.double() * sc.double()[:, None, None, :] + sh.double()[:, None, None, :]
    act = t * torch.sigmoid(t)
    ref = torch.nn.functional.conv2d(act.permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), bias.double(), padding=1).permute(0, 2, 3, 1)
    assert got.shape == (B, H, W, Cout)
    assert rel_err(got, ref) < 3e-6
    nb = ops.conv_out_direct(x.to(DEV), sc.to(DEV), sh.to(DEV), w.reshape(Cout, 9, C).contiguous().to(DEV), None, B, H, W, C, Cout)
    assert rel_err(nb, ref - bias.double()) < 3e-6
    assert not ops.conv_out_direct_ok(H + 1, W, C, Cout, 3) and not ops.conv_out_direct_ok(H, W, C, 8, 3)


def test_upsample2x_split_is_the_split_of_the_interpolated_tensor():
    """muse_upsample2x_split_nhwc: nearest x2 written as the bf16x3 operand planes == hi = bf16(x), lo = bf16(x - hi) of
    F.interpolate(x, scale_factor=2, mode="nearest"), bit for bit; and the patch-slab convolution on those planes == the
    register-staged convolution that gathers the nearest neighbour itself, to f32 round-off"""
    ops = _ops()
    B, H, W, C = 2, 16, 24, 64
    x = rnd((B, H, W, C), 510)
    hi, lo = ops.upsample2x_split(x.to(DEV), B, H, W, C)
    up = torch.nn.functional.interpolate(x.permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest").permute(0, 2, 3, 1).contiguous()
    h = up.to(torch.bfloat16)
    l = (up - h.float()).to(torch.bfloat16)
    assert hi.shape == (B, 2 * H, 2 * W, C)
    assert torch.equal(hi.cpu().view(torch.int16), h.view(torch.int16)) and torch.equal(lo.cpu().view(torch.int16), l.view(torch.int16))
    Cout = 64
    w = rnd((Cout, 3, 3, C), 511, 0.05)
    wh, wl = ops.split_bf16(w.to(DEV))
    bias = rnd((Cout,), 512).to(DEV)
    a = ops.conv2d_nhwc_split2(hi, lo, wh, wl, B, 2 * H, 2 * W, C, Cout, bias=bias)
    b = ops.conv2d_nhwc_split(x.to(DEV), wh, wl, B, 2 * H, 2 * W, C, Cout, 3, bias=bias, upsample=True)
    ref = torch.nn.functional.conv2d(up.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), bias.double().cpu(), padding=1).permute(0, 2, 3, 1)
    assert rel_err(a, ref) < 3e-5 and rel_err(b, ref) < 3e-5 and rel_err(a, b) < 5e-6


def test_f32_gemm_as_three_bf16_products():
    """ops.f32_gemms_as_bf16x3: an f32 GEMM as hi*hi + hi*lo + lo*hi of bf16 operand planes with f32 accumulation (the "bf16x3"
    compute mode of the tape engines).  Error against float64: <= 2^-16 relative per product - between plain bf16 operands (2^-9) and
    the exact-f32 MFMA; every layout, a batched strided product with offsets, epilogue extras on the first launch only, the
    weight-gradient path with its K split; a product with rows that are not 16-byte chunks in bf16 stays on the exact kernel."""
    ops = _ops()
    T, K, N = 300, 256, 384
    x, w = rnd((T, K), 600).to(DEV), rnd((N, K), 601).to(DEV)
    res, bias = rnd((T, N), 602).to(DEV), rnd((N,), 603).to(DEV)
    ref = x.double() @ w.double().t() + bias.double() + res.double()
    exact = ops.linear(x, w, residual=res, bias=bias)
    with ops.f32_gemms_as_bf16x3():
        y3 = ops.linear(x, w, residual=res, bias=bias)
        dx3 = ops.linear_dgrad(res, w)                                          # [T, N] @ [N, K]: k-major B
        dw3 = torch.empty((N, K), device=DEV)
        ops.linear_wgrad(res, x, dw3, False)
        dw3b = dw3.clone()
        ops.linear_wgrad(res, x, dw3b, True)
    # the same products as three launches (MUSE_X3_CAT=0) instead of one over the concatenated 3K-long operands: same terms, another order
    import unittest.mock as um
    with um.patch.object(ops, "X3_CAT", False), ops.f32_gemms_as_bf16x3():
        y3b = ops.linear(x, w, residual=res, bias=bias)
        dx3b = ops.linear_dgrad(res, w)
        dw3c = torch.empty((N, K), device=DEV)
        ops.linear_wgrad(res, x, dw3c, False)
    assert rel_err(y3, y3b.double()) < 2e-6 and rel_err(dx3, dx3b.double()) < 2e-6 and rel_err(dw3, dw3c.double()) < 2e-6
    yb = ops.linear(x.to(torch.bfloat16), w.to(torch.bfloat16), out_dtype=torch.float32)
    scale = float((x.double().abs() @ w.double().abs().t()).max())
    e3 = float((y3.double() - ref).abs().max()) / scale
    eb = float((yb.double() - (x.double() @ w.double().t())).abs().max()) / scale
    ee = float((exact.double() - ref).abs().max()) / scale
    print(f"bf16x3 GEMM error / sum|a||b|: {e3:.2e} (plain bf16 operands {eb:.2e}, exact-f32 MFMA {ee:.2e}; 2^-16 = 1.5e-5)")
    assert e3 < 2.0 ** -16 and e3 < eb / 30
    assert rel_err(dx3, res.double() @ w.double()) < 2e-5
    assert rel_err(dw3, res.double().t() @ x.double()) < 2e-5 and rel_err(dw3b, 2 * (res.double().t() @ x.double())) < 2e-5
    # batched, strided, with an operand offset (attention's P V shape)
    Bn, S, hd = 6, 64, 32
    P, V = rnd((Bn, S, S), 604).to(DEV), rnd((Bn * S, hd * 2), 605).to(DEV)
    o = torch.empty((Bn, S, hd), device=DEV)
    with ops.f32_gemms_as_bf16x3():
        ops.gemm(P, V, o, S, hd, S, la=0, lb=1, lda=S, ldb=2 * hd, ldc=hd, b_off=hd, batch=Bn, zdiv=1, sA=(S * S, 0), sB=(S * 2 * hd, 0),
                 sC=(S * hd, 0))
    refo = torch.bmm(P.double(), V.view(Bn, S, 2 * hd)[:, :, hd:].double())
    assert rel_err(o, refo) < 2e-5
    # rows of 12 floats: no 16-byte chunks in bf16 -> the exact kernel answers, bit for bit
    a, b = rnd((40, 12), 606).to(DEV), rnd((24, 12), 607).to(DEV)
    with ops.f32_gemms_as_bf16x3():
        c1 = ops.linear(a, b)
    assert torch.equal(c1, ops.linear(a, b))


@pytest.mark.parametrize("M,N,K", [(512, 768, 1024), (300, 384, 256), (1000, 520, 200), (264, 136, 72)])
def test_bf16x3_four_plane_gemm(M, N, K):
    """muse_gemm_x3 (csrc/gemm256.h PipeX3): the bf16x3 product as one kernel on the four (hi, lo) operand planes - every operand
    layout, ragged tile edges, K that is no multiple of the 32-wide K-tile, bias + residual + accumulate in the epilogue; against float64
    (2^-16 of sum |a||b|) and against the route it replaces (one product over K-concatenated operands: the same three terms in another
    order).  Then the weight-gradient form with its K split on a long token dimension."""
    ops = _ops()
    import unittest.mock as um
    a, b = rnd((M, K), 700).to(DEV), rnd((N, K), 701).to(DEV)
    at, bt = a.t().contiguous(), b.t().contiguous()          # [K, M], [K, N]: the k-major forms
    res, bias = rnd((M, N), 702).to(DEV), rnd((N,), 703).to(DEV)
    ref = a.double() @ b.double().t()
    scale = float((a.double().abs() @ b.double().abs().t()).max())
    for la, lb, A, B in ((0, 0, a, b), (0, 1, a, bt), (1, 1, at, bt), (1, 0, at, b)):
        lda, ldb = (K if la == 0 else M), (K if lb == 0 else N)
        outs = []
        for native in (True, False):
            c = torch.empty((M, N), device=DEV)
            with um.patch.object(ops, "X3_NATIVE", native), ops.f32_gemms_as_bf16x3():
                ops.gemm(A, B, c, M, N, K, la=la, lb=lb, lda=lda, ldb=ldb, ldc=N)
            outs.append(c)
        e = float((outs[0].double() - ref).abs().max()) / scale
        assert e < 2.0 ** -16, (la, lb, e)
        assert rel_err(outs[0], outs[1].double()) < 2e-6, (la, lb)
    # epilogue extras, then accumulation onto the result
    c = torch.empty((M, N), device=DEV)
    with ops.f32_gemms_as_bf16x3():
        ops.gemm(a, b, c, M, N, K, la=0, lb=0, lda=K, ldb=K, ldc=N, alpha=0.5, bias=bias, residual=res, ldr=N)
        want = 0.5 * ref + bias.double() + res.double()
        assert rel_err(c, want) < 2e-5
        ops.gemm(a, b, c, M, N, K, la=0, lb=0, lda=K, ldb=K, ldc=N, accumulate=True)
        assert rel_err(c, want + ref) < 2e-5
    # the kernel really ran (the route is a silent fallback otherwise): a shape it refuses returns None
    with ops.f32_gemms_as_bf16x3(False):
        a2, b2 = ops.split_planes(a), ops.split_planes(b)
        c2 = torch.empty((M, N), device=DEV)
        assert ops.gemm(a2[0], b2[0], c2, M, N, K, la=0, lb=0, lda=K, ldb=K, ldc=N, x3_lo=(a.numel(), b.numel())) is c2
        assert float((c2.double() - ref).abs().max()) / scale < 2.0 ** -16
        small = torch.empty((64, N), device=DEV)
        assert ops.gemm(a2[0], b2[0], small, 64, N, K, la=0, lb=0, lda=K, ldb=K, ldc=N, x3_lo=(a.numel(), b.numel())) is None


def test_bf16x3_producers_write_operand_planes():
    """Inside a bf16x3 step the kernels whose f32 result feeds a weight GEMM also write its (hi, lo) operand planes (GLU forward /
    backward, AdaLN-norm forward / backward, the fused attention's context and gradients): bit for bit the split of the f32 result
    (muse_split_f32_to_bf16x2), registered under the result tensor, so the product that reads it launches no split; the f32 results
    are the plain kernels' bits.  Outside a step nothing changes."""
    ops = _ops()
    rows, inter, C_, B = 512, 256, 512, 2
    ab, dh = rnd((rows, 2 * inter), 720).to(DEV), rnd((rows, inter), 721).to(DEV)
    x, res, w, ss = rnd((rows, C_), 722).to(DEV), rnd((rows, C_), 723).to(DEV), rnd((C_,), 724).to(DEV) + 1.0, rnd((B, 2 * C_), 725).to(DEV)
    nh, hd, Sq = 2, 64, 256
    H = nh * hd
    qkv, dctx = rnd((B * Sq, 3 * H), 726).to(DEV), rnd((B * Sq, H), 727).to(DEV)
    alpha = 0.125

    def run():
        h = ops.glu_fwd(ab)
        dab = ops.glu_bwd(ab, dh)
        m, pre = ops.norm_adaln_fwd(x, w, ss, B, 1e-6, 1, residual=res)
        dv, dw, dss = ops.norm_adaln_bwd(m, pre, w, ss, B, 1e-6, 1, dpre=x)
        ctx, lse = ops.attention_x3_fwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], B, Sq, Sq, nh, hd, alpha)
        dqkv = torch.empty_like(qkv)
        pl = ops.x3_new_planes(dqkv)
        lo = dqkv.numel()
        ops.attention_x3_bwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], ctx, dctx, lse, B, Sq, Sq, nh, hd, alpha, dq=dqkv[:, :H],
                             dk=dqkv[:, H:2 * H], dv=dqkv[:, 2 * H:],
                             planes=None if pl is None else ((pl[0][:, :H], lo), (pl[0][:, H:2 * H], lo), (pl[0][:, 2 * H:], lo)))
        ops.x3_put_planes(dqkv, pl)
        return dict(h=h, dab=dab, m=m, dv=dv, ctx=ctx, dqkv=dqkv)
    plain = run()
    im = ops.X3Images()
    with ops.f32_gemms_as_bf16x3(True, im):
        fused = run()
        assert im.produced == 6
        for name, t in fused.items():
            assert torch.equal(t, plain[name]), name                       # the f32 results do not change
            misses = im.misses
            planes = ops.split_planes(t)                                   # what a product reading t gets: the producer's planes
            assert im.misses == misses, name
            want = ops._split_planes_now(t)
            assert torch.equal(planes, want), name


def test_bf16x3_planes_only_operands():
    """ops.Planes: a GLU result that only weight GEMMs read exists as its operand planes alone (no f32 tensor).  The forward product, the
    dX product and both weight-gradient roles give the bits of the route that also writes the f32 tensor; a product the four-plane
    kernel does not take raises instead of reading bytes that are not there; outside a bf16x3 step planes_only is refused."""
    ops = _ops()
    from muse._hip import MuseHipError
    rows, inter, N = 512, 256, 384
    ab, dh = rnd((rows, 2 * inter), 730).to(DEV), rnd((rows, inter), 731).to(DEV)
    w, x, wsmall = rnd((N, inter), 732).to(DEV), rnd((rows, N), 733).to(DEV), rnd((64, inter), 734).to(DEV)
    w2 = rnd((2 * inter, N), 735).to(DEV)
    with pytest.raises(MuseHipError):
        ops.glu_fwd(ab, planes_only=True)
    assert not ops.planes_only_ok(rows, inter)
    with ops.f32_gemms_as_bf16x3(True, ops.X3Images()):
        assert ops.planes_only_ok(rows, inter) and not ops.planes_only_ok(64, inter)
        h, hp = ops.glu_fwd(ab), ops.glu_fwd(ab, planes_only=True)
        dab, dabp = ops.glu_bwd(ab, dh), ops.glu_bwd(ab, dh, planes_only=True)
        assert isinstance(hp, ops.Planes) and tuple(hp.shape) == (rows, inter) and torch.equal(hp.planes, ops._split_planes_now(h))
        assert torch.equal(dabp.planes, ops._split_planes_now(dab))
        assert torch.equal(ops.linear(hp, w), ops.linear(h, w))                                       # forward product, A operand
        assert torch.equal(ops.linear_dgrad(dabp, w2), ops.linear_dgrad(dab, w2))                     # dX product, A operand
        g0, g1 = torch.empty((N, inter), device=DEV), torch.empty((N, inter), device=DEV)
        ops.linear_wgrad(x, h, g0, False); ops.linear_wgrad(x, hp, g1, False)                         # dW: planes as the B operand
        assert torch.equal(g0, g1)
        g2, g3 = torch.empty((2 * inter, N), device=DEV), torch.empty((2 * inter, N), device=DEV)
        ops.linear_wgrad(dab, x, g2, False); ops.linear_wgrad(dabp, x, g3, False)                     # dW: planes as the A operand
        assert torch.equal(g2, g3)
        with pytest.raises(MuseHipError):
            ops.linear(hp, wsmall)                                                                    # N = 64: not a four-plane product
        # the fused attention's packed gradient as planes only: the bits of the f32 + planes route
        B, nh, hd, Sq = 2, 2, 64, 256
        H = nh * hd
        qkv, dctx = rnd((B * Sq, 3 * H), 736).to(DEV), rnd((B * Sq, H), 737).to(DEV)
        ctx, lse = ops.attention_x3_fwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], B, Sq, Sq, nh, hd, 0.125)
        dqkv = torch.empty_like(qkv)
        ops.attention_x3_bwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], ctx, dctx, lse, B, Sq, Sq, nh, hd, 0.125, dq=dqkv[:, :H],
                             dk=dqkv[:, H:2 * H], dv=dqkv[:, 2 * H:])
        pl = torch.zeros((2,) + tuple(qkv.shape), dtype=torch.bfloat16, device=DEV)
        lo = qkv.numel()
        r = ops.attention_x3_bwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], ctx, dctx, lse, B, Sq, Sq, nh, hd, 0.125, planes_only=True,
                                 planes=((pl[0][:, :H], lo), (pl[0][:, H:2 * H], lo), (pl[0][:, 2 * H:], lo)))
        assert r == (None, None, None) and torch.equal(pl, ops._split_planes_now(dqkv))


def test_bf16x3_weight_gradient_with_k_split():
    """dW = dY^T X over 8192 tokens in the bf16x3 mode: the four-plane kernel with its K slices through a workspace (fixed summation
    order), plain and accumulating; against float64 and against the K-concatenated route"""
    ops = _ops()
    import unittest.mock as um
    T, N, K = 8192, 384, 512
    dy, x = rnd((T, N), 710).to(DEV), rnd((T, K), 711).to(DEV)
    ref = dy.double().t() @ x.double()
    assert ops.wgrad_splits(N, K, T, torch.bfloat16, slots=256, tile=256, ktile_us=4.7) > 1
    dw = torch.empty((N, K), device=DEV)
    with ops.f32_gemms_as_bf16x3():
        ops.linear_wgrad(dy, x, dw, False)
        dw2 = dw.clone()
        ops.linear_wgrad(dy, x, dw2, True)
        again = torch.empty((N, K), device=DEV)
        ops.linear_wgrad(dy, x, again, False)
    with um.patch.object(ops, "X3_NATIVE", False), ops.f32_gemms_as_bf16x3():
        dwc = torch.empty((N, K), device=DEV)
        ops.linear_wgrad(dy, x, dwc, False)
    assert rel_err(dw, ref) < 1e-5 and rel_err(dw2, 2 * ref) < 1e-5 and rel_err(dw, dwc.double()) < 2e-6
    assert torch.equal(dw, again)                 # deterministic: slices are summed in a fixed order


def _tf32(x):
    """round to nearest even at 10 mantissa bits: the TF32 operand format (f32 in, f32 out)"""
    i = x.contiguous().view(torch.int32)
    return ((i + 0xFFF + ((i >> 13) & 1)) & ~0x1FFF).view(torch.float32)


@pytest.mark.parametrize("M,N,K", [(512, 768, 1024), (300, 384, 256), (1000, 520, 200), (264, 136, 72)])
def test_f16_mode_product_is_the_tf32_product(M, N, K):
    """ops.f32_gemms_as_f16 (muse_gemm dtype MUSE_F16, v_mfma_f32_16x16x32_f16): one product of IEEE-half operand images with f32
    accumulation.  Half and TF32 share the 10-bit mantissa, so on operands inside half's exponent range the result is the TF32 product
    configs/cc12m_uvit_clip.yaml:103 (`enable_tf32`) computes: equal to an emulated one (operands rounded to 10 mantissa bits, float64
    accumulation) up to f32 accumulation error (1e-6 of sum |a||b|), and 2^-11 of sum |a||b| from float64 like it.  Every operand layout,
    ragged tile edges, epilogue extras; a gradient-sized operand through the power-of-two scale; half subnormals through the MFMA."""
    ops = _ops()
    a, b = rnd((M, K), 720).to(DEV), rnd((N, K), 721, 0.05).to(DEV)
    at, bt = a.t().contiguous(), b.t().contiguous()
    res, bias = rnd((M, N), 722).to(DEV), rnd((N,), 723).to(DEV)
    ref = a.double() @ b.double().t()
    emu = _tf32(a).double() @ _tf32(b).double().t()
    mag = a.double().abs() @ b.double().abs().t()
    for la, lb, A, B in ((0, 0, a, b), (0, 1, a, bt), (1, 1, at, bt), (1, 0, at, b)):
        lda, ldb = (K if la == 0 else M), (K if lb == 0 else N)
        c = torch.empty((M, N), device=DEV)
        im = ops.F16Images()
        with ops.f32_gemms_as_f16(True, im):
            ops.gemm(A, B, c, M, N, K, la=la, lb=lb, lda=lda, ldb=ldb, ldc=N)
        if (la and M % 8) or (lb and N % 8):       # rows of a k-major half operand must be whole 16-byte chunks: the product stays exact f32
            assert im.misses == 0 and float(((c.double() - ref).abs() / mag).max()) < 1e-6, (la, lb)
            continue
        assert im.misses == 2                                              # the half kernel ran (no silent exact-f32 product)
        assert float(((c.double() - emu).abs() / mag).max()) < 1e-6, (la, lb)
        assert float(((c.double() - ref).abs() / mag).max()) < 2.0 ** -11, (la, lb)
        assert im.stats()[0] == 0
    c = torch.empty((M, N), device=DEV)
    with ops.f32_gemms_as_f16():
        ops.gemm(a, b, c, M, N, K, la=0, lb=0, lda=K, ldb=K, ldc=N, alpha=0.5, bias=bias, residual=res, ldr=N)
        want = 0.5 * emu + bias.double() + res.double()
        assert rel_err(c, want) < 2e-6
        ops.gemm(a, b, c, M, N, K, la=0, lb=0, lda=K, ldb=K, ldc=N, accumulate=True)
        assert rel_err(c, want + emu) < 2e-6
        tiny = torch.empty((64, N), device=DEV)                           # a shape the half kernels refuse stays exact f32
        ops.gemm(a[:64].contiguous(), b, tiny, 64, N, K, la=0, lb=0, lda=K, ldb=K, ldc=N)
        assert rel_err(tiny, ref[:64]) < 1e-5
    # a gradient-sized operand: unscaled it falls below half's range (elements rounded to zero are counted), with the pass's
    # power-of-two scale it is the TF32 product again - the scale is undone exactly in alpha
    g = a * 1e-7
    for scale, bound in ((None, None), (2.0 ** 24, 1e-6)):
        im = ops.F16Images()
        if scale is not None:
            im.backward = True
            im.set_grad_scale(scale)
        with ops.f32_gemms_as_f16(True, im):
            ops.gemm(g, b, c, M, N, K, la=0, lb=0, lda=K, ldb=K, ldc=N)
        err = float(((c.double() - _tf32(g).double() @ _tf32(b).double().t()).abs() / (mag * 1e-7)).max())
        clamped, flushed = im.stats()
        if scale is None:
            assert flushed > 0.1 * g.numel() and err > 1e-3
        else:
            assert err < bound and clamped == 0 and flushed < 1e-4 * g.numel()
    with pytest.raises(Exception):
        ops.F16Images().set_grad_scale(3.0)                                # not a power of two: the un-scaling would round
    # operands that are half SUBNORMALS (exactly representable): the MFMA does not flush them - the product is exact
    sub = torch.randint(-512, 512, (M, K), device=DEV).float() * 2.0 ** -24
    bb = torch.randint(-8, 8, (N, K), device=DEV).float()
    with ops.f32_gemms_as_f16():
        ops.gemm(sub, bb, c, M, N, K, la=0, lb=0, lda=K, ldb=K, ldc=N)
    assert torch.equal(c.double(), sub.double() @ bb.double().t())


def test_f16_mode_weight_gradient_with_k_split():
    """dW = dY^T X over 8192 tokens in the f16 mode: half images of both operands (dY with the pass's gradient scale), the 256^2
    kernel's K slices through a workspace - the emulated TF32 product to f32 accumulation error, deterministic, accumulating"""
    ops = _ops()
    T, N, K = 8192, 384, 512
    dy, x = (rnd((T, N), 730) * 3e-6).to(DEV), rnd((T, K), 731).to(DEV)
    emu = _tf32(dy).double().t() @ _tf32(x).double()
    im = ops.F16Images()
    im.backward = True
    im.set_grad_scale(2.0 ** 20)
    dw = torch.empty((N, K), device=DEV)
    with ops.f32_gemms_as_f16(True, im):
        ops.linear_wgrad(dy, x, dw, False)
        dw2 = dw.clone()
        ops.linear_wgrad(dy, x, dw2, True)
        again = torch.empty((N, K), device=DEV)
        ops.linear_wgrad(dy, x, again, False)
    assert im.misses == 2 and im.hits == 4 and im.stats() == (0, 0)
    assert rel_err(dw, emu) < 2e-6 and rel_err(dw2, 2 * emu) < 2e-6
    assert torch.equal(dw, again)


@pytest.mark.parametrize("B,Sq,Skv,nh,packed", [(3, 256, 256, 4, True), (2, 256, 77, 3, False), (2, 256, 240, 2, False), (1, 512, 512, 2, True), (1, 512, 77, 2, False)])
def test_fused_attention_f16_mode(B, Sq, Skv, nh, packed):
    """attention3.hip H16: inside an f16 step (muse_operand_images) the fused attention core runs every product as ONE half MFMA
    product - q, k, v, P rounded to half's 10-bit mantissa (the TF32 operand format), dO and dS times the pass's power-of-two gradient
    scale before their conversion, results handed back divided by it.  Against float64: forward and gradients to TF32-class error
    (2e-3 of their scale; the bf16x3 kernels: 2e-5 / 5e-5, a bf16 core: 1.5e-2 / 3e-2); a gradient-sized dO (1e-6) needs the scale
    (without it dS falls below half's range) and with it matches the same bound; the result does not depend on the scale beyond rounding;
    one-tile shapes and the block-by-block form of longer sequences."""
    ops = _ops()
    hd = 64
    H = nh * hd
    alpha = 0.125
    if packed:
        qkv = rnd((B * Sq, 3 * H), 760, 1.0).to(DEV)
        qd, kd, vd = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
    else:
        qd, kv = rnd((B * Sq, H), 761, 1.0).to(DEV), rnd((B * Skv, 2 * H), 762, 1.0).to(DEV)
        kd, vd = kv[:, :H], kv[:, H:]
    dctx = rnd((B * Sq, H), 763).to(DEV) * 1e-6
    q = qd.double().reshape(B, Sq, nh, hd).transpose(1, 2).detach().requires_grad_(True)
    k = kd.double().reshape(B, Skv, nh, hd).transpose(1, 2).detach().requires_grad_(True)
    v = vd.double().reshape(B, Skv, nh, hd).transpose(1, 2).detach().requires_grad_(True)
    ref = (torch.softmax(q @ k.transpose(-1, -2) * alpha, dim=-1) @ v).transpose(1, 2).reshape(B * Sq, H)
    ref.backward(dctx.double())
    gq, gk, gv = (t.grad.transpose(1, 2).reshape(-1, H) for t in (q, k, v))
    outs = {}
    for S in (1.0, 2.0 ** 20, 2.0 ** 22):
        im = ops.F16Images()
        with ops.f32_gemms_as_f16(True, im):
            ctx, lse = ops.attention_x3_fwd(qd, kd, vd, B, Sq, Skv, nh, hd, alpha)
        im.backward = True
        im.set_grad_scale(S)
        with ops.f32_gemms_as_f16(True, im):
            dq, dk, dv = ops.attention_x3_bwd(qd, kd, vd, ctx, dctx, lse, B, Sq, Skv, nh, hd, alpha)
        outs[S] = (ctx, dq, dk, dv)
        ef = rel_err(ctx, ref.detach())
        eg = [rel_err(a, b) for a, b in ((dq, gq), (dk, gk), (dv, gv))]
        print(f"f16-mode attention {Sq} x {Skv}, gradient scale {S:g}: ctx {ef:.1e}, dq / dk / dv {eg[0]:.1e} / {eg[1]:.1e} / {eg[2]:.1e} of float64; overflowed {im.stats()[0]}")
        assert ef < 2e-3
        if S == 1.0:
            assert max(eg) > 1e-2          # dO ~ 1e-6 unscaled: dS is far below half's normal range
        else:
            assert max(eg) < 2e-3 and im.stats()[0] == 0
    for a, b in zip(outs[2.0 ** 20][1:], outs[2.0 ** 22][1:]):
        assert rel_err(a, b.double()) < 2e-3
    # outside the mode the same entry points are the bf16x3 kernels again
    ctx3, _ = ops.attention_x3_fwd(qd, kd, vd, B, Sq, Skv, nh, hd, alpha)
    assert rel_err(ctx3, ref.detach()) < 2e-5


def test_f16_mode_producers_write_half_images():
    """Inside an f16 step the producer kernels (GLU forward / backward, AdaLN-norm forward / backward, the fused attention's context and
    gradients) write their result's IEEE-half operand image next to the f32 result (muse_operand_images): bit for bit what
    muse_cast_f32_to_f16 makes of that result - unscaled in a forward pass, times the pass's gradient scale in a backward pass -
    registered under the result tensor so the product that reads it launches no cast; the f32 results are the plain kernels' bits
    (the attention core's, which computes in half itself inside the mode, to that precision).
    A result only weight GEMMs read exists as its image alone (ops.Planes) and gives the same products."""
    ops = _ops()
    rows, inter, C_, B = 512, 256, 512, 2
    ab, dh = rnd((rows, 2 * inter), 740).to(DEV), rnd((rows, inter), 741, 1e-5).to(DEV)
    x, res, w, ss = rnd((rows, C_), 742).to(DEV), rnd((rows, C_), 743).to(DEV), rnd((C_,), 744).to(DEV) + 1.0, rnd((B, 2 * C_), 745).to(DEV)
    nh, hd, Sq = 2, 64, 256
    H = nh * hd
    qkv, dctx = rnd((B * Sq, 3 * H), 746).to(DEV), rnd((B * Sq, H), 747, 1e-5).to(DEV)
    wl = rnd((384, inter), 748).to(DEV)

    def forward():
        h = ops.glu_fwd(ab)
        m, pre = ops.norm_adaln_fwd(x, w, ss, B, 1e-6, 1, residual=res)
        ctx, lse = ops.attention_x3_fwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], B, Sq, Sq, nh, hd, 0.125)
        return dict(h=h, m=m, ctx=ctx), pre, lse

    def backward(m, pre, ctx, lse):
        dab = ops.glu_bwd(ab, dh)
        dv, dw, dss = ops.norm_adaln_bwd(m * 1e-5, pre, w, ss, B, 1e-6, 1, dpre=x * 1e-5)
        dqkv = torch.empty_like(qkv)
        pl = ops.x3_new_planes(dqkv)
        lo = dqkv.numel()
        ops.attention_x3_bwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], ctx, dctx, lse, B, Sq, Sq, nh, hd, 0.125, dq=dqkv[:, :H],
                             dk=dqkv[:, H:2 * H], dv=dqkv[:, 2 * H:],
                             planes=None if pl is None else ((pl[0][:, :H], lo), (pl[0][:, H:2 * H], lo), (pl[0][:, 2 * H:], lo)))
        ops.x3_put_planes(dqkv, pl)
        return dict(dab=dab, dv=dv, dqkv=dqkv)
    pf, pre, lse = forward()
    pb = backward(pf["m"], pre, pf["ctx"], lse)
    im = ops.F16Images()
    S = 2.0 ** 14
    for bwd, plain in ((False, pf), (True, pb)):
        im.backward = bwd
        im.set_grad_scale(S)
        with ops.f32_gemms_as_f16(True, im):
            fused = backward(pf["m"], pre, pf["ctx"], lse) if bwd else forward()[0]
            for name, t in fused.items():
                if name in ("ctx", "dqkv"):                                # the attention core itself computes in half inside the mode
                    assert rel_err(t, plain[name].double()) < 2e-3, name
                else:
                    assert torch.equal(t, plain[name]), name               # the f32 results do not change
                misses = im.misses
                img = im.image(t, S if bwd else 1.0)                       # what a product reading t gets: the producer's image
                assert im.misses == misses, name
                assert torch.equal(img, ops.cast_to_f16(t, S if bwd else 1.0)), name
                assert img._muse_scale == (S if bwd else 1.0)
            # a result that exists as its image only: the same forward / dX / dW products
            assert ops.planes_only_ok(rows, inter)
            if not bwd:
                hp = ops.glu_fwd(ab, planes_only=True)
                assert isinstance(hp, ops.Planes) and hp.half and torch.equal(hp.planes[0], ops.cast_to_f16(fused["h"]))
                assert torch.equal(ops.linear(hp, wl), ops.linear(fused["h"], wl))
            else:
                dp = ops.glu_bwd(ab, dh, planes_only=True)
                assert dp.half and dp.scale == S and torch.equal(dp.planes[0], ops.cast_to_f16(fused["dab"], S))
                w2 = rnd((2 * inter, 384), 749).to(DEV)
                assert torch.equal(ops.linear_dgrad(dp, w2), ops.linear_dgrad(fused["dab"], w2))
                g0, g1 = torch.empty((2 * inter, 384), device=DEV), torch.empty((2 * inter, 384), device=DEV)
                xx = rnd((rows, 384), 750).to(DEV)
                ops.linear_wgrad(fused["dab"], xx, g0, False); ops.linear_wgrad(dp, xx, g1, False)
                assert torch.equal(g0, g1)
    assert im.produced >= 6
    # outside the mode the same entry points write bf16 planes again (the kernels' image format is restored on exit)
    with ops.f32_gemms_as_bf16x3(True, ops.X3Images()):
        h = ops.glu_fwd(ab)
        assert torch.equal(ops.split_planes(h), ops._split_planes_now(h))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_upsample2x_matches_interpolate(dtype):
    """muse_upsample2x_nhwc == F.interpolate(scale_factor=2, mode="nearest") (taming Upsample without its convolution), bit for bit"""
    ops = _ops()
    B, H, W, C = 2, 5, 7, 24
    x = rnd((B, H, W, C), 520).to(dtype)
    y = ops.upsample2x(x.to(DEV), B, H, W, C)
    ref = torch.nn.functional.interpolate(x.float().permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest").permute(0, 2, 3, 1).to(dtype)
    assert y.shape == (B, 2 * H, 2 * W, C) and torch.equal(y.cpu(), ref)


@pytest.mark.parametrize("M,N,K,out_dtype", [(512, 1024, 1024, torch.float32), (512, 1024, 1024, torch.bfloat16), (300, 5632, 1024, torch.bfloat16),
                                             (2048, 1024, 2816, torch.float32), (77, 2048, 768, torch.bfloat16),
                                             # slice counts that do not divide the K-tiles (16 asked / 15 hold work, 5 / 4): the empty
                                             # trailing slice of the workspace is never written and must not be summed
                                             (256, 1024, 2816, torch.float32), (768, 1024, 1024, torch.bfloat16)])
def test_small_batch_forward_products_split_k(M, N, K, out_dtype, monkeypatch):
    """small-batch decoding path of ops.gemm: a forward Linear of <= 2048 rows whose 128^2 tiles would fill a fraction of the chip is cut
    along K into an f32 workspace and summed by muse_sum_slices_epilogue together with the Linear's epilogue (bias, residual, output
    dtype).  Against float64, and against the plain path (ops.SKINNY off): the same product up to the bf16 rounding of the output
    (f32 outputs: f32 summation-order noise)."""
    ops = _ops()
    x = rnd((M, K), 700).to(DEV).to(torch.bfloat16)
    w = (rnd((N, K), 701) / math.sqrt(K)).to(DEV).to(torch.bfloat16)
    bias = rnd((N,), 702).to(DEV)
    res = rnd((M, N), 703).to(DEV).to(out_dtype)
    ref = x.double() @ w.double().t() + bias.double() + res.double()
    calls = []
    from muse import _hip
    real = _hip.lib().muse_sum_slices_epilogue
    monkeypatch.setattr(ops, "SKINNY", 2)          # (1, the default, takes the path only while a HIP graph is being captured)
    y1 = ops.linear(x, w, out_dtype=out_dtype, residual=res, bias=bias)
    monkeypatch.setattr(ops, "SKINNY", 0)
    y0 = ops.linear(x, w, out_dtype=out_dtype, residual=res, bias=bias)
    tol = 1e-5 if out_dtype == torch.float32 else 6e-3
    assert rel_err(y1, ref) < tol and rel_err(y0, ref) < tol
    assert rel_err(y1, y0.double()) < (2e-6 if out_dtype == torch.float32 else 8e-3)
    if out_dtype == torch.float32:
        assert not torch.equal(y1, y0) or True      # (different summation order: equality is neither required nor excluded)
    # no epilogue extras, plain output
    monkeypatch.setattr(ops, "SKINNY", 2)
    y2 = ops.linear(x, w, out_dtype=out_dtype)
    assert rel_err(y2, x.double() @ w.double().t()) < tol
