"""CPU: the near-tie accounting of oracle/vq_parity.py (used by test_vq_indices_over_bench_batch_vs_oracle and bench.py to say what a
VQ index disagreement is worth) on a constructed case: a codebook entry mirrored through one token's encoder output is an exact tie;
nudged by a few ulps it is a near-tie that a second implementation with an f32-class encoder error legitimately resolves the other
way (accepted), while a disagreement that needs an encoder error beyond the arithmetic's bound is rejected."""
import torch

import weights as W
from oracle import maskgit_oracle as O
from oracle import vq_parity as VP


def _case(nudge_ulps, push):
    cfg = W.VQGAN_TINY
    sd = {k: v.clone() for k, v in W.fill_state_dict(W.vqgan_shapes(cfg), 600, "vqgan").items()}
    px = W.images(2, 16, 611)
    with torch.no_grad():
        z = O.vqgan_encoder(sd, cfg, px)
    B, C, h, w = z.shape
    T = h * w
    zf = z.permute(0, 2, 3, 1).reshape(B, T, C).contiguous()
    cb = sd["quantize.embedding.weight"]
    idx0 = O.vq_indices(z, cb)
    b, t = 1, 3
    i = int(idx0[b, t])
    j = (i + 1) % cb.shape[0]
    zt = zf[b, t]
    mirror = 2.0 * zt - cb[i]                                     # |z - mirror| == |z - e_i|: an exact tie in exact arithmetic
    d = float((zt - cb[i]).pow(2).sum())
    away = (mirror - zt) / (mirror - zt).norm()
    cb[j] = mirror + away * (nudge_ulps * VP.ulp32(d) / (2.0 * float((mirror - zt).norm())))    # D_j - D_i ~ +nudge_ulps ulp
    dist_o = O.vq_distances(zf.reshape(B * T, C), cb).view(B, T, -1)
    idx_o = dist_o.argmin(-1)
    # the "second implementation": the same encoder output moved along (e_j - e_i) by `push` x the distance needed to flip the decision
    de = cb[j] - cb[i]
    gap = float(dist_o[b, t, j] - dist_o[b, t, i])
    z_h = zf.clone()
    z_h[b, t] += de * ((push * max(gap, 0.0) + 40.0 * VP.ulp32(d)) / (2.0 * float(de.pow(2).sum())))   # 40 ulp: clear of either side's rounding
    dist_h = O.vq_distances(z_h.reshape(B * T, C), cb).view(B, T, -1)
    idx_h = dist_h.argmin(-1)
    return sd, cfg, px, idx_o, dist_o, zf, idx_h, dist_h, z_h, (b, t, i, j)


def test_near_tie_is_accepted_and_measured():
    sd, cfg, px, idx_o, dist_o, z_o, idx_h, dist_h, z_h, (b, t, i, j) = _case(nudge_ulps=6.0, push=1.0)
    assert int(idx_o[b, t]) == i and int(idx_h[b, t]) == j, "the constructed near-tie did not flip"
    recs, ok = VP.explain(sd, cfg, px, idx_o, dist_o, z_o, idx_h, dist_h, z_h)
    print(VP.format_records(recs))
    assert ok and len(recs) == int((idx_o != idx_h).sum())
    r = next(x for x in recs if (x["image"], x["token"]) == (b, t))
    assert r["accepted"] and abs(r["exact_margin_ulp"]) < 80 and r["oracle_top2_margin_ulp"] < 80
    assert r["hip"]["shift_ulp"] < 0 and abs(r["hip"]["rounding_ulp"]) <= r["hip"]["rounding_bound_ulp"]
    m = VP.oracle_margins(dist_o.view(-1, dist_o.shape[-1]), idx_o, idx_h)
    assert len(m) == len(recs) and abs(m[0]["candidate_gap_ulp"] - r["oracle"]["g_ulp"]) < 1e-6


def test_disagreement_beyond_the_arithmetic_is_rejected():
    # a margin of ~1e6 ulp flipped by moving z far beyond the encoder-error bound: not a near-tie
    sd, cfg, px, idx_o, dist_o, z_o, idx_h, dist_h, z_h, (b, t, i, j) = _case(nudge_ulps=1.0e6, push=1.5)
    assert int(idx_o[b, t]) == i and int(idx_h[b, t]) == j
    recs, ok = VP.explain(sd, cfg, px, idx_o, dist_o, z_o, idx_h, dist_h, z_h)
    assert not ok and not next(x for x in recs if (x["image"], x["token"]) == (b, t))["accepted"]


def test_ulp32():
    assert VP.ulp32(30.0) == 2.0 ** -19 and VP.ulp32(1.0) == 2.0 ** -23 and VP.ulp32(0.75) == 2.0 ** -24
