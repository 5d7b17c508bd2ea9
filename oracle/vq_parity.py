"""TEST INFRASTRUCTURE (CPU only; imported by tests/, bench.py:cpu_baseline and nothing else): what a disagreement between two
implementations of `VectorQuantizer.get_code` (muse/modeling_maskgit_vqgan.py:303-316,342-348) on one token is worth.

north_star asks for bit-exact VQ token indices.  Two f32-class implementations of the same encoder + nearest-codebook search agree
on a token unless its two best codebook entries are closer than what the arithmetic can resolve.  This module MEASURES that, it does
not wave a relative tolerance at it.  For a token where the oracle (f32, torch CPU) picked entry i and the HIP path picked entry j:

    g_x   = d_x[j] - d_x[i]                 what implementation x evaluated in f32 (x = o: oracle, h: HIP); g_o >= 0 >= g_h
    m64   = D(z64)[j] - D(z64)[i]           the same difference in EXACT arithmetic: the encoder re-run in float64 (`z64`) and the
                                            distances in float64.  D_j - D_i = |e_j|^2 - |e_i|^2 - 2 z.(e_j - e_i) is LINEAR in z, so
    g_x   = m64 + shift_x + r_x             splits exactly into
    shift_x = -2 (z_x - z64).(e_j - e_i)      the encoder's deviation from exact arithmetic, projected on the one direction that matters
    r_x     = g_x - (D(z_x)[j] - D(z_x)[i])   the rounding of x's own f32 distance evaluation (addmm of 256 products + 2 additions)

Each component is held to the bound of ITS arithmetic:
    |r_x|  <= 2 * gamma(lambda = 6, n = K + 2) * S      probabilistic f32 summation bound (Higham & Mary 2019: gamma~_n(lambda) =
                                                        lambda * sqrt(n) * 2^-24 holds with probability >= 1 - 2 n exp(-lambda^2 / 2),
                                                        8e-6 here), S = |z|^2 + |e|^2 + 2 sum_k |z_k e_k| (magnitude sum), once per distance
    |z_x - z64|_inf <= 5 * 2^-16 * |z64|_inf            bf16x3 products carry <= 2^-16 relative error each (hi*hi + hi*lo + lo*hi, lo*lo
                                                        dropped) with random signs, so one convolution's output error stays <= 2^-16 of
                                                        its scale, GroupNorm / SiLU / residual adds do not amplify a relative error, and
                                                        the L = 23 sequential convolutions of the f16 encoder add like a random walk:
                                                        sqrt(23) < 5.  (The f32 sides sit far below it; measured ratios are printed.)
Given g_o >= 0 >= g_h (each side really picked its own minimum) the split gives  -(shift_h + r_h) >= m64 >= -(shift_o + r_o):  the
exact margin lies between the two implementations' measured deviations, and each deviation is within its arithmetic's bound - that
is what "an f32 near-tie" means here.  A disagreement whose margin needs a larger encoder error or a larger rounding than the bound is
REJECTED.
Everything is reported in f32 ulps of the distance (an ulp at d ~ 30 is 1.9e-6) so that "near-tie" is a number."""
from __future__ import annotations

import math
from typing import Dict, List

import torch

from . import maskgit_oracle as O

U32 = 2.0 ** -24


def ulp32(x: float) -> float:
    """spacing of float32 at |x|"""
    x = abs(float(x))
    if x == 0.0:
        return 2.0 ** -149
    return 2.0 ** (math.floor(math.log2(x)) - 23)


def gamma_prob(n: int, lam: float = 6.0) -> float:
    return lam * math.sqrt(n) * U32


def oracle_margins(dist_o: torch.Tensor, idx_o: torch.Tensor, idx_h: torch.Tensor) -> List[Dict]:
    """cheap form (no float64 re-run): for every token where the two index tensors differ, the oracle's own f32 top-2 margin and the
    gap between the two candidates in the oracle's distance row, in f32 ulps.  dist_o [N, Kc] f32, idx_* [N]."""
    out = []
    for n in (idx_o.reshape(-1) != idx_h.reshape(-1)).nonzero().reshape(-1).tolist():
        row = dist_o[n]
        i, j = int(idx_o.reshape(-1)[n]), int(idx_h.reshape(-1)[n])
        top2 = torch.topk(row, 2, largest=False).values
        u = ulp32(float(row[i]))
        out.append({"token": n, "idx_oracle": i, "idx_hip": j, "d": float(row[i]), "oracle_top2_margin_ulp": float(top2[1] - top2[0]) / u,
                    "candidate_gap_ulp": float(row[j] - row[i]) / u})
    return out


def explain(sd, cfg, px: torch.Tensor, idx_o: torch.Tensor, dist_o: torch.Tensor, z_o: torch.Tensor, idx_h: torch.Tensor,
            dist_h: torch.Tensor, z_h: torch.Tensor, max_images: int = 8):
    """Full decomposition for every disagreeing token.  px [B, 3, H, W] f32; idx_* [B, T]; dist_* [B, T, Kc] f32 (each side's own
    distance rows); z_* [B, T, D] f32 (each side's encoder output, NHWC-flattened).  -> (records, ok): one record per disagreement
    with every quantity above in ulps and `ok` = all bounds hold.  The float64 encoder runs once per image that has a disagreement
    (at most `max_images`: more than that is not a near-tie phenomenon and fails)."""
    B, T = idx_o.shape
    cb = sd["quantize.embedding.weight"]
    cb64 = cb.double()
    K = cb.shape[1]
    mism = (idx_o != idx_h).nonzero().tolist()
    images = sorted({b for b, _ in mism})
    recs, ok = [], True
    if len(images) > max_images:
        return [{"error": f"{len(mism)} disagreements over {len(images)} images: not a near-tie phenomenon"}], False
    sd64 = {k: v.double() for k, v in sd.items()}
    g2 = 2.0 * gamma_prob(K + 2)
    for b in images:
        with torch.no_grad():
            z64 = O.vqgan_encoder(sd64, cfg, px[b:b + 1].double())[0].permute(1, 2, 0).reshape(T, K)
        zmax = float(z64.abs().max())
        enc_err_h = float((z_h[b].double() - z64).abs().max())
        enc_err_o = float((z_o[b].double() - z64).abs().max())
        enc_bound = 5.0 * 2.0 ** -16 * zmax
        for bb, t in mism:
            if bb != b:
                continue
            i, j = int(idx_o[b, t]), int(idx_h[b, t])
            de = cb64[j] - cb64[i]
            en = float(cb64[j].pow(2).sum() - cb64[i].pow(2).sum())
            exact = lambda z: en - 2.0 * float(z.double() @ de)   # noqa: E731   D_j - D_i for an encoder output z, float64
            m64 = exact(z64[t])
            u = ulp32(float(dist_o[b, t, i]))
            rec = {"image": b, "token": t, "idx_oracle": i, "idx_hip": j, "d": float(dist_o[b, t, i]), "ulp": u}
            top2 = torch.topk(dist_o[b, t], 2, largest=False).values
            rec["oracle_top2_margin_ulp"] = float(top2[1] - top2[0]) / u
            rec["exact_margin_ulp"] = m64 / u                     # > 0: exact arithmetic sides with the oracle, < 0: with the HIP path
            good = True
            for tag, dist, z in (("oracle", dist_o, z_o), ("hip", dist_h, z_h)):
                g = float(dist[b, t, j]) - float(dist[b, t, i])
                shift = exact(z[b, t]) - m64
                r = g - exact(z[b, t])
                S = float(z[b, t].double().pow(2).sum()) + max(float(cb64[i].pow(2).sum()), float(cb64[j].pow(2).sum())) + \
                    2.0 * max(float((z[b, t].double() * cb64[i]).abs().sum()), float((z[b, t].double() * cb64[j]).abs().sum()))
                rb = g2 * S
                rec[tag] = {"g_ulp": g / u, "shift_ulp": shift / u, "rounding_ulp": r / u, "rounding_bound_ulp": rb / u}
                good &= abs(r) <= rb
            good &= float(dist_o[b, t, j]) >= float(dist_o[b, t, i]) and float(dist_h[b, t, j]) <= float(dist_h[b, t, i])
            good &= enc_err_h <= enc_bound and enc_err_o <= enc_bound
            rec["encoder_err_over_bound"] = {"hip": enc_err_h / enc_bound, "oracle": enc_err_o / enc_bound, "bound_abs": enc_bound}
            rec["accepted"] = bool(good)
            ok &= good
            recs.append(rec)
    return recs, ok


def format_records(recs) -> str:
    lines = []
    for r in recs:
        if "error" in r:
            lines.append(r["error"])
            continue
        lines.append(
            f"  image {r['image']} token {r['token']}: oracle {r['idx_oracle']} / HIP {r['idx_hip']}, d = {r['d']:.4f} (ulp {r['ulp']:.2e}); oracle's "
            f"f32 top-2 margin {r['oracle_top2_margin_ulp']:.1f} ulp; exact (float64) margin {r['exact_margin_ulp']:+.2f} ulp; "
            f"oracle: g {r['oracle']['g_ulp']:+.1f} = exact {r['exact_margin_ulp']:+.2f} + encoder {r['oracle']['shift_ulp']:+.2f} + rounding "
            f"{r['oracle']['rounding_ulp']:+.2f} (bound {r['oracle']['rounding_bound_ulp']:.0f}); HIP: g {r['hip']['g_ulp']:+.1f} = exact + encoder "
            f"{r['hip']['shift_ulp']:+.2f} + rounding {r['hip']['rounding_ulp']:+.2f}; encoder max-error / bound: HIP "
            f"{r['encoder_err_over_bound']['hip']:.3f}, oracle {r['encoder_err_over_bound']['oracle']:.3f}; "
            f"{'accepted' if r['accepted'] else 'REJECTED'}")
    return "\n".join(lines)
