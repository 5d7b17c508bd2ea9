"""CPU emulation of the lane-level index arithmetic of open-muse_amd/csrc/attention2.hip (forward + fused backward) for ONE head.

Not an oracle and not a product path: a bring-up tool.  It restates, lane by lane, the address formulas of the kernel (perm32, the
ds_read_b128 / ds_read_b64_tr_b16 operand addresses, the register <-> row map of the 32x32x16 MFMA's C/D layout, the
v_permlane32_swap store packing, the L2 / DS array order) on top of the DOCUMENTED instruction semantics
(/opt/skills/guides/cdna_hip_programming.md section 3, T10, T21) and checks the result against plain attention in float64.
A formula that is wrong here is wrong on the GPU; one that is right here can still meet a hardware semantic the guide states
differently - the GPU parity tests decide that (they passed on the first run).  The work split emulated here is the first form's
(eight waves, one query / key block each, the 9th block's streamed dimension cut in eight); the shipped kernels give each of four
waves two blocks and two slices - the lane arithmetic, which is what this file checks, is the same.

    python scripts/exp/attn2_emulate.py [S]
"""
import sys

import numpy as np

HD, STR_E, ROWS = 48, 56, 288          # STR_E = row stride in bf16 elements (112 bytes)
KS, NDB = HD // 16, 2
S = int(sys.argv[1]) if len(sys.argv) > 1 else 257
TAIL = S > 256
NB = 9 if TAIL else 8
nsh = S - 256
last_valid = S - 32 * (NB - 1)
alpha = 1.0 / np.sqrt(np.float32(HD))
rng = np.random.default_rng(0)


def bf16(x):
    """round to bf16 (nearest even), kept as float64 values"""
    x = np.asarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).view(np.float32).astype(np.float64)


def perm32(i):
    return ((i >> 4) << 4) + ((i & 3) << 2) + (((i >> 3) & 1) << 1) + ((i >> 2) & 1)


def perm32_inv(r):
    return ((r >> 4) << 4) + (((r >> 1) & 1) << 3) + ((r & 1) << 2) + ((r >> 2) & 3)


assert sorted(perm32(i) for i in range(32)) == list(range(32)) and all(perm32_inv(perm32(i)) == i for i in range(32))

LANES = np.arange(64)
N_, H_ = LANES & 31, LANES >> 5
P16, M_ = LANES & 15, (LANES >> 4) & 1


def image(x):
    """[S][HD] -> LDS image [ROWS + 8][STR_E] (rows past S and the pad slot are zeros; +8 rows of slack like the 32 KiB region)"""
    img = np.zeros((ROWS + 8, STR_E))
    img[:S, :HD] = x
    return img


def frag_rows(img, blk, ks):
    """A operand of a head-dim contraction: per lane 8 elements"""
    out = np.zeros((64, 8))
    for l in LANES:
        row = blk * 32 + perm32(N_[l])
        out[l] = img[row, ks * 16 + 8 * H_[l]: ks * 16 + 8 * H_[l] + 8]
    return out


def load_fragb(x, blk, ks):
    """B operand from global memory: rows blk*32 + n, elements 16 ks + 8 h .. +7 (rows past S read as zeros)"""
    out = np.zeros((64, 8))
    for l in LANES:
        row = blk * 32 + N_[l]
        if row < S:
            out[l] = x[row, ks * 16 + 8 * H_[l]: ks * 16 + 8 * H_[l] + 8]
    return out


def tr_read(img, byte_addr):
    """ds_read_b64_tr_b16 with per-lane addresses (bytes): in each 16-lane group lane p' fetches 4 consecutive bf16 at its address =
    row (p' >> 2), columns 4 (p' & 3) .. + 3 of a 4 x 16 matrix; lane p receives column p of that matrix (rows 0..3)"""
    flat = img.reshape(-1)
    out = np.zeros((64, 4))
    for grp in range(4):
        Mx = np.zeros((4, 16))
        for pp in range(16):
            l = grp * 16 + pp
            e0 = byte_addr[l] // 2
            Mx[pp >> 2, 4 * (pp & 3): 4 * (pp & 3) + 4] = flat[e0: e0 + 4]
        for p in range(16):
            out[grp * 16 + p] = Mx[:, p]
    return out


TR_OFF = (H_ + 4 * (P16 >> 2)) * (STR_E * 2) + (16 * M_ + 4 * (P16 & 3)) * 2


def frag_tr(img, blk, t, db):
    a = TR_OFF + (blk * 32 + t * 16) * (STR_E * 2) + db * 64
    return np.concatenate([tr_read(img, a), tr_read(img, a + 2 * STR_E * 2)], axis=1)


def mfma32(A, B, C):
    """v_mfma_f32_32x32x16_bf16: A lane (i = l & 31, h) holds k-slots (h, 0..7), B lane (n, h) likewise; D lane (n, h) register r is
    row i = (r & 3) + 8 (r >> 2) + 4 h"""
    Am = np.zeros((32, 16))
    Bm = np.zeros((16, 32))
    for l in LANES:
        Am[N_[l], 8 * H_[l]: 8 * H_[l] + 8] = A[l]
        Bm[8 * H_[l]: 8 * H_[l] + 8, N_[l]] = B[l]
    D = Am @ Bm
    out = C.copy()
    for l in LANES:
        for r in range(16):
            out[l, r] += D[(r & 3) + 8 * (r >> 2) + 4 * H_[l], N_[l]]
    return out


def row_of_reg(r, h):
    return 16 * ((r >> 2) >> 1) + 4 * (r & 3) + 2 * ((r >> 2) & 1) + h


def mask16(valid):
    z = np.zeros((64, 16))
    for l in LANES:
        for r in range(16):
            z[l, r] = 0.0 if row_of_reg(r, 0) + H_[l] < valid else -1e30
    return z


def mma_rows(img, blk, fb, acc):
    for ks in range(KS):
        acc = mfma32(frag_rows(img, blk, ks), fb[ks], acc)
    return acc


def pack8(v, t):
    return bf16(v[:, 8 * t: 8 * t + 8])


def mma_seq(img, blk, x, t2, acc):
    for t in range(t2):
        xb = pack8(x, t)
        for db in range(NDB):
            acc[db] = mfma32(frag_tr(img, blk, t, db), xb, acc[db])


def xhalf(v):
    return v[LANES ^ 32]


def swap32(a, b):
    """v_permlane32_swap vdst=a, src=b: lanes 32-63 of a <-> lanes 0-31 of b"""
    a2, b2 = a.copy(), b.copy()
    a2[32:] = b[:32]
    b2[:32] = a[32:]
    return a2, b2


def pack_and_store(acc, scale, out, blk):
    """pack_rows + store_rows: chunk c = 8 consecutive columns starting at 16 c + 8 h of row blk*32 + n"""
    for c in range(HD // 16):
        db, r0 = c >> 1, 8 * (c & 1)
        x = bf16(acc[db][:, r0: r0 + 4] * scale)        # two u32 = 4 bf16 (x0 | x1)
        y = bf16(acc[db][:, r0 + 4: r0 + 8] * scale)
        x, y = swap32(x, y)
        for l in LANES:
            row = blk * 32 + N_[l]
            if row < S:
                out[row, 16 * c + 8 * H_[l]: 16 * c + 8 * H_[l] + 8] = np.concatenate([x[l], y[l]])


# ---------------------------------------------------------------------------------------------------------------------------
q, k, v, do = (bf16(rng.standard_normal((S, HD))) for _ in range(4))
c = float(alpha) * 1.4426950408889634
sc = (q @ k.T) * float(alpha)
pr = np.exp(sc - sc.max(1, keepdims=True))
pr /= pr.sum(1, keepdims=True)
ref_o = pr @ v
ref_lse = np.log(np.exp(sc - sc.max(1, keepdims=True)).sum(1)) + sc.max(1)
dP = do @ v.T
dsum_ref = (dP * pr).sum(1, keepdims=True)
dS = pr * (dP - dsum_ref)
ref_dq, ref_dk, ref_dv = dS @ k * float(alpha), dS.T @ q * float(alpha), pr.T @ do

Kimg, Vimg, Qimg, Dimg = image(k), image(v), image(q), image(do)
zero = np.zeros((64, 16))

# ---- forward ----
out, lse = np.zeros((S, HD)), np.zeros(ROWS)
scratch = {}
for wave in range(8):
    qf = [load_fragb(q, wave, ks) for ks in range(KS)]
    s = [mma_rows(Kimg, kb, qf, mask16(last_valid) if (not TAIL and kb == 7) else zero.copy()) for kb in range(8)]
    st0 = st1 = np.full(64, -1e30)
    if TAIL:
        t = mma_rows(Kimg, 8, qf, mask16(last_valid))
        st0, st1 = t[:, 0], t[:, 4]
    m = np.maximum(st0, st1)
    for kb in range(8):
        m = np.maximum(m, s[kb].max(1))
    m = np.maximum(m, xhalf(m))
    l = np.zeros(64)
    for kb in range(8):
        s[kb] = np.exp2(s[kb] * c - (m * c)[:, None])
        l += s[kb].sum(1)
    pt = zero.copy()
    if TAIL:
        pt[:, 0] = np.exp2(st0 * c - m * c)
        pt[:, 4] = np.exp2(st1 * c - m * c)
        l += pt[:, 0] + pt[:, 4]
    l = l + xhalf(l)
    o = [zero.copy() for _ in range(NDB)]
    for kb in range(8):
        mma_seq(Vimg, kb, s[kb], 2, o)
    if TAIL:
        mma_seq(Vimg, 8, pt, 1, o)
    pack_and_store(o, (1.0 / l)[:, None], out, wave)
    for ln in LANES:
        if H_[ln] == 0 and wave * 32 + N_[ln] < S:
            lse[wave * 32 + N_[ln]] = m[ln] * float(alpha) + np.log(l[ln])
    if TAIL:
        qs = [load_fragb(q, 8, ks) for ks in range(KS)]
        sa = mma_rows(Kimg, wave, qs, zero.copy())
        st0 = st1 = np.full(64, -1e30)
        if wave == 7:
            t = mma_rows(Kimg, 8, qs, mask16(last_valid))
            st0, st1 = t[:, 0], t[:, 4]
        m = np.maximum(np.maximum(st0, st1), sa.max(1))
        m = np.maximum(m, xhalf(m))
        sa = np.exp2(sa * c - (m * c)[:, None])
        l = sa.sum(1)
        pt = zero.copy()
        if wave == 7:
            pt[:, 0] = np.exp2(st0 * c - m * c)
            pt[:, 4] = np.exp2(st1 * c - m * c)
            l += pt[:, 0] + pt[:, 4]
        l = l + xhalf(l)
        o = [zero.copy() for _ in range(NDB)]
        mma_seq(Vimg, wave, sa, 2, o)
        if wave == 7:
            mma_seq(Vimg, 8, pt, 1, o)
        for ln in LANES:
            if N_[ln] < nsh:
                rec = scratch.setdefault((wave, N_[ln]), np.zeros(52))
                for db in range(NDB):
                    for q4 in range(4):
                        if 32 * db + 8 * q4 < HD:
                            rec[32 * db + 8 * q4 + 4 * H_[ln]: 32 * db + 8 * q4 + 4 * H_[ln] + 4] = o[db][ln, 4 * q4: 4 * q4 + 4]
                if H_[ln] == 0:
                    rec[HD], rec[HD + 1] = m[ln], l[ln]
if TAIL:
    for qq in range(nsh):
        mg = max(scratch[(w, qq)][HD] for w in range(8))
        f = [np.exp2((scratch[(w, qq)][HD] - mg) * c) for w in range(8)]
        ls = sum(scratch[(w, qq)][HD + 1] * f[w] for w in range(8))
        out[256 + qq] = bf16(sum(scratch[(w, qq)][:HD] * f[w] for w in range(8)) / ls)
        lse[256 + qq] = mg * float(alpha) + np.log(ls)


def err(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


print(f"S={S} forward : out {err(out, ref_o):.2e}  lse {err(lse[:S], ref_lse):.2e}")
assert err(out, ref_o) < 1.5e-2 and err(lse[:S], ref_lse) < 1e-3

# ---- backward ----
o_bf = bf16(ref_o)          # the forward's stored output
dq, dk, dv = np.zeros((S, HD)), np.zeros((S, HD)), np.zeros((S, HD))
L2, DSa = np.zeros(ROWS), np.zeros(ROWS)


def query_consts(blk, dof):
    of = [load_fragb(o_bf, blk, ks) for ks in range(KS)]
    d = sum((dof[ks] * of[ks]).sum(1) for ks in range(KS))
    d = d + xhalf(d)
    qrow = blk * 32 + N_
    l2 = np.where(qrow < S, ref_lse[np.minimum(qrow, S - 1)] * 1.4426950408889634, 1e30)
    return l2, d


def p_and_ds(s, dp, l2, dsm):
    pv = np.exp2(s * c - l2)
    return pv, pv * (dp - dsm)


scr = {}
for wave in range(8):
    qf = [load_fragb(q, wave, ks) for ks in range(KS)]
    dof = [load_fragb(do, wave, ks) for ks in range(KS)]
    l2, dsm = query_consts(wave, dof)
    for ln in LANES:
        if H_[ln] == 0:
            L2[wave * 32 + perm32_inv(N_[ln])] = l2[ln]
            DSa[wave * 32 + perm32_inv(N_[ln])] = dsm[ln]
    acc = [zero.copy() for _ in range(NDB)]
    for kb in range(NB):
        last = kb == NB - 1
        s = mma_rows(Kimg, kb, qf, mask16(last_valid) if last else zero.copy())
        dp = mma_rows(Vimg, kb, dof, zero.copy())
        s, dp = p_and_ds(s, dp, l2[:, None], dsm[:, None])
        mma_seq(Kimg, kb, dp, 1 if (last and TAIL) else 2, acc)
    pack_and_store(acc, float(alpha), dq, wave)
    if TAIL:
        qf = [load_fragb(q, 8, ks) for ks in range(KS)]
        dof = [load_fragb(do, 8, ks) for ks in range(KS)]
        l2, dsm = query_consts(8, dof)
        if wave == 0:
            for ln in LANES:
                if H_[ln] == 0:
                    L2[256 + perm32_inv(N_[ln])] = l2[ln]
                    DSa[256 + perm32_inv(N_[ln])] = dsm[ln]
        acc = [zero.copy() for _ in range(NDB)]
        s = mma_rows(Kimg, wave, qf, zero.copy())
        dp = mma_rows(Vimg, wave, dof, zero.copy())
        s, dp = p_and_ds(s, dp, l2[:, None], dsm[:, None])
        mma_seq(Kimg, wave, dp, 2, acc)
        if wave == 7:
            s = mma_rows(Kimg, 8, qf, mask16(last_valid))
            dp = mma_rows(Vimg, 8, dof, zero.copy())
            s, dp = p_and_ds(s, dp, l2[:, None], dsm[:, None])
            mma_seq(Kimg, 8, dp, 1, acc)
        for ln in LANES:
            if N_[ln] < nsh:
                rec = scr.setdefault((wave, N_[ln]), np.zeros(100))
                for db in range(NDB):
                    for q4 in range(4):
                        if 32 * db + 8 * q4 < HD:
                            rec[32 * db + 8 * q4 + 4 * H_[ln]: 32 * db + 8 * q4 + 4 * H_[ln] + 4] = acc[db][ln, 4 * q4: 4 * q4 + 4]
if TAIL:
    for qq in range(nsh):
        dq[256 + qq] = bf16(sum(scr[(w, qq)][:HD] for w in range(8)) * float(alpha))


def q_block(qb, kf, vf, t2, dkk, dvv):
    s = mma_rows(Qimg, qb, kf, zero.copy())
    dp = mma_rows(Dimg, qb, vf, zero.copy())
    l2v, dsv = np.zeros((64, 16)), np.zeros((64, 16))
    for ln in LANES:
        for T in range(4):
            base = qb * 32 + 8 * T + 4 * H_[ln]
            l2v[ln, 4 * T: 4 * T + 4] = L2[base: base + 4]
            dsv[ln, 4 * T: 4 * T + 4] = DSa[base: base + 4]
    s, dp = p_and_ds(s, dp, l2v, dsv)
    mma_seq(Dimg, qb, s, t2, dvv)
    mma_seq(Qimg, qb, dp, t2, dkk)


scr = {}
for wave in range(8):
    kf = [load_fragb(k, wave, ks) for ks in range(KS)]
    vf = [load_fragb(v, wave, ks) for ks in range(KS)]
    dkk, dvv = [zero.copy() for _ in range(NDB)], [zero.copy() for _ in range(NDB)]
    for qb in range(NB):
        q_block(qb, kf, vf, 1 if (TAIL and qb == NB - 1) else 2, dkk, dvv)
    pack_and_store(dkk, float(alpha), dk, wave)
    pack_and_store(dvv, 1.0, dv, wave)
    if TAIL:
        kf = [load_fragb(k, 8, ks) for ks in range(KS)]
        vf = [load_fragb(v, 8, ks) for ks in range(KS)]
        dkk, dvv = [zero.copy() for _ in range(NDB)], [zero.copy() for _ in range(NDB)]
        q_block(wave, kf, vf, 2, dkk, dvv)
        if wave == 7:
            q_block(8, kf, vf, 1, dkk, dvv)
        for ln in LANES:
            if N_[ln] < nsh:
                rec = scr.setdefault((wave, N_[ln]), np.zeros(100))
                for db in range(NDB):
                    for q4 in range(4):
                        if 32 * db + 8 * q4 < HD:
                            o0 = 32 * db + 8 * q4 + 4 * H_[ln]
                            rec[o0: o0 + 4] = dkk[db][ln, 4 * q4: 4 * q4 + 4]
                            rec[HD + o0: HD + o0 + 4] = dvv[db][ln, 4 * q4: 4 * q4 + 4]
if TAIL:
    for kk in range(nsh):
        dk[256 + kk] = bf16(sum(scr[(w, kk)][:HD] for w in range(8)) * float(alpha))
        dv[256 + kk] = bf16(sum(scr[(w, kk)][HD: 2 * HD] for w in range(8)))

print(f"S={S} backward: dq {err(dq, ref_dq):.2e}  dk {err(dk, ref_dk):.2e}  dv {err(dv, ref_dv):.2e}")
assert err(dq, ref_dq) < 3e-2 and err(dk, ref_dk) < 3e-2 and err(dv, ref_dv) < 3e-2

# ---- LDS bank-conflict census of the two read patterns (MI355X_MICROARCH.md section LDS) ----
STRB = STR_E * 2
B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_GROUPS += [[x + 32 for x in grp] for grp in B128_GROUPS]
worst = 0
for ks in range(KS):
    addr = np.array([perm32(N_[ln]) * STRB + 16 * H_[ln] + ks * 32 for ln in LANES])
    for grp in B128_GROUPS:
        slots = [(addr[ln] // 16) % 16 for ln in grp]
        worst = max(worst, max(slots.count(x) for x in set(slots)))
print("ds_read_b128 operand rows : worst multiplicity of a 16-byte slot inside a service group =", worst)
worst = 0
for db in range(NDB):
    for second in (0, 1):
        addr = TR_OFF + db * 64 + second * 2 * STRB
        for half in (range(0, 32), range(32, 64)):
            banks = []
            for ln in half:
                banks += [((addr[ln] + 4 * i) // 4) % 64 for i in range(2)]
            worst = max(worst, max(banks.count(x) for x in set(banks)))
print("ds_read_b64_tr_b16 operand: worst multiplicity of a bank inside a 32-lane group =", worst)
print("OK")
