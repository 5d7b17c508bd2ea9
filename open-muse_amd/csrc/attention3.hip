// Fused attention of the "bf16x3" compute mode: f32 tensors in and out, every product (Q K^T, P V, and the five of the backward) as
// three bf16 MFMA products of hi / lo operand planes with f32 accumulation (hi = bf16(x), lo = bf16(x - hi); hi*hi + hi*lo + lo*hi,
// the dropped lo*lo term is 2^-16 relative) - the TF32-class regime configs/cc12m_uvit_clip.yaml:102-103 trains in (f32 tensors,
// enable_tf32), on hardware without an xf32 MFMA.  Replaces, for that mode, the materialised route of the tape engines
// (muse/modeling_transformer_v2.py:881-889 -> reference Attention.attention: baddbmm -> softmax -> matmul and their autograd
// backward): six batched exact-f32 products + two softmax passes over a [B * heads, 256, S_kv] f32 score tensor, 24 % of a
// config-4 step at the YAML's precision (profiles/r05_c4_kernel_stats_before.csv).
//
// Shapes: head_dim 64, 256 queries (the 16 x 16 latent grid of config 4), S_kv <= 256 keys in whole-kernel template steps
// (8 key blocks: self-attention; 3: the 77 text states of the cross-attention).  Everything else: MUSE_ERR_UNSUPPORTED, the
// caller keeps the materialised route.
//
// Built from attention2.hip's pieces (attention_blocks.h): 32 x 32 x 16 MFMA blocks with the scores TRANSPOSED, exact softmax
// with a query block's score blocks in registers, P / dS from the score registers into the B operand of the next product without a
// lane exchange (row permutation pi), operand images with a row stride of head_dim * 2 + 16 bytes that are bank-conflict free for
// both read patterns.  What differs:
//   * the operand images exist twice (hi plane, lo plane) and are WRITTEN BY THE KERNEL: the f32 rows are loaded to registers,
//     split (common.h split4 - the same split muse_split_f32_to_bf16x2 and the GEMMs' operand images use) and stored; no pre-pass
//     over q / k / v in HBM, no LDS-DMA;
//   * a block product is three MFMAs per K step (lo*hi, hi*lo, hi*hi: small terms first); P and dS are split in registers;
//   * four planes of 256 rows are 147 KB: ONE 8-wave workgroup per CU and head, each wave owns one 32-query block (forward,
//     backward phase 1) / one 32-key block (backward phase 2).
// H16 (round 6, the "f16" compute mode: muse_operand_images(1, ..)): the same kernels with ONE IEEE-half plane per operand and one
// v_mfma_f32_32x32x16_f16 per K step - half's 10-bit mantissa is the TF32 operand format, like the mode's GEMMs.  Two planes are 74 KB:
// two workgroups share a CU and overlap each other's phases.  Gradient operands (dO, dS) are converted times the pass's power-of-two
// gradient scale S and the results handed back divided by it: dO S, -dsum S -> S (dP - dsum) -> S dS -> S dQ, S dK; P^T (dO S) = S dV.
#include "attention_blocks.h"
#include "../../include/muse_hip.h"

namespace attn3 {

using namespace attn2;

constexpr int HD = 64;
using C = Cfg<HD>;                       // KS = 4, NDB = 2, STR = 144 (9 slots of 16 bytes)
constexpr int NW3 = 8;                   // waves per workgroup
constexpr int NT = NW3 * 64;
constexpr int SQ = 256;                  // queries
constexpr int PLANE = SQ * C::STR;       // bytes of one 256-row plane

struct Params {
  const float *q, *k, *v, *o, *d_o;
  float *out, *dq, *dk, *dv, *lse;
  long ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;     // elements between consecutive tokens
  long bq, bk, bv, bo, bdo, bdq, bdk, bdv;             // elements between consecutive images
  int nh, skv;
  float alpha;
  // optional: the same results ALSO as (hi, lo) bf16 operand planes for the products that read them (muse_gemm_x3): hi plane pointers
  // addressed like the f32 tensors (same strides, in elements), the lo plane lo_* elements behind
  // ("f16" compute mode, img_format(): lo_* = -1 and the pointers receive ONE IEEE-half image half(x * img_scale) instead)
  bf16_t *outp, *dqp, *dkp, *dvp;
  long lo_out, lo_dq, lo_dk, lo_dv;
  float img_scale;
  int* img_stats;
  float gscale;        // H16 backward: the power of two gradient operands are scaled by before their conversion to half (else 1)
  // streaming forms (longer sequences: whole multiples of 256 queries / keys; q, k, v, o, dO, dq, dk, dv are the FULL-sequence tensors):
  int nqb, nkj;        // query blocks (= gridDim.y of the forward / dQ kernels) and key blocks (= gridDim.y of the dK / dV kernel)
  float* dsum;         // [nqb][B * nh][256]: dO . O per query, written by the dQ kernel, read by the dK / dV kernel
};

typedef _Float16 f16x8_ __attribute__((ext_vector_type(8)));
// eight f32 values (times s) as one half fragment
__device__ __forceinline__ bf16x8 half8(const f32x4& a, const f32x4& b, float s) {
  const f16x8_ h = {(_Float16)(a[0] * s), (_Float16)(a[1] * s), (_Float16)(a[2] * s), (_Float16)(a[3] * s),
                    (_Float16)(b[0] * s), (_Float16)(b[1] * s), (_Float16)(b[2] * s), (_Float16)(b[3] * s)};
  return __builtin_bit_cast(bf16x8, h);
}
template <bool H16> __device__ __forceinline__ f32x16 mfma32(const bf16x8& a, const bf16x8& b, const f32x16& c) {
  if constexpr (H16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_, a), __builtin_bit_cast(f16x8_, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

struct FragB3 { bf16x8 h[C::KS], l[C::KS]; };     // B operand of a head-dim contraction, both planes

__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& lo) {
  u32x2 h0, l0, h1, l1;
  split4(__builtin_bit_cast(u32x4, a), h0, l0);
  split4(__builtin_bit_cast(u32x4, b), h1, l1);
  hi = __builtin_bit_cast(bf16x8, u32x4{h0[0], h0[1], h1[0], h1[1]});
  lo = __builtin_bit_cast(bf16x8, u32x4{l0[0], l0[1], l1[0], l1[1]});
}

// the raw f32 values lane (n, h) supplies as B operand of row `row`: columns 16 ks + 8 h .. + 7
struct Raw { f32x4 a[C::KS], b[C::KS]; };
__device__ __forceinline__ Raw load_raw(const float* rowp, bool valid, int h) {
  Raw r;
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) {
    if (valid) {
      r.a[ks] = *(const f32x4*)(rowp + 16 * ks + 8 * h);
      r.b[ks] = *(const f32x4*)(rowp + 16 * ks + 8 * h + 4);
    } else {
      r.a[ks] = f32x4{0.f, 0.f, 0.f, 0.f};
      r.b[ks] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  return r;
}
template <bool H16 = false>
__device__ __forceinline__ FragB3 split_raw(const Raw& r, float s = 1.f) {
  FragB3 f;
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) {
    if constexpr (H16) f.h[ks] = half8(r.a[ks], r.b[ks], s);
    else split8(r.a[ks], r.b[ks], f.h[ks], f.l[ks]);
  }
  return f;
}

// rows 0 .. nrows-1 of two [.][64] f32 slices -> their (hi, lo) image planes; rows at or past `valid` are zeros.  Every load of both
// slices is in flight before the first conversion.  nrows * 8 sixteen-byte slots per plane, NT threads: slot s = row * 8 + chunk.
// (H16: one half plane each, the lo pointers are not touched; sb scales the second slice - dO times the gradient scale)
template <int NROWS, bool H16 = false>
__device__ __forceinline__ void fill_two(unsigned char* ah, unsigned char* al, const float* a, long lda, unsigned char* bh, unsigned char* bl,
                                         const float* b, long ldb, int valid, float sb = 1.f) {
  constexpr int PER = (NROWS * 8 + NT - 1) / NT;
  f32x4 va[PER][2], vb[PER][2];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int s = (int)threadIdx.x + i * NT, row = s >> 3, c = s & 7;
    if (s < NROWS * 8 && row < valid) {
      const float* pa = a + (long)row * lda + c * 8;
      const float* pb = b + (long)row * ldb + c * 8;
      va[i][0] = *(const f32x4*)pa; va[i][1] = *(const f32x4*)(pa + 4);
      vb[i][0] = *(const f32x4*)pb; vb[i][1] = *(const f32x4*)(pb + 4);
    } else {
      va[i][0] = va[i][1] = vb[i][0] = vb[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int s = (int)threadIdx.x + i * NT, row = s >> 3, c = s & 7;
    if (s < NROWS * 8) {
      if constexpr (H16) {
        *(bf16x8*)(ah + row * C::STR + c * 16) = half8(va[i][0], va[i][1], 1.f);
        *(bf16x8*)(bh + row * C::STR + c * 16) = half8(vb[i][0], vb[i][1], sb);
      } else {
        bf16x8 hi, lo;
        split8(va[i][0], va[i][1], hi, lo);
        *(bf16x8*)(ah + row * C::STR + c * 16) = hi;
        *(bf16x8*)(al + row * C::STR + c * 16) = lo;
        split8(vb[i][0], vb[i][1], hi, lo);
        *(bf16x8*)(bh + row * C::STR + c * 16) = hi;
        *(bf16x8*)(bl + row * C::STR + c * 16) = lo;
      }
    }
  }
}

// acc += rows(blk) of the image x fragment, three products per K step
template <bool H16 = false>
__device__ __forceinline__ f32x16 mma_rows3(const unsigned char* ih, const unsigned char* il, const LaneGeom& g, int blk, const FragB3& b, f32x16 acc) {
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) {
    const bf16x8 ah = frag_rows<HD>(ih, g, blk, ks);
    if constexpr (H16) {
      acc = mfma32<true>(ah, b.h[ks], acc);
    } else {
      const bf16x8 al = frag_rows<HD>(il, g, blk, ks);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b.h[ks], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b.l[ks], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b.h[ks], acc, 0, 0, 0);
    }
  }
  return acc;
}
// registers 8 t .. 8 t + 7 of x as B-operand slots, both planes
__device__ __forceinline__ void pack8x3(const f32x16& x, int t, bf16x8& hi, bf16x8& lo) {
  union { uint32_t w[4]; bf16x8 b; } uh, ul;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float x0 = x[8 * t + 2 * e], x1 = x[8 * t + 2 * e + 1];
    const uint32_t w = pack2_bf16(x0, x1);
    uh.w[e] = w;
    ul.w[e] = pack2_bf16(x0 - __uint_as_float(w << 16), x1 - __uint_as_float(w & 0xffff0000u));
  }
  hi = uh.b; lo = ul.b;
}
// acc[db] += img^T[:, rows of blk] * x[rows, :]   (sequence contraction, transposed image reads)
template <bool H16 = false>
__device__ __forceinline__ void mma_seq3(const unsigned char* ih, const unsigned char* il, const LaneGeom& g, int blk, const f32x16& x,
                                         f32x16 (&acc)[C::NDB]) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    bf16x8 xh, xl;
    if constexpr (H16) {
      xh = half8(f32x4{x[8 * t], x[8 * t + 1], x[8 * t + 2], x[8 * t + 3]}, f32x4{x[8 * t + 4], x[8 * t + 5], x[8 * t + 6], x[8 * t + 7]}, 1.f);
    } else {
      pack8x3(x, t, xh, xl);
    }
#pragma unroll
    for (int db = 0; db < C::NDB; ++db) {
      const bf16x8 ah = frag_tr<HD>(ih, g, blk, t, db);
      if constexpr (H16) {
        acc[db] = mfma32<true>(ah, xh, acc[db]);
      } else {
        const bf16x8 al = frag_tr<HD>(il, g, blk, t, db);
        acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, xh, acc[db], 0, 0, 0);
        acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xl, acc[db], 0, 0, 0);
        acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xh, acc[db], 0, 0, 0);
      }
    }
  }
}
// acc[db][r] = value of row n at column 32 db + 8 (r >> 2) + 4 h + (r & 3): 16-byte stores; planes (optional): the same four values
// split into the hi / lo planes (8-byte stores), bit for bit what muse_split_f32_to_bf16x2 makes of the f32 result
__device__ __forceinline__ void store_rows3(float* rowp, const f32x16 (&acc)[C::NDB], float scale, int h, bf16_t* hip = nullptr, long lo_off = 0,
                                            float img_scale = 1.f, int* img_stats = nullptr) {
#pragma unroll
  for (int db = 0; db < C::NDB; ++db)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const f32x4 v = {acc[db][4 * q4] * scale, acc[db][4 * q4 + 1] * scale, acc[db][4 * q4 + 2] * scale, acc[db][4 * q4 + 3] * scale};
      if (rowp) *(f32x4*)(rowp + 32 * db + 8 * q4 + 4 * h) = v;
      if (hip) store_image4(hip + 32 * db + 8 * q4 + 4 * h, lo_off, img_scale, img_stats, v[0], v[1], v[2], v[3]);
    }
}

// =================================================================================================================
// forward: wave w owns queries 32 w .. 32 w + 31
// =================================================================================================================
template <int NKB, bool H16 = false>
__global__ __launch_bounds__(NT, 2) void fwd_kernel(const Params P) {
  constexpr int PL = NKB * 32 * C::STR;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *Kh = smem, *Kl = smem + PL, *Vh = smem + (H16 ? 1 : 2) * PL, *Vl = smem + 3 * PL;   // (H16: two planes, the lo pointers unused)
  const LaneGeom g = make_geom<HD>();
  const int head = xcd_remap();
  const int b = head / P.nh, hh = head - b * P.nh;
  const int last_valid = P.skv - 32 * (NKB - 1);
  const float c = P.alpha * LOG2E;
  const int q = g.wave * 32 + g.n;

  const Raw qr = load_raw(P.q + b * P.bq + (long)q * P.ldq + hh * HD, true, g.h);
  fill_two<NKB * 32, H16>(Kh, Kl, P.k + b * P.bk + hh * HD, P.ldk, Vh, Vl, P.v + b * P.bv + hh * HD, P.ldv, P.skv);
  const FragB3 qf = split_raw<H16>(qr);
  lds_barrier();

  f32x16 s[NKB];
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) s[kb] = mma_rows3<H16>(Kh, Kl, g, kb, qf, kb == NKB - 1 ? mask16(last_valid, g.h) : zero16());
  float m = NEG_BIG;
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) m = fmaxf(m, s[kb][r]);
  m = fmaxf(m, xhalf(m));
  const float mc = m * c;
  float l = 0.f;
  f32x16 o[C::NDB];
#pragma unroll
  for (int db = 0; db < C::NDB; ++db) o[db] = zero16();
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[kb][r] = __builtin_amdgcn_exp2f(fmaf(s[kb][r], c, -mc)); l += s[kb][r]; }
    mma_seq3<H16>(Vh, Vl, g, kb, s[kb], o);
  }
  l += xhalf(l);
  const long oo = b * P.bo + (long)q * P.ldo + hh * HD;
  store_rows3(P.out + oo, o, 1.0f / l, g.h, P.outp ? P.outp + oo : nullptr, P.lo_out, P.img_scale, P.img_stats);
  if (g.h == 0) P.lse[(long)head * SQ + q] = m * P.alpha + __logf(l);
}

// =================================================================================================================
// backward: phase 1 queries stationary (K, V planes) -> dQ; phase 2 keys stationary (Q, dO planes) -> dK, dV
// =================================================================================================================
template <bool PER_REG>
__device__ __forceinline__ void p_and_ds(f32x16& s, f32x16& dp, float c, const f32x16& l2) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float pv = __builtin_amdgcn_exp2f(fmaf(s[r], c, -(PER_REG ? l2[r] : l2[0])));
    dp[r] = pv * dp[r];       // (dp arrives as dP - dsum: the dP product accumulates onto a C operand that holds -dsum)
    s[r] = pv;
  }
}

template <int NKB, bool H16 = false>
__global__ __launch_bounds__(NT, 2) void bwd_kernel(const Params P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *Ah = smem, *Al = smem + PLANE, *Bh = smem + (H16 ? 1 : 2) * PLANE, *Bl = smem + 3 * PLANE;   // phase 1: K, V   phase 2: Q, dO
  float* L2 = (float*)(smem + (H16 ? 2 : 4) * PLANE);      // lse * log2(e) per query in MFMA-row order: position 32 blk + i <-> query 32 blk + perm32(i)
  const float S = H16 ? P.gscale : 1.f;        // gradient operands are converted times S (see the header)
  float* DS = L2 + SQ;                         // -dsum, same order
  const LaneGeom g = make_geom<HD>();
  const int head = xcd_remap();
  const int b = head / P.nh, hh = head - b * P.nh;
  const int last_valid = P.skv - 32 * (NKB - 1);
  const float c = P.alpha * LOG2E;
  const float* qh = P.q + b * P.bq + hh * HD;
  const float* kh = P.k + b * P.bk + hh * HD;
  const float* vh = P.v + b * P.bv + hh * HD;
  const float* doh = P.d_o + b * P.bdo + hh * HD;

  // ---------------- phase 1 ----------------
  {
    const int q = g.wave * 32 + g.n;
    const Raw qr = load_raw(qh + (long)q * P.ldq, true, g.h);
    const Raw dor = load_raw(doh + (long)q * P.lddo, true, g.h);
    const Raw orr = load_raw(P.o + b * P.bo + (long)q * P.ldo + hh * HD, true, g.h);
    const float lse = P.lse[(long)head * SQ + q];
    fill_two<NKB * 32, H16>(Ah, Al, kh, P.ldk, Bh, Bl, vh, P.ldv, P.skv);
    float d = 0.f;
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks)
#pragma unroll
      for (int e = 0; e < 4; ++e) { d = fmaf(dor.a[ks][e], orr.a[ks][e], d); d = fmaf(dor.b[ks][e], orr.b[ks][e], d); }
    d += xhalf(d);
    d *= S;
    const float l2 = lse * LOG2E;
    if (g.h == 0) { L2[g.wave * 32 + perm32_inv(g.n)] = l2; DS[g.wave * 32 + perm32_inv(g.n)] = -d; }
    const FragB3 qf = split_raw<H16>(qr), dof = split_raw<H16>(dor, S);
    f32x16 l2v, dsv;
    l2v[0] = l2;
#pragma unroll
    for (int r = 0; r < 16; ++r) dsv[r] = -d;
    lds_barrier();
    f32x16 dq[C::NDB];
#pragma unroll
    for (int db = 0; db < C::NDB; ++db) dq[db] = zero16();
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      f32x16 s = mma_rows3<H16>(Ah, Al, g, kb, qf, kb == NKB - 1 ? mask16(last_valid, g.h) : zero16());
      f32x16 dp = mma_rows3<H16>(Bh, Bl, g, kb, dof, dsv);
      p_and_ds<false>(s, dp, c, l2v);
      mma_seq3<H16>(Ah, Al, g, kb, dp, dq);
    }
    const long oq = b * P.bdq + (long)q * P.lddq + hh * HD;
    store_rows3(P.dq ? P.dq + oq : nullptr, dq, P.alpha / S, g.h, P.dqp ? P.dqp + oq : nullptr, P.lo_dq, P.img_scale, P.img_stats);
  }
  lds_barrier();       // everybody is done with the K, V planes; L2 / DS are written
  // ---------------- phase 2 ----------------
  {
    const int key = g.wave * 32 + g.n;
    const bool own = g.wave < NKB;                       // (wave-uniform)
    const bool valid = own && key < P.skv;
    const Raw kr = load_raw(kh + (long)key * P.ldk, valid, g.h);
    const Raw vr = load_raw(vh + (long)key * P.ldv, valid, g.h);
    fill_two<SQ, H16>(Ah, Al, qh, P.ldq, Bh, Bl, doh, P.lddo, SQ, S);
    const FragB3 kf = split_raw<H16>(kr), vf = split_raw<H16>(vr);
    lds_barrier();
    if (own) {
      f32x16 dk[C::NDB], dv[C::NDB];
#pragma unroll
      for (int db = 0; db < C::NDB; ++db) { dk[db] = zero16(); dv[db] = zero16(); }
#pragma unroll
      for (int qb = 0; qb < SQ / 32; ++qb) {
        f32x16 l2v, dsv;
#pragma unroll
        for (int T = 0; T < 4; ++T) {
          const f32x4 a = *(const f32x4*)(L2 + qb * 32 + 8 * T + 4 * g.h);
          const f32x4 dd = *(const f32x4*)(DS + qb * 32 + 8 * T + 4 * g.h);
#pragma unroll
          for (int j = 0; j < 4; ++j) { l2v[4 * T + j] = a[j]; dsv[4 * T + j] = dd[j]; }
        }
        f32x16 s = mma_rows3<H16>(Ah, Al, g, qb, kf, zero16());
        f32x16 dp = mma_rows3<H16>(Bh, Bl, g, qb, vf, dsv);
        p_and_ds<true>(s, dp, c, l2v);
        mma_seq3<H16>(Bh, Bl, g, qb, s, dv);
        mma_seq3<H16>(Ah, Al, g, qb, dp, dk);
      }
      if (valid) {
        const long ok = b * P.bdk + (long)key * P.lddk + hh * HD, ov = b * P.bdv + (long)key * P.lddv + hh * HD;
        store_rows3(P.dk ? P.dk + ok : nullptr, dk, P.alpha / S, g.h, P.dkp ? P.dkp + ok : nullptr, P.lo_dk, P.img_scale, P.img_stats);
        store_rows3(P.dv ? P.dv + ov : nullptr, dv, 1.0f / S, g.h, P.dvp ? P.dvp + ov : nullptr, P.lo_dv, P.img_scale, P.img_stats);
      }
    }
  }
}

// =================================================================================================================
// Streaming forms for sequences of whole 256-row blocks (round 6: BASELINE config 4's 1024 tokens).  The one-tile kernels above, run per
// (query block, key block) pair, reload q / k / v for every pair and leave partial results to be merged; here a workgroup keeps its 256
// queries (forward, dQ) or its 256 keys (dK / dV) and STREAMS the other side's 256-row blocks through the same LDS planes:
//   forward   online softmax - running row maximum m and sum l per query, the accumulated P V rescaled by exp2((m_old - m_new) c) when a
//             block raises the maximum; one pass, the context and its log-sum-exp written once (no partials, no merge kernel)
//   dQ        phase 1 of bwd_kernel per key block with the GLOBAL log-sum-exp: dQ accumulates over the key blocks; also writes dO . O
//   dK / dV   phase 2 of bwd_kernel per query block (Q, dO planes; lse and -dO.O of the block from the arrays the other kernels wrote)
// S and dP are computed twice (once per kernel: 7 block products instead of 5) and every gradient is written once (no partial stacks).
// =================================================================================================================
template <bool H16>
__global__ __launch_bounds__(NT, 2) void fwd_stream_kernel(const Params P) {
  constexpr int NKB = 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *Kh = smem, *Kl = smem + PLANE, *Vh = smem + (H16 ? 1 : 2) * PLANE, *Vl = smem + 3 * PLANE;
  const LaneGeom g = make_geom<HD>();
  const int head = xcd_remap(), qi = blockIdx.y;
  const int b = head / P.nh, hh = head - b * P.nh;
  const float c = P.alpha * LOG2E;
  const int q = qi * SQ + g.wave * 32 + g.n;
  const Raw qr = load_raw(P.q + b * P.bq + (long)q * P.ldq + hh * HD, true, g.h);
  const FragB3 qf = split_raw<H16>(qr);
  float m = NEG_BIG, l = 0.f;
  f32x16 o[C::NDB];
#pragma unroll
  for (int db = 0; db < C::NDB; ++db) o[db] = zero16();
  for (int kj = 0; kj < P.nkj; ++kj) {
    if (kj) lds_barrier();                      // every wave is done with the previous key block's planes
    fill_two<SQ, H16>(Kh, Kl, P.k + b * P.bk + (long)kj * SQ * P.ldk + hh * HD, P.ldk, Vh, Vl, P.v + b * P.bv + (long)kj * SQ * P.ldv + hh * HD, P.ldv, SQ);
    lds_barrier();
    f32x16 s[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) s[kb] = mma_rows3<H16>(Kh, Kl, g, kb, qf, zero16());
    float mj = m;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mj = fmaxf(mj, s[kb][r]);
    mj = fmaxf(mj, xhalf(mj));
    const float resc = __builtin_amdgcn_exp2f((m - mj) * c);     // (first block: exp2(-huge) = 0 on zeros)
    m = mj;
    l *= resc;
#pragma unroll
    for (int db = 0; db < C::NDB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[db][r] *= resc;
    const float mc = m * c;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[kb][r] = __builtin_amdgcn_exp2f(fmaf(s[kb][r], c, -mc)); l += s[kb][r]; }
      mma_seq3<H16>(Vh, Vl, g, kb, s[kb], o);
    }
  }
  l += xhalf(l);
  const long oo = b * P.bo + (long)q * P.ldo + hh * HD;
  store_rows3(P.out + oo, o, 1.0f / l, g.h, P.outp ? P.outp + oo : nullptr, P.lo_out, P.img_scale, P.img_stats);
  if (g.h == 0) P.lse[((long)qi * gridDim.x + head) * SQ + g.wave * 32 + g.n] = m * P.alpha + __logf(l);
}

template <bool H16>
__global__ __launch_bounds__(NT, 2) void bwd_dq_stream_kernel(const Params P) {
  constexpr int NKB = 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *Ah = smem, *Al = smem + PLANE, *Bh = smem + (H16 ? 1 : 2) * PLANE, *Bl = smem + 3 * PLANE;   // K, V of the streamed key block
  const float S = H16 ? P.gscale : 1.f;
  const LaneGeom g = make_geom<HD>();
  const int head = xcd_remap(), qi = blockIdx.y;
  const int b = head / P.nh, hh = head - b * P.nh;
  const float c = P.alpha * LOG2E;
  const int q = qi * SQ + g.wave * 32 + g.n;
  const Raw qr = load_raw(P.q + b * P.bq + (long)q * P.ldq + hh * HD, true, g.h);
  const Raw dor = load_raw(P.d_o + b * P.bdo + (long)q * P.lddo + hh * HD, true, g.h);
  const Raw orr = load_raw(P.o + b * P.bo + (long)q * P.ldo + hh * HD, true, g.h);
  const long li = ((long)qi * gridDim.x + head) * SQ + g.wave * 32 + g.n;
  const float lse = P.lse[li];
  float d = 0.f;
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks)
#pragma unroll
    for (int e = 0; e < 4; ++e) { d = fmaf(dor.a[ks][e], orr.a[ks][e], d); d = fmaf(dor.b[ks][e], orr.b[ks][e], d); }
  d += xhalf(d);
  if (g.h == 0) P.dsum[li] = d;
  d *= S;
  const FragB3 qf = split_raw<H16>(qr), dof = split_raw<H16>(dor, S);
  f32x16 l2v, dsv;
  l2v[0] = lse * LOG2E;
#pragma unroll
  for (int r = 0; r < 16; ++r) dsv[r] = -d;
  f32x16 dq[C::NDB];
#pragma unroll
  for (int db = 0; db < C::NDB; ++db) dq[db] = zero16();
  for (int kj = 0; kj < P.nkj; ++kj) {
    if (kj) lds_barrier();
    fill_two<SQ, H16>(Ah, Al, P.k + b * P.bk + (long)kj * SQ * P.ldk + hh * HD, P.ldk, Bh, Bl, P.v + b * P.bv + (long)kj * SQ * P.ldv + hh * HD, P.ldv, SQ);
    lds_barrier();
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      f32x16 s = mma_rows3<H16>(Ah, Al, g, kb, qf, zero16());
      f32x16 dp = mma_rows3<H16>(Bh, Bl, g, kb, dof, dsv);
      p_and_ds<false>(s, dp, c, l2v);
      mma_seq3<H16>(Ah, Al, g, kb, dp, dq);
    }
  }
  const long oq = b * P.bdq + (long)q * P.lddq + hh * HD;
  store_rows3(P.dq ? P.dq + oq : nullptr, dq, P.alpha / S, g.h, P.dqp ? P.dqp + oq : nullptr, P.lo_dq, P.img_scale, P.img_stats);
}

template <bool H16>
__global__ __launch_bounds__(NT, 2) void bwd_dkv_stream_kernel(const Params P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *Ah = smem, *Al = smem + PLANE, *Bh = smem + (H16 ? 1 : 2) * PLANE, *Bl = smem + 3 * PLANE;   // Q, dO of the streamed query block
  float* L2 = (float*)(smem + (H16 ? 2 : 4) * PLANE);      // lse * log2(e) of the block's queries in MFMA-row order (bwd_kernel)
  float* DS = L2 + SQ;                                     // -dO.O * S, same order
  const float S = H16 ? P.gscale : 1.f;
  const LaneGeom g = make_geom<HD>();
  const int head = xcd_remap(), kj = blockIdx.y;
  const int b = head / P.nh, hh = head - b * P.nh;
  const float c = P.alpha * LOG2E;
  const int key = kj * SQ + g.wave * 32 + g.n;
  const Raw kr = load_raw(P.k + b * P.bk + (long)key * P.ldk + hh * HD, true, g.h);
  const Raw vr = load_raw(P.v + b * P.bv + (long)key * P.ldv + hh * HD, true, g.h);
  const FragB3 kf = split_raw<H16>(kr), vf = split_raw<H16>(vr);
  f32x16 dk[C::NDB], dv[C::NDB];
#pragma unroll
  for (int db = 0; db < C::NDB; ++db) { dk[db] = zero16(); dv[db] = zero16(); }
  for (int qi = 0; qi < P.nqb; ++qi) {
    if (qi) lds_barrier();
    fill_two<SQ, H16>(Ah, Al, P.q + b * P.bq + (long)qi * SQ * P.ldq + hh * HD, P.ldq, Bh, Bl, P.d_o + b * P.bdo + (long)qi * SQ * P.lddo + hh * HD,
                      P.lddo, SQ, S);
    if (threadIdx.x < SQ) {     // query r of the block sits at position 32 (r / 32) + perm32_inv(r % 32)
      const int r = threadIdx.x;
      const long li = ((long)qi * gridDim.x + head) * SQ + r;
      const int pos = (r & ~31) + perm32_inv(r & 31);
      L2[pos] = P.lse[li] * LOG2E;
      DS[pos] = -P.dsum[li] * S;
    }
    lds_barrier();
#pragma unroll
    for (int qb = 0; qb < SQ / 32; ++qb) {
      f32x16 l2v, dsv;
#pragma unroll
      for (int T = 0; T < 4; ++T) {
        const f32x4 a = *(const f32x4*)(L2 + qb * 32 + 8 * T + 4 * g.h);
        const f32x4 dd = *(const f32x4*)(DS + qb * 32 + 8 * T + 4 * g.h);
#pragma unroll
        for (int j = 0; j < 4; ++j) { l2v[4 * T + j] = a[j]; dsv[4 * T + j] = dd[j]; }
      }
      f32x16 s = mma_rows3<H16>(Ah, Al, g, qb, kf, zero16());
      f32x16 dp = mma_rows3<H16>(Bh, Bl, g, qb, vf, dsv);
      p_and_ds<true>(s, dp, c, l2v);
      mma_seq3<H16>(Bh, Bl, g, qb, s, dv);
      mma_seq3<H16>(Ah, Al, g, qb, dp, dk);
    }
  }
  const long ok = b * P.bdk + (long)key * P.lddk + hh * HD, ov = b * P.bdv + (long)key * P.lddv + hh * HD;
  store_rows3(P.dk ? P.dk + ok : nullptr, dk, P.alpha / S, g.h, P.dkp ? P.dkp + ok : nullptr, P.lo_dk, P.img_scale, P.img_stats);
  store_rows3(P.dv ? P.dv + ov : nullptr, dv, 1.0f / S, g.h, P.dvp ? P.dvp + ov : nullptr, P.lo_dv, P.img_scale, P.img_stats);
}

static int nkb_of(int skv) { return (skv + 31) / 32; }
static int check(const muse_attn_desc* d) {
  if (!d || d->seq_q <= 0 || d->seq_kv <= 0 || d->heads <= 0) return MUSE_ERR_BAD_ARG;
  if (d->head_dim != HD || d->seq_q != SQ || d->seq_kv > SQ) return MUSE_ERR_UNSUPPORTED;
  const int nkb = nkb_of(d->seq_kv);
  if (nkb != 8 && nkb != 3) return MUSE_ERR_UNSUPPORTED;
  const int64_t lds[] = {d->ldq, d->ldk, d->ldv, d->ldo, d->bsq, d->bsk, d->bsv, d->bso};
  for (int64_t x : lds) if (x & 3) return MUSE_ERR_ALIGN;     // 16-byte f32 vectors
  const void* ps[] = {d->q, d->k, d->v, d->o};
  for (const void* p : ps) if (((uintptr_t)p) & 15) return MUSE_ERR_ALIGN;
  return 0;
}
static Params base(const muse_attn_desc* d) {
  Params P = {};
  P.q = (const float*)d->q; P.k = (const float*)d->k; P.v = (const float*)d->v;
  P.ldq = d->ldq; P.ldk = d->ldk; P.ldv = d->ldv; P.ldo = d->ldo;
  P.bq = d->bsq; P.bk = d->bsk; P.bv = d->bsv; P.bo = d->bso;
  P.nh = d->heads; P.skv = d->seq_kv; P.alpha = d->alpha;
  return P;
}
template <typename K>
static int launch(K k, const Params& P, int heads_total, size_t lds, hipStream_t st) {
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k, dim3(heads_total), dim3(NT), lds, st, P);
  MUSE_CHECK_LAUNCH();
  return 0;
}

template <typename K>
static int launch2(K k, const Params& P, int heads_total, int ny, size_t lds, hipStream_t st) {
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k, dim3(heads_total, ny), dim3(NT), lds, st, P);
  MUSE_CHECK_LAUNCH();
  return 0;
}
static int check_stream(const muse_attn_desc* d) {
  if (!d || d->seq_q <= 0 || d->seq_kv <= 0 || d->heads <= 0) return MUSE_ERR_BAD_ARG;
  if (d->head_dim != HD || (d->seq_q % SQ) || (d->seq_kv % SQ)) return MUSE_ERR_UNSUPPORTED;
  const int64_t lds[] = {d->ldq, d->ldk, d->ldv, d->ldo, d->bsq, d->bsk, d->bsv, d->bso};
  for (int64_t x : lds) if (x & 3) return MUSE_ERR_ALIGN;
  const void* ps[] = {d->q, d->k, d->v, d->o};
  for (const void* p : ps) if (((uintptr_t)p) & 15) return MUSE_ERR_ALIGN;
  return 0;
}

}  // namespace attn3

extern "C" int muse_attention_x3_fwd(const muse_attn_desc* d, float* lse, void* o_planes, int64_t o_lo, void* stream) {
  using namespace attn3;
  const int rc = check(d);
  if (rc) return rc;
  if (o_planes && ((((uintptr_t)o_planes) & 7) || (o_lo & 3) || o_lo <= 0)) return MUSE_ERR_ALIGN;
  if (d->batch <= 0) return 0;
  Params P = base(d);
  P.out = (float*)d->o; P.lse = lse;
  const ImgFormat f = img_format(false);
  P.outp = (bf16_t*)o_planes; P.lo_out = f.lo_sign < 0 ? -1 : o_lo; P.img_scale = f.scale; P.img_stats = f.stats;
  const int nkb = nkb_of(d->seq_kv);
  if (f.lo_sign < 0) {      // the "f16" compute mode is on: the core's products are single half products like the mode's GEMMs
    const size_t lds = 2 * (size_t)nkb * 32 * C::STR;
    return nkb == 8 ? launch(fwd_kernel<8, true>, P, d->batch * d->heads, lds, (hipStream_t)stream)
                    : launch(fwd_kernel<3, true>, P, d->batch * d->heads, lds, (hipStream_t)stream);
  }
  const size_t lds = 4 * (size_t)nkb * 32 * C::STR;
  return nkb == 8 ? launch(fwd_kernel<8>, P, d->batch * d->heads, lds, (hipStream_t)stream)
                  : launch(fwd_kernel<3>, P, d->batch * d->heads, lds, (hipStream_t)stream);
}

extern "C" int muse_attention_x3_bwd(const muse_attn_desc* d, const void* d_o, int64_t lddo, int64_t bsdo, const float* lse, void* dq,
                                     int64_t lddq, int64_t bsdq, void* dk, int64_t lddk, int64_t bsdk, void* dv, int64_t lddv, int64_t bsdv,
                                     void* dq_planes, int64_t dq_lo, void* dk_planes, int64_t dk_lo, void* dv_planes, int64_t dv_lo,
                                     void* stream) {
  using namespace attn3;
  if (((((uintptr_t)dq_planes) | ((uintptr_t)dk_planes) | ((uintptr_t)dv_planes)) & 7) || ((dq_lo | dk_lo | dv_lo) & 3)) return MUSE_ERR_ALIGN;
  const int rc = check(d);
  if (rc) return rc;
  if ((lddo | bsdo | lddq | bsdq | lddk | bsdk | lddv | bsdv) & 3) return MUSE_ERR_ALIGN;
  if ((((uintptr_t)d_o) | ((uintptr_t)dq) | ((uintptr_t)dk) | ((uintptr_t)dv)) & 15) return MUSE_ERR_ALIGN;
  if ((!dq && !dq_planes) || (!dk && !dk_planes) || (!dv && !dv_planes)) return MUSE_ERR_BAD_ARG;    // (a gradient may exist as planes only)
  if (d->batch <= 0) return 0;
  Params P = base(d);
  P.o = (const float*)d->o; P.d_o = (const float*)d_o; P.lddo = lddo; P.bdo = bsdo;
  P.lse = (float*)lse;
  P.dq = (float*)dq; P.lddq = lddq; P.bdq = bsdq;
  P.dk = (float*)dk; P.lddk = lddk; P.bdk = bsdk;
  P.dv = (float*)dv; P.lddv = lddv; P.bdv = bsdv;
  const ImgFormat f = img_format(true);
  P.dqp = (bf16_t*)dq_planes; P.dkp = (bf16_t*)dk_planes; P.dvp = (bf16_t*)dv_planes; P.img_scale = f.scale; P.img_stats = f.stats;
  P.lo_dq = f.lo_sign < 0 ? -1 : dq_lo; P.lo_dk = f.lo_sign < 0 ? -1 : dk_lo; P.lo_dv = f.lo_sign < 0 ? -1 : dv_lo;
  P.gscale = f.lo_sign < 0 ? f.scale : 1.f;
  if (f.lo_sign < 0) {
    const size_t lds = 2 * (size_t)PLANE + 2 * SQ * sizeof(float);
    return nkb_of(d->seq_kv) == 8 ? launch(bwd_kernel<8, true>, P, d->batch * d->heads, lds, (hipStream_t)stream)
                                  : launch(bwd_kernel<3, true>, P, d->batch * d->heads, lds, (hipStream_t)stream);
  }
  const size_t lds = 4 * (size_t)PLANE + 2 * SQ * sizeof(float);
  return nkb_of(d->seq_kv) == 8 ? launch(bwd_kernel<8>, P, d->batch * d->heads, lds, (hipStream_t)stream)
                                : launch(bwd_kernel<3>, P, d->batch * d->heads, lds, (hipStream_t)stream);
}

// ---- streaming forms: whole multiples of 256 queries AND keys (self-attention of the longer sequences) ---------------------------------------------
// d describes the FULL sequences (seq_q, seq_kv multiples of 256; batch strides of the whole tensors).  lse [seq_q / 256][batch * heads][256].
extern "C" int muse_attention_x3_fwd_stream(const muse_attn_desc* d, float* lse, void* o_planes, int64_t o_lo, void* stream) {
  using namespace attn3;
  const int rc = check_stream(d);
  if (rc) return rc;
  if (o_planes && ((((uintptr_t)o_planes) & 7) || (o_lo & 3) || o_lo <= 0)) return MUSE_ERR_ALIGN;
  if (d->batch <= 0) return 0;
  Params P = base(d);
  P.out = (float*)d->o; P.lse = lse;
  const ImgFormat f = img_format(false);
  P.outp = (bf16_t*)o_planes; P.lo_out = f.lo_sign < 0 ? -1 : o_lo; P.img_scale = f.scale; P.img_stats = f.stats;
  P.nqb = d->seq_q / SQ; P.nkj = d->seq_kv / SQ;
  if (f.lo_sign < 0) return launch2(fwd_stream_kernel<true>, P, d->batch * d->heads, P.nqb, 2 * (size_t)PLANE, (hipStream_t)stream);
  return launch2(fwd_stream_kernel<false>, P, d->batch * d->heads, P.nqb, 4 * (size_t)PLANE, (hipStream_t)stream);
}
// dsum: workspace [seq_q / 256][batch * heads][256] f32 (dO . O per query: written by the dQ pass, read by the dK / dV pass).  Gradients as for
// muse_attention_x3_bwd (f32 tensors and / or operand images; leading dimensions and batch strides of the full-sequence tensors).
extern "C" int muse_attention_x3_bwd_stream(const muse_attn_desc* d, const float* d_o, int64_t lddo, int64_t bsdo, const float* lse, float* dsum,
                                            float* dq, int64_t lddq, int64_t bsdq, float* dk, int64_t lddk, int64_t bsdk, float* dv, int64_t lddv,
                                            int64_t bsdv, void* dq_planes, int64_t dq_lo, void* dk_planes, int64_t dk_lo, void* dv_planes, int64_t dv_lo,
                                            void* stream) {
  using namespace attn3;
  if (((((uintptr_t)dq_planes) | ((uintptr_t)dk_planes) | ((uintptr_t)dv_planes)) & 7) || ((dq_lo | dk_lo | dv_lo) & 3)) return MUSE_ERR_ALIGN;
  const int rc = check_stream(d);
  if (rc) return rc;
  if ((lddo | bsdo | lddq | bsdq | lddk | bsdk | lddv | bsdv) & 3) return MUSE_ERR_ALIGN;
  if ((((uintptr_t)d_o) | ((uintptr_t)dq) | ((uintptr_t)dk) | ((uintptr_t)dv)) & 15) return MUSE_ERR_ALIGN;
  if (!dsum || !lse || (!dq && !dq_planes) || (!dk && !dk_planes) || (!dv && !dv_planes)) return MUSE_ERR_BAD_ARG;
  if (d->batch <= 0) return 0;
  Params P = base(d);
  P.o = (const float*)d->o; P.d_o = d_o; P.lddo = lddo; P.bdo = bsdo;
  P.lse = (float*)lse; P.dsum = dsum;
  P.dq = dq; P.lddq = lddq; P.bdq = bsdq;
  P.dk = dk; P.lddk = lddk; P.bdk = bsdk;
  P.dv = dv; P.lddv = lddv; P.bdv = bsdv;
  const ImgFormat f = img_format(true);
  P.dqp = (bf16_t*)dq_planes; P.dkp = (bf16_t*)dk_planes; P.dvp = (bf16_t*)dv_planes; P.img_scale = f.scale; P.img_stats = f.stats;
  P.lo_dq = f.lo_sign < 0 ? -1 : dq_lo; P.lo_dk = f.lo_sign < 0 ? -1 : dk_lo; P.lo_dv = f.lo_sign < 0 ? -1 : dv_lo;
  P.gscale = f.lo_sign < 0 ? f.scale : 1.f;
  P.nqb = d->seq_q / SQ; P.nkj = d->seq_kv / SQ;
  const bool h16 = f.lo_sign < 0;
  const size_t lds1 = (h16 ? 2 : 4) * (size_t)PLANE, lds2 = lds1 + 2 * SQ * sizeof(float);
  int r = h16 ? launch2(bwd_dq_stream_kernel<true>, P, d->batch * d->heads, P.nqb, lds1, (hipStream_t)stream)
              : launch2(bwd_dq_stream_kernel<false>, P, d->batch * d->heads, P.nqb, lds1, (hipStream_t)stream);
  if (r) return r;
  return h16 ? launch2(bwd_dkv_stream_kernel<true>, P, d->batch * d->heads, P.nkj, lds2, (hipStream_t)stream)
             : launch2(bwd_dkv_stream_kernel<false>, P, d->batch * d->heads, P.nkj, lds2, (hipStream_t)stream);
}

// ---- block-by-block form of the longer sequences (round 6: BASELINE config 4's 1024 tokens; muse/ops.py attention_x3_blocked) -------------------
// The one-tile kernels above run per (256 query rows, <= 256 keys) block pair; these two row kernels put the pieces together.
//
// muse_attention_x3_merge: softmax over ALL keys from the key blocks' partial results.  part[j] [B * S, H] f32 = softmax over key block j times
// its values, lp[j][qi][b * nh + h][r] its log-sum-exp (query block qi, row r of 256): lse = log sum_j exp(lp_j), out = sum_j exp(lp_j - lse) part_j,
// f32 throughout.  Also writes lse [S / 256][B * nh][256] (what the block pairs' backward reads) and - optionally - the (hi, lo) operand planes of out
// for the output projection.  One workgroup per token row, thread t owns four channels of head t / 16 (head_dim 64).
namespace attn3m {
constexpr int MAXB = 8;
struct MergeParams {
  const float* part; const float* lp; float* out; float* lse; bf16_t* planes;
  long part_stride, lp_stride, lo;     // elements between key blocks' partials / their lse arrays; hi -> lo plane distance (-1: half image)
  int nk, B, S, nh;
  float img_scale;
  int* img_stats;
};
__global__ __launch_bounds__(256) void merge_kernel(MergeParams P) {
  const int H = P.nh * 64;
  for (long row = blockIdx.x; row < (long)P.B * P.S; row += gridDim.x) {
    const int b = (int)(row / P.S), s = (int)(row - (long)b * P.S), qi = s >> 8, r = s & 255;
    for (int c = threadIdx.x * 4; c < H; c += 1024) {
      const int h = c >> 6;
      const long li = ((long)qi * P.B * P.nh + (long)b * P.nh + h) * 256 + r;
      float l[MAXB], m = -INFINITY;
#pragma unroll
      for (int j = 0; j < MAXB; ++j)
        if (j < P.nk) { l[j] = P.lp[j * P.lp_stride + li]; m = fmaxf(m, l[j]); }
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < MAXB; ++j)
        if (j < P.nk) sum += expf(l[j] - m);
      const float lse = m + logf(sum);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < MAXB; ++j)
        if (j < P.nk) acc += expf(l[j] - lse) * *(const f32x4*)(P.part + j * P.part_stride + row * H + c);
      *(f32x4*)(P.out + row * H + c) = acc;
      if ((c & 63) == 0) P.lse[li] = lse;
      if (P.planes) store_image4(P.planes + row * H + c, P.lo, P.img_scale, P.img_stats, acc[0], acc[1], acc[2], acc[3]);
    }
  }
}
// muse_sum_parts_strided: out[r, 0..cols) (row pitch ldo; += when accumulate) = sum_j parts[j * part_stride + r * cols + c] in the order j = 0, 1, ...
// - the partial gradients of the block pairs (dq over key blocks, dk / dv over query blocks) into the column block of a packed gradient.
struct SumParams { const float* parts; float* out; long part_stride, rows, ldo; int n, cols, accumulate; };
__global__ __launch_bounds__(256) void sum_parts_kernel(SumParams P) {
  const long per_row = P.cols >> 2, total = P.rows * per_row;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / per_row, c = (i - r * per_row) * 4;
    f32x4 acc = *(const f32x4*)(P.parts + r * P.cols + c);
    for (int j = 1; j < P.n; ++j) acc += *(const f32x4*)(P.parts + j * P.part_stride + r * P.cols + c);
    float* o = P.out + r * P.ldo + c;
    if (P.accumulate) acc += *(const f32x4*)o;
    *(f32x4*)o = acc;
  }
}
}  // namespace attn3m

extern "C" int muse_attention_x3_merge(const float* part, int64_t part_stride, const float* lp, int64_t lp_stride, int32_t nk, float* out, float* lse,
                                       void* out_planes, int64_t out_lo, int32_t batch, int32_t seq, int32_t heads, void* stream) {
  if (nk < 1 || nk > attn3m::MAXB || batch <= 0 || seq <= 0 || (seq & 255) || heads <= 0) return nk < 1 || nk > attn3m::MAXB || (seq & 255) ? MUSE_ERR_UNSUPPORTED : 0;
  if ((((uintptr_t)part) | ((uintptr_t)out)) & 15 || (part_stride & 3) || (out_planes && ((((uintptr_t)out_planes) & 7) || (out_lo & 3) || out_lo <= 0)))
    return MUSE_ERR_ALIGN;
  attn3m::MergeParams P;
  P.part = part; P.lp = lp; P.out = out; P.lse = lse; P.planes = (bf16_t*)out_planes;
  const ImgFormat f = img_format(false);
  P.part_stride = part_stride; P.lp_stride = lp_stride; P.lo = f.lo_sign < 0 ? -1 : out_lo; P.img_scale = f.scale; P.img_stats = f.stats; P.nk = nk; P.B = batch; P.S = seq; P.nh = heads;
  const long rows = (long)batch * seq;
  hipLaunchKernelGGL(attn3m::merge_kernel, dim3((unsigned)(rows < 65536 ? rows : 65536)), dim3(256), 0, (hipStream_t)stream, P);
  return (int)hipGetLastError();
}

extern "C" int muse_sum_parts_strided(const float* parts, int64_t part_stride, int32_t n, int64_t rows, int32_t cols, float* out, int64_t ldo,
                                      int32_t accumulate, void* stream) {
  if (n < 1 || rows <= 0 || cols <= 0) return n < 1 ? MUSE_ERR_BAD_ARG : 0;
  if ((cols & 3) || (ldo & 3) || (part_stride & 3) || ((((uintptr_t)parts) | ((uintptr_t)out)) & 15)) return MUSE_ERR_ALIGN;
  attn3m::SumParams P;
  P.parts = parts; P.out = out; P.part_stride = part_stride; P.rows = rows; P.ldo = ldo; P.n = n; P.cols = cols; P.accumulate = accumulate;
  const long total = rows * (cols >> 2);
  long g = (total + 255) / 256;
  hipLaunchKernelGGL(attn3m::sum_parts_kernel, dim3((unsigned)(g > 16384 ? 16384 : g)), dim3(256), 0, (hipStream_t)stream, P);
  return (int)hipGetLastError();
}
