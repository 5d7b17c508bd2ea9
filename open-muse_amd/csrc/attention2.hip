// Self-attention of ONE-TILE heads (225 <= S <= 260 tokens, head_dim 48: the class-conditional MaskGit transformer at S = 257,
// muse/modeling_transformer.py:190-241) on 32 x 32 MFMA blocks, forward and a single fused backward kernel.  Same C-ABI entry
// points and the same arithmetic contract as attention.hip (bf16 in / out, f32 softmax and accumulation, P rounded to bf16 for
// the second product, lse = log sum exp of the scaled scores); attention.hip keeps every other shape.
//
// Why a second kernel (profiles/r04_attn_pmc.txt, r04_attn_whatif.txt): the 16-row design reads every K / V element from LDS 17
// times per head (one pass per 16-query tile: ~940 KB of LDS reads per head), pays 1125 VALU instructions per 153 MFMAs per wave
// (per-16-key online-softmax bookkeeping, 4-lane row reductions) and re-fetches Q / K / V / dO from HBM in two backward kernels.
// Here:
//   * v_mfma_f32_32x32x16_bf16 with the scores TRANSPOSED (S^T = K Q^T): a lane owns ONE query column and 16 keys of each
//     32-key block, so row maxima / sums are plain register chains + one cross-half exchange, and head_dim 48 = 3 K-steps of 16
//     exactly (no K = 32 + K = 16 pair with its wait states).  A 32-row block re-uses each LDS operand row twice as often.
//   * exact (non-online) softmax: the 8-9 score blocks of a query block stay in registers (two waves per SIMD, up to 256 VGPRs per
//     wave), so there is no per-tile rescale of the output accumulator.
//   * P goes from the score registers to the B operand of the second product WITHOUT lane exchanges: register r of a lane is
//     key-slot pi(r) of the block, and the A operand (V^T, read with ds_read_b64_tr_b16) is addressed through the same
//     permutation pi - a contraction does not care about the order of its slots.  pi is chosen so that BOTH read patterns of an
//     image are bank-conflict free at a row stride of head_dim * 2 + 16 bytes (see perm32).
//   * one head per 4-wave workgroup, two workgroups per CU, operand images by LDS-DMA (buffer_load ... lds, no register pass); the
//     co-resident workgroups are not synchronised with each other, so one's fetch runs under the other's math (see the comment at
//     `constexpr int MAXSH` for the persistent prefetching form this replaced).  Consecutive heads run on one XCD (the 16 heads of
//     an image share the 128-byte lines of its packed qkv rows).
//   * backward as ONE kernel, two phases per head over the same LDS images: (1) queries stationary -> dQ (K, V images; writes
//     lse / dsum per query to LDS), (2) keys stationary -> dK, dV (Q, dO images).  Q, K, V, dO, O are read from HBM once.
//   * the 257th token: query block 8 / key block 8 hold ONE real row.  Every wave takes a slice of that block's streamed
//     dimension and the partial results meet in LDS (a few hundred floats).
#include "attention_blocks.h"

namespace attn2 {

// Two workgroups share a CU and every workgroup takes the same time: started together they stay in the same phase (both fetching,
// then both computing) for the whole launch.  The workgroups of the SECOND dispatch round-robin (blockIdx ncu .. 2 ncu - 1: the
// dispatcher fills one slot per CU before it starts on the second - observed, used for speed only) therefore wait `cycles` before
// their first instruction; every later workgroup starts when a slot frees, i.e. already out of phase.
__device__ __forceinline__ void stagger_start(int ncu, int cycles) {
  if (cycles > 0 && (int)blockIdx.x >= ncu && (int)blockIdx.x < 2 * ncu) {
    const long t0 = (long)__builtin_amdgcn_s_memtime();
    while ((long)__builtin_amdgcn_s_memtime() - t0 < (long)cycles) __builtin_amdgcn_s_sleep(32);
  }
}

#ifdef ATT2_TS   // timing experiments (scripts/exp/attn2_ts.py): s_memtime stamps per wave
__device__ long* g_ts = nullptr;
#define ATT2_STAMP(K) do { __builtin_amdgcn_sched_barrier(0); if (g_ts && g.lane == 0) g_ts[((long)blockIdx.x * (blockDim.x >> 6) + g.wave) * 16 + (K)] = (long)__builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define ATT2_STAMP(K)
#endif

// Workgroup = ONE head, NW = 4 waves (one per SIMD), two workgroups per CU (64 KiB of images each, 256 VGPRs per wave): the two
// co-resident workgroups are not synchronised with each other, so one's operand fetch, softmax VALU or store burst runs under the
// other's MFMAs.  (First form of this file: one persistent 8-wave workgroup per CU with the next head's K / V prefetched by DMA -
// every wave of the CU then sat in the same phase at the same time: a 6 k-cycle vector-memory issue burst, 3.6 k cycles of MFMA with
// idle VALU, 3.6 k of VALU with idle MFMA, per head; 39 us against 42 for attention.hip, profiles/r05_attn2_v1_timeline.txt.)
constexpr int MAXSH = 4;                      // real rows of the 9th block (S - 256)
constexpr int FWD_PW = 52;                    // floats per (wave, query) of the shared block's partials: O[48] | m | l | pad
constexpr int BWD_PW = 100;                   // phase 1 dQ[48]; phase 2 dK[48] | dV[48] (+ pad)

// B operand rows of a block whose only real rows are the first `nreal` (the 9th block): the other lanes load nothing
template <int HD>
__device__ __forceinline__ FragB<HD> load_fragb_few(rsrc_t rs, unsigned ld_bytes, int blk, int nreal, const LaneGeom& g) {
  FragB<HD> r;
#pragma unroll
  for (int ks = 0; ks < Cfg<HD>::KS; ++ks) r.f[ks] = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});
  if (g.n < nreal) r = load_fragb<HD>(rs, ld_bytes, blk, g);
  return r;
}
template <int HD>
__device__ __forceinline__ void write_partial(float* mine, const f32x16 (&acc)[Cfg<HD>::NDB], int h) {
#pragma unroll
  for (int db = 0; db < Cfg<HD>::NDB; ++db)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4)
      if (32 * db + 8 * q4 < HD)
        *(f32x4*)(mine + 32 * db + 8 * q4 + 4 * h) = f32x4{acc[db][4 * q4], acc[db][4 * q4 + 1], acc[db][4 * q4 + 2], acc[db][4 * q4 + 3]};
}

// =================================================================================================================
// forward
// =================================================================================================================
template <int HD, bool TAIL>
__global__ __launch_bounds__(NW * 64, 2) void fwd_kernel(const AttnParams P, int ncu, int stagger) {
  using C = Cfg<HD>;
  constexpr int NKB = TAIL ? 9 : 8;           // key blocks (block 8 holds S - 256 <= MAXSH keys)
  constexpr int NFULL = 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Kimg = smem;
  unsigned char* Vimg = smem + C::IMG;
  float* scratch = (float*)(smem + 2 * C::IMG);
  const LaneGeom g = make_geom<HD>();
  stagger_start(ncu, stagger);
  const int head = xcd_remap();
  const int b = head / P.nh, hh = head - b * P.nh;
  const int S = P.sq;
  const int nsh = S - 256;                                    // TAIL: real rows of block 8
  const int last_valid = S - 32 * (NKB - 1);                  // real keys of the last key block
  const unsigned ldq_b = (unsigned)P.ldq * 2, ldo_b = (unsigned)P.ldo * 2;
  const float c = P.alpha * LOG2E;
  const rsrc_t rq = make_rsrc(P.q + b * P.bq + hh * HD, (unsigned)((S - 1) * P.ldq + HD) * 2);
  const rsrc_t ro = make_rsrc(P.out + b * P.bo + hh * HD, (unsigned)((S - 1) * P.ldo + HD) * 2);

  ATT2_STAMP(0);
  dma_image<HD>(Kimg, P.k + b * P.bk + hh * HD, (unsigned)((S - 1) * P.ldk + HD) * 2, (unsigned)P.ldk * 2, S, g);
  dma_image<HD>(Vimg, P.v + b * P.bv + hh * HD, (unsigned)((S - 1) * P.ldv + HD) * 2, (unsigned)P.ldv * 2, S, g);
  FragB<HD> qf = load_fragb<HD>(rq, ldq_b, g.wave, g);
  ATT2_STAMP(1);
  static_assert(Cfg<HD>::NDMA % NW == 0, "every wave issues the same number of DMA pieces per image: the counted wait below relies on it");
  // K has landed (mine: every operation but the 8 V pieces and the 3 Q loads issued behind it; then everyone's); V may still be in flight
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Cfg<HD>::NDMA / NW + Cfg<HD>::KS) : "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  ATT2_STAMP(2);

  // ---- own query blocks (wave, wave + 4): scores against every key block, exact softmax, P V ----
#pragma unroll 1
  for (int i = 0; i < 2; ++i) {
    const int qb = g.wave + NW * i;
    FragB<HD> qnext;
    if (i == 0) qnext = load_fragb<HD>(rq, ldq_b, g.wave + NW, g);
    else if (TAIL) qnext = load_fragb_few<HD>(rq, ldq_b, 8, nsh, g);     // the 9th block's rows, for the shared pass below
    f32x16 s[NFULL];
    float st0 = NEG_BIG, st1 = NEG_BIG;   // TAIL: the two score registers of block 8 that can hold a real key (rows h and 2 + h)
#pragma unroll
    for (int kb = 0; kb < NFULL; ++kb)
      s[kb] = mma_rows<HD>(Kimg, g, kb, qf, (!TAIL && kb == NFULL - 1) ? mask16(last_valid, g.h) : zero16());
    if (TAIL) {
      const f32x16 t = mma_rows<HD>(Kimg, g, 8, qf, mask16(last_valid, g.h));
      st0 = t[0]; st1 = t[4];
    }
    if (i == 0) ATT2_STAMP(5);
    float m = fmaxf(st0, st1);
#pragma unroll
    for (int kb = 0; kb < NFULL; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, s[kb][r]);
    m = fmaxf(m, xhalf(m));
    const float mc = m * c;
    float l = 0.f;
    if (i == 0) {   // V has landed (the first block's scores and row maxima ran under its flight)
      wait_all_and_barrier();
      ATT2_STAMP(6);
    }
    f32x16 o[C::NDB];
#pragma unroll
    for (int db = 0; db < C::NDB; ++db) o[db] = zero16();
#pragma unroll
    for (int kb = 0; kb < NFULL; ++kb) {      // (a block's exponentials sit next to the previous block's P V in program order)
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[kb][r] = __builtin_amdgcn_exp2f(fmaf(s[kb][r], c, -mc)); l += s[kb][r]; }
      mma_seq<HD>(Vimg, g, kb, s[kb], 2, o);
    }
    if (TAIL) {
      f32x16 pt = zero16();
      pt[0] = __builtin_amdgcn_exp2f(fmaf(st0, c, -mc));
      pt[4] = __builtin_amdgcn_exp2f(fmaf(st1, c, -mc));
      l += pt[0] + pt[4];
      mma_seq<HD>(Vimg, g, 8, pt, 1, o);
    }
    l += xhalf(l);
    if (i == 0) ATT2_STAMP(7);
    u32x4 rows[C::NST];
    pack_rows<HD>(o, 1.0f / l, rows);
    store_rows<HD>(ro, ldo_b, qb, rows, g);
    if (i == 0) ATT2_STAMP(8);
    if (g.h == 0 && qb * 32 + g.n < S) P.lse[(long)head * P.sqp + qb * 32 + g.n] = m * P.alpha + __logf(l);
    qf = qnext;
  }
  ATT2_STAMP(3);

  // ---- the 9th query block (S - 256 real rows): wave w takes key blocks 2 w, 2 w + 1 (wave 3 also block 8), partials meet in LDS ----
  if (TAIL) {
    const bool with_tail = g.wave == NW - 1;
    f32x16 sa = mma_rows<HD>(Kimg, g, 2 * g.wave, qf, zero16());
    f32x16 sb = mma_rows<HD>(Kimg, g, 2 * g.wave + 1, qf, zero16());
    float st0 = NEG_BIG, st1 = NEG_BIG;
    if (with_tail) {
      const f32x16 t = mma_rows<HD>(Kimg, g, 8, qf, mask16(last_valid, g.h));
      st0 = t[0]; st1 = t[4];
    }
    float m = fmaxf(st0, st1);
#pragma unroll
    for (int r = 0; r < 16; ++r) m = fmaxf(m, fmaxf(sa[r], sb[r]));
    m = fmaxf(m, xhalf(m));
    const float mc = m * c;
    float l = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sa[r] = __builtin_amdgcn_exp2f(fmaf(sa[r], c, -mc));
      sb[r] = __builtin_amdgcn_exp2f(fmaf(sb[r], c, -mc));
      l += sa[r] + sb[r];
    }
    f32x16 o[C::NDB];
#pragma unroll
    for (int db = 0; db < C::NDB; ++db) o[db] = zero16();
    mma_seq<HD>(Vimg, g, 2 * g.wave, sa, 2, o);
    mma_seq<HD>(Vimg, g, 2 * g.wave + 1, sb, 2, o);
    if (with_tail) {
      f32x16 pt = zero16();
      pt[0] = __builtin_amdgcn_exp2f(fmaf(st0, c, -mc));
      pt[4] = __builtin_amdgcn_exp2f(fmaf(st1, c, -mc));
      l += pt[0] + pt[4];
      mma_seq<HD>(Vimg, g, 8, pt, 1, o);
    }
    l += xhalf(l);
    if (g.n < nsh) {
      float* mine = scratch + (g.wave * MAXSH + g.n) * FWD_PW;
      write_partial<HD>(mine, o, g.h);
      if (g.h == 0) { mine[HD] = m; mine[HD + 1] = l; }
    }
    lds_barrier();
    if (g.wave == 0) {
      for (int idx = g.lane; idx < nsh * HD; idx += 64) {
        const int q = idx / HD, d = idx - q * HD;
        float mg = NEG_BIG;
#pragma unroll
        for (int w = 0; w < NW; ++w) mg = fmaxf(mg, scratch[(w * MAXSH + q) * FWD_PW + HD]);
        float lsum = 0.f, osum = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          const float* pw = scratch + (w * MAXSH + q) * FWD_PW;
          const float f = __builtin_amdgcn_exp2f((pw[HD] - mg) * c);
          lsum = fmaf(pw[HD + 1], f, lsum);
          osum = fmaf(pw[d], f, osum);
        }
        P.out[b * P.bo + (long)(256 + q) * P.ldo + hh * HD + d] = f32_to_bf16(osum / lsum);
        if (d == 0) P.lse[(long)head * P.sqp + 256 + q] = mg * P.alpha + __logf(lsum);
      }
    }
  }
  ATT2_STAMP(4);
}

// =================================================================================================================
// backward: one kernel, two phases per head over the same two LDS images
// =================================================================================================================
// ATT2_BWD_UNROLL: unroll factor of the backward kernel's block loops (0 = fully unrolled)
#ifndef ATT2_BWD_UNROLL
#define ATT2_BWD_UNROLL 0
#endif
#if ATT2_BWD_UNROLL == 0
#define ATT2_BLOCK_LOOP _Pragma("unroll")
#elif ATT2_BWD_UNROLL == 1
#define ATT2_BLOCK_LOOP _Pragma("unroll 1")
#elif ATT2_BWD_UNROLL == 2
#define ATT2_BLOCK_LOOP _Pragma("unroll 2")
#else
#define ATT2_BLOCK_LOOP _Pragma("unroll 3")
#endif

// P = exp2(s c - lse2), dS = P (dp - dsum) for the 16 registers of a block; lse2 / dsum either per lane (phase 1: the lane's query)
// or per register (phase 2: the register's query)
// (dp arrives as dP - dsum: the dP product accumulates onto a C operand that holds -dsum)
template <bool PER_REG>
__device__ __forceinline__ void p_and_ds(f32x16& s, f32x16& dp, float c, const f32x16& l2) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float pv = __builtin_amdgcn_exp2f(fmaf(s[r], c, -(PER_REG ? l2[r] : l2[0])));
    dp[r] = pv * dp[r];
    s[r] = pv;
  }
}

template <int HD, bool TAIL, int NWB>
__global__ __launch_bounds__(NWB * 64, NWB == 8 ? 4 : 2) void bwd_kernel(const AttnParams P, int ncu, int stagger) {
  using C = Cfg<HD>;
  constexpr int BPW = 8 / NWB;   // own blocks per wave and phase; key / query blocks of the shared (9th) block per wave
  constexpr int NB = TAIL ? 9 : 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* imgA = smem;                           // phase 1: K, phase 2: Q
  unsigned char* imgB = smem + C::IMG;                  // phase 1: V, phase 2: dO
  float* L2 = (float*)(smem + 2 * C::IMG);              // lse * log2(e) per query, in MFMA-row order: position 32 blk + i <-> query 32 blk + perm32(i)
  float* DS = L2 + C::ROWS;                             // -dsum, same order
  float* scratch = DS + C::ROWS;
  const LaneGeom g = make_geom<HD>();
  stagger_start(ncu, stagger);
  const int head = xcd_remap();
  const int b = head / P.nh, hh = head - b * P.nh;
  const int S = P.sq;
  const int nsh = S - 256;
  const int last_valid = S - 32 * (NB - 1);
  const float c = P.alpha * LOG2E;
  auto rsrc_of = [&](const bf16_t* base, long bs, long ld) { return make_rsrc(base + b * bs + hh * HD, (unsigned)((S - 1) * ld + HD) * 2); };
  auto dma_of = [&](unsigned char* img, const bf16_t* base, long bs, long ld) {
    dma_image<HD, NWB>(img, base + b * bs + hh * HD, (unsigned)((S - 1) * ld + HD) * 2, (unsigned)ld * 2, S, g);
  };
  const rsrc_t rq = rsrc_of(P.q, P.bq, P.ldq), rk = rsrc_of(P.k, P.bk, P.ldk), rv = rsrc_of(P.v, P.bv, P.ldv), ro = rsrc_of(P.o, P.bo, P.ldo),
               rdo = rsrc_of(P.d_o, P.bdo, P.lddo);
  const unsigned ldq_b = (unsigned)P.ldq * 2, ldk_b = (unsigned)P.ldk * 2, ldv_b = (unsigned)P.ldv * 2, ldo_b = (unsigned)P.ldo * 2,
                 lddo_b = (unsigned)P.lddo * 2, lddq_b = (unsigned)P.lddq * 2, lddk_b = (unsigned)P.lddk * 2, lddv_b = (unsigned)P.lddv * 2;
  // lse2 / dsum of the queries blk * 32 + n (this lane's query), from global memory
  auto query_consts = [&](int blk, const FragB<HD>& dof, const FragB<HD>& of, float& l2, float& dsm) {
    const int q = blk * 32 + g.n;
    const float lse = q < S ? P.lse[(long)head * P.sqp + q] : 0.f;
    float d = 0.f;
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) d = fmaf((float)dof.f[ks][e], (float)of.f[ks][e], d);
    d += xhalf(d);
    dsm = d;
    l2 = q < S ? lse * LOG2E : 1e30f;      // rows past the end: P = exp2(.. - 1e30) = 0
  };

  // ================= phase 1: queries stationary -> dQ; K, V images =================
  ATT2_STAMP(0);
  dma_of(imgA, P.k, P.bk, P.ldk);
  dma_of(imgB, P.v, P.bv, P.ldv);
  ATT2_STAMP(1);
  wait_all_and_barrier();
  ATT2_STAMP(2);
  // one query block against one key block: dq += (P (dP - dsum)) K
  auto k_block = [&](int kb, bool last, const FragB<HD>& qf, const FragB<HD>& dof, const f32x16& l2v, const f32x16& dsv, f32x16 (&dq)[C::NDB]) {
    f32x16 s = mma_rows<HD>(imgA, g, kb, qf, last ? mask16(last_valid, g.h) : zero16());
    f32x16 dp = mma_rows<HD>(imgB, g, kb, dof, dsv);      // dsv = -dsum in every register
    p_and_ds<false>(s, dp, c, l2v);
    mma_seq<HD>(imgA, g, kb, dp, (last && TAIL) ? 1 : 2, dq);
  };
  // (Measured and dropped, profiles/r05_attn2_steps.txt: two key blocks per step with S / dP of both as four independent accumulator
  //  chains and dQ split over two accumulator sets - 256 VGPRs with 10 spilled, 103 us against 91-96; the operand rows of both own
  //  blocks fetched under the DMA's flight - 214 VGPRs, 93-99 us: no gain.)
#pragma unroll 1
  for (int i = 0; i < BPW; ++i) {
    const int qb = g.wave + NWB * i;
    const FragB<HD> qf = load_fragb<HD>(rq, ldq_b, qb, g);
    const FragB<HD> dof = load_fragb<HD>(rdo, lddo_b, qb, g);
    f32x16 l2v, dsv;
    {
      const FragB<HD> of = load_fragb<HD>(ro, ldo_b, qb, g);
      float l2, dsm;
      query_consts(qb, dof, of, l2, dsm);
      if (g.h == 0) { L2[qb * 32 + perm32_inv(g.n)] = l2; DS[qb * 32 + perm32_inv(g.n)] = -dsm; }   // (DS holds -dsum)
      l2v[0] = l2;
#pragma unroll
      for (int r = 0; r < 16; ++r) dsv[r] = -dsm;
    }
    f32x16 dq[C::NDB];
#pragma unroll
    for (int db = 0; db < C::NDB; ++db) dq[db] = zero16();
    ATT2_BLOCK_LOOP
    for (int kb = 0; kb < NB; ++kb) k_block(kb, kb == NB - 1, qf, dof, l2v, dsv, dq);
    u32x4 rows[C::NST];
    pack_rows<HD>(dq, P.alpha, rows);
    store_rows<HD>(rsrc_of(P.dq, P.bdq, P.lddq), lddq_b, qb, rows, g);
    if (i == 0) ATT2_STAMP(3);
  }
  ATT2_STAMP(4);
  if (TAIL) {   // query block 8: wave w against BPW key blocks (the last wave also block 8); partial dQ rows through LDS
    const FragB<HD> qf = load_fragb_few<HD>(rq, ldq_b, 8, nsh, g);
    const FragB<HD> dof = load_fragb_few<HD>(rdo, lddo_b, 8, nsh, g);
    const FragB<HD> of = load_fragb_few<HD>(ro, ldo_b, 8, nsh, g);
    float l2, dsm;
    query_consts(8, dof, of, l2, dsm);
    if (g.wave == 0 && g.h == 0) { L2[256 + perm32_inv(g.n)] = l2; DS[256 + perm32_inv(g.n)] = -dsm; }
    f32x16 l2v, dsv;
    l2v[0] = l2;
#pragma unroll
    for (int r = 0; r < 16; ++r) dsv[r] = -dsm;
    f32x16 dq[C::NDB];
#pragma unroll
    for (int db = 0; db < C::NDB; ++db) dq[db] = zero16();
#pragma unroll
    for (int j = 0; j < BPW; ++j) k_block(BPW * g.wave + j, false, qf, dof, l2v, dsv, dq);
    if (g.wave == NWB - 1) k_block(8, true, qf, dof, l2v, dsv, dq);
    if (g.n < nsh) write_partial<HD>(scratch + (g.wave * MAXSH + g.n) * BWD_PW, dq, g.h);
  }
  ATT2_STAMP(5);
  lds_barrier();              // everybody is done with the K, V images; L2 / DS and the partials are written
  ATT2_STAMP(6);
  dma_of(imgA, P.q, P.bq, P.ldq);
  dma_of(imgB, P.d_o, P.bdo, P.lddo);
  ATT2_STAMP(7);
  if (TAIL && g.wave == 0) {
    for (int idx = g.lane; idx < nsh * HD; idx += 64) {
      const int q = idx / HD, d = idx - q * HD;
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < NWB; ++w) a += scratch[(w * MAXSH + q) * BWD_PW + d];
      P.dq[b * P.bdq + (long)(256 + q) * P.lddq + hh * HD + d] = f32_to_bf16(a * P.alpha);
    }
  }
  // ================= phase 2: keys stationary -> dK, dV; Q, dO images =================
  wait_all_and_barrier();     // (also: wave 0 has read the phase-1 partials before anybody writes the phase-2 ones)
  ATT2_STAMP(8);
  auto q_block = [&](int qb, const FragB<HD>& kf, const FragB<HD>& vf, int t2, f32x16 (&dk)[C::NDB], f32x16 (&dv)[C::NDB]) {
    f32x16 l2v, dsv;
#pragma unroll
    for (int T = 0; T < 4; ++T) {
      const f32x4 a = *(const f32x4*)(L2 + qb * 32 + 8 * T + 4 * g.h);
      const f32x4 d = *(const f32x4*)(DS + qb * 32 + 8 * T + 4 * g.h);
#pragma unroll
      for (int j = 0; j < 4; ++j) { l2v[4 * T + j] = a[j]; dsv[4 * T + j] = d[j]; }
    }
    f32x16 s = mma_rows<HD>(imgA, g, qb, kf, zero16());
    f32x16 dp = mma_rows<HD>(imgB, g, qb, vf, dsv);       // -dsum of the registers' queries as the C operand
    p_and_ds<true>(s, dp, c, l2v);
    mma_seq<HD>(imgB, g, qb, s, t2, dv);
    mma_seq<HD>(imgA, g, qb, dp, t2, dk);
  };
#pragma unroll 1
  for (int i = 0; i < BPW; ++i) {
    const int kb = g.wave + NWB * i;
    const FragB<HD> kf = load_fragb<HD>(rk, ldk_b, kb, g);
    const FragB<HD> vf = load_fragb<HD>(rv, ldv_b, kb, g);
    f32x16 dk[C::NDB], dv[C::NDB];
#pragma unroll
    for (int db = 0; db < C::NDB; ++db) { dk[db] = zero16(); dv[db] = zero16(); }
    ATT2_BLOCK_LOOP
    for (int qb = 0; qb < NB; ++qb) q_block(qb, kf, vf, (TAIL && qb == NB - 1) ? 1 : 2, dk, dv);
    u32x4 rows[C::NST];
    pack_rows<HD>(dk, P.alpha, rows);
    store_rows<HD>(rsrc_of(P.dk, P.bdk, P.lddk), lddk_b, kb, rows, g);
    pack_rows<HD>(dv, 1.0f, rows);
    store_rows<HD>(rsrc_of(P.dv, P.bdv, P.lddv), lddv_b, kb, rows, g);
    if (i == 0) ATT2_STAMP(9);
  }
  ATT2_STAMP(10);
  if (TAIL) {   // key block 8: wave w against BPW query blocks (the last wave also block 8); partial dK / dV rows through LDS
    const FragB<HD> kf = load_fragb_few<HD>(rk, ldk_b, 8, nsh, g);
    const FragB<HD> vf = load_fragb_few<HD>(rv, ldv_b, 8, nsh, g);
    f32x16 dk[C::NDB], dv[C::NDB];
#pragma unroll
    for (int db = 0; db < C::NDB; ++db) { dk[db] = zero16(); dv[db] = zero16(); }
#pragma unroll
    for (int j = 0; j < BPW; ++j) q_block(BPW * g.wave + j, kf, vf, 2, dk, dv);
    if (g.wave == NWB - 1) q_block(8, kf, vf, 1, dk, dv);
    if (g.n < nsh) {
      float* mine = scratch + (g.wave * MAXSH + g.n) * BWD_PW;
      write_partial<HD>(mine, dk, g.h);
      write_partial<HD>(mine + HD, dv, g.h);
    }
    lds_barrier();
    if (g.wave == 0) {
      for (int idx = g.lane; idx < nsh * HD; idx += 64) {
        const int k = idx / HD, d = idx - k * HD;
        float a = 0.f, e = 0.f;
#pragma unroll
        for (int w = 0; w < NWB; ++w) { a += scratch[(w * MAXSH + k) * BWD_PW + d]; e += scratch[(w * MAXSH + k) * BWD_PW + HD + d]; }
        P.dk[b * P.bdk + (long)(256 + k) * P.lddk + hh * HD + d] = f32_to_bf16(a * P.alpha);
        P.dv[b * P.bdv + (long)(256 + k) * P.lddv + hh * HD + d] = f32_to_bf16(e);
      }
    }
  }
  ATT2_STAMP(11);
}

static bool enabled() {
  const char* e = getenv("MUSE_ATTN2");      // read per call: tests compare both kernel families in one process
  return !(e && e[0] == '0');
}
static int num_cus() {
  static int v = []() {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { (void)hipGetLastError(); return 256; }
    return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }();
  return v;
}
static int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
static bool shape_ok(const AttnParams& P, int head_dim) {
  return enabled() && head_dim == 48 && P.sq == P.skv && P.sq >= 225 && P.sq <= 256 + MAXSH;
}

}  // namespace attn2

#ifdef ATT2_TS
extern "C" int muse_dbg_attn2_ts(void* p) { long* v = (long*)p; return (int)hipMemcpyToSymbol(HIP_SYMBOL(attn2::g_ts), &v, sizeof(v)); }
#endif

int attn2_fwd_try(const AttnParams& P, int head_dim, int batch, hipStream_t st) {
  using namespace attn2;
  if (!shape_ok(P, head_dim)) return 0;
  constexpr int HD = 48;
  const size_t lds = 2 * (size_t)Cfg<HD>::IMG + NW * MAXSH * FWD_PW * 4;
  auto launch = [&](auto k) {
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(batch * P.nh), dim3(NW * 64), lds, st, P, num_cus(), env_int("MUSE_ATTN2_STAGGER_FWD", 6000));
  };
  if (P.sq > 256) launch(fwd_kernel<HD, true>);
  else launch(fwd_kernel<HD, false>);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 1 : -(int)e;
}

int attn2_bwd_try(const AttnParams& P, int head_dim, int batch, hipStream_t st) {
  using namespace attn2;
  if (!shape_ok(P, head_dim)) return 0;
  constexpr int HD = 48;
  constexpr int NWB = ATT2_BWD_NW;
  const size_t lds = 2 * (size_t)Cfg<HD>::IMG + 2 * Cfg<HD>::ROWS * 4 + NWB * MAXSH * BWD_PW * 4;
  auto launch = [&](auto k) {
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(batch * P.nh), dim3(NWB * 64), lds, st, P, num_cus(), env_int("MUSE_ATTN2_STAGGER_BWD", 0));
  };
  if (P.sq > 256) launch(bwd_kernel<HD, true, NWB>);
  else launch(bwd_kernel<HD, false, NWB>);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 1 : -(int)e;
}
