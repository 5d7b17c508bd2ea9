// Fused full-visibility attention, bf16 in/out, f32 softmax / accumulation, head_dim in {16,32,48,64}; any query / key
// length (self-attention at S = 257 / 256 / 1024 and cross-attention against 77 text tokens are the shapes on the path).
// Replaces Attention.attention (muse/modeling_transformer.py:221-241: baddbmm -> softmax -> matmul), the xformers
// memory_efficient_attention seam (:206-210; muse/modeling_transformer_v2.py:881-889) and their autograd backward
// without ever materialising the S x S matrix in HBM.
//
// Structure (all three kernels): a workgroup owns one (image, head) and a chunk of its "stationary" rows (queries for
// forward / dQ, keys for dK,dV); the other operand streams through LDS in tiles of <= 288 rows as [rows][hd] bf16
// images with a row stride == 32 (mod 64) bytes: the same image is conflict-free for ds_read_b128 / b64 (k = head-dim
// contiguous operand) and for ds_read_b64_tr_b16 (k = sequence operand, hardware transpose).  A tile is fetched with
// ALL of a thread's 16-byte buffer loads in flight at once (out-of-range rows come back as zeros from the buffer
// descriptor: no branches), then written to LDS; two or three workgroups per CU overlap one's fetch with another's math.
// Each wave owns 16-row tiles of the stationary operand.  With more than one streamed tile (S_kv > 288) a wave owns exactly
// one stationary tile and carries its state (online-softmax max / sum / O, or the dQ / dK,dV accumulators) across tiles.
//
// Trick that keeps P in registers: scores are produced TRANSPOSED (S^T = K Q^T, MFMA operands swapped), so a lane holds,
// for its one query (lane & 15), the keys {16t + 4g + r}; two consecutive 16-key tiles therefore give exactly the 8
// k-slots one lane must supply as the B operand of the next MFMA (O^T = V^T P^T) under the k-permutation
// slot(g, j) <-> key 32s + 16*(j>>2) + 4g + (j&3), which the V^T operand reproduces through its tr-read row addresses.
// head_dim 48 contracts as one K=32 MFMA plus one K=16 MFMA (v_mfma_f32_16x16x16_bf16) instead of two half-empty K=32 ones.
#include "common.h"
#include "attention_params.h"
#include "../../include/muse_hip.h"
#include <stdlib.h>

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

template <int HD> struct HeadCfg {
  static constexpr int N32 = HD / 32;               // K = 32 steps of the head-dim contraction
  static constexpr int TAIL = (HD % 32) != 0;       // + one K = 16 step (hd 16, 48)
  static constexpr int ND = HD / 16;                // 16-wide output tiles over the head dim
  static constexpr int CPR = HD / 8;                // 16-byte chunks per row
  // LDS row stride in bytes, == 32 (mod 64): 96 B for hd 48 / 32, 160 B for hd 64, 32 B for hd 16
  static constexpr int RS = ((HD * 2) % 64 == 32) ? HD * 2 : HD * 2 + 32;
};

// one 16-row operand of a head-dim contraction: lane (row = lane & 15, g = lane >> 4) holds columns 32*ks + 8g .. +7 of every
// K = 32 step and columns 32*N32 + 4g .. +3 of the K = 16 tail
template <int HD> struct FragHD {
  bf16x8 f[HeadCfg<HD>::N32 > 0 ? HeadCfg<HD>::N32 : 1];
  bf16x4 t;
};

template <int HD>
__device__ __forceinline__ FragHD<HD> frag_lds(const unsigned char* img, int rowbase, int lane) {
  using C = HeadCfg<HD>;
  FragHD<HD> r;
  const unsigned char* p = img + (rowbase + (lane & 15)) * C::RS;
#pragma unroll
  for (int ks = 0; ks < C::N32; ++ks) r.f[ks] = *(const bf16x8*)(p + (ks * 32 + (lane >> 4) * 8) * 2);
  if constexpr (C::TAIL) r.t = *(const bf16x4*)(p + (C::N32 * 32 + (lane >> 4) * 4) * 2);
  return r;
}
// the same operand straight from global memory through a buffer descriptor (rows outside the tensor read as zeros)
template <int HD>
__device__ __forceinline__ FragHD<HD> frag_global(rsrc_t rs, unsigned ld_bytes, int rowbase, int lane) {
  using C = HeadCfg<HD>;
  FragHD<HD> r;
  const unsigned ro = (unsigned)(rowbase + (lane & 15)) * ld_bytes;
#pragma unroll
  for (int ks = 0; ks < C::N32; ++ks) {
    union { u32x4 u; bf16x8 v; } t;
    t.u = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(ro + (ks * 32 + (lane >> 4) * 8) * 2), 0, 0);
    r.f[ks] = t.v;
  }
  if constexpr (C::TAIL) {
    union { u32x2 u; bf16x4 v; } t;
    t.u = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(ro + (C::N32 * 32 + (lane >> 4) * 4) * 2), 0, 0);
    r.t = t.v;
  }
  return r;
}
// head_dim 48 / 16: a K = 16 MFMA and a K = 32 MFMA must not be chained back to back through SrcC, in either order: ROCm 7.2
// emits them with no wait states in between and gfx950 then returns wrong sums (found by the parity tests on MI355X; the
// hazard tables the compiler consults evidently do not cover the mixed pair).  ATT_TAIL_MODE 1 (default): K = 32 step(s)
// first, 16 wait states, then the K = 16 step accumulates onto the result; 2: separate accumulator + VALU add (VALU is the
// busier pipe of these kernels: 5-7 % slower backward); 3: K = 16 first - WRONG on hardware, kept only as the reproducer.
#ifndef ATT_TAIL_MODE
#define ATT_TAIL_MODE 1
#endif
template <int HD>
__device__ __forceinline__ f32x4 mma_hd(const FragHD<HD>& a, const FragHD<HD>& b, f32x4 acc) {
  using C = HeadCfg<HD>;
  if constexpr (C::TAIL && C::N32 > 0) {
    const s16x4 at = __builtin_bit_cast(s16x4, a.t), bt = __builtin_bit_cast(s16x4, b.t);
#if ATT_TAIL_MODE == 3
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(at, bt, acc, 0, 0, 0);
#pragma unroll
    for (int ks = 0; ks < C::N32; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.f[ks], b.f[ks], acc, 0, 0, 0);
#elif ATT_TAIL_MODE == 1
#pragma unroll
    for (int ks = 0; ks < C::N32; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.f[ks], b.f[ks], acc, 0, 0, 0);
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc));
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(at, bt, acc, 0, 0, 0);
#else
#pragma unroll
    for (int ks = 0; ks < C::N32; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.f[ks], b.f[ks], acc, 0, 0, 0);
    acc += __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(at, bt, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#endif
  } else if constexpr (C::TAIL) {
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a.t), __builtin_bit_cast(s16x4, b.t), acc, 0, 0, 0);
  } else {
#pragma unroll
    for (int ks = 0; ks < C::N32; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.f[ks], b.f[ks], acc, 0, 0, 0);
  }
  return acc;
}
// sum over the head dim of a .* b for this lane's row (partial: the 4 lanes g = 0..3 of a row each hold a quarter)
template <int HD>
__device__ __forceinline__ float dot_hd(const FragHD<HD>& a, const FragHD<HD>& b) {
  using C = HeadCfg<HD>;
  float s = 0.f;
#pragma unroll
  for (int ks = 0; ks < C::N32; ++ks)
#pragma unroll
    for (int j = 0; j < 8; ++j) s = fmaf((float)a.f[ks][j], (float)b.f[ks][j], s);
  if constexpr (C::TAIL)
#pragma unroll
    for (int j = 0; j < 4; ++j) s = fmaf((float)a.t[j], (float)b.t[j], s);
  return s;
}

// Cooperative fetch of `rows` (multiple of 32, <= MAXROWS) rows starting at row0 of one head's [S][HD] slice into an LDS image.
// Rows >= S lie beyond the descriptor's range and arrive as zeros.  All loads of a thread are issued before the first LDS write.
template <int HD, int NT, int MAXROWS>
__device__ __forceinline__ void load_tile(unsigned char* img, rsrc_t rs, unsigned ld_bytes, int row0, int rows) {
  using C = HeadCfg<HD>;
  constexpr int NCH = (MAXROWS * C::CPR + NT - 1) / NT;
  u32x4 v[NCH];
  const int total = rows * C::CPR;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = threadIdx.x + i * NT;
    const int row = c / C::CPR, col = c - row * C::CPR;
    const unsigned off = c < total ? (unsigned)(row0 + row) * ld_bytes + col * 16 : ATT_OOB;
    v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = threadIdx.x + i * NT;
    const int row = c / C::CPR, col = c - row * C::CPR;
    if (c < total) *(u32x4*)(img + row * C::RS + col * 16) = v[i];
  }
}

// A operand with k = sequence (32 rows starting at r0, permuted as described above), i = 16 columns starting at c0
template <int HD>
__device__ __forceinline__ bf16x8 frag_seq(const unsigned char* img, int r0, int c0, int lane) {
  const int p = lane & 15, g = lane >> 4;
  const unsigned char* a0 = img + (r0 + 4 * g + (p >> 2)) * HeadCfg<HD>::RS + (c0 + (p & 3) * 4) * 2;
  union { s16x4 h[2]; bf16x8 v; } u;
  u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
  u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 16 * HeadCfg<HD>::RS));
  return u.v;
}
__device__ __forceinline__ bf16x8 pack8(const f32x4& lo, const f32x4& hi) {
  union { uint32_t w[4]; bf16x8 v; } u;
  u.w[0] = pack2_bf16(lo[0], lo[1]); u.w[1] = pack2_bf16(lo[2], lo[3]);
  u.w[2] = pack2_bf16(hi[0], hi[1]); u.w[3] = pack2_bf16(hi[2], hi[3]);
  return u.v;
}
__device__ __forceinline__ void store4_bf16(bf16_t* p, const f32x4& v, float scale) {
  u32x2 t;
  t[0] = pack2_bf16(v[0] * scale, v[1] * scale);
  t[1] = pack2_bf16(v[2] * scale, v[3] * scale);
  *(u32x2*)p = t;
}
__device__ __forceinline__ float quad_g_sum(float v) {   // sum over the 4 lanes (g = 0..3) that share lane & 15
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float quad_g_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}


// blockIdx -> logical work item such that consecutive logical items (chunks of one head) run on one XCD (block b runs on XCD b % 8)
__device__ __forceinline__ int xcd_remap() {
  const int nb = gridDim.x, id = blockIdx.x, xcd = id & 7, slot = id >> 3, q = nb >> 3, r = nb & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

constexpr float LOG2E = 1.4426950408889634f;
constexpr float NEG_BIG = -1e30f;   // "minus infinity" that survives an MFMA accumulate and an fma without producing NaN

// Work balance.  A head with 257 tokens has 17 16-row tiles; 8 waves would do 3,2,2,... of them.  Instead every wave takes
// floor(tiles / NW) whole tiles and the one left over ("shared" tile) is split over all NW waves along the STREAMED dimension
// (wave w takes the 32-row pairs w, w + NW, ...); the per-wave partial results meet in LDS (the K/V or Q/dO images are dead
// by then and provide the space) and wave 0 combines them: softmax partials by the usual (max, sum, O) merge, gradient
// partials by plain addition.

// =================================================================================================================
// forward: out[b, q, h, :] = softmax(alpha * Q K^T) V ; lse[b*nh + h, q] = log sum exp of the scaled scores
// =================================================================================================================
template <int HD> struct FwdAcc {
  float m;                          // running row maximum of the raw scores
  f32x4 l;                          // running row sum of P (all four entries equal: it comes out of an MFMA against ones)
  f32x4 o[HeadCfg<HD>::ND];         // running P V
};

// One online-softmax update over TN consecutive 16-key tiles starting at tile t0 of the LDS images (TN even):
// scores (keys at tile index >= MASK_FROM within the step get the -1e30 init where they lie past the sequence end),
// row max, rescale, P = exp2(...), O += P V and l += P 1 (the row sum as one more MFMA against a ones operand: VALU is the
// busy pipe of this kernel, the matrix pipe has room).
template <int HD, int TN, int MASK_FROM>
__device__ __forceinline__ void fwd_substep(const unsigned char* Kimg, const unsigned char* Vimg, int t0, const FragHD<HD>& qf,
                                            float c, int kvalid, FwdAcc<HD>& A, int lane) {
  using C = HeadCfg<HD>;
  f32x4 sacc[TN];
  float mt = NEG_BIG;
#pragma unroll
  for (int u = 0; u < TN; ++u) {
    f32x4 ci = {0.f, 0.f, 0.f, 0.f};
    if (u >= MASK_FROM) {
#pragma unroll
      for (int r = 0; r < 4; ++r) ci[r] = (16 * (t0 + u) + r < kvalid) ? 0.f : NEG_BIG;
    }
    sacc[u] = mma_hd<HD>(frag_lds<HD>(Kimg, (t0 + u) * 16, lane), qf, ci);
    mt = fmaxf(mt, fmaxf(fmaxf(sacc[u][0], sacc[u][1]), fmaxf(sacc[u][2], sacc[u][3])));
  }
  mt = quad_g_max(mt);
  const float mn = fmaxf(A.m, mt);
  const float scale = __builtin_amdgcn_exp2f((A.m - mn) * c);   // 0 on the first step (m = -1e30: exp2(-huge) = 0)
  const float mc = mn * c;
  A.m = mn;
#pragma unroll
  for (int u = 0; u < TN; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) sacc[u][r] = __builtin_amdgcn_exp2f(fmaf(sacc[u][r], c, -mc));
  A.l *= scale;
#pragma unroll
  for (int d = 0; d < C::ND; ++d) A.o[d] *= scale;
  const s16x4 one4 = {0x3F80, 0x3F80, 0x3F80, 0x3F80};
  union { s16x4 h[2]; bf16x8 v; } ones;
  ones.h[0] = one4; ones.h[1] = one4;
#pragma unroll
  for (int s = 0; s < TN / 2; ++s) {
    const bf16x8 pb = pack8(sacc[2 * s], sacc[2 * s + 1]);
    A.l = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones.v, pb, A.l, 0, 0, 0);
#pragma unroll
    for (int d = 0; d < C::ND; ++d)
      A.o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_seq<HD>(Vimg, 16 * t0 + 32 * s, 16 * d, lane), pb, A.o[d], 0, 0, 0);
  }
}

template <int HD>
__device__ __forceinline__ void fwd_reset(FwdAcc<HD>& A) {
  A.m = NEG_BIG;
  A.l = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int d = 0; d < HeadCfg<HD>::ND; ++d) A.o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// waves per SIMD to keep the register allocation to: what `blocks` co-resident workgroups of NW waves need (2 by LDS where they fit)
constexpr int fwd_waves_per_simd(int hd_rs, int maxt, int nw) {
  const int lds = 2 * maxt * 16 * hd_rs;
  const int blocks = 163840 / lds >= 2 ? 2 : 1;
  return (blocks * nw + 3) / 4;
}

// MAXT = 16-key tiles per streamed K/V tile (the LDS images always hold MAXT * 16 rows; rows past the sequence end are zeros).
// Keys past the sequence end are masked through the MFMA's C operand (-1e30 instead of 0), which costs nothing for the
// tiles that cannot contain the end: with FULL = false only the last two 16-key tiles of a K/V tile can (the host picks
// MAXT = 2 * ceil(S_kv / 32) for one-tile problems and S_kv % 256 == 0 or > 224 for streamed ones), FULL = true masks every tile.
template <int HD, int MAXT, int NW, bool FULL>
__global__ __launch_bounds__(NW * 64, fwd_waves_per_simd(HeadCfg<HD>::RS, MAXT, NW)) void attn_fwd_kernel(const AttnParams P) {
  using C = HeadCfg<HD>;
  constexpr int NT = NW * 64;
  constexpr int ROWS = MAXT * 16;
  constexpr int CT = 6;                     // 16-key tiles per online-softmax step (register budget)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Kimg = smem;
  unsigned char* Vimg = smem + ROWS * C::RS;
  const int item = xcd_remap();
  const int bh = item / P.nchunk, chunk = item - bh * P.nchunk;
  const int b = bh / P.nh, h = bh - b * P.nh;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4;
  const unsigned ldq_b = (unsigned)P.ldq * 2, ldk_b = (unsigned)P.ldk * 2, ldv_b = (unsigned)P.ldv * 2;
  const rsrc_t rq = make_rsrc(P.q + b * P.bq + h * HD, (unsigned)((P.sq - 1) * P.ldq + HD) * 2);
  const rsrc_t rk = make_rsrc(P.k + b * P.bk + h * HD, (unsigned)((P.skv - 1) * P.ldk + HD) * 2);
  const rsrc_t rv = make_rsrc(P.v + b * P.bv + h * HD, (unsigned)((P.skv - 1) * P.ldv + HD) * 2);
  const float c = P.alpha * LOG2E;
  const int q_begin = chunk * P.chunk_rows;
  const int q_end = min(P.sq, q_begin + P.chunk_rows);
  const int ntq = (q_end - q_begin + 15) >> 4;

  FragHD<HD> qf;
  FwdAcc<HD> A;

  auto step = [&](int kv0) {   // this wave's q-tile against the whole LDS tile
    const int kvalid = P.skv - kv0 - 4 * g;   // this lane's keys kv0 + 16 t + 4 g + r are real iff 16 t + r < kvalid
#pragma unroll
    for (int t0 = 0; t0 < MAXT; t0 += CT) {
      if (t0) __builtin_amdgcn_sched_barrier(0);   // keep the next step's LDS reads from being hoisted over this one (registers)
      if (MAXT - t0 >= CT) {
        if (FULL) fwd_substep<HD, CT, 0>(Kimg, Vimg, t0, qf, c, kvalid, A, lane);
        else if (MAXT - t0 == CT) fwd_substep<HD, CT, CT - 2>(Kimg, Vimg, t0, qf, c, kvalid, A, lane);
        else fwd_substep<HD, CT, CT>(Kimg, Vimg, t0, qf, c, kvalid, A, lane);   // (MAXT and CT even: the end cannot be in here)
      } else {
        constexpr int TN = MAXT % CT ? MAXT % CT : 2;
        fwd_substep<HD, TN, FULL ? 0 : TN - 2>(Kimg, Vimg, t0, qf, c, kvalid, A, lane);
      }
    }
  };
  auto finish = [&](int qt) {   // normalise and store this wave's q-tile
    const int q = q_begin + qt * 16 + (lane & 15);
    if (q < q_end) {
      const float inv = 1.0f / A.l[0];
      bf16_t* o = P.out + b * P.bo + (long)q * P.ldo + h * HD + 4 * g;
#pragma unroll
      for (int d = 0; d < C::ND; ++d) store4_bf16(o + 16 * d, A.o[d], inv);
      if (g == 0) P.lse[(long)bh * P.sqp + q] = A.m * P.alpha + __logf(A.l[0]);
    }
  };

  if (P.ntile == 1) {
    const int nfull = P.shared ? ntq - 1 : ntq;                               // tiles handled whole, wave-strided
    FragHD<HD> qn = frag_global<HD>(rq, ldq_b, q_begin + wave * 16, lane);    // in flight under the K/V fetch
    load_tile<HD, NT, ROWS>(Kimg, rk, ldk_b, 0, ROWS);
    load_tile<HD, NT, ROWS>(Vimg, rv, ldv_b, 0, ROWS);
    __syncthreads();
    for (int qt = wave; qt < nfull; qt += NW) {
      qf = qn;
      const int nxt = qt + NW < nfull ? qt + NW : ntq - 1;                    // next whole tile, or the shared one
      qn = frag_global<HD>(rq, ldq_b, q_begin + nxt * 16, lane);
      fwd_reset<HD>(A);
      step(0);
      finish(qt);
    }
    if (P.shared) {   // wave w: key pairs w, w + NW, ... of the last q-tile, then merge through LDS
      const int qs = ntq - 1;
      qf = nfull > wave ? qn : frag_global<HD>(rq, ldq_b, q_begin + qs * 16, lane);
      fwd_reset<HD>(A);
      const int kvalid = P.skv - 4 * g;
      for (int s = wave; s < MAXT / 2; s += NW) fwd_substep<HD, 2, 0>(Kimg, Vimg, 2 * s, qf, c, kvalid, A, lane);
      __syncthreads();                                   // every wave is done with the K / V images
      constexpr int PW = HD + 2;                         // floats per (wave, row): O | m | l
      float* part = (float*)smem;
      float* mine = part + ((wave * 16 + (lane & 15)) * PW);
#pragma unroll
      for (int d = 0; d < C::ND; ++d) *(f32x4*)(mine + 16 * d + 4 * g) = A.o[d];
      if (g == 0) { mine[HD] = A.m; mine[HD + 1] = A.l[0]; }
      __syncthreads();
      if (wave == 0) {
        float mg = NEG_BIG;
#pragma unroll
        for (int w = 0; w < NW; ++w) mg = fmaxf(mg, part[(w * 16 + (lane & 15)) * PW + HD]);
        fwd_reset<HD>(A);
        A.m = mg;
        float lsum = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          const float* pw = part + (w * 16 + (lane & 15)) * PW;
          const float f = __builtin_amdgcn_exp2f((pw[HD] - mg) * c);
          lsum = fmaf(pw[HD + 1], f, lsum);
#pragma unroll
          for (int d = 0; d < C::ND; ++d) A.o[d] += *(const f32x4*)(pw + 16 * d + 4 * g) * f;
        }
        A.l = f32x4{lsum, lsum, lsum, lsum};
        finish(qs);
      }
    }
  } else {   // one q-tile per wave (chunk_rows == NW * 16), state carried over the streamed K/V tiles
    const bool active = wave < ntq;
    qf = frag_global<HD>(rq, ldq_b, q_begin + wave * 16, lane);
    fwd_reset<HD>(A);
    for (int j = 0; j < P.ntile; ++j) {
      const int kv0 = j * ROWS;
      if (j) __syncthreads();
      load_tile<HD, NT, ROWS>(Kimg, rk, ldk_b, kv0, ROWS);
      load_tile<HD, NT, ROWS>(Vimg, rv, ldv_b, kv0, ROWS);
      __syncthreads();
      if (active) step(kv0);
    }
    if (active) finish(wave);
  }
}

// =================================================================================================================
// backward, part 1: dQ (+ the softmax-backward row constant dsum[q] = sum_d dO[q,d] O[q,d], written for part 2).
// Each wave owns 16-query tiles and walks the keys in pairs of 16-key tiles.
// =================================================================================================================
// one 32-key pair: dS for the wave's q-tile, acc += dS K.  MASK: keys past the sequence end (only the last pair of the last
// streamed tile can hold them) get the -1e30 score init, so P = 0 there.
template <int HD, bool MASK>
__device__ __forceinline__ void dq_pair(const unsigned char* Kimg, const unsigned char* Vimg, int s, const FragHD<HD>& qf,
                                        const FragHD<HD>& df, float c, float lq2, float dsq, int kvalid,
                                        f32x4 (&acc)[HeadCfg<HD>::ND], int lane) {
  using C = HeadCfg<HD>;
  f32x4 dsv[2];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int k0 = 32 * s + 16 * half;
    f32x4 ci = {0.f, 0.f, 0.f, 0.f};
    if (MASK) {
#pragma unroll
      for (int r = 0; r < 4; ++r) ci[r] = (k0 + r < kvalid) ? 0.f : NEG_BIG;
    }
    const f32x4 sa = mma_hd<HD>(frag_lds<HD>(Kimg, k0, lane), qf, ci);
    const f32x4 dp = mma_hd<HD>(frag_lds<HD>(Vimg, k0, lane), df, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
    for (int r = 0; r < 4; ++r) dsv[half][r] = __builtin_amdgcn_exp2f(fmaf(sa[r], c, -lq2)) * (dp[r] - dsq);
  }
  const bf16x8 dsb = pack8(dsv[0], dsv[1]);
#pragma unroll
  for (int d = 0; d < C::ND; ++d)
    acc[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_seq<HD>(Kimg, 32 * s, 16 * d, lane), dsb, acc[d], 0, 0, 0);
}

template <int HD, int NW>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dq_kernel(const AttnParams P) {
  using C = HeadCfg<HD>;
  constexpr int NT = NW * 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Kimg = smem;
  unsigned char* Vimg = smem + P.tile_rows * C::RS;
  const int item = xcd_remap();
  const int bh = item / P.nchunk, chunk = item - bh * P.nchunk;
  const int b = bh / P.nh, h = bh - b * P.nh;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4;
  const unsigned ldk_b = (unsigned)P.ldk * 2, ldv_b = (unsigned)P.ldv * 2;
  const rsrc_t rq = make_rsrc(P.q + b * P.bq + h * HD, (unsigned)((P.sq - 1) * P.ldq + HD) * 2);
  const rsrc_t ro = make_rsrc(P.o + b * P.bo + h * HD, (unsigned)((P.sq - 1) * P.ldo + HD) * 2);
  const rsrc_t rd = make_rsrc(P.d_o + b * P.bdo + h * HD, (unsigned)((P.sq - 1) * P.lddo + HD) * 2);
  const rsrc_t rk = make_rsrc(P.k + b * P.bk + h * HD, (unsigned)((P.skv - 1) * P.ldk + HD) * 2);
  const rsrc_t rv = make_rsrc(P.v + b * P.bv + h * HD, (unsigned)((P.skv - 1) * P.ldv + HD) * 2);
  const float c = P.alpha * LOG2E;
  const int q_begin = chunk * P.chunk_rows;
  const int q_end = min(P.sq, q_begin + P.chunk_rows);
  const int ntq = (q_end - q_begin + 15) >> 4;

  FragHD<HD> qf, df;
  float lq2 = 0.f, dsq = 0.f;   // lse * log2(e), dsum of this lane's query
  f32x4 acc[C::ND];
  struct QIn { FragHD<HD> q, d, o; float l; };   // what a q-tile needs from global memory (prefetched one tile ahead)

  auto fetch_tile = [&](int qt) {
    QIn t;
    const int r0 = q_begin + qt * 16;
    t.q = frag_global<HD>(rq, (unsigned)P.ldq * 2, r0, lane);
    t.d = frag_global<HD>(rd, (unsigned)P.lddo * 2, r0, lane);
    t.o = frag_global<HD>(ro, (unsigned)P.ldo * 2, r0, lane);
    const int q = r0 + (lane & 15);
    t.l = q < q_end ? P.lse[(long)bh * P.sqp + q] : 0.f;
    return t;
  };
  auto begin_tile = [&](int qt, const QIn& t, bool write_dsum) {
    qf = t.q; df = t.d;
    const int q = q_begin + qt * 16 + (lane & 15);
    dsq = quad_g_sum(dot_hd<HD>(t.d, t.o));
    lq2 = t.l * LOG2E;
    if (write_dsum && q < q_end && g == 0) P.dsum[(long)bh * P.sqp + q] = dsq;
#pragma unroll
    for (int d = 0; d < C::ND; ++d) acc[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto step = [&](int kv0, int npair) {   // all pairs of the LDS tile; only the last one can hold the sequence end
    const int kvalid = P.skv - kv0 - 4 * g;
    for (int s = 0; s < npair - 1; ++s) dq_pair<HD, false>(Kimg, Vimg, s, qf, df, c, lq2, dsq, kvalid, acc, lane);
    dq_pair<HD, true>(Kimg, Vimg, npair - 1, qf, df, c, lq2, dsq, kvalid, acc, lane);
  };
  auto finish_tile = [&](int qt) {
    const int q = q_begin + qt * 16 + (lane & 15);
    if (q < q_end) {
      bf16_t* o = P.dq + b * P.bdq + (long)q * P.lddq + h * HD + 4 * g;
#pragma unroll
      for (int d = 0; d < C::ND; ++d) store4_bf16(o + 16 * d, acc[d], P.alpha);
    }
  };

  if (P.ntile == 1) {
    const int rows = P.tile_rows, npair = rows >> 5;
    const int nfull = P.shared ? ntq - 1 : ntq;
    QIn tn = fetch_tile(wave < nfull ? wave : ntq - 1);   // in flight under the K/V fetch
    load_tile<HD, NT, 288>(Kimg, rk, ldk_b, 0, rows);
    load_tile<HD, NT, 288>(Vimg, rv, ldv_b, 0, rows);
    __syncthreads();
    for (int qt = wave; qt < nfull; qt += NW) {
      begin_tile(qt, tn, true);
      tn = fetch_tile(qt + NW < nfull ? qt + NW : ntq - 1);
      step(0, npair);
      finish_tile(qt);
    }
    if (P.shared) {   // wave w: key pairs w, w + NW, ... of the last q-tile; partial dQ summed through LDS
      const int qs = ntq - 1;
      begin_tile(qs, tn, wave == 0);
      const int kvalid = P.skv - 4 * g;
      for (int s = wave; s < npair; s += NW) dq_pair<HD, true>(Kimg, Vimg, s, qf, df, c, lq2, dsq, kvalid, acc, lane);
      __syncthreads();                                   // every wave is done with the K / V images
      float* part = (float*)smem;
      float* mine = part + (wave * 16 + (lane & 15)) * HD;
#pragma unroll
      for (int d = 0; d < C::ND; ++d) *(f32x4*)(mine + 16 * d + 4 * g) = acc[d];
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int w = 1; w < NW; ++w)
#pragma unroll
          for (int d = 0; d < C::ND; ++d) acc[d] += *(const f32x4*)(part + (w * 16 + (lane & 15)) * HD + 16 * d + 4 * g);
        finish_tile(qs);
      }
    }
  } else {
    const bool active = wave < ntq;
    begin_tile(wave, fetch_tile(wave), true);
    for (int j = 0; j < P.ntile; ++j) {
      const int kv0 = j * P.tile_rows;
      const int rows = min(P.tile_rows, (P.skv - kv0 + 31) & ~31);
      if (j) __syncthreads();
      load_tile<HD, NT, 288>(Kimg, rk, ldk_b, kv0, rows);
      load_tile<HD, NT, 288>(Vimg, rv, ldv_b, kv0, rows);
      __syncthreads();
      if (active) step(kv0, rows >> 5);
    }
    if (active) finish_tile(wave);
  }
}

// =================================================================================================================
// backward, part 2: dK, dV.  Each wave owns 16-key tiles and walks the queries in pairs of 16-query tiles; the queries'
// lse / dsum ride in LDS next to the Q and dO images (lse = +1e30 for rows past the sequence end, so their P is 0: no
// per-element masks in this kernel; key rows past the end only produce output rows that are never stored).
// =================================================================================================================
template <int HD>
__device__ __forceinline__ void dkv_pair(const unsigned char* Qimg, const unsigned char* Dimg, const float* Limg, const float* Simg,
                                         int s, const FragHD<HD>& kf, const FragHD<HD>& vf, float c,
                                         f32x4 (&dk)[HeadCfg<HD>::ND], f32x4 (&dv)[HeadCfg<HD>::ND], int lane) {
  using C = HeadCfg<HD>;
  const int g = lane >> 4;
  f32x4 pv[2], dsv[2];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int r0 = 32 * s + 16 * half;
    const f32x4 sa = mma_hd<HD>(frag_lds<HD>(Qimg, r0, lane), kf, f32x4{0.f, 0.f, 0.f, 0.f});
    const f32x4 dp = mma_hd<HD>(frag_lds<HD>(Dimg, r0, lane), vf, f32x4{0.f, 0.f, 0.f, 0.f});
    const f32x4 l4 = *(const f32x4*)(Limg + r0 + 4 * g);
    const f32x4 d4 = *(const f32x4*)(Simg + r0 + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      pv[half][r] = __builtin_amdgcn_exp2f(fmaf(sa[r], c, -l4[r]));
      dsv[half][r] = pv[half][r] * (dp[r] - d4[r]);
    }
  }
  const bf16x8 pb = pack8(pv[0], pv[1]), dsb = pack8(dsv[0], dsv[1]);
#pragma unroll
  for (int d = 0; d < C::ND; ++d) {
    dv[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_seq<HD>(Dimg, 32 * s, 16 * d, lane), pb, dv[d], 0, 0, 0);
    dk[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_seq<HD>(Qimg, 32 * s, 16 * d, lane), dsb, dk[d], 0, 0, 0);
  }
}

template <int HD, int NW>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dkv_kernel(const AttnParams P) {
  using C = HeadCfg<HD>;
  constexpr int NT = NW * 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Qimg = smem;
  unsigned char* Dimg = smem + P.tile_rows * C::RS;  // dO image
  float* Limg = (float*)(smem + 2 * P.tile_rows * C::RS);   // lse * log2(e) of the tile's queries
  float* Simg = Limg + P.tile_rows;                        // dsum
  const int item = xcd_remap();
  const int bh = item / P.nchunk, chunk = item - bh * P.nchunk;
  const int b = bh / P.nh, h = bh - b * P.nh;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4;
  const unsigned ldq_b = (unsigned)P.ldq * 2, lddo_b = (unsigned)P.lddo * 2;
  const rsrc_t rq = make_rsrc(P.q + b * P.bq + h * HD, (unsigned)((P.sq - 1) * P.ldq + HD) * 2);
  const rsrc_t rd = make_rsrc(P.d_o + b * P.bdo + h * HD, (unsigned)((P.sq - 1) * P.lddo + HD) * 2);
  const rsrc_t rk = make_rsrc(P.k + b * P.bk + h * HD, (unsigned)((P.skv - 1) * P.ldk + HD) * 2);
  const rsrc_t rv = make_rsrc(P.v + b * P.bv + h * HD, (unsigned)((P.skv - 1) * P.ldv + HD) * 2);
  const float c = P.alpha * LOG2E;
  const int k_begin = chunk * P.chunk_rows;
  const int k_end = min(P.skv, k_begin + P.chunk_rows);
  const int ntk = (k_end - k_begin + 15) >> 4;
  const float* lrow = P.lse + (long)bh * P.sqp;
  const float* drow = P.dsum + (long)bh * P.sqp;

  FragHD<HD> kf, vf;
  f32x4 dv[C::ND], dk[C::ND];

  auto load_q_tile = [&](int q0, int rows) {
    load_tile<HD, NT, 288>(Qimg, rq, ldq_b, q0, rows);
    load_tile<HD, NT, 288>(Dimg, rd, lddo_b, q0, rows);
    for (int i = threadIdx.x; i < rows; i += NT) {
      const bool ok = q0 + i < P.sq;
      Limg[i] = ok ? lrow[q0 + i] * LOG2E : 1e30f;
      Simg[i] = ok ? drow[q0 + i] : 0.f;
    }
  };
  struct KIn { FragHD<HD> k, v; };
  auto fetch_tile = [&](int kt) {
    KIn t;
    t.k = frag_global<HD>(rk, (unsigned)P.ldk * 2, k_begin + kt * 16, lane);
    t.v = frag_global<HD>(rv, (unsigned)P.ldv * 2, k_begin + kt * 16, lane);
    return t;
  };
  auto begin_tile = [&](const KIn& t) {
    kf = t.k; vf = t.v;
#pragma unroll
    for (int d = 0; d < C::ND; ++d) { dv[d] = f32x4{0.f, 0.f, 0.f, 0.f}; dk[d] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  };
  auto finish_tile = [&](int kt) {
    const int key = k_begin + kt * 16 + (lane & 15);
    if (key < k_end) {
      bf16_t* ok_ = P.dk + b * P.bdk + (long)key * P.lddk + h * HD + 4 * g;
      bf16_t* ov_ = P.dv + b * P.bdv + (long)key * P.lddv + h * HD + 4 * g;
#pragma unroll
      for (int d = 0; d < C::ND; ++d) {
        store4_bf16(ok_ + 16 * d, dk[d], P.alpha);
        store4_bf16(ov_ + 16 * d, dv[d], 1.0f);
      }
    }
  };

  if (P.ntile == 1) {
    const int rows = P.tile_rows, npair = rows >> 5;
    const int nfull = P.shared ? ntk - 1 : ntk;
    KIn tn = fetch_tile(wave < nfull ? wave : ntk - 1);   // in flight under the Q / dO fetch
    load_q_tile(0, rows);
    __syncthreads();
    for (int kt = wave; kt < nfull; kt += NW) {
      begin_tile(tn);
      tn = fetch_tile(kt + NW < nfull ? kt + NW : ntk - 1);
      for (int s = 0; s < npair; ++s) dkv_pair<HD>(Qimg, Dimg, Limg, Simg, s, kf, vf, c, dk, dv, lane);
      finish_tile(kt);
    }
    if (P.shared) {   // wave w: query pairs w, w + NW, ... against the last key tile; partial dK, dV summed through LDS
      const int ks = ntk - 1;
      begin_tile(tn);
      for (int s = wave; s < npair; s += NW) dkv_pair<HD>(Qimg, Dimg, Limg, Simg, s, kf, vf, c, dk, dv, lane);
      __syncthreads();                                   // every wave is done with the Q / dO images
      float* part = (float*)smem;
      float* mine = part + (wave * 16 + (lane & 15)) * (2 * HD);
#pragma unroll
      for (int d = 0; d < C::ND; ++d) {
        *(f32x4*)(mine + 16 * d + 4 * g) = dk[d];
        *(f32x4*)(mine + HD + 16 * d + 4 * g) = dv[d];
      }
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int w = 1; w < NW; ++w) {
          const float* pw = part + (w * 16 + (lane & 15)) * (2 * HD);
#pragma unroll
          for (int d = 0; d < C::ND; ++d) {
            dk[d] += *(const f32x4*)(pw + 16 * d + 4 * g);
            dv[d] += *(const f32x4*)(pw + HD + 16 * d + 4 * g);
          }
        }
        finish_tile(ks);
      }
    }
  } else {
    const bool active = wave < ntk;
    begin_tile(fetch_tile(wave));
    for (int j = 0; j < P.ntile; ++j) {
      const int q0 = j * P.tile_rows;
      const int rows = min(P.tile_rows, (P.sq - q0 + 31) & ~31);
      if (j) __syncthreads();
      load_q_tile(q0, rows);
      __syncthreads();
      if (active) for (int s = 0; s < (rows >> 5); ++s) dkv_pair<HD>(Qimg, Dimg, Limg, Simg, s, kf, vf, c, dk, dv, lane);
    }
    if (active) finish_tile(wave);
  }
}

// =================================================================================================================
// host side: work decomposition
// =================================================================================================================
static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
static inline size_t max_sz(size_t a, size_t b) { return a > b ? a : b; }

// waves per workgroup for `tiles` 16-row tiles handled by one workgroup.  Only multiples of 4: a 6-wave workgroup (the balanced
// count for 17 tiles) measured SLOWER than 4 or 8 (54 vs 48 / 40 us forward at 64 x 16 heads x 257): its waves land 2,2,1,1 on
// the four SIMDs and a second workgroup is then not admitted next to it, leaving one workgroup per CU.
static int pick_waves(int tiles) { return tiles > 6 ? 8 : 4; }
// split the leftover tile over all waves when exactly one is left over and every wave also has whole tiles to do
static int use_shared(int tiles, int nw) { return tiles > nw && tiles % nw == 1; }

struct Plan { int nw, nchunk, chunk_rows, tile_rows, ntile, shared; size_t lds; };

// backward kernels: stationary = rows the waves own (sq for dQ, skv for dK,dV); streamed = rows that pass through LDS
// part_floats = per-(wave,row) floats of the shared-tile exchange (HD for dQ, 2 HD for dK,dV)
template <int HD>
static Plan make_plan(int stationary, int streamed, bool extra_f32, int part_floats) {
  using C = HeadCfg<HD>;
  Plan p;
  p.shared = 0;
  if (streamed <= 288) { p.tile_rows = round_up(streamed, 32); p.ntile = 1; }
  else { p.tile_rows = 256; p.ntile = (streamed + 255) / 256; }
  const int tiles = (stationary + 15) / 16;
  if (p.ntile == 1 && tiles <= 24) {          // whole head in one workgroup
    p.nw = pick_waves(tiles); p.nchunk = 1; p.chunk_rows = tiles * 16; p.shared = use_shared(tiles, p.nw);
  } else if (p.ntile == 1) {                   // long stationary side, short streamed side (cross-attention): 2 tiles per wave
    p.nw = 8; p.chunk_rows = 256; p.nchunk = (stationary + 255) / 256;
  } else {                                      // state carried across streamed tiles: one tile per wave
    p.nw = 8; p.chunk_rows = 128; p.nchunk = (stationary + 127) / 128;
  }
  p.lds = 2 * (size_t)p.tile_rows * C::RS + (extra_f32 ? 2 * (size_t)p.tile_rows * 4 : 0);
  if (p.shared) p.lds = max_sz(p.lds, (size_t)p.nw * 16 * part_floats * 4);
  return p;
}

template <int HD, int MAXT, int NW, bool FULL>
static int fwd_launch_k(AttnParams P, int chunk_rows, int nchunk, int ntile, int shared, int batch, hipStream_t st) {
  using C = HeadCfg<HD>;
  P.nchunk = nchunk; P.chunk_rows = chunk_rows; P.tile_rows = MAXT * 16; P.ntile = ntile; P.shared = shared;
  size_t lds = 2 * (size_t)MAXT * 16 * C::RS;
  if (shared) lds = max_sz(lds, (size_t)NW * 16 * (HD + 2) * 4);
  auto k = attn_fwd_kernel<HD, MAXT, NW, FULL>;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k, dim3(batch * P.nh * nchunk), dim3(NW * 64), lds, st, P);
  return (int)hipGetLastError();
}
// forward: the kernel variant is (16-key tiles per K/V tile, waves, mask mode)
template <int HD>
static int attn_fwd_launch(const AttnParams& P, int batch, hipStream_t st) {
  const int qt = (P.sq + 15) / 16;                 // 16-query tiles per head
  if (P.skv <= 288) {
    const int nt = 2 * ((P.skv + 31) / 32);        // exactly the tiles the keys need: only the last two can hold the end
    const bool whole = qt <= 24;                   // whole head in one workgroup, else 256-query chunks (2 tiles per wave)
    const int rows = whole ? qt * 16 : 256, nchunk = whole ? 1 : (P.sq + 255) / 256;
    const bool big = (whole ? qt : 16) > 6;        // 8 waves when there are q-tiles for them
    const int sh = whole ? use_shared(qt, big ? 8 : 4) : 0;
    switch (nt) {
      case 2: return big ? fwd_launch_k<HD, 2, 8, false>(P, rows, nchunk, 1, sh, batch, st) : fwd_launch_k<HD, 2, 4, false>(P, rows, nchunk, 1, sh, batch, st);
      case 4: return big ? fwd_launch_k<HD, 4, 8, false>(P, rows, nchunk, 1, sh, batch, st) : fwd_launch_k<HD, 4, 4, false>(P, rows, nchunk, 1, sh, batch, st);
      case 6: return big ? fwd_launch_k<HD, 6, 8, false>(P, rows, nchunk, 1, sh, batch, st) : fwd_launch_k<HD, 6, 4, false>(P, rows, nchunk, 1, sh, batch, st);
      case 16: return fwd_launch_k<HD, 16, 8, false>(P, rows, nchunk, 1, big ? sh : 0, batch, st);
      case 18: return fwd_launch_k<HD, 18, 8, false>(P, rows, nchunk, 1, big ? sh : 0, batch, st);
      default: return fwd_launch_k<HD, 16, 8, true>(P, rows, nchunk, 1, big ? sh : 0, batch, st);   // 8..14 tiles: the 16-tile kernel, every tile masked
    }
  }
  // streamed K/V (256 keys per tile), one q-tile per wave, 128-query chunks
  const int ntile = (P.skv + 255) / 256, nchunk = (P.sq + 127) / 128;
  const int tail = P.skv - (ntile - 1) * 256;      // keys in the last tile
  if (tail > 224) return fwd_launch_k<HD, 16, 8, false>(P, 128, nchunk, ntile, 0, batch, st);
  return fwd_launch_k<HD, 16, 8, true>(P, 128, nchunk, ntile, 0, batch, st);
}

template <int HD, int NW>
static int bwd_launch_dq(const AttnParams& P, const Plan& pl, int items, hipStream_t st) {
  auto k = attn_bwd_dq_kernel<HD, NW>;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds);
  hipLaunchKernelGGL(k, dim3(items), dim3(NW * 64), pl.lds, st, P);
  return (int)hipGetLastError();
}
template <int HD, int NW>
static int bwd_launch_dkv(const AttnParams& P, const Plan& pl, int items, hipStream_t st) {
  auto k = attn_bwd_dkv_kernel<HD, NW>;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds);
  hipLaunchKernelGGL(k, dim3(items), dim3(NW * 64), pl.lds, st, P);
  return (int)hipGetLastError();
}
template <int HD>
static int attn_bwd_launch(AttnParams P, int batch, hipStream_t st) {
  const Plan plq = make_plan<HD>(P.sq, P.skv, false, HD);       // dQ: queries stationary, keys streamed
  const Plan plk = make_plan<HD>(P.skv, P.sq, true, 2 * HD);    // dK,dV: keys stationary, queries streamed
  AttnParams Pq = P, Pk = P;
  Pq.nchunk = plq.nchunk; Pq.chunk_rows = plq.chunk_rows; Pq.tile_rows = plq.tile_rows; Pq.ntile = plq.ntile; Pq.shared = plq.shared;
  Pk.nchunk = plk.nchunk; Pk.chunk_rows = plk.chunk_rows; Pk.tile_rows = plk.tile_rows; Pk.ntile = plk.ntile; Pk.shared = plk.shared;
  const int iq = batch * P.nh * plq.nchunk, ik = batch * P.nh * plk.nchunk;
  // dQ first: it also produces dsum, which dK,dV consumes
  const int rc = plq.nw == 4 ? bwd_launch_dq<HD, 4>(Pq, plq, iq, st) : bwd_launch_dq<HD, 8>(Pq, plq, iq, st);
  if (rc) return rc;
  return plk.nw == 4 ? bwd_launch_dkv<HD, 4>(Pk, plk, ik, st) : bwd_launch_dkv<HD, 8>(Pk, plk, ik, st);
}

static int check_desc(const muse_attn_desc* d) {
  if (!d || d->seq_q <= 0 || d->seq_kv <= 0 || d->heads <= 0) return MUSE_ERR_BAD_ARG;
  if (d->head_dim != 16 && d->head_dim != 32 && d->head_dim != 48 && d->head_dim != 64) return MUSE_ERR_UNSUPPORTED;
  const int64_t lds[] = {d->ldq, d->ldk, d->ldv, d->ldo, d->bsq, d->bsk, d->bsv, d->bso};
  for (int64_t x : lds) if (x & 7) return MUSE_ERR_ALIGN;     // 16-byte rows
  const void* ps[] = {d->q, d->k, d->v, d->o};
  for (const void* p : ps) if (((uintptr_t)p) & 15) return MUSE_ERR_ALIGN;
  // 32-bit byte offsets inside one image
  if ((int64_t)d->seq_q * d->ldq * 2 >= (1LL << 31) || (int64_t)d->seq_kv * d->ldk * 2 >= (1LL << 31) ||
      (int64_t)d->seq_kv * d->ldv * 2 >= (1LL << 31) || (int64_t)d->seq_q * d->ldo * 2 >= (1LL << 31)) return MUSE_ERR_UNSUPPORTED;
  return 0;
}
static AttnParams base_params(const muse_attn_desc* d) {
  AttnParams P = {};
  P.q = (const bf16_t*)d->q; P.k = (const bf16_t*)d->k; P.v = (const bf16_t*)d->v;
  P.ldq = d->ldq; P.ldk = d->ldk; P.ldv = d->ldv; P.ldo = d->ldo;
  P.bq = d->bsq; P.bk = d->bsk; P.bv = d->bsv; P.bo = d->bso;
  P.nh = d->heads; P.sq = d->seq_q; P.skv = d->seq_kv; P.sqp = muse_attention_seq_pad(d->seq_q);
  P.alpha = d->alpha;
  return P;
}

extern "C" int muse_attention_seq_pad(int32_t seq) { return (seq + 31) / 32 * 32; }

extern "C" int muse_attention_fwd_ex(const muse_attn_desc* d, float* lse, void* stream) {
  const int rc = check_desc(d);
  if (rc) return rc;
  if (d->batch <= 0) return 0;
  AttnParams P = base_params(d);
  P.out = (bf16_t*)d->o; P.lse = lse;
  hipStream_t st = (hipStream_t)stream;
  {   // one-tile self-attention at head_dim 48 (S = 257): the 32 x 32-block kernels of attention2.hip
    const int r2 = attn2_fwd_try(P, d->head_dim, d->batch, st);
    if (r2 != 0) return r2 > 0 ? 0 : -r2;
  }
  switch (d->head_dim) {
    case 16: return attn_fwd_launch<16>(P, d->batch, st);
    case 32: return attn_fwd_launch<32>(P, d->batch, st);
    case 48: return attn_fwd_launch<48>(P, d->batch, st);
    case 64: return attn_fwd_launch<64>(P, d->batch, st);
  }
  return MUSE_ERR_UNSUPPORTED;
}

extern "C" int muse_attention_bwd_ex(const muse_attn_desc* d, const void* d_o, int64_t lddo, int64_t bsdo, const float* lse,
                                     float* dsum, void* dq, int64_t lddq, int64_t bsdq, void* dk, int64_t lddk, int64_t bsdk,
                                     void* dv, int64_t lddv, int64_t bsdv, void* stream) {
  const int rc = check_desc(d);
  if (rc) return rc;
  if ((lddo | bsdo | lddq | bsdq | lddk | bsdk | lddv | bsdv) & 7) return MUSE_ERR_ALIGN;
  if ((((uintptr_t)d_o) | ((uintptr_t)dq) | ((uintptr_t)dk) | ((uintptr_t)dv)) & 7) return MUSE_ERR_ALIGN;
  if ((int64_t)d->seq_q * lddo * 2 >= (1LL << 31)) return MUSE_ERR_UNSUPPORTED;
  if (d->batch <= 0) return 0;
  AttnParams P = base_params(d);
  P.o = (const bf16_t*)d->o; P.d_o = (const bf16_t*)d_o; P.lddo = lddo; P.bdo = bsdo;
  P.lse = (float*)lse; P.dsum = dsum;
  P.dq = (bf16_t*)dq; P.lddq = lddq; P.bdq = bsdq;
  P.dk = (bf16_t*)dk; P.lddk = lddk; P.bdk = bsdk;
  P.dv = (bf16_t*)dv; P.lddv = lddv; P.bdv = bsdv;
  hipStream_t st = (hipStream_t)stream;
  {   // (the fused backward wants 16-byte aligned gradient rows: its stores are 16 bytes wide)
    const bool al16 = (((uintptr_t)d_o | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15) == 0;
    const int r2 = al16 ? attn2_bwd_try(P, d->head_dim, d->batch, st) : 0;
    if (r2 != 0) return r2 > 0 ? 0 : -r2;
  }
  switch (d->head_dim) {
    case 16: return attn_bwd_launch<16>(P, d->batch, st);
    case 32: return attn_bwd_launch<32>(P, d->batch, st);
    case 48: return attn_bwd_launch<48>(P, d->batch, st);
    case 64: return attn_bwd_launch<64>(P, d->batch, st);
  }
  return MUSE_ERR_UNSUPPORTED;
}

// packed-qkv self-attention entry points (qkv [batch*seq, 3*heads*head_dim]: the fused QKV projection's output)
static muse_attn_desc packed_desc(const void* qkv, void* ctx, int32_t batch, int32_t seq, int32_t heads, int32_t head_dim, float alpha) {
  const int64_t H = (int64_t)heads * head_dim;
  muse_attn_desc d = {};
  d.q = qkv; d.k = (const bf16_t*)qkv + H; d.v = (const bf16_t*)qkv + 2 * H; d.o = ctx;
  d.ldq = d.ldk = d.ldv = 3 * H; d.ldo = H;
  d.bsq = d.bsk = d.bsv = (int64_t)seq * 3 * H; d.bso = (int64_t)seq * H;
  d.batch = batch; d.heads = heads; d.head_dim = head_dim; d.seq_q = seq; d.seq_kv = seq; d.alpha = alpha;
  return d;
}
extern "C" int muse_attention_fwd(const void* qkv, void* ctx, float* lse, int32_t batch, int32_t seq, int32_t heads,
                                  int32_t head_dim, float alpha, void* stream) {
  const muse_attn_desc d = packed_desc(qkv, ctx, batch, seq, heads, head_dim, alpha);
  return muse_attention_fwd_ex(&d, lse, stream);
}
extern "C" int muse_attention_bwd(const void* qkv, const void* ctx, const void* dctx, const float* lse, float* dsum,
                                  void* dqkv, int32_t batch, int32_t seq, int32_t heads, int32_t head_dim, float alpha,
                                  void* stream) {
  const muse_attn_desc d = packed_desc(qkv, (void*)ctx, batch, seq, heads, head_dim, alpha);
  const int64_t H = (int64_t)heads * head_dim;
  bf16_t* g = (bf16_t*)dqkv;
  return muse_attention_bwd_ex(&d, dctx, H, (int64_t)seq * H, lse, dsum, g, 3 * H, (int64_t)seq * 3 * H, g + H, 3 * H,
                               (int64_t)seq * 3 * H, g + 2 * H, 3 * H, (int64_t)seq * 3 * H, stream);
}
