// Fused full-visibility attention for the MaskGit sequence lengths (S <= 288, head_dim in {16,32,48,64}), bf16 in/out,
// f32 softmax / accumulation.  Replaces Attention.attention (muse/modeling_transformer.py:221-241: baddbmm -> softmax ->
// matmul) and the xformers memory_efficient_attention seam (:206-210) without ever materialising the S x S matrix in HBM.
//
// One workgroup (4 waves) per (image, head).  The head's K and V (forward / dQ) or Q and dO (dK,dV) live in LDS as
// [S][hd] bf16 images with a row stride == 32 (mod 64) bytes: the same image is conflict-free for ds_read_b128
// (k = head-dim contiguous operand) and for ds_read_b64_tr_b16 (k = sequence operand, hardware transpose).
// Trick that keeps P in registers: scores are produced TRANSPOSED (S^T = K Q^T, MFMA operands swapped), so a lane holds,
// for its one query (lane & 15), the keys {16t + 4g + r}; two consecutive 16-key tiles therefore give exactly the 8
// k-slots one lane must supply as the B operand of the next MFMA (O^T = V^T P^T) under the k-permutation
// slot(g, j) <-> key 32s + 16*(j>>2) + 4g + (j&3), which the V^T operand reproduces through its tr-read row addresses.
#include "common.h"
#include "../../include/muse_hip.h"

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

template <int HD> struct HeadCfg {
  static constexpr int HDP = (HD + 31) / 32 * 32;   // head dim padded to the MFMA K = 32
  static constexpr int KS = HDP / 32;               // k-steps of the (head-dim contracted) MFMAs
  static constexpr int ND = HD / 16;                // 16-wide output tiles over the head dim
  // LDS row stride in bytes, == 32 (mod 64): 96 B for hd 48 / 32, 160 B for hd 64, 32 B for hd 16.  Only HD columns are
  // stored; the zero padding of the last 32-wide k-step (hd 48, 16) is produced in frag_hd instead of in LDS.
  static constexpr int RS = ((HD * 2) % 64 == 32) ? HD * 2 : HD * 2 + 32;
};

// cooperative load of one head's [S][HD] slice (row stride ld elements) into an LDS image of SKP rows, zero padded
template <int HD>
__device__ __forceinline__ void load_head(unsigned char* img, const bf16_t* src, long ld, int S, int SKP) {
  using C = HeadCfg<HD>;
  constexpr int CPR = HD / 8;
  for (int c = threadIdx.x; c < SKP * CPR; c += blockDim.x) {
    const int row = c / CPR, col = (c % CPR) * 8;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (row < S) v = *(const u32x4*)(src + (long)row * ld + col);
    *(u32x4*)(img + row * C::RS + col * 2) = v;
  }
}

// B/A operand with k = head dim: rows [rowbase, +16) of an LDS image
template <int HD>
__device__ __forceinline__ bf16x8 frag_hd(const unsigned char* img, int rowbase, int ks, int lane) {
  const int col = ks * 32 + (lane >> 4) * 8;
  if constexpr (HD % 32 != 0) {
    union { u32x4 u; bf16x8 v; } t;
    t.u = u32x4{0u, 0u, 0u, 0u};
    if (col < HD) t.v = *(const bf16x8*)(img + (rowbase + (lane & 15)) * HeadCfg<HD>::RS + col * 2);
    return t.v;
  } else {
    return *(const bf16x8*)(img + (rowbase + (lane & 15)) * HeadCfg<HD>::RS + col * 2);
  }
}
// the same operand straight from global memory (rows owned by this wave), zero outside [0,S) x [0,HD)
template <int HD>
__device__ __forceinline__ bf16x8 frag_hd_global(const bf16_t* src, long ld, int rowbase, int ks, int S, int lane) {
  const int row = rowbase + (lane & 15), col = ks * 32 + (lane >> 4) * 8;
  union { u32x4 u; bf16x8 v; } t;
  t.u = u32x4{0u, 0u, 0u, 0u};
  if (row < S && col < HD) t.u = *(const u32x4*)(src + (long)row * ld + col);
  return t.v;
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
__device__ __forceinline__ bf16x8 pack8(const float (&lo)[4], const float (&hi)[4]) {
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

// =================================================================================================================
// forward: ctx[b, q, h, :] = softmax(alpha * Q K^T) V ; lse[b*nh + h, q] = log sum exp of the scaled scores
// =================================================================================================================
template <int HD, int MAXT>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ ctx,
                                                       float* __restrict__ lse, int S, int SKP, int nh, float alpha) {
  using C = HeadCfg<HD>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Kimg = smem;
  unsigned char* Vimg = smem + SKP * C::RS;
  const int bh = blockIdx.x, b = bh / nh, h = bh - b * nh;
  const int H = nh * HD;
  const long ld = 3L * H;
  const bf16_t* base = qkv + (long)b * S * ld + h * HD;
  load_head<HD>(Kimg, base + H, ld, S, SKP);
  load_head<HD>(Vimg, base + 2 * H, ld, S, SKP);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
  const int nt = SKP >> 4, nq = (S + 15) >> 4;
  for (int qt = wave; qt < nq; qt += 4) {
    bf16x8 qf[C::KS];
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) qf[ks] = frag_hd_global<HD>(base, ld, qt * 16, ks, S, lane);
    f32x4 sacc[MAXT];
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      sacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (t < nt) {
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks)
          sacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_hd<HD>(Kimg, t * 16, ks, lane), qf[ks], sacc[t], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = t * 16 + 4 * g + r;
          sacc[t][r] = key < S ? sacc[t][r] * alpha : -INFINITY;
          m = fmaxf(m, sacc[t][r]);
        }
      }
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      if (t < nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { sacc[t][r] = __expf(sacc[t][r] - m); sum += sacc[t][r]; }
      }
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    f32x4 oacc[C::ND];
#pragma unroll
    for (int d = 0; d < C::ND; ++d) oacc[d] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < MAXT / 2; ++s) {
      if (2 * s < nt) {
        float lo[4] = {sacc[2 * s][0], sacc[2 * s][1], sacc[2 * s][2], sacc[2 * s][3]};
        float hi[4] = {sacc[2 * s + 1][0], sacc[2 * s + 1][1], sacc[2 * s + 1][2], sacc[2 * s + 1][3]};
        const bf16x8 pb = pack8(lo, hi);
#pragma unroll
        for (int d = 0; d < C::ND; ++d)
          oacc[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_seq<HD>(Vimg, 32 * s, 16 * d, lane), pb, oacc[d], 0, 0, 0);
      }
    }
    const int q = qt * 16 + (lane & 15);
    if (q < S) {
      const float inv = 1.0f / sum;
      bf16_t* o = ctx + ((long)b * S + q) * H + h * HD + 4 * g;
#pragma unroll
      for (int d = 0; d < C::ND; ++d) store4_bf16(o + 16 * d, oacc[d], inv);
      if (g == 0) lse[(long)bh * SKP + q] = m + __logf(sum);
    }
  }
}

// dsum[bh, q] = sum_d dctx[b,q,h,d] * ctx[b,q,h,d]   (the softmax-backward row constant), zero for q in [S, SKP)
template <int HD>
__global__ void attn_bwd_prep_kernel(const bf16_t* __restrict__ ctx, const bf16_t* __restrict__ dctx, float* __restrict__ dsum,
                                     int S, int SKP, int nh, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // over (bh, q in SKP)
  if (i >= total) return;
  const int q = (int)(i % SKP);
  const long bh = i / SKP;
  const int b = (int)(bh / nh), h = (int)(bh - (long)b * nh);
  float s = 0.f;
  if (q < S) {
    const long off = ((long)b * S + q) * (nh * HD) + h * HD;
#pragma unroll
    for (int c = 0; c < HD; c += 8) {
      const u32x4 a = *(const u32x4*)(ctx + off + c), d = *(const u32x4*)(dctx + off + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s = fmaf(__uint_as_float(a[j] << 16), __uint_as_float(d[j] << 16), s);
        s = fmaf(__uint_as_float(a[j] & 0xffff0000u), __uint_as_float(d[j] & 0xffff0000u), s);
      }
    }
  }
  dsum[i] = s;
}

// =================================================================================================================
// backward, part 1: dK, dV.  Each wave owns 16-key tiles and walks all queries in pairs of 16-query tiles.
// =================================================================================================================
template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dctx,
                                                           const float* __restrict__ lse, const float* __restrict__ dsum,
                                                           bf16_t* __restrict__ dqkv, int S, int SKP, int nh, float alpha) {
  using C = HeadCfg<HD>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Qimg = smem;
  unsigned char* Dimg = smem + SKP * C::RS;  // dO image
  const int bh = blockIdx.x, b = bh / nh, h = bh - b * nh;
  const int H = nh * HD;
  const long ld = 3L * H;
  const bf16_t* base = qkv + (long)b * S * ld + h * HD;
  load_head<HD>(Qimg, base, ld, S, SKP);
  load_head<HD>(Dimg, dctx + (long)b * S * H + h * HD, H, S, SKP);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
  const int nkt = (S + 15) >> 4, npair = SKP >> 5;
  const float* lrow = lse + (long)bh * SKP;
  const float* drow = dsum + (long)bh * SKP;
  for (int kt = wave; kt < nkt; kt += 4) {
    bf16x8 kf[C::KS], vf[C::KS];
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      kf[ks] = frag_hd_global<HD>(base + H, ld, kt * 16, ks, S, lane);
      vf[ks] = frag_hd_global<HD>(base + 2 * H, ld, kt * 16, ks, S, lane);
    }
    const bool key_ok = (kt * 16 + (lane & 15)) < S;
    f32x4 dv[C::ND], dk[C::ND];
#pragma unroll
    for (int d = 0; d < C::ND; ++d) { dv[d] = f32x4{0.f, 0.f, 0.f, 0.f}; dk[d] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    for (int s = 0; s < npair; ++s) {
      float pv[2][4], dsv[2][4];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int q0 = 32 * s + 16 * half;
        f32x4 sa = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
          sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_hd<HD>(Qimg, q0, ks, lane), kf[ks], sa, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_hd<HD>(Dimg, q0, ks, lane), vf[ks], dp, 0, 0, 0);
        }
        const f32x4 l4 = *(const f32x4*)(lrow + q0 + 4 * g);
        const f32x4 d4 = *(const f32x4*)(drow + q0 + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ok = key_ok && (q0 + 4 * g + r) < S;
          const float p = ok ? __expf(sa[r] * alpha - l4[r]) : 0.f;
          pv[half][r] = p;
          dsv[half][r] = p * (dp[r] - d4[r]);
        }
      }
      const bf16x8 pb = pack8(pv[0], pv[1]), dsb = pack8(dsv[0], dsv[1]);
#pragma unroll
      for (int d = 0; d < C::ND; ++d) {
        dv[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_seq<HD>(Dimg, 32 * s, 16 * d, lane), pb, dv[d], 0, 0, 0);
        dk[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_seq<HD>(Qimg, 32 * s, 16 * d, lane), dsb, dk[d], 0, 0, 0);
      }
    }
    const int key = kt * 16 + (lane & 15);
    if (key < S) {
      bf16_t* o = dqkv + ((long)b * S + key) * ld + h * HD + 4 * g;
#pragma unroll
      for (int d = 0; d < C::ND; ++d) {
        store4_bf16(o + H + 16 * d, dk[d], alpha);
        store4_bf16(o + 2 * H + 16 * d, dv[d], 1.0f);
      }
    }
  }
}

// =================================================================================================================
// backward, part 2: dQ.  Each wave owns 16-query tiles and walks all keys in pairs of 16-key tiles.
// =================================================================================================================
template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dctx,
                                                          const float* __restrict__ lse, const float* __restrict__ dsum,
                                                          bf16_t* __restrict__ dqkv, int S, int SKP, int nh, float alpha) {
  using C = HeadCfg<HD>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Kimg = smem;
  unsigned char* Vimg = smem + SKP * C::RS;
  const int bh = blockIdx.x, b = bh / nh, h = bh - b * nh;
  const int H = nh * HD;
  const long ld = 3L * H;
  const bf16_t* base = qkv + (long)b * S * ld + h * HD;
  load_head<HD>(Kimg, base + H, ld, S, SKP);
  load_head<HD>(Vimg, base + 2 * H, ld, S, SKP);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
  const int nq = (S + 15) >> 4, npair = SKP >> 5;
  const bf16_t* dbase = dctx + (long)b * S * H + h * HD;
  for (int qt = wave; qt < nq; qt += 4) {
    bf16x8 qf[C::KS], df[C::KS];
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      qf[ks] = frag_hd_global<HD>(base, ld, qt * 16, ks, S, lane);
      df[ks] = frag_hd_global<HD>(dbase, H, qt * 16, ks, S, lane);
    }
    const int q = qt * 16 + (lane & 15);
    const float lq = q < S ? lse[(long)bh * SKP + q] : 0.f;
    const float dq_ = q < S ? dsum[(long)bh * SKP + q] : 0.f;
    f32x4 acc[C::ND];
#pragma unroll
    for (int d = 0; d < C::ND; ++d) acc[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < npair; ++s) {
      float dsv[2][4];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int k0 = 32 * s + 16 * half;
        f32x4 sa = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
          sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_hd<HD>(Kimg, k0, ks, lane), qf[ks], sa, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_hd<HD>(Vimg, k0, ks, lane), df[ks], dp, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ok = (q < S) && (k0 + 4 * g + r) < S;
          const float p = ok ? __expf(sa[r] * alpha - lq) : 0.f;
          dsv[half][r] = p * (dp[r] - dq_);
        }
      }
      const bf16x8 dsb = pack8(dsv[0], dsv[1]);
#pragma unroll
      for (int d = 0; d < C::ND; ++d)
        acc[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_seq<HD>(Kimg, 32 * s, 16 * d, lane), dsb, acc[d], 0, 0, 0);
    }
    if (q < S) {
      bf16_t* o = dqkv + ((long)b * S + q) * ld + h * HD + 4 * g;
#pragma unroll
      for (int d = 0; d < C::ND; ++d) store4_bf16(o + 16 * d, acc[d], alpha);
    }
  }
}

// =================================================================================================================
template <int HD>
static int attn_fwd_launch(const void* qkv, void* ctx, float* lse, int B, int S, int nh, float alpha, hipStream_t st) {
  using C = HeadCfg<HD>;
  const int SKP = (S + 31) / 32 * 32;
  const size_t lds = 2 * (size_t)SKP * C::RS;
  if (SKP <= 64) {
    auto k = attn_fwd_kernel<HD, 4>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(B * nh), dim3(256), lds, st, (const bf16_t*)qkv, (bf16_t*)ctx, lse, S, SKP, nh, alpha);
  } else {
    auto k = attn_fwd_kernel<HD, 18>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(B * nh), dim3(256), lds, st, (const bf16_t*)qkv, (bf16_t*)ctx, lse, S, SKP, nh, alpha);
  }
  return (int)hipGetLastError();
}
template <int HD>
static int attn_bwd_launch(const void* qkv, const void* ctx, const void* dctx, const float* lse, float* dsum, void* dqkv,
                           int B, int S, int nh, float alpha, hipStream_t st) {
  using C = HeadCfg<HD>;
  const int SKP = (S + 31) / 32 * 32;
  const size_t lds = 2 * (size_t)SKP * C::RS;
  const long total = (long)B * nh * SKP;
  hipLaunchKernelGGL(attn_bwd_prep_kernel<HD>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const bf16_t*)ctx,
                     (const bf16_t*)dctx, dsum, S, SKP, nh, total);
  auto k1 = attn_bwd_dkv_kernel<HD>;
  auto k2 = attn_bwd_dq_kernel<HD>;
  (void)hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k1, dim3(B * nh), dim3(256), lds, st, (const bf16_t*)qkv, (const bf16_t*)dctx, lse, (const float*)dsum,
                     (bf16_t*)dqkv, S, SKP, nh, alpha);
  hipLaunchKernelGGL(k2, dim3(B * nh), dim3(256), lds, st, (const bf16_t*)qkv, (const bf16_t*)dctx, lse, (const float*)dsum,
                     (bf16_t*)dqkv, S, SKP, nh, alpha);
  return (int)hipGetLastError();
}

extern "C" int muse_attention_seq_pad(int32_t seq) { return (seq + 31) / 32 * 32; }

extern "C" int muse_attention_fwd(const void* qkv, void* ctx, float* lse, int32_t batch, int32_t seq, int32_t heads,
                                  int32_t head_dim, float alpha, void* stream) {
  if (seq > 288 || seq <= 0) return MUSE_ERR_UNSUPPORTED;
  if (batch <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  switch (head_dim) {
    case 16: return attn_fwd_launch<16>(qkv, ctx, lse, batch, seq, heads, alpha, st);
    case 32: return attn_fwd_launch<32>(qkv, ctx, lse, batch, seq, heads, alpha, st);
    case 48: return attn_fwd_launch<48>(qkv, ctx, lse, batch, seq, heads, alpha, st);
    case 64: return attn_fwd_launch<64>(qkv, ctx, lse, batch, seq, heads, alpha, st);
  }
  return MUSE_ERR_UNSUPPORTED;
}

extern "C" int muse_attention_bwd(const void* qkv, const void* ctx, const void* dctx, const float* lse, float* dsum,
                                  void* dqkv, int32_t batch, int32_t seq, int32_t heads, int32_t head_dim, float alpha,
                                  void* stream) {
  if (seq > 288 || seq <= 0) return MUSE_ERR_UNSUPPORTED;
  if (batch <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  switch (head_dim) {
    case 16: return attn_bwd_launch<16>(qkv, ctx, dctx, lse, dsum, dqkv, batch, seq, heads, alpha, st);
    case 32: return attn_bwd_launch<32>(qkv, ctx, dctx, lse, dsum, dqkv, batch, seq, heads, alpha, st);
    case 48: return attn_bwd_launch<48>(qkv, ctx, dctx, lse, dsum, dqkv, batch, seq, heads, alpha, st);
    case 64: return attn_bwd_launch<64>(qkv, ctx, dctx, lse, dsum, dqkv, batch, seq, heads, alpha, st);
  }
  return MUSE_ERR_UNSUPPORTED;
}
