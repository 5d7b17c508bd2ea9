// MaskGitVQGAN kernels (NHWC): implicit-GEMM convolution entry, GroupNorm+SiLU, avg-pool, layout changes,
// VQ helpers (row |z|^2, argmin, codebook gather), 2-D transpose and the tr16 probe.
#include "gemm_core.h"
#include "../../include/muse_hip.h"

// =================================================================================================================
// conv2d NHWC stride-1 SAME as implicit GEMM:  M = B*H*W pixels, N = Cout, K = KS*KS*Cin
// =================================================================================================================
extern "C" int muse_conv2d_nhwc(const void* in, const void* weight, const float* bias, const void* residual, void* out,
                                int32_t dtype, int32_t batch, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t KS,
                                int32_t upsample, void* stream) {
  const int esz = dtype == MUSE_BF16 ? 2 : 4, ch = 16 / esz;
  if (Cin % ch) return MUSE_ERR_ALIGN;
  if ((((uintptr_t)in) & 15) || (((uintptr_t)weight) & 15)) return MUSE_ERR_ALIGN;
  if (KS != 1 && KS != 3) return MUSE_ERR_UNSUPPORTED;
  if (upsample < 0 || upsample > 2 || (upsample == 2 && KS != 3)) return MUSE_ERR_BAD_ARG;
  if (upsample == 1 && ((H | W) & 1)) return MUSE_ERR_BAD_ARG;
  GemmParams p;
  p.A = in; p.B = weight; p.C = out;
  p.bias = bias; p.rowvec = nullptr; p.residual = residual;
  p.M = batch * H * W; p.N = Cout; p.K = KS * KS * Cin;
  p.lda = 0; p.ldb = p.K; p.ldc = Cout; p.ldr = Cout;
  p.zdiv = 1; p.sA0 = p.sA1 = p.sB0 = p.sB1 = p.sC0 = p.sC1 = 0;
  p.alpha = 1.0f; p.accumulate = 0; p.act = 0; p.split_k = 1; p.split_stride = 0;
  p.cH = H; p.cW = W; p.cCin = Cin; p.cKS = KS; p.cUps = upsample;
  p.cCinShift = -1;
  if ((Cin & (Cin - 1)) == 0) { int sh = 0; while ((1 << sh) < Cin) ++sh; p.cCinShift = sh; }
  hipStream_t s = (hipStream_t)stream;
  const bool big = use_bm256(p, 1);
  if (dtype == MUSE_BF16)
    return big ? launch_gemm<bf16_t, bf16_t, 0, 0, 256, ConvLoader<bf16_t, 256, 512>, PlainLoader<bf16_t, 0, 128, 512>>(p, 1, s)
               : launch_gemm<bf16_t, bf16_t, 0, 0, 128, ConvLoader<bf16_t, 128, 256>, PlainLoader<bf16_t, 0, 128, 256>>(p, 1, s);
  if (dtype == MUSE_F32)
    return big ? launch_gemm<float, float, 0, 0, 256, ConvLoader<float, 256, 512>, PlainLoader<float, 0, 128, 512>>(p, 1, s)
               : launch_gemm<float, float, 0, 0, 128, ConvLoader<float, 128, 256>, PlainLoader<float, 0, 128, 256>>(p, 1, s);
  return MUSE_ERR_BAD_ARG;
}

// =================================================================================================================
// GroupNorm(G) + SiLU over NHWC.  Pass 1: per (image, pixel-chunk) partial sum / sum-of-squares per group in f64.
// Pass 2: each block folds the partials of its image (fixed order), then y = silu((x - mean) * rstd * gamma + beta).
// =================================================================================================================
#define GN_PIX_PER_CHUNK 1024
extern "C" int muse_groupnorm_nchunk(int32_t HW) { return (HW + GN_PIX_PER_CHUNK - 1) / GN_PIX_PER_CHUNK; }

template <typename T, int VEC>
__device__ __forceinline__ void loadv(const T* p, float (&v)[VEC]);
template <> __device__ __forceinline__ void loadv<float, 4>(const float* p, float (&v)[4]) {
  const f32x4 t = *(const f32x4*)p; v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
}
template <> __device__ __forceinline__ void loadv<bf16_t, 8>(const bf16_t* p, float (&v)[8]) {
  const u32x4 t = *(const u32x4*)p;
#pragma unroll
  for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(t[j] << 16); v[2 * j + 1] = __uint_as_float(t[j] & 0xffff0000u); }
}
template <typename T, int VEC>
__device__ __forceinline__ void storev(T* p, const float (&v)[VEC]);
template <> __device__ __forceinline__ void storev<float, 4>(float* p, const float (&v)[4]) { *(f32x4*)p = f32x4{v[0], v[1], v[2], v[3]}; }
template <> __device__ __forceinline__ void storev<bf16_t, 8>(bf16_t* p, const float (&v)[8]) {
  u32x4 t;
#pragma unroll
  for (int j = 0; j < 4; ++j) t[j] = pack2_bf16(v[2 * j], v[2 * j + 1]);
  *(u32x4*)p = t;
}

template <typename T, int VEC>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x, double* __restrict__ partial, int HW, int C, int G) {
  __shared__ double gs[64], gq[64];
  const int chunk = blockIdx.x, b = blockIdx.y, nchunk = gridDim.x;
  if (threadIdx.x < G) { gs[threadIdx.x] = 0.0; gq[threadIdx.x] = 0.0; }
  __syncthreads();
  const int vpp = C / VEC;             // vectors per pixel
  const int cpg = C / G;
  const int p0 = chunk * GN_PIX_PER_CHUNK, p1 = min(HW, p0 + GN_PIX_PER_CHUNK);
  double s[VEC], q[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) { s[j] = 0.0; q[j] = 0.0; }
  const int vc = threadIdx.x % vpp;    // fixed channel-vector per thread (vpp divides 256 or exceeds it)
  if (vpp <= 256) {
    const int ppi = 256 / vpp;         // pixels per block iteration
    for (int p = p0 + threadIdx.x / vpp; p < p1; p += ppi) {
      float v[VEC];
      loadv<T, VEC>(x + ((long)b * HW + p) * C + vc * VEC, v);
#pragma unroll
      for (int j = 0; j < VEC; ++j) { s[j] += (double)v[j]; q[j] += (double)v[j] * (double)v[j]; }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int g = (vc * VEC + j) / cpg;
      atomicAdd(&gs[g], s[j]); atomicAdd(&gq[g], q[j]);
    }
  } else {  // very wide C: walk channel vectors too
    for (int p = p0; p < p1; ++p)
      for (int v0 = threadIdx.x; v0 < vpp; v0 += 256) {
        float v[VEC];
        loadv<T, VEC>(x + ((long)b * HW + p) * C + v0 * VEC, v);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const int g = (v0 * VEC + j) / cpg;
          atomicAdd(&gs[g], (double)v[j]); atomicAdd(&gq[g], (double)v[j] * (double)v[j]);
        }
      }
  }
  __syncthreads();
  if (threadIdx.x < G) {
    double* o = partial + (((long)b * nchunk + chunk) * G + threadIdx.x) * 2;
    o[0] = gs[threadIdx.x]; o[1] = gq[threadIdx.x];
  }
}

template <typename T, int VEC>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const double* __restrict__ partial,
                                                       int HW, int C, int G, int nchunk, float eps, int silu,
                                                       bf16_t* __restrict__ y_hi, bf16_t* __restrict__ y_lo) {
  __shared__ float sc[2048], sh[2048];
  __shared__ float gmean[64], grstd[64];
  __shared__ double gs_[64], gq_[64];
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int cpg = C / G;
  {
    // fold the image's partials: 256 / G threads per group, each a strided subset in a fixed order, then a lane tree
    const int tpg = 256 / G, g = threadIdx.x / tpg, sub = threadIdx.x % tpg;   // G in {32, 64}: tpg = 8 / 4 lanes of one wave
    double s = 0.0, q = 0.0;
    for (int c = sub; c < nchunk; c += tpg) {
      const double* o = partial + (((long)b * nchunk + c) * G + g) * 2;
      s += o[0]; q += o[1];
    }
    for (int o = 1; o < tpg; o <<= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
    if (sub == 0) { gs_[g] = s; gq_[g] = q; }
  }
  __syncthreads();
  if (threadIdx.x < G) {
    const double s = gs_[threadIdx.x], q = gq_[threadIdx.x];
    const double n = (double)HW * (double)cpg;
    const double mean = s / n;
    double var = q / n - mean * mean;
    if (var < 0.0) var = 0.0;
    gmean[threadIdx.x] = (float)mean;
    grstd[threadIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const int g = c / cpg;
    const float scale = grstd[g] * gamma[c];
    sc[c] = scale;
    sh[c] = beta[c] - scale * gmean[g];
  }
  __syncthreads();
  const int vpp = C / VEC;
  const int p0 = chunk * GN_PIX_PER_CHUNK, p1 = min(HW, p0 + GN_PIX_PER_CHUNK);
  if (vpp <= 256) {
    // each thread keeps ONE channel vector (scale / shift in registers) and strides over pixels: no per-element index
    // math or LDS lookups in the streaming loop
    const int vc = threadIdx.x % vpp, ppi = 256 / vpp;
    float scv[VEC], shv[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { scv[j] = sc[vc * VEC + j]; shv[j] = sh[vc * VEC + j]; }
    // four pixels per thread per iteration, their loads issued together: the kernel is bound by loads in flight per CU, not by
    // bandwidth or VALU (one 16-byte load per lane per iteration measured 4.2 TB/s)
    constexpr int UN = 4;
    for (int pb = p0 + threadIdx.x / vpp; pb < p1; pb += UN * ppi) {
      float v[UN][VEC];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int p = pb + u * ppi;
        if (p < p1) loadv<T, VEC>(x + ((long)b * HW + p) * C + vc * VEC, v[u]);
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int p = pb + u * ppi;
        if (p >= p1) break;
        const long off = ((long)b * HW + p) * C + vc * VEC;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          float t = fmaf(v[u][j], scv[j], shv[j]);
          // operand planes of the bf16x3 convolution: the hardware-reciprocal SiLU the fused convolution uses (same bits on both routes);
          // an f32 / bf16 OUTPUT tensor (the exact-f32 parity mode of the tokenizer among them) keeps the correctly rounded division
          if (silu) t = y_hi ? gn_silu(t) : t / (1.0f + __expf(-t));
          v[u][j] = t;
        }
        if constexpr (VEC == 4) {
          if (y_hi) {  // bf16x3 operand planes for the LDS-DMA convolution (conv_dma.hip) instead of the f32 tensor
            const u32x4 raw = {__float_as_uint(v[u][0]), __float_as_uint(v[u][1]), __float_as_uint(v[u][2]), __float_as_uint(v[u][3])};
            u32x2 hi, lo;
            split4(raw, hi, lo);
            *(u32x2*)(y_hi + off) = hi;
            *(u32x2*)(y_lo + off) = lo;
            continue;
          }
        }
        storev<T, VEC>(y + off, v[u]);
      }
    }
  } else {
    const long nv = (long)(p1 - p0) * vpp;
    for (long i = threadIdx.x; i < nv; i += 256) {
      const int p = p0 + (int)(i / vpp), vc = (int)(i % vpp);
      const long off = ((long)b * HW + p) * C + vc * VEC;
      float v[VEC];
      loadv<T, VEC>(x + off, v);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float t = v[j] * sc[vc * VEC + j] + sh[vc * VEC + j];
        if (silu) t = t / (1.0f + __expf(-t));
        v[j] = t;
      }
      storev<T, VEC>(y + off, v);
    }
  }
}

extern "C" int muse_groupnorm_silu_nhwc(const void* x, void* y, int32_t dtype, const float* gamma, const float* beta,
                                        double* partial, int32_t batch, int32_t HW, int32_t C, int32_t groups, float eps,
                                        int32_t apply_silu, void* stream) {
  const int vec = dtype == MUSE_BF16 ? 8 : 4;
  if ((groups != 32 && groups != 64) || C > 2048 || (C % groups) || (C % vec)) return MUSE_ERR_UNSUPPORTED;
  const int vpp = C / vec;
  if (vpp <= 256 && (256 % vpp)) return MUSE_ERR_UNSUPPORTED;
  if (batch <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const int nchunk = muse_groupnorm_nchunk(HW);
  dim3 grid(nchunk, batch);
  if (dtype == MUSE_F32) {
    hipLaunchKernelGGL((gn_stats_kernel<float, 4>), grid, dim3(256), 0, s, (const float*)x, partial, HW, C, groups);
    hipLaunchKernelGGL((gn_apply_kernel<float, 4>), grid, dim3(256), 0, s, (const float*)x, (float*)y, gamma, beta,
                       (const double*)partial, HW, C, groups, nchunk, eps, apply_silu, (bf16_t*)nullptr, (bf16_t*)nullptr);
  } else {
    hipLaunchKernelGGL((gn_stats_kernel<bf16_t, 8>), grid, dim3(256), 0, s, (const bf16_t*)x, partial, HW, C, groups);
    hipLaunchKernelGGL((gn_apply_kernel<bf16_t, 8>), grid, dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, gamma, beta,
                       (const double*)partial, HW, C, groups, nchunk, eps, apply_silu, (bf16_t*)nullptr, (bf16_t*)nullptr);
  }
  return (int)hipGetLastError();
}

// The split-output apply pass with EIGHT channels per thread (two adjacent 16-byte loads, one 16-byte store per plane instead
// of two 8-byte ones) and four pixels per thread in flight.  Same arithmetic per element as gn_apply_kernel<float, 4> -> the
// same bits.  Needs C % 8 == 0 and 256 % (C / 8) == 0.
__global__ __launch_bounds__(256) void gn_apply_split8_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const double* __restrict__ partial,
                                                              int HW, int C, int G, int nchunk, float eps, int silu,
                                                              bf16_t* __restrict__ y_hi, bf16_t* __restrict__ y_lo) {
  __shared__ float sc[1024], sh[1024];
  __shared__ float gmean[64], grstd[64];
  __shared__ double gs_[64], gq_[64];
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int cpg = C / G;
  {
    const int tpg = 256 / G, g = threadIdx.x / tpg, sub = threadIdx.x % tpg;
    double s = 0.0, q = 0.0;
    for (int c = sub; c < nchunk; c += tpg) {
      const double* o = partial + (((long)b * nchunk + c) * G + g) * 2;
      s += o[0]; q += o[1];
    }
    for (int o = 1; o < tpg; o <<= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
    if (sub == 0) { gs_[g] = s; gq_[g] = q; }
  }
  __syncthreads();
  if (threadIdx.x < G) {
    const double s = gs_[threadIdx.x], q = gq_[threadIdx.x];
    const double n = (double)HW * (double)cpg;
    const double mean = s / n;
    double var = q / n - mean * mean;
    if (var < 0.0) var = 0.0;
    gmean[threadIdx.x] = (float)mean;
    grstd[threadIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const int g = c / cpg;
    const float scale = grstd[g] * gamma[c];
    sc[c] = scale;
    sh[c] = beta[c] - scale * gmean[g];
  }
  __syncthreads();
  const int vpp = C >> 3, vc = threadIdx.x % vpp, ppi = 256 / vpp;
  float scv[8], shv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { scv[j] = sc[vc * 8 + j]; shv[j] = sh[vc * 8 + j]; }
  const int p0 = chunk * GN_PIX_PER_CHUNK, p1 = min(HW, p0 + GN_PIX_PER_CHUNK);
  constexpr int UN = 4;
  for (int pb = p0 + threadIdx.x / vpp; pb < p1; pb += UN * ppi) {
    f32x4 v[UN][2];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int p = pb + u * ppi;
      if (p < p1) {
        const float* src = x + ((long)b * HW + p) * C + vc * 8;
        v[u][0] = *(const f32x4*)src;
        v[u][1] = *(const f32x4*)(src + 4);
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int p = pb + u * ppi;
      if (p >= p1) break;
      const long off = ((long)b * HW + p) * C + vc * 8;
      u32x4 hi4, lo4;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        u32x4 raw;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float t = fmaf(v[u][h][j], scv[h * 4 + j], shv[h * 4 + j]);
          if (silu) t = gn_silu(t);
          raw[j] = __float_as_uint(t);
        }
        u32x2 hi, lo;
        split4(raw, hi, lo);
        hi4[2 * h] = hi[0]; hi4[2 * h + 1] = hi[1];
        lo4[2 * h] = lo[0]; lo4[2 * h + 1] = lo[1];
      }
      *(u32x4*)(y_hi + off) = hi4;
      *(u32x4*)(y_lo + off) = lo4;
    }
  }
}

// The affine form of GroupNorm for a consumer that applies it itself (muse_conv2d_nhwc_gn_split2): scale[b][c] = rstd * gamma[c],
// shift[b][c] = beta[c] - scale * mean from the [B, nchunk, G, 2] partial sums; one block per image, the arithmetic of the apply
// kernels' preamble (double sums, float mean / rstd) -> the same floats.
__global__ __launch_bounds__(256) void gn_scale_shift_kernel(const double* __restrict__ partial, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ scale,
                                                             float* __restrict__ shift, int HW, int C, int G, int nchunk, float eps) {
  __shared__ float gmean[64], grstd[64];
  __shared__ double gs_[64], gq_[64];
  const int b = blockIdx.x, cpg = C / G;
  {
    const int tpg = 256 / G, g = threadIdx.x / tpg, sub = threadIdx.x % tpg;
    double s = 0.0, q = 0.0;
    for (int c = sub; c < nchunk; c += tpg) {
      const double* o = partial + (((long)b * nchunk + c) * G + g) * 2;
      s += o[0]; q += o[1];
    }
    for (int o = 1; o < tpg; o <<= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
    if (sub == 0) { gs_[g] = s; gq_[g] = q; }
  }
  __syncthreads();
  if (threadIdx.x < G) {
    const double s = gs_[threadIdx.x], q = gq_[threadIdx.x];
    const double n = (double)HW * (double)cpg;
    const double mean = s / n;
    double var = q / n - mean * mean;
    if (var < 0.0) var = 0.0;
    gmean[threadIdx.x] = (float)mean;
    grstd[threadIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const int g = c / cpg;
    const float sc = grstd[g] * gamma[c];
    scale[(long)b * C + c] = sc;
    shift[(long)b * C + c] = beta[c] - sc * gmean[g];
  }
}
extern "C" int muse_groupnorm_scale_shift(const double* partial, int32_t nchunk, const float* gamma, const float* beta, float* scale,
                                          float* shift, int32_t batch, int32_t HW, int32_t C, int32_t groups, float eps, void* stream) {
  if ((groups != 32 && groups != 64) || (C % groups) || nchunk <= 0) return MUSE_ERR_UNSUPPORTED;
  if (batch <= 0) return 0;
  hipLaunchKernelGGL(gn_scale_shift_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, partial, gamma, beta, scale, shift, HW, C,
                     groups, nchunk, eps);
  return (int)hipGetLastError();
}

// f32 input, output as the two bf16 planes y_hi = bf16(y), y_lo = bf16(y - y_hi) the bf16x3 LDS-DMA convolution reads.
// stats_nchunk == 0: the statistics pass runs here; > 0: `partial` already holds [B, stats_nchunk, G, 2] sums written by the
// producing convolution's epilogue (muse_conv2d_nhwc_split2 with gn_partial) and only the apply pass runs.
extern "C" int muse_groupnorm_silu_nhwc_split(const float* x, void* y_hi, void* y_lo, const float* gamma, const float* beta,
                                              double* partial, int32_t stats_nchunk, int32_t batch, int32_t HW, int32_t C,
                                              int32_t groups, float eps, int32_t apply_silu, void* stream) {
  if ((groups != 32 && groups != 64) || C > 1024 || (C % groups) || (C % 4)) return MUSE_ERR_UNSUPPORTED;
  const int vpp = C / 4;
  if (256 % vpp) return MUSE_ERR_UNSUPPORTED;  // (the split store lives in the one-vector-per-thread loop)
  if (batch <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const int nchunk = muse_groupnorm_nchunk(HW);
  dim3 grid(nchunk, batch);
  if (stats_nchunk <= 0) hipLaunchKernelGGL((gn_stats_kernel<float, 4>), grid, dim3(256), 0, s, x, partial, HW, C, groups);
  static const int wide = []() { const char* e = getenv("MUSE_GN_SPLIT8"); return e ? atoi(e) : 1; }();
  if (wide && (C % 8) == 0 && (256 % (C / 8)) == 0 && !((((uintptr_t)x) | ((uintptr_t)y_hi) | ((uintptr_t)y_lo)) & 15))
    hipLaunchKernelGGL(gn_apply_split8_kernel, grid, dim3(256), 0, s, x, gamma, beta, (const double*)partial, HW, C, groups,
                       stats_nchunk > 0 ? stats_nchunk : nchunk, eps, apply_silu, (bf16_t*)y_hi, (bf16_t*)y_lo);
  else
    hipLaunchKernelGGL((gn_apply_kernel<float, 4>), grid, dim3(256), 0, s, x, (float*)nullptr, gamma, beta, (const double*)partial,
                       HW, C, groups, stats_nchunk > 0 ? stats_nchunk : nchunk, eps, apply_silu, (bf16_t*)y_hi, (bf16_t*)y_lo);
  return (int)hipGetLastError();
}

// =================================================================================================================
// The first convolution of an encoder (conv_in, muse/modeling_maskgit_vqgan.py:175: 3 image channels -> hidden_channels, 3x3,
// padding 1) as a direct f32 convolution on the vector ALUs: K = 27 is no matrix-core problem - the implicit-GEMM kernels spend the
// launch on their 128 x 128 epilogue and write the 2.1 GB output of a 64 x 256 x 256 batch in partial lines (1.51 ms).  Here a block
// owns one image row: its three input rows go to LDS once, a thread keeps the 27 x 4 weights of its four output channels in registers
// (as pairs: one v_pk_fma_f32 feeds two accumulators) and walks the row's pixels; the 32 lanes of a pixel read the same nine LDS
// vectors and store 512 contiguous bytes.  Exact f32 products.  0.76 ms on that batch with the f64 GroupNorm sums (0.63 without; a
// plain 2.1 GB fill takes 0.31 ms).  The first version read its inputs from global memory a pixel ahead and was bound by their
// latency at two waves per SIMD (1.3 ms whatever the ALU count).
// x: [B, H, W, Cpad] f32 (channels >= Cin are ignored), w4: [Cout][9][4] f32 (channel 3 zero when Cin == 3), out [B, H, W, Cout].
// gn_partial (optional): [B, H, groups, 2] f64 sums of the output, one chunk per image row (groups of exactly four channels).
// =================================================================================================================
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
template <int CIN>
__global__ __launch_bounds__(256) void conv_in_direct_kernel(const float* __restrict__ x, const float* __restrict__ w4,
                                                             const float* __restrict__ bias, float* __restrict__ out,
                                                             double* __restrict__ gn_partial, int H, int W, int Cpad, int Cout) {
  __shared__ double red[2][256];
  const int qpp = Cout >> 2;                                   // channel quads (= threads) per pixel: divides 256
  const int cq = threadIdx.x % qpp, px0 = threadIdx.x / qpp, pstep = 256 / qpp;
  const int row = blockIdx.x, yy = row % H;                    // row = b * H + y
  // weights of this thread's four output channels as PAIRS (j, j+1): one v_pk_fma_f32 feeds two accumulators from one input value
  f32x2_t wk[9][CIN][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < CIN; ++c)
#pragma unroll
      for (int jp = 0; jp < 2; ++jp)
        wk[t][c][jp] = f32x2_t{w4[((long)(cq * 4 + 2 * jp) * 9 + t) * 4 + c], w4[((long)(cq * 4 + 2 * jp + 1) * 9 + t) * 4 + c]};
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (bias) bv = *(const f32x4*)(bias + cq * 4);
  // the three input rows of this image row (first four channels of every pixel, one zero pixel either side, zero rows outside the
  // image) are staged in LDS ONCE: the per-pixel loop then has no global load in it - with one pixel of look-ahead and two waves per
  // SIMD it was bound by the memory latency of its nine loads per pixel (1.3 ms on the 64 x 256 x 256 batch, whatever the ALU count)
  extern __shared__ __attribute__((aligned(16))) unsigned char cin_smem[];
  f32x4* xs = (f32x4*)cin_smem;                               // [3][W + 2]
  for (int i = threadIdx.x; i < 3 * (W + 2); i += 256) {
    const int ky = i / (W + 2), ix = i - ky * (W + 2) - 1;
    const bool ok = (ky == 1 || (ky == 0 ? yy > 0 : yy + 1 < H)) && ix >= 0 && ix < W;
    xs[i] = ok ? *(const f32x4*)(x + ((long)(row + ky - 1) * W + ix) * Cpad) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();
  double gs = 0.0, gq = 0.0;
  auto pixel = [&](int xx) {
    f32x2_t a01 = {bv[0], bv[1]}, a23 = {bv[2], bv[3]};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const f32x4 v = xs[ky * (W + 2) + xx + kx];           // pixel xx + kx - 1 of input row ky (all lanes of a pixel: one address)
#pragma unroll
        for (int c = 0; c < CIN; ++c) {
          const f32x2_t in2 = {v[c], v[c]};
          a01 = __builtin_elementwise_fma(in2, wk[ky * 3 + kx][c][0], a01);
          a23 = __builtin_elementwise_fma(in2, wk[ky * 3 + kx][c][1], a23);
        }
      }
    const f32x4 acc = {a01[0], a01[1], a23[0], a23[1]};
    *(f32x4*)(out + ((long)row * W + xx) * Cout + cq * 4) = acc;
    if (gn_partial) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { gs += (double)acc[j]; gq += (double)acc[j] * (double)acc[j]; }
    }
  };
  for (int xx = px0; xx < W; xx += pstep) pixel(xx);
  if (gn_partial) {      // one group = one channel quad: fold the pstep pixel lanes of each quad in a fixed order
    red[0][threadIdx.x] = gs; red[1][threadIdx.x] = gq;
    __syncthreads();
    if (threadIdx.x < qpp) {
      double s = 0.0, q = 0.0;
      for (int k = 0; k < pstep; ++k) { s += red[0][k * qpp + threadIdx.x]; q += red[1][k * qpp + threadIdx.x]; }
      double* o = gn_partial + ((long)row * qpp + threadIdx.x) * 2;
      o[0] = s; o[1] = q;
    }
  }
}
extern "C" int muse_conv_in_direct(const float* x, const float* w4, const float* bias, float* out, double* gn_partial, int32_t gn_groups,
                                   int32_t batch, int32_t H, int32_t W, int32_t Cin, int32_t Cpad, int32_t Cout, void* stream) {
  const int qpp = Cout >> 2;
  if ((Cout & 3) || qpp > 256 || (256 % qpp) || Cpad < 4 || (Cpad & 3) || Cin < 1 || Cin > 4) return MUSE_ERR_UNSUPPORTED;
  if (gn_partial && gn_groups * 4 != Cout) return MUSE_ERR_UNSUPPORTED;      // groups of exactly four channels
  if ((((uintptr_t)x) | ((uintptr_t)w4) | ((uintptr_t)bias) | ((uintptr_t)out)) & 15) return MUSE_ERR_ALIGN;
  if ((long)batch * H <= 0) return 0;
  if ((long)batch * H >= (1L << 31)) return MUSE_ERR_UNSUPPORTED;
  const size_t lds = (size_t)3 * (W + 2) * 16;
  if (lds > 48 * 1024) return MUSE_ERR_UNSUPPORTED;              // (W <= 1022)
#define CID(N) hipLaunchKernelGGL(conv_in_direct_kernel<N>, dim3((unsigned)(batch * H)), dim3(256), lds, (hipStream_t)stream, x, w4, bias, out, \
                                  gn_partial, H, W, Cpad, Cout)
  if (Cin == 3) CID(3); else if (Cin == 4) CID(4); else if (Cin == 1) CID(1); else CID(2);
#undef CID
  return (int)hipGetLastError();
}

// =================================================================================================================
// The LAST convolution of a decoder (conv_out, muse/modeling_maskgit_vqgan.py:236-240: norm_out -> swish -> 3x3, hidden_channels -> 3
// image channels) as a direct exact-f32 convolution that normalises and activates its own input.  Three output channels are no
// matrix-core problem: the implicit-GEMM kernels compute 128-wide channel tiles for 3 live columns (4.8 ms on a 64 x 256 x 256 batch,
// behind a 1.2 ms GroupNorm statistics + apply pass of the 2.1 GB input).  Here a block owns a 16 x 16 pixel patch; per 32-channel
// chunk the 18 x 18 halo'd patch is staged in LDS ONCE as silu(x * scale[b, c] + shift[b, c]) (the GroupNorm in its per-image affine
// form, muse_groupnorm_scale_shift from the producer's partial sums; zero padding AFTER the activation, like the reference pads the
// activated tensor) with a 36-float pixel stride (conflict-free 16-byte reads across the 16 pixels of a row), the weights of the
// chunk next to it (every lane reads the same address: broadcast), and each thread accumulates its pixel's COUT outputs over the
// nine taps x 32 channels with plain f32 fmas.  Input read once (+ 27 % halo), output written once.
// x [B, H, W, C] f32, scale / shift [B, C] f32, w [COUT][9][C] f32, out [B, H, W, COUT] f32.  H, W % 16 == 0, C % 32 == 0.
// =================================================================================================================
template <int COUT>
__global__ __launch_bounds__(256) void conv_out_direct_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ out,
                                                              int H, int W, int C) {
  constexpr int PS = 36;                                   // floats per staged pixel (32 channels + 4 pad)
  __shared__ __attribute__((aligned(16))) float xs[18 * 18 * PS];
  __shared__ __attribute__((aligned(16))) float ws[COUT * 9 * 32];
  const int tiles_x = W >> 4, tiles_y = H >> 4;
  const int t = blockIdx.x, b = t / (tiles_x * tiles_y), ti = t - b * (tiles_x * tiles_y);
  const int y0 = (ti / tiles_x) << 4, x0 = (ti % tiles_x) << 4;
  const int py = threadIdx.x >> 4, px = threadIdx.x & 15;
  float acc[COUT];
#pragma unroll
  for (int o = 0; o < COUT; ++o) acc[o] = bias ? bias[o] : 0.f;
  for (int c0 = 0; c0 < C; c0 += 32) {
    __syncthreads();                                       // the previous chunk's reads are done
    // stage the 18 x 18 x 32 patch: 324 pixels x 8 float4 = 2592 vectors over 256 threads
    for (int i = threadIdx.x; i < 18 * 18 * 8; i += 256) {
      const int pix = i >> 3, q = i & 7;
      const int yy = y0 + pix / 18 - 1, xx = x0 + pix % 18 - 1;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
        const f32x4 r = *(const f32x4*)(x + (((long)b * H + yy) * W + xx) * C + c0 + q * 4);
        const f32x4 sc = *(const f32x4*)(scale + (long)b * C + c0 + q * 4);
        const f32x4 sh = *(const f32x4*)(shift + (long)b * C + c0 + q * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = gn_silu(fmaf(r[j], sc[j], sh[j]));
      }
      *(f32x4*)(xs + pix * PS + q * 4) = v;
    }
    for (int i = threadIdx.x; i < COUT * 9 * 8; i += 256) {
      const int ot = i >> 3, q = i & 7;                    // ot = o * 9 + tap
      *(f32x4*)(ws + ot * 32 + q * 4) = *(const f32x4*)(w + (long)ot * C + c0 + q * 4);
    }
    __syncthreads();
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const float* xp = xs + ((py + ky) * 18 + px + kx) * PS;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const f32x4 v = *(const f32x4*)(xp + q * 4);
#pragma unroll
          for (int o = 0; o < COUT; ++o) {
            const f32x4 wv = *(const f32x4*)(ws + (o * 9 + ky * 3 + kx) * 32 + q * 4);
            acc[o] = fmaf(v[0], wv[0], acc[o]);
            acc[o] = fmaf(v[1], wv[1], acc[o]);
            acc[o] = fmaf(v[2], wv[2], acc[o]);
            acc[o] = fmaf(v[3], wv[3], acc[o]);
          }
        }
      }
  }
  float* op = out + (((long)b * H + y0 + py) * W + x0 + px) * COUT;
#pragma unroll
  for (int o = 0; o < COUT; ++o) op[o] = acc[o];
}
extern "C" int muse_conv_out_direct(const float* x, const float* scale, const float* shift, const float* w, const float* bias, float* out,
                                    int32_t batch, int32_t H, int32_t W, int32_t C, int32_t Cout, void* stream) {
  if ((H & 15) || (W & 15) || (C & 31) || C < 32 || Cout < 1 || Cout > 4) return MUSE_ERR_UNSUPPORTED;
  if ((((uintptr_t)x) | ((uintptr_t)scale) | ((uintptr_t)shift) | ((uintptr_t)w)) & 15) return MUSE_ERR_ALIGN;
  const long tiles = (long)batch * (H >> 4) * (W >> 4);
  if (tiles <= 0) return 0;
  if (tiles >= (1L << 31)) return MUSE_ERR_UNSUPPORTED;
#define COD(N) hipLaunchKernelGGL(conv_out_direct_kernel<N>, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, x, scale, shift, w, bias, out, H, W, C)
  if (Cout == 3) COD(3); else if (Cout == 4) COD(4); else if (Cout == 1) COD(1); else COD(2);
#undef COD
  return (int)hipGetLastError();
}

// =================================================================================================================
// F.interpolate(scale_factor=2, mode="nearest") of an f32 NHWC tensor written directly as the (hi, lo) bf16 operand planes of the
// bf16x3 patch-slab convolution that follows it (UpsamplingBlock, muse/modeling_maskgit_vqgan.py:141-149): the up-sampling
// convolutions of the decoder then run on the LDS-DMA kernel like every other 3 x 3 layer instead of the register-staged
// muse_conv2d_nhwc_split (5.1 / 4.6 ms against ~3.1 ms for the same flops on the 64 x 256 x 256 / 128 x 128 layers).
// x [B, H, W, C] f32 -> y_hi, y_lo [B, 2H, 2W, C] bf16.  Eight channels per thread: two 16-byte loads, four 16-byte stores per plane.
// =================================================================================================================
__global__ __launch_bounds__(256) void upsample2x_split_kernel(const float* __restrict__ x, bf16_t* __restrict__ y_hi,
                                                               bf16_t* __restrict__ y_lo, long npix, int H, int W, int C) {
  const int vpp = C >> 3;
  const long n = npix * vpp;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int vc = (int)(i % vpp);
    long p = i / vpp;
    const int xx = (int)(p % W); p /= W;
    const int yy = (int)(p % H);
    const long b = p / H;
    const float* src = x + (((b * H + yy) * W + xx) * (long)C) + vc * 8;
    const u32x4 v0 = *(const u32x4*)src, v1 = *(const u32x4*)(src + 4);
    u32x2 h0, l0, h1, l1;
    split4(v0, h0, l0);
    split4(v1, h1, l1);
    const u32x4 hv = {h0[0], h0[1], h1[0], h1[1]}, lv = {l0[0], l0[1], l1[0], l1[1]};
    const long W2 = 2L * W;
    const long o00 = (((b * 2 * H + 2 * yy) * W2 + 2 * xx) * (long)C) + vc * 8;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const long o = o00 + ((long)dy * W2 + dx) * C;
        *(u32x4*)(y_hi + o) = hv;
        *(u32x4*)(y_lo + o) = lv;
      }
  }
}
// plain nearest x2 of an NHWC tensor (taming Upsample with resample_with_conv = False, muse/modeling_taming_vqgan.py:36-47)
template <typename T, int VEC>
__global__ __launch_bounds__(256) void upsample2x_kernel(const T* __restrict__ x, T* __restrict__ y, long npix, int H, int W, int C) {
  const int vpp = C / VEC;
  const long n = npix * vpp;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int vc = (int)(i % vpp);
    long p = i / vpp;
    const int xx = (int)(p % W); p /= W;
    const int yy = (int)(p % H);
    const long b = p / H;
    const u32x4 v = *(const u32x4*)(x + (((b * H + yy) * W + xx) * (long)C) + vc * VEC);
    const long W2 = 2L * W;
    const long o00 = (((b * 2 * H + 2 * yy) * W2 + 2 * xx) * (long)C) + vc * VEC;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) *(u32x4*)(y + o00 + ((long)dy * W2 + dx) * C) = v;
  }
}
extern "C" int muse_upsample2x_nhwc(const void* x, void* y, int32_t dtype, int32_t batch, int32_t H, int32_t W, int32_t C, void* stream) {
  const int vec = dtype == MUSE_BF16 ? 8 : 4;
  if (C % vec) return MUSE_ERR_UNSUPPORTED;
  if ((((uintptr_t)x) | ((uintptr_t)y)) & 15) return MUSE_ERR_ALIGN;
  const long npix = (long)batch * H * W;
  if (npix <= 0) return 0;
  long g = (npix * (C / vec) + 255) / 256; if (g > 65536) g = 65536;
  if (dtype == MUSE_F32) hipLaunchKernelGGL((upsample2x_kernel<float, 4>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, npix, H, W, C);
  else hipLaunchKernelGGL((upsample2x_kernel<bf16_t, 8>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, npix, H, W, C);
  return (int)hipGetLastError();
}
extern "C" int muse_upsample2x_split_nhwc(const float* x, void* y_hi, void* y_lo, int32_t batch, int32_t H, int32_t W, int32_t C,
                                          void* stream) {
  if (C & 7) return MUSE_ERR_UNSUPPORTED;
  if ((((uintptr_t)x) | ((uintptr_t)y_hi) | ((uintptr_t)y_lo)) & 15) return MUSE_ERR_ALIGN;
  const long npix = (long)batch * H * W;
  if (npix <= 0) return 0;
  long g = (npix * (C >> 3) + 255) / 256; if (g > 65536) g = 65536;
  hipLaunchKernelGGL(upsample2x_split_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y_hi, (bf16_t*)y_lo, npix, H, W, C);
  return (int)hipGetLastError();
}

// =================================================================================================================
// avg_pool2d(2,2) NHWC
// =================================================================================================================
template <typename T, int VEC>
__global__ void avgpool_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C) {
  const int oh = H >> 1, ow = W >> 1, vpp = C / VEC;
  const long n = (long)B * oh * ow * vpp;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int vc = (int)(i % vpp);
    long t = i / vpp;
    const int ox = (int)(t % ow); t /= ow;
    const int oy = (int)(t % oh);
    const int b = (int)(t / oh);
    const T* src = x + (((long)b * H + 2 * oy) * W + 2 * ox) * C + vc * VEC;
    float a[VEC], c0[VEC], c1[VEC], c2[VEC];
    loadv<T, VEC>(src, a); loadv<T, VEC>(src + C, c0); loadv<T, VEC>(src + (long)W * C, c1); loadv<T, VEC>(src + (long)W * C + C, c2);
#pragma unroll
    for (int j = 0; j < VEC; ++j) a[j] = (((a[j] + c0[j]) + c1[j]) + c2[j]) * 0.25f;
    storev<T, VEC>(y + (((long)b * oh + oy) * ow + ox) * C + vc * VEC, a);
  }
}
extern "C" int muse_avgpool2x2_nhwc(const void* x, void* y, int32_t dtype, int32_t batch, int32_t H, int32_t W, int32_t C,
                                    void* stream) {
  const int vec = dtype == MUSE_BF16 ? 8 : 4;
  if ((C % vec) || ((H | W) & 1)) return MUSE_ERR_BAD_ARG;
  const long n = (long)batch * (H / 2) * (W / 2) * (C / vec);
  if (n <= 0) return 0;
  long g = (n + 255) / 256; if (g > 8192) g = 8192;
  if (dtype == MUSE_F32) hipLaunchKernelGGL((avgpool_kernel<float, 4>), dim3((int)g), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, batch, H, W, C);
  else hipLaunchKernelGGL((avgpool_kernel<bf16_t, 8>), dim3((int)g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, batch, H, W, C);
  return (int)hipGetLastError();
}

// pooled output + its GroupNorm statistics: block (chunk, b) produces output pixels [chunk*1024, +1024) of image b; a thread
// keeps one 4-channel vector and strides over pixels (same decomposition and partial layout as gn_stats_kernel)
__global__ __launch_bounds__(256) void avgpool_stats_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            double* __restrict__ partial, int H, int W, int C, int G) {
  __shared__ double gs[64], gq[64];
  const int chunk = blockIdx.x, b = blockIdx.y, nchunk = gridDim.x;
  if (threadIdx.x < G) { gs[threadIdx.x] = 0.0; gq[threadIdx.x] = 0.0; }
  __syncthreads();
  const int oh = H >> 1, ow = W >> 1, oHW = oh * ow, vpp = C >> 2, cpg = C / G;
  const int p0 = chunk * GN_PIX_PER_CHUNK, p1 = min(oHW, p0 + GN_PIX_PER_CHUNK);
  const int vc = threadIdx.x % vpp, ppi = 256 / vpp;
  double s[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0};
  constexpr int UN = 2;
  for (int pb = p0 + threadIdx.x / vpp; pb < p1; pb += UN * ppi) {
    float a[UN][4], c0[UN][4], c1[UN][4], c2[UN][4];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int p = pb + u * ppi;
      if (p < p1) {
        const int oy = p / ow, ox = p - oy * ow;
        const float* src = x + (((long)b * H + 2 * oy) * W + 2 * ox) * C + vc * 4;
        loadv<float, 4>(src, a[u]); loadv<float, 4>(src + C, c0[u]);
        loadv<float, 4>(src + (long)W * C, c1[u]); loadv<float, 4>(src + (long)W * C + C, c2[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int p = pb + u * ppi;
      if (p >= p1) break;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a[u][j] = (((a[u][j] + c0[u][j]) + c1[u][j]) + c2[u][j]) * 0.25f;
        s[j] += (double)a[u][j]; q[j] += (double)a[u][j] * (double)a[u][j];
      }
      storev<float, 4>(y + ((long)b * oHW + p) * C + vc * 4, a[u]);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int g = (vc * 4 + j) / cpg;
    atomicAdd(&gs[g], s[j]); atomicAdd(&gq[g], q[j]);
  }
  __syncthreads();
  if (threadIdx.x < G) {
    double* o = partial + (((long)b * nchunk + chunk) * G + threadIdx.x) * 2;
    o[0] = gs[threadIdx.x]; o[1] = gq[threadIdx.x];
  }
}
extern "C" int muse_avgpool2x2_nhwc_stats(const float* x, float* y, double* partial, int32_t groups, int32_t batch, int32_t H,
                                          int32_t W, int32_t C, void* stream) {
  if ((C % 4) || ((H | W) & 1)) return MUSE_ERR_BAD_ARG;
  const int vpp = C / 4;
  if ((groups != 32 && groups != 64) || (C % groups) || vpp > 256 || (256 % vpp)) return MUSE_ERR_UNSUPPORTED;
  if (batch <= 0 || H <= 0 || W <= 0) return 0;
  const int nchunk = muse_groupnorm_nchunk((H / 2) * (W / 2));
  hipLaunchKernelGGL(avgpool_stats_kernel, dim3(nchunk, batch), dim3(256), 0, (hipStream_t)stream, x, y, partial, H, W, C, groups);
  return (int)hipGetLastError();
}

// =================================================================================================================
// layout conversion NCHW f32 <-> NHWC (f32 | bf16) with zero channel padding
// =================================================================================================================
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, T* __restrict__ out, int C, int HW, int Cpad, long npix) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
    const long b = i / HW; const int p = (int)(i - b * HW);
    for (int c = 0; c < Cpad; ++c) {
      const float v = c < C ? in[(b * C + c) * HW + p] : 0.f;
      Elem<T>::store(out + i * Cpad + c, v);
    }
  }
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ in, float* __restrict__ out, int C, int HW, int Cpad, long npix) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
    const long b = i / HW; const int p = (int)(i - b * HW);
    for (int c = 0; c < C; ++c) out[(b * C + c) * HW + p] = Elem<T>::load(in + i * Cpad + c);
  }
}
extern "C" int muse_nchw_to_nhwc(const float* in, void* out, int32_t out_dtype, int32_t batch, int32_t C, int32_t HW,
                                 int32_t Cpad, void* stream) {
  const long npix = (long)batch * HW;
  if (npix <= 0) return 0;
  long g = (npix + 255) / 256; if (g > 8192) g = 8192;
  if (out_dtype == MUSE_F32) hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3((int)g), dim3(256), 0, (hipStream_t)stream, in, (float*)out, C, HW, Cpad, npix);
  else hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3((int)g), dim3(256), 0, (hipStream_t)stream, in, (bf16_t*)out, C, HW, Cpad, npix);
  return (int)hipGetLastError();
}
extern "C" int muse_nhwc_to_nchw(const void* in, int32_t in_dtype, float* out, int32_t batch, int32_t C, int32_t HW,
                                 int32_t Cpad, void* stream) {
  const long npix = (long)batch * HW;
  if (npix <= 0) return 0;
  long g = (npix + 255) / 256; if (g > 8192) g = 8192;
  if (in_dtype == MUSE_F32) hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3((int)g), dim3(256), 0, (hipStream_t)stream, (const float*)in, out, C, HW, Cpad, npix);
  else hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3((int)g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, out, C, HW, Cpad, npix);
  return (int)hipGetLastError();
}

// =================================================================================================================
// VQ helpers
// =================================================================================================================
__global__ __launch_bounds__(256) void argmin_rows_kernel(const float* __restrict__ d, int64_t* __restrict__ idx, long rows, int n, long ld) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* r = d + row * ld;
  float best = INFINITY; int bi = 0x7fffffff;
  for (int c = lane; c < n; c += 64) { const float v = r[c]; if (v < best) { best = v; bi = c; } }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
    if (ov < best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) idx[row] = (int64_t)bi;
}
extern "C" int muse_argmin_rows(const float* dist, int64_t* idx, int64_t rows, int32_t ncodes, int64_t ld, void* stream) {
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(argmin_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dist, idx, (long)rows, ncodes, (long)ld);
  return (int)hipGetLastError();
}

__global__ __launch_bounds__(256) void row_sumsq_kernel(const float* __restrict__ x, float* __restrict__ out, long rows, int cols, long ld) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  double s = 0.0;
  for (int c = lane; c < cols; c += 64) { const double v = (double)x[row * ld + c]; s += v * v; }
  s = wave_sum_d(s);
  if (lane == 0) out[row] = (float)s;
}
extern "C" int muse_row_sumsq(const float* x, float* out, int64_t rows, int32_t cols, int64_t ld, void* stream) {
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(row_sumsq_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, out, (long)rows, cols, (long)ld);
  return (int)hipGetLastError();
}

template <typename T>
__global__ void gather_rows_kernel(const float* __restrict__ table, const int64_t* __restrict__ idx, T* __restrict__ out, long rows, int cols) {
  const long r = blockIdx.x;
  const int64_t id = idx[r];
  for (int c = threadIdx.x; c < cols; c += blockDim.x) Elem<T>::store(out + r * cols + c, table[id * cols + c]);
}
extern "C" int muse_gather_rows(const float* table, const int64_t* idx, void* out, int32_t out_dtype, int64_t rows, int32_t cols,
                                void* stream) {
  if (rows <= 0) return 0;
  if (out_dtype == MUSE_F32) hipLaunchKernelGGL(gather_rows_kernel<float>, dim3((unsigned)rows), dim3(64), 0, (hipStream_t)stream, table, idx, (float*)out, (long)rows, cols);
  else hipLaunchKernelGGL(gather_rows_kernel<bf16_t>, dim3((unsigned)rows), dim3(64), 0, (hipStream_t)stream, table, idx, (bf16_t*)out, (long)rows, cols);
  return (int)hipGetLastError();
}

// =================================================================================================================
// 2-D transpose (fallback path only) and the tr16 probe
// =================================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ in, T* __restrict__ out, int rows, int cols, long ldi,
                                                        long ldo, long si, long so) {
  __shared__ T tile[32][33];
  const T* ib = in + (long)blockIdx.z * si;
  T* ob = out + (long)blockIdx.z * so;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int j = ty; j < 32; j += 8) {
    const int r = r0 + j, c = c0 + tx;
    if (r < rows && c < cols) tile[j][tx] = ib[(long)r * ldi + c];
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, r = r0 + tx;
    if (r < rows && c < cols) ob[(long)c * ldo + r] = tile[tx][j];
  }
}
extern "C" int muse_transpose(const void* in, void* out, int32_t dtype, int32_t rows, int32_t cols, int64_t ld_in,
                              int64_t ld_out, int32_t batch, int64_t stride_in, int64_t stride_out, void* stream) {
  if (rows <= 0 || cols <= 0 || batch <= 0) return 0;
  dim3 grid((cols + 31) / 32, (rows + 31) / 32, batch);
  if (dtype == MUSE_F32) hipLaunchKernelGGL(transpose_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)in, (float*)out, rows, cols, (long)ld_in, (long)ld_out, (long)stride_in, (long)stride_out);
  else hipLaunchKernelGGL(transpose_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, (bf16_t*)out, rows, cols, (long)ld_in, (long)ld_out, (long)stride_in, (long)stride_out);
  return (int)hipGetLastError();
}

__global__ void probe_tr16_kernel(const int* __restrict__ addr, int* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const unsigned char* a = (const unsigned char*)lds + addr[threadIdx.x];
  const s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a);
#pragma unroll
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (int)(unsigned short)r[j];
}
extern "C" int muse_probe_tr16(const int32_t* addr, int32_t* out, void* stream) {
  hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, addr, out);
  return (int)hipGetLastError();
}
