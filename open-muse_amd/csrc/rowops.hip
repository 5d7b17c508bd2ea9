// Row / element kernels of the MaskGitTransformer path: LayerNorm, softmax, GLU, GELU, cross-entropy, embedding,
// AdamW, casts, mask sampling.  All HBM-bound: 8/16-byte vector accesses, wave64 shuffle reductions, one wave per
// row where a row fits a wave's registers.
#include "common.h"
#include "../../include/muse_hip.h"

// ---- 4-element vector access helpers ---------------------------------------------------------------------------
template <typename T> struct V4;
template <> struct V4<float> {
  typedef f32x4 raw;
  static __device__ __forceinline__ raw load_raw(const float* p) { return *(const f32x4*)p; }
  static __device__ __forceinline__ void unpack(const raw& t, float (&v)[4]) { v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
  static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
    const f32x4 t = *(const f32x4*)p; v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[4]) { *(f32x4*)p = f32x4{v[0], v[1], v[2], v[3]}; }
};
template <> struct V4<bf16_t> {
  typedef u32x2 raw;
  static __device__ __forceinline__ raw load_raw(const bf16_t* p) { return *(const u32x2*)p; }
  static __device__ __forceinline__ void unpack(const raw& t, float (&v)[4]) {
    v[0] = __uint_as_float(t[0] << 16); v[1] = __uint_as_float(t[0] & 0xffff0000u);
    v[2] = __uint_as_float(t[1] << 16); v[3] = __uint_as_float(t[1] & 0xffff0000u);
  }
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[4]) {
    const u32x2 t = *(const u32x2*)p;
    v[0] = __uint_as_float(t[0] << 16); v[1] = __uint_as_float(t[0] & 0xffff0000u);
    v[2] = __uint_as_float(t[1] << 16); v[3] = __uint_as_float(t[1] & 0xffff0000u);
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[4]) {
    u32x2 t;
    t[0] = pack2_bf16(v[0], v[1]);
    t[1] = pack2_bf16(v[2], v[3]);
    *(u32x2*)p = t;
  }
};

// =================================================================================================================
// LayerNorm (weight only).  One wave per row, 4 rows per 256-thread block.
// =================================================================================================================
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const TI* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ res, TO* __restrict__ y,
                                                     float* __restrict__ mean_o, float* __restrict__ rstd_o, int rows,
                                                     int cols, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const TI* xr = x + (long)row * cols;
  float s = 0.f;
  for (int c = lane * 4; c < cols; c += 256) { float v[4]; V4<TI>::load(xr + c, v); s += (v[0] + v[1]) + (v[2] + v[3]); }
  const float mean = wave_sum(s) / (float)cols;
  float q = 0.f;
  for (int c = lane * 4; c < cols; c += 256) {
    float v[4]; V4<TI>::load(xr + c, v);
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float d = v[j] - mean; q = fmaf(d, d, q); }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)cols + eps);
  if (lane == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
  for (int c = lane * 4; c < cols; c += 256) {
    float v[4], g[4], o[4];
    V4<TI>::load(xr + c, v); V4<float>::load(w + c, g);
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (v[j] - mean) * rstd * g[j];
    if (res) { float r[4]; V4<float>::load(res + (long)row * cols + c, r);
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] += r[j]; }
    V4<TO>::store(y + (long)row * cols + c, o);
  }
}

template <typename TI, typename TO>
static int ln_fwd_launch(const void* x, const float* w, const float* res, void* y, float* mean, float* rstd, int rows,
                         int cols, float eps, hipStream_t s) {
  hipLaunchKernelGGL((ln_fwd_kernel<TI, TO>), dim3((rows + 3) / 4), dim3(256), 0, s, (const TI*)x, w, res, (TO*)y, mean, rstd,
                     rows, cols, eps);
  return (int)hipGetLastError();
}

extern "C" int muse_layernorm_fwd(const void* x, int32_t x_dtype, const float* w, const float* residual, void* y,
                                  int32_t y_dtype, float* mean, float* rstd, int32_t rows, int32_t cols, float eps,
                                  void* stream) {
  if (cols % 4 || rows <= 0) return rows == 0 ? 0 : MUSE_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (x_dtype == MUSE_F32 && y_dtype == MUSE_F32) return ln_fwd_launch<float, float>(x, w, residual, y, mean, rstd, rows, cols, eps, s);
  if (x_dtype == MUSE_F32 && y_dtype == MUSE_BF16) return ln_fwd_launch<float, bf16_t>(x, w, residual, y, mean, rstd, rows, cols, eps, s);
  if (x_dtype == MUSE_BF16 && y_dtype == MUSE_F32) return ln_fwd_launch<bf16_t, float>(x, w, residual, y, mean, rstd, rows, cols, eps, s);
  if (x_dtype == MUSE_BF16 && y_dtype == MUSE_BF16) return ln_fwd_launch<bf16_t, bf16_t>(x, w, residual, y, mean, rstd, rows, cols, eps, s);
  return MUSE_ERR_BAD_ARG;
}

// backward: 16 rows per block (4 per wave, >= 4 blocks per CU in flight); per-column dw partials live in registers.
#define LN_BWD_ROWS 16
template <typename TDY, typename TX, typename TDX, int NIT>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const TDY* __restrict__ dy, const TX* __restrict__ x,
                                                     const float* __restrict__ w, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ dres,
                                                     TDX* __restrict__ dx, bf16_t* __restrict__ dx2, float* __restrict__ dwp, int rows,
                                                     int cols) {
  __shared__ float red[4][NIT * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float dwacc[NIT][4];
#pragma unroll
  for (int it = 0; it < NIT; ++it)
#pragma unroll
    for (int j = 0; j < 4; ++j) dwacc[it][j] = 0.f;
  const int rbeg = blockIdx.x * LN_BWD_ROWS + wave * (LN_BWD_ROWS / 4);
  // (fetching dy and x of the NEXT row while this one is reduced - what pays in ffn_mid_bwd and norm_res_bwd - was measured here
  //  with -DLN_BWD_PREFETCH: 78 -> 118 VGPRs, 6 -> 4 waves per SIMD, 2.98 -> 3.60 ms per step: this kernel lives on occupancy)
#ifdef LN_BWD_PREFETCH
  constexpr bool PF = NIT <= 4;
#else
  constexpr bool PF = false;
#endif
  typename V4<TDY>::raw dn[PF ? NIT : 1];
  typename V4<TX>::raw xn[PF ? NIT : 1];
  auto fetch = [&](int row) {
#pragma unroll
    for (int it = 0; it < (PF ? NIT : 0); ++it) {
      const int c = it * 256 + lane * 4;
      if (c < cols) { dn[it] = V4<TDY>::load_raw(dy + (long)row * cols + c); xn[it] = V4<TX>::load_raw(x + (long)row * cols + c); }
    }
  };
  if (PF && rbeg < rows) fetch(rbeg);
  for (int rr = 0; rr < LN_BWD_ROWS / 4; ++rr) {
    const int row = rbeg + rr;
    if (row >= rows) break;
    const float mu = mean[row], rs = rstd[row];
    const TDY* dyr = dy + (long)row * cols;
    const TX* xr = x + (long)row * cols;
    float s1 = 0.f, s2 = 0.f;
    float gk[NIT][4], xh[NIT][4];
    typename V4<TDY>::raw dc[PF ? NIT : 1];
    typename V4<TX>::raw xc[PF ? NIT : 1];
    if constexpr (PF) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) { dc[it] = dn[it]; xc[it] = xn[it]; }
      if (rr + 1 < LN_BWD_ROWS / 4 && row + 1 < rows) fetch(row + 1);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = it * 256 + lane * 4;
      if (c < cols) {
        float d[4], v[4], g[4];
        if constexpr (PF) { V4<TDY>::unpack(dc[it], d); V4<TX>::unpack(xc[it], v); }
        else { V4<TDY>::load(dyr + c, d); V4<TX>::load(xr + c, v); }
        V4<float>::load(w + c, g);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          xh[it][j] = (v[j] - mu) * rs;
          gk[it][j] = d[j] * g[j];
          s1 += gk[it][j];
          s2 = fmaf(gk[it][j], xh[it][j], s2);
          dwacc[it][j] = fmaf(d[j], xh[it][j], dwacc[it][j]);
        }
      }
    }
    const float c1 = wave_sum(s1) / (float)cols, c2 = wave_sum(s2) / (float)cols;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = it * 256 + lane * 4;
      if (c < cols) {
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = rs * (gk[it][j] - c1 - xh[it][j] * c2);
        // (loading the residual gradient ahead of the row reductions was measured: 90 -> 116 VGPRs, 5 -> 4 waves per SIMD,
        //  2.97 -> 3.77 ms per step - this kernel lives on occupancy)
        if (dres) { float r[4]; V4<float>::load(dres + (long)row * cols + c, r);
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] += r[j]; }
        V4<TDX>::store(dx + (long)row * cols + c, o);
        if (dx2) V4<bf16_t>::store(dx2 + (long)row * cols + c, o);   // the bf16 copy the next GEMMs consume (saves a cast pass)
      }
    }
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it)
#pragma unroll
    for (int j = 0; j < 4; ++j) red[wave][it * 256 + lane * 4 + j] = dwacc[it][j];
  __syncthreads();
  for (int c = threadIdx.x; c < cols; c += 256)
    dwp[(long)blockIdx.x * cols + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

extern "C" int muse_layernorm_bwd_nblk(int32_t rows) { return (rows + LN_BWD_ROWS - 1) / LN_BWD_ROWS; }

template <typename TDY, typename TX, typename TDX>
static int ln_bwd_launch(const void* dy, const void* x, const float* w, const float* mean, const float* rstd,
                         const float* dres, void* dx, void* dx2, float* dwp, int nblk, int rows, int cols, hipStream_t s) {
#define LNB(NIT) hipLaunchKernelGGL((ln_bwd_kernel<TDY, TX, TDX, NIT>), dim3(nblk), dim3(256), 0, s, (const TDY*)dy, \
                                    (const TX*)x, w, mean, rstd, dres, (TDX*)dx, (bf16_t*)dx2, dwp, rows, cols)
  if (cols <= 256) LNB(1);
  else if (cols <= 512) LNB(2);
  else if (cols <= 768) LNB(3);    // hidden 768: three column steps, not four (12 fewer accumulator / operand registers)
  else if (cols <= 1024) LNB(4);
  else if (cols <= 2048) LNB(8);
  else if (cols <= 3072) LNB(12);
  else if (cols <= 4096) LNB(16);
  else return MUSE_ERR_UNSUPPORTED;
#undef LNB
  return (int)hipGetLastError();
}

extern "C" int muse_layernorm_bwd(const void* dy, int32_t dy_dtype, const void* x, int32_t x_dtype, const float* w,
                                  const float* mean, const float* rstd, const float* dres, void* dx, int32_t dx_dtype,
                                  void* dx_bf16, float* dw_partial, int32_t nblk, int32_t rows, int32_t cols, void* stream) {
  if (cols % 4) return MUSE_ERR_BAD_ARG;
  if (rows <= 0) return 0;
  if (nblk != muse_layernorm_bwd_nblk(rows)) return MUSE_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int key = dy_dtype * 4 + x_dtype * 2 + dx_dtype;
  switch (key) {
    case 0: return ln_bwd_launch<float, float, float>(dy, x, w, mean, rstd, dres, dx, dx_bf16, dw_partial, nblk, rows, cols, s);
    case 1: return ln_bwd_launch<float, float, bf16_t>(dy, x, w, mean, rstd, dres, dx, dx_bf16, dw_partial, nblk, rows, cols, s);
    case 2: return ln_bwd_launch<float, bf16_t, float>(dy, x, w, mean, rstd, dres, dx, dx_bf16, dw_partial, nblk, rows, cols, s);
    case 3: return ln_bwd_launch<float, bf16_t, bf16_t>(dy, x, w, mean, rstd, dres, dx, dx_bf16, dw_partial, nblk, rows, cols, s);
    case 4: return ln_bwd_launch<bf16_t, float, float>(dy, x, w, mean, rstd, dres, dx, dx_bf16, dw_partial, nblk, rows, cols, s);
    case 5: return ln_bwd_launch<bf16_t, float, bf16_t>(dy, x, w, mean, rstd, dres, dx, dx_bf16, dw_partial, nblk, rows, cols, s);
    case 6: return ln_bwd_launch<bf16_t, bf16_t, float>(dy, x, w, mean, rstd, dres, dx, dx_bf16, dw_partial, nblk, rows, cols, s);
    case 7: return ln_bwd_launch<bf16_t, bf16_t, bf16_t>(dy, x, w, mean, rstd, dres, dx, dx_bf16, dw_partial, nblk, rows, cols, s);
  }
  return MUSE_ERR_BAD_ARG;
}

// =================================================================================================================
// The two back-to-back LayerNorms of a NormFormer layer (muse/modeling_transformer.py:882-884, :785-789) as ONE pass per direction:
//   forward : x1 = x + LN(ao) * w_post ;  ln2 = LN(x1) * w_pre            (ao bf16, x / x1 f32, ln2 bf16)
//   backward: dx1 = LN_pre'(dln2) + dres ;  dao = LN_post'(dx1)            (dx1 f32 stays the residual gradient, dao bf16)
// Same arithmetic in the same order as two muse_layernorm_fwd / _bwd calls (bit-identical, tested); what goes away is the write of
// x1 / dx1 followed by its immediate re-read by the second kernel (50 MB each way per layer at config B) and one launch each.
// One wave per row with the row in registers (cols <= 256 * NIT).
// =================================================================================================================
template <int NIT>
__global__ __launch_bounds__(256) void ln_pair_fwd_kernel(const bf16_t* __restrict__ ao, const float* __restrict__ x,
                                                          const float* __restrict__ w_post, const float* __restrict__ w_pre,
                                                          float* __restrict__ x1, bf16_t* __restrict__ ln2, float* __restrict__ mean_p,
                                                          float* __restrict__ rstd_p, float* __restrict__ mean_2, float* __restrict__ rstd_2,
                                                          int rows, int cols, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float v[NIT][4];
  float s = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = it * 256 + lane * 4;
    if (c < cols) { V4<bf16_t>::load(ao + (long)row * cols + c, v[it]); s += (v[it][0] + v[it][1]) + (v[it][2] + v[it][3]); }
  }
  float mean = wave_sum(s) / (float)cols;
  float q = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = it * 256 + lane * 4;
    if (c < cols) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float d = v[it][j] - mean; q = fmaf(d, d, q); }
    }
  }
  float rstd = 1.0f / sqrtf(wave_sum(q) / (float)cols + eps);
  if (lane == 0) { mean_p[row] = mean; rstd_p[row] = rstd; }
  s = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = it * 256 + lane * 4;
    if (c < cols) {
      float g[4], r[4];
      V4<float>::load(w_post + c, g); V4<float>::load(x + (long)row * cols + c, r);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[it][j] = __fadd_rn((v[it][j] - mean) * rstd * g[j], r[j]);   // LN output rounded, THEN the residual (no fma): what ln_fwd_kernel and torch do
      V4<float>::store(x1 + (long)row * cols + c, v[it]);
      s += (v[it][0] + v[it][1]) + (v[it][2] + v[it][3]);
    }
  }
  mean = wave_sum(s) / (float)cols;
  q = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = it * 256 + lane * 4;
    if (c < cols) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float d = v[it][j] - mean; q = fmaf(d, d, q); }
    }
  }
  rstd = 1.0f / sqrtf(wave_sum(q) / (float)cols + eps);
  if (lane == 0) { mean_2[row] = mean; rstd_2[row] = rstd; }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = it * 256 + lane * 4;
    if (c < cols) {
      float g[4], o[4];
      V4<float>::load(w_pre + c, g);
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (v[it][j] - mean) * rstd * g[j];
      V4<bf16_t>::store(ln2 + (long)row * cols + c, o);
    }
  }
}

template <int NIT>
__global__ __launch_bounds__(256, 4) void ln_pair_bwd_kernel(const bf16_t* __restrict__ dln2, const float* __restrict__ x1,
                                                          const float* __restrict__ w_pre, const float* __restrict__ mean_2,
                                                          const float* __restrict__ rstd_2, const float* __restrict__ dres,
                                                          const bf16_t* __restrict__ ao, const float* __restrict__ w_post,
                                                          const float* __restrict__ mean_p, const float* __restrict__ rstd_p,
                                                          float* __restrict__ dx1, bf16_t* __restrict__ dao, float* __restrict__ dwp_pre,
                                                          float* __restrict__ dwp_post, int rows, int cols) {
  __shared__ float red[4][NIT * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float dw_pre[NIT][4], dw_post[NIT][4];
#pragma unroll
  for (int it = 0; it < NIT; ++it)
#pragma unroll
    for (int j = 0; j < 4; ++j) { dw_pre[it][j] = 0.f; dw_post[it][j] = 0.f; }
  const int rbeg = blockIdx.x * LN_BWD_ROWS + wave * (LN_BWD_ROWS / 4);
#pragma unroll 1
  for (int rr = 0; rr < LN_BWD_ROWS / 4; ++rr) {
    const int row = rbeg + rr;
    if (row >= rows) break;
    float o[NIT][4];
    {   // ---- dx1 = LN_pre'(dln2) + dres  (ln_bwd_kernel<bf16, float, float> with dres)
      const float mu = mean_2[row], rs = rstd_2[row];
      float s1 = 0.f, s2 = 0.f;
      float gk[NIT][4], xh[NIT][4];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int c = it * 256 + lane * 4;
        if (c < cols) {
          float d[4], v[4], g[4];
          V4<bf16_t>::load(dln2 + (long)row * cols + c, d); V4<float>::load(x1 + (long)row * cols + c, v); V4<float>::load(w_pre + c, g);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            xh[it][j] = (v[j] - mu) * rs;
            gk[it][j] = d[j] * g[j];
            s1 += gk[it][j];
            s2 = fmaf(gk[it][j], xh[it][j], s2);
            dw_pre[it][j] = fmaf(d[j], xh[it][j], dw_pre[it][j]);
          }
        }
      }
      const float c1 = wave_sum(s1) / (float)cols, c2 = wave_sum(s2) / (float)cols;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int c = it * 256 + lane * 4;
        if (c < cols) {
#pragma unroll
          for (int j = 0; j < 4; ++j) o[it][j] = rs * (gk[it][j] - c1 - xh[it][j] * c2);
          if (dres) { float r[4]; V4<float>::load(dres + (long)row * cols + c, r);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[it][j] += r[j]; }
          V4<float>::store(dx1 + (long)row * cols + c, o[it]);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);   // keep the second phase's loads out of the first one (136 -> fewer VGPRs: this kernel lives on occupancy)
    {   // ---- dao = LN_post'(dx1)  (ln_bwd_kernel<float, bf16, bf16>, dy = the dx1 just computed)
      const float mu = mean_p[row], rs = rstd_p[row];
      float s1 = 0.f, s2 = 0.f;
      float gk[NIT][4], xh[NIT][4];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int c = it * 256 + lane * 4;
        if (c < cols) {
          float v[4], g[4];
          V4<bf16_t>::load(ao + (long)row * cols + c, v); V4<float>::load(w_post + c, g);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float d = o[it][j];
            xh[it][j] = (v[j] - mu) * rs;
            gk[it][j] = d * g[j];
            s1 += gk[it][j];
            s2 = fmaf(gk[it][j], xh[it][j], s2);
            dw_post[it][j] = fmaf(d, xh[it][j], dw_post[it][j]);
          }
        }
      }
      const float c1 = wave_sum(s1) / (float)cols, c2 = wave_sum(s2) / (float)cols;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int c = it * 256 + lane * 4;
        if (c < cols) {
          float r[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) r[j] = rs * (gk[it][j] - c1 - xh[it][j] * c2);
          V4<bf16_t>::store(dao + (long)row * cols + c, r);
        }
      }
    }
  }
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    if (pass) __syncthreads();
#pragma unroll
    for (int it = 0; it < NIT; ++it)
#pragma unroll
      for (int j = 0; j < 4; ++j) red[wave][it * 256 + lane * 4 + j] = pass ? dw_post[it][j] : dw_pre[it][j];
    __syncthreads();
    float* dwp = pass ? dwp_post : dwp_pre;
    for (int c = threadIdx.x; c < cols; c += 256)
      dwp[(long)blockIdx.x * cols + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
  }
}

extern "C" int muse_layernorm_pair_fwd(const void* ao, const float* x, const float* w_post, const float* w_pre, float* x1, void* ln2,
                                       float* mean_post, float* rstd_post, float* mean_pre, float* rstd_pre, int32_t rows,
                                       int32_t cols, float eps, void* stream) {
  if (cols % 4 || cols > 1024) return MUSE_ERR_UNSUPPORTED;
  if (rows <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((rows + 3) / 4);
#define LPF(N) hipLaunchKernelGGL(ln_pair_fwd_kernel<N>, grid, dim3(256), 0, s, (const bf16_t*)ao, x, w_post, w_pre, x1, (bf16_t*)ln2, \
                                  mean_post, rstd_post, mean_pre, rstd_pre, rows, cols, eps)
  if (cols <= 256) LPF(1); else if (cols <= 512) LPF(2); else if (cols <= 768) LPF(3); else LPF(4);
#undef LPF
  return (int)hipGetLastError();
}

extern "C" int muse_layernorm_pair_bwd(const void* dln2, const float* x1, const float* w_pre, const float* mean_pre,
                                       const float* rstd_pre, const float* dres, const void* ao, const float* w_post,
                                       const float* mean_post, const float* rstd_post, float* dx1, void* dao, float* dw_partial_pre,
                                       float* dw_partial_post, int32_t nblk, int32_t rows, int32_t cols, void* stream) {
  if (cols % 4 || cols > 1024) return MUSE_ERR_UNSUPPORTED;
  if (rows <= 0) return 0;
  if (nblk != (rows + LN_BWD_ROWS - 1) / LN_BWD_ROWS) return MUSE_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
#define LPB(N) hipLaunchKernelGGL(ln_pair_bwd_kernel<N>, dim3(nblk), dim3(256), 0, s, (const bf16_t*)dln2, x1, w_pre, mean_pre, rstd_pre, \
                                  dres, (const bf16_t*)ao, w_post, mean_post, rstd_post, dx1, (bf16_t*)dao, dw_partial_pre, dw_partial_post, rows, cols)
  if (cols <= 256) LPB(1); else if (cols <= 512) LPB(2); else if (cols <= 768) LPB(3); else LPB(4);
#undef LPB
  return (int)hipGetLastError();
}

// =================================================================================================================
// Fused middle of the NormFormer GLU MLP (muse/modeling_transformer.py:789-797):
//   forward : h = gelu(a) * b ; hm = LayerNorm(h) * w          (ab = [rows, 2I], a first)   - one pass over ab
//   backward: dh = LN'(dhm) ; dab = (dh * b * gelu'(a), dh * gelu(a)) ; dw partials          - dh never touches HBM
// One 256-thread block per row at a time (NV chunks of 4 columns per thread), block reductions through LDS.
// =================================================================================================================
// (Measured and rejected, round 2: software-pipelining the rows of a block - next row's loads issued before this row's reductions,
// LDS-only barriers - raised the kernels to 136-145 VGPRs and made them SLOWER on MI355X: ffn_mid_bwd 4.39 -> 4.91 ms,
// ffn_mid_fwd 2.19 -> 2.40 ms per step.  Occupancy, not loads in flight per block, is what these kernels live on.)
__device__ __forceinline__ float block_sum256(float v, float* red, int slot) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[slot * 4 + (threadIdx.x >> 6)] = v;
  __syncthreads();
  return (red[slot * 4] + red[slot * 4 + 1]) + (red[slot * 4 + 2] + red[slot * 4 + 3]);
}

#define FFN_ROWS 8
template <typename T, int NV>
__global__ __launch_bounds__(256) void ffn_mid_fwd_kernel(const T* __restrict__ ab, const float* __restrict__ w,
                                                          T* __restrict__ h, T* __restrict__ hm, float* __restrict__ mean_o,
                                                          float* __restrict__ rstd_o, int rows, int inter, float eps) {
  __shared__ float red[16];
  float wv[NV][4];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = (k * 256 + threadIdx.x) * 4;
    if (c < inter) V4<float>::load(w + c, wv[k]);
  }
  const int r0 = blockIdx.x * FFN_ROWS;
  for (int rr = 0; rr < FFN_ROWS; ++rr) {
    const int row = r0 + rr;
    if (row >= rows) break;
    const T* abr = ab + (long)row * 2 * inter;
    float hv[NV][4];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = (k * 256 + threadIdx.x) * 4;
      if (c < inter) {
        float a[4], b[4], o[4];
        V4<T>::load(abr + c, a); V4<T>::load(abr + inter + c, b);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = gelu_erf_t<sizeof(T) == 2>(a[j]) * b[j];
        if (h) V4<T>::store(h + (long)row * inter + c, o);
        if (sizeof(T) == 2) {  // LayerNorm sees the stored (bf16-rounded) h, exactly like the unfused path
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = bf16_to_f32(f32_to_bf16(o[j]));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { hv[k][j] = o[j]; s += o[j]; }
      }
    }
    const float mean = block_sum256(s, red, 0) / (float)inter;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = (k * 256 + threadIdx.x) * 4;
      if (c < inter) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = hv[k][j] - mean; q = fmaf(d, d, q); }
      }
    }
    const float rstd = 1.0f / sqrtf(block_sum256(q, red, 1) / (float)inter + eps);
    if (threadIdx.x == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = (k * 256 + threadIdx.x) * 4;
      if (c < inter) {
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (hv[k][j] - mean) * rstd * wv[k][j];
        V4<T>::store(hm + (long)row * inter + c, o);
      }
    }
    __syncthreads();  // red[] is reused by the next row
  }
}

// RECOMP: h is not read but recomputed from ab exactly as the forward produced it (gelu_erf(a) * b, rounded to the storage type
// - the value the forward's LayerNorm normalised): one tensor less to write in the forward and to read here (-17 % of this kernel's
// bytes), the erf it needs is the one the GLU backward computes anyway.  erf(a / sqrt 2) is kept across the row reductions.
template <typename T, int NV, bool RECOMP>
__global__ __launch_bounds__(256) void ffn_mid_bwd_kernel(const T* __restrict__ dhm, const T* __restrict__ h,
                                                          const T* __restrict__ ab, const float* __restrict__ w,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          T* __restrict__ dab, float* __restrict__ dwp, int rows, int inter) {
  __shared__ float red[16];
  float wv[NV][4], dwacc[NV][4];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = (k * 256 + threadIdx.x) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) dwacc[k][j] = 0.f;
    if (c < inter) V4<float>::load(w + c, wv[k]);
  }
  const int r0 = blockIdx.x * FFN_ROWS;
  // every operand of a row is fetched while the PREVIOUS row is being reduced (the block's three barriers per row otherwise expose a
  // full memory round trip per row)
  typename V4<T>::raw ra_n[NV], rb_n[NV], rd_n[NV];
  auto fetch = [&](int row) {
    const T* abr = ab + (long)row * 2 * inter;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = (k * 256 + threadIdx.x) * 4;
      if (c < inter) {
        ra_n[k] = V4<T>::load_raw(abr + c); rb_n[k] = V4<T>::load_raw(abr + inter + c);
        rd_n[k] = V4<T>::load_raw(dhm + (long)row * inter + c);
      }
    }
  };
  if (r0 < rows) fetch(r0);
  for (int rr = 0; rr < FFN_ROWS; ++rr) {
    const int row = r0 + rr;
    if (row >= rows) break;
    const float mu = mean[row], rs = rstd[row];
    float gk[NV][4], xh[RECOMP ? 1 : NV][4], ev[RECOMP ? NV : 1][4];
    float s1 = 0.f, s2 = 0.f;
    typename V4<T>::raw ra[NV], rb[NV], rd[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) { ra[k] = ra_n[k]; rb[k] = rb_n[k]; rd[k] = rd_n[k]; }
    if (rr + 1 < FFN_ROWS && row + 1 < rows) fetch(row + 1);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = (k * 256 + threadIdx.x) * 4;
      if (c < inter) {
        float d[4], x[4];
        V4<T>::unpack(rd[k], d);
        if constexpr (RECOMP) {
          float a[4], b[4];
          V4<T>::unpack(ra[k], a); V4<T>::unpack(rb[k], b);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            ev[k][j] = erf_rsqrt2<sizeof(T) == 2>(a[j]);
            x[j] = 0.5f * a[j] * (1.0f + ev[k][j]) * b[j];           // == gelu_erf(a) * b, bit for bit
            if (sizeof(T) == 2) x[j] = bf16_to_f32(f32_to_bf16(x[j]));
          }
        } else {
          V4<T>::load(h + (long)row * inter + c, x);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float xhj = (x[j] - mu) * rs;
          if constexpr (!RECOMP) xh[k][j] = xhj;
          gk[k][j] = d[j] * wv[k][j];
          s1 += gk[k][j];
          s2 = fmaf(gk[k][j], xhj, s2);
          dwacc[k][j] = fmaf(d[j], xhj, dwacc[k][j]);
        }
      }
    }
    const float c1 = block_sum256(s1, red, 0) / (float)inter;
    const float c2 = block_sum256(s2, red, 1) / (float)inter;
    T* dr = dab + (long)row * 2 * inter;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = (k * 256 + threadIdx.x) * 4;
      if (c < inter) {
        float a[4], b[4], da[4], db[4];
        V4<T>::unpack(ra[k], a); V4<T>::unpack(rb[k], b);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (RECOMP) {
            const float e = ev[k][j], g = 0.5f * a[j] * (1.0f + e);
            float hq = g * b[j];
            if (sizeof(T) == 2) hq = bf16_to_f32(f32_to_bf16(hq));
            const float dh = rs * (gk[k][j] - c1 - ((hq - mu) * rs) * c2);
            const float cdf = 0.5f * (1.0f + e), pdf = 0.39894228040143267794f * __expf(-0.5f * a[j] * a[j]);
            da[j] = dh * b[j] * (cdf + a[j] * pdf);                  // == gelu_erf_grad(a)
            db[j] = dh * g;
          } else {
            const float dh = rs * (gk[k][j] - c1 - xh[k][j] * c2);
            da[j] = dh * b[j] * gelu_erf_grad_t<sizeof(T) == 2>(a[j]);
            db[j] = dh * gelu_erf_t<sizeof(T) == 2>(a[j]);
          }
        }
        V4<T>::store(dr + c, da); V4<T>::store(dr + inter + c, db);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = (k * 256 + threadIdx.x) * 4;
    if (c < inter) V4<float>::store(dwp + (long)blockIdx.x * inter + c, dwacc[k]);
  }
}

// ---- bf16, inter % 1024 == 0: TWO waves per row, 16-byte accesses, every operand of the row in flight at once ------------------
// The 256-threads-per-row kernels above move 8 bytes per lane per access in two dependent phases per row (3.3 TB/s backward,
// 4.4 TB/s forward on MI355X).  Here a row belongs to a pair of waves (128 lanes x 8 elements = 1024 columns per step, NK steps),
// the two pairs of a block work on different rows, all 4 NK (backward) / 2 NK (forward) loads of a row are issued before the
// first use, and the only block barriers are the ones that carry the two-wave row reductions.  Partial-sum layout unchanged:
// partial row i = rows [8 i, 8 i + 8) (a pair walks 8 consecutive rows).
__device__ __forceinline__ void unpack8(const u32x4& t, float (&v)[8]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(t[j] << 16); v[2 * j + 1] = __uint_as_float(t[j] & 0xffff0000u); }
}
__device__ __forceinline__ u32x4 pack8(const float (&v)[8]) {
  return u32x4{pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7])};
}
// sum over the 128 lanes of a wave pair; `slot` = one of 4 exchange cells of the pair (a cell is rewritten only two barriers later)
__device__ __forceinline__ float pair_sum(float v, float (*red)[2], int slot, int pair, int wip) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[pair * 4 + slot][wip] = v;
  __syncthreads();
  return red[pair * 4 + slot][0] + red[pair * 4 + slot][1];
}

template <int NK>
__global__ __launch_bounds__(256) void ffn_mid_fwd2_kernel(const bf16_t* __restrict__ ab, const float* __restrict__ w,
                                                           bf16_t* __restrict__ h, bf16_t* __restrict__ hm, float* __restrict__ mean_o,
                                                           float* __restrict__ rstd_o, int rows, float eps) {
  constexpr int inter = NK * 1024;
  __shared__ float red[8][2];
  const int pair = threadIdx.x >> 7, wip = (threadIdx.x >> 6) & 1, t = threadIdx.x & 127;
  float wv[NK][8];
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const f32x4 w0 = *(const f32x4*)(w + k * 1024 + t * 8), w1 = *(const f32x4*)(w + k * 1024 + t * 8 + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { wv[k][j] = w0[j]; wv[k][4 + j] = w1[j]; }
  }
  const int r0 = (blockIdx.x * 2 + pair) * FFN_ROWS;
#pragma unroll 1
  for (int rr = 0; rr < FFN_ROWS; ++rr) {
    const int row = r0 + rr;
    const bool live = row < rows;     // (both pairs keep taking the barriers)
    const int rowc = live ? row : rows - 1;
    const bf16_t* abr = ab + (long)rowc * 2 * inter;
    u32x4 ra[NK], rb[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) { ra[k] = *(const u32x4*)(abr + k * 1024 + t * 8); rb[k] = *(const u32x4*)(abr + inter + k * 1024 + t * 8); }
    float hv[NK][8];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      float a[8], b[8];
      unpack8(ra[k], a); unpack8(rb[k], b);
#pragma unroll
      for (int j = 0; j < 8; ++j) hv[k][j] = gelu_erf_t<true>(a[j]) * b[j];
      const u32x4 o = pack8(hv[k]);
      if (live && h) *(u32x4*)(h + (long)row * inter + k * 1024 + t * 8) = o;   // (h == nullptr: the backward recomputes it)
      unpack8(o, hv[k]);   // LayerNorm sees the stored (bf16-rounded) h, exactly like the unfused path
#pragma unroll
      for (int j = 0; j < 8; ++j) s += hv[k][j];
    }
    const float mean = pair_sum(s, red, (rr & 1) * 2, pair, wip) / (float)inter;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NK; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = hv[k][j] - mean; q = fmaf(d, d, q); }
    const float rstd = 1.0f / sqrtf(pair_sum(q, red, (rr & 1) * 2 + 1, pair, wip) / (float)inter + eps);
    if (live) {
      if (t == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (hv[k][j] - mean) * rstd * wv[k][j];
        *(u32x4*)(hm + (long)row * inter + k * 1024 + t * 8) = pack8(o);
      }
    }
  }
}

template <int NK>
__global__ __launch_bounds__(256) void ffn_mid_bwd2_kernel(const bf16_t* __restrict__ dhm, const bf16_t* __restrict__ h,
                                                           const bf16_t* __restrict__ ab, const float* __restrict__ w,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           bf16_t* __restrict__ dab, float* __restrict__ dwp, int rows) {
  constexpr int inter = NK * 1024;
  __shared__ float red[8][2];
  const int pair = threadIdx.x >> 7, wip = (threadIdx.x >> 6) & 1, t = threadIdx.x & 127;
  float wv[NK][8], dwacc[NK][8];
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const f32x4 w0 = *(const f32x4*)(w + k * 1024 + t * 8), w1 = *(const f32x4*)(w + k * 1024 + t * 8 + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { wv[k][j] = w0[j]; wv[k][4 + j] = w1[j]; dwacc[k][j] = 0.f; dwacc[k][4 + j] = 0.f; }
  }
  const int prow = blockIdx.x * 2 + pair;     // partial row of this pair
  const int r0 = prow * FFN_ROWS;
#pragma unroll 1
  for (int rr = 0; rr < FFN_ROWS; ++rr) {
    const int row = r0 + rr;
    const bool live = row < rows;
    const int rowc = live ? row : rows - 1;
    const float mu = mean[rowc], rs = rstd[rowc];
    const bf16_t* abr = ab + (long)rowc * 2 * inter;
    u32x4 rd[NK], rx[NK], ra[NK], rb[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const long c = (long)rowc * inter + k * 1024 + t * 8;
      rd[k] = *(const u32x4*)(dhm + c); rx[k] = *(const u32x4*)(h + c);
      ra[k] = *(const u32x4*)(abr + k * 1024 + t * 8); rb[k] = *(const u32x4*)(abr + inter + k * 1024 + t * 8);
    }
    float s1 = 0.f, s2 = 0.f;
    float xh[NK][8], gk[NK][8];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      float d[8], x[8];
      unpack8(rd[k], d); unpack8(rx[k], x);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        xh[k][j] = (x[j] - mu) * rs;
        gk[k][j] = d[j] * wv[k][j];
        s1 += gk[k][j];
        s2 = fmaf(gk[k][j], xh[k][j], s2);
        if (live) dwacc[k][j] = fmaf(d[j], xh[k][j], dwacc[k][j]);
      }
    }
    const float c1 = pair_sum(s1, red, (rr & 1) * 2, pair, wip) / (float)inter;
    const float c2 = pair_sum(s2, red, (rr & 1) * 2 + 1, pair, wip) / (float)inter;
    if (live) {
      bf16_t* dr = dab + (long)row * 2 * inter;
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        float a[8], b[8], da[8], db[8];
        unpack8(ra[k], a); unpack8(rb[k], b);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float dh = rs * (gk[k][j] - c1 - xh[k][j] * c2);
          da[j] = dh * b[j] * gelu_erf_grad_t<true>(a[j]);
          db[j] = dh * gelu_erf_t<true>(a[j]);
        }
        *(u32x4*)(dr + k * 1024 + t * 8) = pack8(da);
        *(u32x4*)(dr + inter + k * 1024 + t * 8) = pack8(db);
      }
    }
  }
  if (r0 < rows) {
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      float* o = dwp + (long)prow * inter + k * 1024 + t * 8;
      *(f32x4*)o = f32x4{dwacc[k][0], dwacc[k][1], dwacc[k][2], dwacc[k][3]};
      *(f32x4*)(o + 4) = f32x4{dwacc[k][4], dwacc[k][5], dwacc[k][6], dwacc[k][7]};
    }
  }
}

// MUSE_FFN_MID_WIDE: bit 0 = forward, bit 1 = backward.  Measured on MI355X (config B, 24 layers): forward 2.24 -> 1.98 ms per
// step; backward 4.45 -> 4.42 ms (199 VGPRs: two waves per SIMD eat what the wider accesses give) - default: forward only.
static bool ffn_mid_wide_ok(int32_t dtype, int32_t inter, int bit) {
  static const int on = []() { const char* e = getenv("MUSE_FFN_MID_WIDE"); return e ? atoi(e) : 1; }();
  return ((on >> bit) & 1) && dtype == MUSE_BF16 && inter % 1024 == 0 && inter <= 4096;
}

extern "C" int muse_ffn_mid_rows_per_block(void) { return FFN_ROWS; }

extern "C" int muse_ffn_mid_fwd(const void* ab, const float* w, void* h, void* hm, float* mean, float* rstd, int32_t dtype,
                                int32_t rows, int32_t inter, float eps, void* stream) {
  if (inter % 4 || inter > 4096) return MUSE_ERR_UNSUPPORTED;
  if (rows <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((rows + FFN_ROWS - 1) / FFN_ROWS);
  const int nv = (inter + 1023) / 1024;
  if (ffn_mid_wide_ok(dtype, inter, 0) && !((((uintptr_t)ab) | ((uintptr_t)h) | ((uintptr_t)hm) | ((uintptr_t)w)) & 15)) {   // (a null h is "aligned")
    const dim3 g2((grid.x + 1) / 2);
#define FF2(NK) hipLaunchKernelGGL((ffn_mid_fwd2_kernel<NK>), g2, dim3(256), 0, s, (const bf16_t*)ab, w, (bf16_t*)h, (bf16_t*)hm, mean, rstd, rows, eps)
    if (nv == 1) FF2(1); else if (nv == 2) FF2(2); else if (nv == 3) FF2(3); else FF2(4);
#undef FF2
    return (int)hipGetLastError();
  }
#define FF(T, NV) hipLaunchKernelGGL((ffn_mid_fwd_kernel<T, NV>), grid, dim3(256), 0, s, (const T*)ab, w, (T*)h, (T*)hm, mean, rstd, rows, inter, eps)
  if (dtype == MUSE_F32) { if (nv == 1) FF(float, 1); else if (nv == 2) FF(float, 2); else if (nv == 3) FF(float, 3); else FF(float, 4); }
  else { if (nv == 1) FF(bf16_t, 1); else if (nv == 2) FF(bf16_t, 2); else if (nv == 3) FF(bf16_t, 3); else FF(bf16_t, 4); }
#undef FF
  return (int)hipGetLastError();
}

extern "C" int muse_ffn_mid_bwd(const void* dhm, const void* h, const void* ab, const float* w, const float* mean,
                                const float* rstd, void* dab, float* dw_partial, int32_t dtype, int32_t rows, int32_t inter,
                                void* stream) {
  if (inter % 4 || inter > 4096) return MUSE_ERR_UNSUPPORTED;
  if (rows <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((rows + FFN_ROWS - 1) / FFN_ROWS);
  const int nv = (inter + 1023) / 1024;
  if (h && ffn_mid_wide_ok(dtype, inter, 1) && !((((uintptr_t)dhm) | ((uintptr_t)h) | ((uintptr_t)ab) | ((uintptr_t)dab) | ((uintptr_t)w) | ((uintptr_t)dw_partial)) & 15)) {
    const dim3 g2((grid.x + 1) / 2);
#define FB2(NK) hipLaunchKernelGGL((ffn_mid_bwd2_kernel<NK>), g2, dim3(256), 0, s, (const bf16_t*)dhm, (const bf16_t*)h, (const bf16_t*)ab, w, mean, rstd, (bf16_t*)dab, dw_partial, rows)
    if (nv == 1) FB2(1); else if (nv == 2) FB2(2); else if (nv == 3) FB2(3); else FB2(4);
#undef FB2
    return (int)hipGetLastError();
  }
#define FB(T, NV) do { if (h) hipLaunchKernelGGL((ffn_mid_bwd_kernel<T, NV, false>), grid, dim3(256), 0, s, (const T*)dhm, (const T*)h, (const T*)ab, w, mean, rstd, (T*)dab, dw_partial, rows, inter); \
                       else hipLaunchKernelGGL((ffn_mid_bwd_kernel<T, NV, true>), grid, dim3(256), 0, s, (const T*)dhm, (const T*)nullptr, (const T*)ab, w, mean, rstd, (T*)dab, dw_partial, rows, inter); } while (0)
  if (dtype == MUSE_F32) { if (nv == 1) FB(float, 1); else if (nv == 2) FB(float, 2); else if (nv == 3) FB(float, 3); else FB(float, 4); }
  else { if (nv == 1) FB(bf16_t, 1); else if (nv == 2) FB(bf16_t, 2); else if (nv == 3) FB(bf16_t, 3); else FB(bf16_t, 4); }
#undef FB
  return (int)hipGetLastError();
}

// out[c] (+)= sum_r in[r,c]; 16 columns x 64 row-groups per block (cols/16 blocks: the partial-sum matrices are short and
// wide, 64-column blocks left a dozen CUs doing a latency-bound row walk), fixed summation order (deterministic)
__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols, int acc) {
  __shared__ float red[64][17];
  const int c16 = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + c16;
  float s0 = 0.f, s1 = 0.f;
  if (c < cols) {
    int r = rg;
    for (; r + 64 < rows; r += 128) { s0 += in[(long)r * cols + c]; s1 += in[(long)(r + 64) * cols + c]; }
    if (r < rows) s0 += in[(long)r * cols + c];
  }
  red[rg][c16] = s0 + s1;
  __syncthreads();
  if (rg == 0 && c < cols) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 64; ++k) s += red[k][c16];
    out[c] = acc ? out[c] + s : s;
  }
}
extern "C" int muse_colsum(const float* in, float* out, int32_t rows, int32_t cols, int32_t accumulate, void* stream) {
  if (cols <= 0) return 0;
  hipLaunchKernelGGL(colsum_kernel, dim3((cols + 15) / 16), dim3(1024), 0, (hipStream_t)stream, in, out, rows, cols, accumulate);
  return (int)hipGetLastError();
}

// =================================================================================================================
// biases of use_bias models (muse/modeling_transformer.py:130, :170-176): the LayerNorm bias added in place, and the first stage of
// d(bias) = sum_r dy[r, :] (second stage: muse_colsum over the per-chunk partial rows; fixed order, no float atomics)
// =================================================================================================================
__global__ __launch_bounds__(256) void add_rowvec_kernel(float* __restrict__ x, const float* __restrict__ b, long n, int cols) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) x[i] += b[i % cols];
}
extern "C" int muse_add_rowvec(float* x, const float* b, int64_t rows, int32_t cols, void* stream) {
  if (rows <= 0 || cols <= 0) return 0;
  const long n = (long)rows * cols;
  hipLaunchKernelGGL(add_rowvec_kernel, dim3((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, b, n, cols);
  return (int)hipGetLastError();
}

constexpr int kBiasGradRows = 128;
template <typename T>
__global__ __launch_bounds__(256) void bias_grad_partial_kernel(const T* __restrict__ dy, float* __restrict__ part, long rows, int cols, long ld) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  const long r0 = (long)blockIdx.y * kBiasGradRows;
  const long r1 = r0 + kBiasGradRows < rows ? r0 + kBiasGradRows : rows;
  float s = 0.f;
  for (long r = r0; r < r1; ++r) s += Elem<T>::load(dy + r * ld + c);      // (a wave reads 64 consecutive columns of one row)
  part[(long)blockIdx.y * cols + c] = s;
}
extern "C" int muse_bias_grad_rows_per_block(void) { return kBiasGradRows; }
extern "C" int muse_bias_grad_partial(const void* dy, int32_t dtype, float* partial, int64_t rows, int32_t cols, int64_t ld, void* stream) {
  if (rows <= 0 || cols <= 0) return 0;
  if (dtype != MUSE_F32 && dtype != MUSE_BF16) return MUSE_ERR_BAD_ARG;
  const dim3 grid((cols + 255) / 256, (unsigned)((rows + kBiasGradRows - 1) / kBiasGradRows));
  if (dtype == MUSE_F32) hipLaunchKernelGGL((bias_grad_partial_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)dy, partial, (long)rows, cols, (long)ld);
  else hipLaunchKernelGGL((bias_grad_partial_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, partial, (long)rows, cols, (long)ld);
  return (int)hipGetLastError();
}

// =================================================================================================================
// softmax over rows of [rows, ld] (cols valid); one wave per row, pad columns [cols, ld) written as 0
// =================================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long rows, int cols, long ld) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + row * ld;
  T* yr = y + row * ld;
  float m = -INFINITY;
  for (int c = lane; c < cols; c += 64) m = fmaxf(m, Elem<T>::load(xr + c));
  m = wave_max(m);
  float s = 0.f;
  for (int c = lane; c < cols; c += 64) s += expf(Elem<T>::load(xr + c) - m);
  s = wave_sum(s);
  const float inv = 1.0f / s;
  for (int c = lane; c < (int)ld; c += 64) {
    const float v = c < cols ? expf(Elem<T>::load(xr + c) - m) * inv : 0.f;
    Elem<T>::store(yr + c, v);
  }
}
extern "C" int muse_softmax_fwd(const void* x, void* y, int32_t dtype, int64_t rows, int32_t cols, int64_t ld, void* stream) {
  if (rows <= 0) return 0;
  dim3 grid((unsigned)((rows + 3) / 4));
  if (dtype == MUSE_F32) hipLaunchKernelGGL(softmax_fwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, (long)rows, cols, (long)ld);
  else hipLaunchKernelGGL(softmax_fwd_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, (long)rows, cols, (long)ld);
  return (int)hipGetLastError();
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const T* __restrict__ p, const T* __restrict__ dp, T* __restrict__ ds,
                                                          long rows, int cols, long ld) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* pr = p + row * ld;
  const T* dr = dp + row * ld;
  T* sr = ds + row * ld;
  float s = 0.f;
  for (int c = lane; c < cols; c += 64) s = fmaf(Elem<T>::load(pr + c), Elem<T>::load(dr + c), s);
  s = wave_sum(s);
  for (int c = lane; c < (int)ld; c += 64) {
    const float v = c < cols ? Elem<T>::load(pr + c) * (Elem<T>::load(dr + c) - s) : 0.f;
    Elem<T>::store(sr + c, v);
  }
}
extern "C" int muse_softmax_bwd(const void* p, const void* dp, void* ds, int32_t dtype, int64_t rows, int32_t cols,
                                int64_t ld, void* stream) {
  if (rows <= 0) return 0;
  dim3 grid((unsigned)((rows + 3) / 4));
  if (dtype == MUSE_F32) hipLaunchKernelGGL(softmax_bwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)p, (const float*)dp, (float*)ds, (long)rows, cols, (long)ld);
  else hipLaunchKernelGGL(softmax_bwd_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)p, (const bf16_t*)dp, (bf16_t*)ds, (long)rows, cols, (long)ld);
  return (int)hipGetLastError();
}

// =================================================================================================================
// GLU / GELU element kernels (4 elements per thread)
// =================================================================================================================
template <typename T>
__global__ void glu_fwd_kernel(const T* __restrict__ ab, T* __restrict__ h, long rows, int inter) {
  const long n4 = rows * (inter / 4);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const long r = i / (inter / 4);
    const int c = (int)(i - r * (inter / 4)) * 4;
    float a[4], b[4], o[4];
    V4<T>::load(ab + r * 2 * inter + c, a);
    V4<T>::load(ab + r * 2 * inter + inter + c, b);
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = gelu_erf_t<sizeof(T) == 2>(a[j]) * b[j];
    V4<T>::store(h + r * inter + c, o);
  }
}
template <typename T>
__global__ void glu_bwd_kernel(const T* __restrict__ ab, const T* __restrict__ dh, T* __restrict__ dab, long rows, int inter) {
  const long n4 = rows * (inter / 4);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const long r = i / (inter / 4);
    const int c = (int)(i - r * (inter / 4)) * 4;
    float a[4], b[4], d[4], da[4], db[4];
    V4<T>::load(ab + r * 2 * inter + c, a);
    V4<T>::load(ab + r * 2 * inter + inter + c, b);
    V4<T>::load(dh + r * inter + c, d);
#pragma unroll
    for (int j = 0; j < 4; ++j) { da[j] = d[j] * b[j] * gelu_erf_grad_t<sizeof(T) == 2>(a[j]); db[j] = d[j] * gelu_erf_t<sizeof(T) == 2>(a[j]); }
    V4<T>::store(dab + r * 2 * inter + c, da);
    V4<T>::store(dab + r * 2 * inter + inter + c, db);
  }
}
static inline int ew_grid(long n) { long g = (n + 255) / 256; return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g)); }
// f32 GLU of the "bf16x3" compute mode: the same expressions as glu_*_kernel<float> (the same f32 bits), and the result ALSO as the
// (hi, lo) bf16 operand planes of the product that reads it (muse_gemm_x3: h feeds the FFN's output projection, d(ab) the dX and dW
// products of its input projection) - the separate split pass (read 4 + write 4 bytes per element) becomes 4 written bytes here.
// planes: [2][rows][cols] bf16, hi plane first; split4 is the split muse_split_f32_to_bf16x2 applies, so the planes are its bits.
// (ImgFormat: the "f16" mode's half image instead - common.h store_image4)
static int g_img_half = 0;
static float g_img_grad_scale = 1.f;
static int* g_img_stats = nullptr;
ImgFormat img_format(bool gradient) {
  return ImgFormat{g_img_half ? -1L : 1L, (g_img_half && gradient) ? g_img_grad_scale : 1.f, g_img_half ? g_img_stats : nullptr};
}
extern "C" int muse_operand_images(int32_t half, float grad_scale, int32_t* stats) {
  int e = 0;
  if ((half != 0 && half != 1) || !(grad_scale > 0.f) || frexpf(grad_scale, &e) != 0.5f) return MUSE_ERR_BAD_ARG;   // a power of two
  g_img_half = half;
  g_img_grad_scale = grad_scale;
  g_img_stats = half ? (int*)stats : nullptr;
  return 0;
}
__device__ __forceinline__ void store_planes4(bf16_t* hi, long plane, const ImgFormat& f, long idx, const float (&o)[4]) {
  store_image4(hi + idx, plane, f.scale, f.stats, o[0], o[1], o[2], o[3]);
}
__global__ void glu_fwd_x3_kernel(const float* __restrict__ ab, float* __restrict__ h, bf16_t* __restrict__ planes, long rows, int inter, ImgFormat f) {
  const long n4 = rows * (inter / 4), plane = f.lo_sign * rows * inter;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const long r = i / (inter / 4);
    const int c = (int)(i - r * (inter / 4)) * 4;
    float a[4], b[4], o[4];
    V4<float>::load(ab + r * 2 * inter + c, a);
    V4<float>::load(ab + r * 2 * inter + inter + c, b);
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = gelu_erf_t<false>(a[j]) * b[j];
    if (h) V4<float>::store(h + r * inter + c, o);
    store_planes4(planes, plane, f, r * inter + c, o);
  }
}
__global__ void glu_bwd_x3_kernel(const float* __restrict__ ab, const float* __restrict__ dh, float* __restrict__ dab, bf16_t* __restrict__ planes,
                                  long rows, int inter, ImgFormat f) {
  const long n4 = rows * (inter / 4), plane = f.lo_sign * rows * 2 * inter;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const long r = i / (inter / 4);
    const int c = (int)(i - r * (inter / 4)) * 4;
    float a[4], b[4], d[4], da[4], db[4];
    V4<float>::load(ab + r * 2 * inter + c, a);
    V4<float>::load(ab + r * 2 * inter + inter + c, b);
    V4<float>::load(dh + r * inter + c, d);
#pragma unroll
    for (int j = 0; j < 4; ++j) { da[j] = d[j] * b[j] * gelu_erf_grad_t<false>(a[j]); db[j] = d[j] * gelu_erf_t<false>(a[j]); }
    if (dab) {
      V4<float>::store(dab + r * 2 * inter + c, da);
      V4<float>::store(dab + r * 2 * inter + inter + c, db);
    }
    store_planes4(planes, plane, f, r * 2 * inter + c, da);
    store_planes4(planes, plane, f, r * 2 * inter + inter + c, db);
  }
}
extern "C" int muse_glu_fwd_x3(const float* ab, float* h, void* planes, int64_t rows, int32_t inter, void* stream) {
  if (inter % 4) return MUSE_ERR_BAD_ARG;
  if (rows <= 0) return 0;
  if (!planes) return MUSE_ERR_BAD_ARG;
  if ((((uintptr_t)ab) | ((uintptr_t)h)) & 15 || (((uintptr_t)planes) & 7) || ((rows * inter) & 3)) return MUSE_ERR_ALIGN;
  hipLaunchKernelGGL(glu_fwd_x3_kernel, dim3(ew_grid(rows * (inter / 4))), dim3(256), 0, (hipStream_t)stream, ab, h, (bf16_t*)planes, (long)rows, inter,
                     img_format(false));
  return (int)hipGetLastError();
}
extern "C" int muse_glu_bwd_x3(const float* ab, const float* dh, float* dab, void* planes, int64_t rows, int32_t inter, void* stream) {
  if (inter % 4) return MUSE_ERR_BAD_ARG;
  if (rows <= 0) return 0;
  if (!planes) return MUSE_ERR_BAD_ARG;
  if ((((uintptr_t)ab) | ((uintptr_t)dh) | ((uintptr_t)dab)) & 15 || (((uintptr_t)planes) & 7)) return MUSE_ERR_ALIGN;
  hipLaunchKernelGGL(glu_bwd_x3_kernel, dim3(ew_grid(rows * (inter / 4))), dim3(256), 0, (hipStream_t)stream, ab, dh, dab, (bf16_t*)planes, (long)rows, inter,
                     img_format(true));
  return (int)hipGetLastError();
}
// bf16, inter % 8 == 0: eight columns of one row per thread (16-byte accesses), rows walked by blockIdx.y - no 64-bit index
// division per element (the generic kernels above spend more on `i / (inter / 4)` than on the arithmetic), erf evaluated once per
// element in the backward.  Same expressions per element as the generic kernels -> the same bits.
__device__ __forceinline__ void unpack8_bf16(const u32x4& v, float (&o)[8]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) { o[2 * j] = __uint_as_float(v[j] << 16); o[2 * j + 1] = __uint_as_float(v[j] & 0xffff0000u); }
}
__device__ __forceinline__ u32x4 pack8_bf16(const float (&x)[8]) {
  return u32x4{pack2_bf16(x[0], x[1]), pack2_bf16(x[2], x[3]), pack2_bf16(x[4], x[5]), pack2_bf16(x[6], x[7])};
}
__global__ __launch_bounds__(256) void glu_fwd8_kernel(const bf16_t* __restrict__ ab, bf16_t* __restrict__ h, long rows, int inter) {
  const int c = (blockIdx.x * 256 + threadIdx.x) * 8;
  if (c >= inter) return;
  for (long r = blockIdx.y; r < rows; r += gridDim.y) {
    const bf16_t* src = ab + r * 2 * inter + c;
    const u32x4 ra = *(const u32x4*)src, rb = *(const u32x4*)(src + inter);
    float a[8], b[8], o[8];
    unpack8_bf16(ra, a); unpack8_bf16(rb, b);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = gelu_erf_t<true>(a[j]) * b[j];
    *(u32x4*)(h + r * inter + c) = pack8_bf16(o);
  }
}
__global__ __launch_bounds__(256) void glu_bwd8_kernel(const bf16_t* __restrict__ ab, const bf16_t* __restrict__ dh, bf16_t* __restrict__ dab,
                                                       long rows, int inter) {
  const int c = (blockIdx.x * 256 + threadIdx.x) * 8;
  if (c >= inter) return;
  for (long r = blockIdx.y; r < rows; r += gridDim.y) {
    const bf16_t* src = ab + r * 2 * inter + c;
    const u32x4 ra = *(const u32x4*)src, rb = *(const u32x4*)(src + inter), rd = *(const u32x4*)(dh + r * inter + c);
    float a[8], b[8], d[8], da[8], db[8];
    unpack8_bf16(ra, a); unpack8_bf16(rb, b); unpack8_bf16(rd, d);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float e = erf_rsqrt2<true>(a[j]);
      const float cdf = 0.5f * (1.0f + e), pdf = 0.39894228040143267794f * __expf(-0.5f * a[j] * a[j]);
      da[j] = d[j] * b[j] * (cdf + a[j] * pdf);               // == d * b * gelu_erf_grad(a)
      db[j] = d[j] * (0.5f * a[j] * (1.0f + e));              // == d * gelu_erf(a)
    }
    bf16_t* dst = dab + r * 2 * inter + c;
    *(u32x4*)dst = pack8_bf16(da);
    *(u32x4*)(dst + inter) = pack8_bf16(db);
  }
}
static inline dim3 glu8_grid(long rows, int inter) {
  const int gx = (inter / 8 + 255) / 256;
  long gy = 16384 / gx; if (gy > rows) gy = rows; if (gy < 1) gy = 1;
  return dim3(gx, (unsigned)gy);
}

extern "C" int muse_glu_fwd(const void* ab, void* h, int32_t dtype, int64_t rows, int32_t inter, void* stream) {
  if (inter % 4) return MUSE_ERR_BAD_ARG;
  if (rows <= 0) return 0;
  if (dtype == MUSE_BF16 && (inter % 8) == 0 && !((((uintptr_t)ab) | ((uintptr_t)h)) & 15)) {
    hipLaunchKernelGGL(glu_fwd8_kernel, glu8_grid(rows, inter), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)ab, (bf16_t*)h, (long)rows, inter);
    return (int)hipGetLastError();
  }
  const int g = ew_grid(rows * (inter / 4));
  if (dtype == MUSE_F32) hipLaunchKernelGGL(glu_fwd_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)ab, (float*)h, (long)rows, inter);
  else hipLaunchKernelGGL(glu_fwd_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)ab, (bf16_t*)h, (long)rows, inter);
  return (int)hipGetLastError();
}
extern "C" int muse_glu_bwd(const void* ab, const void* dh, void* dab, int32_t dtype, int64_t rows, int32_t inter, void* stream) {
  if (inter % 4) return MUSE_ERR_BAD_ARG;
  if (rows <= 0) return 0;
  if (dtype == MUSE_BF16 && (inter % 8) == 0 && !((((uintptr_t)ab) | ((uintptr_t)dh) | ((uintptr_t)dab)) & 15)) {
    hipLaunchKernelGGL(glu_bwd8_kernel, glu8_grid(rows, inter), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)ab, (const bf16_t*)dh, (bf16_t*)dab, (long)rows, inter);
    return (int)hipGetLastError();
  }
  const int g = ew_grid(rows * (inter / 4));
  if (dtype == MUSE_F32) hipLaunchKernelGGL(glu_bwd_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)ab, (const float*)dh, (float*)dab, (long)rows, inter);
  else hipLaunchKernelGGL(glu_bwd_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)ab, (const bf16_t*)dh, (bf16_t*)dab, (long)rows, inter);
  return (int)hipGetLastError();
}

template <typename T, int MODE>  // MODE 0: y = gelu(x); 1: dx = dy * gelu'(x)
__global__ void gelu_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ out, long n) {
  const long n4 = n / 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float a[4], o[4];
    V4<T>::load(x + i * 4, a);
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = gelu_erf_t<sizeof(T) == 2>(a[j]);
    } else {
      float d[4]; V4<T>::load(dy + i * 4, d);
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = d[j] * gelu_erf_grad_t<sizeof(T) == 2>(a[j]);
    }
    V4<T>::store(out + i * 4, o);
  }
}
extern "C" int muse_gelu_fwd(const void* x, void* y, int32_t dtype, int64_t n, void* stream) {
  if (n % 4) return MUSE_ERR_BAD_ARG;
  if (n <= 0) return 0;
  if (dtype == MUSE_F32) hipLaunchKernelGGL((gelu_kernel<float, 0>), dim3(ew_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (const float*)nullptr, (float*)y, (long)n);
  else hipLaunchKernelGGL((gelu_kernel<bf16_t, 0>), dim3(ew_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)nullptr, (bf16_t*)y, (long)n);
  return (int)hipGetLastError();
}
// dx = dy * gelu'(x) from f32 x / dy, written as bf16: the dY operand of the next weight GEMMs in the bf16 compute mode (what a cast
// pass over the f32 result would produce, bit for bit)
__global__ void gelu_bwd_f32_bf16_kernel(const float* __restrict__ x, const float* __restrict__ dy, bf16_t* __restrict__ out, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 a = *(const f32x4*)(x + i * 4), d = *(const f32x4*)(dy + i * 4);
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = d[j] * gelu_erf_grad(a[j]);
    *(u32x2*)(out + i * 4) = u32x2{pack2_bf16(o[0], o[1]), pack2_bf16(o[2], o[3])};
  }
}
extern "C" int muse_gelu_bwd_f32_bf16(const float* x, const float* dy, void* dx_bf16, int64_t n, void* stream) {
  if (n % 4) return MUSE_ERR_BAD_ARG;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(gelu_bwd_f32_bf16_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, x, dy, (bf16_t*)dx_bf16, (long)(n / 4));
  return (int)hipGetLastError();
}
extern "C" int muse_gelu_bwd(const void* x, const void* dy, void* dx, int32_t dtype, int64_t n, void* stream) {
  if (n % 4) return MUSE_ERR_BAD_ARG;
  if (n <= 0) return 0;
  if (dtype == MUSE_F32) hipLaunchKernelGGL((gelu_kernel<float, 1>), dim3(ew_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (const float*)dy, (float*)dx, (long)n);
  else hipLaunchKernelGGL((gelu_kernel<bf16_t, 1>), dim3(ew_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, (long)n);
  return (int)hipGetLastError();
}

// =================================================================================================================
// Embedding: word[ids] + pos[s]  (f32).  Backward is deterministic: no float atomics anywhere.
// =================================================================================================================
__global__ void embed_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ word, const float* __restrict__ pos,
                                 float* __restrict__ out, int seq, int hidden, int vocab) {
  const int t = blockIdx.x;  // token index b*seq + s
  const int s = t % seq;
  long id = ids[t];
  if (id < 0 || id >= vocab) id = 0;  // torch would raise; keep the kernel memory safe
  for (int c = threadIdx.x * 4; c < hidden; c += blockDim.x * 4) {
    float a[4], b[4], o[4];
    V4<float>::load(word + id * hidden + c, a); V4<float>::load(pos + (long)s * hidden + c, b);
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = a[j] + b[j];
    V4<float>::store(out + (long)t * hidden + c, o);
  }
}
extern "C" int muse_embed_fwd(const int64_t* ids, const float* word, const float* pos, float* out, int32_t batch,
                              int32_t seq, int32_t hidden, int32_t vocab, void* stream) {
  if (hidden % 4) return MUSE_ERR_BAD_ARG;
  if (batch * seq <= 0) return 0;
  hipLaunchKernelGGL(embed_fwd_kernel, dim3(batch * seq), dim3(64), 0, (hipStream_t)stream, ids, word, pos, out, seq, hidden, vocab);
  return (int)hipGetLastError();
}

// Token splits of the word-embedding gradient.  partial[split][v][:] costs split * vocab * hidden floats of HBM traffic twice
// (write + reduce), so the split count is bounded by a 64 MiB partial buffer: 8 for vocab 2048 x 768 (was a fixed 32: 201 MB
// written and re-read, 765 us of the 80 ms step), 2 for vocab 8256 x 768, 32 for small tables.
static inline int emb_splits(int hidden, int vocab) {
  long s = (16L << 20) / ((long)hidden * vocab > 0 ? (long)hidden * vocab : 1);
  int p = 1;
  while (p * 2 <= s && p < 32) p *= 2;
  return p;
}
// partial[split][v][:] = sum (in position order) of dout[t,:] over tokens t in this split with ids[t] == v
__global__ __launch_bounds__(256) void embed_bwd_partial_kernel(const int64_t* __restrict__ ids, const float* __restrict__ dout,
                                                                float* __restrict__ partial, int ntok, int hidden, int vocab, int nsplit) {
  const int v = blockIdx.x, sp = blockIdx.y;
  const int per = (ntok + nsplit - 1) / nsplit;
  const int t0 = sp * per, t1 = min(ntok, t0 + per);
  __shared__ int hits[256];
  __shared__ int nhit;
  __shared__ int wcnt[4];
  // up to 16 columns per thread (hidden <= 4096)
  float acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  const int ncol = (hidden - (int)threadIdx.x + 255) / 256;   // columns threadIdx.x + 256 j < hidden
  for (int base = t0; base < t1; base += 256) {
    const int t = base + threadIdx.x;
    const bool hit = (t < t1) && (ids[t] == (int64_t)v);
    // ordered compaction of this chunk's hits
    const unsigned long long bal = __ballot(hit);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wcnt[wave] = __popcll(bal);
    __syncthreads();
    int off = 0;
    for (int w = 0; w < wave; ++w) off += wcnt[w];
    if (hit) hits[off + __popcll(bal & ((1ull << lane) - 1ull))] = t;
    if (threadIdx.x == 0) nhit = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    __syncthreads();
    const int n = nhit;
    int h = 0;
    for (; h + 4 <= n; h += 4) {   // four rows in flight, added in position order
      const float* s0 = dout + (long)hits[h] * hidden;
      const float* s1 = dout + (long)hits[h + 1] * hidden;
      const float* s2 = dout + (long)hits[h + 2] * hidden;
      const float* s3 = dout + (long)hits[h + 3] * hidden;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (j < ncol) {
          const int c = threadIdx.x + j * 256;
          const float a0 = s0[c], a1 = s1[c], a2 = s2[c], a3 = s3[c];
          acc[j] = (((acc[j] + a0) + a1) + a2) + a3;
        }
      }
    }
    for (; h < n; ++h) {
      const float* src = dout + (long)hits[h] * hidden;
#pragma unroll
      for (int j = 0; j < 16; ++j) if (j < ncol) acc[j] += src[threadIdx.x + j * 256];
    }
    __syncthreads();
  }
  float* dst = partial + ((long)sp * vocab + v) * hidden;
#pragma unroll
  for (int j = 0; j < 16; ++j) { const int c = threadIdx.x + j * 256; if (c < hidden) dst[c] = acc[j]; }
}
__global__ void embed_bwd_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dword, long n, int acc, int nsplit) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int sp = 0; sp < nsplit; ++sp) s += partial[(long)sp * n + i];
    dword[i] = acc ? dword[i] + s : s;
  }
}
__global__ void embed_bwd_pos_kernel(const float* __restrict__ dout, float* __restrict__ dpos, int batch, int seq, int hidden, int acc) {
  const int s = blockIdx.x;
  for (int c = threadIdx.x; c < hidden; c += blockDim.x) {
    float a = 0.f;
    for (int b = 0; b < batch; ++b) a += dout[((long)b * seq + s) * hidden + c];
    float* d = dpos + (long)s * hidden + c;
    *d = acc ? *d + a : a;
  }
}
extern "C" int muse_embed_bwd(const int64_t* ids, const float* dout, float* dword, float* dpos, float* scratch,
                              int32_t batch, int32_t seq, int32_t hidden, int32_t vocab, int32_t accumulate, void* stream) {
  if (hidden > 4096) return MUSE_ERR_UNSUPPORTED;
  if (batch * seq <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const int nsplit = emb_splits(hidden, vocab);
  hipLaunchKernelGGL(embed_bwd_partial_kernel, dim3(vocab, nsplit), dim3(256), 0, s, ids, dout, scratch, batch * seq, hidden, vocab, nsplit);
  const long n = (long)vocab * hidden;
  hipLaunchKernelGGL(embed_bwd_reduce_kernel, dim3(ew_grid(n)), dim3(256), 0, s, (const float*)scratch, dword, n, accumulate, nsplit);
  hipLaunchKernelGGL(embed_bwd_pos_kernel, dim3(seq), dim3(256), 0, s, dout, dpos, batch, seq, hidden, accumulate);
  return (int)hipGetLastError();
}
extern "C" int64_t muse_embed_bwd_scratch_floats(int32_t hidden, int32_t vocab) { return (int64_t)emb_splits(hidden, vocab) * vocab * hidden; }

// =================================================================================================================
// Cross entropy (ignore_index = -100, label smoothing, mean over valid rows)
// =================================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const T* __restrict__ logits, const int64_t* __restrict__ labels,
                                                     float* __restrict__ row_loss, float* __restrict__ lse_o, long rows,
                                                     int vocab, long ld, float ls) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = logits + row * ld;
  float m = -INFINITY, sx = 0.f;
  for (int c = lane; c < vocab; c += 64) { const float v = Elem<T>::load(xr + c); m = fmaxf(m, v); sx += v; }
  m = wave_max(m);
  float s = 0.f;
  for (int c = lane; c < vocab; c += 64) s += expf(Elem<T>::load(xr + c) - m);
  s = wave_sum(s);
  sx = wave_sum(sx);
  const float lse = m + logf(s);
  if (lane == 0) {
    lse_o[row] = lse;
    const int64_t lab = labels[row];
    float l = 0.f;
    if (lab >= 0 && lab < vocab) {
      const float nll = lse - Elem<T>::load(xr + lab);
      const float smooth = lse - sx / (float)vocab;
      l = (1.0f - ls) * nll + ls * smooth;
    }
    row_loss[row] = l;
  }
}
__global__ __launch_bounds__(1024) void ce_reduce_kernel(const float* __restrict__ row_loss, const int64_t* __restrict__ labels,
                                                         float* __restrict__ out, long rows, int vocab) {
  __shared__ double ssum[16];
  __shared__ int scnt[16];
  double s = 0.0; int n = 0;
  for (long r = threadIdx.x; r < rows; r += 1024) {
    const int64_t lab = labels[r];
    if (lab >= 0 && lab < vocab) { s += (double)row_loss[r]; ++n; }
  }
  s = wave_sum_d(s);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
  if ((threadIdx.x & 63) == 0) { ssum[threadIdx.x >> 6] = s; scnt[threadIdx.x >> 6] = n; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0; int c = 0;
    for (int i = 0; i < 16; ++i) { t += ssum[i]; c += scnt[i]; }
    out[0] = (float)(t / (double)c);  // 0/0 -> nan, like torch
    out[1] = (float)c;
  }
}
extern "C" int muse_cross_entropy_fwd(const void* logits, int32_t dtype, const int64_t* labels, float* row_loss, float* lse,
                                      float* loss_out, int64_t rows, int32_t vocab, int64_t ld, float label_smoothing,
                                      void* stream) {
  if (rows <= 0) return MUSE_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)((rows + 3) / 4));
  if (dtype == MUSE_F32) hipLaunchKernelGGL(ce_fwd_kernel<float>, grid, dim3(256), 0, s, (const float*)logits, labels, row_loss, lse, (long)rows, vocab, (long)ld, label_smoothing);
  else hipLaunchKernelGGL(ce_fwd_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)logits, labels, row_loss, lse, (long)rows, vocab, (long)ld, label_smoothing);
  hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(1024), 0, s, (const float*)row_loss, labels, loss_out, (long)rows, vocab);
  return (int)hipGetLastError();
}

template <typename T, typename TO>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const T* __restrict__ logits, const int64_t* __restrict__ labels,
                                                     const float* __restrict__ lse, const float* __restrict__ loss_out,
                                                     const float* __restrict__ gout, TO* __restrict__ dl, long rows, int vocab,
                                                     long ld, float ls) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t lab = labels[row];
  TO* dr = dl + row * ld;
  if (lab < 0 || lab >= vocab) {
    for (int c = lane; c < (int)ld; c += 64) Elem<TO>::store(dr + c, 0.f);
    return;
  }
  const float g = gout[0] / loss_out[1];
  const float l = lse[row];
  const T* xr = logits + row * ld;
  const float off = ls / (float)vocab;
  for (int c = lane; c < (int)ld; c += 64) {  // pad columns [vocab, ld) are written as 0 (they feed GEMM K-chunks)
    float p = 0.f;
    if (c < vocab) {
      p = expf(Elem<T>::load(xr + c) - l) - off;
      if (c == (int)lab) p -= (1.0f - ls);
    }
    Elem<TO>::store(dr + c, p * g);
  }
}
extern "C" int muse_cross_entropy_bwd(const void* logits, int32_t dtype, const int64_t* labels, const float* lse,
                                      const float* loss_out, const float* grad_out, void* dlogits, int32_t dl_dtype,
                                      int64_t rows, int32_t vocab, int64_t ld, float label_smoothing, void* stream) {
  if (rows <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)((rows + 3) / 4));
#define CEB(T, TO) hipLaunchKernelGGL((ce_bwd_kernel<T, TO>), grid, dim3(256), 0, s, (const T*)logits, labels, lse, loss_out, grad_out, (TO*)dlogits, (long)rows, vocab, (long)ld, label_smoothing)
  if (dtype == MUSE_F32 && dl_dtype == MUSE_F32) CEB(float, float);
  else if (dtype == MUSE_F32 && dl_dtype == MUSE_BF16) CEB(float, bf16_t);
  else if (dtype == MUSE_BF16 && dl_dtype == MUSE_F32) CEB(bf16_t, float);
  else CEB(bf16_t, bf16_t);
#undef CEB
  return (int)hipGetLastError();
}

// =================================================================================================================
// AdamW over a flat f32 buffer; optional bf16 shadow refresh.  7 x 4 B per parameter of HBM traffic (+2 B shadow).
// =================================================================================================================
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16_t* __restrict__ pb, long n, float lr, float b1,
                                                    float b2, float eps, float decay, float omb1, float omb2,
                                                    float step_size, float bc2_sqrt, float gscale) {
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float pp[4], gg[4], mm[4], vv[4];
    V4<float>::load(p + i * 4, pp); V4<float>::load(g + i * 4, gg); V4<float>::load(m + i * 4, mm); V4<float>::load(v + i * 4, vv);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gr = gg[j] * gscale;
      pp[j] = pp[j] * decay;                           // param.mul_(1 - lr * weight_decay), factor rounded once on the host
      mm[j] = fmaf(omb1, gr - mm[j], mm[j]);           // exp_avg.lerp_(grad, 1 - beta1)
      vv[j] = fmaf(omb2, gr * gr, vv[j] * b2);         // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
      const float denom = sqrtf(vv[j]) / bc2_sqrt + eps;
      pp[j] = pp[j] - step_size * (mm[j] / denom);     // param.addcdiv_(exp_avg, denom, value=-step_size)
    }
    V4<float>::store(p + i * 4, pp); V4<float>::store(m + i * 4, mm); V4<float>::store(v + i * 4, vv);
    if (pb) V4<bf16_t>::store(pb + i * 4, pp);
  }
  // tail (n % 4)
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long i = (n4 << 2) + threadIdx.x;
    const float gr = g[i] * gscale;
    float pp = p[i] * decay;
    const float mm = fmaf(omb1, gr - m[i], m[i]);
    const float vv = fmaf(omb2, gr * gr, v[i] * b2);
    pp = pp - step_size * (mm / (sqrtf(vv) / bc2_sqrt + eps));
    p[i] = pp; m[i] = mm; v[i] = vv;
    if (pb) pb[i] = f32_to_bf16(pp);
  }
}
extern "C" int muse_adamw_flat(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr, float beta1,
                               float beta2, float eps, float weight_decay, int32_t step, float grad_scale, void* stream) {
  if (n <= 0) return 0;
  if ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) return MUSE_ERR_ALIGN;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  const float decay = (float)(1.0 - (double)lr * (double)weight_decay);
  const float omb1 = (float)(1.0 - (double)beta1), omb2 = (float)(1.0 - (double)beta2);
  hipLaunchKernelGGL(adamw_kernel, dim3(ew_grid((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16_t*)p_bf16,
                     (long)n, lr, beta1, beta2, eps, decay, omb1, omb2, step_size, bc2_sqrt, grad_scale);
  return (int)hipGetLastError();
}

// Multi-tensor form: ONE launch over a device-side table of tensors (models whose parameters are ordinary tensors, not views of
// a flat buffer: MaskGiTUViT has ~500, and 500 launches of 9 us each were 4 % of its step).  The table is 6 x int64 per tensor:
// {p, g, m, v, p_bf16 or 0, n}; `chunk_first[t]` = index of tensor t's first 4096-element chunk (exclusive prefix sum, nt + 1
// entries); block b owns chunk b: binary search -> (tensor, offset).  Same arithmetic, same order, as adamw_kernel.
// Overflow guard of the "f16" compute mode (muse_adamw_skip_flag): when set, the multi-tensor kernels read *skip first and leave every
// tensor untouched if it is non-zero - the gradients of a backward pass whose operand images overflowed half's range are NaN, and the
// update is skipped ON THE DEVICE (torch.cuda.amp.GradScaler's found_inf, without a host round trip).  Process state like
// muse_operand_images; NULL (default) = no guard.
static const int* g_adamw_skip = nullptr;
extern "C" int muse_adamw_skip_flag(const int32_t* flag) { g_adamw_skip = (const int*)flag; return 0; }
__global__ __launch_bounds__(256) void adamw_multi_kernel(const long* __restrict__ table, const int* __restrict__ chunk_first, int nt,
                                                          float b2, float eps, float decay, float omb1, float omb2,
                                                          float step_size, float bc2_sqrt, float gscale, const int* __restrict__ skip) {
  if (skip && *skip != 0) return;
  int lo = 0, hi = nt;                    // largest t with chunk_first[t] <= blockIdx.x
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (chunk_first[mid] <= (int)blockIdx.x) lo = mid; else hi = mid; }
  const long* e = table + (long)lo * 6;
  float* p = (float*)e[0]; const float* g = (const float*)e[1]; float* m = (float*)e[2]; float* v = (float*)e[3];
  bf16_t* pb = (bf16_t*)e[4];
  const long n = e[5], base = (long)((int)blockIdx.x - chunk_first[lo]) * 4096;
  const long end = base + 4096 < n ? base + 4096 : n;
  const bool vec = !((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) && !(((uintptr_t)pb) & 7);
  if (vec) {
    for (long i = base + threadIdx.x * 4; i + 3 < end; i += 1024) {
      float pp[4], gg[4], mm[4], vv[4];
      V4<float>::load(p + i, pp); V4<float>::load(g + i, gg); V4<float>::load(m + i, mm); V4<float>::load(v + i, vv);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float gr = gg[j] * gscale;
        pp[j] = pp[j] * decay;
        mm[j] = fmaf(omb1, gr - mm[j], mm[j]);
        vv[j] = fmaf(omb2, gr * gr, vv[j] * b2);
        const float denom = sqrtf(vv[j]) / bc2_sqrt + eps;
        pp[j] = pp[j] - step_size * (mm[j] / denom);
      }
      V4<float>::store(p + i, pp); V4<float>::store(m + i, mm); V4<float>::store(v + i, vv);
      if (pb) V4<bf16_t>::store(pb + i, pp);
    }
  }
  // scalar: the whole chunk when a pointer is unaligned, else the (n % 4) tail of the tensor's last chunk
  const long s0 = vec ? base + ((end - base) & ~3L) : base;
  for (long i = s0 + threadIdx.x; i < end; i += 256) {
    const float gr = g[i] * gscale;
    float pp = p[i] * decay;
    const float mm = fmaf(omb1, gr - m[i], m[i]);
    const float vv = fmaf(omb2, gr * gr, v[i] * b2);
    pp = pp - step_size * (mm / (sqrtf(vv) / bc2_sqrt + eps));
    p[i] = pp; m[i] = mm; v[i] = vv;
    if (pb) pb[i] = f32_to_bf16(pp);
  }
}
// Exponential moving average of the weights (reference muse/modeling_ema.py:118-137, called right behind the optimizer step,
// training/train_muse.py:779-780): shadow -= (1 - decay) * (shadow - param) over EVERY tracked tensor in one launch - the reference
// issues three elementwise kernels per tensor.  Table: 4 x int64 per tensor {shadow, param, n, mode}; mode 0 = the update, mode 1 =
// plain copy (a parameter with requires_grad == False, :134-135).  chunk_first as in adamw_multi_kernel.  The three roundings of the
// reference's expression (subtract, multiply, subtract - each an f32 tensor op there) are kept: no contraction into an fma.
__device__ __forceinline__ float ema_update1(float s, float p, float omd) {
#pragma clang fp contract(off)
  const float t = s - p;
  const float u = omd * t;
  return s - u;
}
__global__ __launch_bounds__(256) void ema_multi_kernel(const long* __restrict__ table, const int* __restrict__ chunk_first, int nt, float omd) {
  int lo = 0, hi = nt;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (chunk_first[mid] <= (int)blockIdx.x) lo = mid; else hi = mid; }
  const long* e = table + (long)lo * 4;
  float* s = (float*)e[0]; const float* p = (const float*)e[1];
  const long n = e[2], base = (long)((int)blockIdx.x - chunk_first[lo]) * 4096;
  const bool copy = e[3] != 0;
  const long end = base + 4096 < n ? base + 4096 : n;
  const bool vec = !((((uintptr_t)s) | ((uintptr_t)p)) & 15);
  if (vec) {
    for (long i = base + threadIdx.x * 4; i + 3 < end; i += 1024) {
      float ss[4], pp[4];
      V4<float>::load(s + i, ss); V4<float>::load(p + i, pp);
#pragma unroll
      for (int j = 0; j < 4; ++j) ss[j] = copy ? pp[j] : ema_update1(ss[j], pp[j], omd);
      V4<float>::store(s + i, ss);
    }
  }
  const long s0 = vec ? base + ((end - base) & ~3L) : base;
  for (long i = s0 + threadIdx.x; i < end; i += 256) s[i] = copy ? p[i] : ema_update1(s[i], p[i], omd);
}
extern "C" int muse_ema_multi(const int64_t* table, const int32_t* chunk_first, int32_t num_tensors, int32_t num_chunks,
                              float one_minus_decay, void* stream) {
  if (num_tensors <= 0 || num_chunks <= 0) return 0;
  if (!table || !chunk_first) return MUSE_ERR_BAD_ARG;
  hipLaunchKernelGGL(ema_multi_kernel, dim3(num_chunks), dim3(256), 0, (hipStream_t)stream, (const long*)table, chunk_first, num_tensors,
                     one_minus_decay);
  return (int)hipGetLastError();
}

extern "C" int muse_adamw_multi(const int64_t* table, const int32_t* chunk_first, int32_t num_tensors, int32_t num_chunks, float lr,
                                float beta1, float beta2, float eps, float weight_decay, int32_t step, float grad_scale, void* stream) {
  if (num_tensors <= 0 || num_chunks <= 0) return 0;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  const float decay = (float)(1.0 - (double)lr * (double)weight_decay);
  const float omb1 = (float)(1.0 - (double)beta1), omb2 = (float)(1.0 - (double)beta2);
  hipLaunchKernelGGL(adamw_multi_kernel, dim3(num_chunks), dim3(256), 0, (hipStream_t)stream, (const long*)table, chunk_first, num_tensors,
                     beta2, eps, decay, omb1, omb2, step_size, bc2_sqrt, grad_scale, g_adamw_skip);
  return (int)hipGetLastError();
}

// ---- parameter groups (training/train_muse.py:425-445: no weight decay on bias / LayerNorm / embedding weights) ---------------------
// torch.optim semantics: every group carries its own lr / betas / eps / weight_decay.  The per-group constants are computed on the
// host exactly like muse_adamw_flat computes its single set and travel BY VALUE in the kernel arguments (they change every step with
// the lr schedule: no host -> device copy).  Same arithmetic, same order as adamw_kernel: a one-group call is bit-identical to it.
#define MUSE_ADAMW_MAX_GROUPS 8
struct AdamHyper { float b2, eps, decay, omb1, omb2, step_size, bc2_sqrt, pad; };
struct AdamGroups { AdamHyper h[MUSE_ADAMW_MAX_GROUPS]; };
static inline int adam_fill_groups(AdamGroups& G, const float* hyper, int ngroups, int step) {
  if (ngroups < 1 || ngroups > MUSE_ADAMW_MAX_GROUPS || !hyper) return MUSE_ERR_BAD_ARG;
  for (int k = 0; k < ngroups; ++k) {
    const float lr = hyper[k * 5 + 0], beta1 = hyper[k * 5 + 1], beta2 = hyper[k * 5 + 2], eps = hyper[k * 5 + 3], wd = hyper[k * 5 + 4];
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    AdamHyper& h = G.h[k];
    h.step_size = (float)((double)lr / bc1);
    h.bc2_sqrt = (float)sqrt(bc2);
    h.decay = (float)(1.0 - (double)lr * (double)wd);
    h.omb1 = (float)(1.0 - (double)beta1); h.omb2 = (float)(1.0 - (double)beta2);
    h.b2 = beta2; h.eps = eps; h.pad = 0.f;
  }
  for (int k = ngroups; k < MUSE_ADAMW_MAX_GROUPS; ++k) G.h[k] = G.h[0];
  return 0;
}
__device__ __forceinline__ void adam_update1(float& pp, float gr, float& mm, float& vv, const AdamHyper& h) {
  pp = pp * h.decay;
  mm = fmaf(h.omb1, gr - mm, mm);
  vv = fmaf(h.omb2, gr * gr, vv * h.b2);
  const float denom = sqrtf(vv) / h.bc2_sqrt + h.eps;
  pp = pp - h.step_size * (mm / denom);
}
// Flat buffer cut into segments: seg_end[s] (ascending, ABSOLUTE element offsets in the flat buffer) closes segment s, seg_group[s]
// names its parameter group.  The call covers elements [base, base + n) of the flat buffer (p, g, m, v, pb point at element `base`):
// any slice, so the in-backward / behind-the-all-reduce range updates share the table.  Block b owns elements [4096 b, 4096 b + 4096)
// of the slice; its first / last segment are found once per block (uniform binary searches), a lane then only steps inside that range.
__global__ __launch_bounds__(256) void adamw_groups_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                           float* __restrict__ v, bf16_t* __restrict__ pb, long n, long base,
                                                           const long* __restrict__ seg_end, const int* __restrict__ seg_group, int nseg,
                                                           AdamGroups G, float gscale) {
  const long c0 = (long)blockIdx.x * 4096, c1 = c0 + 4096 < n ? c0 + 4096 : n;
  auto seg_of = [&](long pos) {   // smallest s with seg_end[s] > pos (positions beyond the last end: the last segment)
    int lo = 0, hi = nseg - 1;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (seg_end[mid] > pos) hi = mid; else lo = mid + 1; }
    return lo;
  };
  const int s0 = seg_of(base + c0), s1 = seg_of(base + c1 - 1);
  if (s0 == s1) {                 // the common case: one group for the whole chunk
    const AdamHyper h = G.h[seg_group[s0] & (MUSE_ADAMW_MAX_GROUPS - 1)];
    long i = c0 + threadIdx.x * 4;
    for (; i + 3 < c1; i += 1024) {
      float pp[4], gg[4], mm[4], vv[4];
      V4<float>::load(p + i, pp); V4<float>::load(g + i, gg); V4<float>::load(m + i, mm); V4<float>::load(v + i, vv);
#pragma unroll
      for (int j = 0; j < 4; ++j) adam_update1(pp[j], gg[j] * gscale, mm[j], vv[j], h);
      V4<float>::store(p + i, pp); V4<float>::store(m + i, mm); V4<float>::store(v + i, vv);
      if (pb) V4<bf16_t>::store(pb + i, pp);
    }
    const long t0 = c0 + ((c1 - c0) & ~3L);
    for (long k = t0 + threadIdx.x; k < c1; k += 256) {
      float pp = p[k], mm = m[k], vv = v[k];
      adam_update1(pp, g[k] * gscale, mm, vv, h);
      p[k] = pp; m[k] = mm; v[k] = vv;
      if (pb) pb[k] = f32_to_bf16(pp);
    }
    return;
  }
  for (long k = c0 + threadIdx.x; k < c1; k += 256) {   // a chunk with a segment boundary inside: element-wise, group per element
    int s = s0;
    while (s < s1 && seg_end[s] <= base + k) ++s;
    const AdamHyper h = G.h[seg_group[s] & (MUSE_ADAMW_MAX_GROUPS - 1)];
    float pp = p[k], mm = m[k], vv = v[k];
    adam_update1(pp, g[k] * gscale, mm, vv, h);
    p[k] = pp; m[k] = mm; v[k] = vv;
    if (pb) pb[k] = f32_to_bf16(pp);
  }
}
extern "C" int muse_adamw_flat_groups(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, int64_t base,
                                      const int64_t* seg_end, const int32_t* seg_group, int32_t nseg, const float* group_hyper,
                                      int32_t ngroups, int32_t step, float grad_scale, void* stream) {
  if (n <= 0) return 0;
  if (nseg < 1 || !seg_end || !seg_group) return MUSE_ERR_BAD_ARG;
  if ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) return MUSE_ERR_ALIGN;
  if (p_bf16 && (((uintptr_t)p_bf16) & 7)) return MUSE_ERR_ALIGN;
  AdamGroups G;
  const int rc = adam_fill_groups(G, group_hyper, ngroups, step);
  if (rc) return rc;
  hipLaunchKernelGGL(adamw_groups_kernel, dim3((unsigned)((n + 4095) / 4096)), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                     (bf16_t*)p_bf16, (long)n, (long)base, (const long*)seg_end, seg_group, nseg, G, grad_scale);
  return (int)hipGetLastError();
}
// Multi-tensor form with groups: `table` is 7 x int64 per tensor {p, g, m, v, p_bf16 or 0, n, group | lo_plane_distance << 8}
// (lo_plane_distance < 0: p_bf16 receives an IEEE-half copy).
__global__ __launch_bounds__(256) void adamw_multi_groups_kernel(const long* __restrict__ table, const int* __restrict__ chunk_first, int nt,
                                                                 AdamGroups G, float gscale, const int* __restrict__ skip) {
  if (skip && *skip != 0) return;       // (the f16 mode's overflow guard: muse_adamw_skip_flag)
  int lo = 0, hi = nt;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (chunk_first[mid] <= (int)blockIdx.x) lo = mid; else hi = mid; }
  const long* e = table + (long)lo * 7;
  float* p = (float*)e[0]; const float* g = (const float*)e[1]; float* m = (float*)e[2]; float* v = (float*)e[3];
  bf16_t* pb = (bf16_t*)e[4];
  const AdamHyper h = G.h[(int)e[6] & (MUSE_ADAMW_MAX_GROUPS - 1)];
  // column 6 above bit 8: p_bf16 is the HI plane of the parameter's bf16x3 operand planes and the lo plane sits that many elements behind
  // it (hi = bf16(p), lo = bf16(p - hi): split_f2bb_kernel's arithmetic) - the planes the next step's weight GEMMs read; 0 = a plain bf16 copy
  // ... and a NEGATIVE value there: p_bf16 is an IEEE-half copy instead (the weight's operand image of the "f16" compute mode)
  const long plo_raw = e[6] >> 8;
  const bool half_copy = plo_raw < 0;
  const long plo = half_copy ? 0 : plo_raw;
  const long n = e[5], base = (long)((int)blockIdx.x - chunk_first[lo]) * 4096;
  const long end = base + 4096 < n ? base + 4096 : n;
  const bool vec = !((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) && !(((uintptr_t)pb) & 7) && !(plo & 3);
  if (vec) {
    for (long i = base + threadIdx.x * 4; i + 3 < end; i += 1024) {
      float pp[4], gg[4], mm[4], vv[4];
      V4<float>::load(p + i, pp); V4<float>::load(g + i, gg); V4<float>::load(m + i, mm); V4<float>::load(v + i, vv);
#pragma unroll
      for (int j = 0; j < 4; ++j) adam_update1(pp[j], gg[j] * gscale, mm[j], vv[j], h);
      V4<float>::store(p + i, pp); V4<float>::store(m + i, mm); V4<float>::store(v + i, vv);
      if (pb && half_copy) {
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        *(h4*)(pb + i) = h4{(_Float16)pp[0], (_Float16)pp[1], (_Float16)pp[2], (_Float16)pp[3]};
      } else if (pb) {
        V4<bf16_t>::store(pb + i, pp);
        if (plo) {
          float rr[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) rr[j] = pp[j] - bf16_to_f32(f32_to_bf16(pp[j]));
          V4<bf16_t>::store(pb + plo + i, rr);
        }
      }
    }
  }
  const long s0 = vec ? base + ((end - base) & ~3L) : base;
  for (long i = s0 + threadIdx.x; i < end; i += 256) {
    float pp = p[i], mm = m[i], vv = v[i];
    adam_update1(pp, g[i] * gscale, mm, vv, h);
    p[i] = pp; m[i] = mm; v[i] = vv;
    if (pb && half_copy) {
      ((_Float16*)pb)[i] = (_Float16)pp;
    } else if (pb) {
      const bf16_t hi = f32_to_bf16(pp);
      pb[i] = hi;
      if (plo) pb[plo + i] = f32_to_bf16(pp - bf16_to_f32(hi));
    }
  }
}
extern "C" int muse_adamw_multi_groups(const int64_t* table, const int32_t* chunk_first, int32_t num_tensors, int32_t num_chunks,
                                       const float* group_hyper, int32_t ngroups, int32_t step, float grad_scale, void* stream) {
  if (num_tensors <= 0 || num_chunks <= 0) return 0;
  AdamGroups G;
  const int rc = adam_fill_groups(G, group_hyper, ngroups, step);
  if (rc) return rc;
  hipLaunchKernelGGL(adamw_multi_groups_kernel, dim3(num_chunks), dim3(256), 0, (hipStream_t)stream, (const long*)table, chunk_first,
                     num_tensors, G, grad_scale, g_adamw_skip);
  return (int)hipGetLastError();
}

// f32 -> the two bf16 planes of the bf16x3 product scheme: hi = bf16(x), lo = bf16(x - hi) (x ~= hi + lo to 2^-16 relative); the
// operand form of every f32 GEMM of the "bf16x3" compute mode of the tape engines (three bf16 MFMA products hi*hi + hi*lo + lo*hi with
// f32 accumulation: TF32-class-or-tighter arithmetic at 1/3 of the bf16 matrix rate - gfx950 has no xf32 MFMA and its exact-f32
// MFMA peaks at 157 TFLOP/s)
__global__ void split_f2bb_kernel(const float* __restrict__ in, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, long n) {
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const u32x4 v = *(const u32x4*)(in + i * 4);
    u32x2 h, l;
    split4(v, h, l);
    *(u32x2*)(hi + i * 4) = h;
    *(u32x2*)(lo + i * 4) = l;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long i = (n4 << 2) + threadIdx.x;
    const bf16_t h = f32_to_bf16(in[i]);
    hi[i] = h;
    lo[i] = f32_to_bf16(in[i] - bf16_to_f32(h));
  }
}
extern "C" int muse_split_f32_to_bf16x2(const float* in, void* hi, void* lo, int64_t n, void* stream) {
  if (n <= 0) return 0;
  if ((((uintptr_t)in) & 15) || ((((uintptr_t)hi) | ((uintptr_t)lo)) & 7)) return MUSE_ERR_ALIGN;
  hipLaunchKernelGGL(split_f2bb_kernel, dim3(ew_grid((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, in, (bf16_t*)hi, (bf16_t*)lo, (long)n);
  return (int)hipGetLastError();
}

// The bf16x3 product as ONE GEMM: C = A_hi B_hi + A_hi B_lo + A_lo B_hi is a single product over a three times longer K with the
// operands laid out as A' = (hi | hi | lo), B' = (hi | lo | hi) along K - one launch, one epilogue, no read-modify-write of C between
// the three terms.  This kernel writes such an operand from the f32 tensor [rows, cols] (row stride ld_in): mode 0 concatenates the
// three planes along the columns (k-contiguous operand: out [rows, 3 cols]), mode 1 stacks them along the rows (k-major operand:
// out [3 rows, cols], row stride ld_out); lo_pos (1 or 2) says which third carries the lo plane.
__global__ void split_cat3_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, long rows, int cols, long ld_in, long ld_out,
                                  int mode, int lo_pos) {
  const int vpr = cols >> 2;
  const long n = rows * vpr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long r = i / vpr;
    const int c = (int)(i - r * vpr) * 4;
    const u32x4 v = *(const u32x4*)(in + r * ld_in + c);
    u32x2 h, l;
    split4(v, h, l);
#pragma unroll
    for (int sgm = 0; sgm < 3; ++sgm) {
      bf16_t* o = mode == 0 ? out + r * ld_out + (long)sgm * cols + c : out + ((long)sgm * rows + r) * ld_out + c;
      *(u32x2*)o = (sgm == lo_pos) ? l : h;
    }
  }
}
extern "C" int muse_split_f32_to_bf16_cat3(const float* in, void* out, int64_t rows, int32_t cols, int64_t ld_in, int64_t ld_out, int32_t mode,
                                           int32_t lo_pos, void* stream) {
  if (rows <= 0 || cols <= 0) return 0;
  if ((cols & 3) || (ld_in & 3) || (ld_out & 3) || (mode != 0 && mode != 1) || (lo_pos != 1 && lo_pos != 2)) return MUSE_ERR_BAD_ARG;
  if ((((uintptr_t)in) & 15) || (((uintptr_t)out) & 7)) return MUSE_ERR_ALIGN;
  hipLaunchKernelGGL(split_cat3_kernel, dim3(ew_grid(rows * (cols >> 2))), dim3(256), 0, (hipStream_t)stream, in, (bf16_t*)out, (long)rows, cols,
                     (long)ld_in, (long)ld_out, mode, lo_pos);
  return (int)hipGetLastError();
}

__global__ void cast_f2b_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, long n) {
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float a[4]; V4<float>::load(in + i * 4, a); V4<bf16_t>::store(out + i * 4, a);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const long i = (n4 << 2) + threadIdx.x; out[i] = f32_to_bf16(in[i]); }
}
__global__ void cast_b2f_kernel(const bf16_t* __restrict__ in, float* __restrict__ out, long n) {
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float a[4]; V4<bf16_t>::load(in + i * 4, a); V4<float>::store(out + i * 4, a);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const long i = (n4 << 2) + threadIdx.x; out[i] = bf16_to_f32(in[i]); }
}
extern "C" int muse_cast_f32_to_bf16(const float* in, void* out, int64_t n, void* stream) {
  if (n <= 0) return 0;
  if ((((uintptr_t)in) & 15) || (((uintptr_t)out) & 7)) return MUSE_ERR_ALIGN;
  hipLaunchKernelGGL(cast_f2b_kernel, dim3(ew_grid((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, in, (bf16_t*)out, (long)n);
  return (int)hipGetLastError();
}
// f32 -> IEEE half operand image of the "f16" compute mode (muse_gemm dtype MUSE_F16): out = half(in * scale), round to nearest even,
// subnormals kept; a finite in * scale beyond half's range becomes inf (the product, and the step's loss, turn NaN: loud - the same
// bits the producer kernels write, common.h store_image4).  stats (optional, int32[2], accumulated with atomics by the waves that see
// one): [0] finite elements that overflowed, [1] non-zero elements that became zero - the caller's gradient scale was too large / too small.
__global__ void cast_f2h_kernel(const float* __restrict__ in, _Float16* __restrict__ out, long n, float scale, int* __restrict__ stats) {
  const long n4 = n >> 2;
  bool sat = false, und = false;
  auto cv = [&](float x) -> _Float16 {
    const float v = x * scale;
    const _Float16 h = (_Float16)v;
    sat = sat || (fabsf(v) <= 3.0e38f && fabsf((float)h) > 65504.f);
    und = und || (x != 0.f && (float)h == 0.f);
    return h;
  };
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float a[4]; V4<float>::load(in + i * 4, a);
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    const h4 o = {cv(a[0]), cv(a[1]), cv(a[2]), cv(a[3])};
    *(h4*)(out + i * 4) = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const long i = (n4 << 2) + threadIdx.x; out[i] = cv(in[i]); }
  if (stats) {
    const unsigned long long bs = __ballot(sat), bu = __ballot(und);
    if ((threadIdx.x & 63) == 0) {
      if (bs) atomicAdd(stats, __popcll(bs));
      if (bu) atomicAdd(stats + 1, __popcll(bu));
    }
  }
}
extern "C" int muse_cast_f32_to_f16(const float* in, void* out, int64_t n, float scale, int32_t* stats, void* stream) {
  if (n <= 0) return 0;
  if ((((uintptr_t)in) & 15) || (((uintptr_t)out) & 7)) return MUSE_ERR_ALIGN;
  hipLaunchKernelGGL(cast_f2h_kernel, dim3(ew_grid((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, in, (_Float16*)out, (long)n, scale, (int*)stats);
  return (int)hipGetLastError();
}
extern "C" int muse_cast_bf16_to_f32(const void* in, float* out, int64_t n, void* stream) {
  if (n <= 0) return 0;
  if ((((uintptr_t)out) & 15) || (((uintptr_t)in) & 7)) return MUSE_ERR_ALIGN;
  hipLaunchKernelGGL(cast_b2f_kernel, dim3(ew_grid((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, out, (long)n);
  return (int)hipGetLastError();
}

// =================================================================================================================
// Mask sampling of the train step: cosine schedule -> count -> argsort(noise) -> threshold the argsort array.
// One block per image; rank by counting (S <= 1024), ties broken by position (stable order).
// =================================================================================================================
__global__ __launch_bounds__(256) void mask_sample_kernel(const int64_t* __restrict__ tokens, const int64_t* __restrict__ class_ids,
                                                          const float* __restrict__ timesteps, const float* __restrict__ noise,
                                                          int64_t* __restrict__ input_ids, int64_t* __restrict__ labels,
                                                          float* __restrict__ mask_prob, int seq, int64_t mask_id,
                                                          int64_t codebook, float min_rate) {
  __shared__ float nz[1024];
  __shared__ int perm[1024];
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < seq; i += 256) nz[i] = noise[(long)b * seq + i];
  __syncthreads();
  // cos(t * pi * 0.5) in the reference is fl32(fl32(t * fl32(pi)) * 0.5) then cosf; evaluate the cosine in f64 and
  // round once so the result is the correctly rounded f32 cosine of that argument.
  const float arg = (timesteps[b] * 3.14159265358979323846f) * 0.5f;
  float prob = (float)cos((double)arg);
  prob = fmaxf(prob, min_rate);
  float kf = rintf((float)seq * prob);  // torch.round = half-to-even
  kf = fmaxf(kf, 1.0f);
  const int k = (int)kf;
  for (int i = threadIdx.x; i < seq; i += 256) {
    const float v = nz[i];
    int r = 0;
    for (int j = 0; j < seq; ++j) { const float u = nz[j]; r += (u < v) || (u == v && j < i); }
    perm[r] = i;
  }
  __syncthreads();
  int64_t* ir = input_ids + (long)b * (seq + 1);
  int64_t* lr = labels + (long)b * (seq + 1);
  if (threadIdx.x == 0) { ir[0] = class_ids[b] + codebook; lr[0] = -100; mask_prob[b] = prob; }
  for (int j = threadIdx.x; j < seq; j += 256) {
    const bool masked = perm[j] < k;
    const int64_t t = tokens[(long)b * seq + j];
    ir[j + 1] = masked ? mask_id : t;
    lr[j + 1] = masked ? t : (int64_t)-100;
  }
}
extern "C" int muse_mask_sample(const int64_t* tokens, const int64_t* class_ids, const float* timesteps, const float* noise,
                                int64_t* input_ids, int64_t* labels, float* mask_prob, int32_t batch, int32_t seq,
                                int64_t mask_id, int64_t codebook_size, float min_masking_rate, void* stream) {
  if (seq > 1024 || seq <= 0) return MUSE_ERR_UNSUPPORTED;
  if (batch <= 0) return 0;
  hipLaunchKernelGGL(mask_sample_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, tokens, class_ids, timesteps, noise,
                     input_ids, labels, mask_prob, seq, mask_id, codebook_size, min_masking_rate);
  return (int)hipGetLastError();
}

// out[i] (+)= sum_s ws[s*stride + i]   (deterministic reduction of split-K partial tiles), n % 4 == 0
__global__ void sum_slices_kernel(const float* __restrict__ ws, float* __restrict__ out, int ns, long n, long stride, int acc) {
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (acc) V4<float>::load(out + i * 4, a);
    for (int s = 0; s < ns; ++s) {
      float t[4]; V4<float>::load(ws + s * stride + i * 4, t);
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] += t[j];
    }
    V4<float>::store(out + i * 4, a);
  }
}
extern "C" int muse_sum_slices(const float* ws, float* out, int32_t nslices, int64_t n, int64_t stride, int32_t accumulate,
                               void* stream) {
  if (n <= 0) return 0;
  if ((n & 3) || (stride & 3) || (((uintptr_t)ws) & 15) || (((uintptr_t)out) & 15)) return MUSE_ERR_ALIGN;
  hipLaunchKernelGGL(sum_slices_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, ws, out, nslices, (long)n,
                     (long)stride, accumulate);
  return (int)hipGetLastError();
}

// ---- several reductions in ONE launch (the weight-gradient stream's tail work of a transformer layer: the split-K slice sums of its
// grouped dW launch and the column sums of its LayerNorm / GLU weight gradients).  Two job kinds, each the literal arithmetic of the
// single-job kernel it replaces (bit-identical results: sum_slices_kernel / colsum_kernel):
//   kind 0  out[i] (+)= sum_{s < ns} ws[s * stride + i], i < n        items of 4096 outputs (1024 threads x 4)
//   kind 1  out[c] (+)= sum_{r < ns} ws[r * n + c], c < n             items of 16 columns, 64 row groups (colsum_kernel's order)
#define MUSE_SUM_MAX_JOBS 16
struct SumJob { const float* ws; float* out; long n, stride; int ns, acc, first_item, kind; };
struct SumJobs { SumJob j[MUSE_SUM_MAX_JOBS]; int n; };
__global__ __launch_bounds__(1024) void sum_multi_kernel(const SumJobs J) {
  __shared__ float red[64][17];
  int k = 0;
#pragma unroll
  for (int i = 1; i < MUSE_SUM_MAX_JOBS; ++i) k += (i < J.n && (int)blockIdx.x >= J.j[i].first_item) ? 1 : 0;
  const SumJob jb = J.j[k];
  const int item = (int)blockIdx.x - jb.first_item;
  if (jb.kind == 0) {
    const long i = (long)item * 4096 + threadIdx.x * 4;
    if (i < jb.n) {             // (n % 4 == 0)
      float a[4] = {0.f, 0.f, 0.f, 0.f};
      if (jb.acc) V4<float>::load(jb.out + i, a);
      for (int s = 0; s < jb.ns; ++s) {
        float t[4]; V4<float>::load(jb.ws + s * jb.stride + i, t);
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] += t[q];
      }
      V4<float>::store(jb.out + i, a);
    }
    return;
  }
  const int c16 = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int cols = (int)jb.n, rows = jb.ns;
  const int c = item * 16 + c16;
  float s0 = 0.f, s1 = 0.f;
  if (c < cols) {
    int r = rg;
    for (; r + 64 < rows; r += 128) { s0 += jb.ws[(long)r * cols + c]; s1 += jb.ws[(long)(r + 64) * cols + c]; }
    if (r < rows) s0 += jb.ws[(long)r * cols + c];
  }
  red[rg][c16] = s0 + s1;
  __syncthreads();
  if (rg == 0 && c < cols) {
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < 64; ++q) sum += red[q][c16];
    jb.out[c] = jb.acc ? jb.out[c] + sum : sum;
  }
}
// kind[i] = 0: slice sum (n[i], stride[i] multiples of 4, 16-byte aligned pointers); 1: column sum of a [nslices[i], n[i]] f32 matrix
extern "C" int muse_sum_multi(const void* const* ws, void* const* out, const int32_t* nslices, const int64_t* n, const int64_t* stride,
                              const int32_t* accumulate, const int32_t* kind, int32_t njobs, void* stream) {
  if (njobs <= 0) return 0;
  if (njobs > MUSE_SUM_MAX_JOBS) return MUSE_ERR_BAD_ARG;
  SumJobs J;
  int items = 0;
  for (int i = 0; i < njobs; ++i) {
    if (n[i] <= 0 || nslices[i] < 1 || (kind[i] != 0 && kind[i] != 1)) return MUSE_ERR_BAD_ARG;
    if (kind[i] == 0 && ((n[i] & 3) || (stride[i] & 3) || (((uintptr_t)ws[i]) & 15) || (((uintptr_t)out[i]) & 15))) return MUSE_ERR_ALIGN;
    SumJob& j = J.j[i];
    j.ws = (const float*)ws[i]; j.out = (float*)out[i]; j.n = n[i]; j.stride = stride[i]; j.ns = nslices[i]; j.acc = accumulate[i];
    j.first_item = items; j.kind = kind[i];
    items += kind[i] == 0 ? (int)((n[i] / 4 + 1023) / 1024) : (int)((n[i] + 15) / 16);
  }
  for (int i = njobs; i < MUSE_SUM_MAX_JOBS; ++i) J.j[i] = J.j[0];
  J.n = njobs;
  hipLaunchKernelGGL(sum_multi_kernel, dim3(items), dim3(1024), 0, (hipStream_t)stream, J);
  return (int)hipGetLastError();
}

// ---- split-K slices -> a Linear's output with its epilogue (small-batch decoding: M <= 2048 rows) -----------------------------------------
// out[r, c] = sum_{s < ns} ws[s * stride + r * cols + c] (+ bias[c]) (+ residual[r, c]) as f32 or bf16: the reduction of the K-slice
// workspace of a forward product whose 128^2 tiles alone would fill a fraction of the chip ([512 x 1024] x [1024 x 1024]^T: 32 tiles on
// 256 CUs, 36 us of a 16-K-tile loop), fused with everything muse_gemm's epilogue would have added.  Fixed summation order.
template <typename TO>
__global__ __launch_bounds__(256) void sum_slices_epi_kernel(const float* __restrict__ ws, int ns, long stride, const float* __restrict__ bias,
                                                            const TO* __restrict__ residual, long ldr, TO* __restrict__ out, long ldc,
                                                            long rows, int cols) {
  const int vpr = cols >> 2;
  const long n = rows * vpr;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long r = i / vpr;
    const int c = (int)(i - r * vpr) * 4;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < ns; ++k) {
      float t[4]; V4<float>::load(ws + k * stride + r * cols + c, t);
#pragma unroll
      for (int q = 0; q < 4; ++q) a[q] += t[q];
    }
    if (bias) { float b[4]; V4<float>::load(bias + c, b);
#pragma unroll
      for (int q = 0; q < 4; ++q) a[q] += b[q]; }
    if (residual) { float t[4]; V4<TO>::load(residual + r * ldr + c, t);
#pragma unroll
      for (int q = 0; q < 4; ++q) a[q] += t[q]; }
    V4<TO>::store(out + r * ldc + c, a);
  }
}
extern "C" int muse_sum_slices_epilogue(const float* ws, int32_t nslices, int64_t stride, const float* bias, const void* residual, int64_t ldr,
                                        void* out, int32_t out_dtype, int64_t ldc, int64_t rows, int32_t cols, void* stream) {
  if (rows <= 0 || cols <= 0) return 0;
  if ((cols & 3) || (stride & 3) || (ldc & 3) || (ldr & 3) || nslices < 1) return MUSE_ERR_BAD_ARG;
  const int ob = out_dtype == MUSE_BF16 ? 7 : 15;
  if ((((uintptr_t)ws) & 15) || (((uintptr_t)bias) & 15) || (((uintptr_t)out) & ob) || (((uintptr_t)residual) & ob)) return MUSE_ERR_ALIGN;
  const long n = rows * (cols >> 2);
  const dim3 grid(ew_grid(n));
  if (out_dtype == MUSE_BF16)
    hipLaunchKernelGGL(sum_slices_epi_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, ws, nslices, (long)stride, bias, (const bf16_t*)residual,
                       (long)ldr, (bf16_t*)out, (long)ldc, (long)rows, cols);
  else
    hipLaunchKernelGGL(sum_slices_epi_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, ws, nslices, (long)stride, bias, (const float*)residual,
                       (long)ldr, (float*)out, (long)ldc, (long)rows, cols);
  return (int)hipGetLastError();
}

extern "C" int muse_version(void) { return 1; }
