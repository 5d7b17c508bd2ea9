// Row / elementwise kernels of MaskGiTUViT_v2 (muse/modeling_transformer_v2.py) that the MaskGit
// path does not have: norm with a pre-norm residual stream (RMSNorm / LayerNorm), AdaLN modulation, SiLU, depthwise 3x3
// convolution (NHWC), GlobalResponseNorm, sinusoidal micro-conditioning, weighted mean of per-token losses - forward and
// backward, f32.
// Everything else of that model (linears, attention, GLU, GELU, gather, cross-entropy) reuses the MaskGit kernels.
#include "common.h"
#include "../../include/muse_hip.h"

// =================================================================================================================
// norm with residual stream: v = x (+ res); pre = v; y = v * rsqrt(mean(v^2) + eps) * w        (mode 0, RMSNorm :673-691)
//                                                    y = (v - mean) * rsqrt(var + eps) * w      (mode 1, LayerNorm :726-737)
// one wave per row, 4 rows per block
// =================================================================================================================
__global__ __launch_bounds__(256) void norm_res_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                           const float* __restrict__ w, float* __restrict__ y,
                                                           float* __restrict__ pre, long rows, int cols, float eps, int mode) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * cols;
  const float* rr = res ? res + row * cols : nullptr;
  float s = 0.f, q = 0.f;
  for (int c = lane * 4; c < cols; c += 256) {
    f32x4 v = *(const f32x4*)(xr + c);
    if (rr) v += *(const f32x4*)(rr + c);
    if (pre) *(f32x4*)(pre + row * cols + c) = v;
    s += (v[0] + v[1]) + (v[2] + v[3]);
    q += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  }
  float mean = 0.f, rstd;
  if (mode == 0) {
    rstd = rsqrtf(wave_sum(q) / (float)cols + eps);
  } else {
    mean = wave_sum(s) / (float)cols;
    float d2 = 0.f;   // two-pass variance (F.layer_norm accuracy)
    for (int c = lane * 4; c < cols; c += 256) {
      f32x4 v = *(const f32x4*)(xr + c);
      if (rr) v += *(const f32x4*)(rr + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float d = v[j] - mean; d2 = fmaf(d, d, d2); }
    }
    rstd = 1.0f / sqrtf(wave_sum(d2) / (float)cols + eps);
  }
  for (int c = lane * 4; c < cols; c += 256) {
    f32x4 v = *(const f32x4*)(xr + c);
    if (rr) v += *(const f32x4*)(rr + c);
    f32x4 g = {1.f, 1.f, 1.f, 1.f};
    if (w) g = *(const f32x4*)(w + c);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (v[j] - mean) * rstd * g[j];
    *(f32x4*)(y + row * cols + c) = o;
  }
}
// cols <= 1024: the row stays in registers between the statistics and the output pass (one read of x / res instead of two or
// three; the general kernel above leans on the caches for the re-reads).  Same expressions and summation order -> the same bits.
template <int NIT>
__global__ __launch_bounds__(256) void norm_res_fwd_reg_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                               const float* __restrict__ w, float* __restrict__ y,
                                                               float* __restrict__ pre, long rows, int cols, float eps, int mode) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  f32x4 v[NIT];
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = lane * 4 + 256 * it;
    if (c < cols) {
      v[it] = *(const f32x4*)(x + row * cols + c);
      if (res) v[it] += *(const f32x4*)(res + row * cols + c);
    }
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = lane * 4 + 256 * it;
    if (c < cols) {
      if (pre) *(f32x4*)(pre + row * cols + c) = v[it];
      s += (v[it][0] + v[it][1]) + (v[it][2] + v[it][3]);
      q += (v[it][0] * v[it][0] + v[it][1] * v[it][1]) + (v[it][2] * v[it][2] + v[it][3] * v[it][3]);
    }
  }
  float mean = 0.f, rstd;
  if (mode == 0) {
    rstd = rsqrtf(wave_sum(q) / (float)cols + eps);
  } else {
    mean = wave_sum(s) / (float)cols;
    float d2 = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it)
      if (lane * 4 + 256 * it < cols) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = v[it][j] - mean; d2 = fmaf(d, d, d2); }
      }
    rstd = 1.0f / sqrtf(wave_sum(d2) / (float)cols + eps);
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = lane * 4 + 256 * it;
    if (c < cols) {
      f32x4 g = {1.f, 1.f, 1.f, 1.f};
      if (w) g = *(const f32x4*)(w + c);
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (v[it][j] - mean) * rstd * g[j];
      *(f32x4*)(y + row * cols + c) = o;
    }
  }
}
extern "C" int muse_norm_res_fwd(const float* x, const float* res, const float* w, float* y, float* pre, int64_t rows,
                                 int32_t cols, float eps, int32_t mode, void* stream) {
  if (cols % 4 || (mode != 0 && mode != 1)) return MUSE_ERR_UNSUPPORTED;
  if (rows <= 0) return 0;
  if (cols <= 1024) {
    const dim3 grid((unsigned)((rows + 3) / 4));
#define NRF(N) hipLaunchKernelGGL(norm_res_fwd_reg_kernel<N>, grid, dim3(256), 0, (hipStream_t)stream, x, res, w, y, pre, (long)rows, cols, eps, mode)
    if (cols <= 256) NRF(1); else if (cols <= 512) NRF(2); else if (cols <= 768) NRF(3); else NRF(4);
#undef NRF
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL(norm_res_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, res, w, y, pre,
                     (long)rows, cols, eps, mode);
  return (int)hipGetLastError();
}

// =================================================================================================================
// AdaLN modulation (:1025-1037): y[b, r, c] = x[b, r, c] * (1 + ss[b, c]) + ss[b, C + c],  ss = mapper(silu(cond)) [B, 2C]
// =================================================================================================================
__global__ __launch_bounds__(256) void adaln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ ss,
                                                        float* __restrict__ y, bf16_t* __restrict__ yb, long rows_per_batch, int C, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const long e = i * 4, row = e / C;
    const int c = (int)(e - row * C);
    const long b = row / rows_per_batch;
    const f32x4 v = *(const f32x4*)(x + e);
    const f32x4 sc = *(const f32x4*)(ss + b * 2 * C + c), sh = *(const f32x4*)(ss + b * 2 * C + C + c);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = v[j] * (1.0f + sc[j]) + sh[j];
    if (y) *(f32x4*)(y + e) = o;
    if (yb) *(u32x2*)(yb + e) = u32x2{pack2_bf16(o[0], o[1]), pack2_bf16(o[2], o[3])};
  }
}
// y (f32) and / or y_bf16 (the GEMM operand of the bf16 compute mode, written here instead of by a separate cast) may be null
extern "C" int muse_adaln_fwd_ex(const float* x, const float* ss, float* y, void* y_bf16, int32_t batch, int64_t rows_per_batch,
                                 int32_t C, void* stream) {
  if (C % 4) return MUSE_ERR_UNSUPPORTED;
  const long n4 = (long)batch * rows_per_batch * C / 4;
  if (n4 <= 0) return 0;
  long g = (n4 + 255) / 256; if (g > 16384) g = 16384;
  hipLaunchKernelGGL(adaln_fwd_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, ss, y, (bf16_t*)y_bf16,
                     (long)rows_per_batch, C, n4);
  return (int)hipGetLastError();
}
extern "C" int muse_adaln_fwd(const float* x, const float* ss, float* y, int32_t batch, int64_t rows_per_batch, int32_t C, void* stream) {
  return muse_adaln_fwd_ex(x, ss, y, nullptr, batch, rows_per_batch, C, stream);
}

__global__ __launch_bounds__(256) void silu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float v = x[i];
    y[i] = v / (1.0f + expf(-v));
  }
}
extern "C" int muse_silu_fwd(const float* x, float* y, int64_t n, void* stream) {
  if (n <= 0) return 0;
  long g = (n + 255) / 256; if (g > 16384) g = 16384;
  hipLaunchKernelGGL(silu_fwd_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, y, (long)n);
  return (int)hipGetLastError();
}

// =================================================================================================================
// depthwise 3x3, padding 1 (ResBlock.depthwise :596-603), NHWC activations, weight [C][3][3] (= the reference's [C,1,3,3])
// =================================================================================================================
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        float* __restrict__ y, int H, int W, int C, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    long t = i / C;
    const int xx = (int)(t % W); t /= W;
    const int yy = (int)(t % H);
    const long b = t / H;
    float acc = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = yy + ky - 1;
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = xx + kx - 1;
        if (ix < 0 || ix >= W) continue;
        acc = fmaf(x[((b * H + iy) * W + ix) * C + c], w[c * 9 + ky * 3 + kx], acc);
      }
    }
    y[i] = acc;
  }
}
// C % 4 == 0: four channels per thread (16-byte accesses), one image row per block row, 32-bit index arithmetic (the scalar kernel
// above spends most of its time in 64-bit div / mod: 0.94 TB/s on the 16 x 16 x 1024 stage of the U-ViT).  FLIP = the backward's
// dx (correlation with the flipped taps).  Same tap order and fma chain per element as the scalar kernels -> the same bits.
template <bool FLIP>
__global__ __launch_bounds__(256) void dwconv3x3_v4_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                                           int H, int W, int C) {
  const int vpp = C >> 2;
  const int vc = blockIdx.x * 256 + threadIdx.x;            // channel vector (vpp >= 256) ...
  int cv, x0, xs;
  if (vpp >= 256) { cv = vc; x0 = 0; xs = 1; }
  else { cv = threadIdx.x % vpp; x0 = threadIdx.x / vpp; xs = 256 / vpp; }   // ... or several pixels of the row per pass
  if (cv >= vpp) return;
  // row = b * H + y.  Workgroups go to the 8 XCDs round-robin: consecutive image rows (which share two of their three input rows)
  // are handed to ONE XCD, so the shared rows come out of its L2 instead of being fetched by three of them
  const int wg = blockIdx.y * gridDim.x + blockIdx.x, nrow = gridDim.y;
  int row = blockIdx.y;
  if (gridDim.x == 1) {
    const int xcd = wg & 7, q = nrow >> 3, r = nrow & 7;
    row = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
  }
  const int yy = row % H;
  f32x4 wk[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int ts = FLIP ? 8 - t : t;
#pragma unroll
    for (int j = 0; j < 4; ++j) wk[t][j] = w[(cv * 4 + j) * 9 + ts];
  }
#pragma unroll 4      // (four pixels' loads in flight: the loop is otherwise one memory round trip per pixel)
  for (int xx = x0; xx < W; xx += xs) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = yy + ky - 1;
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = xx + kx - 1;
        if (ix < 0 || ix >= W) continue;
        const f32x4 v = *(const f32x4*)(x + ((long)(row + ky - 1) * W + ix) * C + cv * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = fmaf(v[j], wk[ky * 3 + kx][j], acc[j]);
      }
    }
    *(f32x4*)(y + ((long)row * W + xx) * C + cv * 4) = acc;
  }
}
template <bool FLIP>
static int launch_dwconv_v4(const float* x, const float* w, float* y, int batch, int H, int W, int C, hipStream_t s) {
  const int vpp = C >> 2;
  if ((C & 3) || (vpp < 256 && (256 % vpp)) || ((((uintptr_t)x) | ((uintptr_t)y)) & 15) || (long)batch * H > 65535) return 0;
  hipLaunchKernelGGL(dwconv3x3_v4_kernel<FLIP>, dim3(vpp >= 256 ? (vpp + 255) / 256 : 1, batch * H), dim3(256), 0, s, x, w, y, H, W, C);
  return 1;
}
// =================================================================================================================
// 2x2 space-to-depth / depth-to-space on channels-last rows: the data movement that turns the stride-2 2x2 convolution of
// DownsampleBlock (reference muse/modeling_transformer_v2.py:510-514) and the stride-2 2x2 transposed convolution of UpsampleBlock
// (:558-562) into plain products on the GEMM kernels.  full [B, H, W, C] <-> packed [B, H/2, W/2, (di, dj, c)], 16 bytes per thread.
// =================================================================================================================
template <bool INVERSE>
__global__ __launch_bounds__(256) void space_depth2_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C4,
                                                           long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C4);
    long r = i / C4;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H);
    const long b = r / H;
    const long packed = (((b * (H >> 1) + (h >> 1)) * (W >> 1) + (w >> 1)) * 4 + ((h & 1) * 2 + (w & 1))) * C4 + c;
    if (INVERSE) ((f32x4*)y)[i] = ((const f32x4*)x)[packed];
    else ((f32x4*)y)[packed] = ((const f32x4*)x)[i];
  }
}
extern "C" int muse_space_to_depth2_nhwc(const float* x, float* y, int32_t batch, int32_t H, int32_t W, int32_t C, int32_t inverse,
                                         void* stream) {
  if (batch < 0 || H <= 0 || W <= 0 || C <= 0) return MUSE_ERR_BAD_ARG;
  if ((H | W) & 1 || (C & 3) || ((((uintptr_t)x) | ((uintptr_t)y)) & 15)) return MUSE_ERR_UNSUPPORTED;
  const long n4 = (long)batch * H * W * (C >> 2);
  if (n4 == 0) return 0;
  long g = (n4 + 255) / 256; if (g > 16384) g = 16384;
  if (inverse) hipLaunchKernelGGL(space_depth2_kernel<true>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, y, H, W, C >> 2, n4);
  else hipLaunchKernelGGL(space_depth2_kernel<false>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, y, H, W, C >> 2, n4);
  return (int)hipGetLastError();
}

extern "C" int muse_dwconv3x3_nhwc(const float* x, const float* w, float* y, int32_t batch, int32_t H, int32_t W, int32_t C,
                                   void* stream) {
  const long n = (long)batch * H * W * C;
  if (n <= 0) return 0;
  if (launch_dwconv_v4<false>(x, w, y, batch, H, W, C, (hipStream_t)stream)) return (int)hipGetLastError();
  long g = (n + 255) / 256; if (g > 65535) g = 65535;
  hipLaunchKernelGGL(dwconv3x3_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, w, y, H, W, C, n);
  return (int)hipGetLastError();
}

// =================================================================================================================
// GlobalResponseNorm (:741-751) on [B, S, C] (S = H*W pixels):  Gx[b,c] = ||x[b,:,c]||_2 ;  Nx = Gx / (mean_c Gx + 1e-6) ;
// y = gamma * (x * Nx) + beta + x.   scratch: 2*B*C floats, on return [G | N] (the backward's `stats`).
// =================================================================================================================
__global__ __launch_bounds__(256) void grn_colnorm_kernel(const float* __restrict__ x, float* __restrict__ gx, long S, int C) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const long b = blockIdx.y;
  float s = 0.f;
  if (c < C)
    for (long r = wv; r < S; r += 4) { const float v = x[(b * S + r) * C + c]; s = fmaf(v, v, s); }
  red[wv][lane] = s;
  __syncthreads();
  if (wv == 0 && c < C) gx[b * C + c] = sqrtf((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]));
}
__global__ __launch_bounds__(256) void grn_scale_kernel(const float* __restrict__ gx, float* __restrict__ nx, int C) {   // N = G / (mean_c G + 1e-6), one block per image
  __shared__ float red[4];
  const long b = blockIdx.x;
  float s = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) s += gx[b * C + c];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / (float)C;
  for (int c = threadIdx.x; c < C; c += 256) nx[b * C + c] = gx[b * C + c] / (mean + 1e-6f);
}
__global__ __launch_bounds__(256) void grn_apply_kernel(const float* __restrict__ x, const float* __restrict__ nx,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ y, long S, int C, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long b = i / ((long)S * C);
    const float v = x[i];
    y[i] = gamma[c] * (v * nx[b * C + c]) + beta[c] + v;
  }
}
// C % 4 == 0: four channels per thread, the image in blockIdx.y (no 64-bit index division per element), f32 and / or bf16 output -
// the bf16 copy is the next GEMM's operand in the bf16 compute mode, written here instead of by a cast pass over the f32 result
__global__ __launch_bounds__(256) void grn_apply4_kernel(const float* __restrict__ x, const float* __restrict__ nx,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* __restrict__ y, bf16_t* __restrict__ yb, int S, int C) {
  const int vpp = C >> 2, b = blockIdx.y;
  const long base = (long)b * S * C;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < S * vpp; i += gridDim.x * 256) {
    const int c = (i % vpp) * 4;
    const long e = base + (long)(i / vpp) * C + c;
    const f32x4 v = *(const f32x4*)(x + e), nn = *(const f32x4*)(nx + (long)b * C + c);
    const f32x4 ga = *(const f32x4*)(gamma + c), be = *(const f32x4*)(beta + c);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = ga[j] * (v[j] * nn[j]) + be[j] + v[j];
    if (y) *(f32x4*)(y + e) = o;
    if (yb) *(u32x2*)(yb + e) = u32x2{pack2_bf16(o[0], o[1]), pack2_bf16(o[2], o[3])};
  }
}
extern "C" int muse_grn_fwd_ex(const float* x, const float* gamma, const float* beta, float* y, void* y_bf16, float* scratch,
                               int32_t batch, int64_t S, int32_t C, void* stream) {
  const long n = (long)batch * S * C;
  if (n <= 0) return 0;
  if (!y && !y_bf16) return MUSE_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  float* nx = scratch + (long)batch * C;   // scratch = [G | N], both kept for the backward
  hipLaunchKernelGGL(grn_colnorm_kernel, dim3((C + 63) / 64, batch), dim3(256), 0, s, x, scratch, (long)S, C);
  hipLaunchKernelGGL(grn_scale_kernel, dim3(batch), dim3(256), 0, s, (const float*)scratch, nx, C);
  if ((C & 3) == 0 && S * (long)(C >> 2) < (1L << 31) && batch <= 65535 && !((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)y_bf16)) & 15)) {
    long gx = (S * (long)(C >> 2) + 255) / 256; if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(grn_apply4_kernel, dim3((unsigned)gx, batch), dim3(256), 0, s, x, (const float*)nx, gamma, beta, y, (bf16_t*)y_bf16,
                       (int)S, C);
    return (int)hipGetLastError();
  }
  if (y_bf16) return MUSE_ERR_UNSUPPORTED;
  long g = (n + 255) / 256; if (g > 65535) g = 65535;
  hipLaunchKernelGGL(grn_apply_kernel, dim3((unsigned)g), dim3(256), 0, s, x, (const float*)nx, gamma, beta, y, (long)S, C, n);
  return (int)hipGetLastError();
}
extern "C" int muse_grn_fwd(const float* x, const float* gamma, const float* beta, float* y, float* scratch, int32_t batch,
                            int64_t S, int32_t C, void* stream) {
  return muse_grn_fwd_ex(x, gamma, beta, y, nullptr, scratch, batch, S, C, stream);
}

// =================================================================================================================
// sinusoidal_encode (:59-76): out[i, j] = cos(f[i] * w_j), out[i, half + j] = sin(f[i] * w_j), w_j = exp(-ln(maxpos)/half * j),
// zero padded when dim is odd
// =================================================================================================================
__global__ __launch_bounds__(256) void sinusoid_kernel(const float* __restrict__ f, float* __restrict__ out, long n, int dim,
                                                       float neg_log_over_half) {
  const int half = dim >> 1;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n * dim; i += (long)gridDim.x * 256) {
    const long r = i / dim;
    const int j = (int)(i - r * dim);
    float v = 0.f;
    if (j < 2 * half) {
      const int k = j < half ? j : j - half;
      const float ang = f[r] * expf((float)k * neg_log_over_half);
      v = j < half ? cosf(ang) : sinf(ang);
    }
    out[i] = v;
  }
}
extern "C" int muse_sinusoidal_encode(const float* f, float* out, int64_t n, int32_t dim, float max_positions, void* stream) {
  if (n <= 0 || dim <= 1) return dim <= 1 ? MUSE_ERR_BAD_ARG : 0;
  const float nl = -(float)(log((double)max_positions) / (double)(dim >> 1));
  long g = (n * dim + 255) / 256; if (g > 4096) g = 4096;
  hipLaunchKernelGGL(sinusoid_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, f, out, (long)n, dim, nl);
  return (int)hipGetLastError();
}

// out[0] = sum(v * w) / sum(w)   (loss weighting :311-316; one block, fixed order)
__global__ __launch_bounds__(1024) void weighted_mean_kernel(const float* __restrict__ v, const float* __restrict__ w,
                                                             float* __restrict__ out, long n) {
  __shared__ double rs[16], rw[16];
  double s = 0.0, t = 0.0;
  for (long i = threadIdx.x; i < n; i += 1024) { s += (double)v[i] * (double)w[i]; t += (double)w[i]; }
  s = wave_sum_d(s); t = wave_sum_d(t);
  if ((threadIdx.x & 63) == 0) { rs[threadIdx.x >> 6] = s; rw[threadIdx.x >> 6] = t; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0;
    for (int k = 0; k < 16; ++k) { a += rs[k]; b += rw[k]; }
    out[0] = (float)(a / b);
  }
}
extern "C" int muse_weighted_mean(const float* v, const float* w, float* out, int64_t n, void* stream) {
  if (n <= 0) return MUSE_ERR_BAD_ARG;
  hipLaunchKernelGGL(weighted_mean_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, v, w, out, (long)n);
  return (int)hipGetLastError();
}

// =================================================================================================================
// ============================================  backward  =========================================================
// =================================================================================================================

// norm with residual stream, backward.  v = x (+ res) was saved by the forward (`pre`, or x itself when there was no residual).
//   RMS : xhat = v r            dv = r (g - xhat mean(g xhat))                  g = dy w
//   LN  : xhat = (v - mu) r     dv = r (g - mean(g) - xhat mean(g xhat))
//   dv += dpre (gradient that reached the pre-norm residual output);  dx = dres = dv;  dw_partial[block] = sum_rows dy xhat
// 16 rows per block (4 per wave); per-column dw partials live in registers, folded across the 4 waves through LDS in a fixed order.
#define NRB_ROWS 16
extern "C" int muse_norm_res_bwd_nblk(int64_t rows) { return (int)((rows + NRB_ROWS - 1) / NRB_ROWS); }

template <int NIT>
__global__ __launch_bounds__(256) void norm_res_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ dpre,
                                                           const float* __restrict__ v, const float* __restrict__ w,
                                                           float* __restrict__ dv, bf16_t* __restrict__ dvb, float* __restrict__ dwp,
                                                           long rows, int cols, float eps, int mode) {
  __shared__ float red[4096];   // cols <= 4096
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float dwacc[NIT][4];          // a wave covers 256 columns per step: NIT = ceil(cols / 256) rounded up to a power of two
#pragma unroll
  for (int it = 0; it < NIT; ++it)
#pragma unroll
    for (int j = 0; j < 4; ++j) dwacc[it][j] = 0.f;
  if constexpr (NIT <= 4) {
    // the row (v, dy, dpre) lives in registers: ONE round trip to memory per row instead of five dependent ones (the kernel is
    // latency-bound: every block of the grid is resident at once and each wave walks its four rows serially); the next row's
    // loads are issued before this row's reductions.  Same expressions and summation orders as the general path below.
    f32x4 wv4[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = lane * 4 + 256 * it;
      wv4[it] = (w && c < cols) ? *(const f32x4*)(w + c) : f32x4{1.f, 1.f, 1.f, 1.f};
    }
    const long row0 = (long)blockIdx.x * NRB_ROWS + wave * (NRB_ROWS / 4);
    f32x4 tn[NIT], dn[NIT], pn[NIT];
    auto fetch = [&](long row) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int c = lane * 4 + 256 * it;
        if (row < rows && c < cols) {
          tn[it] = *(const f32x4*)(v + row * cols + c);
          dn[it] = *(const f32x4*)(dy + row * cols + c);
          if (dpre) pn[it] = *(const f32x4*)(dpre + row * cols + c);
        }
      }
    };
    fetch(row0);
    for (int rr = 0; rr < NRB_ROWS / 4; ++rr) {
      const long row = row0 + rr;
      if (row >= rows) break;
      f32x4 t[NIT], d[NIT], pr[NIT];
#pragma unroll
      for (int it = 0; it < NIT; ++it) { t[it] = tn[it]; d[it] = dn[it]; pr[it] = pn[it]; }
      if (rr + 1 < NRB_ROWS / 4) fetch(row + 1);
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int it = 0; it < NIT; ++it)
        if (lane * 4 + 256 * it < cols) {
          s += (t[it][0] + t[it][1]) + (t[it][2] + t[it][3]);
          q += (t[it][0] * t[it][0] + t[it][1] * t[it][1]) + (t[it][2] * t[it][2] + t[it][3] * t[it][3]);
        }
      float mean = 0.f, rstd;
      if (mode == 0) {
        rstd = rsqrtf(wave_sum(q) / (float)cols + eps);
      } else {
        mean = wave_sum(s) / (float)cols;
        float d2 = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it)
          if (lane * 4 + 256 * it < cols) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float dd = t[it][j] - mean; d2 = fmaf(dd, dd, d2); }
          }
        rstd = 1.0f / sqrtf(wave_sum(d2) / (float)cols + eps);
      }
      float sg = 0.f, sgx = 0.f;
#pragma unroll
      for (int it = 0; it < NIT; ++it)
        if (lane * 4 + 256 * it < cols) {
          f32x4 g = d[it];
          if (w) g *= wv4[it];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float xh = (t[it][j] - mean) * rstd;
            sg += g[j];
            sgx = fmaf(g[j], xh, sgx);
            dwacc[it][j] = fmaf(d[it][j], xh, dwacc[it][j]);
          }
        }
      const float mg = mode == 0 ? 0.f : wave_sum(sg) / (float)cols;
      const float mgx = wave_sum(sgx) / (float)cols;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int c = lane * 4 + 256 * it;
        if (c < cols) {
          f32x4 g = d[it];
          if (w) g *= wv4[it];
          f32x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = rstd * (g[j] - mg - (t[it][j] - mean) * rstd * mgx);
          if (dpre) o += pr[it];
          *(f32x4*)(dv + row * cols + c) = o;
          if (dvb) *(u32x2*)(dvb + row * cols + c) = u32x2{pack2_bf16(o[0], o[1]), pack2_bf16(o[2], o[3])};
        }
      }
    }
  } else
  for (int rr = 0; rr < NRB_ROWS / 4; ++rr) {
    const long row = (long)blockIdx.x * NRB_ROWS + wave * (NRB_ROWS / 4) + rr;
    if (row >= rows) break;
    const float* vr = v + row * cols;
    const float* gr = dy + row * cols;
    float s = 0.f, q = 0.f;
    for (int c = lane * 4; c < cols; c += 256) {
      const f32x4 t = *(const f32x4*)(vr + c);
      s += (t[0] + t[1]) + (t[2] + t[3]);
      q += (t[0] * t[0] + t[1] * t[1]) + (t[2] * t[2] + t[3] * t[3]);
    }
    float mean = 0.f, rstd;
    if (mode == 0) {
      rstd = rsqrtf(wave_sum(q) / (float)cols + eps);
    } else {
      mean = wave_sum(s) / (float)cols;
      float d2 = 0.f;
      for (int c = lane * 4; c < cols; c += 256) {
        const f32x4 t = *(const f32x4*)(vr + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = t[j] - mean; d2 = fmaf(d, d, d2); }
      }
      rstd = 1.0f / sqrtf(wave_sum(d2) / (float)cols + eps);
    }
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = lane * 4 + 256 * it;
      if (c < cols) {
        const f32x4 t = *(const f32x4*)(vr + c), d = *(const f32x4*)(gr + c);
        f32x4 g = d;
        if (w) g *= *(const f32x4*)(w + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float xh = (t[j] - mean) * rstd;
          sg += g[j];
          sgx = fmaf(g[j], xh, sgx);
          dwacc[it][j] = fmaf(d[j], xh, dwacc[it][j]);
        }
      }
    }
    const float mg = mode == 0 ? 0.f : wave_sum(sg) / (float)cols;
    const float mgx = wave_sum(sgx) / (float)cols;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = lane * 4 + 256 * it;
      if (c < cols) {
        const f32x4 t = *(const f32x4*)(vr + c);
        f32x4 g = *(const f32x4*)(gr + c);
        if (w) g *= *(const f32x4*)(w + c);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = rstd * (g[j] - mg - (t[j] - mean) * rstd * mgx);
        if (dpre) o += *(const f32x4*)(dpre + row * cols + c);
        *(f32x4*)(dv + row * cols + c) = o;
        if (dvb) *(u32x2*)(dvb + row * cols + c) = u32x2{pack2_bf16(o[0], o[1]), pack2_bf16(o[2], o[3])};
      }
    }
  }
  // fold the four waves' column partials in wave order
  for (int wv = 0; wv < 4; ++wv) {
    if (wave == wv) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int c = lane * 4 + 256 * it;
        if (c < cols) {
#pragma unroll
          for (int j = 0; j < 4; ++j) red[c + j] = wv == 0 ? dwacc[it][j] : red[c + j] + dwacc[it][j];
        }
      }
    }
    __syncthreads();
  }
  for (int c = threadIdx.x; c < cols; c += 256) dwp[(long)blockIdx.x * cols + c] = red[c];
}
// dv_bf16 (optional): a bf16 copy of dv written in the same pass - dv is the next GEMM's operand in the bf16 compute mode
extern "C" int muse_norm_res_bwd_ex(const float* dy, const float* dpre, const float* v, const float* w, float* dv, void* dv_bf16,
                                    float* dw_partial, int64_t rows, int32_t cols, float eps, int32_t mode, void* stream) {
  if (cols % 4 || cols > 4096 || (mode != 0 && mode != 1)) return MUSE_ERR_UNSUPPORTED;
  if (rows <= 0) return 0;
  const int nblk = muse_norm_res_bwd_nblk(rows), nit = (cols + 255) / 256;
  hipStream_t s = (hipStream_t)stream;
#define NRB(N) hipLaunchKernelGGL(norm_res_bwd_kernel<N>, dim3(nblk), dim3(256), 0, s, dy, dpre, v, w, dv, (bf16_t*)dv_bf16, dw_partial, (long)rows, cols, eps, mode)
  if (nit <= 1) NRB(1); else if (nit <= 2) NRB(2); else if (nit <= 4) NRB(4); else if (nit <= 8) NRB(8); else NRB(16);
#undef NRB
  return (int)hipGetLastError();
}
extern "C" int muse_norm_res_bwd(const float* dy, const float* dpre, const float* v, const float* w, float* dv, float* dw_partial,
                                 int64_t rows, int32_t cols, float eps, int32_t mode, void* stream) {
  return muse_norm_res_bwd_ex(dy, dpre, v, w, dv, nullptr, dw_partial, rows, cols, eps, mode, stream);
}

// =================================================================================================================
// Norm + AdaLN fused (TransformerLayer :757-792: every norm of a layer is followed by an AdaLNModulation of its output).
//   fwd: v = x (+ res); pre = v; n = Norm(v) * w; m[b, r, :] = n * (1 + ss[b, :C]) + ss[b, C:]  - n is never written
//        (22 -> 14 bytes per element against muse_norm_res_fwd + muse_adaln_fwd_ex).
//   bwd: n is recomputed from v; dn = dm (1 + scale); per-block column sums of dm * n and dm (-> d(scale | shift) of the block's
//        image), then the norm backward of muse_norm_res_bwd on dn - one pass instead of adaln_bwd_dx + adaln_bwd_ss + norm_res_bwd
//        (~34 -> 18 bytes per element).  16 rows per block, all of one image (rows_per_batch % 16 == 0); cols <= 1024.
// =================================================================================================================
// the bf16 copy of four results (the operand of the next weight GEMMs in the bf16 mode); lo_off != 0 ("bf16x3" mode): that copy is the hi
// plane and bf16(o - hi) goes lo_off elements behind it - the (hi, lo) operand planes of muse_gemm_x3, bit for bit what
// muse_split_f32_to_bf16x2 makes of the f32 result
// (lo_off < 0, "f16" mode: ONE IEEE-half image half(o * scale) - common.h store_image4)
__device__ __forceinline__ void store_hi_lo(bf16_t* hi, long lo_off, long idx, const f32x4& o, float scale = 1.f, int* stats = nullptr) {
  if (lo_off) {
    store_image4(hi + idx, lo_off, scale, stats, o[0], o[1], o[2], o[3]);
  } else {
    *(u32x2*)(hi + idx) = u32x2{pack2_bf16(o[0], o[1]), pack2_bf16(o[2], o[3])};
  }
}
template <int NIT>
__global__ __launch_bounds__(256) void norm_adaln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                             const float* __restrict__ w, const float* __restrict__ ss,
                                                             float* __restrict__ pre, float* __restrict__ m, bf16_t* __restrict__ mb,
                                                             long rows, long rpb, int cols, float eps, int mode, long lo_off, float img_scale,
                                                             int* img_stats) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* sb = ss + (row / rpb) * 2 * cols;
  f32x4 v[NIT];
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = lane * 4 + 256 * it;
    if (c < cols) {
      v[it] = *(const f32x4*)(x + row * cols + c);
      if (res) v[it] += *(const f32x4*)(res + row * cols + c);
    }
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = lane * 4 + 256 * it;
    if (c < cols) {
      if (pre) *(f32x4*)(pre + row * cols + c) = v[it];
      s += (v[it][0] + v[it][1]) + (v[it][2] + v[it][3]);
      q += (v[it][0] * v[it][0] + v[it][1] * v[it][1]) + (v[it][2] * v[it][2] + v[it][3] * v[it][3]);
    }
  }
  float mean = 0.f, rstd;
  if (mode == 0) {
    rstd = rsqrtf(wave_sum(q) / (float)cols + eps);
  } else {
    mean = wave_sum(s) / (float)cols;
    float d2 = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it)
      if (lane * 4 + 256 * it < cols) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = v[it][j] - mean; d2 = fmaf(d, d, d2); }
      }
    rstd = 1.0f / sqrtf(wave_sum(d2) / (float)cols + eps);
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = lane * 4 + 256 * it;
    if (c < cols) {
      f32x4 g = {1.f, 1.f, 1.f, 1.f};
      if (w) g = *(const f32x4*)(w + c);
      const f32x4 sc = *(const f32x4*)(sb + c), sh = *(const f32x4*)(sb + cols + c);
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float n = (v[it][j] - mean) * rstd * g[j];
        o[j] = n * (1.0f + sc[j]) + sh[j];
      }
      if (m) *(f32x4*)(m + row * cols + c) = o;
      if (mb) store_hi_lo(mb, lo_off, row * cols + c, o, img_scale, img_stats);
    }
  }
}
static int norm_adaln_fwd_launch(const float* x, const float* res, const float* w, const float* ss, float* pre, float* m, void* m_bf16, long lo_off,
                                 float img_scale, int* img_stats, int32_t batch, int64_t rows_per_batch, int32_t cols, float eps, int32_t mode, void* stream) {
  if (cols % 4 || cols > 1024 || (mode != 0 && mode != 1) || (!m && !m_bf16)) return MUSE_ERR_UNSUPPORTED;
  const long rows = (long)batch * rows_per_batch;
  if (rows <= 0) return 0;
  const dim3 grid((unsigned)((rows + 3) / 4));
#define NAF(N) hipLaunchKernelGGL(norm_adaln_fwd_kernel<N>, grid, dim3(256), 0, (hipStream_t)stream, x, res, w, ss, pre, m, (bf16_t*)m_bf16, rows, \
                                  (long)rows_per_batch, cols, eps, mode, lo_off, img_scale, img_stats)
  if (cols <= 256) NAF(1); else if (cols <= 512) NAF(2); else if (cols <= 768) NAF(3); else NAF(4);
#undef NAF
  return (int)hipGetLastError();
}
extern "C" int muse_norm_adaln_fwd(const float* x, const float* res, const float* w, const float* ss, float* pre, float* m, void* m_bf16,
                                   int32_t batch, int64_t rows_per_batch, int32_t cols, float eps, int32_t mode, void* stream) {
  return norm_adaln_fwd_launch(x, res, w, ss, pre, m, m_bf16, 0, 1.f, nullptr, batch, rows_per_batch, cols, eps, mode, stream);
}
// "bf16x3" mode: m as f32 AND as the (hi, lo) operand planes [2][rows][cols] of the products that read it
extern "C" int muse_norm_adaln_fwd_x3(const float* x, const float* res, const float* w, const float* ss, float* pre, float* m, void* planes,
                                      int32_t batch, int64_t rows_per_batch, int32_t cols, float eps, int32_t mode, void* stream) {
  if (!planes) return MUSE_ERR_BAD_ARG;      // (m may be NULL: the planes only - a result that nothing but weight GEMMs reads)
  const ImgFormat f = img_format(false);     // ("f16" mode: planes receives ONE half image [rows][cols] instead)
  return norm_adaln_fwd_launch(x, res, w, ss, pre, m, planes, f.lo_sign * (long)batch * rows_per_batch * cols, f.scale, f.stats, batch, rows_per_batch, cols, eps,
                               mode, stream);
}

template <int NIT>
__global__ __launch_bounds__(256) void norm_adaln_bwd_kernel(const float* __restrict__ dm, const float* __restrict__ dpre,
                                                             const float* __restrict__ v, const float* __restrict__ w,
                                                             const float* __restrict__ ss, float* __restrict__ dv, bf16_t* __restrict__ dvb,
                                                             float* __restrict__ dwp, float* __restrict__ dssp, long rows, long rpb,
                                                             int cols, float eps, int mode, long lo_off, float img_scale, int* img_stats) {
  __shared__ float red[1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* sb = ss + (((long)blockIdx.x * NRB_ROWS) / rpb) * 2 * cols;       // the block's image
  f32x4 wv4[NIT], sc1[NIT];
  float dwacc[NIT][4], dsc[NIT][4], dsh[NIT][4];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = lane * 4 + 256 * it;
    wv4[it] = (w && c < cols) ? *(const f32x4*)(w + c) : f32x4{1.f, 1.f, 1.f, 1.f};
    sc1[it] = f32x4{1.f, 1.f, 1.f, 1.f};
    if (c < cols) sc1[it] += *(const f32x4*)(sb + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) { dwacc[it][j] = 0.f; dsc[it][j] = 0.f; dsh[it][j] = 0.f; }
  }
  const long row0 = (long)blockIdx.x * NRB_ROWS + wave * (NRB_ROWS / 4);
  f32x4 tn[NIT], dn_[NIT], pn[NIT];
  auto fetch = [&](long row) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = lane * 4 + 256 * it;
      if (row < rows && c < cols) {
        tn[it] = *(const f32x4*)(v + row * cols + c);
        dn_[it] = *(const f32x4*)(dm + row * cols + c);
        if (dpre) pn[it] = *(const f32x4*)(dpre + row * cols + c);
      }
    }
  };
  fetch(row0);
  for (int rr = 0; rr < NRB_ROWS / 4; ++rr) {
    const long row = row0 + rr;
    if (row >= rows) break;
    f32x4 t[NIT], d[NIT], pr[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) { t[it] = tn[it]; d[it] = dn_[it]; pr[it] = pn[it]; }
    if (rr + 1 < NRB_ROWS / 4) fetch(row + 1);
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it)
      if (lane * 4 + 256 * it < cols) {
        s += (t[it][0] + t[it][1]) + (t[it][2] + t[it][3]);
        q += (t[it][0] * t[it][0] + t[it][1] * t[it][1]) + (t[it][2] * t[it][2] + t[it][3] * t[it][3]);
      }
    float mean = 0.f, rstd;
    if (mode == 0) {
      rstd = rsqrtf(wave_sum(q) / (float)cols + eps);
    } else {
      mean = wave_sum(s) / (float)cols;
      float d2 = 0.f;
#pragma unroll
      for (int it = 0; it < NIT; ++it)
        if (lane * 4 + 256 * it < cols) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { const float dd = t[it][j] - mean; d2 = fmaf(dd, dd, d2); }
        }
      rstd = 1.0f / sqrtf(wave_sum(d2) / (float)cols + eps);
    }
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it)
      if (lane * 4 + 256 * it < cols) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float xh = (t[it][j] - mean) * rstd;
          const float n = xh * wv4[it][j];                       // == the forward's (v - mean) * rstd * w
          dsc[it][j] = fmaf(d[it][j], n, dsc[it][j]);
          dsh[it][j] += d[it][j];
          const float dn = d[it][j] * sc1[it][j];                // d(norm output)
          d[it][j] = dn;
          const float g = dn * wv4[it][j];
          sg += g;
          sgx = fmaf(g, xh, sgx);
          dwacc[it][j] = fmaf(dn, xh, dwacc[it][j]);
        }
      }
    const float mg = mode == 0 ? 0.f : wave_sum(sg) / (float)cols;
    const float mgx = wave_sum(sgx) / (float)cols;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = lane * 4 + 256 * it;
      if (c < cols) {
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = rstd * (d[it][j] * wv4[it][j] - mg - (t[it][j] - mean) * rstd * mgx);
        if (dpre) o += pr[it];
        *(f32x4*)(dv + row * cols + c) = o;
        if (dvb) store_hi_lo(dvb, lo_off, row * cols + c, o, img_scale, img_stats);
      }
    }
  }
  // fold the four waves' column partials in wave order: dw, then d(scale), then d(shift)
#pragma unroll
  for (int which = 0; which < 3; ++which) {
    for (int wv = 0; wv < 4; ++wv) {
      if (wave == wv) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int c = lane * 4 + 256 * it;
          if (c < cols) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float a = which == 0 ? dwacc[it][j] : (which == 1 ? dsc[it][j] : dsh[it][j]);
              red[c + j] = wv == 0 ? a : red[c + j] + a;
            }
          }
        }
      }
      __syncthreads();
    }
    float* dst = which == 0 ? dwp + (long)blockIdx.x * cols : dssp + (long)blockIdx.x * 2 * cols + (which == 2 ? cols : 0);
    for (int c = threadIdx.x; c < cols; c += 256) dst[c] = red[c];
    __syncthreads();
  }
}
static int norm_adaln_bwd_launch(const float* dm, const float* dpre, const float* v, const float* w, const float* ss, float* dv,
                                 void* dv_bf16, long lo_off, float img_scale, int* img_stats, float* dw_partial, float* dss_partial, int32_t batch, int64_t rows_per_batch,
                                 int32_t cols, float eps, int32_t mode, void* stream);
extern "C" int muse_norm_adaln_bwd(const float* dm, const float* dpre, const float* v, const float* w, const float* ss, float* dv,
                                   void* dv_bf16, float* dw_partial, float* dss_partial, int32_t batch, int64_t rows_per_batch,
                                   int32_t cols, float eps, int32_t mode, void* stream) {
  return norm_adaln_bwd_launch(dm, dpre, v, w, ss, dv, dv_bf16, 0, 1.f, nullptr, dw_partial, dss_partial, batch, rows_per_batch, cols, eps, mode, stream);
}
// "bf16x3" mode: dv as f32 AND as the (hi, lo) operand planes [2][rows][cols] of the dX / dW products that read it
extern "C" int muse_norm_adaln_bwd_x3(const float* dm, const float* dpre, const float* v, const float* w, const float* ss, float* dv,
                                      void* planes, float* dw_partial, float* dss_partial, int32_t batch, int64_t rows_per_batch,
                                      int32_t cols, float eps, int32_t mode, void* stream) {
  if (!planes) return MUSE_ERR_BAD_ARG;
  const ImgFormat f = img_format(true);      // ("f16" mode: ONE half image of dv, scaled by the backward pass's gradient scale)
  return norm_adaln_bwd_launch(dm, dpre, v, w, ss, dv, planes, f.lo_sign * (long)batch * rows_per_batch * cols, f.scale, f.stats, dw_partial, dss_partial, batch, rows_per_batch,
                               cols, eps, mode, stream);
}
static int norm_adaln_bwd_launch(const float* dm, const float* dpre, const float* v, const float* w, const float* ss, float* dv,
                                 void* dv_bf16, long lo_off, float img_scale, int* img_stats, float* dw_partial, float* dss_partial, int32_t batch, int64_t rows_per_batch,
                                 int32_t cols, float eps, int32_t mode, void* stream) {
  if (cols % 4 || cols > 1024 || (mode != 0 && mode != 1) || (rows_per_batch % NRB_ROWS)) return MUSE_ERR_UNSUPPORTED;
  const long rows = (long)batch * rows_per_batch;
  if (rows <= 0) return 0;
  const int nblk = muse_norm_res_bwd_nblk(rows);
#define NAB(N) hipLaunchKernelGGL(norm_adaln_bwd_kernel<N>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, dm, dpre, v, w, ss, dv, (bf16_t*)dv_bf16, \
                                  dw_partial, dss_partial, rows, (long)rows_per_batch, cols, eps, mode, lo_off, img_scale, img_stats)
  if (cols <= 256) NAB(1); else if (cols <= 512) NAB(2); else if (cols <= 768) NAB(3); else NAB(4);
#undef NAB
  return (int)hipGetLastError();
}
// out[s, c] = sum_{k < seg_rows} part[s * seg_rows + k, c]   (fixed order): the per-image fold of norm_adaln_bwd's d(scale | shift) partials
__global__ __launch_bounds__(256) void colsum_segments_kernel(const float* __restrict__ part, float* __restrict__ out, int seg_rows, int cols) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  const float* p = part + (long)blockIdx.y * seg_rows * cols + c;
  float s = 0.f;
  for (int k = 0; k < seg_rows; ++k) s += p[(long)k * cols];
  out[(long)blockIdx.y * cols + c] = s;
}
extern "C" int muse_colsum_segments(const float* part, float* out, int32_t nseg, int32_t seg_rows, int32_t cols, void* stream) {
  if (nseg <= 0 || cols <= 0) return 0;
  if (nseg > 65535) return MUSE_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(colsum_segments_kernel, dim3((cols + 255) / 256, nseg), dim3(256), 0, (hipStream_t)stream, part, out, seg_rows, cols);
  return (int)hipGetLastError();
}

// AdaLN backward: dx = dy (1 + scale[b]);  dss[b, c] = sum_r dy x,  dss[b, C + c] = sum_r dy   (r over the rows of image b)
__global__ __launch_bounds__(256) void adaln_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ ss,
                                                           float* __restrict__ dx, long rows_per_batch, int C, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const long e = i * 4, row = e / C;
    const int c = (int)(e - row * C);
    const long b = row / rows_per_batch;
    const f32x4 g = *(const f32x4*)(dy + e), sc = *(const f32x4*)(ss + b * 2 * C + c);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = g[j] * (1.0f + sc[j]);
    *(f32x4*)(dx + e) = o;
  }
}
__global__ __launch_bounds__(256) void adaln_bwd_ss_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           float* __restrict__ dss, long S, int C) {
  __shared__ float r0[4][64], r1[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const long b = blockIdx.y;
  float a = 0.f, s = 0.f;
  if (c < C)
    for (long r = wv; r < S; r += 4) { const float g = dy[(b * S + r) * C + c]; a = fmaf(g, x[(b * S + r) * C + c], a); s += g; }
  r0[wv][lane] = a; r1[wv][lane] = s;
  __syncthreads();
  if (wv == 0 && c < C) {
    dss[b * 2 * C + c] = (r0[0][lane] + r0[1][lane]) + (r0[2][lane] + r0[3][lane]);
    dss[b * 2 * C + C + c] = (r1[0][lane] + r1[1][lane]) + (r1[2][lane] + r1[3][lane]);
  }
}
extern "C" int muse_adaln_bwd(const float* dy, const float* x, const float* ss, float* dx, float* dss, int32_t batch,
                              int64_t rows_per_batch, int32_t C, void* stream) {
  if (C % 4) return MUSE_ERR_UNSUPPORTED;
  const long n4 = (long)batch * rows_per_batch * C / 4;
  if (n4 <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  long g = (n4 + 255) / 256; if (g > 16384) g = 16384;
  hipLaunchKernelGGL(adaln_bwd_dx_kernel, dim3((unsigned)g), dim3(256), 0, s, dy, ss, dx, (long)rows_per_batch, C, n4);
  hipLaunchKernelGGL(adaln_bwd_ss_kernel, dim3((C + 63) / 64, batch), dim3(256), 0, s, dy, x, dss, (long)rows_per_batch, C);
  return (int)hipGetLastError();
}

__global__ __launch_bounds__(256) void silu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                       float* __restrict__ dx, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float v = x[i], sg = 1.0f / (1.0f + expf(-v));
    dx[i] = dy[i] * sg * (1.0f + v * (1.0f - sg));
  }
}
extern "C" int muse_silu_bwd(const float* x, const float* dy, float* dx, int64_t n, void* stream) {
  if (n <= 0) return 0;
  long g = (n + 255) / 256; if (g > 16384) g = 16384;
  hipLaunchKernelGGL(silu_bwd_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, dy, dx, (long)n);
  return (int)hipGetLastError();
}

// depthwise 3x3 backward: dx = correlation of dy with the flipped taps;  dw[c][t] = sum_pixels dy[p] x[p + tap t]
// dw partials: one block per 256 pixels -> dwp[chunk][c * 9 + t], folded by muse_colsum
#define DW_PIX 64
extern "C" int muse_dwconv3x3_bwd_nchunk(int64_t pixels) { return (int)((pixels + DW_PIX - 1) / DW_PIX); }
__global__ __launch_bounds__(256) void dwconv3x3_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                               float* __restrict__ dx, int H, int W, int C, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    long t = i / C;
    const int xx = (int)(t % W); t /= W;
    const int yy = (int)(t % H);
    const long b = t / H;
    float acc = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int oy = yy - (ky - 1);            // output pixel that read this input through tap (ky, kx)
      if (oy < 0 || oy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ox = xx - (kx - 1);
        if (ox < 0 || ox >= W) continue;
        acc = fmaf(dy[((b * H + oy) * W + ox) * C + c], w[c * 9 + ky * 3 + kx], acc);
      }
    }
    dx[i] = acc;
  }
}
// four channels per lane (C % 4 == 0), 32-bit pixel arithmetic; same pixel order per wave and the same final (0+1)+(2+3) fold as
// the scalar kernel below -> the same partial sums
__global__ __launch_bounds__(256) void dwconv3x3_bwd_dw_v4_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                  float* __restrict__ dwp, int H, int W, int C, int pixels) {
  __shared__ float red[3][64 * 36];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int cv = blockIdx.x * 64 + lane, vpp = C >> 2;
  const int p0 = blockIdx.y * DW_PIX, p1 = min(pixels, p0 + DW_PIX);
  f32x4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (cv < vpp) {
#pragma unroll 2
    for (int p = p0 + wv; p < p1; p += 4) {
      const int xx = p % W, row = p / W, yy = row % H;
      const f32x4 g = *(const f32x4*)(dy + (long)p * C + cv * 4);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int iy = yy + ky - 1;
        if (iy < 0 || iy >= H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int ix = xx + kx - 1;
          if (ix < 0 || ix >= W) continue;
          const f32x4 v = *(const f32x4*)(x + ((long)(row + ky - 1) * W + ix) * C + cv * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[ky * 3 + kx][j] = fmaf(g[j], v[j], acc[ky * 3 + kx][j]);
        }
      }
    }
  }
  if (wv > 0) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) red[wv - 1][lane * 36 + j * 9 + t] = acc[t][j];
  }
  __syncthreads();
  if (wv == 0 && cv < vpp) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < 9; ++t)
        dwp[(long)blockIdx.y * C * 9 + (long)(cv * 4 + j) * 9 + t] =
            (acc[t][j] + red[0][lane * 36 + j * 9 + t]) + (red[1][lane * 36 + j * 9 + t] + red[2][lane * 36 + j * 9 + t]);
  }
}
__global__ __launch_bounds__(256) void dwconv3x3_bwd_dw_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                               float* __restrict__ dwp, int H, int W, int C, long pixels) {
  __shared__ float red[4][64 * 9];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const long p0 = (long)blockIdx.y * DW_PIX, p1 = min(pixels, p0 + DW_PIX);
  float acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = 0.f;
  if (c < C) {
    for (long p = p0 + wv; p < p1; p += 4) {
      const int xx = (int)(p % W), yy = (int)((p / W) % H);
      const long b = p / ((long)W * H);
      const float g = dy[p * C + c];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int iy = yy + ky - 1;
        if (iy < 0 || iy >= H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int ix = xx + kx - 1;
          if (ix < 0 || ix >= W) continue;
          acc[ky * 3 + kx] = fmaf(g, x[((b * H + iy) * W + ix) * C + c], acc[ky * 3 + kx]);
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) red[wv][lane * 9 + t] = acc[t];
  __syncthreads();
  if (wv == 0 && c < C) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
      dwp[(long)blockIdx.y * C * 9 + (long)c * 9 + t] = (red[0][lane * 9 + t] + red[1][lane * 9 + t]) + (red[2][lane * 9 + t] + red[3][lane * 9 + t]);
  }
}
extern "C" int muse_dwconv3x3_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw_partial, int32_t batch,
                                  int32_t H, int32_t W, int32_t C, void* stream) {
  const long pixels = (long)batch * H * W, n = pixels * C;
  if (n <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (!launch_dwconv_v4<true>(dy, w, dx, batch, H, W, C, s)) {
    long g = (n + 255) / 256; if (g > 65535) g = 65535;
    hipLaunchKernelGGL(dwconv3x3_bwd_dx_kernel, dim3((unsigned)g), dim3(256), 0, s, dy, w, dx, H, W, C, n);
  }
  if ((C & 3) == 0 && pixels < (1L << 31) - DW_PIX && !((((uintptr_t)dy) | ((uintptr_t)x)) & 15))
    hipLaunchKernelGGL(dwconv3x3_bwd_dw_v4_kernel, dim3((C / 4 + 63) / 64, muse_dwconv3x3_bwd_nchunk(pixels)), dim3(256), 0, s, dy, x,
                       dw_partial, H, W, C, (int)pixels);
  else
    hipLaunchKernelGGL(dwconv3x3_bwd_dw_kernel, dim3((C + 63) / 64, muse_dwconv3x3_bwd_nchunk(pixels)), dim3(256), 0, s, dy, x,
                       dw_partial, H, W, C, pixels);
  return (int)hipGetLastError();
}

// GlobalResponseNorm backward.  Saved by the forward: stats[0 : B*C] = G (column norms), stats[B*C : 2*B*C] = N = G / (mean_c G + 1e-6).
//   S0[b,c] = sum_s dy,  S1[b,c] = sum_s dy x;   dbeta = sum_b S0,  dgamma = sum_b N S1
//   dL/dN_c = gamma_c S1_c;  dL/dG_c = (dL/dN_c - mean_j(dL/dN_j N_j)) / (m + 1e-6);   dx = dy (gamma N + 1) + x dL/dG_c / G_c
// work: 4 * B * C floats = [S0 | S1 -> N*S1 | K = dLdG / G | unused]
__global__ __launch_bounds__(256) void grn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             float* __restrict__ s0, float* __restrict__ s1, long S, int C) {
  __shared__ float r0[4][64], r1[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const long b = blockIdx.y;
  float a = 0.f, s = 0.f;
  if (c < C)
    for (long r = wv; r < S; r += 4) { const float g = dy[(b * S + r) * C + c]; s += g; a = fmaf(g, x[(b * S + r) * C + c], a); }
  r0[wv][lane] = s; r1[wv][lane] = a;
  __syncthreads();
  if (wv == 0 && c < C) {
    s0[b * C + c] = (r0[0][lane] + r0[1][lane]) + (r0[2][lane] + r0[3][lane]);
    s1[b * C + c] = (r1[0][lane] + r1[1][lane]) + (r1[2][lane] + r1[3][lane]);
  }
}
__global__ __launch_bounds__(256) void grn_bwd_coef_kernel(const float* __restrict__ G, const float* __restrict__ N,
                                                           const float* __restrict__ gamma, float* __restrict__ s1,
                                                           float* __restrict__ K, int C) {   // one block per image
  __shared__ float red[4], redg[4];
  const long b = blockIdx.x;
  float t = 0.f, sg = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) { t = fmaf(gamma[c] * s1[b * C + c], N[b * C + c], t); sg += G[b * C + c]; }
  t = wave_sum(t); sg = wave_sum(sg);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = t; redg[threadIdx.x >> 6] = sg; }
  __syncthreads();
  const float mean_an = ((red[0] + red[1]) + (red[2] + red[3])) / (float)C;
  const float denom = ((redg[0] + redg[1]) + (redg[2] + redg[3])) / (float)C + 1e-6f;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float a = gamma[c] * s1[b * C + c], g = G[b * C + c];
    K[b * C + c] = g > 0.f ? (a - mean_an) / denom / g : 0.f;
    s1[b * C + c] = N[b * C + c] * s1[b * C + c];     // -> dgamma contribution of this image
  }
}
__global__ __launch_bounds__(256) void grn_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                         const float* __restrict__ N, const float* __restrict__ K,
                                                         const float* __restrict__ gamma, float* __restrict__ dx, long S, int C, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long b = i / ((long)S * C);
    dx[i] = dy[i] * (gamma[c] * N[b * C + c] + 1.0f) + x[i] * K[b * C + c];
  }
}
extern "C" int muse_grn_bwd(const float* dy, const float* x, const float* gamma, const float* stats, float* dx, float* work,
                            int32_t batch, int64_t S, int32_t C, void* stream) {
  const long n = (long)batch * S * C, bc = (long)batch * C;
  if (n <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  float *s0 = work, *s1 = work + bc, *K = work + 2 * bc;
  hipLaunchKernelGGL(grn_bwd_reduce_kernel, dim3((C + 63) / 64, batch), dim3(256), 0, s, dy, x, s0, s1, (long)S, C);
  hipLaunchKernelGGL(grn_bwd_coef_kernel, dim3(batch), dim3(256), 0, s, stats, stats + bc, gamma, s1, K, C);
  long g = (n + 255) / 256; if (g > 65535) g = 65535;
  hipLaunchKernelGGL(grn_bwd_dx_kernel, dim3((unsigned)g), dim3(256), 0, s, dy, x, stats + bc, (const float*)K, gamma, dx, (long)S, C, n);
  return (int)hipGetLastError();
}

// x[r, :] *= w[r] * scal[0] / scal[1]   (weighted cross-entropy backward: per-token weight, n_valid / sum of weights)
__global__ __launch_bounds__(256) void scale_rows_kernel(float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ num,
                                                         const float* __restrict__ den, long rows, int cols, long ld) {
  const float k = num[0] / den[0];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < rows * cols; i += (long)gridDim.x * 256) {
    const long r = i / cols;
    const int c = (int)(i - r * cols);
    x[r * ld + c] *= w[r] * k;
  }
}
extern "C" int muse_scale_rows(float* x, const float* w, const float* num, const float* den, int64_t rows, int32_t cols, int64_t ld,
                               void* stream) {
  if (rows <= 0 || cols <= 0) return 0;
  long g = (rows * cols + 255) / 256; if (g > 65535) g = 65535;
  hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, w, num, den, (long)rows, cols, (long)ld);
  return (int)hipGetLastError();
}
