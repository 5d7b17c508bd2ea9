// Row / elementwise kernels of MaskGiTUViT_v2 (muse/modeling_transformer_v2.py) that the MaskGit
// path does not have: norm with a pre-norm residual stream (RMSNorm / LayerNorm), AdaLN modulation, SiLU, depthwise 3x3
// convolution (NHWC), GlobalResponseNorm, sinusoidal micro-conditioning, weighted mean of per-token losses.  f32, forward.
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
extern "C" int muse_norm_res_fwd(const float* x, const float* res, const float* w, float* y, float* pre, int64_t rows,
                                 int32_t cols, float eps, int32_t mode, void* stream) {
  if (cols % 4 || (mode != 0 && mode != 1)) return MUSE_ERR_UNSUPPORTED;
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(norm_res_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, res, w, y, pre,
                     (long)rows, cols, eps, mode);
  return (int)hipGetLastError();
}

// =================================================================================================================
// AdaLN modulation (:1025-1037): y[b, r, c] = x[b, r, c] * (1 + ss[b, c]) + ss[b, C + c],  ss = mapper(silu(cond)) [B, 2C]
// =================================================================================================================
__global__ __launch_bounds__(256) void adaln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ ss,
                                                        float* __restrict__ y, long rows_per_batch, int C, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const long e = i * 4, row = e / C;
    const int c = (int)(e - row * C);
    const long b = row / rows_per_batch;
    const f32x4 v = *(const f32x4*)(x + e);
    const f32x4 sc = *(const f32x4*)(ss + b * 2 * C + c), sh = *(const f32x4*)(ss + b * 2 * C + C + c);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = v[j] * (1.0f + sc[j]) + sh[j];
    *(f32x4*)(y + e) = o;
  }
}
extern "C" int muse_adaln_fwd(const float* x, const float* ss, float* y, int32_t batch, int64_t rows_per_batch, int32_t C, void* stream) {
  if (C % 4) return MUSE_ERR_UNSUPPORTED;
  const long n4 = (long)batch * rows_per_batch * C / 4;
  if (n4 <= 0) return 0;
  long g = (n4 + 255) / 256; if (g > 16384) g = 16384;
  hipLaunchKernelGGL(adaln_fwd_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, ss, y, (long)rows_per_batch, C, n4);
  return (int)hipGetLastError();
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
extern "C" int muse_dwconv3x3_nhwc(const float* x, const float* w, float* y, int32_t batch, int32_t H, int32_t W, int32_t C,
                                   void* stream) {
  const long n = (long)batch * H * W * C;
  if (n <= 0) return 0;
  long g = (n + 255) / 256; if (g > 65535) g = 65535;
  hipLaunchKernelGGL(dwconv3x3_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, w, y, H, W, C, n);
  return (int)hipGetLastError();
}

// =================================================================================================================
// GlobalResponseNorm (:741-751) on [B, S, C] (S = H*W pixels):  Gx[b,c] = ||x[b,:,c]||_2 ;  Nx = Gx / (mean_c Gx + 1e-6) ;
// y = gamma * (x * Nx) + beta + x.   scratch: B*C floats.
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
__global__ __launch_bounds__(256) void grn_scale_kernel(float* __restrict__ gx, int C) {   // in place: Gx -> Nx, one block per image
  __shared__ float red[4];
  const long b = blockIdx.x;
  float s = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) s += gx[b * C + c];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / (float)C;
  for (int c = threadIdx.x; c < C; c += 256) gx[b * C + c] = gx[b * C + c] / (mean + 1e-6f);
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
extern "C" int muse_grn_fwd(const float* x, const float* gamma, const float* beta, float* y, float* scratch, int32_t batch,
                            int64_t S, int32_t C, void* stream) {
  const long n = (long)batch * S * C;
  if (n <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(grn_colnorm_kernel, dim3((C + 63) / 64, batch), dim3(256), 0, s, x, scratch, (long)S, C);
  hipLaunchKernelGGL(grn_scale_kernel, dim3(batch), dim3(256), 0, s, scratch, C);
  long g = (n + 255) / 256; if (g > 65535) g = 65535;
  hipLaunchKernelGGL(grn_apply_kernel, dim3((unsigned)g), dim3(256), 0, s, x, (const float*)scratch, gamma, beta, y, (long)S, C, n);
  return (int)hipGetLastError();
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
