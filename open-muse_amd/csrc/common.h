// Shared device helpers for the muse_hip kernels (gfx950 / CDNA4 only: wave64, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;  // raw bfloat16 storage
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define MUSE_WAVE 64

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even (hardware v_cvt_pk_bf16_f32 on gfx950; same rounding torch uses for .to(bfloat16))
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) float f32x2_;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_){lo, hi}, bf16x2_));
}

// SiLU behind a GroupNorm: the ONE expression the GroupNorm kernels (vqgan.hip) and the convolution that applies the norm itself
// (conv_dma.hip) share, so both routes produce the same bits.  Hardware reciprocal (v_rcp_f32, 1 ulp): the correctly rounded division
// is eleven VALU instructions per element, which the fused convolution could not hide under its MFMAs.  Only the bf16x3 operand
// routes use it; GroupNorm kernels that write an f32 / bf16 tensor (the exact-f32 parity mode) divide with correct rounding.
__device__ __forceinline__ float gn_silu(float t) { return t * __builtin_amdgcn_rcpf(1.0f + __expf(-t)); }

// bf16x3 operand split of 4 floats: hi = bf16(x), lo = bf16(x - hi)  (conv_split.hip / conv_dma.hip / GroupNorm split output)
__device__ __forceinline__ void split4(const u32x4& v, u32x2& hi, u32x2& lo) {
  const float x0 = __uint_as_float(v[0]), x1 = __uint_as_float(v[1]), x2 = __uint_as_float(v[2]), x3 = __uint_as_float(v[3]);
  hi[0] = pack2_bf16(x0, x1);
  hi[1] = pack2_bf16(x2, x3);
  lo[0] = pack2_bf16(x0 - __uint_as_float(hi[0] << 16), x1 - __uint_as_float(hi[0] & 0xffff0000u));
  lo[1] = pack2_bf16(x2 - __uint_as_float(hi[1] << 16), x3 - __uint_as_float(hi[1] & 0xffff0000u));
}

// the split of four COMPUTED values (a producer kernel that writes its f32 result and that result's operand planes): the values are
// pinned in registers first - otherwise the compiler may contract a trailing multiply of their computation into the `x - hi` of the
// split (one fma on the unrounded product), and the lo plane is no longer the split of the f32 value that was stored
__device__ __forceinline__ void split4_values(float x0, float x1, float x2, float x3, u32x2& hi, u32x2& lo) {
  asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
  split4(u32x4{__float_as_uint(x0), __float_as_uint(x1), __float_as_uint(x2), __float_as_uint(x3)}, hi, lo);
}

// ---- operand images written by producer kernels -----------------------------------------------------------------------------------
// A kernel whose f32 result feeds weight GEMMs also writes that result as the GEMMs' operand image, four values per call:
//   lo_off > 0   "bf16x3" mode: the (hi, lo) bf16 planes, lo plane lo_off elements behind the hi plane (split4_values)
//   lo_off < 0   "f16" mode: ONE IEEE-half image, half(x * scale) with scale a power of two (1 for forward results, the backward pass's
//                gradient scale for gradients) - the bits muse_cast_f32_to_f16 makes of the f32 result; an overflow becomes inf (the
//                product turns NaN: loud) and is counted in *stats (the host's dynamic gradient scale backs off on it)
// Which of the two the *_x3 entry points write is process state set by muse_operand_images (rowops.hip); launchers ask img_format().
struct ImgFormat { long lo_sign; float scale; int* stats; };      // lo_sign: -1 = half image, +1 = bf16 planes
ImgFormat img_format(bool gradient);
__device__ __forceinline__ void store_image4(bf16_t* p, long lo_off, float scale, int* stats, float x0, float x1, float x2, float x3) {
  if (lo_off < 0) {
    asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
    typedef _Float16 h4_ __attribute__((ext_vector_type(4)));
    const h4_ h = {(_Float16)(x0 * scale), (_Float16)(x1 * scale), (_Float16)(x2 * scale), (_Float16)(x3 * scale)};
    const u32x2 w = __builtin_bit_cast(u32x2, h);
    *(u32x2*)p = w;
    // exponent all ones in any of the four halves (inf / NaN): rare, one atomic per offending lane
    const unsigned e0 = w[0] & 0x7c007c00u, e1 = w[1] & 0x7c007c00u;
    if (stats && (((e0 & 0xffffu) == 0x7c00u) | ((e0 >> 16) == 0x7c00u) | ((e1 & 0xffffu) == 0x7c00u) | ((e1 >> 16) == 0x7c00u))) atomicAdd(stats, 1);
  } else {
    u32x2 hi, lo;
    split4_values(x0, x1, x2, x3, hi, lo);
    *(u32x2*)p = hi;
    *(u32x2*)(p + lo_off) = lo;
  }
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static __device__ __forceinline__ float load(const float* p) { return *p; }
  static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static __device__ __forceinline__ float load(const bf16_t* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void store(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// ---- wave64 reductions -------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// exact erf-GELU (F.gelu default) and its derivative
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// erf(x / sqrt(2)) for the kernels whose STORAGE type is bf16 (FAST = true): Abramowitz & Stegun 7.1.26 - branch-free, one reciprocal,
// one exponential (exp(-x^2 / 2), the very factor the GELU derivative's density needs), five fmas; |error| <= 1.5e-7 (+ ~2 ulp of f32
// evaluation), four orders below the bf16 rounding of the values these kernels store.  The bf16 GLU / ffn_mid kernels are vector-ALU
// bound (~80 instructions per element with the library erff, whose two branches both execute in a divergent wave: ffn_mid_bwd 103 us
// of VALU issue for 92 us of memory time); f32 storage keeps the library erff.  Forward and backward of a pair use the same
// function, so the backward's recomputed gelu(a) * b is still the forward's tensor bit for bit.
template <bool FAST> __device__ __forceinline__ float erf_rsqrt2(float x) {
  if constexpr (!FAST) {
    return erff(x * 0.70710678118654752440f);
  } else {
    const float t = __builtin_amdgcn_rcpf(fmaf(0.23164189f, fabsf(x), 1.0f));       // 1 / (1 + p |x| / sqrt(2)), p = 0.3275911
    const float e = __expf(-0.5f * x * x);
    float q = fmaf(1.061405429f, t, -1.453152027f);
    q = fmaf(q, t, 1.421413741f);
    q = fmaf(q, t, -0.284496736f);
    q = fmaf(q, t, 0.254829592f);
    return copysignf(1.0f - q * t * e, x);
  }
}
template <bool FAST> __device__ __forceinline__ float gelu_erf_t(float x) { return 0.5f * x * (1.0f + erf_rsqrt2<FAST>(x)); }
template <bool FAST> __device__ __forceinline__ float gelu_erf_grad_t(float x) {
  const float cdf = 0.5f * (1.0f + erf_rsqrt2<FAST>(x));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

#define MUSE_CHECK_LAUNCH() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return (int)e__; } while (0)
