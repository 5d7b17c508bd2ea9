// MFMA GEMM core for gfx950, shared by the dense/batched GEMM entry points (gemm.hip) and the NHWC
// implicit-GEMM convolution (conv.hip).
//
//   C[m,n] = epilogue( alpha * sum_k A(m,k) * B(n,k) )
//
// Operand layouts (per operand): 0 = "k-contiguous"  X(r,k) at X + r*ld + k   (nn.Linear weight, activations)
//                                1 = "k-major"       X(r,k) at X + k*ld + r   (the transposed views backward needs)
// so Linear fwd is (0,0), dX = dY*W is (0,1), dW = dY^T*X is (1,1); attention's P*V is (0,1), dV/dK are (1,1).
//
// Tiling: 128x128 block tile, 4 waves (2x2), each wave a 64x64 sub-tile as 4x4 MFMA 16x16 fragments:
//   bf16 : v_mfma_f32_16x16x32_bf16, BK=64   fp32 : v_mfma_f32_16x16x4_f32 (exact fp32 fma chain), BK=16
// The MFMA is issued with the operands swapped (B-fragment as the "A" input) so that each lane ends up with 4
// consecutive n of one output row -> 8/16-byte row-major stores.
// Staging: global -> VGPR (16 B per lane, issued one K-tile ahead) -> LDS.  k-contiguous tiles use a +32 B padded row
// stride (conflict-free ds_read_b128); k-major bf16 tiles keep the [k][r] image and are read with
// ds_read_b64_tr_b16 (hardware transpose), rows stored with k-bits 2/3 swapped so the 32 lanes of a half-wave cover
// 8 distinct rows (all 64 banks).
#pragma once
#include "common.h"
#include <stdlib.h>

struct GemmParams {
  const void* A;
  const void* B;
  void* C;
  const float* bias;     // [N] fp32, added per column (may be null)
  const float* rowvec;   // [M] fp32, added per row    (may be null)
  const void* residual;  // [M, ldr] of the output type, added after the activation (may be null)
  int M, N, K;
  long lda, ldb, ldc, ldr;
  int zdiv;  // batch index z -> (z / zdiv, z % zdiv)
  long sA0, sA1, sB0, sB1, sC0, sC1;
  float alpha;
  int accumulate;  // C += result
  int act;         // 0 none, 1 erf-GELU
  int split_k;     // > 1: blockIdx.y selects a K slice, partial results are added to C with f32 atomics (C pre-initialised)
  // implicit-GEMM convolution geometry (conv A loader only)
  int cH, cW, cCin, cKS, cUps, cCinShift;
};

template <typename T> struct TileCfg;
template <> struct TileCfg<bf16_t> {
  static constexpr int BK = 64, CH = 8;
  static constexpr int KC_STRIDE = 160;  // bytes per row, k-contiguous image (128 + 32)
};
template <> struct TileCfg<float> {
  static constexpr int BK = 64, CH = 4;  // 8192 MFMA cycles per wave between barrier pairs (f32 MFMA is 16x slower per flop)
  static constexpr int KC_STRIDE = 288;  // 256 + 32
};
// k-major image: [BK k-rows][ROWS] with a row stride of ROWS*sizeof(T) + pad, pad chosen so that the rows read together
// by one half-wave fall on distinct banks (bf16 tr-reads: stride/4 == 8 mod 16 banks; f32 b32 reads: 4 rows -> +16 banks)
template <typename T, int ROWS> struct KmCfg {
  static constexpr int STRIDE = ROWS * (int)sizeof(T) + (sizeof(T) == 2 ? 32 : 16);
};
template <typename T, int ROWS> struct TileBytes {
  static constexpr int KC = ROWS * TileCfg<T>::KC_STRIDE;
  static constexpr int KM = TileCfg<T>::BK * KmCfg<T, ROWS>::STRIDE;
  static constexpr int VALUE = KC > KM ? KC : KM;
};

__device__ __forceinline__ int km_phys_row_bf16(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }

// ---------------------------------------------------------------------------------------------------------------
// global -> register tile loaders.  Each thread owns NCH 16-byte chunks of the 128 x BK operand tile.
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int LAYOUT, int ROWS, int NT>
struct PlainLoader {
  using Cfg = TileCfg<T>;
  static constexpr int NCH = ROWS * Cfg::BK * (int)sizeof(T) / 16 / NT;  // 16-byte chunks per thread
  const T* base;
  long ld;
  int R, K, r0;
  __device__ __forceinline__ void init(const void* ptr, long ld_, int R_, int K_, int r0_, const GemmParams&) {
    base = (const T*)ptr; ld = ld_; R = R_; K = K_; r0 = r0_;
  }
  __device__ __forceinline__ u32x4 load(int i, int k0) const {
    const int c = threadIdx.x + NT * i;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (LAYOUT == 0) {
      constexpr int CPR = Cfg::BK / Cfg::CH;
      const int row = r0 + c / CPR, k = k0 + (c % CPR) * Cfg::CH;
      if (row < R && k < K) v = *(const u32x4*)(base + (long)row * ld + k);
    } else {
      constexpr int CPR = ROWS / Cfg::CH;
      const int kr = k0 + c / CPR, r = r0 + (c % CPR) * Cfg::CH;
      if (kr < K && r < R) v = *(const u32x4*)(base + (long)kr * ld + r);
    }
    return v;
  }
  static __device__ __forceinline__ int lds_off(int i) {
    const int c = threadIdx.x + NT * i;
    if (LAYOUT == 0) {
      constexpr int CPR = Cfg::BK / Cfg::CH;
      return (c / CPR) * Cfg::KC_STRIDE + (c % CPR) * 16;
    } else {
      constexpr int CPR = ROWS / Cfg::CH;
      int kr = c / CPR;
      if (sizeof(T) == 2) kr = km_phys_row_bf16(kr);
      return kr * KmCfg<T, ROWS>::STRIDE + (c % CPR) * 16;
    }
  }
};

// implicit-GEMM A operand of an NHWC stride-1 SAME convolution:  m = (b, y, x),  k = (ky, kx, cin)
// optional nearest x2 upsample of the input folded into the index math (decoder upsample_conv).
template <typename T, int ROWS, int NT>
struct ConvLoader {
  using Cfg = TileCfg<T>;
  static constexpr int CPR = Cfg::BK / Cfg::CH;
  static constexpr int NCH = ROWS * Cfg::BK * (int)sizeof(T) / 16 / NT;
  const T* base;
  int K, H, W, Cin, KS, ups, cshift;
  int py[NCH], px[NCH];
  long pb[NCH];
  bool pv[NCH];
  __device__ __forceinline__ void init(const void* ptr, long, int R_, int K_, int r0_, const GemmParams& p) {
    base = (const T*)ptr; K = K_; H = p.cH; W = p.cW; Cin = p.cCin; KS = p.cKS; ups = p.cUps; cshift = p.cCinShift;
    const int ih = ups ? (H >> 1) : H, iw = ups ? (W >> 1) : W;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = threadIdx.x + NT * i;
      const int m = r0_ + c / CPR;
      pv[i] = m < R_;
      const int b = m / (H * W), rem = m - b * (H * W);
      py[i] = rem / W; px[i] = rem - py[i] * W;
      pb[i] = (long)b * ih * iw;
    }
  }
  __device__ __forceinline__ u32x4 load(int i, int k0) const {
    const int c = threadIdx.x + NT * i;
    const int k = k0 + (c % CPR) * Cfg::CH;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (pv[i] && k < K) {
      const int kpos = cshift >= 0 ? (k >> cshift) : (k / Cin), ci = k - kpos * Cin;
      const int ky = KS == 3 ? ((kpos * 11) >> 5) : 0, kx = kpos - ky * KS, pad = (KS - 1) >> 1;  // kpos < 9
      int iy = py[i] + ky - pad, ix = px[i] + kx - pad;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
        int iw = W;
        if (ups) { iy >>= 1; ix >>= 1; iw = W >> 1; }
        v = *(const u32x4*)(base + (pb[i] + (long)iy * iw + ix) * Cin + ci);
      }
    }
    return v;
  }
  static __device__ __forceinline__ int lds_off(int i) {
    const int c = threadIdx.x + NT * i;
    return (c / CPR) * Cfg::KC_STRIDE + (c % CPR) * 16;
  }
};

// ---------------------------------------------------------------------------------------------------------------
// LDS -> MFMA fragment readers.  A fragment is (16 rows) x (one MFMA K-group); lane l supplies row (l & 15).
// ---------------------------------------------------------------------------------------------------------------
template <int LAYOUT, int ROWS>
__device__ __forceinline__ bf16x8 frag_bf16(const unsigned char* tile, int rowbase, int ks, int lane) {
  if (LAYOUT == 0) {
    return *(const bf16x8*)(tile + (rowbase + (lane & 15)) * TileCfg<bf16_t>::KC_STRIDE + ks * 64 + (lane >> 4) * 16);
  } else {
    const int p = lane & 15, g = lane >> 4;
    const int kr = ks * 32 + 8 * g + (p >> 2);
    const int col = (rowbase + (p & 3) * 4) * 2;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const unsigned char* a0 = tile + km_phys_row_bf16(kr) * KmCfg<bf16_t, ROWS>::STRIDE + col;
    const unsigned char* a1 = tile + km_phys_row_bf16(kr + 4) * KmCfg<bf16_t, ROWS>::STRIDE + col;
    union { s16x4 h[2]; bf16x8 v; } u;
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a1);
    return u.v;
  }
}

template <int LAYOUT, int ROWS>
__device__ __forceinline__ f32x4 frag_f32(const unsigned char* tile, int rowbase, int ks, int lane) {
  // returns the 4 k-values (k = 16*ks + 4*(lane>>4) + s, s = 0..3) of row (rowbase + lane&15); MFMA step s uses [s]
  if (LAYOUT == 0) {
    return *(const f32x4*)(tile + (rowbase + (lane & 15)) * TileCfg<float>::KC_STRIDE + ks * 64 + (lane >> 4) * 16);
  } else {
    constexpr int ST = KmCfg<float, ROWS>::STRIDE;
    const unsigned char* q = tile + (16 * ks + 4 * (lane >> 4)) * ST + (rowbase + (lane & 15)) * 4;
    f32x4 v;
    v[0] = *(const float*)(q);
    v[1] = *(const float*)(q + ST);
    v[2] = *(const float*)(q + 2 * ST);
    v[3] = *(const float*)(q + 3 * ST);
    return v;
  }
}

template <typename TC> struct OutVec;
template <> struct OutVec<float> {
  static __device__ __forceinline__ void load4(const float* p, float (&v)[4]) {
    const f32x4 t = *(const f32x4*)p; v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  }
  static __device__ __forceinline__ void store4(float* p, const float (&v)[4]) {
    f32x4 t = {v[0], v[1], v[2], v[3]}; *(f32x4*)p = t;
  }
};
template <> struct OutVec<bf16_t> {
  static __device__ __forceinline__ void load4(const bf16_t* p, float (&v)[4]) {
    const u32x2 t = *(const u32x2*)p;
    v[0] = __uint_as_float(t[0] << 16); v[1] = __uint_as_float(t[0] & 0xffff0000u);
    v[2] = __uint_as_float(t[1] << 16); v[3] = __uint_as_float(t[1] & 0xffff0000u);
  }
  static __device__ __forceinline__ void store4(bf16_t* p, const float (&v)[4]) {
    u32x2 t;
    t[0] = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
    t[1] = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
    *(u32x2*)p = t;
  }
};

// ---------------------------------------------------------------------------------------------------------------
template <typename T, typename TC, int AL, int BL, int BM, typename ALoader, typename BLoader>
__global__ __launch_bounds__(BM * 2, 2) void gemm_kernel(GemmParams p) {
  using Cfg = TileCfg<T>;
  constexpr int TA_BYTES = TileBytes<T, BM>::VALUE, TB_BYTES = TileBytes<T, 128>::VALUE;
  __shared__ __attribute__((aligned(16))) unsigned char smem[TA_BYTES + TB_BYTES];
  unsigned char* tA = smem;
  unsigned char* tB = smem + TA_BYTES;

  // XCD-aware tile order: blocks b, b+8, b+16, ... (one XCD, one L2) get a contiguous range of tile ids ...
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + 127) >> 7, ntiles = ntm * ntn;
  const int bq = ntiles >> 3, br = ntiles & 7, xcd = blockIdx.x & 7, bi = blockIdx.x >> 3;
  const int tid_ = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + bi;
  // ... walked as a grouped raster: GM M-tiles x all N-tiles per group, M fastest.  The blocks an XCD runs at once then
  // share a few A row-panels and B column-panels (~ one 4 MB L2) instead of streaming every B panel from fabric.
  constexpr int GM = 1024 / BM;
  const int grp = tid_ / (GM * ntn), first_m = grp * GM;
  const int gm = min(ntm - first_m, GM), in_grp = tid_ - grp * (GM * ntn);
  const int m0 = (first_m + in_grp % gm) * BM, n0 = (in_grp / gm) << 7;

  const int z = blockIdx.z, zq = z / p.zdiv, zr = z - zq * p.zdiv;
  const T* Ap = (const T*)p.A + zq * p.sA0 + zr * p.sA1;
  const T* Bp = (const T*)p.B + zq * p.sB0 + zr * p.sB1;
  TC* Cp = (TC*)p.C + zq * p.sC0 + zr * p.sC1;

  ALoader la; la.init(Ap, p.lda, p.M, p.K, m0, p);
  BLoader lb; lb.init(Bp, p.ldb, p.N, p.K, n0, p);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64;  // BM/64 x 2 waves, 64 x 64 each

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 ra[ALoader::NCH], rb[BLoader::NCH];
  int nk = (p.K + Cfg::BK - 1) / Cfg::BK, kt0 = 0;
  if (p.split_k > 1) {
    const int per = (nk + p.split_k - 1) / p.split_k;
    kt0 = blockIdx.y * per;
    nk = min(nk, kt0 + per);
    if (kt0 >= nk) return;
  }
#pragma unroll
  for (int i = 0; i < ALoader::NCH; ++i) ra[i] = la.load(i, kt0 * Cfg::BK);
#pragma unroll
  for (int i = 0; i < BLoader::NCH; ++i) rb[i] = lb.load(i, kt0 * Cfg::BK);
#pragma unroll
  for (int i = 0; i < ALoader::NCH; ++i) *(u32x4*)(tA + ALoader::lds_off(i)) = ra[i];
#pragma unroll
  for (int i = 0; i < BLoader::NCH; ++i) *(u32x4*)(tB + BLoader::lds_off(i)) = rb[i];
  __syncthreads();

  for (int kt = kt0; kt < nk; ++kt) {
    const bool more = (kt + 1) < nk;
    if (more) {
#pragma unroll
      for (int i = 0; i < ALoader::NCH; ++i) ra[i] = la.load(i, (kt + 1) * Cfg::BK);
#pragma unroll
      for (int i = 0; i < BLoader::NCH; ++i) rb[i] = lb.load(i, (kt + 1) * Cfg::BK);
    }
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 af[4], bf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          af[i] = frag_bf16<AL, BM>(tA, wr + i * 16, ks, lane);
          bf[i] = frag_bf16<BL, 128>(tB, wc + i * 16, ks, lane);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < Cfg::BK / 16; ++ks) {
        f32x4 af[4], bf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          af[i] = frag_f32<AL, BM>(tA, wr + i * 16, ks, lane);
          bf[i] = frag_f32<BL, 128>(tB, wc + i * 16, ks, lane);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j][s], af[i][s], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
    if (more) {
#pragma unroll
      for (int i = 0; i < ALoader::NCH; ++i) *(u32x4*)(tA + ALoader::lds_off(i)) = ra[i];
#pragma unroll
      for (int i = 0; i < BLoader::NCH; ++i) *(u32x4*)(tB + BLoader::lds_off(i)) = rb[i];
      __syncthreads();
    }
  }

  // ---- epilogue: lane holds C[m][n..n+3], m = m0+wr+16i+(lane&15), n = n0+wc+16j+4*(lane>>4) ----
  const bool vec_ok = ((p.ldc & 3) == 0) && ((((uintptr_t)Cp) & 15) == 0) &&
                      (p.residual == nullptr || (((p.ldr & 3) == 0) && ((((uintptr_t)p.residual) & 15) == 0)));
  const TC* Rp = (const TC*)p.residual;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wr + i * 16 + (lane & 15);
    if (m >= p.M) continue;
    const float rv = p.rowvec ? p.rowvec[m] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wc + j * 16 + 4 * (lane >> 4);
      if (n >= p.N) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float add = rv;  // (rowvec + bias) first, then one rounding with alpha*acc: addmm(beta*input + alpha*mm) order
        if (p.bias && (n + r) < p.N) add += p.bias[n + r];
        float x = p.alpha * acc[i][j][r] + add;
        if (p.act == 1) x = gelu_erf(x);
        v[r] = x;
      }
      TC* cptr = Cp + (long)m * p.ldc + n;
      if constexpr (sizeof(TC) == 4) {
        if (p.split_k > 1) {  // partial sum of one K slice: hardware f32 atomics straight into C
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if ((n + r) < p.N) unsafeAtomicAdd((float*)cptr + r, v[r]);
          continue;
        }
      }
      if (vec_ok && (n + 3) < p.N) {
        if (Rp) { float t[4]; OutVec<TC>::load4(Rp + (long)m * p.ldr + n, t); v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3]; }
        if (p.accumulate) { float t[4]; OutVec<TC>::load4(cptr, t); v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3]; }
        OutVec<TC>::store4(cptr, v);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if ((n + r) < p.N) {
            float x = v[r];
            if (Rp) x += Elem<TC>::load(Rp + (long)m * p.ldr + n + r);
            if (p.accumulate) x += Elem<TC>::load(cptr + r);
            Elem<TC>::store(cptr + r, x);
          }
        }
      }
    }
  }
}

template <typename T, typename TC, int AL, int BL, int BM, typename ALoader, typename BLoader>
static inline int launch_gemm(const GemmParams& p, int batch, hipStream_t stream) {
  if (p.M <= 0 || p.N <= 0 || batch <= 0) return 0;
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + 127) / 128;
  dim3 grid(ntm * ntn, p.split_k > 1 ? p.split_k : 1, batch);
  hipLaunchKernelGGL((gemm_kernel<T, TC, AL, BL, BM, ALoader, BLoader>), grid, dim3(BM * 2), 0, stream, p);
  return (int)hipGetLastError();
}

// big-M problems use the 256 x 128 tile (8 waves): 25 % fewer operand bytes per flop through L2 / LDS
static inline bool use_bm256(const GemmParams& p, int batch) {
  static const int enabled = [] { const char* e = getenv("MUSE_BM256"); return (e && e[0] == '1') ? 1 : 0; }();
  if (!enabled) return false;  // 1 block/CU at 150+ VGPRs: kept behind a switch until it wins the A/B on hardware
  const long tiles256 = (long)((p.M + 255) / 256) * ((p.N + 127) / 128) * batch * (p.split_k > 1 ? p.split_k : 1);
  return p.M >= 1024 && tiles256 >= 512;
}
