// MFMA GEMM core for gfx950 (register-staged 128 x 128 tile), shared by the dense/batched GEMM entry point (gemm.hip), the
// NHWC implicit-GEMM convolutions (vqgan.hip: exact f32 / bf16; conv_split.hip: bf16x3 on f32 input) and, for its loaders and
// epilogue helpers, by the 256 x 256 LDS-DMA kernel (gemm256.h) that takes the large bf16 products.  This kernel keeps: f32
// operands (exact-f32 MFMA), small / ragged / batched products, GELU epilogues, atomic split-K.
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
  int split_k;     // > 1: blockIdx.y selects a K slice
  long split_stride;  // != 0: slice y stores its partial tile to C + y*split_stride (plain stores, reduced by
                      // muse_sum_slices); == 0: slices are added to C with f32 atomics (C pre-initialised)
  // implicit-GEMM convolution geometry (conv A loader only)
  int cH, cW, cCin, cKS, cUps, cCinShift;
  // bf16x3 product (gemm256.h PipeX3): elements from A / B (the hi planes) to the lo planes
  long a_lo, b_lo;
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
// global -> register tile loaders.  Each thread owns NCH 16-byte chunks of the ROWS x BK operand tile.
// Loads are buffer_load_dwordx4 through a per-block (wave-uniform, SGPR) buffer descriptor: the chunk's byte offset inside
// the operand lives in one VGPR computed once, the K advance is a scalar offset, and an invalid chunk (row outside the
// matrix, k beyond K, conv tap in the zero padding) is redirected to an out-of-range offset, which the hardware
// returns as zeros - no exec-mask branches and no 64-bit address arithmetic in the K loop.
// ---------------------------------------------------------------------------------------------------------------
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 buf_load16(rsrc_t rs, unsigned voff, unsigned soff) {
  return __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)soff, 0);
}

template <typename T, int LAYOUT, int ROWS, int NT>
struct PlainLoader {
  using Cfg = TileCfg<T>;
  static constexpr int NCH = ROWS * Cfg::BK * (int)sizeof(T) / 16 / NT;  // 16-byte chunks per thread
  static constexpr int CPR = LAYOUT == 0 ? Cfg::BK / Cfg::CH : ROWS / Cfg::CH;
  rsrc_t rs;
  unsigned voff[NCH];  // byte offset of the chunk at k0 = 0 (or `oob` for a row outside the matrix)
  unsigned oob, kstep;
  int K;
  __device__ __forceinline__ void init(const void* ptr, long ld, int R, int K_, int r0, const GemmParams&) {
    constexpr int E = (int)sizeof(T);
    K = K_;
    const long cols = LAYOUT == 0 ? K : R, rows = LAYOUT == 0 ? R : K;
    const unsigned bytes = (unsigned)(((rows - 1) * ld + (cols + Cfg::CH - 1) / Cfg::CH * Cfg::CH) * E);
    rs = make_rsrc(ptr, bytes);
    oob = (bytes + 15u) & ~15u;
    kstep = LAYOUT == 0 ? (unsigned)E : (unsigned)(ld * E);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = threadIdx.x + NT * i;
      if (LAYOUT == 0) {
        const int row = r0 + c / CPR, col = (c % CPR) * Cfg::CH;
        voff[i] = row < R ? (unsigned)(((long)row * ld + col) * E) : oob;
      } else {
        const int kr = c / CPR, r = r0 + (c % CPR) * Cfg::CH;
        voff[i] = r < R ? (unsigned)(((long)kr * ld + r) * E) : oob;
      }
    }
  }
  __device__ __forceinline__ u32x4 load(int i, int k0) const {
    const int c = threadIdx.x + NT * i;
    const int kk = k0 + (LAYOUT == 0 ? (c % CPR) * Cfg::CH : c / CPR);  // first k of this chunk
    return buf_load16(rs, kk < K ? voff[i] : oob, (unsigned)k0 * kstep);
  }
  static __device__ __forceinline__ int lds_off(int i) {
    const int c = threadIdx.x + NT * i;
    if (LAYOUT == 0) {
      return (c / CPR) * Cfg::KC_STRIDE + (c % CPR) * 16;
    } else {
      int kr = c / CPR;
      if (sizeof(T) == 2) kr = km_phys_row_bf16(kr);
      return kr * KmCfg<T, ROWS>::STRIDE + (c % CPR) * 16;
    }
  }
};

// implicit-GEMM A operand of an NHWC convolution:  m = (b, y, x) over the OUTPUT pixels,  k = (ky, kx, cin).  cUps selects
// the input geometry:  0 = stride 1, SAME padding;  1 = nearest x2 upsample of the input folded into the index math
// (decoder upsample_conv);  2 = stride 2 over a (2H x 2W) input zero-padded by one row / column at the bottom / right
// (taming Downsample: F.pad(0,1,0,1) + Conv2d(3, stride 2, padding 0), muse/modeling_taming_vqgan.py:53-59).
// Per chunk: `center` = byte offset of the pixel's first-tap data, `tapmask` bit t = tap t lies inside the image.
template <typename T, int ROWS, int NT>
struct ConvLoader {
  using Cfg = TileCfg<T>;
  static constexpr int CPR = Cfg::BK / Cfg::CH;
  static constexpr int NCH = ROWS * Cfg::BK * (int)sizeof(T) / 16 / NT;
  rsrc_t rs;
  unsigned center[NCH], tapmask[NCH], pyx[NCH];
  unsigned oob;
  int iw, Cin, KS, ups, cshift, padv;
  __device__ __forceinline__ void init(const void* ptr, long, int R_, int K_, int r0_, const GemmParams& p) {
    constexpr int E = (int)sizeof(T);
    const int H = p.cH, W = p.cW;
    Cin = p.cCin; KS = p.cKS; ups = p.cUps; cshift = p.cCinShift;
    const int ih = ups == 1 ? (H >> 1) : ups == 2 ? (H << 1) : H;
    iw = ups == 1 ? (W >> 1) : ups == 2 ? (W << 1) : W;
    const unsigned bytes = (unsigned)((long)(R_ / (H * W)) * ih * iw * Cin * E);
    rs = make_rsrc(ptr, bytes);
    oob = (bytes + 15u) & ~15u;
    const int pad = (KS - 1) >> 1;
    padv = ups == 2 ? 0 : pad;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = threadIdx.x + NT * i;
      const int m = r0_ + c / CPR;
      const int b = m / (H * W), rem = m - b * (H * W);
      const int y = rem / W, x = rem - y * W;
      unsigned mask = 0;
      if (m < R_) {
        for (int t = 0; t < KS * KS; ++t) {
          const int ky = t / KS, kx = t - ky * KS;
          if (ups == 2) {
            if (2 * y + ky < ih && 2 * x + kx < iw) mask |= 1u << t;
          } else {
            const int iy = y + ky - pad, ix = x + kx - pad;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) mask |= 1u << t;
          }
        }
      }
      tapmask[i] = mask;
      pyx[i] = (unsigned)y | ((unsigned)x << 16);
      center[i] = ups == 1 ? (unsigned)(b * ih * iw)
                : ups == 2 ? (unsigned)(((long)(b * ih + 2 * y) * iw + 2 * x) * Cin * E)
                           : (unsigned)(((long)(b * H + y) * W + x) * Cin * E);
    }
  }
  __device__ __forceinline__ u32x4 load(int i, int k0) const {
    constexpr int E = (int)sizeof(T);
    const int c = threadIdx.x + NT * i;
    const int k = k0 + (c % CPR) * Cfg::CH;
    const int kpos = cshift >= 0 ? (k >> cshift) : (k / Cin), ci = k - kpos * Cin;
    const int ky = KS == 3 ? ((kpos * 11) >> 5) : 0, kx = kpos - ky * KS;  // valid taps: kpos < 9
    const bool ok = kpos < 9 && ((tapmask[i] >> kpos) & 1u);
    unsigned off;
    if (ups != 1) {
      off = center[i] + (unsigned)((((ky - padv) * iw + (kx - padv)) * Cin + ci) * E);
    } else {
      const int pad = (KS - 1) >> 1;
      const int iy = ((int)(pyx[i] & 0xffffu) + ky - pad) >> 1, ix = ((int)(pyx[i] >> 16) + kx - pad) >> 1;
      off = (unsigned)(((long)(center[i] + (unsigned)(iy * iw + ix)) * Cin + ci) * E);
    }
    return buf_load16(rs, ok ? off : oob, 0u);
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
    t[0] = pack2_bf16(v[0], v[1]);
    t[1] = pack2_bf16(v[2], v[3]);
    *(u32x2*)p = t;
  }
};

// ---------------------------------------------------------------------------------------------------------------
template <typename T, typename TC, int AL, int BL, int BM, int NSTAGE, typename ALoader, typename BLoader>
__global__ __launch_bounds__(BM * 2, 2) void gemm_kernel(GemmParams p) {
  using Cfg = TileCfg<T>;
  constexpr int TA_BYTES = TileBytes<T, BM>::VALUE, TB_BYTES = TileBytes<T, 128>::VALUE;
  // bf16: two LDS stages + two register stages (K-tile t+2 in flight while t computes and t+1 is written to LDS);
  // f32: one stage (its MFMA phase is 16x longer per byte and the tiles are twice as large).
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* tA = smem;
  unsigned char* tB = smem + TA_BYTES;
  constexpr int STAGE_BYTES = TA_BYTES + TB_BYTES;

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
  TC* Cp = (TC*)p.C + zq * p.sC0 + zr * p.sC1 + (p.split_k > 1 ? (long)blockIdx.y * p.split_stride : 0L);
  const bool atomic_out = p.split_k > 1 && p.split_stride == 0;

  int nk = (p.K + Cfg::BK - 1) / Cfg::BK, kt0 = 0;
  if (p.split_k > 1) {
    const int per = (nk + p.split_k - 1) / p.split_k;
    kt0 = blockIdx.y * per;
    nk = min(nk, kt0 + per);
    if (kt0 >= nk) return;
  }
  const int kend = min(p.K, nk * Cfg::BK);  // K range of this block's slice: tiles past it load zeros
  ALoader la; la.init(Ap, p.lda, p.M, kend, m0, p);
  BLoader lb; lb.init(Bp, p.ldb, p.N, kend, n0, p);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64;  // BM/64 x 2 waves, 64 x 64 each

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](const unsigned char* cA, const unsigned char* cB) {
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 af[4], bf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          af[i] = frag_bf16<AL, BM>(cA, wr + i * 16, ks, lane);
          bf[i] = frag_bf16<BL, 128>(cB, wc + i * 16, ks, lane);
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
          af[i] = frag_f32<AL, BM>(cA, wr + i * 16, ks, lane);
          bf[i] = frag_f32<BL, 128>(cB, wc + i * 16, ks, lane);
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
  };

  if constexpr (NSTAGE == 1) {
    u32x4 ra[ALoader::NCH], rb[BLoader::NCH];
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
      compute(tA, tB);
      __syncthreads();
      if (more) {
#pragma unroll
        for (int i = 0; i < ALoader::NCH; ++i) *(u32x4*)(tA + ALoader::lds_off(i)) = ra[i];
#pragma unroll
        for (int i = 0; i < BLoader::NCH; ++i) *(u32x4*)(tB + BLoader::lds_off(i)) = rb[i];
        __syncthreads();
      }
    }
  } else {
    // register sets r0 / r1 and LDS stages 0 / 1 alternate; the loop is unrolled by two so every index is static.
    u32x4 ra0[ALoader::NCH], rb0[BLoader::NCH], ra1[ALoader::NCH], rb1[BLoader::NCH];
    unsigned char* tA1 = tA + STAGE_BYTES;
    unsigned char* tB1 = tB + STAGE_BYTES;
#define MUSE_FETCH(RA, RB, KT)                                                          \
  {                                                                                     \
    _Pragma("unroll") for (int i = 0; i < ALoader::NCH; ++i) RA[i] = la.load(i, (KT) * Cfg::BK); \
    _Pragma("unroll") for (int i = 0; i < BLoader::NCH; ++i) RB[i] = lb.load(i, (KT) * Cfg::BK); \
  }
#define MUSE_STAGE(RA, RB, DA, DB)                                                      \
  {                                                                                     \
    _Pragma("unroll") for (int i = 0; i < ALoader::NCH; ++i) *(u32x4*)((DA) + ALoader::lds_off(i)) = RA[i]; \
    _Pragma("unroll") for (int i = 0; i < BLoader::NCH; ++i) *(u32x4*)((DB) + BLoader::lds_off(i)) = RB[i]; \
  }
    // The tile count is rounded up to even: a tile beyond K is all out-of-range chunks (zeros, no memory traffic), so the
    // loop body is two symmetric, unconditional half-steps.
    const int nk2 = kt0 + ((nk - kt0 + 1) & ~1);
    MUSE_FETCH(ra0, rb0, kt0)
    MUSE_FETCH(ra1, rb1, kt0 + 1)
    MUSE_STAGE(ra0, rb0, tA, tB)
    __syncthreads();
    for (int kt = kt0; kt < nk2; kt += 2) {
      // even half: compute stage 0 (tile kt); r1 holds tile kt+1 (issued one whole half-step ago); refill r0 with kt+2
      MUSE_FETCH(ra0, rb0, kt + 2)
      __builtin_amdgcn_sched_barrier(0);  // keep the prefetch at the top of the half-step (a full step of latency cover)
      compute(tA, tB);
      MUSE_STAGE(ra1, rb1, tA1, tB1)
      __syncthreads();
      // odd half: compute stage 1 (tile kt+1); r0 holds tile kt+2; refill r1 with kt+3
      MUSE_FETCH(ra1, rb1, kt + 3)
      __builtin_amdgcn_sched_barrier(0);
      compute(tA1, tB1);
      MUSE_STAGE(ra0, rb0, tA, tB)
      __syncthreads();
    }
#undef MUSE_FETCH
#undef MUSE_STAGE
  }

  // ---- epilogue: lane holds C[m][n..n+3], m = m0+wr+16i+(lane&15), n = n0+wc+16j+4*(lane>>4) ----
  // Fast path: the C tile goes through LDS (free now) so that global stores are whole 16-byte chunks of contiguous rows
  // (a wave writes 1 KiB of full lines per instruction instead of sixteen 32-byte row fragments); residual / accumulate
  // operands are read the same way.  Needs 16-byte-aligned rows; anything else takes the direct path below.
  if constexpr (BM == 128) {
    constexpr int EPC = 16 / (int)sizeof(TC);               // elements per 16-byte chunk
    constexpr int CST = 128 * (int)sizeof(TC) + 16;         // LDS row stride of the staged tile
    const bool fast = !atomic_out && (p.N % EPC) == 0 && (p.ldc % EPC) == 0 && ((((uintptr_t)Cp) & 15) == 0) &&
                      (p.residual == nullptr || (((p.ldr % EPC) == 0) && ((((uintptr_t)p.residual) & 15) == 0)));
    if (fast) {
      // f32 tiles are staged 64 rows at a time so the epilogue never needs more LDS than the operand stages did
      constexpr int NH = sizeof(TC) == 4 ? 2 : 1, RPP = 128 / NH;  // passes, rows per pass
      constexpr int CPRo = 128 / EPC;                               // 16-byte chunks per tile row
      const TC* Rq = (const TC*)p.residual;
#pragma unroll
      for (int half = 0; half < NH; ++half) {
        __syncthreads();  // operand stages (first pass) / previous staged rows (second pass) are no longer read
        if (wr / RPP == half) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int ml = wr + i * 16 + (lane & 15);
            const float rv = (p.rowvec && (m0 + ml) < p.M) ? p.rowvec[m0 + ml] : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int nl = wc + j * 16 + 4 * (lane >> 4);
              float v[4];
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                float add = rv;
                if (p.bias && (n0 + nl + r) < p.N) add += p.bias[n0 + nl + r];
                float x = p.alpha * acc[i][j][r] + add;
                if (p.act == 1) x = gelu_erf(x);
                v[r] = x;
              }
              OutVec<TC>::store4((TC*)(smem + (ml - half * RPP) * CST) + nl, v);
            }
          }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < (RPP * CPRo) / 256; ++it) {
          const int c = threadIdx.x + 256 * it;
          const int row = c / CPRo, col = (c % CPRo) * EPC;
          const int m = m0 + half * RPP + row, n = n0 + col;
          if (m < p.M && n < p.N) {
            u32x4 w = *(const u32x4*)(smem + row * CST + col * (int)sizeof(TC));
            if (Rq || p.accumulate) {
              float o[EPC];
              if constexpr (sizeof(TC) == 4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = __uint_as_float(w[e]);
              } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) { o[2 * e] = __uint_as_float(w[e] << 16); o[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
              }
              auto add16 = [&](const TC* src) {
                const u32x4 t = *(const u32x4*)src;
                if constexpr (sizeof(TC) == 4) {
#pragma unroll
                  for (int e = 0; e < 4; ++e) o[e] += __uint_as_float(t[e]);
                } else {
#pragma unroll
                  for (int e = 0; e < 4; ++e) { o[2 * e] += __uint_as_float(t[e] << 16); o[2 * e + 1] += __uint_as_float(t[e] & 0xffff0000u); }
                }
              };
              if (Rq) add16(Rq + (long)m * p.ldr + n);
              if (p.accumulate) add16(Cp + (long)m * p.ldc + n);
              if constexpr (sizeof(TC) == 4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = __float_as_uint(o[e]);
              } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = pack2_bf16(o[2 * e], o[2 * e + 1]);
              }
            }
            *(u32x4*)(Cp + (long)m * p.ldc + n) = w;
          }
        }
      }
      return;
    }
  }
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
        if (atomic_out) {  // partial sum of one K slice: hardware f32 atomics straight into C
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

template <typename T, typename TC, int AL, int BL, int BM, int NSTAGE, typename ALoader, typename BLoader>
static inline int launch_gemm_st(const GemmParams& p, int batch, hipStream_t stream) {
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + 127) / 128;
  dim3 grid(ntm * ntn, p.split_k > 1 ? p.split_k : 1, batch);
  constexpr size_t tiles = (size_t)NSTAGE * (TileBytes<T, BM>::VALUE + TileBytes<T, 128>::VALUE);
  constexpr size_t ctile = BM == 128 ? (size_t)(sizeof(TC) == 4 ? 64 : 128) * (128 * sizeof(TC) + 16) : 0;  // staged C rows
  constexpr size_t lds = tiles > ctile ? tiles : ctile;
  auto kern = gemm_kernel<T, TC, AL, BL, BM, NSTAGE, ALoader, BLoader>;
  static bool attr_set = false;  // one flag per template instantiation
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(BM * 2), lds, stream, p);
  return (int)hipGetLastError();
}

// bf16 operands: the k-contiguous x k-contiguous kernel runs one stage (3 blocks per CU beat the deeper prefetch:
// 775 vs 650 TFLOP/s on [16448x768]x[6144x768]^T), kernels with a transposing (tr-read) operand run 2 LDS + 2 register
// stages (592 vs 568 dgrad, 467 vs 432 wgrad).  MUSE_GEMM_STAGES=1|2 forces one choice.  f32 operands: always one stage.
static inline int gemm_stages(int al, int bl) {
  static const int st = [] { const char* e = getenv("MUSE_GEMM_STAGES"); return e ? (e[0] == '1' ? 1 : 2) : 0; }();
  if (st) return st;
  return (al == 0 && bl == 0) ? 1 : 2;
}
template <typename T, typename TC, int AL, int BL, int BM, typename ALoader, typename BLoader>
static inline int launch_gemm(const GemmParams& p, int batch, hipStream_t stream) {
  if (p.M <= 0 || p.N <= 0 || batch <= 0) return 0;
  if constexpr (sizeof(T) == 2) {
    if (gemm_stages(AL, BL) == 2) return launch_gemm_st<T, TC, AL, BL, BM, 2, ALoader, BLoader>(p, batch, stream);
  }
  return launch_gemm_st<T, TC, AL, BL, BM, 1, ALoader, BLoader>(p, batch, stream);
}

// big-M problems use the 256 x 128 tile (8 waves): 25 % fewer operand bytes per flop through L2 / LDS
static inline bool use_bm256(const GemmParams& p, int batch) {
  static const int enabled = [] { const char* e = getenv("MUSE_BM256"); return (e && e[0] == '1') ? 1 : 0; }();
  if (!enabled) return false;  // 1 block/CU at 150+ VGPRs: kept behind a switch until it wins the A/B on hardware
  const long tiles256 = (long)((p.M + 255) / 256) * ((p.N + 127) / 128) * batch * (p.split_k > 1 ? p.split_k : 1);
  return p.M >= 1024 && tiles256 >= 512;
}
