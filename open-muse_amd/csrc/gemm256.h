// 256 x 256 x 64 bf16 GEMM kernel for the large k-contiguous x k-contiguous products (Linear forward, and dX through the
// transposed weight copies).  Ablation of the 128 x 128 kernel on MI355X (scripts/exp/ablate_gemm.hip) showed that at
// 128^2 the per-CU vector-memory path (64 B/clk) and the VGPR->LDS store path (~79 B/clk) each need as many cycles per tile
// as the MFMAs do; a 256^2 tile moves half the bytes per flop through both.
//
//   8 waves (512 threads) as 2 (M) x 4 (N); each wave owns a 128 x 64 block = 8 x 4 MFMA 16x16 fragments (128 acc VGPRs).
//   Operands: buffer_load_dwordx4 one K-tile ahead into 32 VGPRs, then ds_write_b128 into k-contiguous LDS images with the
//   +32 B row pad (conflict-free ds_read_b128).  One 80 KiB stage, one block per CU (two waves per SIMD).
//   Epilogue: alpha / bias / rowvec / GELU in registers, C staged through LDS in row passes, 16-byte row-contiguous global
//   stores with residual / accumulate applied on the way out.
//
// STATUS (round 1): correct (tests/test_gpu_kernels.py::test_gemm_256_tile_kernel with MUSE_GEMM256=1|2) but NOT the default:
// with register staging there is room for only one K-tile of prefetch and one block per CU, so load latency is exposed
// (500 TFLOP/s on [16448x768]x[6144x768]^T vs 810 for the 128^2 kernel; 811 vs 717 on the K=6144 dgrad).  It needs
// direct-to-LDS loads (global_load_lds) with a counted-vmcnt multi-stage ring to pay off - the round-2 work item.
#pragma once
#include "gemm_core.h"

template <typename TC>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(GemmParams p) {
  using T = bf16_t;
  using Cfg = TileCfg<T>;
  constexpr int BM = 256, BN = 256, NT = 512, MI = 8, NI = 4;
  using ALoader = PlainLoader<T, 0, BM, NT>;  // 4 chunks / thread
  using BLoader = PlainLoader<T, 0, BN, NT>;
  constexpr int TA_BYTES = BM * Cfg::KC_STRIDE;  // 40960
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* tA = smem;
  unsigned char* tB = smem + TA_BYTES;

  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN, ntiles = ntm * ntn;
  const int bq = ntiles >> 3, br = ntiles & 7, xcd = blockIdx.x & 7, bi = blockIdx.x >> 3;
  const int tid_ = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + bi;
  constexpr int GM = 4;
  const int grp = tid_ / (GM * ntn), first_m = grp * GM;
  const int gm = min(ntm - first_m, GM), in_grp = tid_ - grp * (GM * ntn);
  const int m0 = (first_m + in_grp % gm) * BM, n0 = (in_grp / gm) * BN;

  const int z = blockIdx.z, zq = z / p.zdiv, zr = z - zq * p.zdiv;
  const T* Ap = (const T*)p.A + zq * p.sA0 + zr * p.sA1;
  const T* Bp = (const T*)p.B + zq * p.sB0 + zr * p.sB1;
  TC* Cp = (TC*)p.C + zq * p.sC0 + zr * p.sC1;

  ALoader la; la.init(Ap, p.lda, p.M, p.K, m0, p);
  BLoader lb; lb.init(Bp, p.ldb, p.N, p.K, n0, p);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = (wave >> 2) * 128, wc = (wave & 3) * 64;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 ra[ALoader::NCH], rb[BLoader::NCH];
  const int nk = (p.K + Cfg::BK - 1) / Cfg::BK;
#pragma unroll
  for (int i = 0; i < ALoader::NCH; ++i) ra[i] = la.load(i, 0);
#pragma unroll
  for (int i = 0; i < BLoader::NCH; ++i) rb[i] = lb.load(i, 0);
#pragma unroll
  for (int i = 0; i < ALoader::NCH; ++i) *(u32x4*)(tA + ALoader::lds_off(i)) = ra[i];
#pragma unroll
  for (int i = 0; i < BLoader::NCH; ++i) *(u32x4*)(tB + BLoader::lds_off(i)) = rb[i];
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const bool more = (kt + 1) < nk;
    if (more) {
#pragma unroll
      for (int i = 0; i < ALoader::NCH; ++i) ra[i] = la.load(i, (kt + 1) * Cfg::BK);
#pragma unroll
      for (int i = 0; i < BLoader::NCH; ++i) rb[i] = lb.load(i, (kt + 1) * Cfg::BK);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 bf[NI];
#pragma unroll
      for (int j = 0; j < NI; ++j) bf[j] = frag_bf16<0, BN>(tB, wc + j * 16, ks, lane);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const bf16x8 af = frag_bf16<0, BM>(tA, wr + i * 16, ks, lane);
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af, acc[i][j], 0, 0, 0);
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

  // ---- epilogue: C staged through LDS, RPP rows per pass, 16-byte row-contiguous stores ----
  constexpr int EPC = 16 / (int)sizeof(TC);
  constexpr int CST = BN * (int)sizeof(TC) + 16;          // 528 (bf16) / 1040 (f32) bytes per staged row
  constexpr int RPP = sizeof(TC) == 2 ? 128 : 64;         // 67.6 KB / 66.6 KB per pass
  constexpr int NPASS = BM / RPP;
  constexpr int CPRo = BN / EPC;
  const TC* Rq = (const TC*)p.residual;
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int rowblk = wr + i * 16;                     // wave-uniform
      if (rowblk / RPP == pass) {
        const int ml = rowblk + (lane & 15);
        const float rv = (p.rowvec && (m0 + ml) < p.M) ? p.rowvec[m0 + ml] : 0.f;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
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
          OutVec<TC>::store4((TC*)(smem + (ml - pass * RPP) * CST) + nl, v);
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < (RPP * CPRo) / NT; ++it) {
      const int c = threadIdx.x + NT * it;
      const int row = c / CPRo, col = (c % CPRo) * EPC;
      const int m = m0 + pass * RPP + row, n = n0 + col;
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
}

// eligibility: bf16 k-contiguous operands, no split-K, 16-byte aligned output rows
template <typename TC>
static inline bool gemm256_ok(const GemmParams& p) {
  constexpr int EPC = 16 / (int)sizeof(TC);
  return p.split_k <= 1 && (p.N % EPC) == 0 && (p.ldc % EPC) == 0 && ((((uintptr_t)p.C) & 15) == 0) &&
         (p.sC0 % EPC) == 0 && (p.sC1 % EPC) == 0 &&
         (p.residual == nullptr || (((p.ldr % EPC) == 0) && ((((uintptr_t)p.residual) & 15) == 0)));
}

// Pick the tile by estimated time: rounds of resident blocks x per-block cost.  The 128^2 kernel keeps 3 blocks per CU
// (768 slots), the 256^2 kernel 1 block per CU (256 slots) with 4x the work per block; R256/R128 is the measured
// throughput ratio of fully occupied rounds.
static inline bool gemm256_preferred(const GemmParams& p, int batch) {
  static const int mode = [] { const char* e = getenv("MUSE_GEMM256"); return e ? (e[0] - '0') : 0; }();  // 0 off (default), 1 force, 2 auto
  if (mode == 0) return false;
  if (p.K < 128 || p.M < 512 || p.N < 256) return false;
  if (mode == 1) return true;
  const long t128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128) * batch;
  const long t256 = (long)((p.M + 255) / 256) * ((p.N + 255) / 256) * batch;
  const double rounds128 = (double)((t128 + 767) / 768), rounds256 = (double)((t256 + 255) / 256);
  const double cost128 = rounds128 * 1.0, cost256 = rounds256 * (4.0 / 3.0) / 1.35;  // per-round time in 128^2-round units
  return cost256 < cost128;
}

template <typename TC>
static inline int launch_gemm256(const GemmParams& p, int batch, hipStream_t stream) {
  const int ntm = (p.M + 255) / 256, ntn = (p.N + 255) / 256;
  constexpr size_t tiles = 2 * 256 * TileCfg<bf16_t>::KC_STRIDE;
  constexpr size_t ctile = (size_t)(sizeof(TC) == 2 ? 128 : 64) * (256 * sizeof(TC) + 16);
  constexpr size_t lds = tiles > ctile ? tiles : ctile;
  auto kern = gemm256_kernel<TC>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(ntm * ntn, 1, batch), dim3(512), lds, stream, p);
  return (int)hipGetLastError();
}
