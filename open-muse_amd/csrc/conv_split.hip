// f32 convolution on the bf16 matrix cores by operand splitting ("bf16x3"):  x = x_hi + x_lo with x_hi = bf16(x),
// x_lo = bf16(x - x_hi), and  a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  accumulated in f32 (the dropped a_lo*b_lo term is
// <= 2^-16 |a||b|).  Three v_mfma_f32_16x16x32_bf16 per product at 16x the f32-MFMA rate: an f32-class (better than
// TF32: ~16 vs 10 explicit mantissa bits) implicit-GEMM convolution for the frozen VQGAN.  The reference's own GPU path
// runs these convolutions in TF32 (torch.backends.cudnn.allow_tf32 defaults to True; configs/imagenet.yaml:86
// enable_tf32), gfx950 has no xf32 MFMA, so this is the CDNA4 counterpart.
//
// Same structure as gemm_kernel (gemm_core.h): 128 x 128 x 64 tile, 4 waves, f32 activations gathered NHWC
// (ConvLoader<float>) one K-tile ahead into VGPRs, split to hi/lo bf16 on the way into LDS; weights are pre-split on the
// host into two bf16 [Cout][KS*KS*Cin] images.
#include "gemm_core.h"
#include "../../include/muse_hip.h"

struct SplitParams {
  GemmParams g;      // A = f32 NHWC input, B = weight hi image, C = f32 output
  const void* Blo;   // weight lo image
  double* gn_partial;  // optional [B, H*W/128, gn_groups, 2] sum / sum-of-squares of the output (next GroupNorm's statistics)
  int gn_groups, gn_cpg;
};

__global__ __launch_bounds__(256, 2) void conv_split_kernel(SplitParams sp) {
  const GemmParams& p = sp.g;
  constexpr int BK = 64, IMG = 128 * 160;  // one bf16 [128][64] k-contiguous image, +32 B row pad
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* tAh = smem;
  unsigned char* tAl = smem + IMG;
  unsigned char* tBh = smem + 2 * IMG;
  unsigned char* tBl = smem + 3 * IMG;

  const int ntm = (p.M + 127) >> 7, ntn = (p.N + 127) >> 7, ntiles = ntm * ntn;
  const int bq = ntiles >> 3, br = ntiles & 7, xcd = blockIdx.x & 7, bi = blockIdx.x >> 3;
  const int tid_ = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + bi;
  const int m0 = (tid_ / ntn) << 7, n0 = (tid_ % ntn) << 7;

  using ALoader = ConvLoader<float, 128, 256>;           // 8 chunks of 4 floats per thread per K-tile
  using BLoader = PlainLoader<bf16_t, 0, 128, 256>;      // 4 chunks of 8 bf16 per thread per K-tile and image
  ALoader la; la.init(p.A, 0, p.M, p.K, m0, p);
  BLoader lh; lh.init(p.B, p.ldb, p.N, p.K, n0, p);
  BLoader ll; ll.init(sp.Blo, p.ldb, p.N, p.K, n0, p);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 ra[ALoader::NCH], rh[BLoader::NCH], rl[BLoader::NCH];
  const int nk = (p.K + BK - 1) / BK;

  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < ALoader::NCH; ++i) ra[i] = la.load(i, k0);
#pragma unroll
    for (int i = 0; i < BLoader::NCH; ++i) { rh[i] = lh.load(i, k0); rl[i] = ll.load(i, k0); }
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < ALoader::NCH; ++i) {
      const int c = threadIdx.x + 256 * i;           // 16 four-float chunks per row
      const int off = (c >> 4) * 160 + (c & 15) * 8;
      u32x2 hi, lo;
      split4(ra[i], hi, lo);
      *(u32x2*)(tAh + off) = hi;
      *(u32x2*)(tAl + off) = lo;
    }
#pragma unroll
    for (int i = 0; i < BLoader::NCH; ++i) {
      *(u32x4*)(tBh + BLoader::lds_off(i)) = rh[i];
      *(u32x4*)(tBl + BLoader::lds_off(i)) = rl[i];
    }
  };

  fetch(0);
  stage();
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = (kt + 1) < nk;
    if (more) fetch((kt + 1) * BK);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 ah[4], al[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ah[i] = frag_bf16<0, 128>(tAh, wr + i * 16, ks, lane);
        al[i] = frag_bf16<0, 128>(tAl, wr + i * 16, ks, lane);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bf16x8 bh = frag_bf16<0, 128>(tBh, wc + j * 16, ks, lane);
        const bf16x8 bl = frag_bf16<0, 128>(tBl, wc + j * 16, ks, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah[i], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al[i], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah[i], acc[i][j], 0, 0, 0);
        }
      }
    }
    __syncthreads();
    if (more) {
      stage();
      __syncthreads();
    }
  }

  // epilogue (f32): bias per output channel, optional residual, 16-byte stores
  float* Cp = (float*)p.C;
  const float* Rp = (const float*)p.residual;
  const bool vec_ok = ((p.ldc & 3) == 0) && ((((uintptr_t)Cp) & 15) == 0) &&
                      (Rp == nullptr || (((p.ldr & 3) == 0) && ((((uintptr_t)Rp) & 15) == 0)));
  double gs[4] = {0.0, 0.0, 0.0, 0.0}, gq[4] = {0.0, 0.0, 0.0, 0.0};   // per column fragment j: this lane's 4 channels x 4 rows
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wr + i * 16 + (lane & 15);
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wc + j * 16 + 4 * (lane >> 4);
      if (n >= p.N) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + ((p.bias && (n + r) < p.N) ? p.bias[n + r] : 0.f);
      float* cptr = Cp + (long)m * p.ldc + n;
      if (vec_ok && (n + 3) < p.N) {
        if (Rp) { float t[4]; OutVec<float>::load4(Rp + (long)m * p.ldr + n, t); v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3]; }
        OutVec<float>::store4(cptr, v);
        if (sp.gn_partial) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { gs[j] += (double)v[r]; gq[j] += (double)v[r] * (double)v[r]; }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if ((n + r) < p.N) cptr[r] = v[r] + (Rp ? Rp[(long)m * p.ldr + n + r] : 0.f);
      }
    }
  }
  // GroupNorm partials (host guarantees H*W % 128 == 0, N % 4 == 0, 4 | cpg: the tile's 128 pixels are one image and a lane's
  // 4 channels one group).  Fold the 16 row-lanes of each 4-channel chunk, the two row-halves (waves) through LDS, then the
  // chunks of each group -- all in a fixed order.
  if (sp.gn_partial) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { gs[j] += __shfl_xor(gs[j], o, 64); gq[j] += __shfl_xor(gq[j], o, 64); }
    double* red = (double*)smem;   // [2 row halves][32 chunks][2]; the operand stages are idle (loop ended on a barrier)
    if ((lane & 15) == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int chunk = (wc >> 2) + j * 4 + (lane >> 4);
        red[((wave >> 1) * 32 + chunk) * 2] = gs[j];
        red[((wave >> 1) * 32 + chunk) * 2 + 1] = gq[j];
      }
    }
    __syncthreads();
    if (threadIdx.x < 32) {
      double cs = red[threadIdx.x * 2] + red[(32 + threadIdx.x) * 2];
      double cq = red[threadIdx.x * 2 + 1] + red[(32 + threadIdx.x) * 2 + 1];
      const int cpc = sp.gn_cpg >> 2;   // column chunks per group (1, 2, 4, 8)
      for (int o = 1; o < cpc; o <<= 1) { cs += __shfl_xor(cs, o, 64); cq += __shfl_xor(cq, o, 64); }
      const int n = n0 + (int)threadIdx.x * 4;
      if ((threadIdx.x & (cpc - 1)) == 0 && n < p.N) {
        const int hw = p.cH * p.cW, b = m0 / hw, chunk = (m0 - b * hw) >> 7, nchunk = hw >> 7;
        double* o2 = sp.gn_partial + (((long)b * nchunk + chunk) * sp.gn_groups + n / sp.gn_cpg) * 2;
        o2[0] = cs; o2[1] = cq;
      }
    }
  }
}

extern "C" int muse_conv2d_nhwc_split(const float* in, const void* w_hi, const void* w_lo, const float* bias,
                                      const float* residual, float* out, double* gn_partial, int32_t gn_groups, int32_t batch,
                                      int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t KS, int32_t upsample,
                                      void* stream) {
  if (Cin % 8) return MUSE_ERR_ALIGN;  // bf16 weight rows in 16-byte chunks
  if ((((uintptr_t)in) & 15) || (((uintptr_t)w_hi) & 15) || (((uintptr_t)w_lo) & 15)) return MUSE_ERR_ALIGN;
  if (KS != 1 && KS != 3) return MUSE_ERR_UNSUPPORTED;
  if (upsample < 0 || upsample > 2 || (upsample == 2 && KS != 3)) return MUSE_ERR_BAD_ARG;
  if (upsample == 1 && ((H | W) & 1)) return MUSE_ERR_BAD_ARG;
  SplitParams sp;
  sp.gn_partial = nullptr; sp.gn_groups = 0; sp.gn_cpg = 0;
  if (gn_partial) {   // output statistics for the next GroupNorm, [batch, H*W/128, gn_groups, 2] doubles
    const int cpg = gn_groups > 0 ? Cout / gn_groups : 0;
    if (gn_groups <= 0 || (Cout % gn_groups) || (cpg & 3) || cpg > 32 || (cpg & (cpg - 1)) || ((H * W) & 127) || (Cout & 3) ||
        (((uintptr_t)out) & 15) || (((uintptr_t)residual) & 15))
      return MUSE_ERR_UNSUPPORTED;
    sp.gn_partial = gn_partial; sp.gn_groups = gn_groups; sp.gn_cpg = cpg;
  }
  GemmParams& p = sp.g;
  p.A = in; p.B = w_hi; p.C = out; sp.Blo = w_lo;
  p.bias = bias; p.rowvec = nullptr; p.residual = residual;
  p.M = batch * H * W; p.N = Cout; p.K = KS * KS * Cin;
  p.lda = 0; p.ldb = p.K; p.ldc = Cout; p.ldr = Cout;
  p.zdiv = 1; p.sA0 = p.sA1 = p.sB0 = p.sB1 = p.sC0 = p.sC1 = 0;
  p.alpha = 1.0f; p.accumulate = 0; p.act = 0; p.split_k = 1; p.split_stride = 0;
  p.cH = H; p.cW = W; p.cCin = Cin; p.cKS = KS; p.cUps = upsample;
  p.cCinShift = -1;
  if ((Cin & (Cin - 1)) == 0) { int sh = 0; while ((1 << sh) < Cin) ++sh; p.cCinShift = sh; }
  if (p.M <= 0 || p.N <= 0) return 0;
  const int ntm = (p.M + 127) / 128, ntn = (p.N + 127) / 128;
  const size_t lds = 4 * 128 * 160;
  (void)hipFuncSetAttribute((const void*)conv_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(conv_split_kernel, dim3(ntm * ntn), dim3(256), lds, (hipStream_t)stream, sp);
  return (int)hipGetLastError();
}
