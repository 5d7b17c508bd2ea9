// Ablation harness for the NN bf16 GEMM main loop (not part of the product library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I open-muse_amd/csrc scripts/exp/ablate_gemm.hip -o /tmp/ablate && /tmp/ablate
#include "gemm_core.h"
#include <cstdio>
#include <vector>

template <int MODE>  // 0 full, 1 no global loads (stale regs), 2 no LDS writes either, 3 no barriers, 4 no ds_read (frags hoisted), 5 mfma only
__global__ __launch_bounds__(256, 2) void abl_kernel(GemmParams p) {
  using T = bf16_t; using Cfg = TileCfg<T>;
  using ALoader = PlainLoader<T, 0, 128, 256>; using BLoader = PlainLoader<T, 0, 128, 256>;
  constexpr int TA_BYTES = TileBytes<T, 128>::VALUE;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* tA = smem; unsigned char* tB = smem + TA_BYTES;
  const int ntn = (p.N + 127) >> 7;
  const int m0 = (blockIdx.x / ntn) << 7, n0 = (blockIdx.x % ntn) << 7;
  ALoader la; la.init(p.A, p.lda, p.M, p.K, m0, p);
  BLoader lb; lb.init(p.B, p.ldb, p.N, p.K, n0, p);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64;
  f32x4 acc[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 ra[4], rb[4];
  const int nk = p.K / 64;
  for (int i = 0; i < 4; ++i) { ra[i] = la.load(i, 0); rb[i] = lb.load(i, 0); }
  for (int i = 0; i < 4; ++i) { *(u32x4*)(tA + ALoader::lds_off(i)) = ra[i]; *(u32x4*)(tB + BLoader::lds_off(i)) = rb[i]; }
  __syncthreads();
  bf16x8 haf[2][4], hbf[2][4];
  if (MODE >= 4) for (int ks = 0; ks < 2; ++ks) for (int i = 0; i < 4; ++i) { haf[ks][i] = frag_bf16<0, 128>(tA, wr + i * 16, ks, lane); hbf[ks][i] = frag_bf16<0, 128>(tB, wc + i * 16, ks, lane); }
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = (kt + 1) < nk;
    if (MODE == 0 && more) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { ra[i] = la.load(i, (kt + 1) * 64); rb[i] = lb.load(i, (kt + 1) * 64); }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (MODE >= 4) { af[i] = haf[ks][i]; bf[i] = hbf[ks][i]; }
        else { af[i] = frag_bf16<0, 128>(tA, wr + i * 16, ks, lane); bf[i] = frag_bf16<0, 128>(tB, wc + i * 16, ks, lane); }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
    }
    if (MODE < 3) __syncthreads();
    if (more && MODE < 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { *(u32x4*)(tA + ALoader::lds_off(i)) = ra[i]; *(u32x4*)(tB + BLoader::lds_off(i)) = rb[i]; }
    }
    if (MODE < 3) __syncthreads();
  }
  // minimal epilogue so nothing is dead: one value per lane
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  ((float*)p.C)[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE> static void run(const char* name, GemmParams p) {
  const int ntm = (p.M + 127) / 128, ntn = (p.N + 127) / 128;
  const size_t lds = 2 * TileBytes<bf16_t, 128>::VALUE;
  hipFuncSetAttribute((const void*)abl_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(abl_kernel<MODE>, dim3(ntm * ntn), dim3(256), lds, 0, p);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(abl_kernel<MODE>, dim3(ntm * ntn), dim3(256), lds, 0, p);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  printf("%-34s K=%5d: %8.1f us  %7.1f TFLOP/s\n", name, p.K, ms * 1e3, 2.0 * p.M * p.N * p.K / ms / 1e9);
}

int main() {
  const int M = 16384, N = 6144;
  for (int K : {768, 3072}) {
    void *A, *B, *C;
    hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)N * K * 2); hipMalloc(&C, (size_t)M * N * 2);
    std::vector<unsigned short> h((size_t)M * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (unsigned short)((i * 2654435761u) >> 24);  // bf16 values near 0.0078..
    hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice);
    hipMemcpy(B, h.data(), (size_t)N * K * 2 < h.size() * 2 ? (size_t)N * K * 2 : h.size() * 2, hipMemcpyHostToDevice);
    GemmParams p{}; p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = K; p.ldb = K; p.ldc = N; p.zdiv = 1; p.alpha = 1.f; p.split_k = 1;
    run<0>("full (1 stage, toy epilogue)", p);
    run<1>("no global loads", p);
    run<2>("no global loads, no LDS writes", p);
    run<3>("+ no barriers", p);
    run<4>("+ no ds_read (frags hoisted)", p);
    hipFree(A); hipFree(B); hipFree(C);
  }
  return 0;
}
