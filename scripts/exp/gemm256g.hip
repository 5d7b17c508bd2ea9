// Stand-alone harness for the 256 x 256 LDS-DMA GEMM kernel (open-muse_amd/csrc/gemm256.h): correctness of every operand
// layout against a naive kernel (non-symmetric random data, ragged M / N / K, split-K) and timing on the transformer shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -I open-muse_amd/csrc scripts/exp/gemm256g.hip -o /tmp/g256g && /tmp/g256g
#include "gemm256.h"
#include <cstdio>
#include <cstring>
#include <vector>
#include <cmath>

__global__ void naive_kernel(const bf16_t* A, const bf16_t* B, float* C, int M, int N, int K, long lda, long ldb, int la, int lb) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
  if (n >= N) return;
  float s = 0.f;
  for (int k = 0; k < K; ++k) {
    const float a = bf16_to_f32(la == 0 ? A[(long)m * lda + k] : A[(long)k * lda + m]);
    const float b = bf16_to_f32(lb == 0 ? B[(long)n * ldb + k] : B[(long)k * ldb + n]);
    s += a * b;
  }
  C[(long)m * N + n] = s;
}

static unsigned short f2b(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return u >> 16; }
static float b2f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }

static GemmParams make_params(const void* A, const void* B, void* C, int M, int N, int K, long lda, long ldb, long ldc) {
  GemmParams p; memset(&p, 0, sizeof(p));
  p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.zdiv = 1; p.alpha = 1.f;
  p.split_k = 1; p.cCinShift = -1;
  return p;
}

template <typename TC> static void run(int M, int N, int K, int la, int lb, bool check, int split_k = 1) {
  static const int pad = getenv("PAD") ? atoi(getenv("PAD")) : 0;
  const long lda = (la == 0 ? K : M) + pad, ldb = (lb == 0 ? K : N) + pad;
  const size_t na = (size_t)(la == 0 ? M : K) * lda, nb = (size_t)(lb == 0 ? N : K) * ldb;
  std::vector<unsigned short> hA(na), hB(nb);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (auto& v : hA) v = f2b(rnd());
  for (auto& v : hB) v = f2b(rnd() * 0.1f);
  bf16_t *A, *B; TC* C; float* R;
  const size_t cslice = (size_t)M * N;
  hipMalloc(&A, na * 2); hipMalloc(&B, nb * 2); hipMalloc(&C, cslice * sizeof(TC) * split_k);
  hipMemcpy(A, hA.data(), na * 2, hipMemcpyHostToDevice); hipMemcpy(B, hB.data(), nb * 2, hipMemcpyHostToDevice);
  hipMemset(C, 0xff, cslice * sizeof(TC) * split_k);
  GemmParams p = make_params(A, B, C, M, N, K, lda, ldb, N);
  p.split_k = split_k; p.split_stride = split_k > 1 ? (long)cslice : 0;
  if (!gemm256_ok<TC>(p, la, lb)) { printf("not eligible\n"); return; }
  int rc = launch_gemm256<TC>(p, la, lb, 1, 0);
  hipError_t e = hipDeviceSynchronize();
  if (rc || e != hipSuccess) { printf("launch failed: %d %s\n", rc, hipGetErrorString(e)); return; }
  if (check) {
    hipMalloc(&R, cslice * 4);
    hipLaunchKernelGGL(naive_kernel, dim3((N + 255) / 256, M), dim3(256), 0, 0, A, B, R, M, N, K, lda, ldb, la, lb);
    std::vector<float> hR(cslice); std::vector<TC> hC(cslice * split_k);
    hipMemcpy(hR.data(), R, cslice * 4, hipMemcpyDeviceToHost); hipMemcpy(hC.data(), C, hC.size() * sizeof(TC), hipMemcpyDeviceToHost);
    double maxerr = 0, maxref = 0; size_t bad = 0;
    for (size_t i = 0; i < cslice; ++i) {
      double v = 0;
      for (int y = 0; y < split_k; ++y) { if constexpr (sizeof(TC) == 2) v += b2f(hC[y * cslice + i]); else v += hC[y * cslice + i]; }
      double d = fabs(v - hR[i]);
      if (!(d <= maxerr)) { maxerr = d; bad = i; }
      if (fabs(hR[i]) > maxref) maxref = fabs(hR[i]);
    }
    printf("check la=%d lb=%d out=%s M=%d N=%d K=%d sk=%d: max|err| %.4g (max|ref| %.4g) at (%zu,%zu) %s\n", la, lb, sizeof(TC) == 2 ? "bf16" : "f32", M, N, K,
           split_k, maxerr, maxref, bad / N, bad % N, maxerr <= (sizeof(TC) == 2 ? 0.01 : 2e-4) * (maxref + 1) ? "OK" : "FAIL");
    hipFree(R);
  } else {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) launch_gemm256<TC>(p, la, lb, 1, 0);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) launch_gemm256<TC>(p, la, lb, 1, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("la=%d lb=%d out=%s M=%d N=%d K=%d sk=%d: %.1f us  %.1f TFLOP/s\n", la, lb, sizeof(TC) == 2 ? "bf16" : "f32", M, N, K, split_k, ms * 1e3,
           2.0 * M * N * K / ms / 1e9);
  }
  hipFree(A); hipFree(B); hipFree(C);
}

int main(int argc, char** argv) {
  const bool quick = argc > 1 && argv[1][0] == 'q';
  if (argc > 1 && argv[1][0] == 't') {  // timing only (ablation builds: results are wrong by construction)
    run<bf16_t>(16384, 6144, 3072, 0, 0, false);
    run<bf16_t>(16384, 6144, 768, 0, 0, false);
    run<bf16_t>(16384, 768, 6144, 0, 1, false);
    run<float>(6144, 768, 16448, 1, 1, false, 3);
    return 0;
  }
  for (int la = 0; la < 2; ++la)
    for (int lb = 0; lb < 2; ++lb) {
      run<bf16_t>(1000, 520, 512, la, lb, true);
      run<float>(264, 776, 200, la, lb, true);      // ragged K (tail tile + odd tile count), ragged M / N
      run<float>(520, 264, 1160, la, lb, true, 3);  // split-K with an odd slice
    }
  if (quick) return 0;
  run<bf16_t>(16384, 6144, 768, 0, 0, false);
  run<bf16_t>(16384, 6144, 3072, 0, 0, false);
  run<bf16_t>(16384, 768, 3072, 0, 0, false);
  run<bf16_t>(16384, 2304, 768, 0, 0, false);
  run<bf16_t>(16448, 6144, 768, 0, 0, false);
  run<bf16_t>(16384, 768, 6144, 0, 1, false);   // dX = dY W
  run<bf16_t>(16384, 3072, 768, 0, 1, false);
  run<float>(6144, 768, 16448, 1, 1, false, 3);  // dW = dY^T X, split-K workspace
  run<float>(768, 3072, 16448, 1, 1, false, 7);
  return 0;
}
