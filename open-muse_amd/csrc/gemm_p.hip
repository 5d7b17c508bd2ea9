// Host side of the persistent 256 x 256 GEMM (gemm256p.h): eligibility, tile queues, launch.  Called from muse_gemm (gemm.hip).
#include "gemm256p.h"
#include "gemm_p.h"
#include <mutex>
#include <stdio.h>

namespace {

// One queue-counter slot (64 bytes) per (device, stream): launches on one stream are ordered, and every launch leaves its slot
// zeroed.  The table is PER DEVICE (a counter block lives in the memory of the device whose kernels draw tickets from it; a process
// that drives several GPUs gets one block each).  A stream that is being captured into a HIP graph gets no slot: a replayed graph
// and an eager launch on a pooled stream that happens to reuse the handle would share one counter - the caller falls back to the
// launch-per-tile kernel while capturing.
constexpr int kMaxDev = 16;
struct Slots {
  unsigned* base = nullptr;
  hipStream_t streams[64];
  int n = 0;
  int cus = 0;
  bool failed = false;
  bool attr_set[5] = {false, false, false, false, false};
};
std::mutex g_mu;
Slots g_slots[kMaxDev];

Slots* slots_of_current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) { (void)hipGetLastError(); return nullptr; }
  Slots& S = g_slots[dev];
  if (S.failed) return nullptr;
  if (!S.base) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess || hipMalloc((void**)&S.base, 64 * 64) != hipSuccess ||
        hipMemset(S.base, 0, 64 * 64) != hipSuccess) {
      (void)hipGetLastError();
      S.failed = true;
      S.base = nullptr;
      return nullptr;
    }
    S.cus = prop.multiProcessorCount;
  }
  return &S;
}

unsigned* counters_for(hipStream_t s, int* cus, Slots** out) {
  std::lock_guard<std::mutex> lk(g_mu);
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) != hipSuccess) (void)hipGetLastError();
  if (cap != hipStreamCaptureStatusNone) return nullptr;
  Slots* S = slots_of_current_device();
  if (!S) return nullptr;
  *cus = S->cus;
  *out = S;
  for (int i = 0; i < S->n; ++i)
    if (S->streams[i] == s) return S->base + i * 16;
  if (S->n == 64) return nullptr;
  S->streams[S->n] = s;
  return S->base + (S->n++) * 16;
}

// MUSE_G256P_DEBUG=1: say on stderr why a product does not take the persistent kernel
#define P_REJECT(why) do { if (getenv("MUSE_G256P_DEBUG")) fprintf(stderr, "[g256p] %dx%dx%d la%d lb%d out%d: %s\n", p.M, p.N, p.K, la, lb, (int)sizeof(TC), why); return false; } while (0)
template <typename TC>
bool eligible(const GemmParams& p, int la, int lb, int batch) {
  constexpr long E = (long)sizeof(TC);
  if (batch != 1 || p.split_k > 1 || p.act != 0 || p.bias || p.rowvec) P_REJECT("batch / split-K / activation / bias / row vector");
  // instantiated forms: A k-contiguous; bf16 output with either B layout (Linear forward, dX), f32 output with B k-contiguous
  // (forward into the f32 residual stream).  The other layout pairs compile with a few spilled registers and stay with the
  // launch-per-tile kernel.
  if (la != 0 || (sizeof(TC) == 4 && lb != 0)) P_REJECT("operand layout");
  if (p.K <= 128) P_REJECT("K <= 128");                            // at least two K-tile pairs per output tile
  if (sizeof(TC) == 2 && (p.residual || p.accumulate)) P_REJECT("bf16 residual / accumulate");
  if (sizeof(TC) == 4 && ((p.residual && p.accumulate) || ((p.residual || p.accumulate) && p.alpha != 1.f))) P_REJECT("residual + accumulate / alpha");
  if (!gemm256_ok<TC>(p, la, lb)) P_REJECT("gemm256_ok");
  // every tensor is addressed as (per-lane offset | out-of-range marker) + scalar tile offset, 32 bits, through a buffer descriptor:
  // keep each below 2 GiB so that marker + tile offset cannot wrap back into range
  const long ra = la == 0 ? p.M : p.K, rb = lb == 0 ? p.N : p.K;
  if ((ra * p.lda + 16) * 2 >= (1L << 31) || (rb * p.ldb + 16) * 2 >= (1L << 31)) P_REJECT("operand >= 2 GiB");
  if (((long)p.M * p.ldc + 16) * E >= (1L << 31)) P_REJECT("C >= 2 GiB");
  if (p.residual && ((long)p.M * p.ldr + 16) * E >= (1L << 31)) P_REJECT("residual >= 2 GiB");
  return true;
}

template <typename TC, int AL, int BL, bool H = false>
int launch(const GemmParams& p, hipStream_t stream) {
  int cus = 0;
  Slots* S = nullptr;
  unsigned* counters = counters_for(stream, &cus, &S);
  if (!counters || cus < 8) {
    if (getenv("MUSE_G256P_DEBUG")) fprintf(stderr, "[g256p] no queue slot (counters %p, %d CUs)\n", (void*)counters, cus);
    return -1;
  }
  g256p::PArgs a;
  a.g = p;
  a.counters = counters;
  g256p::Sched& sc = a.sched;
  sc.ntm_full = p.M / 256;
  sc.ntn = (p.N + 255) / 256;
  sc.nstrip = (p.M % 256) ? sc.ntn : 0;
  const int nfull = sc.ntm_full * sc.ntn;
  sc.q = nfull / 8; sc.r = nfull % 8;
  int longest = 0;
  for (int x = 0; x < 8; ++x) longest = sc.count(x) > longest ? sc.count(x) : longest;
  const int per_xcd = cus / 8;
  const int nbx = longest < per_xcd ? longest : per_xcd;
  a.dyn = longest > nbx;
  a.nk2 = (((p.K + 63) / 64) + 1) & ~1;
  static const int epi = []() { const char* e = getenv("MUSE_G256P_EPI"); return e ? atoi(e) : 2; }();
  a.epi = epi;
  static const int staux = []() { const char* e = getenv("MUSE_G256P_STAUX"); return e ? atoi(e) : 0; }();
  static const int stagger = []() { const char* e = getenv("MUSE_G256P_STAGGER"); return e ? atoi(e) : 1; }();
  a.staux = staux;
  a.stagger = stagger;
  auto kern = g256p::kernel<TC, AL, BL, H>;
  constexpr int form = H ? 4 : (sizeof(TC) == 4 ? 2 : 0) + BL;      // the dynamic-LDS attribute is per function AND per device
  if (!S->attr_set[form]) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, g256p::LDS_BYTES_P);
    S->attr_set[form] = true;
  }
  hipLaunchKernelGGL(kern, dim3(8 * nbx), dim3(512), g256p::LDS_BYTES_P, stream, a);
  return (int)hipGetLastError();
}

int launch_l(const GemmParams& p, int la, int lb, bool f32_out, hipStream_t s, bool half_ops) {
  if (la != 0) return -1;
  if (half_ops) return (f32_out && lb == 0) ? launch<float, 0, 0, true>(p, s) : -1;
  if (f32_out) return lb == 0 ? launch<float, 0, 0>(p, s) : -1;
  return lb == 0 ? launch<bf16_t, 0, 0>(p, s) : launch<bf16_t, 0, 1>(p, s);
}

}  // namespace

// MUSE_G256P = 0: never (launch-per-tile kernel), 1 (default): whenever eligible
bool gemm256p_takes(const GemmParams& p, int la, int lb, int batch, bool f32_out) {
  const char* e = getenv("MUSE_G256P");   // read per call (cheap) so tests can compare both kernels in one process
  if (e && e[0] == '0') return false;
  return f32_out ? eligible<float>(p, la, lb, batch) : eligible<bf16_t>(p, la, lb, batch);
}

int launch_gemm256p(const GemmParams& p, int la, int lb, bool f32_out, hipStream_t stream, bool half_ops) {
  return launch_l(p, la, lb, f32_out, stream, half_ops);
}
