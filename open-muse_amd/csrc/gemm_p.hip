// Host side of the persistent 256 x 256 GEMM (gemm256p.h): eligibility, tile queues, launch.  Called from muse_gemm (gemm.hip).
#include "gemm256p.h"
#include "gemm_p.h"
#include <mutex>

namespace {

// One queue-counter slot (64 bytes) per stream: launches on one stream are ordered, and every launch leaves its slot zeroed.
struct Slots {
  std::mutex mu;
  unsigned* base = nullptr;
  hipStream_t streams[64];
  int n = 0;
  int cus = 0;
  bool failed = false;
};
Slots g_slots;

unsigned* counters_for(hipStream_t s, int* cus) {
  std::lock_guard<std::mutex> lk(g_slots.mu);
  if (g_slots.failed) return nullptr;
  if (!g_slots.base) {
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
        hipMalloc((void**)&g_slots.base, 64 * 64) != hipSuccess || hipMemset(g_slots.base, 0, 64 * 64) != hipSuccess) {
      (void)hipGetLastError();
      g_slots.failed = true;
      g_slots.base = nullptr;
      return nullptr;
    }
    g_slots.cus = prop.multiProcessorCount;
  }
  *cus = g_slots.cus;
  for (int i = 0; i < g_slots.n; ++i)
    if (g_slots.streams[i] == s) return g_slots.base + i * 16;
  if (g_slots.n == 64) return nullptr;
  g_slots.streams[g_slots.n] = s;
  return g_slots.base + (g_slots.n++) * 16;
}

template <typename TC>
bool eligible(const GemmParams& p, int la, int lb, int batch) {
  constexpr long E = (long)sizeof(TC);
  if (batch != 1 || p.split_k > 1 || p.act != 0 || p.bias || p.rowvec) return false;
  if (p.K <= 128) return false;                                   // at least two K-tile pairs per output tile
  if (sizeof(TC) == 2 && (p.residual || p.accumulate)) return false;
  if (sizeof(TC) == 4 && ((p.residual && p.accumulate) || ((p.residual || p.accumulate) && p.alpha != 1.f))) return false;
  if (!gemm256_ok<TC>(p, la, lb)) return false;
  // every tensor is addressed as (per-lane offset | out-of-range marker) + scalar tile offset, 32 bits, through a buffer descriptor:
  // keep each below 2 GiB so that marker + tile offset cannot wrap back into range
  const long ra = la == 0 ? p.M : p.K, rb = lb == 0 ? p.N : p.K;
  if ((ra * p.lda + 16) * 2 >= (1L << 31) || (rb * p.ldb + 16) * 2 >= (1L << 31)) return false;
  if (((long)p.M * p.ldc + 16) * E >= (1L << 31)) return false;
  if (p.residual && ((long)p.M * p.ldr + 16) * E >= (1L << 31)) return false;
  return true;
}

template <typename TC, int AL, int BL>
int launch(const GemmParams& p, hipStream_t stream) {
  int cus = 0;
  unsigned* counters = counters_for(stream, &cus);
  if (!counters || cus < 8) return -1;
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
  auto kern = g256p::kernel<TC, AL, BL>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, g256p::LDS_BYTES_P);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(8 * nbx), dim3(512), g256p::LDS_BYTES_P, stream, a);
  return (int)hipGetLastError();
}

template <typename TC>
int launch_l(const GemmParams& p, int la, int lb, hipStream_t s) {
  if (la == 0 && lb == 0) return launch<TC, 0, 0>(p, s);
  if (la == 0 && lb == 1) return launch<TC, 0, 1>(p, s);
  if (la == 1 && lb == 1) return launch<TC, 1, 1>(p, s);
  return launch<TC, 1, 0>(p, s);
}

}  // namespace

// MUSE_G256P = 0: never (launch-per-tile kernel), 1 (default): whenever eligible
bool gemm256p_takes(const GemmParams& p, int la, int lb, int batch, bool f32_out) {
  static const int mode = []() { const char* e = getenv("MUSE_G256P"); return e ? atoi(e) : 1; }();
  if (!mode) return false;
  return f32_out ? eligible<float>(p, la, lb, batch) : eligible<bf16_t>(p, la, lb, batch);
}

int launch_gemm256p(const GemmParams& p, int la, int lb, bool f32_out, hipStream_t stream) {
  return f32_out ? launch_l<float>(p, la, lb, stream) : launch_l<bf16_t>(p, la, lb, stream);
}
