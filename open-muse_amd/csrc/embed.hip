// Embedding backward (muse/modeling_transformer.py:942-957 under autograd: index_add of the [T, H] gradient into the [V, H] table)
// as a deterministic SORT-BASED segmented sum:
//   1. stable radix sort of (token id, position) pairs            rocprim::radix_sort_pairs on 32-bit keys (positions ascend inside an id)
//   2. row starts by binary search, segment table by one block scan: a row with n hits is cut into ceil(n / 64) segments, so the
//      mask token's row (half of all tokens) is shared by ~125 blocks instead of serialising one
//   3. one block per segment sums its <= 64 gradient rows in position order (4 rows in flight) -> partial[segment][H]
//   4. one block per table row adds its segments in order (+ the old gradient when accumulating)
// Every gradient row is read exactly once (50 MB at config B); no float atomics, the summation order is fixed.
// The previous kernel (rowops.hip: one block per (table row, token split), every block scanning all ids of its split) took
// 608 us per step on MI355X for this 50 MB gather.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include "common.h"
#include "../../include/muse_hip.h"

namespace emb {
constexpr int SEG = 64;

__global__ void prep_kernel(const int64_t* __restrict__ ids, unsigned* __restrict__ keys, unsigned* __restrict__ vals, int ntok, int vocab) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < ntok) {
    const int64_t id = ids[t];
    keys[t] = (id >= 0 && id < vocab) ? (unsigned)id : (unsigned)vocab;   // out-of-range ids sort behind every row and are ignored
    vals[t] = (unsigned)t;
  }
}

// start[v] = first sorted slot with key >= v (v = 0..vocab); segfirst[v] = number of segments of the rows before v
__global__ __launch_bounds__(1024) void table_kernel(const unsigned* __restrict__ keys, int ntok, int vocab, int* __restrict__ start,
                                                     int* __restrict__ segfirst) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  for (int v = threadIdx.x; v <= vocab; v += 1024) {
    int lo = 0, hi = ntok;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[mid] < (unsigned)v) lo = mid + 1; else hi = mid; }
    start[v] = lo;
  }
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();   // (start[] written by this block, read below by other threads of it)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int base = 0; base <= vocab; base += 1024) {
    const int v = base + threadIdx.x;
    int n = 0;
    if (v < vocab) n = (start[v + 1] - start[v] + SEG - 1) / SEG;
    int incl = n;   // inclusive scan inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(incl, o, 64); if (lane >= o) incl += y; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    const int carry = carry_s;
    if (v <= vocab) segfirst[v] = carry + woff + incl - n;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + woff + incl;
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void segment_kernel(const unsigned* __restrict__ order, const int* __restrict__ start,
                                                      const int* __restrict__ segfirst, const float* __restrict__ dout,
                                                      float* __restrict__ partial, int hidden, int vocab) {
  const int b = blockIdx.x;
  if (b >= segfirst[vocab]) return;
  int lo = 0, hi = vocab;          // largest v with segfirst[v] <= b (rows without hits share their successor's value: skip them)
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (segfirst[mid] <= b) lo = mid; else hi = mid; }
  const int v = lo;
  const int p0 = start[v] + (b - segfirst[v]) * SEG, p1 = min(start[v + 1], p0 + SEG);
  float acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  const int ncol = (hidden - (int)threadIdx.x + 255) / 256;
  int h = p0;
  for (; h + 4 <= p1; h += 4) {   // four rows in flight, added in position order
    const float* s0 = dout + (long)order[h] * hidden;
    const float* s1 = dout + (long)order[h + 1] * hidden;
    const float* s2 = dout + (long)order[h + 2] * hidden;
    const float* s3 = dout + (long)order[h + 3] * hidden;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (j < ncol) {
        const int c = threadIdx.x + j * 256;
        const float a0 = s0[c], a1 = s1[c], a2 = s2[c], a3 = s3[c];
        acc[j] = (((acc[j] + a0) + a1) + a2) + a3;
      }
    }
  }
  for (; h < p1; ++h) {
    const float* src = dout + (long)order[h] * hidden;
#pragma unroll
    for (int j = 0; j < 16; ++j) if (j < ncol) acc[j] += src[threadIdx.x + j * 256];
  }
  float* dst = partial + (long)b * hidden;
#pragma unroll
  for (int j = 0; j < 16; ++j) { const int c = threadIdx.x + j * 256; if (c < hidden) dst[c] = acc[j]; }
}

__global__ __launch_bounds__(256) void row_kernel(const int* __restrict__ segfirst, const float* __restrict__ partial,
                                                  float* __restrict__ dword, int hidden, int accumulate) {
  const int v = blockIdx.x, s0 = segfirst[v], s1 = segfirst[v + 1];
  for (int c = threadIdx.x; c < hidden; c += 256) {
    float s = 0.f;
    for (int k = s0; k < s1; ++k) s += partial[(long)k * hidden + c];
    float* d = dword + (long)v * hidden + c;
    *d = accumulate ? *d + s : s;
  }
}

__global__ void pos_kernel(const float* __restrict__ dout, float* __restrict__ dpos, int batch, int seq, int hidden, int acc) {
  const int s = blockIdx.x;
  for (int c = threadIdx.x; c < hidden; c += blockDim.x) {
    float a = 0.f;
    for (int b = 0; b < batch; ++b) a += dout[((long)b * seq + s) * hidden + c];
    float* d = dpos + (long)s * hidden + c;
    *d = acc ? *d + a : a;
  }
}

static inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }
struct Layout { size_t keys_in, keys_out, vals_in, vals_out, start, segfirst, partial, temp, temp_bytes, total; int maxseg; };
static inline int key_bits(int vocab) { int b = 1; while ((1 << b) <= vocab) ++b; return b; }
static int layout(int ntok, int hidden, int vocab, Layout& L) {
  size_t tb = 0;
  const hipError_t e = rocprim::radix_sort_pairs((void*)nullptr, tb, (const unsigned*)nullptr, (unsigned*)nullptr, (const unsigned*)nullptr,
                                                 (unsigned*)nullptr, (size_t)ntok, 0u, (unsigned)key_bits(vocab), (hipStream_t)0);
  if (e != hipSuccess) return (int)e;
  L.maxseg = vocab + ntok / SEG + 1;
  size_t o = 0;
  L.keys_in = o; o += al((size_t)ntok * 4);
  L.keys_out = o; o += al((size_t)ntok * 4);
  L.vals_in = o; o += al((size_t)ntok * 4);
  L.vals_out = o; o += al((size_t)ntok * 4);
  L.start = o; o += al((size_t)(vocab + 1) * 4);
  L.segfirst = o; o += al((size_t)(vocab + 1) * 4);
  L.partial = o; o += al((size_t)L.maxseg * hidden * 4);
  L.temp = o; L.temp_bytes = tb; o += al(tb);
  L.total = o;
  return 0;
}
}  // namespace emb

extern "C" int64_t muse_embed_bwd2_scratch_bytes(int32_t batch, int32_t seq, int32_t hidden, int32_t vocab) {
  emb::Layout L;
  if (batch <= 0 || seq <= 0) return 0;
  if (emb::layout(batch * seq, hidden, vocab, L)) return -1;
  return (int64_t)L.total;
}

extern "C" int muse_embed_bwd2(const int64_t* ids, const float* dout, float* dword, float* dpos, void* scratch, int64_t scratch_bytes,
                               int32_t batch, int32_t seq, int32_t hidden, int32_t vocab, int32_t accumulate, void* stream) {
  if (hidden > 4096 || vocab <= 0) return MUSE_ERR_UNSUPPORTED;
  const int ntok = batch * seq;
  if (ntok <= 0) return 0;
  emb::Layout L;
  if (int e = emb::layout(ntok, hidden, vocab, L)) return e;
  if ((size_t)scratch_bytes < L.total || (((uintptr_t)scratch) & 255)) return MUSE_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  unsigned char* base = (unsigned char*)scratch;
  unsigned* keys_in = (unsigned*)(base + L.keys_in);
  unsigned* keys_out = (unsigned*)(base + L.keys_out);
  unsigned* vals_in = (unsigned*)(base + L.vals_in);
  unsigned* vals_out = (unsigned*)(base + L.vals_out);
  int* start = (int*)(base + L.start);
  int* segfirst = (int*)(base + L.segfirst);
  float* partial = (float*)(base + L.partial);
  hipLaunchKernelGGL(emb::prep_kernel, dim3((ntok + 255) / 256), dim3(256), 0, s, ids, keys_in, vals_in, ntok, vocab);
  size_t tb = L.temp_bytes;
  const hipError_t e = rocprim::radix_sort_pairs((void*)(base + L.temp), tb, (const unsigned*)keys_in, keys_out, (const unsigned*)vals_in, vals_out,
                                                 (size_t)ntok, 0u, (unsigned)emb::key_bits(vocab), s);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(emb::table_kernel, dim3(1), dim3(1024), 0, s, (const unsigned*)keys_out, ntok, vocab, start, segfirst);
  hipLaunchKernelGGL(emb::segment_kernel, dim3(L.maxseg), dim3(256), 0, s, (const unsigned*)vals_out, (const int*)start, (const int*)segfirst,
                     dout, partial, hidden, vocab);
  hipLaunchKernelGGL(emb::row_kernel, dim3(vocab), dim3(256), 0, s, (const int*)segfirst, (const float*)partial, dword, hidden, accumulate);
  hipLaunchKernelGGL(emb::pos_kernel, dim3(seq), dim3(256), 0, s, dout, dpos, batch, seq, hidden, accumulate);
  return (int)hipGetLastError();
}
