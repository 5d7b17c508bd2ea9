// What one CU's vector-memory path sustains, alone and with every other CU doing the same (round 6; the ceiling table's inputs).
//   stores : every wave of a 512-thread block stores 128 KiB tiles from registers (16 B per lane, 512-byte rows) to its own region
//   dma    : every wave pulls 1 KiB pieces of an L2-resident region into LDS (buffer_load_dwordx4 ... lds), 64 KiB per round
//   loads  : the same bytes through registers (buffer_load_dwordx4), HBM-resident (each block its own region) or L2-resident
// For G = 1, 8, 32, 64, 128, 256 blocks (one per CU): bytes per shader cycle per CU from s_memtime stamps around the timed loop, and
// the wall-clock aggregate.   hipcc --offload-arch=gfx950 -O3 scripts/exp/vmem_path.hip -o /tmp/vmem_path && /tmp/vmem_path
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef __attribute__((address_space(3))) void lds_void_t;

__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// mode 0: stores, mode 1: LDS-DMA loads, mode 2: register loads.  `region` bytes per block (stride between blocks: `stride`).
template <int MODE>
__global__ __launch_bounds__(512, 2) void k(unsigned char* base, long stride, unsigned region, int rounds, long* stamps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned char* mine = base + (long)blockIdx.x * stride;
  const rsrc_t rs = make_rsrc(mine, region);
  // a round = 64 KiB per block = 8 KiB per wave = 8 wave-instructions of 1 KiB
  u32x4 v = {(unsigned)threadIdx.x, 1u, 2u, 3u};
  u32x4 acc = {0u, 0u, 0u, 0u};
  __syncthreads();
  const long t0 = (long)__builtin_amdgcn_s_memtime();
  unsigned off = (unsigned)(wave * 8192 + lane * 16);
  for (int r = 0; r < rounds; ++r) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const unsigned o = off + i * 1024;
      if constexpr (MODE == 0) __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)o, 0, 0);
      if constexpr (MODE == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(smem + wave * 8192 + i * 1024), 16, (int)o, 0, 0, 0);
      if constexpr (MODE == 2) { const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)o, 0, 0); acc ^= x; }
    }
    off += 65536;
    if (off >= region) off -= region;
    if constexpr (MODE == 1) { if ((r & 3) == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const long t1 = (long)__builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t0; stamps[blockIdx.x * 2 + 1] = t1; }
  if (MODE == 2 && acc[0] == 0x12345678u) stamps[0] = acc[1];
}

// mode 3 / 4: the persistent GEMM's mix - 12 rounds of LDS-DMA (64 KiB per round from an L2-resident region: one K-tile) then the tile's C
// stores, 16 per wave (128 KiB per workgroup) to the workgroup's own HBM stream; STRIDED = the stores of an instruction go to 8 rows of
// 128 bytes a row pitch apart (the register epilogue's whole-line form) instead of 1 KiB contiguous.  `with_stores` = 0 gives the same
// loop without them: the difference is what the stores cost.
template <int STRIDED>
__global__ __launch_bounds__(512, 2) void kmix(unsigned char* l2buf, unsigned region, unsigned char* cbase, long cstride, unsigned cregion, int tiles,
                                               int with_stores, unsigned pitch, int pace, int slack, long* stamps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const rsrc_t rs = make_rsrc(l2buf, region);
  const rsrc_t rc = make_rsrc(cbase + (long)blockIdx.x * cstride, cregion);
  u32x4 v = {(unsigned)threadIdx.x, 1u, 2u, 3u};
  __syncthreads();
  const long t0 = (long)__builtin_amdgcn_s_memtime();
  unsigned off = (unsigned)(wave * 8192 + lane * 16), coff = 0;
  for (int t = 0; t < tiles; ++t) {
    for (int r = 0; r < 12; ++r) {
      const long tr = (long)__builtin_amdgcn_s_memtime();
#pragma unroll
      for (int i = 0; i < 8; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(smem + wave * 8192 + i * 1024), 16, (int)(off + i * 1024), 0, 0, 0);
      off += 65536;
      if (off >= region) off -= region;
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      if (pace) {   // the GEMM's K-tile time: the next round is issued `pace` ticks after this one
        while ((long)__builtin_amdgcn_s_memtime() - tr < pace) __builtin_amdgcn_s_sleep(2);   // (a round that took longer than `pace` is not caught up)
      }
    }
    if (with_stores) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        unsigned o;
        if (STRIDED) o = coff + (unsigned)(wave * 16 + i) * 8u * pitch + (unsigned)(lane >> 3) * pitch + (unsigned)(lane & 7) * 16u;   // 8 rows x 128 B
        else o = coff + (unsigned)((wave * 16 + i) * 1024 + lane * 16);
        __builtin_amdgcn_raw_buffer_store_b128(v, rc, (int)o, 0, 0);
      }
      coff += STRIDED ? 128u : 131072u;
      if (coff + (STRIDED ? 1024u * pitch : 131072u) > cregion) coff = 0;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const long t1 = (long)__builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t0; stamps[blockIdx.x * 2 + 1] = t1; }
}

template <int STRIDED> void run_mix(const char* name, unsigned char* buf, long big, long* d_st, int pace) {
  const int gs[] = {8, 32, 64, 256};
  const int tiles = 24;
  for (int g : gs) {
    double med[2];
    for (int ws = 0; ws < 2; ++ws) {
      for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(kmix<STRIDED>, dim3(g), dim3(512), 65536, 0, buf, 1u << 20, buf + (2L << 20), big, 32u << 20, tiles, ws, 12288u, pace, 0, d_st);
        CHECK(hipDeviceSynchronize());
      }
      std::vector<long> st(2 * g); CHECK(hipMemcpy(st.data(), d_st, sizeof(long) * 2 * g, hipMemcpyDeviceToHost));
      std::vector<double> tk(g);
      for (int b = 0; b < g; ++b) tk[b] = (double)(st[2 * b + 1] - st[2 * b]) / tiles;
      std::sort(tk.begin(), tk.end());
      med[ws] = tk[g / 2];
    }
    printf("%-44s G %3d   ticks per tile (12 x 64 KiB DMA): %8.0f without stores, %8.0f with 128 x 1 KiB stores -> %6.1f ticks per store instruction\n",
           name, g, med[0], med[1], (med[1] - med[0]) / 128.0);
  }
}

// mode 5: the GEMM's operand stream as it really is - 256 rows x 128 bytes per operand tile and K-tile, rows `pitch` bytes apart (a k-contiguous
// [rows][K] bf16 operand: pitch = 2 K), 8 rows per wave-instruction; the 32 workgroups of an XCD (blockIdx & 7) share 4 A panels x 8 B panels
// like the kernel's grouped raster does, XCDs work on different panels.  No compute: what the memory side delivers for this access pattern.
__global__ __launch_bounds__(512, 2) void kpanel(unsigned char* base, unsigned pitch, int npan_a, int npan_b, int rounds, long* stamps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int xcd = blockIdx.x & 7, bi = blockIdx.x >> 3;
  const long panel = 256L * pitch;
  const int pa = bi % npan_a, pb = (bi / npan_a) % npan_b;
  const unsigned char* A = base + ((long)xcd * (npan_a + npan_b) + pa) * panel;
  const unsigned char* Bp = base + ((long)xcd * (npan_a + npan_b) + npan_a + pb) * panel;
  const rsrc_t ra = make_rsrc(A, (unsigned)panel), rb = make_rsrc(Bp, (unsigned)panel);
  unsigned vo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) vo[j] = (unsigned)(((wave * 4 + j) * 8 + (lane >> 3)) * pitch + (lane & 7) * 16);
  __syncthreads();
  const long t0 = (long)__builtin_amdgcn_s_memtime();
  unsigned koff = 0;
  for (int r = 0; r < rounds; ++r) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_t*)(smem + (wave * 4 + j) * 1024), 16, (int)vo[j], (int)koff, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void_t*)(smem + 32768 + (wave * 4 + j) * 1024), 16, (int)vo[j], (int)koff, 0, 0);
    }
    koff += 128;
    if (koff >= pitch) koff = 0;
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const long t1 = (long)__builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t0; stamps[blockIdx.x * 2 + 1] = t1; }
}
void run_panel(unsigned char* buf, unsigned pitch, int na, int nb, long* d_st) {
  const int rounds = 1024;
  for (int g : {32, 256}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(kpanel, dim3(g), dim3(512), 65536, 0, buf, pitch, na, nb, rounds, d_st);
      CHECK(hipDeviceSynchronize());
    }
    std::vector<long> st(2 * g); CHECK(hipMemcpy(st.data(), d_st, sizeof(long) * 2 * g, hipMemcpyDeviceToHost));
    std::vector<double> bpc(g);
    for (int b = 0; b < g; ++b) bpc[b] = (double)rounds * 65536.0 / (double)(st[2 * b + 1] - st[2 * b]);
    std::sort(bpc.begin(), bpc.end());
    printf("operand panels, row pitch %5u B, %d A x %d B panels per XCD (%5.1f MB per XCD)   G %3d   B/tick/CU median %6.2f  min %6.2f  max %6.2f\n", pitch, na, nb,
           (na + nb) * 256.0 * pitch * 1e-6, g, bpc[g / 2], bpc[0], bpc[g - 1]);
  }
}

template <int MODE> void run(const char* name, unsigned char* buf, long stride, unsigned region, int rounds, long* d_st) {
  const int gs[] = {1, 8, 32, 64, 128, 256};
  for (int g : gs) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(k<MODE>, dim3(g), dim3(512), 65536, 0, buf, stride, region, rounds, d_st);
      CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
    }
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long> st(2 * g); CHECK(hipMemcpy(st.data(), d_st, sizeof(long) * 2 * g, hipMemcpyDeviceToHost));
    std::vector<double> bpc(g);
    for (int b = 0; b < g; ++b) bpc[b] = (double)rounds * 65536.0 / (double)(st[2 * b + 1] - st[2 * b]);
    std::sort(bpc.begin(), bpc.end());
    const double bytes = (double)g * rounds * 65536.0;
    printf("%-34s G %3d   B/tick/CU median %6.2f  min %6.2f  max %6.2f   wall %8.1f us   aggregate %7.3f TB/s  (%.1f MB)\n", name, g, bpc[g / 2], bpc[0],
           bpc[g - 1], ms * 1e3, bytes / (ms * 1e-3) * 1e-12, bytes * 1e-6);
  }
}

int main() {
  const long big = 64L << 20;                 // 64 MiB per block: HBM-resident streams
  unsigned char* buf; CHECK(hipMalloc(&buf, 256 * big));
  CHECK(hipMemset(buf, 1, 256 * big));
  long* d_st; CHECK(hipMalloc(&d_st, sizeof(long) * 512));
  CHECK(hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  CHECK(hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  CHECK(hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  // s_memtime rate: a known wall time
  run<0>("stores, own 32 MiB stream", buf, big, 32u << 20, 512, d_st);
  run<0>("stores, own 128 KiB re-written", buf, big, 128u << 10, 512, d_st);
  run<1>("LDS-DMA, own 32 MiB stream (HBM)", buf, big, 32u << 20, 512, d_st);
  run<1>("LDS-DMA, shared 1 MiB (L2)", buf, 0, 1u << 20, 512, d_st);
  run<1>("LDS-DMA, own 256 KiB (L2)", buf, big, 256u << 10, 512, d_st);
  run<2>("register loads, own 32 MiB (HBM)", buf, big, 32u << 20, 512, d_st);
  run<2>("register loads, shared 1 MiB (L2)", buf, 0, 1u << 20, 512, d_st);
  CHECK(hipFuncSetAttribute((const void*)kmix<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  CHECK(hipFuncSetAttribute((const void*)kmix<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  CHECK(hipFuncSetAttribute((const void*)kpanel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  run_panel(buf, 1536, 4, 8, d_st);      // K = 768
  run_panel(buf, 6144, 4, 8, d_st);      // K = 3072
  run_panel(buf, 12288, 4, 8, d_st);     // K = 6144
  run_panel(buf, 6144 + 128, 4, 8, d_st);
  run_panel(buf, 6144, 1, 1, d_st);      // every workgroup of an XCD on the same two panels
  run_panel(buf, 6144, 2, 16, d_st);
  run_panel(buf, 6144, 8, 4, d_st);
  run_mix<0>("DMA flat out + C stores 1 KiB contiguous", buf, big, d_st, 0);
  run_mix<1>("DMA flat out + C stores 8 rows x 128 B", buf, big, d_st, 0);
  run_mix<0>("DMA paced 2500/K-tile + stores contiguous", buf, big, d_st, 2500);
  run_mix<1>("DMA paced 2500/K-tile + stores 8 x 128 B", buf, big, d_st, 2500);
  return 0;
}
