// 32 x 32 MFMA-block building pieces shared by attention2.hip (bf16 one-tile self-attention) and attention3.hip (the bf16x3 form on f32
// tensors): the row permutation that makes a score register a B-operand slot, the lane geometry of both LDS read patterns, LDS-DMA
// images, operand fragments, the block products.  See the header comment of attention2.hip for the design.
#pragma once
#include "attention_params.h"
#include <stdlib.h>

namespace attn2 {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr float LOG2E = 1.4426950408889634f;
constexpr float NEG_BIG = -1e30f;

template <int HD> struct Cfg {
  static constexpr int KS = HD / 16;              // K = 16 steps of a head-dim contraction
  static constexpr int NDB = (HD + 31) / 32;      // 32-wide blocks over the head dim (the last one may hang over: columns never stored)
  static constexpr int STR = HD * 2 + 16;         // image row stride in bytes: an ODD number of 16-byte slots
  static constexpr int SLOTS = STR / 16;
  static constexpr int ROWS = 288;
  static constexpr int NDMA = (ROWS * SLOTS + 63) / 64;   // 1 KiB LDS-DMA instructions per image
  static constexpr int IMG = NDMA * 1024;
  static constexpr int NST = HD / 16;             // 16-byte stores per lane for one 32-row block of an output
};

// MFMA row i of a 32-row operand block <-> image row perm32(i) of the block.  i = 8 T + 4 h + j is the row register 4 T + j of the
// lanes of half h receives (C/D layout of the 32 x 32 MFMA).  With pi(8 T + 4 h + j) = 16 (T >> 1) + 4 j + 2 (T & 1) + h
//   * the 16 lanes one ds_read_b128 services together read 16 rows that are distinct mod 16 -> at an odd slot stride all 64 banks;
//   * the 8 key-slots of one B-operand register group (registers 8 t .. 8 t + 7) are the rows 16 t + h + {0, 4, 8, 12} and
//     16 t + 2 + h + {0, 4, 8, 12}: each ds_read_b64_tr_b16 of the transposed operand fetches 4 rows 4 apart = 4 x 64 bytes on four
//     different quarter-rows of the 256-byte bank line.
__device__ __forceinline__ int perm32(int i) { return ((i >> 4) << 4) + ((i & 3) << 2) + (((i >> 3) & 1) << 1) + ((i >> 2) & 1); }
__device__ __forceinline__ int perm32_inv(int r) { return ((r >> 4) << 4) + (((r >> 1) & 1) << 3) + ((r & 1) << 2) + ((r >> 2) & 3); }
// image row (inside its 32-row block) behind register r of a lane of half h
__host__ __device__ constexpr int row_of_reg(int r, int h) { return 16 * ((r >> 2) >> 1) + 4 * (r & 3) + 2 * ((r >> 2) & 1) + h; }

struct LaneGeom {
  int lane, wave, n, h;
  unsigned a_off;    // ds_read_b128 operand rows: perm32(n) * STR + 16 h      (+ block * 32 * STR + ks * 32)
  unsigned tr_off;   // ds_read_b64_tr_b16:        (h + 4 (p >> 2)) * STR + (16 m + 4 (p & 3)) * 2   (+ (block * 32 + 16 t) * STR + db * 64)
};
template <int HD> __device__ __forceinline__ LaneGeom make_geom() {
  using C = Cfg<HD>;
  LaneGeom g;
  g.lane = threadIdx.x & 63;
  g.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  g.n = g.lane & 31; g.h = g.lane >> 5;
  const int p = g.lane & 15, m = (g.lane >> 4) & 1;
  g.a_off = (unsigned)(perm32(g.n) * C::STR + 16 * g.h);
  g.tr_off = (unsigned)((g.h + 4 * (p >> 2)) * C::STR + (16 * m + 4 * (p & 3)) * 2);
  return g;
}

// 1 KiB LDS-DMA piece (buffer_load_dwordx4 ... lds: lane l's 16 bytes land at lds_addr + 16 l) in INLINE ASM, on purpose: with the
// builtin hipcc treats the in-flight DMA as a pending write to all of LDS and puts `s_waitcnt vmcnt(..)` in front of the first
// ds_read_b64_tr_b16 behind it - the next head's fetch then completes BEFORE the current head's math instead of under it
// (measured: 38.7 us forward, the DMA round trip exposed once per head).  Hidden from the compiler, the pieces are ordered by this
// file's own `s_waitcnt vmcnt(0)` + s_barrier at the top of each head / phase.  M0 (the LDS base of the DMA) is saved and restored
// inside the statement; the descriptor is built from wave-uniform words.
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((address_space(3))) unsigned char lds_u8;
__device__ __forceinline__ i32x4 dma_desc(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  i32x4 d;
  d[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  d[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
  d[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  d[3] = 0x00020000;
  return d;
}
__device__ __forceinline__ void lds_dma16(const i32x4& desc, unsigned lds_addr, unsigned voff) {
  unsigned keep;
  // (s_nop 4: the descriptor / offset registers may come straight from v_readfirstlane / VALU - hipcc pads nothing inside asm)
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds_addr), "s"(desc) : "memory");
}
// one operand image [288 rows][STR] <- rows 0 .. rows_valid-1 of a head's [S][HD] slice (row r at r * ld_bytes); everything else zeros.
// 32 (hd 48) wave-instructions, 8 per wave; the lane <-> slot map is linear (that is what the LDS side of the DMA does), the source
// offset is per lane.
constexpr int NW = 4;   // waves per workgroup of the forward kernel
#ifndef ATT2_BWD_NW
#define ATT2_BWD_NW 4    // ... of the backward kernel.  (8 = one block per wave and phase, 4 waves per SIMD: does not fit 128 VGPRs - phase 2
                         // holds dK, dV (64) + the K, V operand rows (24) + S, dP (32) before any temporary: 36-51 spilled dwords, not pursued)
#endif
template <int HD, int NWV = NW>
__device__ __forceinline__ void dma_image(unsigned char* img, const void* base, unsigned bytes, unsigned ld_bytes, int rows_valid,
                                          const LaneGeom& g) {
  using C = Cfg<HD>;
  const i32x4 desc = dma_desc(base, bytes);
  const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)(lds_u8*)img);
#pragma unroll
  for (int n0 = 0; n0 < C::NDMA; n0 += NWV) {
    const int n = n0 + g.wave;
    if (n < C::NDMA) {
      const int s = n * 64 + g.lane;
      const int row = s / C::SLOTS, c = s - row * C::SLOTS;
      const unsigned off = (row < rows_valid && c < C::SLOTS - 1) ? (unsigned)row * ld_bytes + (unsigned)c * 16u : ATT_OOB;
      lds_dma16(desc, lds0 + (unsigned)n * 1024u, off);
    }
  }
}

template <int HD> struct FragB { bf16x8 f[Cfg<HD>::KS]; };   // B operand of a head-dim contraction: 32 rows (lane n), k-slots (h, 0..7)

// rows blk * 32 + n of a [S][HD] slice straight from global memory (rows past the end read as zeros)
template <int HD>
__device__ __forceinline__ FragB<HD> load_fragb(rsrc_t rs, unsigned ld_bytes, int blk, const LaneGeom& g) {
  FragB<HD> r;
  const unsigned ro = (unsigned)(blk * 32 + g.n) * ld_bytes + 16u * g.h;
#pragma unroll
  for (int ks = 0; ks < Cfg<HD>::KS; ++ks) {
    union { u32x4 u; bf16x8 v; } t;
    t.u = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(ro + ks * 32), 0, 0);
    r.f[ks] = t.v;
  }
  return r;
}
// A operand of a head-dim contraction from an image: rows blk * 32 + perm32(n)
template <int HD>
__device__ __forceinline__ bf16x8 frag_rows(const unsigned char* img, const LaneGeom& g, int blk, int ks) {
  return *(const bf16x8*)(img + g.a_off + blk * 32 * Cfg<HD>::STR + ks * 32);
}
// A operand of a sequence contraction (transposed read): key-slots of register group t of block blk, columns 32 db + n
template <int HD>
__device__ __forceinline__ bf16x8 frag_tr(const unsigned char* img, const LaneGeom& g, int blk, int t, int db) {
  const unsigned char* a = img + g.tr_off + (blk * 32 + t * 16) * Cfg<HD>::STR + db * 64;
  union { s16x4 hh[2]; bf16x8 v; } u;
  u.hh[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a);
  u.hh[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 2 * Cfg<HD>::STR));
  return u.v;
}
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}
// C operand that masks the rows of a block at or past `valid` (rows inside the block, 1..32)
__device__ __forceinline__ f32x16 mask16(int valid, int h) {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = row_of_reg(r, 0) + h < valid ? 0.f : NEG_BIG;
  return z;
}
template <int HD>
__device__ __forceinline__ f32x16 mma_rows(const unsigned char* img, const LaneGeom& g, int blk, const FragB<HD>& b, f32x16 acc) {
#pragma unroll
  for (int ks = 0; ks < Cfg<HD>::KS; ++ks)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<HD>(img, g, blk, ks), b.f[ks], acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ bf16x8 pack8(const f32x16& v, int t) {
  union { uint32_t w[4]; bf16x8 b; } u;
#pragma unroll
  for (int e = 0; e < 4; ++e) u.w[e] = pack2_bf16(v[8 * t + 2 * e], v[8 * t + 2 * e + 1]);
  return u.b;
}
// acc[db] += sum over the 32 (t2 = 2) or first 16 (t2 = 1) rows of block blk: img^T[:, rows] * x[rows, :]
template <int HD>
__device__ __forceinline__ void mma_seq(const unsigned char* img, const LaneGeom& g, int blk, const f32x16& x, int t2,
                                        f32x16 (&acc)[Cfg<HD>::NDB]) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (t < t2) {
      const bf16x8 xb = pack8(x, t);
#pragma unroll
      for (int db = 0; db < Cfg<HD>::NDB; ++db)
        acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<HD>(img, g, blk, t, db), xb, acc[db], 0, 0, 0);
    }
  }
}
__device__ __forceinline__ float xhalf(float v) { return __shfl_xor(v, 32, 64); }   // the other half-wave's value

// v_permlane32_swap with its wait states inside the statement (operands come straight from v_cvt_pk; the instruction rewrites
// BOTH registers): lanes 32-63 of a <-> lanes 0-31 of b
__device__ __forceinline__ void swap32(unsigned& a, unsigned& b) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 3" : "+v"(a), "+v"(b));
}
// acc[db][r] = value of row n at column 32 db + 8 (r >> 2) + 4 h + (r & 3)  ->  HD / 16 chunks of 16 contiguous bytes of row n:
// chunk c at column 16 c + 8 h
template <int HD>
__device__ __forceinline__ void pack_rows(const f32x16 (&acc)[Cfg<HD>::NDB], float scale, u32x4 (&out)[Cfg<HD>::NST]) {
#pragma unroll
  for (int c = 0; c < Cfg<HD>::NST; ++c) {
    const int db = c >> 1, r0 = 8 * (c & 1);
    unsigned x0 = pack2_bf16(acc[db][r0 + 0] * scale, acc[db][r0 + 1] * scale), x1 = pack2_bf16(acc[db][r0 + 2] * scale, acc[db][r0 + 3] * scale);
    unsigned y0 = pack2_bf16(acc[db][r0 + 4] * scale, acc[db][r0 + 5] * scale), y1 = pack2_bf16(acc[db][r0 + 6] * scale, acc[db][r0 + 7] * scale);
    swap32(x0, y0);
    swap32(x1, y1);
    out[c] = u32x4{x0, x1, y0, y1};
  }
}
template <int HD>
__device__ __forceinline__ void store_rows(rsrc_t rs, unsigned ld_bytes, int blk, const u32x4 (&v)[Cfg<HD>::NST], const LaneGeom& g) {
  const unsigned ro = (unsigned)(blk * 32 + g.n) * ld_bytes + 16u * g.h;   // (rows past the end fall outside the descriptor: dropped)
#pragma unroll
  for (int c = 0; c < Cfg<HD>::NST; ++c) __builtin_amdgcn_raw_buffer_store_b128(v[c], rs, (int)(ro + c * 32), 0, 0);
}

// blockIdx -> head such that consecutive heads (the 16 heads of an image share the 128-byte lines of its packed qkv rows) run on one
// XCD (block b runs on XCD b % 8)
__device__ __forceinline__ int xcd_remap() {
  const int nb = gridDim.x, id = blockIdx.x, xcd = id & 7, slot = id >> 3, q = nb >> 3, r = nb & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}
// my LDS-DMA pieces have landed, and so have everyone's (nothing else orders a ds_read behind a DMA)
__device__ __forceinline__ void wait_all_and_barrier() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

}  // namespace attn2
