// 256 x 256 x 64 bf16 GEMM kernel with direct-to-LDS operand loads, for every operand layout (Linear forward, dX, dW).
//
// Why: ablation of the 128 x 128 kernel (scripts/exp/ablate_gemm.hip) showed the per-CU vector-memory return path and the
// VGPR -> LDS store path each cost as many cycles per tile as the MFMAs.  Here operands never touch VGPRs on the way in:
//
//   * 8 waves (2 M x 4 N), wave tile 128 x 64 = 8 x 4 MFMA 16x16x32 fragments (128 accumulator VGPRs), one block per CU.
//   * Operand tiles arrive by LDS-DMA (`buffer_load_dwordx4 ... lds`, 1 KiB per wave-instruction, 8 per wave per K-tile)
//     into two 64 KiB stages.  The LDS image is lane-linear, so the bank swizzle is applied to the per-lane SOURCE offset:
//       k-contiguous operand: [256 rows][128 B]; 16-byte chunk c of row r sits at chunk (c ^ ((r >> 1) & 7))
//                             -> each 16-lane group of a ds_read_b128 fragment read covers all 64 banks once
//       k-major operand:      [64 k-rows][512 B]; chunk c of k-row k sits at chunk (c ^ (s(k) << 1)), s(k) = k[1:0] | k[3] << 2
//                             -> the 8 k-rows one half-wave of ds_read_b64_tr_b16 touches fall on 8 distinct 32-byte bank slots
//     Rows outside the matrix and k beyond K are redirected to an out-of-range buffer offset (hardware returns zeros).
//   * K loop, software-pipelined inside each wave (fragment registers double-buffered, reads issued by inline asm so the
//     waits are counted by hand):  the ks=1 fragment reads run under the ks=0 MFMAs; then `vmcnt(0) lgkmcnt(0)` + ONE raw
//     s_barrier per K-tile; right after it the DMA for tile t+2 goes into the stage just drained and the ks=0 fragment reads
//     of tile t+1 run under the ks=1 MFMAs.  The DMA has a whole K-tile of MFMA time (>= 2048 cycles) to land.
//   * Epilogue: alpha / bias / rowvec / GELU in registers, C staged through LDS in row passes, 16-byte row-contiguous
//     global stores with residual / accumulate applied on the way out; split-K slices go to a workspace.
#pragma once
#include "gemm_core.h"
#include <type_traits>

namespace g256 {

typedef __attribute__((address_space(3))) void lds_void_t;
constexpr int BM = 256, BN = 256, BK = 64, NT = 512, MI = 8, NI = 4;
constexpr int TILE = 32768;                 // one operand tile of one stage
constexpr int A_BASE = 0, B_BASE = 65536;   // stage s of an operand at BASE + s * TILE
constexpr int LDS_BYTES = 131072;

__device__ __forceinline__ int km_sw(int kr) { return (kr & 3) | (((kr >> 3) & 1) << 2); }

template <int OFF> __device__ __forceinline__ void lds_read128(bf16x8& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int OFF> __device__ __forceinline__ void lds_read64_tr(u32x2& d, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
// One MFMA of the K loop: bf16 operands, or (H) the same 16-bit lanes read as IEEE half - 11 significant bits per operand, the TF32
// operand precision, at the bf16 rate (gfx950 has no xf32 MFMA); fragments, LDS images and DMA are the same bytes either way.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <bool H> __device__ __forceinline__ f32x4 mfma16(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  if constexpr (H) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// ---- LDS-DMA of one operand (256 rows x 64 k) --------------------------------------------------------------------------
// Each wave issues 4 pieces of 1 KiB per K-tile; piece q = wave * 4 + j lands at tile + q * 1024 (+ lane * 16 by hardware).
template <int L>
struct Dma {
  rsrc_t rs;
  unsigned voff[4];  // byte offset of this lane's chunk of piece j at K-tile 0 (or `oob`)
  unsigned kk0;      // k (inside the tile) of this lane's chunk of piece 0
  unsigned oob;
  unsigned kstep;    // bytes per K-tile
  int kend;
  __device__ __forceinline__ void init(const bf16_t* ptr, long ld, int R, int K, int kend_, int r0, int wave, int lane) {
    kend = kend_;
    const long cols = L == 0 ? K : R, rows = L == 0 ? R : K;
    const unsigned bytes = (unsigned)(((rows - 1) * ld + (cols + 7) / 8 * 8) * 2);
    rs = make_rsrc(ptr, bytes);
    oob = (bytes + 15u) & ~15u;
    if constexpr (L == 0) {
      const int r8 = lane >> 3, pc = lane & 7;
      kstep = 128u;
      kk0 = (unsigned)((pc ^ (r8 >> 1)) * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = (wave * 4 + j) * 8 + r8;           // (row >> 1) & 7 == ((j & 1) << 2) | (r8 >> 1)
        const int csrc = pc ^ ((row >> 1) & 7);
        voff[j] = (r0 + row) < R ? (unsigned)(((long)(r0 + row) * ld + csrc * 8) * 2) : oob;
      }
    } else {
      const int half = lane >> 5, pos = lane & 31;
      kstep = (unsigned)(64 * ld * 2);
      kk0 = (unsigned)(wave * 8 + half);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kr = wave * 8 + j * 2 + half;
        const int csrc = pos ^ (km_sw(kr) << 1);
        voff[j] = (r0 + csrc * 8) < R ? (unsigned)(((long)kr * ld + r0 + csrc * 8) * 2) : oob;
      }
    }
  }
  // k of this lane's chunk of piece j inside its tile
  __device__ __forceinline__ unsigned kk(int j) const { return L == 0 ? (kk0 ^ ((unsigned)(j & 1) << 5)) : (kk0 + 2u * j); }
  template <int LDS_OFF>
  __device__ __forceinline__ void issue(unsigned char* smem, int kt, int wave) const {
    const unsigned soff = (unsigned)kt * kstep;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned vo = ((unsigned)kt * 64u + kk(j)) < (unsigned)kend ? voff[j] : oob;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(smem + LDS_OFF + (wave * 4 + j) * 1024), 16, (int)vo, (int)soff, 0, 0);
    }
  }
  // piece J alone (a tile at or beyond kend is all out-of-range chunks: zeros, no memory traffic)
  template <int LDS_OFF, int J>
  __device__ __forceinline__ void issue1(unsigned char* smem, int kt, int wave) const {
    const unsigned vo = ((unsigned)kt * 64u + kk(J)) < (unsigned)kend ? voff[J] : oob;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(smem + LDS_OFF + (wave * 4 + J) * 1024), 16, (int)vo, (int)((unsigned)kt * kstep), 0, 0);
  }
};

// ---- LDS -> MFMA fragments of one operand: NF fragments of 16 rows; lane supplies row (lane & 15), k = 8 * (lane >> 4) .. +7 ---
template <int L, int NF>
struct Frags {
  static constexpr int NADDR = L == 0 ? 2 : NF;
  static constexpr int READS_PER_FRAG = L == 0 ? 1 : 2;
  unsigned addr[NADDR];
  __device__ __forceinline__ void init(int base, int wb, int lane) {
    const int p = lane & 15, g = lane >> 4;
    if constexpr (L == 0) {
      const int sw = (p >> 1) & 7;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) addr[ks] = (unsigned)(base + (wb + p) * 128 + (((ks * 4 + g) ^ sw) << 4));
    } else {
      const int kr = 8 * g + (p >> 2), s = km_sw(kr);
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const int chunk = (wb >> 3) + 2 * f + ((p & 3) >> 1);
        addr[f] = (unsigned)(base + kr * 512 + ((chunk ^ (s << 1)) << 4) + (p & 1) * 8);
      }
    }
  }
  // fragment F of K-group KS of stage STAGE
  template <int STAGE, int KS, int F>
  __device__ __forceinline__ void read(bf16x8& d) const {
    if constexpr (L == 0) {
      lds_read128<STAGE * TILE + F * 2048>(d, addr[KS]);
    } else {
      u32x2 h0, h1;
      lds_read64_tr<STAGE * TILE + KS * 16384>(h0, addr[F]);
      lds_read64_tr<STAGE * TILE + KS * 16384 + 2048>(h1, addr[F]);
      const u32x4 w = {h0[0], h0[1], h1[0], h1[1]};
      d = __builtin_bit_cast(bf16x8, w);
    }
  }
  template <int STAGE, int KS, int F0, int F1>
  __device__ __forceinline__ void read_range(bf16x8 (&d)[NF]) const {
    if constexpr (F0 < F1) {
      read<STAGE, KS, F0>(d[F0]);
      read_range<STAGE, KS, F0 + 1, F1>(d);
    }
  }
};

// ---- the K loop: one wave's view ----------------------------------------------------------------------------------------
// A K-tile is 16 MFMA "groups" g = ks * 8 + i (the 4 MFMAs of A-fragment row i against the 4 B fragments of K-group ks).
// A fragments live in a ring of NSLOT registers and are read DIST groups ahead of their use; the B fragments of both
// K-groups stay resident (read once per tile: ks=1 at group GB1, the next tile's ks=0 at group GBAR, right after the
// barrier).  The barrier sits before group GBAR = 16 - DIST, the first group whose look-ahead read targets the next stage:
//   wait vmcnt(0) lgkmcnt(0)   my DMA pieces of tile t+1 have landed, my reads of this stage are complete
//   s_barrier                  ... and so have everyone's
//   DMA tile t+2 -> this stage; fragment reads of tile t+1 begin, under the last DIST groups of MFMAs of tile t.
// LDS reads return in order, so "fragment a(g) has arrived" == at most (reads issued after it) outstanding: waitN(g).
template <int AL, int BL, bool SPREAD = false, bool H = false>
struct Pipe {
  static constexpr int RA = AL ? 2 : 1, RB = BL ? 2 : 1, NB = NI * RB;
  static constexpr int NSLOT = 4, DIST = NSLOT - 1;
  static constexpr int GB1 = 8 - DIST, GBAR = 16 - DIST;
  Dma<AL> da;
  Dma<BL> db;
  Frags<AL, MI> fa;
  Frags<BL, NI> fb;
  f32x4 acc[MI][NI];
  bf16x8 ring[NSLOT];
  bf16x8 bk[2][NI];
  unsigned char* smem;
  int wave;
#ifdef G256_L2_TOUCH   // experiment (scripts/exp/ablate256c.sh): one 4-byte load per lane and K-tile that touches every 128-byte line of the operand
  unsigned touch_vo;   // tiles G256_L2_TOUCH K-tiles ahead, so that their LDS-DMA finds them in L2 (k-contiguous operands only)
  unsigned touch_a, touch_b;   // destination registers of the touches in flight: reserved from one barrier to the next
  __device__ __forceinline__ void touch_init(int m0, int n0, int M, int N, long lda, long ldb, int lane) {
    const int row = wave * 32 + (lane & 31);
    if (lane < 32) touch_vo = (m0 + row) < M ? (unsigned)((long)(m0 + row) * lda * 2) : da.oob;
    else touch_vo = (n0 + row) < N ? (unsigned)((long)(n0 + row) * ldb * 2) : db.oob;
    touch_a = touch_b = 0;
  }
  __device__ __forceinline__ void touch(int kt) {
    if constexpr (AL == 0 && BL == 0) {
      asm volatile("" ::"v"(touch_a), "v"(touch_b));   // (the previous touches have been waited for by the barrier's vmcnt(0))
      const unsigned so = (unsigned)(kt + G256_L2_TOUCH) * 128u;
      const bool ok = (unsigned)(kt + G256_L2_TOUCH) * 64u < (unsigned)da.kend;
      // lanes 0-31 touch the A tile's rows, lanes 32-63 the B tile's: two descriptors -> two instructions, half the lanes each
      unsigned va = lane_lo() && ok ? touch_vo : da.oob, vb = !lane_lo() && ok ? touch_vo : db.oob;
      asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(touch_a) : "v"(va), "s"(da.rs), "s"(so) : "memory");
      asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(touch_b) : "v"(vb), "s"(db.rs), "s"(so) : "memory");
    }
  }
  static __device__ __forceinline__ bool lane_lo() { return (threadIdx.x & 63) < 32; }
#endif

  static constexpr int nB(int x) { x = ((x % 16) + 16) % 16; return (x == GB1 || x == GBAR) ? NB : 0; }
  static constexpr int waitN(int g) {
    int n = DIST * RA;
    for (int x = g - DIST + 1; x <= g; ++x) n += nB(x);
    return n > 15 ? 15 : n;
  }
  // A fragment of virtual group GA: 0..15 = this tile (stage S), 16.. = next tile (stage S ^ 1)
  template <int S, int GA>
  __device__ __forceinline__ void read_a() {
    constexpr int st = GA < 16 ? S : (S ^ 1), g = GA & 15;
    fa.template read<st, (g >> 3), (g & 7)>(ring[GA & (NSLOT - 1)]);
  }
  // SPREAD: the 8 DMA pieces of a tile are not issued in one burst behind the barrier but one per MFMA group - pieces 0..2 of tile
  // t+2 in groups 13..15 of tile t (its stage is free from the barrier on), pieces 3..7 in groups 0..4 of tile t+1.  A burst of 8
  // pieces costs each wave ~150 cycles of issue per piece with both waves of the SIMD doing the same thing at the same time
  // (MI355X_MICROARCH.md: "LDS-DMA piece issue cost"); a single piece among MFMAs costs ~60 and the partner wave's MFMAs run under it.
  template <int STAGE, int Q>
  __device__ __forceinline__ void issue_piece(int kt) {
    if constexpr (Q < 4) da.template issue1<A_BASE + STAGE * TILE, Q>(smem, kt, wave);
#if defined(G256_ABLATE_NO_B)      // timing experiments (scripts/exp/ablate256b.sh): the B operand's pieces are not issued at all ...
    else (void)kt;
#elif defined(G256_ABLATE_B_OOB)   // ... or issued out of range: the LDS write of a zero-filled piece happens, no memory traffic does
    else db.template issue1<B_BASE + STAGE * TILE, Q - 4>(smem, 0x3fffff, wave);
#else
    else db.template issue1<B_BASE + STAGE * TILE, Q - 4>(smem, kt, wave);
#endif
  }
  __device__ __forceinline__ void prologue(int kt0) {
    da.template issue<A_BASE>(smem, kt0, wave);
    db.template issue<B_BASE>(smem, kt0, wave);
    if constexpr (SPREAD) {   // pieces 3..7 of the second tile follow in groups 0..4 of the first
      issue_piece<1, 0>(kt0 + 1); issue_piece<1, 1>(kt0 + 1); issue_piece<1, 2>(kt0 + 1);
#if defined(G256_SPREAD_EARLY) && G256_SPREAD_EARLY == 1
      issue_piece<1, 3>(kt0 + 1); issue_piece<1, 4>(kt0 + 1); issue_piece<1, 5>(kt0 + 1); issue_piece<1, 6>(kt0 + 1); issue_piece<1, 7>(kt0 + 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#elif defined(G256_SPREAD_EARLY) && G256_SPREAD_EARLY == 2
      issue_piece<1, 3>(kt0 + 1); issue_piece<1, 4>(kt0 + 1); issue_piece<1, 5>(kt0 + 1);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
#else
      asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
#endif
    } else {
      da.template issue<A_BASE + TILE>(smem, kt0 + 1, wave);
      db.template issue<B_BASE + TILE>(smem, kt0 + 1, wave);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    fb.template read_range<0, 0, 0, NI>(bk[0]);
    prologue_a<0>();
    __builtin_amdgcn_sched_barrier(0);
  }
  template <int G> __device__ __forceinline__ void prologue_a() {
    if constexpr (G < DIST) { read_a<0, G>(); prologue_a<G + 1>(); }
  }
  template <int S, int G>
  __device__ __forceinline__ void group(int kt, int kt_last) {
    if constexpr (G == GBAR) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
#ifndef G256_ABLATE_NO_DMA
      if constexpr (!SPREAD) {
        if (kt + 2 < kt_last) {
          da.template issue<A_BASE + S * TILE>(smem, kt + 2, wave);
          db.template issue<B_BASE + S * TILE>(smem, kt + 2, wave);
        }
      }
#endif
      fb.template read_range<S ^ 1, 0, 0, NI>(bk[0]);  // (after the last tile: a stale stage, never used)
#ifdef G256_L2_TOUCH
      touch(kt);
#endif
    }
#ifndef G256_ABLATE_NO_DMA
    if constexpr (SPREAD) {   // (tiles at or beyond kend are zero-fill pieces: the piece count per tile stays uniform for the waits)
#if defined(G256_SPREAD_EARLY) && G256_SPREAD_EARLY == 1   // experiment: all eight pieces of tile t + 2 in the three groups behind the barrier (3 + 3 + 2)
      if constexpr (G == GBAR) { issue_piece<S, 0>(kt + 2); issue_piece<S, 1>(kt + 2); issue_piece<S, 2>(kt + 2); }
      if constexpr (G == GBAR + 1) { issue_piece<S, 3>(kt + 2); issue_piece<S, 4>(kt + 2); issue_piece<S, 5>(kt + 2); }
      if constexpr (G == GBAR + 2) { issue_piece<S, 6>(kt + 2); issue_piece<S, 7>(kt + 2); }
#elif defined(G256_SPREAD_EARLY) && G256_SPREAD_EARLY == 2  // experiment: two per group: 13, 14, 15 of tile t and group 0 of tile t + 1
      if constexpr (G >= GBAR) { issue_piece<S, 2 * (G - GBAR)>(kt + 2); issue_piece<S, 2 * (G - GBAR) + 1>(kt + 2); }
      else if constexpr (G == 0) { issue_piece<S ^ 1, 6>(kt + 1); issue_piece<S ^ 1, 7>(kt + 1); }
#else
      if constexpr (G >= GBAR) issue_piece<S, G - GBAR>(kt + 2);
      else if constexpr (G < 8 - (16 - GBAR)) issue_piece<S ^ 1, G + (16 - GBAR)>(kt + 1);
#endif
    }
#endif
    if constexpr (G == GB1) fb.template read_range<S, 1, 0, NI>(bk[1]);
    read_a<S, G + DIST>();
    wait_lgkm<waitN(G)>();
#ifndef G256_ABLATE_NO_MFMA
#pragma unroll
    for (int j = 0; j < NI; ++j)
      acc[G & 7][j] = mfma16<H>(bk[G >> 3][j], ring[G & (NSLOT - 1)], acc[G & 7][j]);
#else
    asm volatile("" ::"v"(ring[G & (NSLOT - 1)]), "v"(bk[G >> 3][0]), "v"(bk[G >> 3][1]), "v"(bk[G >> 3][2]), "v"(bk[G >> 3][3]));
#endif
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (G + 1 < 16) group<S, G + 1>(kt, kt_last);
  }
};

// =====================================================================================================================
// BK = 32 variant with five 32 KiB stages (160 KiB of LDS): three K-tiles of DMA in flight instead of one.
// Ablation of the BK = 64 / 2-stage pipeline (scripts/exp/ablate256.sh, [16384x3072]x[6144x3072]^T): 594 us full,
// 467 us without the in-loop DMA, 342 us without the MFMAs, 153 us with neither - the DMA stream alone runs at 1.19 us per
// 64-wide K-tile although its bytes need 0.52 us of the CU's vector-memory path: with one stage in flight it is bound by
// issue -> landed latency (Little: 64 KiB in flight per CU vs ~75 KiB needed at MFMA speed).  Same wave tiling, same
// ring, same waits; per K-tile: 8 MFMA groups, barrier before group 5, `vmcnt(8)` (tiles t+2, t+3 stay in flight).
// 64-byte LDS rows for k-contiguous operands: chunk c of row r at chunk c ^ s4((r >> 2) & 3), s4 = {0,3,2,1}.
// =====================================================================================================================
constexpr int BK32 = 32, NST32 = 5, TILE32 = 16384, STAGE32 = 32768, LDS_BYTES32 = NST32 * STAGE32;
__device__ __forceinline__ int sw4(int q) { return (4 - q) & 3; }

template <int L>
struct Dma32 {
  rsrc_t rs;
  unsigned voff[2];
  unsigned kk0, oob, kstep;
  int kend;
  __device__ __forceinline__ void init(const bf16_t* ptr, long ld, int R, int K, int kend_, int r0, int wave, int lane) {
    kend = kend_;
    const long cols = L == 0 ? K : R, rows = L == 0 ? R : K;
    const unsigned bytes = (unsigned)(((rows - 1) * ld + (cols + 7) / 8 * 8) * 2);
    rs = make_rsrc(ptr, bytes);
    oob = (bytes + 15u) & ~15u;
    if constexpr (L == 0) {
      const int csrc = (lane & 3) ^ sw4((lane >> 4) & 3);
      kstep = 64u;
      kk0 = (unsigned)(csrc * 8);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = (wave * 2 + j) * 16 + (lane >> 2);
        voff[j] = (r0 + row) < R ? (unsigned)(((long)(r0 + row) * ld + csrc * 8) * 2) : oob;
      }
    } else {
      const int half = lane >> 5, pos = lane & 31;
      kstep = (unsigned)(32 * ld * 2);
      kk0 = (unsigned)(wave * 4 + half);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int kr = wave * 4 + j * 2 + half;
        const int csrc = pos ^ (km_sw(kr) << 1);
        voff[j] = (r0 + csrc * 8) < R ? (unsigned)(((long)kr * ld + r0 + csrc * 8) * 2) : oob;
      }
    }
  }
  __device__ __forceinline__ unsigned kk(int j) const { return L == 0 ? kk0 : (kk0 + 2u * j); }
  // piece J of tile kt alone
  template <int J>
  __device__ __forceinline__ void issue1(unsigned char* smem, int lds_off, int kt, int wave) const {
    const unsigned vo = ((unsigned)kt * 32u + kk(J)) < (unsigned)kend ? voff[J] : oob;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(smem + lds_off + (wave * 2 + J) * 1024), 16, (int)vo, (int)((unsigned)kt * kstep), 0, 0);
  }
  // tile kt -> LDS byte offset `lds_off` (stage base + operand base; wave-uniform)
  __device__ __forceinline__ void issue(unsigned char* smem, int lds_off, int kt, int wave) const {
    const unsigned soff = (unsigned)kt * kstep;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const unsigned vo = ((unsigned)kt * 32u + kk(j)) < (unsigned)kend ? voff[j] : oob;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(smem + lds_off + (wave * 2 + j) * 1024), 16, (int)vo, (int)soff, 0, 0);
    }
  }
};

template <int L, int NF>
struct Frags32 {
  static constexpr int NADDR = L == 0 ? 1 : NF;
  static constexpr int READS_PER_FRAG = L == 0 ? 1 : 2;
  unsigned base[NADDR];  // stage-0 addresses
  unsigned cur[NADDR];   // addresses in the stage being read
  __device__ __forceinline__ void init(int opbase, int wb, int lane) {
    const int p = lane & 15, g = lane >> 4;
    if constexpr (L == 0) {
      base[0] = (unsigned)(opbase + (wb + p) * 64 + ((g ^ sw4((p >> 2) & 3)) << 4));
    } else {
      const int kr = 8 * g + (p >> 2), s = km_sw(kr);
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const int chunk = (wb >> 3) + 2 * f + ((p & 3) >> 1);
        base[f] = (unsigned)(opbase + kr * 512 + ((chunk ^ (s << 1)) << 4) + (p & 1) * 8);
      }
    }
  }
  __device__ __forceinline__ void point_at(unsigned stage_off) {
#pragma unroll
    for (int f = 0; f < NADDR; ++f) cur[f] = base[f] + stage_off;
  }
  template <int F>
  __device__ __forceinline__ void read(bf16x8& d) const {
    if constexpr (L == 0) {
      lds_read128<F * 1024>(d, cur[0]);
    } else {
      u32x2 h0, h1;
      lds_read64_tr<0>(h0, cur[F]);
      lds_read64_tr<2048>(h1, cur[F]);
      const u32x4 w = {h0[0], h0[1], h1[0], h1[1]};
      d = __builtin_bit_cast(bf16x8, w);
    }
  }
  template <int F0, int F1>
  __device__ __forceinline__ void read_range(bf16x8 (&d)[NF]) const {
    if constexpr (F0 < F1) {
      read<F0>(d[F0]);
      read_range<F0 + 1, F1>(d);
    }
  }
};

template <int AL, int BL>
struct Pipe32 {
  static constexpr int RA = AL ? 2 : 1, RB = BL ? 2 : 1, NB = NI * RB;
  static constexpr int NSLOT = 4, DIST = NSLOT - 1, GBAR = 8 - DIST;
  Dma32<AL> da;
  Dma32<BL> db;
  Frags32<AL, MI> fa;
  Frags32<BL, NI> fb;
  f32x4 acc[MI][NI];
  bf16x8 ring[NSLOT];
  bf16x8 bk[2][NI];
  unsigned char* smem;
  int wave;
  int so_next, so_issue, kt_issue;   // stage byte offsets of tile t+1 / of the next tile to DMA, and that tile's index

  static constexpr int waitN(int g) {
    int n = DIST * RA;
    for (int x = g - DIST + 1; x <= g; ++x) n += (((x % 8) + 8) % 8 == GBAR) ? NB : 0;
    return n > 15 ? 15 : n;
  }
  __device__ __forceinline__ void issue_next() {
    da.issue(smem, so_issue, kt_issue, wave);
    db.issue(smem, so_issue + TILE32, kt_issue, wave);
    so_issue = so_issue + STAGE32 == LDS_BYTES32 ? 0 : so_issue + STAGE32;
    ++kt_issue;
  }
  template <int G> __device__ __forceinline__ void prologue_a() {
    if constexpr (G < DIST) { fa.template read<G>(ring[G]); prologue_a<G + 1>(); }
  }
  __device__ __forceinline__ void prologue(int kt0) {
    so_issue = 0; kt_issue = kt0;
    issue_next(); issue_next(); issue_next(); issue_next();
    so_next = STAGE32;
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    fa.point_at(0u);
    fb.point_at((unsigned)TILE32);
    fb.template read_range<0, NI>(bk[0]);
    prologue_a<0>();
    __builtin_amdgcn_sched_barrier(0);
  }
  // one K-tile whose B fragments sit in bk[P]; G = MFMA group (A fragment row)
  template <int P, int G>
  __device__ __forceinline__ void group() {
    if constexpr (G == GBAR) {
      asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");  // tile t+1 landed (t+2, t+3 in flight); my reads of tile t done
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      issue_next();                                  // tile t+4 -> the stage tile t-1 used
      fa.point_at((unsigned)so_next);
      fb.point_at((unsigned)(so_next + TILE32));
      so_next = so_next + STAGE32 == LDS_BYTES32 ? 0 : so_next + STAGE32;
      fb.template read_range<0, NI>(bk[P ^ 1]);
    }
    fa.template read<((G + DIST) & 7)>(ring[(G + DIST) & (NSLOT - 1)]);
    wait_lgkm<waitN(G)>();
#ifndef G256_ABLATE_NO_MFMA
#pragma unroll
    for (int j = 0; j < NI; ++j)
      acc[G][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bk[P][j], ring[G & (NSLOT - 1)], acc[G][j], 0, 0, 0);
#else
    asm volatile("" ::"v"(ring[G & (NSLOT - 1)]), "v"(bk[P][0]), "v"(bk[P][1]), "v"(bk[P][2]), "v"(bk[P][3]));
#endif
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (G + 1 < 8) group<P, G + 1>();
  }
};

// =====================================================================================================================
// "bf16x3" K loop: C = A_hi B_hi + A_hi B_lo + A_lo B_hi from FOUR operand planes (A_hi, A_lo at A + a_lo elements, B_hi, B_lo at
// B + b_lo; same leading dimension), f32 accumulation - the f32-class product of the tape engines' "bf16x3" mode as one kernel.
// Why not the plain kernel over K-concatenated operands (hi | hi | lo) x (hi | lo | hi) (ops.split_cat3, rounds 4-5): that form moves
// SIX operand tiles through the LDS-DMA path for three products, and the K loop of this tiling is bound by exactly that path
// (~27 B/clk per CU sustained against 31.8 B/clk needed at MFMA speed).  Here a 32-wide K-tile brings its four
// planes in once (64 KiB per stage, two stages) and the waves run 3 x 32 = 96 MFMAs on them: 21 B/clk of DMA at MFMA speed, and
// each fragment read from LDS feeds 1.5 x as many MFMAs.  Tile formats, swizzles and fragment addressing are the BK = 32 pipeline's
// (Dma32 / Frags32); per A-fragment row i: reads (a_hi, a_lo) of row i + 1, then b_lo x a_hi, b_hi x a_lo, b_hi x a_hi over the four B
// fragments (small terms first; 4 independent accumulators between two MFMAs on the same one).  One barrier per K-tile, before the
// last row: all fragment reads of the stage are complete there (row 7's were issued during row 6), so the DMA of tile t + 2 goes
// into it, and the next tile's B fragments and row-0 A fragments are read under row 7's MFMAs.
// =====================================================================================================================
constexpr int X3_STAGE = 4 * TILE32, LDS_BYTES_X3 = 2 * X3_STAGE;

template <int AL, int BL>
struct PipeX3 {
  static constexpr int RA = AL ? 2 : 1, RB = BL ? 2 : 1;
  Dma32<AL> dah, dal;
  Dma32<BL> dbh, dbl;
  Frags32<AL, MI> fa;      // plane 0 of A inside a stage; the lo plane TILE32 bytes further
  Frags32<BL, NI> fb;      // B_hi at 2 * TILE32, B_lo at 3 * TILE32
  f32x4 acc[MI][NI];
  bf16x8 ah[2], al[2];
  bf16x8 bh[2][NI], bl[2][NI];
  unsigned char* smem;
  int wave;

  template <int L, int F, int OFF, typename FR>
  __device__ __forceinline__ static void rd(const FR& fr, bf16x8& d) {
    if constexpr (L == 0) {
      lds_read128<F * 1024 + OFF>(d, fr.cur[0]);
    } else {
      u32x2 h0, h1;
      lds_read64_tr<OFF>(h0, fr.cur[F]);
      lds_read64_tr<OFF + 2048>(h1, fr.cur[F]);
      const u32x4 w = {h0[0], h0[1], h1[0], h1[1]};
      d = __builtin_bit_cast(bf16x8, w);
    }
  }
  template <int I, int SLOT> __device__ __forceinline__ void read_a() {
    rd<AL, I, 0>(fa, ah[SLOT]);
    rd<AL, I, TILE32>(fa, al[SLOT]);
  }
  template <int P, int J> __device__ __forceinline__ void read_b() {
    if constexpr (J < NI) {
      rd<BL, J, 0>(fb, bh[P][J]);
      rd<BL, J, TILE32>(fb, bl[P][J]);
      read_b<P, J + 1>();
    }
  }
  __device__ __forceinline__ void issue_tile(int stage_off, int kt) {
    dah.issue(smem, stage_off, kt, wave);
    dal.issue(smem, stage_off + TILE32, kt, wave);
    dbh.issue(smem, stage_off + 2 * TILE32, kt, wave);
    dbl.issue(smem, stage_off + 3 * TILE32, kt, wave);
  }
  // The 8 DMA pieces of a tile are not issued as a burst behind the barrier (both waves of a SIMD sit at the same barrier: ~150 cycles
  // of issue per piece with no MFMA under it, measured on the plain kernel - Pipe::SPREAD) but one between two MFMA loops: pieces 0..2 of
  // tile t+2 in row 7 of tile t (its stage is free from the barrier on), pieces 3..7 in rows 0 and 1 of tile t+1 - five rows (~1900
  // cycles) before the barrier that needs them.
  template <int Q>
  __device__ __forceinline__ void issue_piece(int stage_off, int kt, int kt_last) {
    if (kt < kt_last) {
      if constexpr (Q < 2) dah.template issue1<Q>(smem, stage_off, kt, wave);
      else if constexpr (Q < 4) dal.template issue1<Q - 2>(smem, stage_off + TILE32, kt, wave);
      else if constexpr (Q < 6) dbh.template issue1<Q - 4>(smem, stage_off + 2 * TILE32, kt, wave);
      else dbl.template issue1<Q - 6>(smem, stage_off + 3 * TILE32, kt, wave);
    }
  }
  // behind MFMA loop LOOP (0..2) of row I of the tile kt in stage S
  template <int S, int I, int LOOP>
  __device__ __forceinline__ void spread(int kt, int kt_last) {
    if constexpr (I == MI - 1) issue_piece<LOOP>(S * X3_STAGE, kt + 2, kt_last);
    else if constexpr (I == 0) issue_piece<3 + LOOP>((S ^ 1) * X3_STAGE, kt + 1, kt_last);
    else if constexpr (I == 1 && LOOP < 2) issue_piece<6 + LOOP>((S ^ 1) * X3_STAGE, kt + 1, kt_last);
  }
  __device__ __forceinline__ void prologue(int kt0, int kt_last) {
    issue_tile(0, kt0);
    issue_piece<0>(X3_STAGE, kt0 + 1, kt_last);
    issue_piece<1>(X3_STAGE, kt0 + 1, kt_last);
    issue_piece<2>(X3_STAGE, kt0 + 1, kt_last);
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    fa.point_at(0u);
    fb.point_at(0u);
    read_b<0, 0>();
    read_a<0, 0>();
    __builtin_amdgcn_sched_barrier(0);
  }
  // row I of the tile in stage S (its B fragments sit in bh[S] / bl[S])
  template <int S, int I>
  __device__ __forceinline__ void row(int kt, int kt_last) {
    if constexpr (I == MI - 1) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // tile t+1 has landed; every read of this stage is complete
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      fa.point_at((unsigned)((S ^ 1) * X3_STAGE));
      fb.point_at((unsigned)((S ^ 1) * X3_STAGE));
      read_b<S ^ 1, 0>();                                           // (after the last tile: a stale stage, never used)
      read_a<0, 0>();
    } else {
      read_a<I + 1, (I + 1) & 1>();
      wait_lgkm<2 * RA>();
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[I][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[S][j], ah[I & 1], acc[I][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    spread<S, I, 0>(kt, kt_last);
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[I][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[S][j], al[I & 1], acc[I][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    spread<S, I, 1>(kt, kt_last);
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[I][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[S][j], ah[I & 1], acc[I][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    spread<S, I, 2>(kt, kt_last);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (I + 1 < MI) row<S, I + 1>(kt, kt_last);
  }
};

#ifdef G256_TIMESTAMPS   // timing experiments (scripts/exp/gemm_ts.py): s_memtime stamps per block - 0 entry, 1 prologue done, 2 K loop done, 3 exit
__device__ long* g_gemm_ts = nullptr;
#define G256_TS(IDX) if (g_gemm_ts && threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) g_gemm_ts[(long)blockIdx.x * 4 + (IDX)] = (long)__builtin_amdgcn_s_memtime();
#else
#define G256_TS(IDX)
#endif
// one 256 x 256 output tile (or one K slice of it): `tid_` = the tile's index in the problem's grouped raster
template <typename TC, int AL, int BL, int BKV, bool SPREAD, bool X3 = false, bool H = false>
__device__ __forceinline__ void tile_body(const GemmParams& p, const int tid_, unsigned char* smem) {
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
  constexpr int GM = 4;
  const int grp = tid_ / (GM * ntn), first_m = grp * GM;
  const int gm = min(ntm - first_m, GM), in_grp = tid_ - grp * (GM * ntn);
  const int m0 = (first_m + in_grp % gm) * BM, n0 = (in_grp / gm) * BN;

  const int z = blockIdx.z, zq = z / p.zdiv, zr = z - zq * p.zdiv;
  const bf16_t* Ap = (const bf16_t*)p.A + zq * p.sA0 + zr * p.sA1;
  const bf16_t* Bp = (const bf16_t*)p.B + zq * p.sB0 + zr * p.sB1;
  TC* Cp = (TC*)p.C + zq * p.sC0 + zr * p.sC1 + (p.split_k > 1 ? (long)blockIdx.y * p.split_stride : 0L);

  constexpr int BKc = BKV;
  int nk = (p.K + BKc - 1) / BKc, kt0 = 0;
  if (p.split_k > 1) {
    const int nk64 = (p.K + 63) / 64;                 // slices are cut on 64-wide boundaries for either pipeline
    const int per = (nk64 + p.split_k - 1) / p.split_k;
    if (blockIdx.y * per >= nk64) return;
    kt0 = blockIdx.y * per * (64 / BKc);
    nk = min(nk, (blockIdx.y + 1) * per * (64 / BKc));
  }
  const int kend = min(p.K, nk * BKc);

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = (wave >> 2) * 128, wc = (wave & 3) * 64;

  using PipeT = std::conditional_t<X3, PipeX3<AL, BL>, std::conditional_t<BKV == 64, Pipe<AL, BL, SPREAD, H>, Pipe32<AL, BL>>>;
  static_assert(!X3 || BKV == 32, "the bf16x3 K loop works on 32-wide K-tiles");
  static_assert(!H || (BKV == 64 && !X3 && sizeof(TC) == 4), "half operands: the 64-wide K loop with f32 output");
  PipeT pp;
  pp.smem = smem; pp.wave = wave;
  if constexpr (X3) {
    pp.dah.init(Ap, p.lda, p.M, p.K, kend, m0, wave, lane);
    pp.dal.init(Ap + p.a_lo, p.lda, p.M, p.K, kend, m0, wave, lane);
    pp.dbh.init(Bp, p.ldb, p.N, p.K, kend, n0, wave, lane);
    pp.dbl.init(Bp + p.b_lo, p.ldb, p.N, p.K, kend, n0, wave, lane);
    pp.fa.init(0, wr, lane);
    pp.fb.init(2 * TILE32, wc, lane);
  } else {
    pp.da.init(Ap, p.lda, p.M, p.K, kend, m0, wave, lane);
    pp.db.init(Bp, p.ldb, p.N, p.K, kend, n0, wave, lane);
#ifdef G256_L2_TOUCH
    if constexpr (BKV == 64) pp.touch_init(m0, n0, p.M, p.N, p.lda, p.ldb, lane);
#endif
  }
  if constexpr (X3) {
  } else if constexpr (BKV == 64) {
    pp.fa.init(A_BASE, wr, lane);
    pp.fb.init(B_BASE, wc, lane);
  } else {
    pp.fa.init(0, wr, lane);
    pp.fb.init(0, wc, lane);
  }
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) pp.acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // the tile count is rounded up to even (a tile beyond kend is all out-of-range chunks: zeros, no memory traffic)
  const int kt_last = kt0 + ((nk - kt0 + 1) & ~1);
  if constexpr (X3) pp.prologue(kt0, kt_last);
  else pp.prologue(kt0);
  G256_TS(1)
  if constexpr (X3) {
    for (int kt = kt0; kt < kt_last; kt += 2) {
      pp.template row<0, 0>(kt, kt_last);
      pp.template row<1, 0>(kt + 1, kt_last);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the trailing look-ahead reads
    __builtin_amdgcn_sched_barrier(0);
  } else if constexpr (BKV == 64) {
    for (int kt = kt0; kt < kt_last; kt += 2) {
      pp.template group<0, 0>(kt, kt_last);
      pp.template group<1, 0>(kt + 1, kt_last);
    }
    wait_lgkm<0>();  // the trailing (unused) fragment reads must not land in registers the epilogue reuses
    if constexpr (SPREAD) {   // ... nor the trailing zero-fill DMA pieces in the LDS bytes the epilogue stages C in
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    for (int kt = kt0; kt < kt_last; kt += 2) {
      pp.template group<0, 0>();
      pp.template group<1, 0>();
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // trailing zero-fill DMAs and look-ahead reads
    __builtin_amdgcn_sched_barrier(0);
  }
  auto& acc = pp.acc;
  G256_TS(2)

  // ---- epilogue: C staged through LDS, RPP rows per pass, 16-byte row-contiguous stores ----
  // lane holds C[m][n..n+3], m = m0 + wr + 16 i + (lane & 15), n = n0 + wc + 16 j + 4 (lane >> 4)
  // (kept small on purpose: 128 accumulator registers per lane make every per-element branch 128 copies of code, and a
  //  one-block-per-CU kernel cannot hide an instruction-cache-missing epilogue behind another block)
  // Loads and stores share the in-order vmcnt queue, so any load consumed after a store waits for that store's acknowledgement,
  // and __syncthreads() is a global-memory fence (vmcnt(0)) as well.  Hence: (1) everything that has to be READ - bias, row
  // vector and, for f32 outputs, the residual and the old C of an accumulating product - is folded into the accumulators in
  // fragment layout BEFORE the first store; (2) the staging passes synchronise with LDS-only barriers.  (A bf16 residual /
  // accumulate keeps its loads in the store loop: 8-byte fragment accesses would waste half of every request.)
  constexpr int EPC = 16 / (int)sizeof(TC);
  constexpr int CST = BN * (int)sizeof(TC) + 16;          // 528 (bf16) / 1040 (f32) bytes per staged row
  constexpr int RPP = sizeof(TC) == 2 ? 128 : 64;         // 67.6 KB / 66.6 KB per pass
  constexpr int NPASS = BM / RPP, PPW = 128 / RPP, IPP = MI / PPW;  // passes, passes per wave row, fragment rows per pass
  constexpr int CPRo = BN / EPC;
  const TC* Rq = (const TC*)p.residual;
  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  {
    float bias_v[NI][4];
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wc + j * 16 + 4 * (lane >> 4) + r;
        bias_v[j][r] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
      }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = m0 + wr + i * 16 + (lane & 15);
      const float rv = (p.rowvec && m < p.M) ? p.rowvec[m] : 0.f;
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = p.alpha * acc[i][j][r] + (rv + bias_v[j][r]);
    }
  }
  bool late_add = (Rq != nullptr) || p.accumulate;   // residual / old C still to be added in the store loop
  if constexpr (sizeof(TC) == 4) {
    if (late_add) {
      // one fragment row (NI x 16 bytes per lane) in flight ahead of the one being added
      const int nf = n0 + wc + 4 * (lane >> 4);
      auto load_row = [&](int i, f32x4 (&dst)[NI], const float* src, long ld) {
        const int m = m0 + wr + i * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < NI; ++j)
          dst[j] = (m < p.M && (nf + j * 16) < p.N) ? *(const f32x4*)(src + (long)m * ld + nf + j * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
      };
      auto add_all = [&](const float* src, long ld) {
        f32x4 rr[2][NI];
        load_row(0, rr[0], src, ld);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          if (i + 1 < MI) load_row(i + 1, rr[(i + 1) & 1], src, ld);
#pragma unroll
          for (int j = 0; j < NI; ++j) acc[i][j] += rr[i & 1][j];
        }
      };
      if (Rq) add_all((const float*)Rq, p.ldr);
      if (p.accumulate) add_all((const float*)Cp, p.ldc);
      late_add = false;
    }
  }
  // every load so far has been consumed (the compiler waited for it at its use): from here on the f32 path only stores
  // (Measured, no gain: taking each pass's rows from both wave rows so that all eight waves write LDS in every pass - the
  //  epilogue's 9.5-10.6 k cycles are the 128 KiB of stores going through the CU's vector-memory path, not the staging.)
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
    lds_barrier();
    if ((wave >> 2) == pass / PPW) {
#pragma unroll
      for (int ii = 0; ii < IPP; ++ii) {
        const int i = (pass % PPW) * IPP + ii;
        const int ml = wr + i * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int nl = wc + j * 16 + 4 * (lane >> 4);
          const float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
          OutVec<TC>::store4((TC*)(smem + (ml - pass * RPP) * CST) + nl, v);
        }
      }
    }
    lds_barrier();
#pragma unroll
    for (int it = 0; it < (RPP * CPRo) / NT; ++it) {
      const int c = threadIdx.x + NT * it;
      const int row = c / CPRo, col = (c % CPRo) * EPC;
      const int m = m0 + pass * RPP + row, n = n0 + col;
      if (m < p.M && n < p.N) {
        u32x4 w = *(const u32x4*)(smem + row * CST + col * (int)sizeof(TC));
        if constexpr (sizeof(TC) == 2) {
          if (late_add) {
            float o[EPC];
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[2 * e] = __uint_as_float(w[e] << 16); o[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
            auto add16 = [&](const TC* src) {
              const u32x4 t = *(const u32x4*)src;
#pragma unroll
              for (int e = 0; e < 4; ++e) { o[2 * e] += __uint_as_float(t[e] << 16); o[2 * e + 1] += __uint_as_float(t[e] & 0xffff0000u); }
            };
            if (Rq) add16(Rq + (long)m * p.ldr + n);
            if (p.accumulate) add16(Cp + (long)m * p.ldc + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = pack2_bf16(o[2 * e], o[2 * e + 1]);
          }
        }
        *(u32x4*)(Cp + (long)m * p.ldc + n) = w;
      }
    }
  }
  G256_TS(3)
}

template <typename TC, int AL, int BL, int BKV, bool SPREAD = false, bool H = false>
__global__ __launch_bounds__(512, 2) void kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  G256_TS(0)
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN, ntiles = ntm * ntn;
  const int bq = ntiles >> 3, br = ntiles & 7, xcd = blockIdx.x & 7, bi = blockIdx.x >> 3;
  const int tid_ = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + bi;
  tile_body<TC, AL, BL, BKV, SPREAD, false, H>(p, tid_, smem);
}

template <typename TC, int AL, int BL>
__global__ __launch_bounds__(512, 2) void kernel_x3(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  G256_TS(0)
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN, ntiles = ntm * ntn;
  const int bq = ntiles >> 3, br = ntiles & 7, xcd = blockIdx.x & 7, bi = blockIdx.x >> 3;
  const int tid_ = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + bi;
  tile_body<TC, AL, BL, 32, false, true>(p, tid_, smem);
}

// ---- grouped launch: the tiles of up to MAXG independent products in ONE grid (the four weight gradients of a transformer layer:
// 72 + 36 + 27 + 9 tiles at config B).  Each product alone needs a deep K split to fill 256 CUs (out-proj: 9 tiles x 28 slices of 9
// K-tiles each, a third of such a block is prologue + f32 epilogue) and ends in its own ragged round; together they fill the chip with
// one or two slices of >= 128 K-tiles per tile.  blockIdx.x walks the concatenated tile lists (XCD-swizzled as a whole: neighbouring
// tiles of a product still share their operand panels in one XCD's L2), blockIdx.y is the K slice, the same for every product.
constexpr int MAXG = 8;
struct GroupParams {
  GemmParams p[MAXG];
  int tile_start[MAXG + 1];   // first global tile index of each product (tile_start[n] = total)
  int n;
};
template <typename TC, int AL, int BL, bool SPREAD, bool H = false>
__global__ __launch_bounds__(512, 2) void kernel_group(const GroupParams gp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int ntiles = gp.tile_start[gp.n];
  const int bq = ntiles >> 3, br = ntiles & 7, xcd = blockIdx.x & 7, bi = blockIdx.x >> 3;
  const int gt = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + bi;
  int pi = 0;
#pragma unroll
  for (int k = 1; k < MAXG; ++k) pi += (k < gp.n && gt >= gp.tile_start[k]) ? 1 : 0;
  pi = __builtin_amdgcn_readfirstlane(pi);
  const GemmParams p = gp.p[pi];
  tile_body<TC, AL, BL, 64, SPREAD, false, H>(p, gt - gp.tile_start[pi], smem);
}

}  // namespace g256

// eligibility: bf16 operands; 16-byte aligned output rows; whole 16-byte chunks along the contiguous operand dimension;
// split-K only through the workspace (atomics stay with the 128^2 kernel)
template <typename TC>
static inline bool gemm256_ok(const GemmParams& p, int la, int lb) {
  constexpr int EPC = 16 / (int)sizeof(TC);
  if (p.split_k > 1 && p.split_stride == 0) return false;
  if (p.act != 0) return false;  // (GELU epilogue stays with the 128^2 kernel)
  if ((la == 0 || lb == 0) && (p.K % 8)) return false;
  if ((la == 1 && (p.M % 8)) || (lb == 1 && (p.N % 8))) return false;
  return (p.N % EPC) == 0 && (p.ldc % EPC) == 0 && ((((uintptr_t)p.C) & 15) == 0) && (p.sC0 % EPC) == 0 &&
         (p.sC1 % EPC) == 0 && (p.split_stride % EPC) == 0 &&
         (p.residual == nullptr || (((p.ldr % EPC) == 0) && ((((uintptr_t)p.residual) & 15) == 0)));
}

// (Peeling the short ragged last row-tile of M = 16448 = 64 * 256 + 64 off to the 128 x 128 kernel was tried and measured: the
// extra launch costs more (~20 us) than the round it saves on [16448x768]x[6144x768]^T (14 us).)
// Pick the tile by estimated time (microseconds on MI355X, fitted to scripts/exp/gemm256g.hip and scripts/gemm_shapes.py):
//   256^2: one block per CU (256 slots); a round costs nk * 1.8 + 6        (K loop ~1000 TFLOP/s + prologue / epilogue)
//   128^2: 3 (k-contiguous x k-contiguous, one stage) or 2 blocks per CU; a round costs nk * 1.7 + 2.5 / nk * 1.25 + 2.5
// MUSE_GEMM256 = 0 never, 1 whenever eligible, 2 (default) by the estimate.
static inline bool gemm256_preferred(const GemmParams& p, int la, int lb, int batch, bool persistent = false) {
  const char* e = getenv("MUSE_GEMM256");  // read per call (cheap) so tests can force either kernel
  const int mode = e ? (e[0] - '0') : 2;
  if (mode == 0) return false;
  if (mode == 1) return true;
  if (p.K < 128 || p.M < 256 || p.N < 256) return false;
  const long sk = p.split_k > 1 ? p.split_k : 1;
  const long t128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128) * batch * sk;
  const long t256 = (long)((p.M + 255) / 256) * ((p.N + 255) / 256) * batch * sk;
  const double nk = (double)((p.K + 63) / 64) / (double)sk;
  const bool nn = la == 0 && lb == 0;
  const long slots128 = nn ? 768 : 512;
  // persistent form (gemm256p.h): tiles are handed out dynamically and a tile change costs neither prologue nor a staged epilogue -
  // rounds count fractionally beyond the first (MI355X: [16448x768]x[2048x768]^T 61.7 us against 70.5 us on the 128^2 kernel)
  const double rounds = persistent ? (t256 <= 256 ? 1.0 : (double)t256 / 256.0) : (double)((t256 + 255) / 256);
  const double cost256 = persistent ? rounds * (nk * 1.8 + 2.0) + 4.0 : rounds * (nk * 1.8 + 6.0);
  const double cost128 = (double)((t128 + slots128 - 1) / slots128) * (nk * (nn ? 1.7 : 1.25) + 2.5);
  return cost256 < cost128;
}

template <typename TC, int AL, int BL, int BKV, bool SPREAD = false, bool H = false>
static inline int launch_gemm256_lb(const GemmParams& p, int batch, hipStream_t stream) {
  const int ntm = (p.M + 255) / 256, ntn = (p.N + 255) / 256;
  constexpr int lds = BKV == 64 ? g256::LDS_BYTES : g256::LDS_BYTES32;
  auto kern = g256::kernel<TC, AL, BL, BKV, SPREAD, H>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(ntm * ntn, p.split_k > 1 ? p.split_k : 1, batch), dim3(512), lds, stream, p);
  return (int)hipGetLastError();
}
template <typename TC, int AL, int BL>
static inline int launch_gemm256_x3_l(const GemmParams& p, hipStream_t stream) {
  const int ntm = (p.M + 255) / 256, ntn = (p.N + 255) / 256;
  auto kern = g256::kernel_x3<TC, AL, BL>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, g256::LDS_BYTES_X3);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(ntm * ntn, p.split_k > 1 ? p.split_k : 1, 1), dim3(512), g256::LDS_BYTES_X3, stream, p);
  return (int)hipGetLastError();
}
template <typename TC>
static inline int launch_gemm256_x3(const GemmParams& p, int la, int lb, hipStream_t stream) {
  if (la == 0 && lb == 0) return launch_gemm256_x3_l<TC, 0, 0>(p, stream);
  if (la == 0 && lb == 1) return launch_gemm256_x3_l<TC, 0, 1>(p, stream);
  if (la == 1 && lb == 1) return launch_gemm256_x3_l<TC, 1, 1>(p, stream);
  return launch_gemm256_x3_l<TC, 1, 0>(p, stream);
}
// MUSE_G256_BK = 64 (default): two 64 KiB stages of 64-wide K-tiles; 32: five 32 KiB stages of 32-wide K-tiles.  Measured
// equal within 2-3 % (BK = 64 ahead: 574 vs 586 us on [16384x3072]x[6144x3072]^T): the LDS-DMA stream is throughput-bound
// (~28 B/clk per CU whatever the depth), so more tiles in flight buy nothing and twice the barriers cost a little.
template <typename TC, int AL, int BL>
static inline int launch_gemm256_l(const GemmParams& p, int batch, hipStream_t stream) {
  const char* e = getenv("MUSE_G256_BK");
  if (e && e[0] == '3') return launch_gemm256_lb<TC, AL, BL, 32>(p, batch, stream);
  // MUSE_G256_SPREAD (default 1): the DMA pieces of a K-tile issued one per MFMA group instead of as a burst behind the barrier
  // (Pipe::SPREAD).  MI355X, [16448x768]x[6144x768]^T and its dX / dW: 874 -> 920, 873 -> 970, 869 -> 955 TFLOP/s; bit-identical.
  static const int spread = []() { const char* s = getenv("MUSE_G256_SPREAD"); return s ? atoi(s) : 1; }();
  return spread ? launch_gemm256_lb<TC, AL, BL, 64, true>(p, batch, stream) : launch_gemm256_lb<TC, AL, BL, 64>(p, batch, stream);
}
template <typename TC>
static inline int launch_gemm256(const GemmParams& p, int la, int lb, int batch, hipStream_t stream) {
  if (la == 0 && lb == 0) return launch_gemm256_l<TC, 0, 0>(p, batch, stream);
  if (la == 0 && lb == 1) return launch_gemm256_l<TC, 0, 1>(p, batch, stream);
  if (la == 1 && lb == 1) return launch_gemm256_l<TC, 1, 1>(p, batch, stream);
  return launch_gemm256_l<TC, 1, 0>(p, batch, stream);
}

// half operands (muse_gemm with dtype MUSE_F16: the TF32-class product of the tape engines' "f16" compute mode): f32 output, the 64-wide
// K loop with the DMA pieces spread over the MFMA groups
static inline int launch_gemm256_f16(const GemmParams& p, int la, int lb, int batch, hipStream_t stream) {
  if (la == 0 && lb == 0) return launch_gemm256_lb<float, 0, 0, 64, true, true>(p, batch, stream);
  if (la == 0 && lb == 1) return launch_gemm256_lb<float, 0, 1, 64, true, true>(p, batch, stream);
  if (la == 1 && lb == 1) return launch_gemm256_lb<float, 1, 1, 64, true, true>(p, batch, stream);
  return launch_gemm256_lb<float, 1, 0, 64, true, true>(p, batch, stream);
}
