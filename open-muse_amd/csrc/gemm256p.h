// Persistent form of the 256 x 256 x 64 bf16 LDS-DMA GEMM (gemm256.h): one workgroup per CU WALKS output tiles.
//
// Why (profiles/r02_gemm_timeline.txt): at K = 768 a launch-per-tile block spends 4.7 k cycles in its prologue (arguments, addresses,
// first DMA round trip), 30 k in the K loop and 10.6 k in the epilogue (C staged through LDS in two passes, four barriers, and every
// CU of the chip storing its 128 KiB at the same moment) - a third of every QKV / FFN-in / out-projection block is not MFMA.  Here
//
//   * the K pipeline never stops: the LDS-DMA look-ahead of a tile's last two K-tiles already fetches the NEXT tile's first two
//     (the per-lane source offsets are tile independent, the tile origin rides in the scalar offset and a 4-bit validity mask),
//     so a tile change costs no prologue and no barrier;
//   * the epilogue needs no LDS and no barrier: accumulators are converted in registers, pairs of 16 x 16 fragments are exchanged
//     between the 16-lane rows of the wave (v_permlane16_swap) so that every lane owns 16 contiguous bytes of an output row, and
//     go out as buffer stores (rows / columns outside the matrix are dropped by the descriptor's range check).  The stores are
//     fire-and-forget: the wave re-zeroes its accumulators and continues with the next tile's first MFMA group while they drain.
//     Loads and stores retire in order on one counter, so the rest of the next tile's second K-tile is issued AHEAD of the stores
//     and the first barrier of the new tile waits with vmcnt(#stores) instead of vmcnt(0);
//   * f32 outputs with a residual (or accumulating into C) start the accumulators FROM that tensor instead of from zero: the
//     256 KiB read overlaps the first DMA round trip instead of sitting between the last MFMA and the first store;
//   * tiles are handed out per XCD: XCD x owns a contiguous range of the grouped raster (the tiles its 32 CUs work on at any
//     moment share ~4 A row panels and ~8 B column panels = one L2), its workgroups take the first 32 statically and the rest
//     from a device counter, one returning atomic per tile issued a whole tile ahead and broadcast through one LDS word.  Full
//     row tiles first, the ragged last row tile (M = 16448 = 64 x 256 + 64) last.  A workgroup that starts late because another
//     stream's kernel held its CU simply takes fewer tiles.  The counters clean themselves: the last workgroup to leave zeroes them.
//
// The wave tiling, LDS images, swizzles, fragment ring and the per-group DMA spreading are those of g256::Pipe<AL, BL, true>; the
// products and their accumulation order per output element are identical (bf16 outputs are bit-identical to the launch-per-tile
// kernel; f32 outputs with a residual differ by the order of one addition).
#pragma once
#include "gemm256.h"

namespace g256p {
using namespace g256;

constexpr int TK_OFF = LDS_BYTES;            // LDS word behind the two stages: next tile (queue position), written by wave 0
constexpr int LDS_BYTES_P = LDS_BYTES + 64;

struct Sched {
  int ntm_full, ntn, nstrip, q, r;           // full row tiles, column tiles, tiles of the ragged row strip; full tiles = 8 q + r
  // number of tiles in XCD x's queue
  __host__ __device__ int count(int x) const {
    const int s0 = (x - r) & 7;
    return q + (x < r ? 1 : 0) + (s0 < nstrip ? (nstrip - s0 + 7) / 8 : 0);
  }
  // queue position l of XCD x -> tile origin
  __device__ __forceinline__ bool decode(int x, int l, int& m0, int& n0) const {
    const int full = q + (x < r ? 1 : 0);
    if (l < 0) return false;
    if (l < full) {
      const int id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + l;
      constexpr int GM = 4;
      const int grp = id / (GM * ntn), first_m = grp * GM;
      const int gm = min(ntm_full - first_m, GM), in_grp = id - grp * (GM * ntn);
      m0 = (first_m + in_grp % gm) * BM;
      n0 = (in_grp / gm) * BN;
      return true;
    }
    const int s = ((x - r) & 7) + 8 * (l - full);   // the strip continues the round robin where the full tiles stopped
    if (s < nstrip) { m0 = ntm_full * BM; n0 = s * BN; return true; }
    return false;
  }
};

struct PArgs {
  GemmParams g;
  Sched sched;
  unsigned* counters;   // [0..7] queue heads per XCD, [8] workgroups that have left; all zero between launches
  int nk2;              // K-tiles per output tile, rounded up to even (>= 4)
  int dyn;              // some queue is longer than the workgroups of its XCD
  int epi;              // bf16 epilogue: 2 whole 128-byte lines per store (permlane16_swap + DPP pair exchange), 1 64-byte row segments
                        // (permlane16_swap only), 0 plain 8-byte stores, 3 no stores (ablation)
  int staux;            // cache policy bits of the C stores (0 default, 2 non-temporal)
  int stagger;          // > 1: the workgroups of an XCD start in that many phases, a tile time / stagger apart (dyn only)
};

// ---- LDS-DMA of one operand, tile-relative: the lane's source offsets are computed once for a tile at origin 0 -------------------
template <int L>
struct DmaP {
  rsrc_t rs;
  unsigned voff0[4];
  unsigned kk0, oob, kstep, ld2;
  unsigned vmask;   // bit j: piece j of the current tile lies inside the matrix (this lane's row / column chunk)
  unsigned r0off;   // byte offset of the current tile's first row (k-contiguous) / first column (k-major); wave-uniform
  int kend, R, rel0;
  __device__ __forceinline__ void init(const bf16_t* ptr, long ld, int R_, int K, int wave, int lane) {
    kend = K; R = R_;
    const long cols = L == 0 ? K : R_, rows = L == 0 ? R_ : K;
    const unsigned bytes = (unsigned)(((rows - 1) * ld + (cols + 7) / 8 * 8) * 2);
    rs = make_rsrc(ptr, bytes);
    oob = (bytes + 15u) & ~15u;
    ld2 = (unsigned)(ld * 2);
    vmask = 0; r0off = 0;
    if constexpr (L == 0) {
      const int r8 = lane >> 3, pc = lane & 7;
      kstep = 128u;
      kk0 = (unsigned)((pc ^ (r8 >> 1)) * 8);
      rel0 = wave * 32 + r8;                                   // row of piece j inside the tile = rel0 + 8 j
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = (wave * 4 + j) * 8 + r8;
        const int csrc = pc ^ ((row >> 1) & 7);
        voff0[j] = (unsigned)(((long)row * ld + csrc * 8) * 2);
      }
    } else {
      const int half = lane >> 5, pos = lane & 31;
      kstep = (unsigned)(64 * ld * 2);
      kk0 = (unsigned)(wave * 8 + half);
      rel0 = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kr = wave * 8 + j * 2 + half;
        const int csrc = pos ^ (km_sw(kr) << 1);
        voff0[j] = (unsigned)(((long)kr * ld + csrc * 8) * 2);
      }
    }
  }
  // first row / column of this lane's chunk of piece j, relative to the tile origin
  __device__ __forceinline__ int rel(int j, int wave, int lane) const {
    if constexpr (L == 0) return rel0 + 8 * j;
    const int half = lane >> 5, pos = lane & 31, kr = wave * 8 + j * 2 + half;
    return (pos ^ (km_sw(kr) << 1)) * 8;
  }
  __device__ __forceinline__ void set_tile(int r0, bool valid, int wave, int lane) {
    unsigned m = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) m |= (valid && (r0 + rel(j, wave, lane)) < R) ? (1u << j) : 0u;
    vmask = m;
    r0off = L == 0 ? (unsigned)r0 * ld2 : (unsigned)r0 * 2u;
  }
  __device__ __forceinline__ unsigned kk(int j) const { return L == 0 ? (kk0 ^ ((unsigned)(j & 1) << 5)) : (kk0 + 2u * j); }
  template <int LDS_OFF, int J>
  __device__ __forceinline__ void issue1(unsigned char* smem, int kt, int wave) const {
    const bool ok = ((unsigned)kt * 64u + kk(J)) < (unsigned)kend && ((vmask >> J) & 1u);
    const unsigned vo = ok ? voff0[J] : oob;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(smem + LDS_OFF + (wave * 4 + J) * 1024), 16, (int)vo,
                                             (int)((unsigned)kt * kstep + r0off), 0, 0);
  }
  template <int LDS_OFF>
  __device__ __forceinline__ void issue(unsigned char* smem, int kt, int wave) const {
    issue1<LDS_OFF, 0>(smem, kt, wave); issue1<LDS_OFF, 1>(smem, kt, wave);
    issue1<LDS_OFF, 2>(smem, kt, wave); issue1<LDS_OFF, 3>(smem, kt, wave);
  }
};

enum { M_NORMAL = 0, M_FIRST = 1, M_LAST = 2, M_PUBLISH = 3 };

template <typename TC, int AL, int BL, bool H = false>   // H: half operands (g256::mfma16)
struct PipeP {
  static constexpr bool F32OUT = sizeof(TC) == 4;
  static constexpr int RA = AL ? 2 : 1, RB = BL ? 2 : 1, NB = NI * RB;
  static constexpr int NSLOT = 4, DIST = NSLOT - 1;
  static constexpr int GB1 = 8 - DIST, GBAR = 16 - DIST;
  // stores one lane issues per tile: the first barrier of the next tile may leave that many operations in flight
  // (the 8-byte bring-up epilogue issues 32; waiting with the smaller count is merely stricter)
  static constexpr int NST = F32OUT ? 32 : 16;
  DmaP<AL> da;
  DmaP<BL> db;
  Frags<AL, MI> fa;
  Frags<BL, NI> fb;
  f32x4 acc[MI][NI];
  bf16x8 ring[NSLOT];
  bf16x8 bk[2][NI];
  unsigned char* smem;
  Sched sched;
  unsigned* counters;
  int nk2, dyn, epi, staux, stagger, M, N;
  unsigned ldc;
  float alpha;
  rsrc_t rsC, rsI;         // output; residual or old C (accumulator start values)
  unsigned oobC, oobI, ldI;
  const void* init_src;
  unsigned tk;             // wave 0, lane 0: ticket drawn for the next tile
  int wave, lane, wr, wc, xcd, nbx;
  int kti;                 // K-tile index, relative to the tile the DMA currently targets
  int nm0, nn0;
  bool has_next, pend;

  static constexpr int nB(int x) { x = ((x % 16) + 16) % 16; return (x == GB1 || x == GBAR) ? NB : 0; }
  static constexpr int waitN(int g) {
    int n = DIST * RA;
    for (int x = g - DIST + 1; x <= g; ++x) n += nB(x);
    return n > 15 ? 15 : n;
  }
  template <int S, int GA>
  __device__ __forceinline__ void read_a() {
    constexpr int st = GA < 16 ? S : (S ^ 1), g = GA & 15;
    fa.template read<st, (g >> 3), (g & 7)>(ring[GA & (NSLOT - 1)]);
  }
  template <int STAGE, int Q>
  __device__ __forceinline__ void issue_piece(int kt) {
    if constexpr (Q < 4) da.template issue1<A_BASE + STAGE * TILE, Q>(smem, kt, wave);
    else db.template issue1<B_BASE + STAGE * TILE, Q - 4>(smem, kt, wave);
  }
  template <int G> __device__ __forceinline__ void prologue_a() {
    if constexpr (G < DIST) { read_a<0, G>(); prologue_a<G + 1>(); }
  }

  // compiler-level fence: memory operations (DMA pieces, stores, loads) keep their program order across it
  static __device__ __forceinline__ void fence() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
  // the tile after this one: queue position from the LDS word wave 0 published a K-tile pair ago
  __device__ __forceinline__ void next_tile() {
    int l = -1;
    if (dyn) {
      unsigned v;
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(TK_OFF) : "memory");
      l = __builtin_amdgcn_readfirstlane((int)v);
    }
    has_next = sched.decode(xcd, l, nm0, nn0);
    if (!has_next) { nm0 = 0; nn0 = 0; }
    da.set_tile(nm0, has_next, wave, lane);
    db.set_tile(nn0, has_next, wave, lane);
    kti -= nk2;
  }
  __device__ __forceinline__ void draw_ticket() {
    if (dyn && wave == 0) {
      if (lane == 0) tk = __hip_atomic_fetch_add(counters + xcd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __device__ __forceinline__ void publish_ticket() {
    if (dyn && wave == 0) {
      if (lane == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(TK_OFF), "v"((unsigned)nbx + tk) : "memory");
    }
  }

  // one MFMA group (A-fragment row G & 7 against the 4 B fragments of K-group G >> 3) of the K-tile in stage S
  template <int S, int G, int MODE>
  __device__ __forceinline__ void group() {
    // the ticket for the tile after this one: behind the previous tile's stores when the barrier below lets them drain on (it
    // then has until the NEXT K-tile's barrier to return), else behind this K-tile's barrier
    if constexpr (G == 0 && MODE == M_FIRST) { if (pend) draw_ticket(); }
    if constexpr (G == GBAR) {
      if constexpr (MODE == M_FIRST) {
        // the other stage's tile was issued before the previous tile's stores: let the stores drain on
        if (pend) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NST) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (MODE == M_FIRST) { if (!pend) draw_ticket(); }
      if constexpr (MODE == M_LAST) next_tile();
      if constexpr (MODE == M_PUBLISH) publish_ticket();
      __builtin_amdgcn_sched_barrier(0);
      fb.template read_range<S ^ 1, 0, 0, NI>(bk[0]);
    }
    if constexpr (G >= GBAR) issue_piece<S, G - GBAR>(kti + 2);
    else if constexpr (G < 8 - (16 - GBAR) && MODE != M_FIRST) issue_piece<S ^ 1, G + (16 - GBAR)>(kti + 1);
    if constexpr (G == GB1) fb.template read_range<S, 1, 0, NI>(bk[1]);
    read_a<S, G + DIST>();
    wait_lgkm<waitN(G)>();
#pragma unroll
    for (int j = 0; j < NI; ++j)
      acc[G & 7][j] = mfma16<H>(bk[G >> 3][j], ring[G & (NSLOT - 1)], acc[G & 7][j]);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (G + 1 < 16) group<S, G + 1, MODE>();
    else ++kti;
  }
  // (Measured and dropped: skipping the MFMA groups of fragment rows outside the matrix on the tiles of the ragged last row strip
  //  - M = 16448 leaves 64 rows in the 65th row tile - with a wave-uniform branch per group.  The strip tiles get ~2x cheaper and the
  //  GEMM micro-benchmarks gain 2-5 %, but the branch costs every other tile a little and the train step, where other streams fill
  //  the tail anyway, came out 0.15 ms SLOWER: 60.00 vs 59.85 ms, same box, profiles/r03_skip_ab.txt.)
  template <int S, int MODE>
  __device__ __forceinline__ void ktile() {
    group<S, 0, MODE>();
  }

  // accumulator start values: zero, or (f32 outputs) the residual / the old C, in fragment layout
  __device__ __forceinline__ bool init_acc(int m0, int n0) {
    if constexpr (F32OUT) {
      if (init_src) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int pr = ln & 15, g = ln >> 4;
        const int rowlim = M - m0 - wr - pr;
        const unsigned soff = (unsigned)m0 * ldI * 4u + (unsigned)n0 * 4u, rstep = 16u * ldI * 4u;
        const unsigned vbase = (unsigned)(wr + pr) * ldI * 4u + (unsigned)(wc + 4 * g) * 4u;
        unsigned vb[NI];   // per fragment column: this lane's offset inside fragment row 0, or the out-of-range marker
#pragma unroll
        for (int j = 0; j < NI; ++j) vb[j] = (n0 + wc + 4 * g + 16 * j) < N ? vbase + (unsigned)j * 64u : oobI;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const bool rowok = 16 * i < rowlim;
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = __builtin_bit_cast(f32x4, buf_load16(rsI, rowok ? vb[j] : oobI, soff + (unsigned)i * rstep));
          __builtin_amdgcn_sched_barrier(0);   // (one fragment row's addresses at a time: 32 of them up front do not fit)
        }
        return true;
      }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    return false;
  }
  // v_permlane16_swap in inline asm with its wait states inside the statement: two before (the operands come straight from
  // v_cvt_pk; hipcc pads only its own builtin) and four behind it - the instruction also rewrites its SOURCE operand, and a VALU
  // or store that read either register in the next issue slots got the old value on some waves of some launches (bring-up:
  // builtin form, big grids, results off by O(1) and different from run to run; never with the 8-byte stores).
  static __device__ __forceinline__ void swap16(unsigned& a, unsigned& b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 3" : "+v"(a), "+v"(b));
  }
  // chunk q of fragment row i after the row exchange (16 contiguous bytes of this lane's output row)
  __device__ __forceinline__ void swapped_chunk(int i, int q, u32x4& w) {
    unsigned x0 = pack2_bf16(acc[i][2 * q][0], acc[i][2 * q][1]), x1 = pack2_bf16(acc[i][2 * q][2], acc[i][2 * q][3]);
    unsigned y0 = pack2_bf16(acc[i][2 * q + 1][0], acc[i][2 * q + 1][1]), y1 = pack2_bf16(acc[i][2 * q + 1][2], acc[i][2 * q + 1][3]);
    swap16(x0, y0);
    swap16(x1, y1);
    w = u32x4{x0, x1, y0, y1};
  }
  __device__ __forceinline__ void store16(const u32x4& w, unsigned vo, unsigned soff) {
    if (staux == 6) { asm volatile("" ::"v"(w), "v"(vo)); return; }   // ablation (round 6): the whole register epilogue, minus the store instructions
    if (staux == 2) __builtin_amdgcn_raw_buffer_store_b128(w, rsC, (int)vo, (int)soff, 2);
    else __builtin_amdgcn_raw_buffer_store_b128(w, rsC, (int)vo, (int)soff, 0);
  }

  // accumulators -> C, from registers
  __device__ __forceinline__ void epilogue(int m0, int n0) {
    if (alpha != 1.f) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] *= alpha;
    }
    if (epi == 3) {   // ablation: no stores (the accumulators stay live)
#pragma unroll
      for (int i = 0; i < MI; ++i)
        asm volatile("" ::"v"(acc[i][0]), "v"(acc[i][1]), "v"(acc[i][2]), "v"(acc[i][3]));
      return;
    }
    int ln = lane;
    asm volatile("" : "+v"(ln));   // (keeps the per-lane store offsets from being hoisted out of the tile loop into ~32 registers)
    const int pr = ln & 15, g = ln >> 4;
    const int rowlim = M - m0 - wr - pr;   // fragment row i lies inside the matrix iff 16 i < rowlim
    constexpr unsigned E = (unsigned)sizeof(TC);
    // fragment row i rides in the scalar offset; the lane's offset inside fragment row 0 (or the out-of-range marker for a column
    // beyond N) is computed once per tile; a row beyond M swaps the marker in per store
    const unsigned soff = (unsigned)m0 * ldc * E + (unsigned)n0 * E, rstep = 16u * ldc * E;
    const unsigned rowb = (unsigned)(wr + pr) * ldc * E;
    if constexpr (F32OUT) {
      const int cb = wc + 4 * g;
      unsigned vb[NI];
#pragma unroll
      for (int j = 0; j < NI; ++j) vb[j] = (n0 + cb + 16 * j) < N ? rowb + (unsigned)(cb + 16 * j) * 4u : oobC;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const bool rowok = 16 * i < rowlim;
#pragma unroll
        for (int j = 0; j < NI; ++j)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rsC, (int)(rowok ? vb[j] : oobC),
                                                 (int)(soff + (unsigned)i * rstep), 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      // (Round 6, measured and dropped: the lane exchange through a wave-private LDS buffer instead of the register network - four ds_write_b64 +
      //  two ds_read_b128 per fragment row, bit-identical - ran the layer's products in 735 us against 684: the K loop is LDS-bound and the
      //  epilogue's LDS round trips are latency the register form does not have.  profiles/r06_g256p_epi7_lds_transpose.txt; source under
      //  scripts/exp/rejected/.)
      if (epi == 0) {   // plain 8-byte stores (bring-up reference)
        const int cb = wc + 4 * g;
        unsigned vb[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) vb[j] = (n0 + cb + 16 * j) < N ? rowb + (unsigned)(cb + 16 * j) * 2u : oobC;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const bool rowok = 16 * i < rowlim;
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            const u32x2 w = {pack2_bf16(acc[i][j][0], acc[i][j][1]), pack2_bf16(acc[i][j][2], acc[i][j][3])};
            __builtin_amdgcn_raw_buffer_store_b64(w, rsC, (int)(rowok ? vb[j] : oobC), (int)(soff + (unsigned)i * rstep), 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        // lane (pr, g) holds columns 16 j + 4 g .. + 3 of row pr for j = 0..3 (8 bytes each).  v_permlane16_swap exchanges the odd
        // 16-lane rows of its first operand with the even rows of its second: with X = fragment 2q, Y = fragment 2q + 1,
        //   g even: keeps its X, receives X of lane g + 1  -> columns 32 q      + 8 (g >> 1) .. + 7
        //   g odd : receives Y of lane g - 1, keeps its Y  -> columns 32 q + 16 + 8 (g >> 1) .. + 7
        // so every lane owns 16 contiguous bytes: chunk q of its row, at element cq = 16 (g & 1) + 8 (g >> 1) + 32 q of the
        // wave's 64 columns.
        const int cg = 16 * (g & 1) + 8 * (g >> 1);
        if (epi == 1) {   // one chunk per store: the four lanes of a row cover 64 contiguous bytes
          unsigned vb[2];
#pragma unroll
          for (int q = 0; q < 2; ++q) vb[q] = (n0 + wc + cg + 32 * q) < N ? rowb + (unsigned)(wc + cg + 32 * q) * 2u : oobC;
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            const bool rowok = 16 * i < rowlim;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              u32x4 w;
              swapped_chunk(i, q, w);
              store16(w, rowok ? vb[q] : oobC, soff + (unsigned)i * rstep);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
          // epi 2 - whole 128-byte lines per store: neighbouring lanes pr, pr ^ 1 (rows 2 t, 2 t + 1) trade a chunk through DPP
          // (quad_perm [1,0,3,2]): the even lane gives its chunk 1 and takes the odd lane's chunk 0.  Store 0 then writes row 2 t
          // (even lane: bytes 0..63, odd lane: 64..127 of the wave's 128), store 1 writes row 2 t + 1 the same way.
          const int odd = pr & 1;
          const int col = wc + cg + 32 * odd;
          const int rl0 = rowlim + odd;            // row 16 i + (pr & ~1) inside the matrix  <=>  16 i < rl0
          const unsigned vb0 = (n0 + col) < N ? rowb - (unsigned)odd * ldc * 2u + (unsigned)col * 2u : oobC;
          const unsigned vb1 = (n0 + col) < N ? vb0 + ldc * 2u : oobC;
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            u32x4 c0, c1, d0, d1;
            swapped_chunk(i, 0, c0);
            swapped_chunk(i, 1, c1);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const unsigned from1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)c1[e], 0xB1, 0xf, 0xf, false);
              const unsigned from0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)c0[e], 0xB1, 0xf, 0xf, false);
              d0[e] = odd ? from1 : c0[e];
              d1[e] = odd ? c1[e] : from0;
            }
            store16(d0, 16 * i < rl0 ? vb0 : oobC, soff + (unsigned)i * rstep);
            store16(d1, 16 * i + 1 < rl0 ? vb1 : oobC, soff + (unsigned)i * rstep);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
  }

  // the tile loop
  __device__ __forceinline__ void tiles(int& m0, int& n0) {
    for (;;) {
      ktile<0, M_FIRST>();
      ktile<1, M_PUBLISH>();
      for (int it = 2; it < nk2 - 2; it += 2) {
        ktile<0, M_NORMAL>();
        ktile<1, M_NORMAL>();
      }
      ktile<0, M_LAST>();      // from its barrier on the DMA fetches the next tile
      ktile<1, M_NORMAL>();
      // tile change: the rest of the next tile's second K-tile goes out ahead of the stores
      issue_piece<1, 3>(1); issue_piece<1, 4>(1); issue_piece<1, 5>(1); issue_piece<1, 6>(1); issue_piece<1, 7>(1);
      fence();
      epilogue(m0, n0);
      fence();
      if (!has_next) return;
      m0 = nm0; n0 = nn0;
      const bool ld = init_acc(m0, n0);
      fence();
      pend = !ld && epi != 3;
    }
  }

  __device__ __forceinline__ void run(int m0, int n0) {
    if (dyn && stagger > 1) {
      // de-phase the CUs: all tiles take the same time, so without this every CU of the chip reaches its epilogue in the same
      // microsecond and the 32 MiB burst of C stores drains at HBM speed while every pipeline waits behind it (stores and DMA
      // loads retire in order on one counter).  A tile costs ~2500 cycles per K-tile.
      const long wait = (long)((blockIdx.x >> 3) % stagger) * ((long)nk2 * 2500L / stagger);
      const long t0 = (long)__builtin_amdgcn_s_memtime();
      while ((long)__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(16);
    }
    da.set_tile(m0, true, wave, lane);
    db.set_tile(n0, true, wave, lane);
    da.template issue<A_BASE>(smem, 0, wave);
    db.template issue<B_BASE>(smem, 0, wave);
    da.template issue<A_BASE + TILE>(smem, 1, wave);
    db.template issue<B_BASE + TILE>(smem, 1, wave);
    fence();   // (the waits below count on this issue order: DMA pieces, then accumulator loads)
    const bool loads = init_acc(m0, n0);
    fence();
    // K-tile 0 has landed; K-tile 1 (and the accumulator loads behind it) may stay in flight
    if (loads) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    fb.template read_range<0, 0, 0, NI>(bk[0]);
    prologue_a<0>();
    __builtin_amdgcn_sched_barrier(0);
    kti = 0;
    pend = false;
    has_next = false;
    tiles(m0, n0);
    // trailing zero-fill DMA pieces and look-ahead reads must not outlive the workgroup's LDS allocation
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
};

template <typename TC, int AL, int BL, bool H = false>
__global__ __launch_bounds__(512, 2) void kernel(PArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const GemmParams& p = a.g;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & 7, nbx = gridDim.x >> 3;

  int m0, n0;
  if (a.sched.decode(xcd, blockIdx.x >> 3, m0, n0)) {
    PipeP<TC, AL, BL, H> pp;
    pp.smem = smem; pp.sched = a.sched; pp.counters = a.counters; pp.nk2 = a.nk2; pp.dyn = a.dyn; pp.epi = a.epi; pp.staux = a.staux; pp.stagger = a.stagger;
    pp.M = p.M; pp.N = p.N; pp.ldc = (unsigned)p.ldc; pp.alpha = p.alpha;
    pp.wave = wave; pp.lane = lane; pp.xcd = xcd; pp.nbx = nbx;
    pp.wr = (wave >> 2) * 128; pp.wc = (wave & 3) * 64;
    pp.tk = 0;
    pp.da.init((const bf16_t*)p.A, p.lda, p.M, p.K, wave, lane);
    pp.db.init((const bf16_t*)p.B, p.ldb, p.N, p.K, wave, lane);
    pp.fa.init(A_BASE, pp.wr, lane);
    pp.fb.init(B_BASE, pp.wc, lane);
    {
      const unsigned bytes = (unsigned)(((long)(p.M - 1) * p.ldc + p.N) * (long)sizeof(TC));
      pp.rsC = make_rsrc(p.C, bytes);
      pp.oobC = (bytes + 15u) & ~15u;
    }
    pp.init_src = nullptr; pp.ldI = 0; pp.oobI = 0; pp.rsI = pp.rsC;
    if constexpr (sizeof(TC) == 4) {
      if (p.residual || p.accumulate) {
        pp.init_src = p.residual ? p.residual : (const void*)p.C;
        pp.ldI = (unsigned)(p.residual ? p.ldr : p.ldc);
        const unsigned bytes = (unsigned)(((long)(p.M - 1) * pp.ldI + p.N) * 4L);
        pp.rsI = make_rsrc(pp.init_src, bytes);
        pp.oobI = (bytes + 15u) & ~15u;
      }
    }
    pp.run(m0, n0);
  }
  // leave: the last workgroup out zeroes the queue heads for the next launch on this stream
  if (a.dyn && threadIdx.x == 0) {
    const unsigned left = __hip_atomic_fetch_add(a.counters + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (left == gridDim.x - 1) {
#pragma unroll
      for (int x = 0; x < 9; ++x) __hip_atomic_exchange(a.counters + x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace g256p
