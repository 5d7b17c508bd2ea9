// "bf16x3" 3x3 SAME convolution (see conv_split.hip for the numerics) as an LDS-DMA implicit GEMM: the activations arrive
// PRE-SPLIT into two bf16 NHWC planes x_hi / x_lo (written by the GroupNorm+SiLU kernel that produces every 3x3 conv input
// of the VQGAN - same bytes as one f32 plane), the weights as the usual w_hi / w_lo images, and all four operand images go
// global -> LDS by `buffer_load_dwordx4 ... lds` with no VGPR staging, no in-kernel split and no ds_write.
//
//   tile 256 pixels x 128 output channels x 32 k; 8 waves (4 M x 2 N), wave tile 64 x 64 (4 x 4 MFMA fragments, 3 MFMAs per
//   fragment pair: lo*hi, hi*lo, hi*hi in the order conv_split.hip uses -> bit-identical results).
//   One stage = x_hi [256][64 B] + x_lo + w_hi [128][64 B] + w_lo = 48 KiB; three stages (144 KiB), one block per CU.
//   64-byte LDS rows: 16-byte chunk c of row r sits at chunk c ^ s((r >> 2) & 3), s = {0,3,2,1}: every 16-lane group of a
//   ds_read_b128 fragment read covers the 64 banks once (swizzle applied to the per-lane DMA source offset).
//   im2col gather: K-tile kt is tap (ky,kx) = kt / (Cin/32), channels (kt % (Cin/32)) * 32 ..+31; a lane's DMA offset is
//   center(pixel) + delta(tap) or an out-of-range offset (zero fill) when the tap falls in the SAME padding.
//   Pipeline per K-tile t (registers double-buffered, stage = t % 3):
//       wait vmcnt(6) lgkmcnt(0) ; s_barrier        tile t+1 has landed everywhere, tile t's fragments are in registers
//       DMA tile t+3 -> stage t % 3 (just drained), 16 fragment reads of tile t+1, 48 MFMAs of tile t - interleaved
//   so each DMA has two K-tiles of MFMA time to land and each fragment read one.
#include "gemm256.h"
#include "../../include/muse_hip.h"

namespace cdma {
using g256::lds_read128;
using g256::lds_void_t;

constexpr int BM = 256, BN = 128, BK = 32, NT = 512, MI = 4, NI = 4, NSTAGE = 3;
constexpr int AH = 0, AL = 16384, BH = 32768, BL = 40960, STAGE = 49152, LDS_BYTES = NSTAGE * STAGE;

struct Params {
  const bf16_t *xh, *xl, *wh, *wl;
  const float* bias;
  const float* residual;
  float* out;
  double* gn_partial;  // optional: per (image, 256-pixel tile, group) sum / sum-of-squares of the output, for the next GroupNorm
  int gn_cpg, gn_groups;
  int M, N, K, H, W, Cin;
  // conv_slab_kernel<true>: the f32 activation and the GroupNorm scale / shift [batch][Cin] applied (with SiLU and the hi / lo
  // split) while the slab is staged; xh / xl are unused then
  const float *xf, *gsc, *gsh;
};

__device__ __forceinline__ int sw4(int q) { return (4 - q) & 3; }

struct Frag16 { bf16x8 ah[MI], al[MI], bh[NI], bl[NI]; };

__global__ __launch_bounds__(512, 2) void conv_dma_kernel(Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN, ntiles = ntm * ntn;
  const int bq = ntiles >> 3, br = ntiles & 7, xcd = blockIdx.x & 7, bi = blockIdx.x >> 3;
  const int tid_ = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + bi;
  const int m0 = (tid_ / ntn) * BM, n0 = (tid_ % ntn) * BN;   // the N-tiles of one pixel block run side by side (shared L2 lines)

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int H = p.H, W = p.W, Cin = p.Cin;

  // ---- DMA state ----
  const unsigned bytesA = (unsigned)((long)p.M * Cin * 2);
  const rsrc_t rs_xh = make_rsrc(p.xh, bytesA), rs_xl = make_rsrc(p.xl, bytesA);
  const unsigned oobA = (bytesA + 15u) & ~15u;
  const int srcchunk = (lane & 3) ^ sw4((lane >> 4) & 3);
  unsigned centerA[2], maskA[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = m0 + (2 * wave + j) * 16 + (lane >> 2);
    const int pix = m % (H * W), y = pix / W, x = pix - y * W;
    unsigned mask = 0;
    if (m < p.M) {
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int iy = y + t / 3 - 1, ix = x + t % 3 - 1;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) mask |= 1u << t;
      }
    }
    maskA[j] = mask;
    centerA[j] = (unsigned)(((long)m * Cin + srcchunk * 8) * 2);
  }
  const int img = wave >> 2;  // this wave's weight image: 0 = hi, 1 = lo
  const unsigned bytesB = (unsigned)((long)p.N * p.K * 2);
  const rsrc_t rs_w = make_rsrc(img ? p.wl : p.wh, bytesB);
  const unsigned oobB = (bytesB + 15u) & ~15u;
  unsigned voffB[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + ((wave & 3) * 2 + j) * 16 + (lane >> 2);
    voffB[j] = n < p.N ? (unsigned)(((long)n * p.K + srcchunk * 8) * 2) : oobB;
  }
  const unsigned kkB = (unsigned)(srcchunk * 8);
  int tap = 0, cc = 0, kti = 0;   // next K-tile to issue: tap, first channel, index

  auto issue_a = [&](int stage_off, int j) {
    const int ty = (tap * 11) >> 5, tx = tap - ty * 3;
    const unsigned delta = (unsigned)((((ty - 1) * W + (tx - 1)) * Cin + cc) * 2);
    const bool ok = tap < 9 && ((maskA[j] >> tap) & 1u);
    const unsigned vo = ok ? centerA[j] + delta : oobA;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_xh, (lds_void_t*)(smem + stage_off + AH + (2 * wave + j) * 1024), 16, (int)vo, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_xl, (lds_void_t*)(smem + stage_off + AL + (2 * wave + j) * 1024), 16, (int)vo, 0, 0, 0);
  };
  auto issue_b = [&](int stage_off) {
    const unsigned kb = (unsigned)kti * 32u;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const unsigned vo = (kb + kkB) < (unsigned)p.K ? voffB[j] : oobB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void_t*)(smem + stage_off + BH + img * 8192 + ((wave & 3) * 2 + j) * 1024), 16,
                                               (int)vo, (int)(kb * 2u), 0, 0);
    }
    cc += 32;
    if (cc >= Cin) { cc = 0; ++tap; }
    ++kti;
  };

  // ---- fragment addresses (one per stage: the immediate offset field cannot reach past 64 KiB) ----
  unsigned addrA[NSTAGE], addrB[NSTAGE];
  {
    const int pr = lane & 15, g = lane >> 4, pos = g ^ sw4((pr >> 2) & 3);
#pragma unroll
    for (int s = 0; s < NSTAGE; ++s) {
      addrA[s] = (unsigned)(s * STAGE + AH + (wm * 64 + pr) * 64 + pos * 16);
      addrB[s] = (unsigned)(s * STAGE + BH + (wn * 64 + pr) * 64 + pos * 16);
    }
  }

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  Frag16 f0, f1;

  // read #q (0..15) of the tile in stage S into F
#define CDMA_READ(F, S, Q)                                                                    \
  {                                                                                           \
    if constexpr ((Q) < 4) lds_read128<((Q) & 3) * 1024>(F.ah[(Q) & 3], addrA[S]);            \
    else if constexpr ((Q) < 8) lds_read128<16384 + ((Q) & 3) * 1024>(F.al[(Q) & 3], addrA[S]); \
    else if constexpr ((Q) < 12) lds_read128<((Q) & 3) * 1024>(F.bh[(Q) & 3], addrB[S]);      \
    else lds_read128<8192 + ((Q) & 3) * 1024>(F.bl[(Q) & 3], addrB[S]);                       \
  }
#ifndef CDMA_ABLATE_NO_MFMA
#define CDMA_PAIR(F, Q)                                                                                                        \
  {                                                                                                                            \
    constexpr int i_ = (Q) >> 2, j_ = (Q) & 3;                                                                                 \
    acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F.bl[j_], F.ah[i_], acc[i_][j_], 0, 0, 0);                            \
    acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F.bh[j_], F.al[i_], acc[i_][j_], 0, 0, 0);                            \
    acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F.bh[j_], F.ah[i_], acc[i_][j_], 0, 0, 0);                            \
  }
#else  // timing experiments only (scripts/exp): keep the fragment registers live, issue no MFMA
#define CDMA_PAIR(F, Q) asm volatile("" ::"v"(F.ah[(Q) >> 2]), "v"(F.al[(Q) >> 2]), "v"(F.bh[(Q) & 3]), "v"(F.bl[(Q) & 3]));
#endif
#ifndef CDMA_ABLATE_NO_DMA
#define CDMA_ISSUE(S, Q)                             \
  if constexpr ((Q) == 0) issue_a((S) * STAGE, 0);   \
  if constexpr ((Q) == 1) issue_a((S) * STAGE, 1);   \
  if constexpr ((Q) == 2) issue_b((S) * STAGE);
#define CDMA_TOP_WAIT "s_waitcnt vmcnt(6) lgkmcnt(0)"
#else
#define CDMA_ISSUE(S, Q)
#define CDMA_TOP_WAIT "s_waitcnt vmcnt(0) lgkmcnt(0)"
#endif
  // one (MFMA pair, DMA part, fragment read) slot of tile t: compute from CUR, prefetch tile t+1 (stage SN) into NXT,
  // DMA tile t+3 into stage S
#define CDMA_SLOT(CUR, NXT, S, SN, Q)                          \
  CDMA_PAIR(CUR, Q)                                            \
  __builtin_amdgcn_sched_barrier(0);                           \
  CDMA_ISSUE(S, Q)                                             \
  CDMA_READ(NXT, SN, Q)                                        \
  __builtin_amdgcn_sched_barrier(0);
#define CDMA_TILE(CUR, NXT, S, SN)                                                     \
  asm volatile(CDMA_TOP_WAIT ::: "memory");                                            \
  __builtin_amdgcn_s_barrier();                                                        \
  __builtin_amdgcn_sched_barrier(0);                                                   \
  CDMA_SLOT(CUR, NXT, S, SN, 0) CDMA_SLOT(CUR, NXT, S, SN, 1) CDMA_SLOT(CUR, NXT, S, SN, 2) CDMA_SLOT(CUR, NXT, S, SN, 3)     \
  CDMA_SLOT(CUR, NXT, S, SN, 4) CDMA_SLOT(CUR, NXT, S, SN, 5) CDMA_SLOT(CUR, NXT, S, SN, 6) CDMA_SLOT(CUR, NXT, S, SN, 7)     \
  CDMA_SLOT(CUR, NXT, S, SN, 8) CDMA_SLOT(CUR, NXT, S, SN, 9) CDMA_SLOT(CUR, NXT, S, SN, 10) CDMA_SLOT(CUR, NXT, S, SN, 11)   \
  CDMA_SLOT(CUR, NXT, S, SN, 12) CDMA_SLOT(CUR, NXT, S, SN, 13) CDMA_SLOT(CUR, NXT, S, SN, 14) CDMA_SLOT(CUR, NXT, S, SN, 15)

  // prologue: tiles 0, 1, 2 in flight; tile 0's fragments into f0
  issue_a(0 * STAGE, 0); issue_a(0 * STAGE, 1); issue_b(0 * STAGE);
  issue_a(1 * STAGE, 0); issue_a(1 * STAGE, 1); issue_b(1 * STAGE);
  issue_a(2 * STAGE, 0); issue_a(2 * STAGE, 1); issue_b(2 * STAGE);
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  CDMA_READ(f0, 0, 0) CDMA_READ(f0, 0, 1) CDMA_READ(f0, 0, 2) CDMA_READ(f0, 0, 3)
  CDMA_READ(f0, 0, 4) CDMA_READ(f0, 0, 5) CDMA_READ(f0, 0, 6) CDMA_READ(f0, 0, 7)
  CDMA_READ(f0, 0, 8) CDMA_READ(f0, 0, 9) CDMA_READ(f0, 0, 10) CDMA_READ(f0, 0, 11)
  CDMA_READ(f0, 0, 12) CDMA_READ(f0, 0, 13) CDMA_READ(f0, 0, 14) CDMA_READ(f0, 0, 15)
  __builtin_amdgcn_sched_barrier(0);

  // K-tiles beyond K are DMA'd as zeros (out-of-range offsets), so the loop runs whole groups of 6 (3 stages x 2 register sets)
  const int nkt = (p.K + BK - 1) / BK, ngroups = (nkt + 5) / 6;
  for (int grp = 0; grp < ngroups; ++grp) {
    CDMA_TILE(f0, f1, 0, 1)
    CDMA_TILE(f1, f0, 1, 2)
    CDMA_TILE(f0, f1, 2, 0)
    CDMA_TILE(f1, f0, 0, 1)
    CDMA_TILE(f0, f1, 1, 2)
    CDMA_TILE(f1, f0, 2, 0)
  }
#undef CDMA_TILE
#undef CDMA_ISSUE
#undef CDMA_TOP_WAIT
#undef CDMA_SLOT
#undef CDMA_PAIR
#undef CDMA_READ
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // trailing (zero) DMAs and fragment reads are done before LDS / registers are reused
  __builtin_amdgcn_sched_barrier(0);

  // ---- epilogue: + bias, staged through LDS, 16-byte row-contiguous stores with the residual added ----
  constexpr int CST = BN * 4 + 16;  // 528 bytes per staged row
  float bias_v[NI][4];
#pragma unroll
  for (int j = 0; j < NI; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + wn * 64 + j * 16 + 4 * (lane >> 4) + r;
      bias_v[j][r] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
    }
  double gs = 0.0, gq = 0.0;   // GroupNorm statistics of this thread's column chunk (4 channels = part of one group)
  // the whole 256 x 128 f32 tile fits the (now idle) operand stages: one staging pass, every wave busy
  __syncthreads();
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int lr = wm * 64 + i * 16 + (lane & 15);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int nl = wn * 64 + j * 16 + 4 * (lane >> 4);
      const f32x4 v = {acc[i][j][0] + bias_v[j][0], acc[i][j][1] + bias_v[j][1], acc[i][j][2] + bias_v[j][2], acc[i][j][3] + bias_v[j][3]};
      *(f32x4*)(smem + lr * CST + nl * 4) = v;
    }
  }
  __syncthreads();
  constexpr int NIT = (BM * 32) / NT;   // 16 row-contiguous 16-byte chunks per thread
  const int col = (threadIdx.x & 31) * 4, n = n0 + col, rbase = threadIdx.x >> 5;
  if (n < p.N) {
    f32x4 rv[NIT];
    if (p.residual) {   // all residual loads in flight before the first store
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int m = m0 + rbase + 16 * it;
        rv[it] = m < p.M ? *(const f32x4*)(p.residual + (long)m * p.N + n) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = rbase + 16 * it, m = m0 + row;
      if (m < p.M) {
        f32x4 w = *(const f32x4*)(smem + row * CST + col * 4);
        if (p.residual) w += rv[it];
        *(f32x4*)(p.out + (long)m * p.N + n) = w;
        if (p.gn_partial) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { gs += (double)w[e]; gq += (double)w[e] * (double)w[e]; }
        }
      }
    }
  }
  // GroupNorm partials: the tile's 256 pixels belong to one image (host guarantees H*W % 256 == 0); thread t owns column
  // chunk t & 31 in every iteration above -> fold the 16 threads per chunk, then the chunks of each group, in a fixed order.
  if (p.gn_partial) {
    gs += __shfl_xor(gs, 32, 64);
    gq += __shfl_xor(gq, 32, 64);
    __syncthreads();
    double* red = (double*)smem;   // [8 waves][32 chunks][2]
    if (lane < 32) { red[(wave * 32 + lane) * 2] = gs; red[(wave * 32 + lane) * 2 + 1] = gq; }
    __syncthreads();
    if (threadIdx.x < 32) {
      double cs = 0.0, cq = 0.0;
#pragma unroll
      for (int w8 = 0; w8 < 8; ++w8) { cs += red[(w8 * 32 + threadIdx.x) * 2]; cq += red[(w8 * 32 + threadIdx.x) * 2 + 1]; }
      const int cpc = p.gn_cpg >> 2;   // column chunks per group (1, 2, 4, 8)
      for (int o = 1; o < cpc; o <<= 1) { cs += __shfl_xor(cs, o, 64); cq += __shfl_xor(cq, o, 64); }
      const int n = n0 + (int)threadIdx.x * 4;
      if ((threadIdx.x & (cpc - 1)) == 0 && n < p.N) {
        const int hw = p.H * p.W, b = m0 / hw, chunk = (m0 - b * hw) >> 8, nchunk = hw >> 8;
        double* o2 = p.gn_partial + (((long)b * nchunk + chunk) * p.gn_groups + n / p.gn_cpg) * 2;
        o2[0] = cs; o2[1] = cq;
      }
    }
  }
}

}  // namespace cdma


// =====================================================================================================================
// "Patch-slab" variant (taken when H % 16 == 0, W % 16 == 0, Cin % 64 == 0): the same bf16x3 products, but the activation
// side of the implicit GEMM is staged ONCE per 32-channel chunk instead of once per tap.
//
//   The kernel above DMAs 48 KiB per K-tile (x_hi/x_lo [256][32] + w_hi/w_lo [128][32]) for 1546 MFMA cycles: 31.8 B/clk/CU
//   against the ~28 B/clk the LDS-DMA path sustains - its K loop is DMA-bound, and the nine taps
//   re-load the same pixels nine times.  Here a block's 256 pixels are a 16 x 16 PATCH of one image and, per 32-channel
//   chunk, the 18 x 18 halo'd patch ("slab", 324 rows of 64 B per plane) is DMA'd once and serves all nine taps: tap
//   (ky,kx) of patch row py reads slab rows (py+ky)*18 + kx + 0..15.  DMA bytes per K-tile: 4.6 KiB of activations + 16 KiB
//   of weights = 13.7 B/clk/CU; K order = (channel chunk, tap, channel) instead of (tap, channel).
//
//   LDS: slab buffer = hi plane [21 KiB] + lo plane [21 KiB], two buffers (chunk c+1 lands while chunk c computes), then a
//   3-stage weight ring of 16 KiB (w_hi [128][64 B] + w_lo) = 132 KiB; the epilogue stages the f32 tile in the same bytes.
//   Slab rows are read at ARBITRARY 16-row windows (kx shifts), so the bank swizzle has to be conflict-free for every start
//   row: chunk c of row r sits at position c ^ ((r >> 1) & 2) (scripts/lds_swizzle_search.py enumerates the ds_read_b128
//   lane groups of the guide for all alignments: 4 LDS cycles = no conflict for every window; the {0,3,2,1} swizzle of the
//   aligned kernel costs 8 on unaligned ones).  Read address = (P + c') * 64 + pos with P = 72 * wm + (lane & 15) per lane
//   and c' = (i + ky) * 18 + kx a compile-time constant: the swizzle bit depends on bit 2 of P + c', i.e. on c' & 3 (carry
//   into bit 2) and bit 2 of c' - eight per-lane base registers, everything else is the instruction's immediate offset.
//   Pipeline per tap t (one K-tile): wait for the weights of tap t+1, barrier, then 48 MFMAs of tap t interleaved with the
//   16 fragment reads of tap t+1 and the DMA of the weights of tap t+3 (+ in taps 0-2 a third of the next chunk's slab).
namespace cslab {
using g256::lds_read128;
using g256::lds_void_t;
using cdma::Params;
using cdma::sw4;

constexpr int BM = 256, BN = 128, NT = 512, MI = 4, NI = 4;
constexpr int PLANE = 21 * 1024;            // 324 slab rows x 64 B, rounded up to whole 1 KiB DMA pieces (21)
constexpr int SLAB = 2 * PLANE;             // hi + lo
constexpr int B_BASE = 2 * SLAB, BSTAGE = 16384;
constexpr int CST = BN * 4 + 16;            // staged output row (epilogue)
constexpr int LDS_BYTES = BM * CST;         // 135168 >= B_BASE + 3 * BSTAGE (135168)
static_assert(B_BASE + 3 * BSTAGE <= LDS_BYTES, "LDS map");

struct Frag16 { bf16x8 ah[MI], al[MI], bh[NI], bl[NI]; };

#ifdef CDMA_TIMESTAMPS   // timing experiments (scripts/exp/conv_ts.py): per-block s_memtime stamps at the phase boundaries
__device__ long* g_conv_ts = nullptr;
#define CDMA_TS(IDX) if (g_conv_ts && threadIdx.x == 0) g_conv_ts[(long)blockIdx.x * 8 + (IDX)] = (long)__builtin_amdgcn_s_memtime();
#define CDMA_TSV(V, IDX) if (g_conv_ts && threadIdx.x == 0) g_conv_ts[(long)(V) * 8 + (IDX)] = (long)__builtin_amdgcn_s_memtime();
#else
#define CDMA_TS(IDX)
#define CDMA_TSV(V, IDX)
#endif
// GN = true: GroupNorm + SiLU + the bf16x3 split of the INPUT happen here (muse/modeling_maskgit_vqgan.py:73-80: norm -> swish ->
// conv).  The slab thirds are loaded as f32 into registers (same two vector-memory operations per third as the two plane DMAs,
// so the counted waits keep their meaning; issued in taps 0 / 2 / 4 instead of 0 / 1 / 2), transformed two taps later - the
// top-of-tap wait of tap s+2 leaves only tap s+1's operations outstanding - with the arithmetic of gn_apply_split8_kernel
// (vqgan.hip; same bits) spread over that tap's MFMA slots, and written to the slab buffer in the layout the plane DMA produces.
// Saves the write and the re-read of both planes (8 of the 12 bytes per element the unfused pair moves) and a launch.
template <bool GN>
__global__ __launch_bounds__(512, 2) void conv_slab_kernel(Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  CDMA_TS(0)
  const int H = p.H, W = p.W, Cin = p.Cin;
  const int tpr = W >> 4, tpi = (H >> 4) * tpr;     // patches per image row / per image
  const int ntm = p.M >> 8, ntn = (p.N + BN - 1) / BN, ntiles = ntm * ntn;
  const int bq = ntiles >> 3, br = ntiles & 7, xcd = blockIdx.x & 7, bi = blockIdx.x >> 3;
  const int tid_ = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + bi;
  const int mt = tid_ / ntn, n0 = (tid_ % ntn) * BN;
  const int img = mt / tpi, prem = mt - img * tpi, ty = prem / tpr, tx = prem - ty * tpr;
  const int y0 = ty << 4, x0 = tx << 4;

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // ---- slab DMA: 21 pieces of 1 KiB per plane; wave w issues pieces w, w+8, w+16 of both planes (a piece index past 20
  //      repeats the wave's previous piece: same bytes to the same place, keeps the per-wave DMA count uniform) ----
  constexpr int XE = GN ? 4 : 2;             // bytes per activation element in global memory
  const unsigned bytesA = (unsigned)((long)p.M * Cin * XE);
  const rsrc_t rs_xh = make_rsrc(GN ? (const void*)p.xf : (const void*)p.xh, bytesA), rs_xl = make_rsrc(GN ? (const void*)p.xf : (const void*)p.xl, bytesA);
  const unsigned oobA = (bytesA + 15u) & ~15u;
  unsigned slab_off[3];
  int slab_piece[3];
  const int srcchunk = (lane & 3) ^ (((lane >> 4) & 1) << 1);
  {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      int piece = wave + 8 * k;
      if (piece > 20) piece -= 8;
      slab_piece[k] = piece;
      const int r = piece * 16 + (lane >> 2), sy = r / 18, sx = r - sy * 18;
      const int y = y0 - 1 + sy, x = x0 - 1 + sx;
      const bool ok = r < 324 && y >= 0 && y < H && x >= 0 && x < W;
      slab_off[k] = ok ? (unsigned)(((((long)img * H + y) * W + x) * Cin + srcchunk * 8) * XE) : oobA;
    }
  }
  // GN: the image's scale / shift behind the tile's LDS map, [2][Cin] f32
  float* const gss = (float*)(smem + LDS_BYTES);
  if constexpr (GN) {
    for (int c = threadIdx.x; c < Cin; c += NT) { gss[c] = p.gsc[(long)img * Cin + c]; gss[Cin + c] = p.gsh[(long)img * Cin + c]; }
    __syncthreads();
  }
  u32x4 xr[2];                  // GN: the slab third in flight / being transformed (one at a time inside the K loop)
  float scv[4], shv[4];
  // third `k` of the slab of channel chunk `chunk`, f32, into registers (zero outside the image / past the last chunk)
  auto issue_slab_f = [&](u32x4 (&r)[2], int k, int chunk) {
    const unsigned vo = chunk * 32 < Cin ? slab_off[k] : oobA;
    r[0] = buf_load16(rs_xh, vo, (unsigned)(chunk * 128));
    r[1] = buf_load16(rs_xh, vo + 16u, (unsigned)(chunk * 128));
  };
  // half `h` (four channels) of this lane's eight channels of chunk `chunk`
  auto load_gss = [&](int chunk, int h) {
    const float* q = gss + chunk * 32 + srcchunk * 8 + h * 4;
#pragma unroll
#ifndef GNX_CONST_SS      // (timing experiments, scripts/exp/conv_gn_variants.sh: what the scale / shift reads and the SiLU cost)
    for (int j = 0; j < 4; ++j) { scv[j] = q[j]; shv[j] = q[Cin + j]; }
#else
    for (int j = 0; j < 4; ++j) { scv[j] = 1.25f; shv[j] = 0.125f; }
    (void)q;
#endif
  };
  // element e (0..7) of a third: GroupNorm affine + SiLU, exactly gn_apply_split8_kernel's expression; in place
  auto xform_elem = [&](u32x4 (&r)[2], int e) {
    float t = fmaf(__uint_as_float(r[e >> 2][e & 3]), scv[e & 3], shv[e & 3]);
#ifndef GNX_NO_SILU
    t = gn_silu(t);
#endif
    asm volatile("" : "+v"(t));   // the ROUNDED product is what gets split: no contraction of (x * r) - hi into one fma
    r[e >> 2][e & 3] = __float_as_uint(t);
  };
  // hi / lo split of third k and its two 16-byte LDS writes (where the plane DMA would have put them); zero padding stays zero
  auto store_slab_f = [&](u32x4 (&r)[2], int buf, int k) {
    u32x4 hi4, lo4;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      u32x2 hi, lo;
      split4(r[h], hi, lo);
      hi4[2 * h] = hi[0]; hi4[2 * h + 1] = hi[1];
      lo4[2 * h] = lo[0]; lo4[2 * h + 1] = lo[1];
    }
    if (slab_off[k] == oobA) { hi4 = u32x4{0u, 0u, 0u, 0u}; lo4 = hi4; }
    unsigned char* dst = smem + buf * SLAB + slab_piece[k] * 1024 + lane * 16;
    *(u32x4*)dst = hi4;
    *(u32x4*)(dst + PLANE) = lo4;
  };

  // third `k` of the slab of channel chunk `chunk` into buffer `buf`
  auto issue_slab = [&](int buf, int k, int chunk) {
    const unsigned vo = chunk * 32 < Cin ? slab_off[k] : oobA;
    const int soff = chunk * 64;
    unsigned char* dst = smem + buf * SLAB + slab_piece[k] * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_xh, (lds_void_t*)dst, 16, (int)vo, soff, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_xl, (lds_void_t*)(dst + PLANE), 16, (int)vo, soff, 0, 0);
  };

  // ---- weight DMA (as in cdma): waves 0-3 the hi image, 4-7 the lo image, two 16-row pieces each per K-tile ----
  const int wimg = wave >> 2;
  const unsigned bytesB = (unsigned)((long)p.N * p.K * 2);
  const rsrc_t rs_w = make_rsrc(wimg ? p.wl : p.wh, bytesB);
  const unsigned oobB = (bytesB + 15u) & ~15u;
  unsigned voffB[2];
  {
    const int srcchunk = (lane & 3) ^ sw4((lane >> 4) & 3);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + ((wave & 3) * 2 + j) * 16 + (lane >> 2);
      voffB[j] = n < p.N ? (unsigned)(((long)n * p.K + srcchunk * 8) * 2) : oobB;
    }
  }
  // weights of (chunk, tap) into ring stage `stage`: k index = tap * Cin + chunk * 32
  auto issue_b = [&](int stage, int chunk, int tap) {
    const bool ok = chunk * 32 < Cin;
    const int soff = (tap * Cin + chunk * 32) * 2;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void_t*)(smem + B_BASE + stage * BSTAGE + wimg * 8192 + ((wave & 3) * 2 + j) * 1024),
                                               16, (int)(ok ? voffB[j] : oobB), soff, 0, 0);
  };

  // ---- fragment read addresses ----
  unsigned baseA[4][2], addrB;
  {
    const int pr = lane & 15, g = lane >> 4, P = 72 * wm + pr;
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int bit2 = ((P >> 2) & 1) ^ (((P & 3) + e) >> 2) ^ b;
        baseA[e][b] = (unsigned)((P << 6) | ((g << 4) ^ (bit2 << 5)));
      }
    addrB = (unsigned)(B_BASE + (wn * 64 + pr) * 64 + ((g ^ sw4((pr >> 2) & 3)) << 4));
  }

  auto flip_base = [&](int d) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { baseA[e][0] += d; baseA[e][1] += d; }
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  Frag16 f0, f1;

  // read #Q (0..15) of tap TAP (0..8) from the slab buffer baseA points into / weight stage ST into F.   A reads: c' = (i + ky) * 18 + kx.
#define SLAB_CP(TAP, I) (((I) + (TAP) / 3) * 18 + (TAP) % 3)
#define SLAB_READ(F, ST, TAP, Q)                                                                                             \
  {                                                                                                                               \
    constexpr int q_ = (Q) & 3;                                                                                                   \
    constexpr int cp_ = SLAB_CP(TAP, q_);                                                                                         \
    if constexpr ((Q) < 4) lds_read128<cp_ * 64>(F.ah[q_], baseA[cp_ & 3][(cp_ >> 2) & 1]);                                       \
    else if constexpr ((Q) < 8) lds_read128<PLANE + cp_ * 64>(F.al[q_], baseA[cp_ & 3][(cp_ >> 2) & 1]);                          \
    else if constexpr ((Q) < 12) lds_read128<(ST) * BSTAGE + q_ * 1024>(F.bh[q_], addrB);                                         \
    else lds_read128<(ST) * BSTAGE + 8192 + q_ * 1024>(F.bl[q_], addrB);                                                          \
  }
#if defined(CDMA_TWO_PRODUCTS)   // accuracy experiment (scripts/exp/conv_two_products.sh): drop one cross term of the bf16x3 product
#define SLAB_PAIR(F, Q)                                                                                                        \
  {                                                                                                                            \
    constexpr int i_ = (Q) >> 2, j_ = (Q) & 3;                                                                                 \
    if constexpr (CDMA_TWO_PRODUCTS == 1) acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F.bh[j_], F.al[i_], acc[i_][j_], 0, 0, 0);  /* w_lo dropped */ \
    else acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F.bl[j_], F.ah[i_], acc[i_][j_], 0, 0, 0);                       /* x_lo dropped */ \
    acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F.bh[j_], F.ah[i_], acc[i_][j_], 0, 0, 0);                            \
  }
#elif !defined(CDMA_ABLATE_NO_MFMA)
#define SLAB_PAIR(F, Q)                                                                                                        \
  {                                                                                                                            \
    constexpr int i_ = (Q) >> 2, j_ = (Q) & 3;                                                                                 \
    acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F.bl[j_], F.ah[i_], acc[i_][j_], 0, 0, 0);                            \
    acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F.bh[j_], F.al[i_], acc[i_][j_], 0, 0, 0);                            \
    acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F.bh[j_], F.ah[i_], acc[i_][j_], 0, 0, 0);                            \
  }
#else
#define SLAB_PAIR(F, Q) asm volatile("" ::"v"(F.ah[(Q) >> 2]), "v"(F.al[(Q) >> 2]), "v"(F.bh[(Q) & 3]), "v"(F.bl[(Q) & 3]));
#endif
  // slot Q of tap T (0..17 inside the two-chunk loop body; chunk parity CP = T / 9, tap TAP = T % 9) computing from CUR while
  // prefetching the fragments of tap T+1 into NXT.  DMA: slot 0/1 = a third of the NEXT chunk's slab (taps 0..2 only), slot 2 =
  // the weights of tap T+3.
#ifndef CDMA_ABLATE_NO_DMA
#define SLAB_ISSUE(T, Q)                                                                          \
  if constexpr (!GN && (Q) == 0 && ((T) % 9) < 3) issue_slab(1 - (T) / 9, (T) % 9, chunk0 + (T) / 9 + 1);  \
  if constexpr (GN && (Q) == 0 && ((T) % 9) == 0) issue_slab_f(xr, 0, chunk0 + (T) / 9 + 1);           \
  if constexpr (GN && (Q) == 12 && (((T) % 9) == 2 || ((T) % 9) == 4)) issue_slab_f(xr, ((T) % 9) / 2, chunk0 + (T) / 9 + 1);  \
  if constexpr ((Q) == 2) issue_b(((T) + 3) % 3, chunk0 + ((T) + 3) / 9, ((T) + 3) % 9);
// GN: one register set: third 0 is issued at the top of tap 0, thirds 1 / 2 at slot 12 of taps 2 / 4 (once the previous third has
// left the registers); third k is transformed in tap 2k + 2: scale / shift halves at slots 1 / 6, one element per slot 2..5 and
// 7..10, split + LDS write at slot 11 - all of it done before tap 8 reads the next chunk's first fragments
#define SLAB_XF(T, Q)                                                                             \
  if constexpr (GN && ((T) % 9) >= 2 && ((T) % 9) <= 6 && ((T) % 9) % 2 == 0) {                    \
    constexpr int k_ = ((T) % 9) / 2 - 1;                                                         \
    if constexpr ((Q) == 1) load_gss(chunk0 + (T) / 9 + 1, 0);                                    \
    if constexpr ((Q) >= 2 && (Q) < 6) xform_elem(xr, (Q) - 2);                                   \
    if constexpr ((Q) == 6) load_gss(chunk0 + (T) / 9 + 1, 1);                                    \
    if constexpr ((Q) >= 7 && (Q) < 11) xform_elem(xr, (Q) - 3);                                  \
    if constexpr ((Q) == 11) store_slab_f(xr, 1 - (T) / 9, k_);                                   \
  }
#else
#define SLAB_ISSUE(T, Q)
#define SLAB_XF(T, Q)
#endif
#define SLAB_SLOT(CUR, NXT, T, Q)                                                   \
  SLAB_PAIR(CUR, Q)                                                                 \
  __builtin_amdgcn_sched_barrier(0);                                                \
  SLAB_ISSUE(T, Q)                                                                  \
  SLAB_READ(NXT, (((T) + 1) % 3), (((T) + 1) % 9), Q)                               \
  SLAB_XF(T, Q)                                                                     \
  __builtin_amdgcn_sched_barrier(0);
  // top-of-tap wait: the weights of tap T+1 (issued in tap T-1... see header) have landed once at most the DMAs issued after
  // them are outstanding: 2 (weights of T+2) + 2 more when tap T-1 also issued a slab third (T-1 in 0..2 of its chunk).
  // GN: thirds are issued at (tap 0, slot 0) and (taps 2 / 4, slot 12), i.e. after the weights of tap T+1 when T-1 is 0, 2 or 4 or
  // T-2 is 2 or 4: two more at the top of taps 1, 3, 4, 5 and 6; the third itself is waited for by the compiler where the
  // transform first reads it (a tap and a half after its issue)
#ifndef CDMA_ABLATE_NO_DMA
#define SLAB_WAIT(T)                                                                     \
  if constexpr (GN ? (((T) % 9) == 1 || (((T) % 9) >= 3 && ((T) % 9) <= 6)) : (((T) % 9) >= 1 && ((T) % 9) <= 3))  \
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");                          \
  else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
#else
#define SLAB_WAIT(T) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
  // the base registers point into the slab buffer the NEXT tap's fragments come from: flipped when that tap starts a new chunk
  // (ds_read immediates are 16 bits, the two buffers span 84 KiB)
#define SLAB_TAP(CUR, NXT, T)                                                                         \
  SLAB_WAIT(T)                                                                                        \
  __builtin_amdgcn_s_barrier();                                                                       \
  if constexpr ((T) == 8) flip_base(SLAB);                                                            \
  if constexpr ((T) == 17) flip_base(-SLAB);                                                          \
  __builtin_amdgcn_sched_barrier(0);                                                                  \
  SLAB_SLOT(CUR, NXT, T, 0) SLAB_SLOT(CUR, NXT, T, 1) SLAB_SLOT(CUR, NXT, T, 2) SLAB_SLOT(CUR, NXT, T, 3)     \
  SLAB_SLOT(CUR, NXT, T, 4) SLAB_SLOT(CUR, NXT, T, 5) SLAB_SLOT(CUR, NXT, T, 6) SLAB_SLOT(CUR, NXT, T, 7)     \
  SLAB_SLOT(CUR, NXT, T, 8) SLAB_SLOT(CUR, NXT, T, 9) SLAB_SLOT(CUR, NXT, T, 10) SLAB_SLOT(CUR, NXT, T, 11)   \
  SLAB_SLOT(CUR, NXT, T, 12) SLAB_SLOT(CUR, NXT, T, 13) SLAB_SLOT(CUR, NXT, T, 14) SLAB_SLOT(CUR, NXT, T, 15)

  // prologue: slab of chunk 0 and the weights of taps 0, 1, 2 in flight; tap 0's fragments into f0
  int chunk0 = 0;
  u32x4 pr[GN ? 3 : 1][2];      // (prologue only: the fragment registers are still free)
  if constexpr (GN) { issue_slab_f(pr[0], 0, 0); issue_slab_f(pr[GN ? 1 : 0], 1, 0); issue_slab_f(pr[GN ? 2 : 0], 2, 0); }
  else { issue_slab(0, 0, 0); issue_slab(0, 1, 0); issue_slab(0, 2, 0); }
  issue_b(0, 0, 0); issue_b(1, 0, 1); issue_b(2, 0, 2);
  CDMA_TS(1)
  if constexpr (GN) {   // chunk 0's slab through the register path before the loop (the compiler waits for its six loads)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      load_gss(0, h);
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) xform_elem(pr[GN ? k : 0], h * 4 + e);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) store_slab_f(pr[GN ? k : 0], 0, k);
  }
  asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  CDMA_TS(2)
  SLAB_READ(f0, 0, 0, 0) SLAB_READ(f0, 0, 0, 1) SLAB_READ(f0, 0, 0, 2) SLAB_READ(f0, 0, 0, 3)
  SLAB_READ(f0, 0, 0, 4) SLAB_READ(f0, 0, 0, 5) SLAB_READ(f0, 0, 0, 6) SLAB_READ(f0, 0, 0, 7)
  SLAB_READ(f0, 0, 0, 8) SLAB_READ(f0, 0, 0, 9) SLAB_READ(f0, 0, 0, 10) SLAB_READ(f0, 0, 0, 11)
  SLAB_READ(f0, 0, 0, 12) SLAB_READ(f0, 0, 0, 13) SLAB_READ(f0, 0, 0, 14) SLAB_READ(f0, 0, 0, 15)
  __builtin_amdgcn_sched_barrier(0);

  // two channel chunks (18 taps) per iteration: slab buffer = chunk parity, weight stage = tap % 3, registers ping-pong
  const int nchunks = Cin >> 5;
  for (; chunk0 < nchunks; chunk0 += 2) {
    SLAB_TAP(f0, f1, 0) SLAB_TAP(f1, f0, 1) SLAB_TAP(f0, f1, 2) SLAB_TAP(f1, f0, 3) SLAB_TAP(f0, f1, 4) SLAB_TAP(f1, f0, 5)
    SLAB_TAP(f0, f1, 6) SLAB_TAP(f1, f0, 7) SLAB_TAP(f0, f1, 8) SLAB_TAP(f1, f0, 9) SLAB_TAP(f0, f1, 10) SLAB_TAP(f1, f0, 11)
    SLAB_TAP(f0, f1, 12) SLAB_TAP(f1, f0, 13) SLAB_TAP(f0, f1, 14) SLAB_TAP(f1, f0, 15) SLAB_TAP(f0, f1, 16) SLAB_TAP(f1, f0, 17)
  }
#undef SLAB_TAP
#undef SLAB_WAIT
#undef SLAB_SLOT
#undef SLAB_XF
#undef SLAB_ISSUE
#undef SLAB_PAIR
#undef SLAB_READ
#undef SLAB_CP
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // trailing (zero) DMAs and fragment reads done before LDS is reused
  __builtin_amdgcn_sched_barrier(0);
  CDMA_TS(3)

  // ---- epilogue: + bias, staged through LDS, 16-byte row-contiguous stores with the residual added (tile row = patch
  //      pixel py * 16 + px), GroupNorm partials of the output ----
  float bias_v[NI][4];
#pragma unroll
  for (int j = 0; j < NI; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + wn * 64 + j * 16 + 4 * (lane >> 4) + r;
      bias_v[j][r] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
    }
  double gs = 0.0, gq = 0.0;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int lr = wm * 64 + i * 16 + (lane & 15);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int nl = wn * 64 + j * 16 + 4 * (lane >> 4);
      const f32x4 v = {acc[i][j][0] + bias_v[j][0], acc[i][j][1] + bias_v[j][1], acc[i][j][2] + bias_v[j][2], acc[i][j][3] + bias_v[j][3]};
      *(f32x4*)(smem + lr * CST + nl * 4) = v;
    }
  }
  __syncthreads();
  CDMA_TS(4)
  constexpr int NIT = (BM * 32) / NT;   // 16 chunks of 16 bytes per thread: patch row `it`, pixel rbase
  const int col = (threadIdx.x & 31) * 4, n = n0 + col, rbase = threadIdx.x >> 5;
  const long pix0 = ((long)img * H + y0) * W + x0 + rbase;
  if (n < p.N) {
    f32x4 rv[NIT];
    if (p.residual) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) rv[it] = *(const f32x4*)(p.residual + (pix0 + (long)it * W) * p.N + n);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      f32x4 w = *(const f32x4*)(smem + (rbase + 16 * it) * CST + col * 4);
      if (p.residual) w += rv[it];
#ifndef CDMA_ABLATE_NO_STORE   // (timing experiments: scripts/exp/conv_seam.sh)
      *(f32x4*)(p.out + (pix0 + (long)it * W) * p.N + n) = w;
#else
      if (w[0] == 123.456f) p.out[0] = w[1];   // keeps the staged read alive, stores nothing
#endif
      if (p.gn_partial) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { gs += (double)w[e]; gq += (double)w[e] * (double)w[e]; }
      }
    }
  }
  if (p.gn_partial) {
    gs += __shfl_xor(gs, 32, 64);
    gq += __shfl_xor(gq, 32, 64);
    __syncthreads();
    double* red = (double*)smem;   // [8 waves][32 chunks][2]
    if (lane < 32) { red[(wave * 32 + lane) * 2] = gs; red[(wave * 32 + lane) * 2 + 1] = gq; }
    __syncthreads();
    if (threadIdx.x < 32) {
      double cs = 0.0, cq = 0.0;
#pragma unroll
      for (int w8 = 0; w8 < 8; ++w8) { cs += red[(w8 * 32 + threadIdx.x) * 2]; cq += red[(w8 * 32 + threadIdx.x) * 2 + 1]; }
      const int cpc = p.gn_cpg >> 2;
      for (int o = 1; o < cpc; o <<= 1) { cs += __shfl_xor(cs, o, 64); cq += __shfl_xor(cq, o, 64); }
      const int nn = n0 + (int)threadIdx.x * 4;
      if ((threadIdx.x & (cpc - 1)) == 0 && nn < p.N) {
        double* o2 = p.gn_partial + (((long)img * tpi + prem) * p.gn_groups + nn / p.gn_cpg) * 2;   // chunk = patch index in the image
        o2[0] = cs; o2[1] = cq;
      }
    }
  }
  CDMA_TS(5)
}


// ---------------------------------------------------------------------------------------------------------------------
// Persistent form of the patch-slab kernel: one block per CU walks its tiles, and the K loop's own look-ahead (next chunk's
// slab in taps 0-2, weights three taps ahead, fragments one tap ahead) simply runs on INTO THE NEXT TILE: when a tile's last
// tap retires, the next tile's first slab, its first three weight stages and its tap-0 fragments are already on chip.
// What is left between two tiles is the epilogue alone - no block launch, no address set-up, no cold DMA round trip (the
// non-persistent kernel spends ~13 us of a 43 us tile there).
//   LDS map (bytes): slab buffer 0 [0, 43008) | weight stages 1, 2, 0 [43008, 92160) | slab buffer 1 [92160, 135168) | spare.
//   At a tile boundary buffer 0 and the three weight stages hold the NEXT tile's operands; buffer 1 and the spare bytes behind
//   it are free: the f32 output tile is staged there in two passes of 128 rows (67584 bytes, 159744 in all).
//   vmcnt: the epilogue drains every load before its first store, so the stores are the only old entries in the queue when
//   the next tile starts; taps 0 and 1 of a tile's first iteration need no vector-memory wait at all (their operands were
//   waited for in the previous tile / the epilogue), from tap 2 on the counted waits of the K loop apply unchanged (loads
//   and stores retire in order within their kind; a counted wait that still sees stores is merely conservative).
constexpr int P_BUF1 = 92160, P_BBASE = 43008, P_STAGING = P_BUF1, P_HALF_ROWS = 128;
constexpr int P_LDS_BYTES = P_STAGING + P_HALF_ROWS * CST;   // 159744
// GN = true: two [2][Cin] f32 scale / shift tables behind the tile's map (this tile's image, the next tile's image): Cin <= 256
constexpr int P_GSS = P_LDS_BYTES, P_GSS_MAX_CIN = 256, P_LDS_BYTES_GN = P_LDS_BYTES + 2 * 2 * P_GSS_MAX_CIN * 4;   // 163840 = all of it
__host__ __device__ constexpr int p_stage_off(int st) { return st == 0 ? 32768 : (st - 1) * 16384; }   // relative to P_BBASE
static_assert(P_BBASE + 3 * BSTAGE == P_BUF1 && P_BUF1 + SLAB <= P_LDS_BYTES && P_LDS_BYTES_GN <= 163840, "LDS map");

// GN = true is the persistent form of conv_slab_kernel<true> (round 6): the f32 activation goes through registers, GroupNorm affine +
// SiLU + hi / lo split are applied on the way into the slab buffer - same slots, same arithmetic, same bits as the launch-per-tile
// kernel - and the look-ahead across the tile boundary carries the NEXT tile's first chunk through that same register path with the
// next tile's image's scale / shift (second table, fetched by one 256-byte LDS-DMA per wave early in the tile).
template <bool GN>
__global__ __launch_bounds__(512, 2) void conv_slab_persist_kernel(Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int H = p.H, W = p.W, Cin = p.Cin;
  const int tpr = W >> 4, tpi = (H >> 4) * tpr;
  const int ntm = p.M >> 8, ntn = (p.N + BN - 1) / BN, ntiles = ntm * ntn;
  const int bq = ntiles >> 3, br = ntiles & 7;
  const int nblocks = gridDim.x;

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  constexpr int XE = GN ? 4 : 2;             // bytes per activation element in global memory
  const unsigned bytesA = (unsigned)((long)p.M * Cin * XE);
  const rsrc_t rs_xh = make_rsrc(GN ? (const void*)p.xf : (const void*)p.xh, bytesA), rs_xl = make_rsrc(GN ? (const void*)p.xf : (const void*)p.xl, bytesA);
  const unsigned oobA = (bytesA + 15u) & ~15u;
  const int wimg = wave >> 2;
  const unsigned bytesB = (unsigned)((long)p.N * p.K * 2);
  const rsrc_t rs_w = make_rsrc(wimg ? p.wl : p.wh, bytesB);
  const unsigned oobB = (bytesB + 15u) & ~15u;
  const int nchunks = Cin >> 5;
  const unsigned bytesO = (unsigned)((long)p.M * p.N * 4);   // < 4 GiB (host check)
  const rsrc_t rs_out = make_rsrc(p.out, bytesO), rs_res = make_rsrc(p.residual ? (const void*)p.residual : (const void*)p.out, bytesO);

  int slab_piece[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { int piece = wave + 8 * k; if (piece > 20) piece -= 8; slab_piece[k] = piece; }
  const int srcchunkA = (lane & 3) ^ (((lane >> 4) & 1) << 1);

  // geometry of virtual block v (the XCD-aware walk of the launch-per-tile kernels: v -> tile)
  struct Geo { int img, prem, y0, x0, n0; };
  auto geo_of = [&](int v) {
    const int xcd = v & 7, bi = v >> 3;
    const int tid_ = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + bi;
    const int mt = tid_ / ntn;
    Geo g;
    g.n0 = (tid_ - mt * ntn) * BN;
    g.img = mt / tpi; g.prem = mt - g.img * tpi;
    const int ty = g.prem / tpr, tx = g.prem - ty * tpr;
    g.y0 = ty << 4; g.x0 = tx << 4;
    return g;
  };
  // (`ln` = the lane id behind an empty asm: per-lane geometry is recomputed per tile instead of living in registers across the K loop)
  auto offsets_of = [&](const Geo& g, bool valid, unsigned (&so)[3], unsigned (&vb)[2], int ln) {
    const int srcchunkA = (ln & 3) ^ (((ln >> 4) & 1) << 1);
    const int srcchunkB = (ln & 3) ^ sw4((ln >> 4) & 3);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int r = slab_piece[k] * 16 + (ln >> 2), sy = r / 18, sx = r - sy * 18;
      const int y = g.y0 - 1 + sy, x = g.x0 - 1 + sx;
      const bool ok = valid && r < 324 && y >= 0 && y < H && x >= 0 && x < W;
      so[k] = ok ? (unsigned)(((((long)g.img * H + y) * W + x) * Cin + srcchunkA * 8) * XE) : oobA;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = g.n0 + ((wave & 3) * 2 + j) * 16 + (ln >> 2);
      vb[j] = (valid && n < p.N) ? (unsigned)(((long)n * p.K + srcchunkB * 8) * 2) : oobB;
    }
  };

  unsigned slab_off[3], voffB[2], slab_off_n[3], voffB_n[2];
  // slab third k of chunk `chunk` of the current tile, or (chunk == nchunks) of chunk 0 of the next tile
  auto issue_slab = [&](int buf, int k, int chunk) {
    const bool cur = chunk < nchunks;
    const unsigned vo = cur ? slab_off[k] : slab_off_n[k];
    const int soff = cur ? chunk * 64 : 0;
    unsigned char* dst = smem + (buf ? P_BUF1 : 0) + slab_piece[k] * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_xh, (lds_void_t*)dst, 16, (int)vo, soff, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_xl, (lds_void_t*)(dst + PLANE), 16, (int)vo, soff, 0, 0);
  };
  auto issue_b = [&](int stage_off, int chunk, int tap) {
    const bool cur = chunk < nchunks;
    const int soff = (tap * Cin + (cur ? chunk * 32 : 0)) * 2;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void_t*)(smem + P_BBASE + stage_off + wimg * 8192 + ((wave & 3) * 2 + j) * 1024),
                                               16, (int)(cur ? voffB[j] : voffB_n[j]), soff, 0, 0);
  };

  // ---- GN: the register path of the slab (conv_slab_kernel<true>), with the tile-boundary case ----
  float* gss_cur = (float*)(smem + P_GSS);                       // [2][Cin] of the current tile's image
  float* gss_nxt = gss_cur + 2 * P_GSS_MAX_CIN;                  // ... of the next tile's image
  u32x4 xr[2];
  float scv[4], shv[4];
  auto issue_slab_f = [&](u32x4 (&r)[2], int k, int chunk) {
    const bool cur = chunk < nchunks;
    const unsigned vo = cur ? slab_off[k] : slab_off_n[k];
    const unsigned soff = cur ? (unsigned)(chunk * 128) : 0u;
    r[0] = buf_load16(rs_xh, vo, soff);
    r[1] = buf_load16(rs_xh, vo + 16u, soff);
  };
  auto load_gss = [&](int chunk, int h) {
    const float* q = (chunk < nchunks ? gss_cur + chunk * 32 : gss_nxt) + srcchunkA * 8 + h * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) { scv[j] = q[j]; shv[j] = q[Cin + j]; }
  };
  auto xform_elem = [&](u32x4 (&r)[2], int e) {
    float t = fmaf(__uint_as_float(r[e >> 2][e & 3]), scv[e & 3], shv[e & 3]);
    t = gn_silu(t);
    asm volatile("" : "+v"(t));   // the ROUNDED product is what gets split: no contraction of (x * r) - hi into one fma
    r[e >> 2][e & 3] = __float_as_uint(t);
  };
  auto store_slab_f = [&](u32x4 (&r)[2], int buf, int k, int chunk) {
    u32x4 hi4, lo4;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      u32x2 hi, lo;
      split4(r[h], hi, lo);
      hi4[2 * h] = hi[0]; hi4[2 * h + 1] = hi[1];
      lo4[2 * h] = lo[0]; lo4[2 * h + 1] = lo[1];
    }
    if ((chunk < nchunks ? slab_off[k] : slab_off_n[k]) == oobA) { hi4 = u32x4{0u, 0u, 0u, 0u}; lo4 = hi4; }
    unsigned char* dst = smem + (buf ? P_BUF1 : 0) + slab_piece[k] * 1024 + lane * 16;
    *(u32x4*)dst = hi4;
    *(u32x4*)(dst + PLANE) = lo4;
  };
  // scale / shift of image `img` into table `dst`: 2 * Cin / 64 pieces of 64 floats, one LDS-DMA per wave (the pieces repeat over the waves)
  auto issue_gss = [&](float* dst, int img) {
    const int npc = Cin >> 6, piece = wave % (2 * npc), isshift = piece >= npc ? 1 : 0, pc = piece - isshift * npc;
    const int batch = p.M / (H * W);
    const rsrc_t rs = make_rsrc(isshift ? p.gsh : p.gsc, (unsigned)((long)batch * Cin * 4));
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(dst + isshift * Cin + pc * 64), 4, (int)((unsigned)((long)img * Cin + pc * 64 + lane) * 4u), 0, 0, 0);
  };

  unsigned baseA[4][2], addrB;
  {
    const int pr = lane & 15, g = lane >> 4, P = 72 * wm + pr;
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int bit2 = ((P >> 2) & 1) ^ (((P & 3) + e) >> 2) ^ b;
        baseA[e][b] = (unsigned)((P << 6) | ((g << 4) ^ (bit2 << 5)));
      }
    addrB = (unsigned)(P_BBASE + (wn * 64 + pr) * 64 + ((g ^ sw4((pr >> 2) & 3)) << 4));
  }
  auto flip_base = [&](int d) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { baseA[e][0] += d; baseA[e][1] += d; }
  };

  // LDS-only barrier: a full block barrier is also a fence for global memory, i.e. an `s_waitcnt vmcnt(0)` on the output
  // stores still in flight - exactly the wait the persistent walk exists to avoid
  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  f32x4 acc[MI][NI];
  Frag16 f0, f1;

#define SLAB_CP(TAP, I) (((I) + (TAP) / 3) * 18 + (TAP) % 3)
#define SLAB_READ(F, ST, TAP, Q)                                                                                                  \
  {                                                                                                                               \
    constexpr int q_ = (Q) & 3;                                                                                                   \
    constexpr int cp_ = SLAB_CP(TAP, q_);                                                                                         \
    if constexpr ((Q) < 4) lds_read128<cp_ * 64>(F.ah[q_], baseA[cp_ & 3][(cp_ >> 2) & 1]);                                       \
    else if constexpr ((Q) < 8) lds_read128<PLANE + cp_ * 64>(F.al[q_], baseA[cp_ & 3][(cp_ >> 2) & 1]);                          \
    else if constexpr ((Q) < 12) lds_read128<p_stage_off(ST) + q_ * 1024>(F.bh[q_], addrB);                                       \
    else lds_read128<p_stage_off(ST) + 8192 + q_ * 1024>(F.bl[q_], addrB);                                                        \
  }
#define SLAB_PAIR(F, Q)                                                                                                        \
  {                                                                                                                            \
    constexpr int i_ = (Q) >> 2, j_ = (Q) & 3;                                                                                 \
    acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F.bl[j_], F.ah[i_], acc[i_][j_], 0, 0, 0);                            \
    acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F.bh[j_], F.al[i_], acc[i_][j_], 0, 0, 0);                            \
    acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F.bh[j_], F.ah[i_], acc[i_][j_], 0, 0, 0);                            \
  }
  // GN: thirds issued at (tap 0, slot 0) and (taps 2 / 4, slot 12), transformed in taps 2 / 4 / 6 - the slots of conv_slab_kernel<true>;
  // the next tile's scale / shift table at (tap 3, slot 5) of a tile's first chunk pair (landed and barrier-published long before
  // tap 11 first reads it)
#define SLAB_ISSUE(T, Q)                                                                          \
  if constexpr (!GN && (Q) == 0 && ((T) % 9) < 3) issue_slab(1 - (T) / 9, (T) % 9, chunk0 + (T) / 9 + 1);  \
  if constexpr (GN && (Q) == 0 && ((T) % 9) == 0) issue_slab_f(xr, 0, chunk0 + (T) / 9 + 1);           \
  if constexpr (GN && (Q) == 12 && (((T) % 9) == 2 || ((T) % 9) == 4)) issue_slab_f(xr, ((T) % 9) / 2, chunk0 + (T) / 9 + 1);  \
  if constexpr (GN && (T) == 3 && (Q) == 5) { if (first) issue_gss(gss_nxt, gn.img); }               \
  if constexpr ((Q) == 2) issue_b(p_stage_off(((T) + 3) % 3), chunk0 + ((T) + 3) / 9, ((T) + 3) % 9);
#define SLAB_XF(T, Q)                                                                             \
  if constexpr (GN && ((T) % 9) >= 2 && ((T) % 9) <= 6 && ((T) % 9) % 2 == 0) {                    \
    constexpr int k_ = ((T) % 9) / 2 - 1;                                                         \
    if constexpr ((Q) == 1) load_gss(chunk0 + (T) / 9 + 1, 0);                                    \
    if constexpr ((Q) >= 2 && (Q) < 6) xform_elem(xr, (Q) - 2);                                   \
    if constexpr ((Q) == 6) load_gss(chunk0 + (T) / 9 + 1, 1);                                    \
    if constexpr ((Q) >= 7 && (Q) < 11) xform_elem(xr, (Q) - 3);                                  \
    if constexpr ((Q) == 11) store_slab_f(xr, 1 - (T) / 9, k_, chunk0 + (T) / 9 + 1);             \
  }
#define SLAB_SLOT(CUR, NXT, T, Q)                                                   \
  SLAB_PAIR(CUR, Q)                                                                 \
  __builtin_amdgcn_sched_barrier(0);                                                \
  SLAB_ISSUE(T, Q)                                                                  \
  SLAB_READ(NXT, (((T) + 1) % 3), (((T) + 1) % 9), Q)                               \
  SLAB_XF(T, Q)                                                                     \
  __builtin_amdgcn_sched_barrier(0);
  // counted waits as in the launch-per-tile kernels (GN: two more at the top of taps 1, 3, 4, 5, 6 of a chunk); the extra scale /
  // shift DMA of a tile's first pass makes the waits of its taps 4 and 5 merely stricter
#define SLAB_WAIT(T)                                                                                                        \
  if constexpr ((T) < 2) {                                                                                                  \
    if (first) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                          \
    else if constexpr ((T) == 1) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");                               \
    else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");                                                        \
  } else if constexpr (GN ? (((T) % 9) == 1 || (((T) % 9) >= 3 && ((T) % 9) <= 6)) : (((T) % 9) >= 1 && ((T) % 9) <= 3))   \
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");                                                             \
  else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
#define SLAB_TAP(CUR, NXT, T)                                                                         \
  SLAB_WAIT(T)                                                                                        \
  __builtin_amdgcn_s_barrier();                                                                       \
  if constexpr ((T) == 8) flip_base(P_BUF1);                                                          \
  if constexpr ((T) == 17) flip_base(-P_BUF1);                                                        \
  __builtin_amdgcn_sched_barrier(0);                                                                  \
  SLAB_SLOT(CUR, NXT, T, 0) SLAB_SLOT(CUR, NXT, T, 1) SLAB_SLOT(CUR, NXT, T, 2) SLAB_SLOT(CUR, NXT, T, 3)     \
  SLAB_SLOT(CUR, NXT, T, 4) SLAB_SLOT(CUR, NXT, T, 5) SLAB_SLOT(CUR, NXT, T, 6) SLAB_SLOT(CUR, NXT, T, 7)     \
  SLAB_SLOT(CUR, NXT, T, 8) SLAB_SLOT(CUR, NXT, T, 9) SLAB_SLOT(CUR, NXT, T, 10) SLAB_SLOT(CUR, NXT, T, 11)   \
  SLAB_SLOT(CUR, NXT, T, 12) SLAB_SLOT(CUR, NXT, T, 13) SLAB_SLOT(CUR, NXT, T, 14) SLAB_SLOT(CUR, NXT, T, 15)

  // ---- first tile: its operands as the "next tile" of an empty predecessor ----
  int v = blockIdx.x;
  Geo g = geo_of(v);
  offsets_of(g, true, slab_off_n, voffB_n, lane);
  if constexpr (GN) {
    // chunk 0 of the first tile through the register path, its image's table by plain loads
    for (int c = threadIdx.x; c < Cin; c += NT) { gss_nxt[c] = p.gsc[(long)g.img * Cin + c]; gss_nxt[Cin + c] = p.gsh[(long)g.img * Cin + c]; }
    u32x4 pr[3][2];
    issue_slab_f(pr[0], 0, nchunks); issue_slab_f(pr[1], 1, nchunks); issue_slab_f(pr[2], 2, nchunks);
    issue_b(p_stage_off(0), nchunks, 0); issue_b(p_stage_off(1), nchunks, 1); issue_b(p_stage_off(2), nchunks, 2);
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      load_gss(nchunks, h);
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) xform_elem(pr[k], h * 4 + e);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) store_slab_f(pr[k], 0, k, nchunks);
  } else {
    issue_slab(0, 0, nchunks); issue_slab(0, 1, nchunks); issue_slab(0, 2, nchunks);
    issue_b(p_stage_off(0), nchunks, 0); issue_b(p_stage_off(1), nchunks, 1); issue_b(p_stage_off(2), nchunks, 2);
  }
  // (Measured and rejected: starting the blocks in eight phases an eighth of a tile apart, to keep 256 CUs from writing their
  //  32 MB of output at the same instant.  scripts/exp/conv_seam.py: T(Cin) = 13-15 us + 8.1 us per 32-channel chunk per tile
  //  with or without the stagger, launch-per-tile or persistent.)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  SLAB_READ(f0, 0, 0, 0) SLAB_READ(f0, 0, 0, 1) SLAB_READ(f0, 0, 0, 2) SLAB_READ(f0, 0, 0, 3)
  SLAB_READ(f0, 0, 0, 4) SLAB_READ(f0, 0, 0, 5) SLAB_READ(f0, 0, 0, 6) SLAB_READ(f0, 0, 0, 7)
  SLAB_READ(f0, 0, 0, 8) SLAB_READ(f0, 0, 0, 9) SLAB_READ(f0, 0, 0, 10) SLAB_READ(f0, 0, 0, 11)
  SLAB_READ(f0, 0, 0, 12) SLAB_READ(f0, 0, 0, 13) SLAB_READ(f0, 0, 0, 14) SLAB_READ(f0, 0, 0, 15)
  __builtin_amdgcn_sched_barrier(0);

  for (;;) {
    // this tile's offsets are the ones prefetched as "next"; then look one tile ahead
#pragma unroll
    for (int k = 0; k < 3; ++k) slab_off[k] = slab_off_n[k];
    voffB[0] = voffB_n[0]; voffB[1] = voffB_n[1];
    if constexpr (GN) { float* t = gss_cur; gss_cur = gss_nxt; gss_nxt = t; }
    const int vn = v + nblocks;
    const bool has_next = vn < ntiles;
    const Geo gn = geo_of(has_next ? vn : v);
    {
      int ln = lane;
      asm volatile("" : "+v"(ln));
      offsets_of(gn, has_next, slab_off_n, voffB_n, ln);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    CDMA_TSV(v, 0)
    bool first = true;
    for (int chunk0 = 0; chunk0 < nchunks; chunk0 += 2) {
      SLAB_TAP(f0, f1, 0) SLAB_TAP(f1, f0, 1) SLAB_TAP(f0, f1, 2) SLAB_TAP(f1, f0, 3) SLAB_TAP(f0, f1, 4) SLAB_TAP(f1, f0, 5)
      SLAB_TAP(f0, f1, 6) SLAB_TAP(f1, f0, 7) SLAB_TAP(f0, f1, 8) SLAB_TAP(f1, f0, 9) SLAB_TAP(f0, f1, 10) SLAB_TAP(f1, f0, 11)
      SLAB_TAP(f0, f1, 12) SLAB_TAP(f1, f0, 13) SLAB_TAP(f0, f1, 14) SLAB_TAP(f1, f0, 15) SLAB_TAP(f0, f1, 16) SLAB_TAP(f1, f0, 17)
      first = false;
    }

    CDMA_TSV(v, 1)
    // ---- epilogue of tile v.  (The tap-17 look-ahead reads already fetched the next tile's tap-0 fragments, but keeping them
    //      live through the epilogue - 64 more registers next to the accumulators and the residual tile - made hipcc spill into
    //      the store loop, and a spill reload is a `vmcnt(0)` behind every store.  They are re-read after the epilogue.) ----
    const int n0 = g.n0;
    constexpr int NITH = (P_HALF_ROWS * 32) / NT;   // 8 chunks of 16 bytes per thread and pass
    int tx = threadIdx.x;
    asm volatile("" : "+v"(tx));                    // (the epilogue's per-thread geometry is not to be hoisted over the K loop)
    const int lane_e = tx & 63;
    const int col = (tx & 31) * 4, n = n0 + col, rbase = tx >> 5;
    // output / residual through buffer descriptors: one 32-bit offset register per thread, the patch row as a scalar offset
    const long pixp = ((long)g.img * H + g.y0) * W + g.x0;   // first pixel of the patch
    const unsigned obase = (unsigned)(((pixp + rbase) * p.N + n) * 4);
    const unsigned rowstep = (unsigned)((long)W * p.N * 4);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // look-ahead DMAs and the tap-17 fragment reads
    __builtin_amdgcn_sched_barrier(0);
    CDMA_TSV(v, 2)
    // bias and residual are added to the accumulators in FRAGMENT layout (lane = pixel lane & 15 of patch row wm * 4 + i, four
    // channels), in the launch-per-tile kernels' order - (products + bias) + residual: same bits -, the residual one patch row ahead,
    // so that every global load of the epilogue retires before its first store: a load waited for behind a store is a wait for
    // that store's acknowledgement (reads and writes share vmcnt)
    {
      const int nf = n0 + wn * 64 + 4 * (lane_e >> 4);
      f32x4 bj[NI];
#pragma unroll
      for (int j = 0; j < NI; ++j)
        bj[j] = (p.bias && (nf + j * 16) < p.N) ? *(const f32x4*)(p.bias + nf + j * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.residual) {
        const unsigned fbase = (unsigned)((((pixp + (long)(wm * 4) * W + (lane_e & 15)) * p.N) + nf) * 4);
        u32x4 rr[2][NI];
#pragma unroll
        for (int j = 0; j < NI; ++j)
          rr[0][j] = (nf + j * 16) < p.N ? __builtin_amdgcn_raw_buffer_load_b128(rs_res, (int)(fbase + j * 64), 0, 0) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          if (i + 1 < MI) {
#pragma unroll
            for (int j = 0; j < NI; ++j)
              rr[(i + 1) & 1][j] = (nf + j * 16) < p.N ? __builtin_amdgcn_raw_buffer_load_b128(rs_res, (int)(fbase + j * 64), (int)(rowstep * (i + 1)), 0)
                                                       : u32x4{0u, 0u, 0u, 0u};
          }
#pragma unroll
          for (int j = 0; j < NI; ++j) acc[i][j] = (acc[i][j] + bj[j]) + __builtin_bit_cast(f32x4, rr[i & 1][j]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) acc[i][j] += bj[j];
      }
    }
    // nothing but loads in the queue: a cheap drain.  The builtin (not inline asm) so that hipcc's own wait-count bookkeeping sees the
    // drain here, in straight-line code, and inserts no `vmcnt(0)` of its own behind the stores below
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt untouched
    __builtin_amdgcn_sched_barrier(0);
    CDMA_TSV(v, 3)
    double gs = 0.0, gq = 0.0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      lds_barrier();   // the staging bytes are free (pass 0: everyone's fragment reads are done; pass 1: pass 0's rows are out)
      if ((wm >> 1) == half) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int lr = (wm & 1) * 64 + i * 16 + (lane_e & 15);
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            const int nl = wn * 64 + j * 16 + 4 * (lane_e >> 4);
            *(f32x4*)(smem + P_STAGING + lr * CST + nl * 4) = acc[i][j];
          }
        }
      }
      lds_barrier();
      if (n < p.N) {
#pragma unroll
        for (int it = 0; it < NITH; ++it) {
          const f32x4 w = *(const f32x4*)(smem + P_STAGING + (rbase + 16 * it) * CST + col * 4);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, w), rs_out, (int)obase, (int)(rowstep * (half * NITH + it)), 0);
          if (p.gn_partial) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { gs += (double)w[e]; gq += (double)w[e] * (double)w[e]; }
          }
        }
      }
    }
    CDMA_TSV(v, 4)
    if (p.gn_partial) {
      gs += __shfl_xor(gs, 32, 64);
      gq += __shfl_xor(gq, 32, 64);
      lds_barrier();
      double* red = (double*)(smem + P_STAGING);   // [8 waves][32 chunks][2]
      if (lane_e < 32) { red[(wave * 32 + lane_e) * 2] = gs; red[(wave * 32 + lane_e) * 2 + 1] = gq; }
      lds_barrier();
      if (tx < 32) {
        double cs = 0.0, cq = 0.0;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) { cs += red[(w8 * 32 + tx) * 2]; cq += red[(w8 * 32 + tx) * 2 + 1]; }
        const int cpc = p.gn_cpg >> 2;
        for (int o = 1; o < cpc; o <<= 1) { cs += __shfl_xor(cs, o, 64); cq += __shfl_xor(cq, o, 64); }
        const int nn = n0 + tx * 4;
        if ((tx & (cpc - 1)) == 0 && nn < p.N) {
          double* o2 = p.gn_partial + (((long)g.img * tpi + g.prem) * p.gn_groups + (nn >> __builtin_ctz(p.gn_cpg))) * 2;   // gn_cpg is a power of two (host check)
          o2[0] = cs; o2[1] = cq;
        }
      }
    }
    CDMA_TSV(v, 5)
    if (!has_next) break;
    v = vn; g = gn;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    SLAB_READ(f0, 0, 0, 0) SLAB_READ(f0, 0, 0, 1) SLAB_READ(f0, 0, 0, 2) SLAB_READ(f0, 0, 0, 3)
    SLAB_READ(f0, 0, 0, 4) SLAB_READ(f0, 0, 0, 5) SLAB_READ(f0, 0, 0, 6) SLAB_READ(f0, 0, 0, 7)
    SLAB_READ(f0, 0, 0, 8) SLAB_READ(f0, 0, 0, 9) SLAB_READ(f0, 0, 0, 10) SLAB_READ(f0, 0, 0, 11)
    SLAB_READ(f0, 0, 0, 12) SLAB_READ(f0, 0, 0, 13) SLAB_READ(f0, 0, 0, 14) SLAB_READ(f0, 0, 0, 15)
    __builtin_amdgcn_sched_barrier(0);
    // (the top-of-tap-0 wait + barrier of the next tile orders its first DMAs / fragment reads behind these staging reads)
  }
#undef SLAB_TAP
#undef SLAB_WAIT
#undef SLAB_SLOT
#undef SLAB_XF
#undef SLAB_ISSUE
#undef SLAB_PAIR
#undef SLAB_READ
#undef SLAB_CP
}

}  // namespace cslab

#ifdef CDMA_TIMESTAMPS
extern "C" int muse_debug_conv_ts(long* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(cslab::g_conv_ts), &buf, sizeof(buf)); }
#endif
extern "C" int muse_conv2d_nhwc_split2(const void* in_hi, const void* in_lo, const void* w_hi, const void* w_lo, const float* bias,
                                       const float* residual, float* out, double* gn_partial, int32_t gn_groups, int32_t batch,
                                       int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t KS, void* stream) {
  if (KS != 3) return MUSE_ERR_UNSUPPORTED;
  if ((Cin % 32) || (Cout % 4)) return MUSE_ERR_ALIGN;
  if ((((uintptr_t)in_hi) | ((uintptr_t)in_lo) | ((uintptr_t)w_hi) | ((uintptr_t)w_lo) | ((uintptr_t)out) | ((uintptr_t)residual)) & 15)
    return MUSE_ERR_ALIGN;
  const long M = (long)batch * H * W;
  if (M <= 0 || Cout <= 0) return 0;
  // 32-bit buffer offsets: each plane / weight image must stay below 4 GiB (and M below 2^31)
  if (M * Cin * 2 >= (1L << 32) - 64 || (long)Cout * 9 * Cin * 2 >= (1L << 32) - 64 || M >= (1L << 31) - 256) return MUSE_ERR_UNSUPPORTED;
  cdma::Params p;
  p.xh = (const bf16_t*)in_hi; p.xl = (const bf16_t*)in_lo; p.wh = (const bf16_t*)w_hi; p.wl = (const bf16_t*)w_lo;
  p.bias = bias; p.residual = residual; p.out = out;
  p.gn_partial = gn_partial; p.gn_groups = gn_groups; p.gn_cpg = gn_groups > 0 ? Cout / gn_groups : 0;
  if (gn_partial) {  // fused GroupNorm statistics: whole 256-pixel tiles per image, groups of 4 * 2^k channels inside one 128-channel tile
    const int cpg = p.gn_cpg;
    if (gn_groups <= 0 || (Cout % gn_groups) || ((H * W) % 256) || cpg < 4 || cpg > 128 || (cpg & (cpg - 1))) return MUSE_ERR_UNSUPPORTED;
  }
  p.M = (int)M; p.N = Cout; p.K = 9 * Cin; p.H = H; p.W = W; p.Cin = Cin;
  const int ntm = (p.M + cdma::BM - 1) / cdma::BM, ntn = (p.N + cdma::BN - 1) / cdma::BN;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)cdma::conv_dma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, cdma::LDS_BYTES);
    attr_set = true;
  }
  static const int use_slab = []() { const char* e = getenv("MUSE_CONV_SLAB"); return e ? atoi(e) : 1; }();
  if (use_slab && (H % 16) == 0 && (W % 16) == 0 && (Cin % 64) == 0) {   // patch-slab kernel (K order: chunk, tap, channel)
    static bool slab_attr = false;
    if (!slab_attr) {
      (void)hipFuncSetAttribute((const void*)cslab::conv_slab_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, cslab::LDS_BYTES);
      slab_attr = true;
    }
    const int nslab = (p.M >> 8) * ntn;
    if (use_slab >= 2 && (long)p.M * p.N * 4 < (1L << 32) - 64) {   // persistent: one block per CU walks its tiles (32-bit output offsets)
      static int ncu = 0;
      if (!ncu) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return MUSE_ERR_UNSUPPORTED;
        ncu = prop.multiProcessorCount;
        (void)hipFuncSetAttribute((const void*)cslab::conv_slab_persist_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, cslab::P_LDS_BYTES);
      }
      hipLaunchKernelGGL(cslab::conv_slab_persist_kernel<false>, dim3(nslab < ncu ? nslab : ncu), dim3(512), cslab::P_LDS_BYTES, (hipStream_t)stream, p);
      return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(cslab::conv_slab_kernel<false>, dim3(nslab), dim3(512), cslab::LDS_BYTES, (hipStream_t)stream, p);
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL(cdma::conv_dma_kernel, dim3(ntm * ntn), dim3(512), cdma::LDS_BYTES, (hipStream_t)stream, p);
  return (int)hipGetLastError();
}

// GroupNorm(32) + SiLU of the input fused into the patch-slab convolution (conv_slab_kernel<true>): `x` is the f32 NHWC activation,
// gn_scale / gn_shift [batch][Cin] the per-image affine form of the normalisation (muse_groupnorm_scale_shift).  Same result, bit
// for bit, as muse_groupnorm_silu_nhwc_split followed by muse_conv2d_nhwc_split2.  Patch-slab shapes only (H, W multiples of 16,
// Cin a multiple of 64, Cin <= 2048); anything else returns MUSE_ERR_UNSUPPORTED and the caller keeps the two-kernel route.
// Persistent form of the fused convolution on / off for the launches that follow on this host thread's behalf (mode 0 / 1; -1 = query).
// Default: MUSE_CONV_PERSIST, else ON - the persistent kernel is 6-8 % faster whenever the convolution has the chip to itself (a tokenizer
// or decoder pass on its own: BASELINE config 5, inline tokenizing, pre-encoding).  muse.TrainStep switches it OFF around the tokenizer
// pass it enqueues BESIDE a train step: there a workgroup that holds its CU for a whole launch costs the step 2.2 ms (profiles/r06_ceiling.md).
extern "C" int muse_conv_persistent(int32_t mode) {
  static int state = []() { const char* e = getenv("MUSE_CONV_PERSIST"); return e ? atoi(e) : 1; }();
  if (mode >= 0) state = mode;
  return state;
}
extern "C" int muse_conv2d_nhwc_gn_split2_ok(int32_t batch, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t KS) {
  const long M = (long)batch * H * W;
  return KS == 3 && (H % 16) == 0 && (W % 16) == 0 && (Cin % 64) == 0 && Cin <= 2048 && (Cout % 4) == 0 && M > 0 &&
         M * Cin * 4 < (1L << 32) - 64 && (long)Cout * 9 * Cin * 2 < (1L << 32) - 64 && M < (1L << 31) - 256;
}
extern "C" int muse_conv2d_nhwc_gn_split2(const float* x, const float* gn_scale, const float* gn_shift, const void* w_hi, const void* w_lo,
                                          const float* bias, const float* residual, float* out, double* gn_partial, int32_t gn_groups,
                                          int32_t batch, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t KS, void* stream) {
  const long M = (long)batch * H * W;
  if (M <= 0 || Cout <= 0) return 0;
  if (!muse_conv2d_nhwc_gn_split2_ok(batch, H, W, Cin, Cout, KS)) return MUSE_ERR_UNSUPPORTED;
  if ((((uintptr_t)x) | ((uintptr_t)w_hi) | ((uintptr_t)w_lo) | ((uintptr_t)out) | ((uintptr_t)residual)) & 15) return MUSE_ERR_ALIGN;
  cdma::Params p;
  p.xh = p.xl = nullptr; p.xf = x; p.gsc = gn_scale; p.gsh = gn_shift;
  p.wh = (const bf16_t*)w_hi; p.wl = (const bf16_t*)w_lo;
  p.bias = bias; p.residual = residual; p.out = out;
  p.gn_partial = gn_partial; p.gn_groups = gn_groups; p.gn_cpg = gn_groups > 0 ? Cout / gn_groups : 0;
  if (gn_partial) {
    const int cpg = p.gn_cpg;
    if (gn_groups <= 0 || (Cout % gn_groups) || cpg < 4 || cpg > 128 || (cpg & (cpg - 1))) return MUSE_ERR_UNSUPPORTED;
  }
  p.M = (int)M; p.N = Cout; p.K = 9 * Cin; p.H = H; p.W = W; p.Cin = Cin;
  const int ntn = (p.N + cslab::BN - 1) / cslab::BN;
  // persistent form (round 6): one workgroup per CU walks the tiles, the K loop's look-ahead runs on into the next tile.  Taken when a
  // CU gets at least MUSE_CONV_PERSIST_MIN tiles (default 2), Cin <= 256 (two scale / shift tables behind the LDS map) and the output
  // stays below 4 GiB (32-bit store offsets).  MUSE_CONV_PERSIST=0 keeps the launch-per-tile kernel; MUSE_CONV_PERSIST_GRID sets the
  // number of workgroups (default: one per CU).
  const int persist = muse_conv_persistent(-1);
  if (persist && Cin <= cslab::P_GSS_MAX_CIN && (long)p.M * p.N * 4 < (1L << 32) - 64) {
    static int ncu = 0, grid = 0, min_tiles = 2, min_given = 0;
    if (!ncu) {
      int dev = 0; hipDeviceProp_t prop;
      if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return MUSE_ERR_UNSUPPORTED;
      (void)hipFuncSetAttribute((const void*)cslab::conv_slab_persist_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, cslab::P_LDS_BYTES_GN);
      const char* e = getenv("MUSE_CONV_PERSIST_GRID");
      grid = e ? atoi(e) : prop.multiProcessorCount;
      if (grid < 1) grid = 1;
      const char* m = getenv("MUSE_CONV_PERSIST_MIN");
      if (m) { min_tiles = atoi(m); min_given = 1; }
      ncu = prop.multiProcessorCount;
    }
    const int nslab = (p.M >> 8) * ntn;
    // MUSE_CONV_PERSIST_TILES=k: workgroups of k tiles each (grid = tiles / k) instead of one workgroup per CU for the whole launch - a CU
    // is handed back to the dispatcher (and to the other streams' kernels) every k tiles
    static const int per_wg = []() { const char* e = getenv("MUSE_CONV_PERSIST_TILES"); return e ? atoi(e) : 0; }();
    if (per_wg > 0) {
      if (nslab >= (min_given ? (min_tiles > 1 ? min_tiles : 1) * per_wg : 2 * per_wg * ncu)) {
        hipLaunchKernelGGL(cslab::conv_slab_persist_kernel<true>, dim3((nslab + per_wg - 1) / per_wg), dim3(512), cslab::P_LDS_BYTES_GN, (hipStream_t)stream, p);
        return (int)hipGetLastError();
      }
    } else if (nslab >= min_tiles * grid) {
      hipLaunchKernelGGL(cslab::conv_slab_persist_kernel<true>, dim3(nslab < grid ? nslab : grid), dim3(512), cslab::P_LDS_BYTES_GN, (hipStream_t)stream, p);
      return (int)hipGetLastError();
    }
  }
  constexpr int lds = cslab::LDS_BYTES + 2 * 2048 * 4;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)cslab::conv_slab_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr = true;
  }
  hipLaunchKernelGGL(cslab::conv_slab_kernel<true>, dim3((p.M >> 8) * ntn), dim3(512), cslab::LDS_BYTES + 2 * Cin * 4, (hipStream_t)stream, p);
  return (int)hipGetLastError();
}
