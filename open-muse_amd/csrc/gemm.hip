// Dense / strided-batched GEMM entry point of libmuse_hip (see include/muse_hip.h: muse_gemm).
#include "gemm_core.h"
#include "gemm256.h"
#include "gemm_p.h"
#include "../../include/muse_hip.h"

template <typename T, typename TC, int BM>
static int dispatch_bm(const GemmParams& p, int la, int lb, int batch, hipStream_t s) {
  constexpr int NT = BM * 2;
  if (la == 0 && lb == 0) return launch_gemm<T, TC, 0, 0, BM, PlainLoader<T, 0, BM, NT>, PlainLoader<T, 0, 128, NT>>(p, batch, s);
  if (la == 0 && lb == 1) return launch_gemm<T, TC, 0, 1, BM, PlainLoader<T, 0, BM, NT>, PlainLoader<T, 1, 128, NT>>(p, batch, s);
  if (la == 1 && lb == 1) return launch_gemm<T, TC, 1, 1, BM, PlainLoader<T, 1, BM, NT>, PlainLoader<T, 1, 128, NT>>(p, batch, s);
  if (la == 1 && lb == 0) return launch_gemm<T, TC, 1, 0, BM, PlainLoader<T, 1, BM, NT>, PlainLoader<T, 0, 128, NT>>(p, batch, s);
  return MUSE_ERR_BAD_ARG;
}
template <typename T, typename TC>
static int dispatch_layout(const GemmParams& p, int la, int lb, int batch, hipStream_t s) {
  return use_bm256(p, batch) ? dispatch_bm<T, TC, 256>(p, la, lb, batch, s) : dispatch_bm<T, TC, 128>(p, la, lb, batch, s);
}

static int fill_params(const muse_gemm_desc* d, GemmParams& p) {
  if (!d || !d->A || !d->B || !d->C) return MUSE_ERR_BAD_ARG;
  const int esz = (d->dtype == MUSE_BF16 || d->dtype == MUSE_F16) ? 2 : 4;
  const int ch = 16 / esz;
  // 16-byte vector loads: leading dimensions and sub-matrix offsets must be multiples of one chunk
  if ((d->lda % ch) || (d->ldb % ch) || (((uintptr_t)d->A) & 15) || (((uintptr_t)d->B) & 15)) return MUSE_ERR_ALIGN;
  if ((d->sA0 % ch) || (d->sA1 % ch) || (d->sB0 % ch) || (d->sB1 % ch)) return MUSE_ERR_ALIGN;
  {  // the operand loaders address one batch slice with 32-bit byte offsets through a buffer descriptor: a slice of 4 GiB or more
     // would wrap silently (e.g. f32 logits [131072, 8192] as a wgrad operand) - refuse it instead
    const int64_t ra = d->layout_a == 0 ? d->M : d->K, ca = d->layout_a == 0 ? d->K : d->M;
    const int64_t rb = d->layout_b == 0 ? d->N : d->K, cb = d->layout_b == 0 ? d->K : d->N;
    if (((ra - 1) * d->lda + ca + ch) * esz >= (1LL << 32) || ((rb - 1) * d->ldb + cb + ch) * esz >= (1LL << 32)) return MUSE_ERR_UNSUPPORTED;
  }
  p.A = d->A; p.B = d->B; p.C = d->C;
  p.bias = (const float*)d->bias; p.rowvec = (const float*)d->rowvec; p.residual = d->residual;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc; p.ldr = d->ldr;
  p.zdiv = d->zdiv > 0 ? d->zdiv : 1;
  p.sA0 = d->sA0; p.sA1 = d->sA1; p.sB0 = d->sB0; p.sB1 = d->sB1; p.sC0 = d->sC0; p.sC1 = d->sC1;
  p.alpha = d->alpha; p.accumulate = d->accumulate; p.act = d->act;
  p.split_k = d->split_k > 1 ? d->split_k : 1;
  p.split_stride = p.split_k > 1 ? d->split_stride : 0;
  if (p.split_k > 1 && (d->out_dtype != MUSE_F32 || d->bias || d->rowvec || d->residual || d->act)) return MUSE_ERR_BAD_ARG;
  p.cH = p.cW = p.cCin = p.cKS = p.cUps = 0; p.cCinShift = -1;
  p.a_lo = p.b_lo = 0;
  return 0;
}

// does this descriptor go to the 256 x 256 LDS-DMA kernel (gemm256.h)?
static bool takes_gemm256(const muse_gemm_desc* d, const GemmParams& p, int batch) {
  if (d->dtype == MUSE_F16) {
    // half operands exist on the 256^2 kernels only, with f32 output: no cost model - whatever those kernels take (the caller keeps
    // a product they refuse in exact f32)
    return d->out_dtype == MUSE_F32 && d->M >= 128 && d->N >= 128 && d->K >= 64 && gemm256_ok<float>(p, d->layout_a, d->layout_b);
  }
  if (d->dtype != MUSE_BF16) return false;
  const bool pers = (d->out_dtype == MUSE_BF16 || d->out_dtype == MUSE_F32) &&
                    gemm256p_takes(p, d->layout_a, d->layout_b, batch, d->out_dtype == MUSE_F32);
  if (!gemm256_preferred(p, d->layout_a, d->layout_b, batch, pers)) return false;
  if (d->out_dtype == MUSE_BF16) return gemm256_ok<bf16_t>(p, d->layout_a, d->layout_b);
  if (d->out_dtype == MUSE_F32) return gemm256_ok<float>(p, d->layout_a, d->layout_b);
  return false;
}

extern "C" int muse_gemm_tile(const muse_gemm_desc* d) {
  GemmParams p;
  const int rc = fill_params(d, p);
  if (rc) return rc;
  if (takes_gemm256(d, p, d->batch > 0 ? d->batch : 1)) return 256;
  return d->dtype == MUSE_F16 ? MUSE_ERR_UNSUPPORTED : 128;
}

extern "C" int muse_gemm_path(const muse_gemm_desc* d) {
  GemmParams p;
  const int rc = fill_params(d, p);
  if (rc) return rc;
  const int batch = d->batch > 0 ? d->batch : 1;
  if (!takes_gemm256(d, p, batch)) return d->dtype == MUSE_F16 ? MUSE_ERR_UNSUPPORTED : 128;
  return gemm256p_takes(p, d->layout_a, d->layout_b, batch, d->out_dtype == MUSE_F32) ? 257 : 256;
}

extern "C" int muse_gemm(const muse_gemm_desc* d, void* stream) {
  GemmParams p;
  const int rc = fill_params(d, p);
  if (rc) return rc;
  const int batch = d->batch > 0 ? d->batch : 1;
  hipStream_t s = (hipStream_t)stream;
  if (d->dtype == MUSE_F16 && !takes_gemm256(d, p, batch)) return MUSE_ERR_UNSUPPORTED;
  if (takes_gemm256(d, p, batch)) {
    // persistent tile-walking form first (gemm256p.h); -1 = no queue slot for this stream
    const bool half_ops = d->dtype == MUSE_F16;
    if (gemm256p_takes(p, d->layout_a, d->layout_b, batch, d->out_dtype == MUSE_F32)) {
      const int rp = launch_gemm256p(p, d->layout_a, d->layout_b, d->out_dtype == MUSE_F32, s, half_ops);
      if (rp != -1) return rp;
    }
    if (half_ops) return launch_gemm256_f16(p, d->layout_a, d->layout_b, batch, s);
    if (d->out_dtype == MUSE_BF16) return launch_gemm256<bf16_t>(p, d->layout_a, d->layout_b, batch, s);
    return launch_gemm256<float>(p, d->layout_a, d->layout_b, batch, s);
  }
  if (d->dtype == MUSE_BF16) {
    if (d->out_dtype == MUSE_BF16) return dispatch_layout<bf16_t, bf16_t>(p, d->layout_a, d->layout_b, batch, s);
    if (d->out_dtype == MUSE_F32) return dispatch_layout<bf16_t, float>(p, d->layout_a, d->layout_b, batch, s);
  } else if (d->dtype == MUSE_F32 && d->out_dtype == MUSE_F32) {
    return dispatch_layout<float, float>(p, d->layout_a, d->layout_b, batch, s);
  }
  return MUSE_ERR_BAD_ARG;
}

// ---- the bf16x3 product on four operand planes (gemm256.h: PipeX3) -----------------------------------------------------------------
extern "C" int muse_gemm_x3(const muse_gemm_desc* d, int64_t a_lo, int64_t b_lo, void* stream) {
  GemmParams p;
  const int rc = fill_params(d, p);
  if (rc) return rc;
  if (d->dtype != MUSE_BF16 || d->out_dtype != MUSE_F32 || d->batch > 1 || d->act) return MUSE_ERR_UNSUPPORTED;
  if (a_lo <= 0 || b_lo <= 0 || (a_lo & 7) || (b_lo & 7)) return MUSE_ERR_ALIGN;
  if (d->M < 128 || d->N < 128 || d->K < 64 || !gemm256_ok<float>(p, d->layout_a, d->layout_b)) return MUSE_ERR_UNSUPPORTED;
  {  // the lo planes are addressed through their own 32-bit buffer descriptors: same span as the hi planes (checked by fill_params)
    p.a_lo = a_lo; p.b_lo = b_lo;
  }
  return launch_gemm256_x3<float>(p, d->layout_a, d->layout_b, (hipStream_t)stream);
}

// ---- grouped weight gradients (gemm256.h: kernel_group) ---------------------------------------------------------------------------
// n <= 8 products C_i[M_i, N_i] = A_i^T B_i with both operands k-major bf16 (dW = dY^T X: k = the token dimension), f32 outputs, in
// ONE launch; split_k slices (the same count for every product) go to C_i + s * split_stride_i and are folded by
// muse_sum_slices_multi.  Every product must be one the 256^2 LDS-DMA kernel takes (muse_gemm_group_ok).
static int group_fill(const muse_gemm_desc* d, int n, int split_k, g256::GroupParams& gp) {
  if (!d || n < 1 || n > g256::MAXG || split_k < 1) return MUSE_ERR_BAD_ARG;
  int start = 0;
  for (int i = 0; i < n; ++i) {
    if ((d[i].dtype != MUSE_BF16 && d[i].dtype != MUSE_F16) || d[i].dtype != d[0].dtype || d[i].out_dtype != MUSE_F32 || d[i].layout_a != 1 ||
        d[i].layout_b != 1)
      return MUSE_ERR_UNSUPPORTED;
    if ((d[i].batch > 1) || d[i].bias || d[i].rowvec || d[i].residual || d[i].act) return MUSE_ERR_UNSUPPORTED;
    muse_gemm_desc di = d[i];
    di.split_k = split_k;
    if (split_k > 1 && di.split_stride == 0) return MUSE_ERR_BAD_ARG;
    const int rc = fill_params(&di, gp.p[i]);
    if (rc) return rc;
    if (di.M < 256 || di.N < 256 || di.K < 128 || !gemm256_ok<float>(gp.p[i], 1, 1)) return MUSE_ERR_UNSUPPORTED;
    if (split_k > 1 && d[i].accumulate) return MUSE_ERR_BAD_ARG;     // (accumulation happens in muse_sum_slices_multi)
    gp.tile_start[i] = start;
    start += ((di.M + 255) / 256) * ((di.N + 255) / 256);
  }
  for (int i = n; i <= g256::MAXG; ++i) gp.tile_start[i] = start;
  for (int i = n; i < g256::MAXG; ++i) gp.p[i] = gp.p[0];
  gp.n = n;
  return 0;
}
extern "C" int muse_gemm_group_ok(const muse_gemm_desc* d, int32_t n, int32_t split_k) {
  g256::GroupParams gp;
  return group_fill(d, n, split_k, gp);
}
extern "C" int muse_gemm_group(const muse_gemm_desc* d, int32_t n, int32_t split_k, void* stream) {
  g256::GroupParams gp;
  const int rc = group_fill(d, n, split_k, gp);
  if (rc) return rc;
  const bool half_ops = d[0].dtype == MUSE_F16;
  auto kern = half_ops ? g256::kernel_group<float, 1, 1, true, true> : g256::kernel_group<float, 1, 1, true>;
  static bool attr_set[2] = {false, false};
  if (!attr_set[half_ops]) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, g256::LDS_BYTES);
    attr_set[half_ops] = true;
  }
  hipLaunchKernelGGL(kern, dim3(gp.tile_start[n], split_k, 1), dim3(512), g256::LDS_BYTES, (hipStream_t)stream, gp);
  return (int)hipGetLastError();
}

#ifdef G256_TIMESTAMPS
extern "C" int muse_debug_gemm_ts(long* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g256::g_gemm_ts), &buf, sizeof(buf)); }
#endif
