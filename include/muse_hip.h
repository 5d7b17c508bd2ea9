/* libmuse_hip — C-ABI of the MI355X (gfx950) kernels behind the open-muse MaskGit hot path.
 *
 * The reference (huggingface/open-muse) has no first-party native code: its hot path reaches native kernels only
 * through torch / xformers / apex / flash_attn call sites.  Each entry point below names
 * the reference call site it replaces (paths relative to the reference repo root).  INTEGRATION.md shows the
 * ctypes binding a maintainer would add on the reference side.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch allocates; the library holds no memory);
 *   - every call only enqueues work on `stream` (a hipStream_t passed as void*) and never synchronises;
 *   - return 0 on success, a hipError_t (> 0) for launch errors, or a negative MUSE_ERR_* for argument errors;
 *   - bf16 tensors are raw uint16 storage; "f32" is IEEE binary32; indices are int64 like torch.long;
 *   - activations are row-major [tokens, features] (transformer) or NHWC (VQGAN).
 */
#ifndef MUSE_HIP_H
#define MUSE_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { MUSE_F32 = 0, MUSE_BF16 = 1, MUSE_F16 = 2 /* IEEE half: GEMM operands only (muse_gemm, muse_gemm_group) */ };
enum { MUSE_ERR_BAD_ARG = -1, MUSE_ERR_ALIGN = -2, MUSE_ERR_UNSUPPORTED = -3 };

int muse_version(void); /* ABI version of this header */

/* ------------------------------------------------------------------------------------------------------------
 * muse_gemm — C = epilogue(alpha * A * B^T), strided-batched, MFMA (bf16: v_mfma_f32_16x16x32_bf16,
 * f32: v_mfma_f32_16x16x4_f32 = exact fp32 fma chain).
 * layout_x = 0: X(r,k) at X + r*ld + k ("k-contiguous", the nn.Linear weight layout);
 * layout_x = 1: X(r,k) at X + k*ld + r ("k-major", the transposed view backward needs).
 * Replaces: every nn.Linear on the path (muse/modeling_transformer.py:198-200,218,789-798,979-984) and their
 * autograd backward, torch.baddbmm / torch.matmul of Attention.attention (:226-238), and the VQ addmm
 * (muse/modeling_maskgit_vqgan.py:310-315, via rowvec/bias: C = (zn[m] + en[n]) - 2 z.e).
 * Batch index z in [0,batch) addresses X + (z / zdiv) * sX0 + (z % zdiv) * sX1  (e.g. (image, head)).
 * Requirements: A, B 16-byte aligned; lda/ldb and batch strides multiples of 16 bytes; for a k-contiguous operand
 * with K % (16/elsize) != 0 the row must be physically padded with zeros up to the next 16-byte chunk.
 * dtype MUSE_F16 (round 6): IEEE-half operands on v_mfma_f32_16x16x32_f16 with f32 accumulation and f32 output - 11 significant bits
 * per operand = the TF32 operand format `enable_tf32` of configs/cc12m_uvit_clip.yaml:103 multiplies in (gfx950 has no xf32 MFMA), at
 * the bf16 matrix rate; the narrower exponent is the caller's business (muse_cast_f32_to_f16 scales by a power of two, alpha undoes
 * it).  Only the 256 x 256 LDS-DMA kernels carry it: M, N >= 128, K >= 64, no activation, batch / split_k as for bf16 operands with
 * f32 output; anything else MUSE_ERR_UNSUPPORTED (muse_gemm_tile says so beforehand).
 */
typedef struct muse_gemm_desc {
  const void* A;
  const void* B;
  void* C;
  const void* bias;     /* f32 [N] or NULL: added per output column                       */
  const void* rowvec;   /* f32 [M] or NULL: added per output row                          */
  const void* residual; /* out_dtype [M, ldr] or NULL: added after the activation         */
  int32_t dtype;        /* MUSE_F32 | MUSE_BF16 | MUSE_F16 (A and B)                       */
  int32_t out_dtype;    /* MUSE_F32 | MUSE_BF16 (bf16 only with bf16 inputs)               */
  int32_t layout_a, layout_b;
  int32_t M, N, K;
  int32_t batch, zdiv;
  int64_t lda, ldb, ldc, ldr;
  int64_t sA0, sA1, sB0, sB1, sC0, sC1;
  float alpha;
  int32_t accumulate; /* C += ...                                                          */
  int32_t act;        /* 0 none, 1 erf-GELU (F.gelu)                                       */
  int32_t split_k;    /* > 1: K is cut into that many slices (f32 output, no epilogue extras)                    */
  int64_t split_stride; /* != 0: slice s stores its partial result at C + s*split_stride elements (workspace, plain
                         stores; reduce with muse_sum_slices - deterministic); == 0: slices are added to C with hardware
                         f32 atomics, the caller pre-initialises C (zeros, or the value to accumulate into)          */
} muse_gemm_desc;
int muse_gemm(const muse_gemm_desc* d, void* stream);
/* Block-tile edge muse_gemm will use for this descriptor: 256 (LDS-DMA kernel, one block per CU) or 128 (two / three
 * blocks per CU); < 0 = the error muse_gemm would return.  Host code sizes split_k with it (ops.wgrad_splits). */
int muse_gemm_tile(const muse_gemm_desc* d);
/* The f32-class "bf16x3" product as ONE kernel on four bf16 operand planes: C = alpha (A_hi B_hi^T + A_hi B_lo^T + A_lo B_hi^T) with f32
 * accumulation, where x ~= hi + lo (muse_split_f32_to_bf16x2; 2^-16 relative per product: at or above the TF32 products
 * configs/cc12m_uvit_clip.yaml:102-103 trains with).  d describes the product as for muse_gemm with dtype MUSE_BF16, out_dtype MUSE_F32,
 * A / B = the hi planes; the lo planes sit a_lo / b_lo ELEMENTS behind them with the same leading dimensions (multiples of 8).  Every
 * layout, bias / rowvec / residual / accumulate, split_k through a workspace; batch 1, no activation, M, N >= 128, K >= 64 and the
 * 256-tile kernel's alignment rules - otherwise MUSE_ERR_UNSUPPORTED (the caller runs three muse_gemm products or one over
 * K-concatenated operands, muse_split_f32_to_bf16_cat3). */
int muse_gemm_x3(const muse_gemm_desc* d, int64_t a_lo, int64_t b_lo, void* stream);
/* Kernel form behind that tile: 128, 256 (launch-per-tile LDS-DMA kernel) or 257 (the persistent tile-walking form of the 256 kernel,
 * csrc/gemm256p.h: bf16 operands, one batch, no split-K, no bias / activation; k-contiguous A); < 0 = muse_gemm's error.  Pure host
 * logic (tests assert that the train step's products take the persistent form; MUSE_G256P=0 turns it off). */
int muse_gemm_path(const muse_gemm_desc* d);

/* GROUPED weight gradients: n <= 8 products C_i[M_i, N_i] = A_i^T B_i (layout_a = layout_b = 1: k-major bf16 operands, k = the token
 * dimension; f32 output; the four dW = dY^T X of a transformer layer, muse/modeling_transformer.py:770-778,973-977 under autograd)
 * in ONE launch of the 256^2 LDS-DMA kernel over the concatenated tile lists.  `split_k` (the same for every product) cuts K into
 * slices written to C_i + s * split_stride_i (reduce with muse_sum_multi: deterministic); split_k = 1 writes (or, accumulate = 1,
 * adds to) C_i directly.  muse_gemm_group_ok: 0 if muse_gemm_group would take the list, else the error it would return (products
 * the 256^2 kernel does not take: MUSE_ERR_UNSUPPORTED - the caller then uses muse_gemm per product). */
int muse_gemm_group_ok(const muse_gemm_desc* d, int32_t n, int32_t split_k);
int muse_gemm_group(const muse_gemm_desc* d, int32_t n, int32_t split_k, void* stream);

/* 2-D transpose out[c, r] = in[r, c] (strided-batched); used only by the fallback that feeds k-major operands to
 * the k-contiguous GEMM path (MUSE_GEMM_TR=0). */
int muse_transpose(const void* in, void* out, int32_t dtype, int32_t rows, int32_t cols, int64_t ld_in, int64_t ld_out,
                   int32_t batch, int64_t stride_in, int64_t stride_out, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Transformer element/row kernels.
 * muse_layernorm_fwd: y = LayerNorm(x) * w (+ residual), weight-only LN (muse/modeling_transformer.py:124-137 at
 *   :878,:883,:786,:796,:1269,:983) with the residual add of :884 / :903 fused.  mean/rstd [rows] f32 are saved.
 * muse_layernorm_bwd: dx = LN'(dy) (+ dres);  dw_partial [nblk, cols] f32 holds per-block column sums of dy*xhat,
 *   reduced into dw by muse_colsum (deterministic two-stage reduction).  dx_bf16 (may be NULL): a second, bf16 copy of dx
 *   for the GEMMs that consume it next (the f32 dx stays the residual-stream gradient).
 */
int muse_layernorm_fwd(const void* x, int32_t x_dtype, const float* w, const float* residual, void* y, int32_t y_dtype,
                       float* mean, float* rstd, int32_t rows, int32_t cols, float eps, void* stream);
int muse_layernorm_bwd(const void* dy, int32_t dy_dtype, const void* x, int32_t x_dtype, const float* w,
                       const float* mean, const float* rstd, const float* dres, void* dx, int32_t dx_dtype,
                       void* dx_bf16, float* dw_partial, int32_t nblk, int32_t rows, int32_t cols, void* stream);
int muse_layernorm_bwd_nblk(int32_t rows);
/* The two back-to-back LayerNorms of a NormFormer layer (muse/modeling_transformer.py:882-884 then :785-789) in one pass:
 *   fwd: x1 = x + LN(ao) * w_post (f32) ; ln2 = LN(x1) * w_pre (bf16); statistics of both rows.   ao bf16 [rows, cols], cols <= 1024.
 *   bwd: dx1 = LN_pre'(dln2) + dres (f32) ; dao = LN_post'(dx1) (bf16); dw partials [nblk, cols] for both weights
 *        (nblk = muse_layernorm_bwd_nblk(rows), reduced by muse_colsum).
 * Bit-identical to two muse_layernorm_fwd / _bwd calls; saves the write + re-read of x1 / dx1. */
int muse_layernorm_pair_fwd(const void* ao, const float* x, const float* w_post, const float* w_pre, float* x1, void* ln2,
                            float* mean_post, float* rstd_post, float* mean_pre, float* rstd_pre, int32_t rows, int32_t cols,
                            float eps, void* stream);
int muse_layernorm_pair_bwd(const void* dln2, const float* x1, const float* w_pre, const float* mean_pre, const float* rstd_pre,
                            const float* dres, const void* ao, const float* w_post, const float* mean_post, const float* rstd_post,
                            float* dx1, void* dao, float* dw_partial_pre, float* dw_partial_post, int32_t nblk, int32_t rows,
                            int32_t cols, void* stream);
/* out[c] (+)= sum_r in[r, c], f32 */
int muse_colsum(const float* in, float* out, int32_t rows, int32_t cols, int32_t accumulate, void* stream);
/* Biases of `use_bias=True` models (muse/modeling_transformer.py:130 LayerNorm bias, :170-176 / :770-778 / :973-977 / :1155
 * nn.Linear biases; the Linear bias itself rides in muse_gemm's `bias` epilogue):
 *   muse_add_rowvec: x[r, c] += b[c], f32 in place (the LayerNorm bias behind muse_norm_res_fwd / muse_layernorm_fwd).
 *   muse_bias_grad_partial: partial[k, c] = sum of dy[r, c] over the k-th chunk of muse_bias_grad_rows_per_block() rows (dy f32 or
 *   bf16 with row stride ld); muse_colsum over the ceil(rows / R) partial rows gives d(bias), in a fixed order. */
int muse_add_rowvec(float* x, const float* b, int64_t rows, int32_t cols, void* stream);
int muse_bias_grad_rows_per_block(void);
int muse_bias_grad_partial(const void* dy, int32_t dtype, float* partial, int64_t rows, int32_t cols, int64_t ld, void* stream);

/* softmax over the last dim of [rows, ld] (first `cols` valid, pad columns written as 0), in place capable.
 * Replaces F.softmax at muse/modeling_transformer.py:236.  bwd: ds = p * (dp - sum(p*dp)). */
int muse_softmax_fwd(const void* x, void* y, int32_t dtype, int64_t rows, int32_t cols, int64_t ld, void* stream);
int muse_softmax_bwd(const void* p, const void* dp, void* ds, int32_t dtype, int64_t rows, int32_t cols, int64_t ld,
                     void* stream);

/* Fused full-visibility attention (bf16 in/out, f32 softmax / accumulation), head_dim in {16,32,48,64}, any query and
 * key length: replaces Attention.attention (muse/modeling_transformer.py:221-241: baddbmm -> softmax -> matmul), the
 * xformers memory_efficient_attention seam (:206-210; muse/modeling_transformer_v2.py:881-889, self- and cross-attention)
 * and their autograd backward, without materialising the S x S matrix.  K / V stream through LDS in tiles of <= 288 keys
 * (online softmax across tiles), so seq 257 / 256 / 1024 and the 77-token text cross-attention run on the same kernels.
 * Token t of image b, head h of tensor X lives at X + b*bsX + t*ldX + h*head_dim (elements); all strides multiples of 8.
 * lse / dsum are f32 [batch*heads, muse_attention_seq_pad(seq_q)].  alpha = 1/sqrt(head_dim) (the baddbmm alpha, :168,:230). */
typedef struct muse_attn_desc {
  const void* q;
  const void* k;
  const void* v;
  void* o;                       /* forward output (ctx); read by the backward */
  int64_t ldq, ldk, ldv, ldo;    /* elements between consecutive tokens        */
  int64_t bsq, bsk, bsv, bso;    /* elements between consecutive images        */
  int32_t batch, heads, head_dim, seq_q, seq_kv;
  float alpha;
} muse_attn_desc;
int muse_attention_seq_pad(int32_t seq);
int muse_attention_fwd_ex(const muse_attn_desc* d, float* lse, void* stream);
/* backward: d_o = upstream gradient of o; writes dq / dk / dv (bf16, own strides) and the scratch row constants dsum */
int muse_attention_bwd_ex(const muse_attn_desc* d, const void* d_o, int64_t lddo, int64_t bsdo, const float* lse, float* dsum,
                          void* dq, int64_t lddq, int64_t bsdq, void* dk, int64_t lddk, int64_t bsdk, void* dv, int64_t lddv,
                          int64_t bsdv, void* stream);
/* The same seam for the "bf16x3" compute mode (f32 tensors, TF32-class products: configs/cc12m_uvit_clip.yaml:102-103 trains with
 * mixed_precision "no" + enable_tf32): q / k / v / o / d_o / dq / dk / dv are FLOAT tensors (strides in elements, multiples of 4),
 * every product of the forward and the backward is three bf16 MFMA products of (hi, lo) operand planes the kernel makes itself
 * (<= 2^-16 relative per product), softmax and accumulation in f32.  head_dim 64, seq_q = 256, seq_kv in 225..256 or 65..96
 * (self-attention of config 4's 16 x 16 grid; its 77 text states); anything else: MUSE_ERR_UNSUPPORTED (the caller keeps the
 * materialised route: muse_gemm batched + muse_softmax_*).  lse is f32 [batch*heads, 256]. */
int muse_attention_x3_fwd(const muse_attn_desc* d, float* lse, void* o_planes, int64_t o_lo, void* stream);
int muse_attention_x3_bwd(const muse_attn_desc* d, const void* d_o, int64_t lddo, int64_t bsdo, const float* lse, void* dq,
                          int64_t lddq, int64_t bsdq, void* dk, int64_t lddk, int64_t bsdk, void* dv, int64_t lddv, int64_t bsdv,
                          void* dq_planes, int64_t dq_lo, void* dk_planes, int64_t dk_lo, void* dv_planes, int64_t dv_lo, void* stream);
/* (dq / dk / dv may be NULL when their planes are given: a gradient that only weight GEMMs read exists as planes alone.)
 * (*_planes, optional: the result ALSO as the bf16 operand planes of the products that read it - muse_gemm_x3 - so that no split pass
 *  runs over it: the hi plane is addressed exactly like the f32 tensor (same strides, in elements), the lo plane sits *_lo elements
 *  behind it; NULL = f32 only) */
/* Streaming forms for sequences of whole 256-row blocks on BOTH sides (round 6: the self-attention of BASELINE config 4's 1024 tokens;
 * reference modeling_transformer_v2.py:881-915).  d describes the FULL sequences (seq_q, seq_kv multiples of 256, head_dim 64; batch
 * strides of the whole tensors).  _fwd_stream: a workgroup keeps 256 queries and streams the key blocks through LDS with an online
 * softmax - context and lse [seq_q/256][batch*heads][256] written once (o_planes as for muse_attention_x3_fwd).  _bwd_stream: dQ per
 * query block over the streamed key blocks (it also writes dO.O per query into dsum, a workspace shaped like lse), then dK / dV per
 * key block over the streamed query blocks; every gradient written once (f32 and / or operand images, as for muse_attention_x3_bwd). */
int muse_attention_x3_fwd_stream(const muse_attn_desc* d, float* lse, void* o_planes, int64_t o_lo, void* stream);
int muse_attention_x3_bwd_stream(const muse_attn_desc* d, const float* d_o, int64_t lddo, int64_t bsdo, const float* lse, float* dsum,
                                 float* dq, int64_t lddq, int64_t bsdq, float* dk, int64_t lddk, int64_t bsdk, float* dv, int64_t lddv,
                                 int64_t bsdv, void* dq_planes, int64_t dq_lo, void* dk_planes, int64_t dk_lo, void* dv_planes, int64_t dv_lo,
                                 void* stream);
/* Block-by-block form of the longer sequences (round 6; reference modeling_transformer_v2.py:757-792 at the 1024 tokens of BASELINE config 4):
 * muse_attention_x3_fwd / _bwd run per (256 query rows, <= 256 keys) block pair, these two put the pieces together.
 * _merge: part[j] [batch*seq, heads*64] f32 (j < nk <= 8, part_stride elements apart) = key block j's softmax times its values, lp[j]
 *   [seq/256][batch*heads][256] (lp_stride apart) its log-sum-exp: lse = log sum_j exp(lp_j), out = sum_j exp(lp_j - lse) part_j; writes out,
 *   lse [seq/256][batch*heads][256] and optionally out's (hi, lo) bf16 operand planes (out_planes, lo plane out_lo elements behind).
 * muse_sum_parts_strided: out[r, 0..cols) (row pitch ldo, += when accumulate) = sum over j < n of parts[j*part_stride + r*cols + c]. */
int muse_attention_x3_merge(const float* part, int64_t part_stride, const float* lp, int64_t lp_stride, int32_t nk, float* out, float* lse,
                            void* out_planes, int64_t out_lo, int32_t batch, int32_t seq, int32_t heads, void* stream);
int muse_sum_parts_strided(const float* parts, int64_t part_stride, int32_t n, int64_t rows, int32_t cols, float* out, int64_t ldo,
                           int32_t accumulate, void* stream);
/* packed self-attention: qkv [B*S, 3*H] (q | k | v, H = heads*head_dim: the fused QKV projection), ctx [B*S, H], dqkv [B*S, 3*H] */
int muse_attention_fwd(const void* qkv, void* ctx, float* lse, int32_t batch, int32_t seq, int32_t heads,
                       int32_t head_dim, float alpha, void* stream);
int muse_attention_bwd(const void* qkv, const void* ctx, const void* dctx, const float* lse, float* dsum, void* dqkv,
                       int32_t batch, int32_t seq, int32_t heads, int32_t head_dim, float alpha, void* stream);

/* GLU: h = gelu_erf(a) * b with ab = [rows, 2*inter] (a = first half).  muse/modeling_transformer.py:789-792. */
int muse_glu_fwd(const void* ab, void* h, int32_t dtype, int64_t rows, int32_t inter, void* stream);
int muse_glu_bwd(const void* ab, const void* dh, void* dab, int32_t dtype, int64_t rows, int32_t inter, void* stream);
/* The f32 GLU of the "bf16x3" compute mode: the same f32 results, written ALSO as the (hi, lo) bf16 operand planes muse_gemm_x3 reads
 * (planes = [2][rows][cols] bf16, hi plane first; the bits muse_split_f32_to_bf16x2 would produce from the f32 result).  h / dab may be
 * NULL: planes only (a result that nothing but weight GEMMs reads - the GLU output and its input gradient inside an MLP). */
int muse_glu_fwd_x3(const float* ab, float* h, void* planes, int64_t rows, int32_t inter, void* stream);
int muse_glu_bwd_x3(const float* ab, const float* dh, float* dab, void* planes, int64_t rows, int32_t inter, void* stream);
/* Fused middle of the NormFormer GLU MLP (muse/modeling_transformer.py:789-797), one pass over ab = [rows, 2*inter]:
 *   fwd: h = gelu_erf(a) * b, hm = LayerNorm(h) * w (mean/rstd saved);
 *   bwd: dh = LN'(dhm) never leaves the CU, dab = (dh*b*gelu'(a), dh*gelu(a)); dw_partial [ceil(rows/R), inter] f32 with
 *        R = muse_ffn_mid_rows_per_block(), reduced by muse_colsum.
 * `h` may be NULL in both: the forward then does not write it and the backward recomputes gelu_erf(a) * b (rounded to the storage
 * type, i.e. exactly the tensor the forward normalised) from ab. */
int muse_ffn_mid_rows_per_block(void);
int muse_ffn_mid_fwd(const void* ab, const float* w, void* h, void* hm, float* mean, float* rstd, int32_t dtype,
                     int32_t rows, int32_t inter, float eps, void* stream);
int muse_ffn_mid_bwd(const void* dhm, const void* h, const void* ab, const float* w, const float* mean, const float* rstd,
                     void* dab, float* dw_partial, int32_t dtype, int32_t rows, int32_t inter, void* stream);
/* y = gelu_erf(x); dx = dy * gelu'(x).  muse/modeling_transformer.py:981. */
int muse_gelu_fwd(const void* x, void* y, int32_t dtype, int64_t n, void* stream);
int muse_gelu_bwd(const void* x, const void* dy, void* dx, int32_t dtype, int64_t n, void* stream);
/* ... from f32 x / dy with the result written as bf16 (the dY operand of the next weight GEMMs in the bf16 compute mode) */
int muse_gelu_bwd_f32_bf16(const float* x, const float* dy, void* dx_bf16, int64_t n, void* stream);

/* Embed.forward (muse/modeling_transformer.py:942-957): out[b,s,:] = word[ids[b,s],:] + pos[s,:], f32 out.
 * bwd: deterministic (sorted by id on device, no float atomics): dword[v,:] (+)= sum over positions with id v,
 * dpos[s,:] (+)= sum_b dout[b,s,:].  `scratch` needs muse_embed_bwd_scratch_floats(hidden, vocab) f32. */
int muse_embed_fwd(const int64_t* ids, const float* word, const float* pos, float* out, int32_t batch, int32_t seq,
                   int32_t hidden, int32_t vocab, void* stream);
int muse_embed_bwd(const int64_t* ids, const float* dout, float* dword, float* dpos, float* scratch, int32_t batch,
                   int32_t seq, int32_t hidden, int32_t vocab, int32_t accumulate, void* stream);
int64_t muse_embed_bwd_scratch_floats(int32_t hidden, int32_t vocab);
/* The same gradients by a stable (token id, position) sort and segmented row sums (csrc/embed.hip): every gradient row is read
 * once, rows with many hits (the mask token) are shared by several blocks, the summation order is fixed.  `scratch`: 256-byte
 * aligned, muse_embed_bwd2_scratch_bytes(batch, seq, hidden, vocab) bytes.  Ids outside [0, vocab) are ignored. */
int muse_embed_bwd2(const int64_t* ids, const float* dout, float* dword, float* dpos, void* scratch, int64_t scratch_bytes,
                    int32_t batch, int32_t seq, int32_t hidden, int32_t vocab, int32_t accumulate, void* stream);
int64_t muse_embed_bwd2_scratch_bytes(int32_t batch, int32_t seq, int32_t hidden, int32_t vocab);

/* F.cross_entropy(logits, labels, ignore_index=-100, label_smoothing) (muse/modeling_transformer.py:1276-1279).
 * fwd: row_loss[r], lse[r] per row, then loss = sum(row_loss over valid) / n_valid into loss_out[0], n_valid into
 * loss_out[1] (f32).  bwd: dlogits = (softmax - target) * gscale / n_valid, 0 for ignored rows. */
int muse_cross_entropy_fwd(const void* logits, int32_t dtype, const int64_t* labels, float* row_loss, float* lse,
                           float* loss_out, int64_t rows, int32_t vocab, int64_t ld, float label_smoothing, void* stream);
int muse_cross_entropy_bwd(const void* logits, int32_t dtype, const int64_t* labels, const float* lse,
                           const float* loss_out, const float* grad_out, void* dlogits, int32_t dl_dtype, int64_t rows,
                           int32_t vocab, int64_t ld, float label_smoothing, void* stream);

/* AdamW over one flat f32 buffer (torch.optim.AdamW / apex FusedAdam(adam_w_mode) semantics,
 * training/train_maskgit_imagenet.py:242-261,438); optionally refreshes the bf16 compute copy of the weights. */
int muse_adamw_flat(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr, float beta1,
                    float beta2, float eps, float weight_decay, int32_t step, float grad_scale, void* stream);
/* Exponential moving average of the weights, every tracked tensor in ONE launch (EMAModel.step, reference muse/modeling_ema.py:118-137,
 * called behind the optimizer step at training/train_muse.py:779-780): shadow = shadow - one_minus_decay * (shadow - param), the three
 * f32 roundings of the reference's tensor expression kept.  `table` (device): 4 x int64 per tensor {shadow, param, n, mode}, mode 0 =
 * update, 1 = copy (requires_grad == False, :134-135); `chunk_first` as for muse_adamw_multi. */
int muse_ema_multi(const int64_t* table, const int32_t* chunk_first, int32_t num_tensors, int32_t num_chunks, float one_minus_decay,
                   void* stream);
/* The same update for many separate f32 tensors in ONE launch (models whose parameters are ordinary tensors: MaskGiTUViT_v2,
 * muse/modeling_transformer_v2.py; the reference reaches this through apex FusedAdam's multi_tensor_apply).  `table` (device):
 * 6 x int64 per tensor {p, g, m, v, p_bf16 or 0, n}; `chunk_first` (device, num_tensors + 1 x int32): exclusive prefix sum of
 * ceil(n / 4096); num_chunks = chunk_first[num_tensors]. */
int muse_adamw_multi(const int64_t* table, const int32_t* chunk_first, int32_t num_tensors, int32_t num_chunks, float lr,
                     float beta1, float beta2, float eps, float weight_decay, int32_t step, float grad_scale, void* stream);
/* AdamW with PARAMETER GROUPS (training/train_muse.py:425-445 builds two: weight decay on the matrices, none on bias / LayerNorm /
 * embedding weights; torch.optim semantics - each group its own lr / betas / eps / weight_decay).  `group_hyper` (HOST memory, read
 * during the call): ngroups <= 8 rows of {lr, beta1, beta2, eps, weight_decay}.
 * _flat_groups: the flat buffer is cut into segments, seg_end[s] (device int64, ascending ABSOLUTE element offsets of the flat
 * buffer) closes segment s and seg_group[s] (device int32) names its group; the call updates elements [base, base + n) - p, g, m, v,
 * p_bf16 point at element `base` - so range-wise updates (inside backward, behind an all-reduce bucket) share one table.
 * _multi_groups: muse_adamw_multi's table with a seventh column, the tensor's group in its low 8 bits; above them (optional, round 6) the
 *   element distance from p_bf16 to a second bf16 plane: p_bf16 then receives hi = bf16(p) and that plane lo = bf16(p - hi), the
 *   bf16x3 operand planes of the updated weight (muse_gemm_x3 reads them next step; 0 = plain bf16 copy).
 * A one-group call is bit-identical to muse_adamw_flat / muse_adamw_multi. */
int muse_adamw_flat_groups(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, int64_t base,
                           const int64_t* seg_end, const int32_t* seg_group, int32_t nseg, const float* group_hyper,
                           int32_t ngroups, int32_t step, float grad_scale, void* stream);
int muse_adamw_multi_groups(const int64_t* table, const int32_t* chunk_first, int32_t num_tensors, int32_t num_chunks,
                            const float* group_hyper, int32_t ngroups, int32_t step, float grad_scale, void* stream);
/* out[i] (+)= sum over s < nslices of ws[s*stride + i]: reduction of split-K partial results (n, stride % 4 == 0) */
int muse_sum_slices(const float* ws, float* out, int32_t nslices, int64_t n, int64_t stride, int32_t accumulate, void* stream);
/* njobs <= 16 reductions in ONE launch, each bit-identical to the single-job kernel it stands for: kind 0 = muse_sum_slices
 * (ws[i] -> out[i], nslices[i] slices of n[i] elements stride[i] apart), kind 1 = muse_colsum (out[i][c] (+)= sum over the
 * nslices[i] rows of the [nslices[i], n[i]] f32 matrix ws[i]).  All arrays are HOST arrays read during the call. */
int muse_sum_multi(const void* const* ws, void* const* out, const int32_t* nslices, const int64_t* n, const int64_t* stride,
                   const int32_t* accumulate, const int32_t* kind, int32_t njobs, void* stream);
/* hi = bf16(in), lo = bf16(in - hi): the operand planes of a bf16x3 product (in ~= hi + lo to 2^-16 relative) */
int muse_split_f32_to_bf16x2(const float* in, void* hi, void* lo, int64_t n, void* stream);
/* out[r, c] = sum over nslices of ws[s * stride + r * cols + c] (+ bias[c]) (+ residual[r, c]) written as out_dtype (f32 / bf16; the
 * residual has the output's dtype, like muse_gemm's): reduction of a forward product's K-slice workspace fused with the Linear's
 * epilogue - the small-batch decoding path (ops.gemm: products of <= 2048 rows whose tiles would fill a fraction of the chip). */
int muse_sum_slices_epilogue(const float* ws, int32_t nslices, int64_t stride, const float* bias, const void* residual, int64_t ldr,
                             void* out, int32_t out_dtype, int64_t ldc, int64_t rows, int32_t cols, void* stream);
/* The same split written as ONE GEMM operand of three times the K length: the bf16x3 product hi*hi + hi*lo + lo*hi is a single product
 * of A' = (hi | hi | lo) and B' = (hi | lo | hi) along K.  in [rows, cols] f32 (row stride ld_in); mode 0: planes concatenated along the
 * columns, out [rows, 3 cols] (k-contiguous operand), mode 1: stacked along the rows, out [3 rows, cols] (k-major operand), both with row
 * stride ld_out; lo_pos = 1 | 2: the third that carries the lo plane.  cols % 4 == 0. */
int muse_split_f32_to_bf16_cat3(const float* in, void* out, int64_t rows, int32_t cols, int64_t ld_in, int64_t ld_out, int32_t mode,
                                int32_t lo_pos, void* stream);
int muse_cast_f32_to_bf16(const float* in, void* out, int64_t n, void* stream);
/* out = half(in * scale): the operand image of a MUSE_F16 product (round to nearest even, subnormals kept; a finite value beyond half's
 * range becomes inf - the product turns NaN rather than silently wrong).  stats: NULL or int32[2], incremented by [0] the finite
 * elements that overflowed, [1] the non-zero elements rounded to zero. */
int muse_cast_f32_to_f16(const float* in, void* out, int64_t n, float scale, int32_t* stats, void* stream);
/* What the producer entry points below that write operand images next to (or instead of) their f32 result - muse_glu_fwd_x3 / _bwd_x3,
 * muse_norm_adaln_fwd_x3 / _bwd_x3, muse_attention_x3_fwd / _bwd / _merge - write as that image.  half = 0 (default): the (hi, lo) bf16
 * planes of the "bf16x3" mode, as documented with each.  half = 1 ("f16" mode): ONE IEEE-half image [rows][cols] at the plane pointer =
 * half(result * s), the bits muse_cast_f32_to_f16 makes of the f32 result, with s = 1 for forward results and s = grad_scale (a power
 * of two) for the gradients the backward entry points produce; lo-plane distances are ignored; muse_attention_x3_fwd / _bwd also COMPUTE in
 * that format (one half plane per operand, one half MFMA per K step; dO and dS times grad_scale, results divided by it); stats (device int32[2] or NULL): [0] is
 * incremented per 4-element group that holds an inf / NaN half (muse_cast_f32_to_f16's overflow counter: a dynamic gradient scale backs
 * off on it).  Process state (host code sets it around a pass: muse/ops.py f32_gemms_as_f16), not thread safe. */
int muse_operand_images(int32_t half, float grad_scale, int32_t* stats);
/* Overflow guard of the "f16" mode for the multi-tensor optimizer kernels (muse_adamw_multi, muse_adamw_multi_groups): while flag is
 * non-NULL those kernels read *flag (device int32: the overflow counter muse_operand_images / muse_cast_f32_to_f16 increment) and leave
 * parameters and moments untouched when it is non-zero - GradScaler's found_inf without a host round trip.  Process state; NULL = off. */
int muse_adamw_skip_flag(const int32_t* flag);
int muse_cast_bf16_to_f32(const void* in, float* out, int64_t n, void* stream);

/* prepare_inputs_and_labels (training/train_maskgit_imagenet.py:371-394) given the two uniform draws.
 * tokens [B,S] i64, class_ids [B] i64, timesteps [B] f32, noise [B,S] f32 ->
 * input_ids, labels [B,S+1] i64, mask_prob [B] f32.  S <= 1024. */
int muse_mask_sample(const int64_t* tokens, const int64_t* class_ids, const float* timesteps, const float* noise,
                     int64_t* input_ids, int64_t* labels, float* mask_prob, int32_t batch, int32_t seq, int64_t mask_id,
                     int64_t codebook_size, float min_masking_rate, void* stream);

/* One iteration of MaskGit parallel decoding on the device (muse/modeling_transformer.py:1409-1454, :generate2;
 * muse/modeling_transformer_v2.py:434-474; muse/sampling.py:30-35 mask_by_random_topk): replaces ~10 ATen launches per step
 * (softmax, multinomial, where, gather, log, sort, gather, compare, where).
 *   the logits of image b, position s start at logits + b*img_stride + s*ld (f32; lets the caller skip a class-token row);
 *   logits = uncond + guidance_scale * (cond - uncond) over the first `vocab` columns (uncond_logits NULL: logits = cond);
 *   p = softmax(logits);  sampled = argmax_j p_j / q_j, q ~ Exp(1) (how torch.multinomial(num_samples = 1) draws), known
 *   tokens (input_ids != mask_id) are kept;  confidence = log(clamp(p_sampled, 1e-20)) + temperature * gumbel(u) (FLT_MAX as p
 *   for known tokens);  mask_len = max(1, min(#unknown - 1, sched_mask_len));  next_ids = mask_id where confidence < the
 *   mask_len-th smallest confidence of the image (0-based), else the sampled id.
 * noise_exp [batch*seq, vocab] / noise_u [batch, seq]: the caller's random draws (parity tests replay the reference's CPU
 * generator); NULL: Philox4x32-10 keyed by (seed, step).  raw_sampled (may be NULL) receives the samples before known tokens
 * are restored (the reference's `intermediate` list).  conf_scratch: f32 [batch*seq].  2 <= seq <= 4096. */
int muse_sample_step(const float* cond_logits, const float* uncond_logits, float guidance_scale, int64_t img_stride, int64_t ld,
                     int32_t vocab,
                     const int64_t* input_ids, int64_t mask_id, const float* noise_exp, const float* noise_u, uint64_t seed,
                     uint32_t step, float temperature, int32_t sched_mask_len, int32_t batch, int32_t seq, int64_t* raw_sampled,
                     int64_t* sampled, int64_t* next_ids, float* conf_scratch, void* stream);

/* training/train_muse.py:149-226 mask_or_random_replace_tokens on the device.  mask_prob = cos(pi/2 t) clipped to
 * min_masking_rate (timesteps given) or mask_prob_in as is (the eval_mask_ratios branch); k = round(seq * mask_prob) >= 1;
 * mask = argsort(noise) < k (noise given) or the rectangle rects[b] = (y0, x0, h, w) on the sqrt(seq) grid (contiguous-region
 * branch, drawn on the host like the reference does); input_ids = mask_id where masked (the reference's `noise_type` test at
 * :202 is always true, so "random_replace" also masks: kept); labels = tokens where masked else -100, or all tokens when
 * all_labels (predict_all_tokens / random_replace), with loss_weight = 1 - (1 - mask) * (1 - mask_prob) * (1 - weight_min)
 * (:145-146; loss_weight may be NULL). */
int muse_mask_tokens(const int64_t* tokens, const float* timesteps, const float* mask_prob_in, const float* noise,
                     const int32_t* rects, int64_t* input_ids, int64_t* labels, float* loss_weight, float* mask_prob,
                     int32_t batch, int32_t seq, int64_t mask_id, float min_masking_rate, int32_t all_labels, float weight_min,
                     void* stream);

/* nn.Dropout (muse/modeling_transformer.py:237 attention probabilities, :797 feed-forward, :956 embeddings): y = x * keep / (1 - p),
 * keep_i = [Philox(seed, offset + i / 4) lane (i % 4) >= p]; in place capable.  There is no mask tensor: the backward pass applies
 * the same call (same seed / offset) to the gradient. */
int muse_dropout(const void* x, void* y, int32_t dtype, int64_t n, float p, uint64_t seed, uint64_t offset, void* stream);

/* training/train_muse.py:715-731 conditioning dropout: keep_b = uniforms[b] < prob;  out = (x * keep != 0) ? x : empty
 * (x [batch, per_image] f32, empty [per_image] f32) - the reference's expression verbatim. */
int muse_cond_dropout(const float* x, const float* empty, const float* uniforms, float* out, int32_t batch, int64_t per_image,
                      float prob, void* stream);


/* ------------------------------------------------------------------------------------------------------------
 * MaskGitVQGAN kernels (NHWC activations).
 * muse_conv2d_nhwc: stride-1 SAME convolution (Conv2dSame, muse/modeling_maskgit_vqgan.py:33-45) as implicit GEMM
 *   on MFMA; weight pre-permuted to [Cout][KS][KS][Cin]; optional bias (f32 [Cout]), optional residual add
 *   (ResnetBlock :85), optional nearest x2 upsample of the input folded into the gather (UpsamplingBlock :146-147).
 *   H, W are the OUTPUT spatial dims.  Cin must be a multiple of 16/elsize.
 *   upsample: 0 = stride 1;  1 = input is [H/2, W/2], nearest x2 upsampled on the fly;  2 = input is [2H, 2W], stride-2
 *   3x3 convolution over it zero-padded by one row / column at the bottom / right (taming-VQGAN Downsample,
 *   muse/modeling_taming_vqgan.py:53-59: F.pad(0,1,0,1) + Conv2d(3, stride 2, padding 0)).  Same meaning in
 *   muse_conv2d_nhwc_split.
 */
int muse_conv2d_nhwc(const void* in, const void* weight, const float* bias, const void* residual, void* out,
                     int32_t dtype, int32_t batch, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t KS,
                     int32_t upsample, void* stream);
/* f32-class convolution on the bf16 matrix cores by hi/lo operand splitting (3 bf16 MFMAs per product, f32 accumulate,
 * error <= 2^-16 |a||b| per product: tighter than the TF32 the reference's cuDNN path uses by PyTorch default).
 * in / out / residual f32 NHWC, w_hi / w_lo bf16 [Cout][KS][KS][Cin] with w ~= w_hi + w_lo; Cin % 8 == 0. */
int muse_conv2d_nhwc_split(const float* in, const void* w_hi, const void* w_lo, const float* bias, const float* residual,
                           float* out, double* gn_partial, int32_t gn_groups, int32_t batch, int32_t H, int32_t W,
                           int32_t Cin, int32_t Cout, int32_t KS, int32_t upsample, void* stream);
/*   gn_partial != NULL (conv_in :168, the 1x1 nin_shortcut :82-85): GroupNorm(gn_groups) sum / sum-of-squares of the output
 *   per (image, 128-pixel tile, group), [B, H*W/128, gn_groups, 2] f64, as for muse_conv2d_nhwc_split2 below.  Needs
 *   H*W % 128 == 0 and Cout/gn_groups a power of two in [4, 32]. */
/* The same convolution for the 3x3 layers whose input comes out of GroupNorm+SiLU (conv1 / conv2 of every ResnetBlock
 * :73-80, conv_out :189): the activation arrives pre-split as two bf16 NHWC planes (muse_groupnorm_silu_nhwc_split) and all
 * operands go global -> LDS by DMA (csrc/conv_dma.hip).  Results are bit-identical to muse_conv2d_nhwc_split on the f32
 * tensor hi + lo came from.  KS == 3, Cin % 32 == 0, Cout % 4 == 0, each plane < 4 GiB. */
int muse_conv2d_nhwc_split2(const void* in_hi, const void* in_lo, const void* w_hi, const void* w_lo, const float* bias,
                            const float* residual, float* out, double* gn_partial, int32_t gn_groups, int32_t batch,
                            int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t KS, void* stream);
/*   gn_partial != NULL: the epilogue also writes the GroupNorm(gn_groups) sum / sum-of-squares of its output (bias and
 *   residual included) per (image, 256-pixel tile, group) as [B, H*W/256, gn_groups, 2] f64 - the statistics pass of the
 *   GroupNorm that consumes this tensor (muse_groupnorm_silu_nhwc_split with stats_nchunk = H*W/256) then never reads it.
 *   Needs H*W % 256 == 0 and Cout/gn_groups a power of two in [4, 128]. */
/* GroupNorm(32, eps, affine) + SiLU (muse/modeling_maskgit_vqgan.py:61,73-78,186-187,236-237).
 * stats: partial [B, nchunk, G, 2] f64 -> apply.  `partial` needs B*nchunk*G*2 doubles (nchunk from _nchunk). */
int muse_groupnorm_silu_nhwc(const void* x, void* y, int32_t dtype, const float* gamma, const float* beta,
                             double* partial, int32_t batch, int32_t HW, int32_t C, int32_t groups, float eps,
                             int32_t apply_silu, void* stream);
int muse_groupnorm_nchunk(int32_t HW);
/* GroupNorm(groups) + SiLU of the INPUT fused into the 3x3 bf16x3 convolution (muse/modeling_maskgit_vqgan.py:73-80 norm1 -> swish
 * -> conv1, norm2 -> swish -> conv2; :186-189 norm_out -> swish -> conv_out): `x` is the f32 NHWC activation, gn_scale / gn_shift
 * [batch][Cin] f32 the affine form of its normalisation (muse_groupnorm_scale_shift: rstd * gamma, beta - rstd * gamma * mean from
 * the [B, nchunk, groups, 2] f64 partial sums a producer's epilogue left).  The kernel normalises, activates and splits the
 * activation into the hi / lo bf16 operands while it stages them in LDS: bit-identical to muse_groupnorm_silu_nhwc_split followed
 * by muse_conv2d_nhwc_split2, without the write and re-read of the two planes.  Shapes: muse_conv2d_nhwc_gn_split2_ok (3x3, H and W
 * multiples of 16, Cin a multiple of 64); others return MUSE_ERR_UNSUPPORTED.  bias / residual / gn_partial as for _split2. */
int muse_conv2d_nhwc_gn_split2_ok(int32_t batch, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t KS);
/* muse_conv2d_nhwc_gn_split2's persistent form (one workgroup per CU walks the tiles; bit-identical; 6-8 % faster when the convolution has
 * the chip to itself, slower for a step that shares the chip with it): mode 1 on (default unless MUSE_CONV_PERSIST=0), 0 off, -1 query;
 * returns the mode in force.  A host-side switch read at launch time (round 6). */
int muse_conv_persistent(int32_t mode);
int muse_conv2d_nhwc_gn_split2(const float* x, const float* gn_scale, const float* gn_shift, const void* w_hi, const void* w_lo,
                               const float* bias, const float* residual, float* out, double* gn_partial, int32_t gn_groups,
                               int32_t batch, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t KS, void* stream);
int muse_groupnorm_scale_shift(const double* partial, int32_t nchunk, const float* gamma, const float* beta, float* scale,
                               float* shift, int32_t batch, int32_t HW, int32_t C, int32_t groups, float eps, void* stream);
/* Encoder.conv_in (muse/modeling_maskgit_vqgan.py:175; taming :380): 3x3, padding 1, Cin <= 4 image channels -> Cout, as a direct
 * exact-f32 convolution (K = 9 * Cin is no matrix-core problem; the output write is what costs).  x [B, H, W, Cpad] f32 NHWC (Cpad a
 * multiple of 4, channels >= 4 ignored), w4 [Cout][9][4] f32 (tap-major, input channels zero-padded to 4), bias f32 [Cout] or NULL.
 * gn_partial (optional, needs gn_groups * 4 == Cout): [B, H, gn_groups, 2] f64 sum / sum of squares of the output per image row -
 * the `partial` / nchunk = H a following GroupNorm consumes.  Cout % 4 == 0, Cout / 4 divides 256. */
int muse_conv_in_direct(const float* x, const float* w4, const float* bias, float* out, double* gn_partial, int32_t gn_groups,
                        int32_t batch, int32_t H, int32_t W, int32_t Cin, int32_t Cpad, int32_t Cout, void* stream);
/* Decoder.conv_out behind norm_out + swish (muse/modeling_maskgit_vqgan.py:236-240; hidden_channels -> 3 image channels) as ONE
 * direct exact-f32 convolution: x [B, H, W, C] f32 is normalised with the per-image affine form of the GroupNorm (scale / shift
 * [B, C] f32 from muse_groupnorm_scale_shift), activated (SiLU) and convolved 3x3 / padding 1 (zero padding after the activation)
 * with w [Cout][9][C] f32 (+ bias [Cout]) -> out [B, H, W, Cout] f32.  H, W % 16 == 0, C % 32 == 0, Cout <= 4. */
int muse_conv_out_direct(const float* x, const float* scale, const float* shift, const float* w, const float* bias, float* out,
                         int32_t batch, int32_t H, int32_t W, int32_t C, int32_t Cout, void* stream);
/* F.interpolate(scale_factor=2, mode="nearest") of an NHWC tensor (f32, C % 4 == 0; bf16, C % 8 == 0): the taming Upsample without
 * its convolution (muse/modeling_taming_vqgan.py:36-47, resample_with_conv = False) -> y [B, 2H, 2W, C] */
int muse_upsample2x_nhwc(const void* x, void* y, int32_t dtype, int32_t batch, int32_t H, int32_t W, int32_t C, void* stream);
/* F.interpolate(scale_factor=2, "nearest") of x [B, H, W, C] f32 written as the (hi, lo) bf16 operand planes [B, 2H, 2W, C] of
 * muse_conv2d_nhwc_split2 (UpsamplingBlock, :141-149): hi = bf16(x), lo = bf16(x - hi).  C % 8 == 0. */
int muse_upsample2x_split_nhwc(const float* x, void* y_hi, void* y_lo, int32_t batch, int32_t H, int32_t W, int32_t C, void* stream);
/* f32 in; output as y_hi = bf16(y), y_lo = bf16(y - y_hi) (two [B, HW, C] bf16 planes) for muse_conv2d_nhwc_split2.
 * stats_nchunk == 0: statistics computed here into `partial`; > 0: `partial` = [B, stats_nchunk, G, 2] already filled. */
int muse_groupnorm_silu_nhwc_split(const float* x, void* y_hi, void* y_lo, const float* gamma, const float* beta,
                                   double* partial, int32_t stats_nchunk, int32_t batch, int32_t HW, int32_t C,
                                   int32_t groups, float eps, int32_t apply_silu, void* stream);
int muse_avgpool2x2_nhwc(const void* x, void* y, int32_t dtype, int32_t batch, int32_t H, int32_t W, int32_t C,
                         void* stream); /* F.avg_pool2d(2,2), :112; H,W = input dims */
/* f32 avg_pool2d(2,2) that also leaves the GroupNorm(groups) statistics of its OUTPUT in `partial`
 * ([B, muse_groupnorm_nchunk(H/2*W/2), groups, 2] f64): the first norm of the next encoder level (:73) skips its own pass. */
int muse_avgpool2x2_nhwc_stats(const float* x, float* y, double* partial, int32_t groups, int32_t batch, int32_t H, int32_t W,
                               int32_t C, void* stream);
/* layout / dtype conversion: NCHW f32 <-> NHWC (f32|bf16), channel padding with zeros up to Cpad */
int muse_nchw_to_nhwc(const float* in, void* out, int32_t out_dtype, int32_t batch, int32_t C, int32_t HW, int32_t Cpad,
                      void* stream);
int muse_nhwc_to_nchw(const void* in, int32_t in_dtype, float* out, int32_t batch, int32_t C, int32_t HW, int32_t Cpad,
                      void* stream);
/* argmin over codes of dist [rows, ncodes] f32 (first index wins ties, like torch.argmin), :279/:346 */
int muse_argmin_rows(const float* dist, int64_t* idx, int64_t rows, int32_t ncodes, int64_t ld, void* stream);
/* sum of squares per row, f32 in / f32 out (VectorQuantizer.compute_distances :308-309) */
int muse_row_sumsq(const float* x, float* out, int64_t rows, int32_t cols, int64_t ld, void* stream);
/* codebook gather (get_codebook_entry :318-324): out[r, :] = codebook[idx[r], :] (f32 -> out_dtype) */
int muse_gather_rows(const float* table, const int64_t* idx, void* out, int32_t out_dtype, int64_t rows, int32_t cols,
                     void* stream);

/* Probe used by the test-suite to pin the ds_read_b64_tr_b16 lane mapping this library relies on:
 * out[lane*4 + j] for a 64-lane wave reading lds[i] = i (uint16) with per-lane byte address addr[lane]. */
int muse_probe_tr16(const int32_t* addr, int32_t* out, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * MaskGiTUViT_v2 (muse/modeling_transformer_v2.py) - the kernels that model needs beyond the
 * MaskGit ones.  f32.  (csrc/uvit.hip)
 * muse_norm_res_fwd: v = x (+ res); pre = v (optional); y = RMSNorm(v) * w (mode 0, unfused_rms_norm :673-691) or
 *   LayerNorm(v) * w (mode 1, unfused_layer_norm :726-737); w may be NULL.  cols % 4 == 0.
 * muse_adaln_fwd: AdaLNModulation :1025-1037, y[b,r,:] = x[b,r,:] * (1 + ss[b,:C]) + ss[b,C:], ss = mapper(silu(cond)).
 * muse_dwconv3x3_nhwc: ResBlock.depthwise :596-603 (groups = C, padding 1), weight [C][3][3].
 * muse_grn_fwd: GlobalResponseNorm :741-751 on [B, S, C]; scratch 2*B*C floats, returns [G | N] for the backward.
 * muse_sinusoidal_encode: sinusoidal_encode :59-76, out [n, dim].
 * muse_weighted_mean: out[0] = sum(v*w) / sum(w)  (per-token loss weighting :311-316). */
int muse_norm_res_fwd(const float* x, const float* res, const float* w, float* y, float* pre, int64_t rows, int32_t cols,
                      float eps, int32_t mode, void* stream);
int muse_adaln_fwd(const float* x, const float* ss, float* y, int32_t batch, int64_t rows_per_batch, int32_t C, void* stream);
/* ... with the f32 result and / or its bf16 copy (the next GEMM's operand in the bf16 compute mode); either pointer may be null */
int muse_adaln_fwd_ex(const float* x, const float* ss, float* y, void* y_bf16, int32_t batch, int64_t rows_per_batch, int32_t C,
                      void* stream);
int muse_silu_fwd(const float* x, float* y, int64_t n, void* stream);
int muse_dwconv3x3_nhwc(const float* x, const float* w, float* y, int32_t batch, int32_t H, int32_t W, int32_t C, void* stream);
/* 2x2 space-to-depth (inverse = 0: x full [B, H, W, C] -> y packed [B, H/2, W/2, 4C], channel order (di, dj, c)) and its inverse
 * (inverse = 1: x packed -> y full); H, W are the FULL-resolution sides in both directions (even), C % 4 == 0.  With it the stride-2
 * 2x2 convolution of DownsampleBlock (reference muse/modeling_transformer_v2.py:510-514) and the stride-2 2x2 transposed convolution
 * of UpsampleBlock (:558-562) are single products on muse_gemm; each direction is the other's backward. */
int muse_space_to_depth2_nhwc(const float* x, float* y, int32_t batch, int32_t H, int32_t W, int32_t C, int32_t inverse, void* stream);
int muse_grn_fwd(const float* x, const float* gamma, const float* beta, float* y, float* scratch, int32_t batch, int64_t S,
                 int32_t C, void* stream);
/* ... with the f32 result and / or its bf16 copy (the next GEMM's operand in the bf16 compute mode); either pointer may be null */
int muse_grn_fwd_ex(const float* x, const float* gamma, const float* beta, float* y, void* y_bf16, float* scratch, int32_t batch,
                    int64_t S, int32_t C, void* stream);
int muse_sinusoidal_encode(const float* f, float* out, int64_t n, int32_t dim, float max_positions, void* stream);
int muse_weighted_mean(const float* v, const float* w, float* out, int64_t n, void* stream);
/* backward of the above (f32).  v = the forward's pre-norm sum x (+ res); dv = dx = dres; dw_partial [nblk, cols] is folded
 * with muse_colsum (nblk from _nblk; cols <= 4096).  muse_adaln_bwd: dx and dss [B, 2C] = (sum_r dy x | sum_r dy).
 * muse_dwconv3x3_bwd: dx and dw_partial [nchunk, C*9] (muse_colsum -> dw [C][3][3]).  muse_grn_bwd: `stats` = the forward's
 * scratch [G | N]; work 4*B*C floats, on return work[0:B*C] = per-image dbeta terms, work[B*C:2*B*C] = per-image dgamma terms
 * (muse_colsum over the B rows).  muse_scale_rows: x[r, :cols] *= w[r] * num[0] / den[0] (weighted-loss gradient). */
int muse_norm_res_bwd_nblk(int64_t rows);
int muse_norm_res_bwd(const float* dy, const float* dpre, const float* v, const float* w, float* dv, float* dw_partial,
                      int64_t rows, int32_t cols, float eps, int32_t mode, void* stream);
/* ... also writing a bf16 copy of dv (may be null) */
int muse_norm_res_bwd_ex(const float* dy, const float* dpre, const float* v, const float* w, float* dv, void* dv_bf16,
                         float* dw_partial, int64_t rows, int32_t cols, float eps, int32_t mode, void* stream);
/* Norm + AdaLN as one op (every norm of a MaskGiTUViT_v2 TransformerLayer feeds an AdaLNModulation, :757-792):
 *   fwd: v = x (+ res); pre = v (optional); n = Norm(v) * w (mode 0 RMSNorm / 1 LayerNorm); m[b,r,:] = n * (1 + ss[b,:C]) + ss[b,C:],
 *        written as f32 (m) and / or bf16 (m_bf16); n itself is not written.  cols % 4 == 0, cols <= 1024.
 *   bwd: dm = d(m), dpre = the gradient that reached `pre` directly (optional), v = pre.  dv = d(x) = d(res) (+ bf16 copy, optional);
 *        dw_partial [nblk, cols] (muse_colsum -> d(w)); dss_partial [nblk, 2 cols] = (sum dm n | sum dm) over each block's 16 rows,
 *        nblk = muse_norm_res_bwd_nblk(rows); rows_per_batch % 16 == 0, so image b owns rows_per_batch / 16 consecutive blocks:
 *        muse_colsum_segments(dss_partial, dss, batch, rows_per_batch / 16, 2 cols) gives d(ss) [batch, 2 cols].
 * muse_colsum_segments: out[s, c] = sum_{k < seg_rows} part[s * seg_rows + k, c], fixed order. */
int muse_norm_adaln_fwd(const float* x, const float* res, const float* w, const float* ss, float* pre, float* m, void* m_bf16,
                        int32_t batch, int64_t rows_per_batch, int32_t cols, float eps, int32_t mode, void* stream);
int muse_norm_adaln_bwd(const float* dm, const float* dpre, const float* v, const float* w, const float* ss, float* dv, void* dv_bf16,
                        float* dw_partial, float* dss_partial, int32_t batch, int64_t rows_per_batch, int32_t cols, float eps,
                        int32_t mode, void* stream);
/* "bf16x3" mode forms: the f32 result AND its (hi, lo) bf16 operand planes [2][rows][cols] (what muse_split_f32_to_bf16x2 makes of it),
 * so that the products reading m / dv (muse_gemm_x3) need no split pass */
int muse_norm_adaln_fwd_x3(const float* x, const float* res, const float* w, const float* ss, float* pre, float* m, void* planes,
                           int32_t batch, int64_t rows_per_batch, int32_t cols, float eps, int32_t mode, void* stream);
int muse_norm_adaln_bwd_x3(const float* dm, const float* dpre, const float* v, const float* w, const float* ss, float* dv, void* planes,
                           float* dw_partial, float* dss_partial, int32_t batch, int64_t rows_per_batch, int32_t cols, float eps,
                           int32_t mode, void* stream);
int muse_colsum_segments(const float* part, float* out, int32_t nseg, int32_t seg_rows, int32_t cols, void* stream);
int muse_adaln_bwd(const float* dy, const float* x, const float* ss, float* dx, float* dss, int32_t batch,
                   int64_t rows_per_batch, int32_t C, void* stream);
int muse_silu_bwd(const float* x, const float* dy, float* dx, int64_t n, void* stream);
int muse_dwconv3x3_bwd_nchunk(int64_t pixels);
int muse_dwconv3x3_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw_partial, int32_t batch, int32_t H,
                       int32_t W, int32_t C, void* stream);
int muse_grn_bwd(const float* dy, const float* x, const float* gamma, const float* stats, float* dx, float* work,
                 int32_t batch, int64_t S, int32_t C, void* stream);
int muse_scale_rows(float* x, const float* w, const float* num, const float* den, int64_t rows, int32_t cols, int64_t ld,
                    void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MUSE_HIP_H */
