// Token-level sampling / masking kernels around the transformer (the "next" rows f2 / f3 of SURVEY.md section 8):
//   muse_sample_step   - one iteration of MaskGit parallel decoding (muse/modeling_transformer.py:1409-1454 and
//                        muse/modeling_transformer_v2.py:434-474 with muse/sampling.py:30-35): classifier-free-guidance mix,
//                        softmax, categorical sample, confidence with Gumbel noise, k-th-smallest threshold, re-mask
//   muse_mask_tokens   - training/train_muse.py:149-226 mask_or_random_replace_tokens (random or contiguous-region masks,
//                        labels / loss weights variants)
//   muse_cond_dropout  - training/train_muse.py:715-731 (conditioning dropout for classifier-free guidance)
// All are latency-class kernels (a few MB at most); the point is one launch instead of ~10 ATen launches per decoding step and
// bit-exact agreement with the reference when the random draws are supplied by the caller.
#include "common.h"
#include "../../include/muse_hip.h"
#include <float.h>

// ---- Philox4x32-10 (counter-based; Salmon et al. 2011), used when the caller supplies no random draws -------------------
struct u4 { uint32_t x, y, z, w; };
__device__ __forceinline__ u4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return u4{c0, c1, c2, c3};
}
__device__ __forceinline__ float u01_open_low(uint32_t r) { return ((float)(r >> 8) + 1.0f) * (1.0f / 16777216.0f); }   // (0, 1]
__device__ __forceinline__ float u01_open_high(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }            // [0, 1)

__device__ __forceinline__ float clamp_log(float t) { return logf(fmaxf(t, 1e-20f)); }   // muse/sampling.py:9-10

// =================================================================================================================
// phase 1: one wave per (image, position) row
// =================================================================================================================
__global__ __launch_bounds__(256) void sample_rows_kernel(const float* __restrict__ cond, const float* __restrict__ uncond, float guidance,
                                                          int64_t img_stride, int64_t ld, int seq, int V, const int64_t* __restrict__ input_ids, int64_t mask_id,
                                                          const float* __restrict__ noise_exp, const float* __restrict__ noise_u,
                                                          uint64_t seed, uint32_t step, float temperature, int64_t rows,
                                                          int64_t* __restrict__ raw_ids, int64_t* __restrict__ ids, float* __restrict__ conf) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const int64_t img = row / seq, pos = row - img * seq;
  const float* c = cond + img * img_stride + pos * ld;
  const float* u = uncond ? uncond + img * img_stride + pos * ld : nullptr;
  auto logit = [&](int j) {
    if (!u) return c[j];
    const float uu = u[j];
    return __fadd_rn(uu, __fmul_rn(guidance, __fsub_rn(c[j], uu)));   // uncond + scale * (cond - uncond), three roundings (:403)
  };
  float m = -INFINITY;
  for (int j = lane; j < V; j += 64) m = fmaxf(m, logit(j));
  m = wave_max(m);
  float s = 0.f;
  for (int j = lane; j < V; j += 64) s += expf(logit(j) - m);
  s = wave_sum(s);
  // categorical sample the way torch.multinomial(num_samples = 1) draws it: argmax_j p_j / q_j with q_j ~ Exp(1) (first index on ties)
  float best = -INFINITY;
  int besti = 0x7fffffff;
  for (int j = lane; j < V; j += 64) {
    const float p = expf(logit(j) - m) / s;
    float q;
    if (noise_exp) q = noise_exp[row * V + j];
    else {
      const u4 r = philox4x32_10((uint32_t)row, (uint32_t)(row >> 32), (uint32_t)j, step * 2u, (uint32_t)seed, (uint32_t)(seed >> 32));
      q = -logf(u01_open_low(r.x));
    }
    const float v = p / q;
    if (v > best) { best = v; besti = j; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(besti, o, 64);
    if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
  }
  if (lane == 0) {
    const int64_t cur = input_ids[row];
    const bool unknown = cur == mask_id;
    const int64_t fin = unknown ? (int64_t)besti : cur;
    if (raw_ids) raw_ids[row] = besti;
    ids[row] = fin;
    // confidence (muse/sampling.py:30-31): log(clamp(p_selected)) + temperature * gumbel;  known tokens carry FLT_MAX as their p
    float psel = FLT_MAX;
    if (unknown) psel = expf(logit(besti) - m) / s;
    float un;
    if (noise_u) un = noise_u[row];
    else {
      const u4 r = philox4x32_10((uint32_t)row, (uint32_t)(row >> 32), 0u, step * 2u + 1u, (uint32_t)seed, (uint32_t)(seed >> 32));
      un = u01_open_high(r.x);
    }
    const float gumbel = -clamp_log(-clamp_log(un));
    conf[row] = __fadd_rn(clamp_log(psel), __fmul_rn(temperature, gumbel));
  }
}

// =================================================================================================================
// phase 2: one workgroup per image: mask_len = max(1, min(#unknown - 1, scheduled)), threshold = the mask_len-th smallest
// confidence (0-based, torch.sort ascending + gather), next input = masked where confidence < threshold
// =================================================================================================================
__global__ __launch_bounds__(256) void remask_kernel(const int64_t* __restrict__ input_ids, const int64_t* __restrict__ ids,
                                                     const float* __restrict__ conf, int64_t* __restrict__ next_ids, int seq,
                                                     int64_t mask_id, int sched_len) {
  __shared__ float cf[4096];
  __shared__ int cnt;
  __shared__ float cut;
  const int b = blockIdx.x;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  int local = 0;
  for (int i = threadIdx.x; i < seq; i += 256) {
    cf[i] = conf[(long)b * seq + i];
    local += input_ids[(long)b * seq + i] == mask_id;
  }
  local = (int)wave_sum((float)local);
  if ((threadIdx.x & 63) == 0) atomicAdd(&cnt, local);
  __syncthreads();
  int k = min(cnt - 1, sched_len);
  k = max(1, k);
  k = min(k, seq - 1);   // (torch.gather would raise past the end; seq >= 2 on every path that reaches here)
  // the element of ascending rank k: value v with  #(x < v) <= k < #(x <= v)
  for (int i = threadIdx.x; i < seq; i += 256) {
    const float v = cf[i];
    int lt = 0, le = 0;
    for (int j = 0; j < seq; ++j) { const float x = cf[j]; lt += x < v; le += x <= v; }
    if (lt <= k && k < le) cut = v;   // every thread that qualifies writes the same value
  }
  __syncthreads();
  const float c = cut;
  for (int i = threadIdx.x; i < seq; i += 256) {
    const long o = (long)b * seq + i;
    next_ids[o] = cf[i] < c ? mask_id : ids[o];
  }
}

extern "C" int muse_sample_step(const float* cond_logits, const float* uncond_logits, float guidance_scale, int64_t img_stride, int64_t ld,
                                int32_t vocab,
                                const int64_t* input_ids, int64_t mask_id, const float* noise_exp, const float* noise_u,
                                uint64_t seed, uint32_t step, float temperature, int32_t sched_mask_len, int32_t batch, int32_t seq,
                                int64_t* raw_sampled, int64_t* sampled, int64_t* next_ids, float* conf_scratch, void* stream) {
  if (!cond_logits || !input_ids || !sampled || !next_ids || !conf_scratch || vocab <= 0) return MUSE_ERR_BAD_ARG;
  if (seq > 4096 || seq < 2) return MUSE_ERR_UNSUPPORTED;
  if (batch <= 0) return 0;
  const int64_t rows = (int64_t)batch * seq;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sample_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, cond_logits, uncond_logits, guidance_scale,
                     img_stride, ld, seq, vocab, input_ids, mask_id, noise_exp, noise_u, seed, step, temperature, rows, raw_sampled, sampled, conf_scratch);
  MUSE_CHECK_LAUNCH();
  hipLaunchKernelGGL(remask_kernel, dim3(batch), dim3(256), 0, st, input_ids, (const int64_t*)sampled, (const float*)conf_scratch,
                     next_ids, seq, mask_id, sched_mask_len);
  return (int)hipGetLastError();
}

// =================================================================================================================
// training/train_muse.py:149-226 mask_or_random_replace_tokens
// =================================================================================================================
// One workgroup per image.  mask source: rects != NULL: rectangle (y0, x0, h, w) on the sqrt(seq) grid (the contiguous-region
// branch; the reference draws it with Python's `random` on the host, so does the caller); else the argsort-threshold rule
// mask[j] = argsort(noise)[j] < k (k = round(seq * p) clamped to >= 1) of :175-176, the same rule as muse_mask_sample.
__global__ __launch_bounds__(256) void mask_tokens_kernel(const int64_t* __restrict__ tokens, const float* __restrict__ timesteps,
                                                          const float* __restrict__ mask_prob_in, const float* __restrict__ noise,
                                                          const int32_t* __restrict__ rects, int64_t* __restrict__ input_ids,
                                                          int64_t* __restrict__ labels, float* __restrict__ loss_weight,
                                                          float* __restrict__ mask_prob, int seq, int side, int64_t mask_id,
                                                          float min_rate, int all_labels, float weight_min) {
  __shared__ float nz[4096];
  __shared__ int perm[4096];
  const int b = blockIdx.x;
  float prob;
  if (mask_prob_in) prob = mask_prob_in[b];   // eval_mask_ratios branch (:152-154): no schedule, no clip
  else {
    const float arg = (timesteps[b] * 3.14159265358979323846f) * 0.5f;   // cosine_schedule (muse/sampling.py:38-39), see muse_mask_sample
    prob = fmaxf((float)cos((double)arg), min_rate);
  }
  if (!rects) {
    for (int i = threadIdx.x; i < seq; i += 256) nz[i] = noise[(long)b * seq + i];
    __syncthreads();
    for (int i = threadIdx.x; i < seq; i += 256) {
      const float v = nz[i];
      int r = 0;
      for (int j = 0; j < seq; ++j) { const float u = nz[j]; r += (u < v) || (u == v && j < i); }
      perm[r] = i;
    }
    __syncthreads();
  }
  const int k = (int)fmaxf(rintf((float)seq * prob), 1.0f);
  if (threadIdx.x == 0) mask_prob[b] = prob;
  const float wk = __fmul_rn(__fsub_rn(1.0f, prob), __fsub_rn(1.0f, weight_min));   // (1 - t) * (1 - min_val) (:145-146, t = mask_prob)
  for (int j = threadIdx.x; j < seq; j += 256) {
    bool masked;
    if (rects) {
      const int y = j / side, x = j - y * side;
      const int32_t* r = rects + 4 * b;
      masked = y >= r[0] && y < r[0] + r[2] && x >= r[1] && x < r[1] + r[3];
    } else masked = perm[j] < k;
    const long o = (long)b * seq + j;
    const int64_t t = tokens[o];
    input_ids[o] = masked ? mask_id : t;
    labels[o] = (all_labels || masked) ? t : (int64_t)-100;
    if (loss_weight) loss_weight[o] = __fsub_rn(1.0f, __fmul_rn(masked ? 0.0f : 1.0f, wk));
  }
}

extern "C" int muse_mask_tokens(const int64_t* tokens, const float* timesteps, const float* mask_prob_in, const float* noise,
                                const int32_t* rects, int64_t* input_ids, int64_t* labels, float* loss_weight, float* mask_prob,
                                int32_t batch, int32_t seq, int64_t mask_id, float min_masking_rate, int32_t all_labels,
                                float weight_min, void* stream) {
  if (!tokens || !input_ids || !labels || !mask_prob || (!timesteps && !mask_prob_in) || (!noise && !rects)) return MUSE_ERR_BAD_ARG;
  if (seq > 4096 || seq <= 0) return MUSE_ERR_UNSUPPORTED;
  int side = 0;
  if (rects) {
    while (side * side < seq) ++side;
    if (side * side != seq) return MUSE_ERR_BAD_ARG;
  }
  if (batch <= 0) return 0;
  hipLaunchKernelGGL(mask_tokens_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, tokens, timesteps, mask_prob_in, noise, rects,
                     input_ids, labels, loss_weight, mask_prob, seq, side, mask_id, min_masking_rate, all_labels, weight_min);
  return (int)hipGetLastError();
}

// =================================================================================================================
// training/train_muse.py:715-731: per image keep = (u < p); out = (x * keep != 0) ? x : empty   (the reference's expression:
// an element that is exactly zero takes the empty embedding's value even when the image is kept)
// =================================================================================================================
__global__ void cond_dropout_kernel(const float* __restrict__ x, const float* __restrict__ empty, const float* __restrict__ u,
                                    float* __restrict__ out, long per_image, long total, float p) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / per_image, j = i - b * per_image;
    const float keep = u[b] < p ? 1.0f : 0.0f;
    const float v = x[i];
    out[i] = (v * keep != 0.0f) ? v : empty[j];   // (NaN * 1 != 0 is true, as `.bool()` of NaN is)
  }
}
extern "C" int muse_cond_dropout(const float* x, const float* empty, const float* uniforms, float* out, int32_t batch,
                                 int64_t per_image, float prob, void* stream) {
  if (!x || !empty || !uniforms || !out || per_image <= 0) return MUSE_ERR_BAD_ARG;
  if (batch <= 0) return 0;
  const long total = (long)batch * per_image;
  const unsigned grid = (unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(cond_dropout_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, empty, uniforms, out, (long)per_image,
                     total, prob);
  return (int)hipGetLastError();
}

// =================================================================================================================
// Dropout (nn.Dropout of muse/modeling_transformer.py:177,237 attention probabilities, :779,797 feed-forward, :933,956
// embeddings): y = x * keep / (1 - p), keep_i = [u_i >= p], u_i the Philox stream (seed, offset + i).  No mask tensor: the
// backward pass calls the same function on the gradient with the same (seed, offset).  One Philox block serves 4 elements.
// =================================================================================================================
template <typename T>
__global__ void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, long n, float p, float scale, uint64_t seed, uint64_t offset) {
  const long n4 = (n + 3) >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const uint64_t c = offset + (uint64_t)i;
    const u4 r = philox4x32_10((uint32_t)c, (uint32_t)(c >> 32), 0x6D757365u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long e = i * 4 + j;
      if (e < n) Elem<T>::store(y + e, u01_open_high(rr[j]) >= p ? Elem<T>::load(x + e) * scale : 0.0f);
    }
  }
}
extern "C" int muse_dropout(const void* x, void* y, int32_t dtype, int64_t n, float p, uint64_t seed, uint64_t offset, void* stream) {
  if (!x || !y || p < 0.0f || p >= 1.0f) return MUSE_ERR_BAD_ARG;
  if (n <= 0) return 0;
  const long n4 = (n + 3) / 4;
  const unsigned grid = (unsigned)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  const float scale = 1.0f / (1.0f - p);
  if (dtype == MUSE_F32) hipLaunchKernelGGL(dropout_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, (long)n, p, scale, seed, offset);
  else if (dtype == MUSE_BF16) hipLaunchKernelGGL(dropout_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, (long)n, p, scale, seed, offset);
  else return MUSE_ERR_BAD_ARG;
  return (int)hipGetLastError();
}
