// Shared between attention.hip (general fused attention) and attention2.hip (the 32 x 32-block kernels for one-tile self-attention).
#pragma once
#include "common.h"

typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}
#define ATT_OOB 0x7ffffff0u

struct AttnParams {
  const bf16_t *q, *k, *v, *o, *d_o;       // o / d_o: forward output (read by backward), upstream gradient
  bf16_t *out, *dq, *dk, *dv;              // out: forward ctx
  float *lse, *dsum;                       // [batch*heads, sqp]
  long ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;   // elements between consecutive tokens
  long bq, bk, bv, bo, bdo, bdq, bdk, bdv;           // elements between consecutive images
  int nh, sq, skv, sqp;
  int nchunk, chunk_rows;                  // stationary rows per workgroup (multiple of 16)
  int tile_rows, ntile;                    // streamed rows per LDS tile (multiple of 32), number of tiles
  int shared;                              // 1: the workgroup's last stationary tile is split over all waves (see below)
  float alpha;
};

// attention2.hip: 1 = launched, 0 = shape not taken (the caller runs the general kernels), < 0 = launch error
int attn2_fwd_try(const AttnParams& P, int head_dim, int batch, hipStream_t st);
int attn2_bwd_try(const AttnParams& P, int head_dim, int batch, hipStream_t st);
