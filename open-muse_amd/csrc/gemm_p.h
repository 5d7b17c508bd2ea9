// persistent 256 x 256 GEMM (gemm256p.h / gemm_p.hip), as seen from muse_gemm
#pragma once
#include "gemm_core.h"
bool gemm256p_takes(const GemmParams& p, int la, int lb, int batch, bool f32_out);
// -1: no queue slot for this stream (the caller falls back to the launch-per-tile kernel); otherwise a hipError_t
// half_ops: the operands are IEEE half (muse_gemm dtype MUSE_F16; f32 output, both operands k-contiguous)
int launch_gemm256p(const GemmParams& p, int la, int lb, bool f32_out, hipStream_t stream, bool half_ops = false);
