// Experiment helper (NOT part of libmuse_hip.so, NOT yet run on hardware - written when the round's GPU budget was spent):
// HIP streams restricted to a set of CUs, for scripts/exp/stream_placement.py.
//   hipcc --offload-arch=gfx950 -shared -fPIC -o libstream_placement.so stream_placement.hip
#include <hip/hip_runtime.h>
#include <cstdint>

extern "C" int sp_stream_create(const uint32_t* mask_words, int32_t n_words, void** stream_out) {
  hipStream_t s = nullptr;
  const hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, mask_words);
  if (e != hipSuccess) return (int)e;
  *stream_out = (void*)s;
  return 0;
}
extern "C" int sp_stream_destroy(void* stream) { return (int)hipStreamDestroy((hipStream_t)stream); }
// the CU mask a stream really got (the runtime may intersect it with HSA_CU_MASK / ROC_GLOBAL_CU_MASK)
extern "C" int sp_stream_get_mask(void* stream, uint32_t* mask_words, int32_t n_words) {
  return (int)hipExtStreamGetCUMask((hipStream_t)stream, (uint32_t)n_words, mask_words);
}
