// Library introspection entry points.
#include "common.h"

extern "C" int mvt_abi_version(void) { return 3; }
extern "C" const char* mvt_build_arch(void) { return "gfx950"; }

// A HIP stream restricted to a subset of the compute units (hipExtStreamCreateWithCUMask): the tracker encodes later frames on
// such a stream while the updater of the earlier windows runs on the caller's stream -- with the whole chip open to the encoder
// its convolutions (thousands of workgroups) occupy every CU slot and each of the updater's small kernels on the serial
// virtual-track chain waits for slots to drain; a masked side stream leaves part of the chip permanently free for them.
// mask: n_words x 32 bits, bit i = CU i enabled (host array).  Returns NULL on failure.  Host-side helper, not on the data path.
extern "C" void* mvt_stream_create_cu_mask(const unsigned* mask, int n_words) {
  if (!mask || n_words <= 0) return nullptr;
  hipStream_t s = nullptr;
  if (hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, mask) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return (void*)s;
}
extern "C" int mvt_stream_destroy(void* stream) {
  return stream && hipStreamDestroy((hipStream_t)stream) == hipSuccess ? MVT_OK : MVT_ERR_ARG;
}
