// Batched 2D FFT feature op (placeholder until the LDS radix kernel lands).
#include "kernels.h"
namespace lshm {
int fft2_ortho_shift_cat_clamp(const float*, float*, int, int, float, hipStream_t) {
  set_last_error("fft2: not built yet");
  return LSHM_ERR_UNSUPPORTED;
}
}  // namespace lshm
