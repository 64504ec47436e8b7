// Dev aid: semantics of DPP wave_shr:1 / wave_shl:1 on this GPU (which neighbour a lane receives, and what
// the end lanes get).  hipcc --offload-arch=gfx950 -o /tmp/d profiles/dpp_wave_shift.hip && /tmp/d
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* o) {
  const float v = 100.f + (float)threadIdx.x;
  const int r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, false);  // wave_shr:1
  const int l = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, false);  // wave_shl:1
  o[2 * threadIdx.x] = __builtin_bit_cast(float, r);
  o[2 * threadIdx.x + 1] = __builtin_bit_cast(float, l);
}
int main() {
  float* d; (void)hipMalloc(&d, 128 * sizeof(float));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[128]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l : {0, 1, 15, 16, 17, 31, 32, 62, 63}) printf("lane %2d: wave_shr:1 -> %6.1f   wave_shl:1 -> %6.1f\n", l, h[2 * l], h[2 * l + 1]);
  return 0;
}
