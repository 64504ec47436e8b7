// Dev aid: issue rate of v_mfma_f32_4x4x1_16b_f32 vs v_mfma_f32_16x16x4_f32 (FLOP/clk/SIMD should match).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0) {
  float a = a0 + threadIdx.x, b = a0 * 0.5f + threadIdx.x;
  f32x4 c[8];
  for (int i = 0; i < 8; ++i) c[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      c[i] = MODE ? __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[i], 0, 0, 0)
                  : __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
static void run(const char* name, double flop_per_inst) {
  float* d; hipMalloc(&d, 1024 * 256 * sizeof(float));
  const int iters = 4096, blocks = 1024;  // 4 blocks per CU = 4 waves per SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 16, 1.0f);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double insts = (double)blocks * 4 * iters * 8;  // wave-level MFMA instructions
  printf("%s: %.3f ms, %.1f TFLOP/s, %.2f ns per instruction per SIMD\n", name, ms, insts * flop_per_inst / ms / 1e9,
         ms * 1e6 / (insts / 1024.0));
  hipFree(d);
}
int main() {
  run<0>("v_mfma_f32_16x16x4_f32   ", 2.0 * 16 * 16 * 4);
  run<1>("v_mfma_f32_4x4x1_16b_f32 ", 2.0 * 16 * 4 * 4 * 1);
  return 0;
}
