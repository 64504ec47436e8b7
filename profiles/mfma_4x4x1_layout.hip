// Dev aid: prints the operand/result layout of v_mfma_f32_4x4x1_16b_f32 on this GPU.
// Build + run:  hipcc --offload-arch=gfx950 -o /tmp/l profiles/mfma_4x4x1_layout.hip && /tmp/l
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
  const int l = threadIdx.x;
  // A value encodes (lane) in units of 1, B value encodes (lane) in units of 1000: product tells both lanes
  const float a = 1.0f + l;            // distinct per lane
  const float b = 1.0f + 100.0f * l;   // distinct per lane
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main() {
  float* d; hipMalloc(&d, 256 * sizeof(float));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  // decode: h = (1+la)*(1+100 lb) for the A lane la and B lane lb that met in this output
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    int fa = -1, fb = -1;
    for (int la = 0; la < 64 && fa < 0; ++la) for (int lb = 0; lb < 64; ++lb)
      if (h[l * 4 + r] == (1.0f + la) * (1.0f + 100.0f * lb)) { fa = la; fb = lb; break; }
    if (l < 12 || l >= 60) printf("D lane %2d reg %d = A(lane %2d) * B(lane %2d)\n", l, r, fa, fb);
  }
  // check hypothesis: D[lane l][reg r] = A[lane 4*(l/4)+r] * B[lane l]
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r)
    if (h[l * 4 + r] != (1.0f + (4 * (l / 4) + r)) * (1.0f + 100.0f * l)) ++bad;
  printf("hypothesis D[l][r] = A[4*(l/4)+r] * B[l]: %s (%d mismatches)\n", bad ? "WRONG" : "OK", bad);
  return 0;
}
