// Batched 2-D FFT feature step of the notebook pipeline (Demo.ipynb:169-175,
// src/lofar_tools.py:24-30):
//   F = fftn(x, dim=(2,3), norm='ortho'); roll both dims by 64 (fftshift);
//   y = cat(Re F, Im F, dim=1); clamp(-c, c)
// x (B,C,128,128) real -> y (B,2C,128,128).
//
// One workgroup per (b,c) image.  The whole 128x128 complex image lives in LDS
// (128 KiB of the CU's 160 KiB), so HBM sees exactly one read of the image and
// one write of the two output planes (196,608 B per image).  Row transforms,
// then column transforms, are in-place radix-2 decimation-in-frequency passes;
// consecutive lanes always touch consecutive complex elements (ds_read/write_b64,
// conflict-free).  DIF leaves both axes in bit-reversed order, which the store
// pass undoes together with the fftshift (index XOR 64), the 1/128 'ortho' scale
// and the clamp, writing coalesced rows.
#include "kernels.h"

namespace lshm {

#define FFT_N 128
#define FFT_THREADS 1024

__device__ __forceinline__ int bitrev7(int v) { return (int)(__brev((unsigned)v) >> 25); }

__global__ __launch_bounds__(FFT_THREADS) void fft2_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                           int C, float clampv) {
  extern __shared__ float2 img[];  // [128][128]
  __shared__ float2 tw[FFT_N / 2];
  const int t = threadIdx.x;
  const int bc = blockIdx.x;  // b*C + c
  const int b = bc / C, c = bc - b * C;
  const float* src = x + (size_t)bc * FFT_N * FFT_N;
  if (t < FFT_N / 2) {
    float sn, cs;
    sincospif(-2.0f * (float)t / (float)FFT_N, &sn, &cs);
    tw[t] = make_float2(cs, sn);
  }
  // load (coalesced float4 per thread)
  for (int i = t; i < FFT_N * FFT_N / 4; i += FFT_THREADS) {
    const f32x4 v = reinterpret_cast<const f32x4*>(src)[i];
    img[4 * i + 0] = make_float2(v[0], 0.f);
    img[4 * i + 1] = make_float2(v[1], 0.f);
    img[4 * i + 2] = make_float2(v[2], 0.f);
    img[4 * i + 3] = make_float2(v[3], 0.f);
  }
  __syncthreads();
  // ---- rows: butterflies (row r, pair index q in [0,64)): 8192 per stage
  for (int h = FFT_N / 2; h >= 1; h >>= 1) {
    for (int idx = t; idx < FFT_N * FFT_N / 2; idx += FFT_THREADS) {
      const int r = idx >> 6, q = idx & 63;
      const int pos = q & (h - 1);
      const int i = ((q - pos) << 1) + pos;  // start of the 2h block + position
      float2* p0 = &img[r * FFT_N + i];
      float2* p1 = p0 + h;
      const float2 a = *p0, bb = *p1;
      const float2 w = tw[pos * (FFT_N / 2 / h)];
      const float dr = a.x - bb.x, di = a.y - bb.y;
      *p0 = make_float2(a.x + bb.x, a.y + bb.y);
      *p1 = make_float2(dr * w.x - di * w.y, dr * w.y + di * w.x);
    }
    __syncthreads();
  }
  // ---- columns: butterflies (pair index q along rows, column cc); lanes run along the column index
  for (int h = FFT_N / 2; h >= 1; h >>= 1) {
    for (int idx = t; idx < FFT_N * FFT_N / 2; idx += FFT_THREADS) {
      const int q = idx >> 7, cc = idx & 127;
      const int pos = q & (h - 1);
      const int i = ((q - pos) << 1) + pos;
      float2* p0 = &img[i * FFT_N + cc];
      float2* p1 = p0 + h * FFT_N;
      const float2 a = *p0, bb = *p1;
      const float2 w = tw[pos * (FFT_N / 2 / h)];
      const float dr = a.x - bb.x, di = a.y - bb.y;
      *p0 = make_float2(a.x + bb.x, a.y + bb.y);
      *p1 = make_float2(dr * w.x - di * w.y, dr * w.y + di * w.x);
    }
    __syncthreads();
  }
  // ---- store: output (u,v) <- frequency ((u+64)%128, (v+64)%128), held at bit-reversed indices
  float* ore = out + ((size_t)b * 2 * C + c) * FFT_N * FFT_N;
  float* oim = out + ((size_t)b * 2 * C + C + c) * FFT_N * FFT_N;
  const float scale = 1.0f / (float)FFT_N;
  for (int idx = t; idx < FFT_N * FFT_N; idx += FFT_THREADS) {
    const int u = idx >> 7, v = idx & 127;
    const float2 f = img[bitrev7(u ^ 64) * FFT_N + bitrev7(v ^ 64)];
    ore[idx] = fminf(fmaxf(f.x * scale, -clampv), clampv);
    oim[idx] = fminf(fmaxf(f.y * scale, -clampv), clampv);
  }
}

int fft2_ortho_shift_cat_clamp(const float* x, float* out, int B, int C, float clampv, hipStream_t st) {
  const size_t shmem = (size_t)FFT_N * FFT_N * sizeof(float2);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fft2_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  if (e != hipSuccess) { set_last_error("fft2: cannot raise dynamic LDS limit to 128 KiB"); return (int)e; }
  hipLaunchKernelGGL(fft2_kernel, dim3(B * C), dim3(FFT_THREADS), shmem, st, x, out, C, clampv);
  return check_launch("fft2");
}

}  // namespace lshm
