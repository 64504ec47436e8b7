// Batched 2-D FFT feature step of the notebook pipeline (Demo.ipynb:169-175,
// src/lofar_tools.py:24-30):
//   F = fftn(x, dim=(2,3), norm='ortho'); roll both dims by 64 (fftshift);
//   y = cat(Re F, Im F, dim=1); clamp(-c, c)
// x (B,C,128,128) real -> y (B,2C,128,128).  HBM-bound: 196,608 B per image, ~1.1 MFLOP.
//
// One 512-thread workgroup per (b,c) image, 66 KB of LDS (two workgroups per CU overlap each other's
// load / store phases).  The input is real, so
//   * two rows ride in one complex transform, z = row(2r) + i row(2r+1); 64 row transforms, not 128;
//   * only columns k2 = 0..64 of the row spectrum are independent (Hermitian); the two real-valued ones
//     (0 and 64) are packed into one complex sequence as well: 64 column transforms, not 128;
//   * F[k1, 128-k2] = conj F[128-k1, k2] supplies the other half of the plane at store time.
// A 128-point transform is 8 threads x 16 points: a 16-point DFT in registers (two radix-4 stages) over
// the points t, t+8, ..., the twiddle W128^(t kj), one exchange through LDS, then 8-point DFTs.  Lanes
// run over the row pair r (rows pass) or the column k2 (columns pass), so LDS accesses are conflict-free
// and every global store instruction covers one 256-byte segment.  LDS traffic per image: 0.6 MB
// (the radix-2 in-place version it replaces: 3.6 MB, which bounded it at 114 us for B=256, C=4).
#include "kernels.h"

namespace lshm {

#define FFT_N 128
#define FFT_THREADS 512
#define FFT_LD 129  // float2 row stride of the pair-interleaved image: 2-way at worst for 64-lane b64 access

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// forward 4-point DFT in place: y_c = sum_a x_a (-i)^(a c)
__device__ __forceinline__ void dft4(float2& a0, float2& a1, float2& a2, float2& a3) {
  const float2 s02 = cadd(a0, a2), d02 = csub(a0, a2), s13 = cadd(a1, a3), d13 = csub(a1, a3);
  const float2 mid = make_float2(d13.y, -d13.x);  // -i * d13
  a0 = cadd(s02, s13);
  a2 = csub(s02, s13);
  a1 = cadd(d02, mid);
  a3 = csub(d02, mid);
}
// forward 16-point DFT of v[j], j = 4a + b; the result Y[kj], kj = c + 4d, is left at v[4c + d]
__device__ __forceinline__ void dft16(float2 (&v)[16]) {
  constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
#pragma unroll
  for (int b = 0; b < 4; ++b) dft4(v[b], v[4 + b], v[8 + b], v[12 + b]);  // v[4c + b] = P_b[c]
  // W16^(b c) = exp(-2 pi i b c / 16)
  v[4 * 1 + 1] = cmul(v[4 * 1 + 1], make_float2(C1, -S1));   // bc = 1
  v[4 * 1 + 2] = cmul(v[4 * 1 + 2], make_float2(R2, -R2));   // 2
  v[4 * 1 + 3] = cmul(v[4 * 1 + 3], make_float2(S1, -C1));   // 3
  v[4 * 2 + 1] = cmul(v[4 * 2 + 1], make_float2(R2, -R2));   // 2
  v[4 * 2 + 2] = make_float2(v[4 * 2 + 2].y, -v[4 * 2 + 2].x);  // 4: -i
  v[4 * 2 + 3] = cmul(v[4 * 2 + 3], make_float2(-R2, -R2));  // 6
  v[4 * 3 + 1] = cmul(v[4 * 3 + 1], make_float2(S1, -C1));   // 3
  v[4 * 3 + 2] = cmul(v[4 * 3 + 2], make_float2(-R2, -R2));  // 6
  v[4 * 3 + 3] = cmul(v[4 * 3 + 3], make_float2(-C1, S1));   // 9
#pragma unroll
  for (int c = 0; c < 4; ++c) dft4(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);  // v[4c + d] = Y[c + 4d]
}
// forward 8-point DFT of q[t], t = 2a + b; result X[m], m = c + 4d, in natural order
__device__ __forceinline__ void dft8(float2 (&q)[8]) {
  constexpr float R2 = 0.70710678118654752f;
  dft4(q[0], q[2], q[4], q[6]);  // q[2c] = Q_0[c]
  dft4(q[1], q[3], q[5], q[7]);  // q[2c + 1] = Q_1[c]
  q[3] = cmul(q[3], make_float2(R2, -R2));      // W8^1
  q[5] = make_float2(q[5].y, -q[5].x);          // W8^2 = -i
  q[7] = cmul(q[7], make_float2(-R2, -R2));     // W8^3
  float2 x[8];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    x[c] = cadd(q[2 * c], q[2 * c + 1]);
    x[c + 4] = csub(q[2 * c], q[2 * c + 1]);
  }
#pragma unroll
  for (int m = 0; m < 8; ++m) q[m] = x[m];
}

__global__ __launch_bounds__(FFT_THREADS) void fft2_feature_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                                   int C, float clampv) {
  extern __shared__ float2 buf[];  // max(64 x FFT_LD, 128 x 64) float2
  __shared__ float2 tw[FFT_N];     // W128^i
  __shared__ float2 Gs[FFT_N];     // spectrum of the packed (k2 = 0, 64) column
  const int tid = threadIdx.x;
  const int t = tid >> 6, lane = tid & 63;  // t: which of the 8 threads of a transform; lane: row pair / column
  const int bc = blockIdx.x;
  const int b = bc / C, c = bc - b * C;
  const float* src = x + (size_t)bc * FFT_N * FFT_N;
  if (tid < FFT_N) {
    float sn, cs;
    sincospif(-2.0f * (float)tid / (float)FFT_N, &sn, &cs);
    tw[tid] = make_float2(cs, sn);
  }
  // ---- load: coalesced float4, rows 2r / 2r+1 interleaved as the real / imaginary part of z_r
  float* bf = reinterpret_cast<float*>(buf);
#pragma unroll
  for (int i = 0; i < FFT_N * FFT_N / 4 / FFT_THREADS; ++i) {
    const int i4 = tid + FFT_THREADS * i;
    const int row = i4 >> 5, n = (i4 & 31) * 4;
    const f32x4 v = reinterpret_cast<const f32x4*>(src)[i4];
    float* d = bf + (((row >> 1) * FFT_LD + n) << 1) + (row & 1);
    d[0] = v[0]; d[2] = v[1]; d[4] = v[2]; d[6] = v[3];
  }
  __syncthreads();
  float2 v[16];
  // ---- rows, stage A: thread (t, r = lane): points n = t + 8 j of z_r
#pragma unroll
  for (int j = 0; j < 16; ++j) v[j] = buf[lane * FFT_LD + t + 8 * j];
  dft16(v);
  __syncthreads();  // everybody has its inputs in registers: the buffer becomes the exchange area
#pragma unroll
  for (int kj = 0; kj < 16; ++kj)
    buf[lane * FFT_LD + kj * 8 + t] = cmul(v[4 * (kj & 3) + (kj >> 2)], tw[t * kj]);
  __syncthreads();
  // ---- rows, stage B: thread (t, r) finishes kj = 2t, 2t+1:  Z_r[kj + 16 m], m = 0..7
  float2 q0[8], q1[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    q0[u] = buf[lane * FFT_LD + (2 * t) * 8 + u];
    q1[u] = buf[lane * FFT_LD + (2 * t + 1) * 8 + u];
  }
  dft8(q0);
  dft8(q1);
  __syncthreads();
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    buf[lane * FFT_LD + 2 * t + 16 * m] = q0[m];
    buf[lane * FFT_LD + 2 * t + 1 + 16 * m] = q1[m];
  }
  __syncthreads();
  // ---- columns, stage A: thread (t, k2 = lane): points n1 = t + 8 j of column k2 of the row spectrum.
  // Row n1 = 2r + p: R_even = (Z_r[k] + conj Z_r[-k]) / 2,  R_odd = -i (Z_r[k] - conj Z_r[-k]) / 2.
  // Lane 0 carries the two real columns k2 = 0 and k2 = 64 as one complex sequence.
  {
    const int p = t & 1;
    const int km = (FFT_N - lane) & (FFT_N - 1);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int r = (t >> 1) + 4 * j;
      const float2 a = buf[r * FFT_LD + lane];
      const float2 bm = buf[r * FFT_LD + km];
      float2 rv;
      if (lane == 0) {
        const float2 z64 = buf[r * FFT_LD + 64];
        rv = p ? make_float2(a.y, z64.y) : make_float2(a.x, z64.x);
      } else {
        const float2 bcj = make_float2(bm.x, -bm.y);
        if (p) {
          const float2 w = csub(a, bcj);
          rv = make_float2(0.5f * w.y, -0.5f * w.x);
        } else {
          const float2 w = cadd(a, bcj);
          rv = make_float2(0.5f * w.x, 0.5f * w.y);
        }
      }
      v[j] = rv;
    }
  }
  dft16(v);
  __syncthreads();
#pragma unroll
  for (int kj = 0; kj < 16; ++kj) buf[(kj * 8 + t) * 64 + lane] = cmul(v[4 * (kj & 3) + (kj >> 2)], tw[t * kj]);
  __syncthreads();
  // ---- columns, stage B: F[k1 = kj + 16 m][k2], kj = 2t, 2t+1
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    q0[u] = buf[((2 * t) * 8 + u) * 64 + lane];
    q1[u] = buf[((2 * t + 1) * 8 + u) * 64 + lane];
  }
  dft8(q0);
  dft8(q1);
  if (lane == 0) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      Gs[2 * t + 16 * m] = q0[m];
      Gs[2 * t + 1 + 16 * m] = q1[m];
    }
  }
  __syncthreads();
  // ---- store: output (u, v) <- frequency ((u+64)%128, (v+64)%128), 'ortho' scale, clamp
  float* ore = out + ((size_t)b * 2 * C + c) * FFT_N * FFT_N;
  float* oim = out + ((size_t)b * 2 * C + C + c) * FFT_N * FFT_N;
  const float scale = 1.0f / (float)FFT_N;
  auto put = [&](int u, int vv, float re, float im) {
    ore[u * FFT_N + vv] = fminf(fmaxf(re * scale, -clampv), clampv);
    oim[u * FFT_N + vv] = fminf(fmaxf(im * scale, -clampv), clampv);
  };
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int k1 = 2 * t + g + 16 * m;
      const float2 f = g ? q1[m] : q0[m];
      const int u = (k1 + 64) & 127;
      if (lane == 0) {
        // unpack: F[k1, 0] = (G[k1] + conj G[-k1]) / 2,  F[k1, 64] = -i (G[k1] - conj G[-k1]) / 2
        const float2 gm = Gs[(FFT_N - k1) & 127];
        const float2 gc = make_float2(gm.x, -gm.y);
        const float2 s = cadd(f, gc), d = csub(f, gc);
        put(u, 64, 0.5f * s.x, 0.5f * s.y);
        put(u, 0, 0.5f * d.y, -0.5f * d.x);
      } else {
        put(u, lane + 64, f.x, f.y);
        put((((FFT_N - k1) & 127) + 64) & 127, 64 - lane, f.x, -f.y);  // F[-k1, -k2] = conj F[k1, k2]
      }
    }
}

int fft2_ortho_shift_cat_clamp(const float* x, float* out, int B, int C, float clampv, hipStream_t st) {
  const size_t shmem = (size_t)64 * FFT_LD * sizeof(float2);  // >= 128 * 64 float2 of the column exchange
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fft2_feature_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  if (e != hipSuccess) { set_last_error("fft2: cannot raise the dynamic LDS limit"); return (int)e; }
  hipLaunchKernelGGL(fft2_feature_kernel, dim3(B * C), dim3(FFT_THREADS), shmem, st, x, out, C, clampv);
  return check_launch("fft2");
}

// ---- backward of the feature step.  With Z[k] = F[(k + 64) mod 128] (per dimension), F the orthonormal DFT of the
// real image x, and G = gRe + i gIm the gradient w.r.t. (Re Z, Im Z) -- zero where the forward clamp was active --
//   dL/dx[m] = sum_k Re(conj(G[k]) dZ[k]/dx[m]) = (-1)^(m1+m2) * ( Re F(gRe)[m] + Im F(gIm)[m] ),
// i.e. two more transforms of REAL images: the forward kernel runs on the 2C masked gradient planes (no clamp), and
// because it stores F shifted, F(.)[m] is found at index (m + 64) mod 128 of its output.
// g, y: (B, 2C, 128, 128) [Re planes | Im planes]; masked: same shape, scratch; spec: (B, 4C, 128, 128) scratch.
__global__ void fft_mask_kernel(const float* __restrict__ g, const float* __restrict__ y, float clampv, long n,
                                float* __restrict__ masked) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = y[i];
    // torch.clamp passes the gradient where min <= input <= max; an output strictly inside the bounds was not clamped,
    // one AT a bound came from an input at or beyond it (measure-zero difference, resolved as torch does for '>' / '<')
    masked[i] = (v > -clampv && v < clampv) ? g[i] : 0.f;
  }
}
__global__ void fft_combine_kernel(const float* __restrict__ spec, int C, long nimg, float* __restrict__ dx) {
  // image (b, c): planes of spec for batch b are [Re F(gRe_0..C-1) | Re F(gIm_0..C-1) | Im F(gRe) | Im F(gIm)]
  const long n = nimg * FFT_N * FFT_N;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int m2 = (int)(i % FFT_N), m1 = (int)((i / FFT_N) % FFT_N);
    const long img = i / (FFT_N * FFT_N);
    const long b = img / C, c = img - b * C;
    const int s1 = (m1 + 64) & 127, s2 = (m2 + 64) & 127;
    const long base = b * 4 * C * (long)(FFT_N * FFT_N) + (long)s1 * FFT_N + s2;
    const float re_gre = spec[base + c * (long)(FFT_N * FFT_N)];
    const float im_gim = spec[base + (3 * C + c) * (long)(FFT_N * FFT_N)];
    const float v = re_gre + im_gim;
    dx[i] = ((m1 + m2) & 1) ? -v : v;
  }
}
size_t fft2_backward_workspace_floats(int B, int C) { return (size_t)B * 6 * C * FFT_N * FFT_N; }
int fft2_feature_backward(const float* g, const float* y, float* dx, int B, int C, float clampv, float* ws, size_t wsf,
                          hipStream_t st) {
  if (wsf < fft2_backward_workspace_floats(B, C)) { set_last_error("fft2 backward: workspace too small"); return LSHM_ERR_WORKSPACE; }
  const long plane = (long)FFT_N * FFT_N;
  float* masked = ws;                              // (B, 2C, 128, 128)
  float* spec = ws + (size_t)B * 2 * C * plane;    // (B, 4C, 128, 128)
  const long n = (long)B * 2 * C * plane;
  hipLaunchKernelGGL(fft_mask_kernel, dim3(min(cdiv(n, 256), 8192)), dim3(256), 0, st, g, y, clampv, n, masked);
  int rc = check_launch("fft_mask");
  if (rc) return rc;
  if ((rc = fft2_ortho_shift_cat_clamp(masked, spec, B, 2 * C, 3.0e38f, st))) return rc;
  const long nd = (long)B * C * plane;
  hipLaunchKernelGGL(fft_combine_kernel, dim3(min(cdiv(nd, 256), 8192)), dim3(256), 0, st, spec, C, (long)B * C, dx);
  return check_launch("fft_combine");
}

}  // namespace lshm
