// Device-side patch pipeline of the LOFAR minibatch loader (src/lofar_tools.py:113-193, the
// num_channels == 4 branch): int8 visibilities x per-(baseline, frequency, polarisation) fp32
// scale factors -> zero-padded (>= patch) dynamic spectra -> 50 %-overlapping patch x patch tiles in
// PATCH-MAJOR order (upstream :170-173) -> clamp to +-1e3 -> optional normalisation by the mean and
// unbiased standard deviation of the whole minibatch.  One pass writes the clamped patches and
// per-workgroup (sum, sum of squares) in double; a second tiny kernel finishes mean/std; a third
// pass normalises in place.  The int8 data are uploaded once; every output element is produced
// directly from its source byte (overlap re-reads are L2 hits).
#include "kernels.h"

namespace lshm {

// vis: (nb, ntime, nfreq, 4 pol, 2 re/im) int8; scale: (nb, nfreq, 4); y: (px*py*nb, NC, P, P)
// NC == 4 (upstream :112-124): channel c -> (pol, part): 0:(0,re) 1:(0,im) 2:(3,re) 3:(3,im)
// NC == 8 (upstream :101-111): channel c -> (pol c/2, part c%2), all four polarisations
__global__ __launch_bounds__(256) void patches_kernel(const int8_t* __restrict__ vis,
                                                      const float* __restrict__ scale, int nb, int ntime,
                                                      int nfreq, int P, int px, int py, int NC, float clampv,
                                                      float* __restrict__ y, double* __restrict__ partial) {
  __shared__ double red[16];
  const long total = (long)px * py * nb * NC * P * P;
  double s1 = 0.0, s2 = 0.0;
  const int stride = P / 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int j = (int)(i % P);
    long r = i / P;
    const int ii = (int)(r % P); r /= P;
    const int c = (int)(r % NC); r /= NC;
    const int b = (int)(r % nb);
    const int ck = (int)(r / nb);
    const int ci = ck / py, cj = ck - ci * py;
    const int t = ci * stride + ii, f = cj * stride + j;
    float v = 0.f;
    if (t < ntime && f < nfreq) {
      const int pol = NC == 4 ? (c >> 1) * 3 : (c >> 1), part = c & 1;
      const long src = ((((long)b * ntime + t) * nfreq + f) * 4 + pol) * 2 + part;
      v = (float)vis[src] * scale[((long)b * nfreq + f) * 4 + pol];
    }
    v = fminf(fmaxf(v, -clampv), clampv);
    y[i] = v;
    s1 += (double)v;
    s2 += (double)v * (double)v;
  }
  const double t1 = block_sum<double>(s1, red);
  const double t2 = block_sum<double>(s2, red);
  if (threadIdx.x == 0) { partial[2 * blockIdx.x] = t1; partial[2 * blockIdx.x + 1] = t2; }
}
// moments (optional): [sum, sum of squares, count] of this call's clamped values -- what ranks of a
// data-parallel job add up before they normalise (upstream normalises by the WHOLE minibatch, :190-193)
__global__ __launch_bounds__(256) void patches_moments_kernel(const double* __restrict__ partial, int nblk,
                                                              double n, double* __restrict__ mean_std,
                                                              double* __restrict__ moments) {
  __shared__ double red[16];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nblk; i += blockDim.x) { a += partial[2 * i]; b += partial[2 * i + 1]; }
  const double s1 = block_sum<double>(a, red);
  const double s2 = block_sum<double>(b, red);
  if (threadIdx.x == 0) {
    const double mean = s1 / n;
    double var = (s2 - n * mean * mean) / (n - 1.0);  // torch.std: unbiased
    if (var < 0.0) var = 0.0;
    mean_std[0] = mean;
    mean_std[1] = sqrt(var);
    if (moments) { moments[0] = s1; moments[1] = s2; moments[2] = n; }
  }
}
// y = (y - mean) / std with mean / unbiased std from [sum, sum of squares, count]
__global__ void patches_normalize_moments_kernel(float* __restrict__ y, long n, const double* __restrict__ moments) {
  const double cnt = moments[2], mean = moments[0] / cnt;
  double var = (moments[1] - cnt * mean * mean) / (cnt - 1.0);
  if (var < 0.0) var = 0.0;
  const float m = (float)mean, inv = (float)(1.0 / sqrt(var));
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = (y[i] - m) * inv;
}
__global__ void patches_normalize_kernel(float* __restrict__ y, long n, const double* __restrict__ mean_std) {
  const float m = (float)mean_std[0], inv = (float)(1.0 / mean_std[1]);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = (y[i] - m) * inv;
}

#define PATCH_BLOCKS 1024
size_t patches_workspace_floats() { return 2 * (2 * PATCH_BLOCKS + 2); }

int patches_normalize_moments(float* y, long n, const double* moments, hipStream_t st) {
  const int grid = (int)((n + 255) / 256 < PATCH_BLOCKS ? (n + 255) / 256 : PATCH_BLOCKS);
  hipLaunchKernelGGL(patches_normalize_moments_kernel, dim3(grid > 0 ? grid : 1), dim3(256), 0, st, y, n, moments);
  return check_launch("patches_normalize_moments");
}

int patches_from_vis(const int8_t* vis, const float* scale, int nb, int ntime, int nfreq, int P, int NC, float clampv,
                     int normalize, float* y, double* mean_std, double* moments, float* ws, hipStream_t st) {
  const int T = ntime > P ? ntime : P, F = nfreq > P ? nfreq : P;
  const int px = (T - P) / (P / 2) + 1, py = (F - P) / (P / 2) + 1;
  const long total = (long)px * py * nb * NC * P * P;
  double* partial = reinterpret_cast<double*>(ws);
  const int grid = (int)((total + 255) / 256 < PATCH_BLOCKS ? (total + 255) / 256 : PATCH_BLOCKS);
  hipLaunchKernelGGL(patches_kernel, dim3(grid), dim3(256), 0, st, vis, scale, nb, ntime, nfreq, P, px, py, NC, clampv,
                     y, partial);
  int rc = check_launch("patches");
  if (rc) return rc;
  hipLaunchKernelGGL(patches_moments_kernel, dim3(1), dim3(256), 0, st, partial, grid, (double)total, mean_std, moments);
  if ((rc = check_launch("patches_moments")) || !normalize) return rc;
  hipLaunchKernelGGL(patches_normalize_kernel, dim3(grid), dim3(256), 0, st, y, total, mean_std);
  return check_launch("patches_normalize");
}

}  // namespace lshm
