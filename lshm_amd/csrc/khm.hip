// K-harmonic means: fused all-pairs squared distance + soft-min reduction,
// forward + backward in one pass over X (reference: Kmeans.forward,
// src/lofar_models.py:199-209; closed-form gradient SURVEY.md Appendix A.1),
// Zhang's offline-update partial sums (intent of :231-261) and the evaluation
// distances of src/evaluate_clustering.py:111-115.
//
// Streaming layout: one 64-lane wavefront per sample row, lane l owns columns
// l, l+64, ... (256-byte coalesced segments), centroids staged once per
// workgroup in LDS, squared distances by wavefront shuffle reduction, the K
// per-centroid scalars spread over lanes (lane k owns centroid k), dX written in
// the same pass while the row is still in registers.  The centroid-side sums
// (sum_i w_ik and sum_i w_ik x_i) live in registers per wave, are combined in a
// fixed order through LDS per workgroup and over workgroups by a second kernel:
// bitwise reproducible, no float atomics.
#include "kernels.h"

namespace lshm {

enum { KHM_FWD_BWD = 0, KHM_OFFLINE = 1, KHM_DIST = 2 };

__device__ __forceinline__ float pow_half(float s, float p, int pint) {
  // s^(p/2) for s >= 0
  if (pint == 2) return s;
  if (pint == 4) return s * s;
  if (pint == 6) return s * s * s;
  if (pint == 3) return s * sqrtf(s);
  if (pint == 1) return sqrtf(s);
  return s > 0.f ? powf(s, 0.5f * p) : 0.f;
}
__device__ __forceinline__ float pow_half_m1(float s, float p, int pint) {
  // s^(p/2-1)
  if (pint == 2) return 1.f;
  if (pint == 4) return s;
  if (pint == 6) return s * s;
  if (pint == 3) return sqrtf(s);
  return s > 0.f ? powf(s, 0.5f * p - 1.f) : (p > 2.f ? 0.f : INFINITY);
}

// NC = ceil(D/64) columns per lane, KT = compile-time bound on K (K <= KT <= 64)
template <int MODE, int NC, int KT, int NW>
__global__ __launch_bounds__(NW * 64) void khm_kernel(
    const float* __restrict__ X, long ldx, const float* __restrict__ Mg, int N, int D, int K,
    float p, int pint, float eps, float wscale /* gscale*inv_count*K */, float* __restrict__ dX,
    long lddx, int accumulate_dx, float* __restrict__ partial /* [grid][K*D + K] */,
    double* __restrict__ loss_partial /* [grid] */) {
  extern __shared__ float lds[];
  float* Ms = lds;   // K*D centroids during the row loop
  float* red = lds;  // reused afterwards: NW slabs of (K*D + K) for the block-level combine
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < K * D; i += blockDim.x) Ms[i] = Mg[i];
  __syncthreads();

  float T[KT][NC];  // sum_i w_ik x_i (this lane's columns)
  float S[KT];      // sum_i w_ik (replicated in every lane)
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    S[k] = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) T[k][c] = 0.f;
  }
  double lsum = 0.0;

  const int nwaves = gridDim.x * NW;
  for (int row = blockIdx.x * NW + wave; row < N; row += nwaves) {
    float xr[NC];
    const float* xp = X + (long)row * ldx;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int col = lane + 64 * c;
      xr[c] = col < D ? xp[col] : 0.f;
    }
    // squared distances: lane k ends up owning s_k
    float s_mine = 0.f;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      if (k < K) {
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const int col = lane + 64 * c;
          const float d = col < D ? xr[c] - Ms[k * D + col] : 0.f;
          a = fmaf(d, d, a);
        }
        a = wave_sum(a);
        if (lane == k) s_mine = a;
      }
    }
    const bool act = lane < K;
    const float ph = pow_half(s_mine, p, pint);
    float w_mine;
    if (MODE == KHM_DIST) {
      w_mine = act ? ph : 0.f;
    } else {
      const float g = ph + eps;
      const float inv = act ? 1.f / g : 0.f;
      const float e = wave_sum(inv);
      if (MODE == KHM_FWD_BWD) {
        const float ee = e + eps;
        if (lane == 0) lsum += (double)((float)K / ee);
        // W_ik = c*K/(e+eps)^2 * p * s^(p/2-1) / g^2
        w_mine = act ? wscale / (ee * ee) * p * pow_half_m1(s_mine, p, pint) * inv * inv : 0.f;
      } else {  // Zhang: alpha_i/(s^((p+2)/2)+eps), alpha_i = 1/(e^2+eps)
        const float alpha = 1.f / (e * e + eps);
        w_mine = act ? alpha / (ph * s_mine + eps) : 0.f;
      }
    }
    float dx[NC];
    float wsum = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) dx[c] = 0.f;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      if (k < K) {
        const float wk = __shfl(w_mine, k, 64);
        S[k] += wk;
        if (MODE != KHM_DIST) {
#pragma unroll
          for (int c = 0; c < NC; ++c) T[k][c] = fmaf(wk, xr[c], T[k][c]);
        }
        if (MODE == KHM_FWD_BWD) {
          wsum += wk;
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            const int col = lane + 64 * c;
            if (col < D) dx[c] = fmaf(-wk, Ms[k * D + col], dx[c]);
          }
        }
      }
    }
    if (MODE == KHM_FWD_BWD && dX) {
      float* dp = dX + (long)row * lddx;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int col = lane + 64 * c;
        if (col < D) {
          const float v = fmaf(wsum, xr[c], dx[c]);
          dp[col] = accumulate_dx ? dp[col] + v : v;
        }
      }
    }
  }
  // ---- block-level combine in fixed wave order, then one partial slab per workgroup
  const int slab = K * D + K;
  __syncthreads();
  float* mine = red + (size_t)wave * slab;
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    if (k < K) {
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int col = lane + 64 * c;
        if (col < D) mine[k * D + col] = T[k][c];
      }
      if (lane == 0) mine[K * D + k] = S[k];
    }
  }
  __shared__ double lred[NW];
  if (lane == 0) lred[wave] = lsum;
  __syncthreads();
  float* out = partial + (size_t)blockIdx.x * slab;
  for (int i = threadIdx.x; i < slab; i += blockDim.x) {
    float v = red[i];
#pragma unroll
    for (int w = 1; w < NW; ++w) v += red[(size_t)w * slab + i];
    out[i] = v;
  }
  if (threadIdx.x == 0 && loss_partial) {
    double v = lred[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) v += lred[w];
    loss_partial[blockIdx.x] = v;
  }
}

// second stage.  MODE fwd_bwd: dM[k,:] (+)= (sum S_k) M[k,:] - sum T_k ; loss = sum partial
//               offline  : num = sum T, den = sum S
//               dist     : dist[k] = sum S_k / N
template <int MODE>
__global__ __launch_bounds__(256) void khm_reduce_kernel(const float* __restrict__ partial, int nblk,
                                                         const float* __restrict__ Mg, int N, int D,
                                                         int K, float* __restrict__ out0,
                                                         float* __restrict__ out1,
                                                         const double* __restrict__ loss_partial,
                                                         double* __restrict__ loss_out,
                                                         int accumulate) {
  const int slab = K * D + K;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (MODE == KHM_DIST) {
    if (i < K) {
      float acc = 0.f;
      for (int b = 0; b < nblk; ++b) acc += partial[(size_t)b * slab + K * D + i];
      out0[i] = acc / (float)N;
    }
    return;
  }
  if (i < K * D) {
    const int k = i / D;
    float t = 0.f, s = 0.f;
    for (int b = 0; b < nblk; ++b) {
      t += partial[(size_t)b * slab + i];
      s += partial[(size_t)b * slab + K * D + k];
    }
    if (MODE == KHM_FWD_BWD) {
      const float v = s * Mg[i] - t;
      out0[i] = accumulate ? out0[i] + v : v;
    } else {
      out0[i] = t;
      if (i % D == 0) out1[k] = s;
    }
  }
  if (MODE == KHM_FWD_BWD && blockIdx.x == 0 && loss_out) {
    __shared__ double red[16];
    double acc = 0.0;
    for (int b = threadIdx.x; b < nblk; b += blockDim.x) acc += loss_partial[b];
    const double tot = block_sum<double>(acc, red);
    if (threadIdx.x == 0) loss_out[0] = tot;
  }
}

static int khm_nw(int K) { return K <= 16 ? 4 : 1; }
static int khm_grid(int N, int K) {
  int g = cdiv(N, khm_nw(K));
  const int cap = K <= 16 ? 1024 : 512;
  return g < 1 ? 1 : (g > cap ? cap : g);
}
size_t khm_workspace_floats(int N, int D, int K) {
  const size_t g = (size_t)khm_grid(N, K);
  return g * ((size_t)K * D + K) + 2 * g + 16;
}

template <int MODE, int NC, int KT, int NW>
static int khm_launch(dim3 grid, size_t shmem, hipStream_t st, const float* X, long ldx,
                      const float* M, int N, int D, int K, float p, int pint, float eps,
                      float wscale, float* dX, long lddx, int acc_dx, float* partial,
                      double* lpart) {
  auto kern = khm_kernel<MODE, NC, KT, NW>;
  if (shmem > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) { set_last_error("khm: cannot raise dynamic LDS limit"); return (int)e; }
  }
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), shmem, st, X, ldx, M, N, D, K, p, pint, eps, wscale,
                     dX, lddx, acc_dx, partial, lpart);
  return check_launch("khm");
}

template <int MODE, int NC>
static int khm_launch_kt(dim3 grid, size_t shmem, hipStream_t st, const float* X, long ldx,
                         const float* M, int N, int D, int K, float p, int pint, float eps,
                         float wscale, float* dX, long lddx, int acc_dx, float* partial,
                         double* lpart) {
#define KHM_GO(KTV, NWV)                                                                         \
  return khm_launch<MODE, NC, KTV, NWV>(grid, shmem, st, X, ldx, M, N, D, K, p, pint, eps, wscale, \
                                        dX, lddx, acc_dx, partial, lpart)
  if (K <= 4) KHM_GO(4, 4);
  if (K <= 8) KHM_GO(8, 4);
  if (K <= 12) KHM_GO(12, 4);
  if (K <= 16) KHM_GO(16, 4);
  if (K <= 32) KHM_GO(32, 1);
  KHM_GO(64, 1);
#undef KHM_GO
}

template <int MODE>
static int khm_run(const float* X, long ldx, const float* M, int N, int D, int K, float p, float eps,
                   float wscale, float* dX, long lddx, int acc_dx, float* out0, float* out1,
                   double* loss_out, int accumulate, float* ws, size_t ws_floats, hipStream_t st) {
  if (!X || !M || !ws || N < 0 || D < 1 || K < 1) { set_last_error("khm: bad argument"); return LSHM_ERR_ARG; }
  if (K > 64 || D > 512) {
    set_last_error("khm: supports K <= 64 and latent_dim <= 512");
    return LSHM_ERR_UNSUPPORTED;
  }
  if (ws_floats < khm_workspace_floats(N, D, K)) { set_last_error("khm: workspace too small"); return LSHM_ERR_WORKSPACE; }
  const int g = khm_grid(N, K);
  const size_t slab = (size_t)K * D + K;
  const size_t shmem = (size_t)khm_nw(K) * slab * sizeof(float);
  if (shmem > 150 * 1024) { set_last_error("khm: K*latent_dim too large for LDS"); return LSHM_ERR_UNSUPPORTED; }
  float* partial = ws;
  size_t off = (size_t)g * slab;
  off = (off + 1) & ~(size_t)1;  // keep the double partials 8-byte aligned
  double* lpart = reinterpret_cast<double*>(ws + off);
  const float pr = roundf(p);
  const int pint = (fabsf(p - pr) < 1e-6f) ? (int)pr : 0;
  dim3 grid(g);
  int rc;
  const int nc = cdiv(D, 64);
  if (nc <= 1) rc = khm_launch_kt<MODE, 1>(grid, shmem, st, X, ldx, M, N, D, K, p, pint, eps, wscale, dX, lddx, acc_dx, partial, lpart);
  else if (nc <= 2) rc = khm_launch_kt<MODE, 2>(grid, shmem, st, X, ldx, M, N, D, K, p, pint, eps, wscale, dX, lddx, acc_dx, partial, lpart);
  else if (nc <= 4) rc = khm_launch_kt<MODE, 4>(grid, shmem, st, X, ldx, M, N, D, K, p, pint, eps, wscale, dX, lddx, acc_dx, partial, lpart);
  else rc = khm_launch_kt<MODE, 8>(grid, shmem, st, X, ldx, M, N, D, K, p, pint, eps, wscale, dX, lddx, acc_dx, partial, lpart);
  if (rc) return rc;
  hipLaunchKernelGGL((khm_reduce_kernel<MODE>), dim3(cdiv(K * D, 256)), dim3(256), 0, st, partial, g,
                     M, N, D, K, out0, out1, lpart, loss_out, accumulate);
  return check_launch("khm_reduce");
}

int khm_fwd_bwd(const float* X, long ldx, const float* M, int N, int D, int K, float p, float eps,
                double inv_count, float gscale, double* loss_sum, float* dX, long lddx, float* dM,
                int accumulate_dx, float* ws, size_t ws_floats, hipStream_t st) {
  const float wscale = (float)((double)gscale * inv_count * (double)K);
  return khm_run<KHM_FWD_BWD>(X, ldx, M, N, D, K, p, eps, wscale, dX, lddx, accumulate_dx, dM,
                              nullptr, loss_sum, 0, ws, ws_floats, st);
}
int khm_offline_partials(const float* X, long ldx, const float* M, int N, int D, int K, float p,
                         float eps, float* num, float* den, float* ws, size_t ws_floats,
                         hipStream_t st) {
  return khm_run<KHM_OFFLINE>(X, ldx, M, N, D, K, p, eps, 0.f, nullptr, 0, 0, num, den, nullptr, 0,
                              ws, ws_floats, st);
}
int khm_mean_distances(const float* X, long ldx, const float* M, int N, int D, int K, float p,
                       float* dist, float* ws, size_t ws_floats, hipStream_t st) {
  return khm_run<KHM_DIST>(X, ldx, M, N, D, K, p, 0.f, 0.f, nullptr, 0, 0, dist, nullptr, nullptr,
                           0, ws, ws_floats, st);
}

}  // namespace lshm
