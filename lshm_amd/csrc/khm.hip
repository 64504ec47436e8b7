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
#include <stdlib.h>

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

// --------------------------------------------------------------------------
// K > 16 (config 5: K = 64): RS = ceil(D/64) wavefronts share each row, wavefront w owning columns
// 64w .. 64w+63 (one per lane).  The centroid-side sums then need KT accumulator registers per lane
// instead of KT*RS (64 instead of 256 at K = 64, D = 256): a workgroup no longer fills a SIMD's register
// file, so the decoders running beside it on the other stream keep finding CUs to start on (with one
// wavefront per row, 434 VGPRs, the first decoder GEMM waited 156 us for a 12 us launch).  The partial
// squared distances of the RS column chunks meet in LDS (one barrier per row, double-buffered, added in
// wavefront order); every wavefront then derives the same soft-min weights.  Disjoint columns: the slab
// is written without a block-level combine.
// --------------------------------------------------------------------------
template <int MODE, int KT, int RS>
__global__ __launch_bounds__(RS * 64) void khm_rowsplit_kernel(
    const float* __restrict__ X, long ldx, const float* __restrict__ Mg, int N, int D, int K,
    float p, int pint, float eps, float wscale, float* __restrict__ dX, long lddx, int accumulate_dx,
    float* __restrict__ partial /* [grid][K*D + K] */, double* __restrict__ loss_partial /* [grid] */) {
  extern __shared__ float lds[];
  float* Ms = lds;                    // K*D centroids
  float* exch = lds + (size_t)K * D;  // [2][RS][64] partial squared distances
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = 64 * wave + lane;
  const bool incol = col < D;
  for (int i = threadIdx.x; i < K * D; i += blockDim.x) Ms[i] = Mg[i];
  __syncthreads();

  float T[KT];        // sum_i w_ik x_i[col]
  float S_mine = 0.f; // lane k: sum_i w_ik
#pragma unroll
  for (int k = 0; k < KT; ++k) T[k] = 0.f;
  double lsum = 0.0;
  int buf = 0;
  for (int row = blockIdx.x; row < N; row += gridDim.x) {  // workgroup-uniform bound: barriers are safe
    const float xr = incol ? X[(long)row * ldx + col] : 0.f;
    float part = 0.f;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      if (k < K) {
        const float d = incol ? xr - Ms[k * D + col] : 0.f;
        const float a = wave_sum(d * d);
        if (lane == k) part = a;
      }
    }
    float s_mine = part;
    if (RS > 1) {
      float* e = exch + buf * (RS * 64);
      e[wave * 64 + lane] = part;
      __syncthreads();
      s_mine = e[lane];
#pragma unroll
      for (int w = 1; w < RS; ++w) s_mine += e[w * 64 + lane];
      buf ^= 1;  // the other buffer is free again: its readers passed this iteration's barrier
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
        if (threadIdx.x == 0) lsum += (double)((float)K / ee);
        w_mine = act ? wscale / (ee * ee) * p * pow_half_m1(s_mine, p, pint) * inv * inv : 0.f;
      } else {
        const float alpha = 1.f / (e * e + eps);
        w_mine = act ? alpha / (ph * s_mine + eps) : 0.f;
      }
    }
    S_mine += w_mine;
    float dx = 0.f, wsum = 0.f;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      if (k < K) {
        const float wk = __shfl(w_mine, k, 64);
        if (MODE != KHM_DIST) T[k] = fmaf(wk, xr, T[k]);
        if (MODE == KHM_FWD_BWD) {
          wsum += wk;
          if (incol) dx = fmaf(-wk, Ms[k * D + col], dx);
        }
      }
    }
    if (MODE == KHM_FWD_BWD && dX && incol) {
      float* dp = dX + (long)row * lddx + col;
      const float v = fmaf(wsum, xr, dx);
      *dp = accumulate_dx ? *dp + v : v;
    }
  }
  const int slab = K * D + K;
  float* out = partial + (size_t)blockIdx.x * slab;
#pragma unroll
  for (int k = 0; k < KT; ++k)
    if (k < K && incol) out[k * D + col] = T[k];
  if (wave == 0 && lane < K) out[K * D + lane] = S_mine;
  if (threadIdx.x == 0 && loss_partial) loss_partial[blockIdx.x] = lsum;
}

// --------------------------------------------------------------------------
// Streaming fast path for latent_dim == 256 and K <= 16 (the training configuration):
//   * a wavefront handles 4 rows at a time, 16 lanes per row; lane j of a row owns columns
//     64q + 4j .. 64q + 4j + 3 (q = 0..3): every load/store instruction moves four 256-byte
//     contiguous row segments as float4 per lane;
//   * the K squared distances are reduced over the 16 lanes of a row with DPP row rotations
//     (pure VALU, no LDS crossbar), leaving every lane with all K values, so the soft-min
//     scalars need no further cross-lane traffic;
//   * centroids are read from LDS as ds_read_b128 (the four rows of a wave read the same
//     addresses: broadcast, conflict-free);
//   * the centroid-side sums T = W^T X ride on the matrix cores: one v_mfma_f32_16x16x4_f32 per
//     16-column chunk with A = W (lane (cluster, row)) and B = X (lane (row, column)), exact fp32;
//   * the next 4 rows are prefetched into registers while the current ones are processed.
// --------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL,
                                                               0xF, 0xF, false));
}
// sum over the 16 lanes of a DPP row; every lane receives the total
__device__ __forceinline__ float row16_allsum(float v) {
  v += dpp_mov<0x128>(v);  // row_ror:8
  v += dpp_mov<0x124>(v);  // row_ror:4
  v += dpp_mov<0x122>(v);  // row_ror:2
  v += dpp_mov<0x121>(v);  // row_ror:1
  return v;
}

// PI: compile-time integer exponent (4 = the training configuration) or 0 = runtime p
template <int PI>
__device__ __forceinline__ float ph_t(float s, float p, int pint) { return PI == 4 ? s * s : pow_half(s, p, pint); }
template <int PI>
__device__ __forceinline__ float phm1_t(float s, float p, int pint) { return PI == 4 ? s : pow_half_m1(s, p, pint); }
// v_rcp_f32: 1 ulp, against ~10 instructions for an IEEE divide
__device__ __forceinline__ float fast_rcp(float v) { return __builtin_amdgcn_rcpf(v); }

// 4-wave workgroups, 3 per CU at the kernel's 3 waves/SIMD (12-wave workgroups measured 25% slower)
#define KHM_FAST_WAVES 4
#define KHM_FAST_THREADS (KHM_FAST_WAVES * 64)
#define KHM_FAST_ROWS (KHM_FAST_WAVES * 4)
template <int MODE, int KT, int PI>
__global__ __launch_bounds__(KHM_FAST_THREADS) void khm256_kernel(
    const float* __restrict__ X, long ldx, const float* __restrict__ Mg, int N, int K, float p,
    int pint, float eps, float wscale, float* __restrict__ dX, long lddx, int accumulate_dx,
    float* __restrict__ partial /* [grid][K*256 + K] */, double* __restrict__ loss_partial) {
  constexpr int D = 256;
  __shared__ __attribute__((aligned(16))) float Ms[KT * D];  // centroids; reused for the block combine
  __shared__ float Ssum[16];
  __shared__ double lred[KHM_FAST_WAVES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, rg = lane >> 4;
  for (int i = threadIdx.x; i < KT * D; i += KHM_FAST_THREADS) Ms[i] = i < K * D ? Mg[i] : 0.f;
  __syncthreads();

  f32x4 acc[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float s_acc = 0.f;
  double lsum = 0.0;

  const long stride = (long)gridDim.x * KHM_FAST_ROWS;
  long row = (long)blockIdx.x * KHM_FAST_ROWS + wave * 4 + rg;
  f32x4 xn[4];
  auto load_rows = [&](long r) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      xn[q] = r < N ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(X + r * ldx + 64 * q + 4 * j))
                    : (f32x4){0.f, 0.f, 0.f, 0.f};  // streamed once: keep it out of L2's way
  };
  load_rows(row);
  // the loop bound is wave-uniform: all four row groups of a wave iterate together
  for (long base = (long)blockIdx.x * KHM_FAST_ROWS + wave * 4; base < N; base += stride, row += stride) {
    f32x4 x[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) x[q] = xn[q];
    load_rows(row + stride);
    const bool valid = row < N;
    float s[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      s[k] = 0.f;
      if (k < K) {
        // vector form so the compiler emits packed v_pk_add_f32 / v_pk_fma_f32
        f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 m = *reinterpret_cast<const f32x4*>(&Ms[k * D + 64 * q + 4 * j]);
          const f32x4 d = x[q] - m;
          a4 = __builtin_elementwise_fma(d, d, a4);
        }
        s[k] = row16_allsum((a4[0] + a4[1]) + (a4[2] + a4[3]));
      }
    }
    // soft-min scalars, replicated in the 16 lanes of the row
    float w[KT];
    float wsum = 0.f;
    if (MODE == KHM_DIST) {
#pragma unroll
      for (int k = 0; k < KT; ++k) w[k] = (k < K && valid) ? ph_t<PI>(s[k], p, pint) : 0.f;
    } else {
      float e = 0.f;
      float inv[KT];
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        inv[k] = 0.f;
        if (k < K) {
          inv[k] = fast_rcp(ph_t<PI>(s[k], p, pint) + eps);
          e += inv[k];
        }
      }
      if (MODE == KHM_FWD_BWD) {
        const float ee = e + eps;
        if (j == 0 && valid) lsum += (double)((float)K / ee);
        const float ri = fast_rcp(ee);
        const float basew = valid ? wscale * ri * ri * p : 0.f;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          w[k] = (k < K) ? basew * phm1_t<PI>(s[k], p, pint) * inv[k] * inv[k] : 0.f;
          wsum += w[k];
        }
      } else {
        const float alpha = valid ? fast_rcp(e * e + eps) : 0.f;
#pragma unroll
        for (int k = 0; k < KT; ++k)
          w[k] = (k < K) ? alpha * fast_rcp(ph_t<PI>(s[k], p, pint) * s[k] + eps) : 0.f;
      }
    }
    // lane (cluster j, row rg) of the MFMA A operand
    float w_mine = 0.f;
#pragma unroll
    for (int k = 0; k < KT; ++k) w_mine = (j == k) ? w[k] : w_mine;
    s_acc += w_mine;
    if (MODE != KHM_DIST) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2)
          acc[4 * q + e2] = __builtin_amdgcn_mfma_f32_16x16x4f32(w_mine, x[q][e2], acc[4 * q + e2], 0, 0, 0);
    }
    if (MODE == KHM_FWD_BWD && dX) {
      f32x4 dx[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) dx[q] = x[q] * wsum;
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        if (k < K) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 m = *reinterpret_cast<const f32x4*>(&Ms[k * D + 64 * q + 4 * j]);
            dx[q] -= m * w[k];
          }
        }
      }
      if (valid) {
        float* dp = dX + row * lddx + 4 * j;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4* d4 = reinterpret_cast<f32x4*>(dp + 64 * q);
          if (accumulate_dx) *d4 = *d4 + dx[q];
          else __builtin_nontemporal_store(dx[q], d4);
        }
      }
    }
  }
  // ---- block combine in fixed wave order (deterministic), then one slab per workgroup
  s_acc += __shfl_xor(s_acc, 16, 64);
  s_acc += __shfl_xor(s_acc, 32, 64);
  lsum += __shfl_xor(lsum, 16, 64);
  lsum += __shfl_xor(lsum, 32, 64);
  if (lane == 0) lred[wave] = lsum;
  __syncthreads();  // everyone is done reading the centroids
  for (int wv = 0; wv < KHM_FAST_WAVES; ++wv) {
    if (wave == wv) {
#pragma unroll
      for (int c = 0; c < 16; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = 4 * rg + r;  // cluster
          if (m < KT) {
            const int col = 64 * (c >> 2) + 4 * j + (c & 3);
            float* d = &Ms[m * D + col];
            *d = wv == 0 ? acc[c][r] : *d + acc[c][r];
          }
        }
      if (lane < 16) Ssum[lane] = wv == 0 ? s_acc : Ssum[lane] + s_acc;
    }
    __syncthreads();
  }
  const int slab = K * D + K;
  float* out = partial + (size_t)blockIdx.x * slab;
  for (int i = threadIdx.x; i < K * D; i += KHM_FAST_THREADS) out[i] = Ms[i];
  if (threadIdx.x < K) out[K * D + threadIdx.x] = Ssum[threadIdx.x];
  if (threadIdx.x == 0 && loss_partial) {
    double v = 0.0;
    for (int wv = 0; wv < KHM_FAST_WAVES; ++wv) v += lred[wv];
    loss_partial[blockIdx.x] = v;
  }
}

// --------------------------------------------------------------------------
// 16 < K <= 64, latent_dim == 256 (config 5: K = 64) on the matrix cores.  At K = 64 the all-pairs work is
// ~7 N K D flop for 8 N D bytes (arithmetic intensity ~56 flop/B): compute-bound, and as per-centroid wavefront
// reductions it is also instruction-bound (the row-split form above: 45 ms at N = 2^20).  Here the three K x D x N
// products are v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulation):
//   S^T = M X^T            squared distances as  s_ik = |x_i|^2 + |m_k|^2 - 2 S_ik   (clamped at 0)
//   (W M)^T = M^T W^T      dX_i = (sum_k w_ik) x_i - (W M)_i
//   T = W^T X              dM_k = (sum_i w_ik) m_k - T_k          (finished by khm_reduce_kernel)
// The expansion of the squared distance cancels where x_i is within rounding of m_k: its absolute error is
// ~1e-7 (|x|^2 + |m|^2), against the exact (x - m)^2 form of the other kernels -- stated in the tests that compare
// this path (relative tolerance 2e-5 on well-separated data, as the others; a latent that coincides with a
// centroid to 1e-3 relative loses the p-th power's leading digits).
// A workgroup = 4 wavefronts walks tiles of 64 rows, 16 per wavefront.  Register layouts (lane = 16 lk + lm):
//   X   : xv[e][t] = X[row lm][16 e + 4 lk + t]                   (float4 loads, 64 B per row per instruction)
//   S^T : acc[j][r] = S[row lm][centroid 16 j + 4 lk + r]           -> a lane owns 16 centroids of ITS row: the
//         soft-min sums over k are in-lane adds plus two shuffles, and the same registers are the B operand
//         (k index = centroid) of the second product with no data movement
//   dX  : C[r] = (W M)[row lm][16 dt + 4 lk + r]                    -> same positions as xv[dt][r]: float4 stores
//   T   : per wavefront w the columns 64 w .. 64 w + 63 of all K rows, K dimension = the 64 rows of the tile,
//         operands from LDS copies of the tile (rows must run along lk there).
// LDS: the centroids in fragment order (one conflict-free ds_read_b128 per A fragment of the first product, 64 KB),
// the X tile (68 KB), the W tile (20 KB): 152 KB, one workgroup per CU.
// --------------------------------------------------------------------------
#define KHM_MM_XP 272  // X tile row pitch (floats): == 16 (mod 32), rows 4 s + lk land on distinct banks
#define KHM_MM_WP 80   // W tile row pitch
template <int MODE>
__global__ __launch_bounds__(256) void khm_mfma_kernel(
    const float* __restrict__ X, long ldx, const float* __restrict__ Mg, int N, int K, float p, int pint, float eps,
    float wscale, float* __restrict__ dX, long lddx, int accumulate_dx, float* __restrict__ partial /* [grid][K*256 + K] */,
    double* __restrict__ loss_partial /* [grid] */) {
  constexpr int D = 256;
  extern __shared__ float lds[];
  float* MA = lds;                       // [4 j][16 e][64 lanes][4]: M[16 j + lm][16 e + 4 lk + t]
  float* XT = MA + 4 * 16 * 64 * 4;      // [64 rows][KHM_MM_XP]
  float* WT = XT + 64 * KHM_MM_XP;       // [64 rows][KHM_MM_WP]
  float* mn = WT + 64 * KHM_MM_WP;       // [64] |m_k|^2
  float* sred = mn + 64;                 // [4 waves][64] closing sums
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lm = lane & 15, lk = lane >> 4;
  // ---- centroids -> fragment order (rows >= K are zero), |m_k|^2
  for (int i = t; i < 4 * 16 * 64; i += 256) {
    const int ln = i & 63, e = (i >> 6) & 15, j = i >> 10;
    const int c = 16 * j + (ln & 15), col = 16 * e + 4 * (ln >> 4);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (c < K) v = *reinterpret_cast<const f32x4*>(Mg + (long)c * D + col);
    *reinterpret_cast<f32x4*>(MA + 4 * i) = v;
  }
  if (t < 64) {
    float a = 0.f;
    if (t < K)
      for (int d = 0; d < D; ++d) { const float v = Mg[(long)t * D + d]; a = fmaf(v, v, a); }
    mn[t] = a;
  }
  __syncthreads();

  f32x4 T[4][4];     // T[j][n][r] = sum_i w[i][16 j + 4 lk + r] * X[i][64 wave + 16 n + lm]
  float Ssum[4][4];  // this lane's row only: sum over tiles of w[row][16 j + 4 lk + r]
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int n = 0; n < 4; ++n) { T[j][n] = (f32x4){0.f, 0.f, 0.f, 0.f}; Ssum[j][n] = 0.f; }
  double lsum = 0.0;
  const int ntiles = (N + 63) / 64;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int row = tile * 64 + 16 * wave + lm;
    const bool rok = row < N;
    const float* xr = X + (long)(rok ? row : 0) * ldx + 4 * lk;
    f32x4 xv[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) xv[e] = rok ? *reinterpret_cast<const f32x4*>(xr + 16 * e) : (f32x4){0.f, 0.f, 0.f, 0.f};
    // ---- S^T = M X^T
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float x2 = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
#pragma unroll
      for (int q = 0; q < 4; ++q) x2 = fmaf(xv[e][q], xv[e][q], x2);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(MA + ((j * 16 + e) * 64 + lane) * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], xv[e][q], acc[j], 0, 0, 0);
      }
    }
    x2 += __shfl_xor(x2, 16, 64);
    x2 += __shfl_xor(x2, 32, 64);
    // ---- soft-min weights of this lane's 16 centroids
    float w[4][4], ph[4][4], s2[4][4];
    float einv = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = 16 * j + 4 * lk + r;
        const bool ok = rok && c < K;
        const float sq = fmaxf(x2 + mn[c] - 2.f * acc[j][r], 0.f);
        s2[j][r] = sq;
        ph[j][r] = pow_half(sq, p, pint);
        if (MODE != KHM_DIST) einv += ok ? 1.f / (ph[j][r] + eps) : 0.f;
      }
    float wsum = 0.f;
    if (MODE == KHM_DIST) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) w[j][r] = (rok && 16 * j + 4 * lk + r < K) ? ph[j][r] : 0.f;
    } else {
      einv += __shfl_xor(einv, 16, 64);
      einv += __shfl_xor(einv, 32, 64);
      const float ee = einv + eps;
      if (MODE == KHM_FWD_BWD && rok && lk == 0) lsum += (double)((float)K / ee);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ok = rok && 16 * j + 4 * lk + r < K;
          const float inv = 1.f / (ph[j][r] + eps);
          float wv;
          if (MODE == KHM_FWD_BWD) wv = wscale / (ee * ee) * p * pow_half_m1(s2[j][r], p, pint) * inv * inv;
          else wv = (1.f / (einv * einv + eps)) / (ph[j][r] * s2[j][r] + eps);
          w[j][r] = ok ? wv : 0.f;
          wsum += w[j][r];
        }
      wsum += __shfl_xor(wsum, 16, 64);
      wsum += __shfl_xor(wsum, 32, 64);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) Ssum[j][r] += w[j][r];
    if (MODE == KHM_DIST) continue;  // (no barriers on this path)
    // ---- dX = wsum x - W M   (A = M^T gathered from the fragment image, B = the w registers)
    if (MODE == KHM_FWD_BWD && dX) {
      float* dr = dX + (long)(rok ? row : 0) * lddx + 4 * lk;
#pragma unroll
      for (int dt = 0; dt < 16; ++dt) {  // fully unrolled: xv[dt] must stay in registers
        f32x4 c4 = {0.f, 0.f, 0.f, 0.f};
        // element M[16 j + 4 lk + r][16 dt + lm] of the fragment image (j, e = dt): lane' = 16 (lm / 4) + 4 lk + r, t = lm % 4
        const float* mb = MA + (dt * 64 + (lm >> 2) * 16 + 4 * lk) * 4 + (lm & 3);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(mb[(j * 16 * 64 + r) * 4], w[j][r], c4, 0, 0, 0);
        if (rok) {
          f32x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = fmaf(wsum, xv[dt][r], -c4[r]);
          f32x4* dp = reinterpret_cast<f32x4*>(dr + 16 * dt);
          *dp = accumulate_dx ? *dp + o : o;
        }
      }
    }
    // ---- T += W^T X over the 64 rows of the tile: rows must run along lk, so both go through LDS
    __syncthreads();  // the previous tile's readers are done
    {
      float* xw = XT + (16 * wave + lm) * KHM_MM_XP + 4 * lk;
#pragma unroll
      for (int e = 0; e < 16; ++e) *reinterpret_cast<f32x4*>(xw + 16 * e) = xv[e];
      float* ww = WT + (16 * wave + lm) * KHM_MM_WP + 4 * lk;
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(ww + 16 * j) = (f32x4){w[j][0], w[j][1], w[j][2], w[j][3]};
    }
    __syncthreads();
#pragma unroll 4
    for (int s = 0; s < 16; ++s) {
      const float* wrow = WT + (4 * s + lk) * KHM_MM_WP + lm;
      const float* xrow = XT + (4 * s + lk) * KHM_MM_XP + 64 * wave + lm;
      float a[4], b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = wrow[16 * j];
#pragma unroll
      for (int n = 0; n < 4; ++n) b[n] = xrow[16 * n];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int n = 0; n < 4; ++n) T[j][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[n], T[j][n], 0, 0, 0);
    }
  }
  // ---- slab of this workgroup: T (K x 256), S_k = sum_i w_ik (K), loss partial
  const int slab = K * D + K;
  float* out = partial + (size_t)blockIdx.x * slab;
  if (MODE != KHM_DIST) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 16 * j + 4 * lk + r;
          if (c < K) out[c * D + 64 * wave + 16 * n + lm] = T[j][n][r];
        }
  }
  // S_k: sum over the 16 rows of a wavefront (DPP row = the 16 lanes that share lk), then over the 4 wavefronts
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = row16_allsum(Ssum[j][r]);
      if (lm == 0) sred[wave * 64 + 16 * j + 4 * lk + r] = v;
    }
  __syncthreads();
  if (t < K) out[K * D + t] = (sred[t] + sred[64 + t]) + (sred[128 + t] + sred[192 + t]);
  if (loss_partial) {
    // rows are spread over the lk == 0 lanes of all wavefronts: fixed-order sum
    __shared__ double lred[4];
    const double tot = wave_sum_d(lsum);
    if (lane == 0) lred[wave] = tot;
    __syncthreads();
    if (t == 0) loss_partial[blockIdx.x] = (lred[0] + lred[1]) + (lred[2] + lred[3]);
  }
}
static size_t khm_mfma_lds_bytes() { return (size_t)(4 * 16 * 64 * 4 + 64 * KHM_MM_XP + 64 * KHM_MM_WP + 64 + 256) * sizeof(float); }
static bool khm_mfma_ok(int D, int K, long ldx, long lddx, const float* X, const float* dX, const float* M) {
  const bool off = sched(LSHM_SCHED_NO_KHM_MFMA);
  // 152 KB of LDS per workgroup: only where the device has it (gfx950: 160 KB); elsewhere the row-split kernel runs
  const int lds = device_lds_bytes();
  if (lds > 0 && khm_mfma_lds_bytes() > (size_t)lds) return false;
  return !off && D == 256 && K > 16 && K <= 64 && (ldx % 4) == 0 && (lddx % 4) == 0 &&
         ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(dX) | reinterpret_cast<uintptr_t>(M)) & 15) == 0;
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
  // block = 16 outputs x 16 slab lanes; lanes are combined through LDS in lane order
  __shared__ float redt[256], reds[256];
  const int slab = K * D + K;
  const int ol = threadIdx.x & 15, sl = threadIdx.x >> 4;
  if (MODE == KHM_DIST) {
    const int i = blockIdx.x * 16 + ol;
    float acc = 0.f;
    if (i < K)
      for (int b = sl; b < nblk; b += 16) acc += partial[(size_t)b * slab + K * D + i];
    reds[threadIdx.x] = acc;
    __syncthreads();
    if (sl == 0 && i < K) {
      for (int q = 1; q < 16; ++q) acc += reds[q * 16 + ol];
      out0[i] = acc / (float)N;
    }
    return;
  }
  const int i = blockIdx.x * 16 + ol;
  const int k = i < K * D ? i / D : 0;
  float t = 0.f, s = 0.f;
  if (i < K * D)
    for (int b = sl; b < nblk; b += 16) {
      t += partial[(size_t)b * slab + i];
      s += partial[(size_t)b * slab + K * D + k];
    }
  redt[threadIdx.x] = t;
  reds[threadIdx.x] = s;
  __syncthreads();
  if (sl == 0 && i < K * D) {
    for (int q = 1; q < 16; ++q) {
      t += redt[q * 16 + ol];
      s += reds[q * 16 + ol];
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
static bool khm_fast(int D, int K, long ldx, long lddx) {
  return D == 256 && K <= 16 && (ldx % 4) == 0 && (lddx % 4) == 0;
}
static int khm_grid(int N, int D, int K) {
  if (D == 256 && K > 16) {  // matrix-core path: 64 rows per workgroup pass, one workgroup per CU (also bounds the
    int g = cdiv(N, 64);     // row-split fallback's grid: the workspace is sized from this number)
    return g < 1 ? 1 : (g > 256 ? 256 : g);
  }
  if (D == 256 && K <= 16) {  // fast path: 16 rows per workgroup pass, persistent grid
    int g = cdiv(N, 16);
    return g < 1 ? 1 : (g > 768 ? 768 : g);
  }
  int g = cdiv(N, khm_nw(K));
  const int cap = K <= 16 ? 1024 : 512;
  return g < 1 ? 1 : (g > cap ? cap : g);
}
size_t khm_workspace_floats(int N, int D, int K) {
  const size_t g = (size_t)khm_grid(N, D, K);
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
    if (e != hipSuccess) { (void)hipGetLastError(); set_last_error("khm: cannot raise dynamic LDS limit"); return LSHM_ERR_UNSUPPORTED; }
  }
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), shmem, st, X, ldx, M, N, D, K, p, pint, eps, wscale,
                     dX, lddx, acc_dx, partial, lpart);
  return check_launch("khm");
}

template <int MODE, int KT, int RS>
static int khm_launch_rowsplit(dim3 grid, hipStream_t st, const float* X, long ldx, const float* M, int N, int D,
                               int K, float p, int pint, float eps, float wscale, float* dX, long lddx,
                               int acc_dx, float* partial, double* lpart) {
  const size_t shmem = ((size_t)K * D + 2 * RS * 64) * sizeof(float);
  auto kern = khm_rowsplit_kernel<MODE, KT, RS>;
  if (shmem > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) { (void)hipGetLastError(); set_last_error("khm: cannot raise dynamic LDS limit"); return LSHM_ERR_UNSUPPORTED; }
  }
  hipLaunchKernelGGL(kern, grid, dim3(RS * 64), shmem, st, X, ldx, M, N, D, K, p, pint, eps, wscale, dX, lddx,
                     acc_dx, partial, lpart);
  return check_launch("khm_rowsplit");
}
template <int MODE, int RS>
static int khm_launch_rowsplit_kt(dim3 grid, hipStream_t st, const float* X, long ldx, const float* M, int N,
                                  int D, int K, float p, int pint, float eps, float wscale, float* dX,
                                  long lddx, int acc_dx, float* partial, double* lpart) {
  if (K <= 32)
    return khm_launch_rowsplit<MODE, 32, RS>(grid, st, X, ldx, M, N, D, K, p, pint, eps, wscale, dX, lddx, acc_dx, partial, lpart);
  return khm_launch_rowsplit<MODE, 64, RS>(grid, st, X, ldx, M, N, D, K, p, pint, eps, wscale, dX, lddx, acc_dx, partial, lpart);
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
  KHM_GO(16, 4);  // K > 16 takes the row-split kernel (khm_run)
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
  const int g = khm_grid(N, D, K);
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
  const bool aligned = ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(dX)) & 15) == 0;
  if (khm_fast(D, K, ldx, dX ? lddx : 0) && aligned) {
#define KHM_FAST(KTV)                                                                              \
  do {                                                                                             \
    if (pint == 4)                                                                                 \
      hipLaunchKernelGGL((khm256_kernel<MODE, KTV, 4>), grid, dim3(KHM_FAST_THREADS), 0, st, X, ldx, M, N, K, p, \
                         pint, eps, wscale, dX, lddx, acc_dx, partial, lpart);                     \
    else                                                                                           \
      hipLaunchKernelGGL((khm256_kernel<MODE, KTV, 0>), grid, dim3(KHM_FAST_THREADS), 0, st, X, ldx, M, N, K, p, \
                         pint, eps, wscale, dX, lddx, acc_dx, partial, lpart);                     \
  } while (0)
    if (K <= 4) KHM_FAST(4);
    else if (K <= 8) KHM_FAST(8);
    else if (K <= 12) KHM_FAST(12);
    else KHM_FAST(16);
#undef KHM_FAST
    rc = check_launch("khm256");
  } else if (khm_mfma_ok(D, K, ldx, dX ? lddx : 0, X, dX, M)) {
    auto kern = khm_mfma_kernel<MODE>;
    const size_t lds_bytes = khm_mfma_lds_bytes();
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds_bytes);
    if (e != hipSuccess) { (void)hipGetLastError(); set_last_error("khm: cannot raise dynamic LDS limit"); return LSHM_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, st, X, ldx, M, N, K, p, pint, eps, wscale, dX, lddx, acc_dx,
                       partial, lpart);
    rc = check_launch("khm_mfma");
  } else if (K > 16) {  // row-split form: RS wavefronts per row
#define KHM_RS(RSV) rc = khm_launch_rowsplit_kt<MODE, RSV>(grid, st, X, ldx, M, N, D, K, p, pint, eps, wscale, dX, lddx, acc_dx, partial, lpart)
    if (nc <= 1) KHM_RS(1);
    else if (nc <= 2) KHM_RS(2);
    else if (nc <= 4) KHM_RS(4);
    else KHM_RS(8);
#undef KHM_RS
  } else if (nc <= 1) rc = khm_launch_kt<MODE, 1>(grid, shmem, st, X, ldx, M, N, D, K, p, pint, eps, wscale, dX, lddx, acc_dx, partial, lpart);
  else if (nc <= 2) rc = khm_launch_kt<MODE, 2>(grid, shmem, st, X, ldx, M, N, D, K, p, pint, eps, wscale, dX, lddx, acc_dx, partial, lpart);
  else if (nc <= 4) rc = khm_launch_kt<MODE, 4>(grid, shmem, st, X, ldx, M, N, D, K, p, pint, eps, wscale, dX, lddx, acc_dx, partial, lpart);
  else rc = khm_launch_kt<MODE, 8>(grid, shmem, st, X, ldx, M, N, D, K, p, pint, eps, wscale, dX, lddx, acc_dx, partial, lpart);
  if (rc) return rc;
  hipLaunchKernelGGL((khm_reduce_kernel<MODE>), dim3(cdiv(K * D, 16)), dim3(256), 0, st, partial, g,
                     M, N, D, K, out0, out1, lpart, loss_out, accumulate);
  return check_launch("khm_reduce");
}

int khm_fwd_bwd(const float* X, long ldx, const float* M, int N, int D, int K, float p, float eps,
                double inv_count, float gscale, double* loss_sum, float* dX, long lddx, float* dM,
                int accumulate_dx, float* ws, size_t ws_floats, hipStream_t st, int accumulate_dm) {
  const float wscale = (float)((double)gscale * inv_count * (double)K);
  return khm_run<KHM_FWD_BWD>(X, ldx, M, N, D, K, p, eps, wscale, dX, lddx, accumulate_dx, dM,
                              nullptr, loss_sum, accumulate_dm, ws, ws_floats, st);
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
