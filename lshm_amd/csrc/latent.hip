// Small latent-space penalties, forward + backward in one launch each:
//  * cluster_similarity  (Kmeans.cluster_similarity, src/lofar_models.py:214-229)
//  * augmented_loss      (src/kharmonic_lofar.py:97-110)
// Both are Gram-matrix problems of a few dozen rows: one workgroup per Gram,
// rows staged in LDS, dot products by wavefront shuffle reduction.
#include "kernels.h"

namespace lshm {

// loss = gscale/(K*D) * sum_i [sum_{j!=i} E_ij] / (E_ii + eps),  E_ij = exp(G_ij/(n_i n_j + eps))
// dM = C M with the K x K coefficient matrix derived in DESIGN.md (section "cluster similarity").
__global__ __launch_bounds__(1024) void cluster_sim_kernel(const float* __restrict__ Mg, int K, int D,
                                                          float eps, float gscale,
                                                          double* __restrict__ loss,
                                                          float* __restrict__ dM, int accumulate) {
  extern __shared__ float lds[];
  float* G = lds;           // K*K
  float* C = G + K * K;     // K*K
  float* nrm = C + K * K;   // K
  float* a = nrm + K;       // K : 1/(E_ii+eps)
  float* num = a + K;       // K
  float* M = num + K;       // K*D centroids (read K times each below: keep them on chip)
  __shared__ double red[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int i = threadIdx.x; i < K * D; i += blockDim.x) M[i] = Mg[i];
  __syncthreads();
  // Gram matrix (upper triangle incl. diagonal), one wave per pair
  const int npairs = K * (K + 1) / 2;
  for (int pr = wave; pr < npairs; pr += nw) {
    // unrank pair index -> (i <= j): rows 0..i-1 hold i K - i (i-1)/2 pairs
    int i = (int)(((2 * K + 1) - sqrtf((float)((2 * K + 1) * (2 * K + 1) - 8 * pr))) * 0.5f);
    while (i > 0 && i * K - i * (i - 1) / 2 > pr) --i;
    while ((i + 1) * K - (i + 1) * i / 2 <= pr) ++i;
    const int j = i + (pr - (i * K - i * (i - 1) / 2));
    float acc = 0.f;
    for (int c = lane; c < D; c += 64) acc = fmaf(M[(long)i * D + c], M[(long)j * D + c], acc);
    acc = wave_sum(acc);
    if (lane == 0) { G[i * K + j] = acc; G[j * K + i] = acc; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < K; i += blockDim.x) nrm[i] = sqrtf(G[i * K + i]);
  __syncthreads();
  // E_ij in place of C (temporarily), a_i, num_i
  for (int idx = threadIdx.x; idx < K * K; idx += blockDim.x) {
    const int i = idx / K, j = idx - i * K;
    C[idx] = expf(G[idx] / (nrm[i] * nrm[j] + eps));
  }
  __syncthreads();
  for (int i = threadIdx.x; i < K; i += blockDim.x) {
    float s = 0.f;
    for (int j = 0; j < K; ++j)
      if (j != i) s += C[i * K + j];
    num[i] = s;
    a[i] = 1.f / (C[i * K + i] + eps);
  }
  __syncthreads();
  const float c = gscale / ((float)K * (float)D);
  {
    double part = 0.0;
    for (int i = threadIdx.x; i < K; i += blockDim.x) part += (double)(num[i] * a[i]);
    const double tot = block_sum<double>(part, red);
    if (threadIdx.x == 0 && loss) loss[0] = tot * (double)c;
  }
  if (!dM) return;
  // off-diagonal coefficients beta_ij/t_ij (overwrite E with the coefficient, keep E in a register)
  // C_ij = c (a_i + a_j) E_ij / t_ij ; diagonal handled afterwards from row sums
  // need E for the diagonal term: stash E_ii first
  __shared__ float eii[64];
  for (int i = threadIdx.x; i < K; i += blockDim.x) eii[i] = C[i * K + i];
  __syncthreads();
  // diagonal: gamma_i - sum_{j!=i} beta_ij G_ij n_j / (t_ij^2 n_i)
  for (int i = threadIdx.x; i < K; i += blockDim.x) {
    const float n2 = nrm[i] * nrm[i] + eps;
    float diag = -c * num[i] * a[i] * a[i] * eii[i] * 2.f * eps / (n2 * n2);
    for (int j = 0; j < K; ++j) {
      if (j == i) continue;
      const float tij = nrm[i] * nrm[j] + eps;
      const float beta = c * (a[i] + a[j]) * C[i * K + j];
      diag -= beta * G[i * K + j] * nrm[j] / (tij * tij * nrm[i]);
    }
    num[i] = diag;  // reuse
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < K * K; idx += blockDim.x) {
    const int i = idx / K, j = idx - i * K;
    if (i == j) continue;
    const float tij = nrm[i] * nrm[j] + eps;
    C[idx] = c * (a[i] + a[j]) * C[idx] / tij;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < K; i += blockDim.x) C[i * K + i] = num[i];
  __syncthreads();
  for (int idx = threadIdx.x; idx < K * D; idx += blockDim.x) {
    const int i = idx / D, col = idx - i * D;
    float acc = 0.f;
    for (int j = 0; j < K; ++j) acc = fmaf(C[i * K + j], M[(long)j * D + col], acc);
    dM[idx] = accumulate ? dM[idx] + acc : acc;
  }
}

int cluster_sim_fwd_bwd(const float* M, int K, int D, float eps, float gscale, double* loss,
                        float* dM, int accumulate, hipStream_t st) {
  if (!M || K < 1 || D < 1) { set_last_error("cluster_sim: bad argument"); return LSHM_ERR_ARG; }
  if (K > 64) { set_last_error("cluster_sim: supports K <= 64"); return LSHM_ERR_UNSUPPORTED; }
  const size_t shmem = ((size_t)2 * K * K + 3 * K + (size_t)K * D) * sizeof(float);
  if (shmem > 150 * 1024) { set_last_error("cluster_sim: K*latent_dim too large for LDS"); return LSHM_ERR_UNSUPPORTED; }
  if (shmem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cluster_sim_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) { set_last_error("cluster_sim: cannot raise the dynamic LDS limit"); return (int)e; }
  }
  hipLaunchKernelGGL(cluster_sim_kernel, dim3(1), dim3(1024), shmem, st, M, K, D, eps, gscale, loss,
                     dM, accumulate);
  return check_launch("cluster_sim");
}

// --------------------------------------------------------------------------
// augmented loss: per group g of `bpb` consecutive rows
//   L_g = (1/bpb) sum_{i<j} exp(-zh_i . zh_j),  zh = z/(|z|+1e-6);   loss = gscale * sum_g L_g/(bs*bpb)
// one workgroup per group; rows >= bs*bpb are ignored (their gradient is zero).
// --------------------------------------------------------------------------
#define AUG_MAX_ROWS 32
__global__ __launch_bounds__(256) void aug_loss_kernel(const float* __restrict__ Z, long ldz, int rows,
                                                       int D, int bpb, float coef /* gscale/(bs*bpb*bpb) */,
                                                       double* __restrict__ gpart, float* __restrict__ dZ,
                                                       long lddz, int accumulate) {
  extern __shared__ float lds[];
  float* zs = lds;                       // nr * D  (normalised rows)
  float* P = zs + (size_t)bpb * D;       // bpb*bpb : -coef*exp(-zh_i.zh_j), zero diagonal
  float* nrm = P + bpb * bpb;            // bpb
  float* dotgz = nrm + bpb;              // bpb : dzh_i . z_i
  __shared__ double red[16];
  const int g = blockIdx.x;
  const int r0 = g * bpb;
  const int nr = min(bpb, rows - r0);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int i = wave; i < nr; i += nw) {
    float acc = 0.f;
    for (int c = lane; c < D; c += 64) {
      const float v = Z[(long)(r0 + i) * ldz + c];
      zs[i * D + c] = v;
      acc = fmaf(v, v, acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) nrm[i] = sqrtf(acc);
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < nr * D; idx += blockDim.x) zs[idx] /= (nrm[idx / D] + 1e-6f);
  __syncthreads();
  double lpart = 0.0;
  for (int pr = wave; pr < nr * nr; pr += nw) {
    const int i = pr / nr, j = pr - i * nr;
    if (j <= i) { if (lane == 0 && j == i) P[i * bpb + i] = 0.f; continue; }
    float acc = 0.f;
    for (int c = lane; c < D; c += 64) acc = fmaf(zs[i * D + c], zs[j * D + c], acc);
    acc = wave_sum(acc);
    if (lane == 0) {
      const float e = expf(-acc);
      lpart += (double)e;
      P[i * bpb + j] = -coef * e;
      P[j * bpb + i] = -coef * e;
    }
  }
  {
    const double tot = block_sum<double>(lpart, red);
    if (threadIdx.x == 0 && gpart) gpart[g] = tot * (double)coef;
  }
  if (!dZ) return;
  __syncthreads();
  // dzh_i = sum_j P_ij zh_j ;  dz_i = dzh_i/(n_i+d) - (dzh_i . z_i) z_i / ((n_i+d)^2 n_i),  z_i = zh_i (n_i+d)
  // => dz_i = [dzh_i - (dzh_i . zh_i) zh_i (n_i+d)/n_i ... ] / (n_i+d)   (kept in the explicit form below)
  for (int i = wave; i < nr; i += nw) {
    float acc = 0.f;
    for (int c = lane; c < D; c += 64) {
      float gzh = 0.f;
      for (int j = 0; j < nr; ++j) gzh = fmaf(P[i * bpb + j], zs[j * D + c], gzh);
      acc = fmaf(gzh, zs[i * D + c], acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) dotgz[i] = acc;  // dzh_i . zh_i
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < nr * D; idx += blockDim.x) {
    const int i = idx / D, c = idx - i * D;
    float gzh = 0.f;
    for (int j = 0; j < nr; ++j) gzh = fmaf(P[i * bpb + j], zs[j * D + c], gzh);
    const float nd = nrm[i] + 1e-6f;
    // z_i = zh_i*nd ; (dzh.z_i) z_i /(nd^2 n_i) = (dzh.zh_i) zh_i / n_i
    const float corr = nrm[i] > 0.f ? dotgz[i] * zs[idx] / nrm[i] : 0.f;
    const float v = gzh / nd - corr;
    float* d = dZ + (long)(r0 + i) * lddz + c;
    *d = accumulate ? *d + v : v;
  }
}
__global__ void sum_doubles_kernel(const double* __restrict__ in, int n, double* __restrict__ out) {
  __shared__ double red[16];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) acc += in[i];
  const double tot = block_sum<double>(acc, red);
  if (threadIdx.x == 0) out[0] = tot;
}
__global__ void zero_rows_kernel(float* __restrict__ d, long ld, int r0, int r1, int D) {
  const long n = (long)(r1 - r0) * D;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    d[(r0 + i / D) * ld + i % D] = 0.f;
}

// `loss` doubles as scratch: loss[1..groups] hold the per-group partials (caller provides
// at least 1 + ceil(rows/bpb) doubles).
int aug_loss_fwd_bwd(const float* Z, long ldz, int rows, int D, int bpb, int batch_size,
                     float gscale, double* loss, float* dZ, long lddz, int accumulate,
                     hipStream_t st) {
  if (!Z || !loss || rows < 0 || bpb < 1 || batch_size < 1) { set_last_error("aug_loss: bad argument"); return LSHM_ERR_ARG; }
  if (bpb > AUG_MAX_ROWS) { set_last_error("aug_loss: supports at most 32 patches per baseline"); return LSHM_ERR_UNSUPPORTED; }
  const int used = min(rows, batch_size * bpb);
  const int groups = cdiv(used, bpb);
  const size_t shmem = ((size_t)bpb * D + bpb * bpb + 2 * bpb) * sizeof(float);
  if (shmem > 60 * 1024) { set_last_error("aug_loss: latent_dim too large"); return LSHM_ERR_UNSUPPORTED; }
  const float coef = gscale / ((float)batch_size * (float)bpb * (float)bpb);
  if (dZ && !accumulate && used < rows) {
    hipLaunchKernelGGL(zero_rows_kernel, dim3(cdiv((long)(rows - used) * D, 256)), dim3(256), 0, st, dZ,
                       lddz, used, rows, D);
  }
  if (groups > 0) {
    hipLaunchKernelGGL(aug_loss_kernel, dim3(groups), dim3(256), shmem, st, Z, ldz, used, D, bpb, coef,
                       loss + 1, dZ, lddz, accumulate);
    int rc = check_launch("aug_loss");
    if (rc) return rc;
  }
  hipLaunchKernelGGL(sum_doubles_kernel, dim3(1), dim3(64), 0, st, loss + 1, groups, loss);
  return check_launch("aug_loss_sum");
}

// Epilogues of the per-baseline distance vector (the downstream consumers of Kmeans' distances):
//   argmin[0] = index of the smallest dist (first one on ties: torch.min, src/evaluate_clustering.py:116-119)
//   prob[k]   = softmax_k(-dist[k] / mean(dist))   (src/train_graph_stat.py:206-210)
// K <= 64: one wavefront, lane k holds dist[k].
__global__ __launch_bounds__(64) void dist_epilogue_kernel(const float* __restrict__ dist, int K, int* __restrict__ argmin,
                                                           float* __restrict__ prob) {
  const int k = threadIdx.x;
  const float d = k < K ? dist[k] : 0.f;
  const float mean = wave_sum(d) / (float)K;
  // arg-min with the lowest index winning ties; a NaN distance never wins (as torch.min would propagate NaN,
  // the caller sees it in dist itself)
  float bv = k < K ? d : __builtin_inff();
  int bi = k < K ? k : 0x7fffffff;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float ov = __shfl_xor(bv, off, 64);
    const int oi = __shfl_xor(bi, off, 64);
    if (ov < bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if (argmin && k == 0) argmin[0] = bi;
  if (prob) {
    const float z = k < K ? -d / mean : -__builtin_inff();
    float zmax = z;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) zmax = fmaxf(zmax, __shfl_xor(zmax, off, 64));
    const float e = k < K ? expf(z - zmax) : 0.f;
    const float tot = wave_sum(e);
    if (k < K) prob[k] = e / tot;
  }
}
int dist_epilogue(const float* dist, int K, int* argmin, float* prob, hipStream_t st) {
  if (!dist || K < 1 || K > 64) { set_last_error("dist_epilogue: K must be 1..64"); return LSHM_ERR_ARG; }
  hipLaunchKernelGGL(dist_epilogue_kernel, dim3(1), dim3(64), 0, st, dist, K, argmin, prob);
  return check_launch("dist_epilogue");
}

}  // namespace lshm
