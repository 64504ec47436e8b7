// Dictionary learning of src/rica_lofar.py:  X ~ A S,  X (L x B) the minibatch of
// vectorised patches, A (L x M) the dictionary, S (M x B) the codes.
//   closure (:72-81):  loss = ||X - A S||^2 / (B L) + lambda1 ||S||_1 / (M B)
//                      (torch.linalg.norm(S, 1) of a matrix: the largest column sum of |S|)
//   update  (:84-93):  E = X - A S;  dA = E S^T / B;  A += eta dA;  logs ||dA||_F
// Everything is held transposed, patch-major as the loader delivers it: Xt = x.view(-1, L) (B x L),
// St = S^T (B x M).  Then (A S)^T = St A^T is a dense layer with weight A (lshm linear_fwd), the code
// gradient E^T-side product is its data gradient and E S^T its weight gradient: three fp32-MFMA GEMMs of
// 2 B L M flop each, the only matrix-core-bound workload of the repository.
#include "common.h"
#include "kernels.h"

namespace lshm {

// E = X - Y in place of Y; per-workgroup partial sums of E^2 (double), fixed order
__global__ __launch_bounds__(256) void rica_residual_kernel(const float* __restrict__ X, float* __restrict__ Y,
                                                            long n4, long n, double* __restrict__ part) {
  __shared__ double red[16];
  double acc = 0.0;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
  const f32x4* X4 = reinterpret_cast<const f32x4*>(X);
  f32x4* Y4 = reinterpret_cast<f32x4*>(Y);
  for (long i = gid; i < n4; i += stride) {
    const f32x4 x = __builtin_nontemporal_load(X4 + i), y = Y4[i];
    const f32x4 e = x - y;
    Y4[i] = e;
    acc += (double)(e[0] * e[0] + e[1] * e[1]) + (double)(e[2] * e[2] + e[3] * e[3]);
  }
  for (long i = 4 * n4 + gid; i < n; i += stride) {  // n % 4 tail
    const float e = X[i] - Y[i];
    Y[i] = e;
    acc += (double)e * e;
  }
  const double tot = block_sum<double>(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

// rowsum[b] = sum_m |St[b, m]|   (= column sums of |S|), one wavefront per row
__global__ __launch_bounds__(256) void rica_rowabs_kernel(const float* __restrict__ St, int B, int M,
                                                          float* __restrict__ rowsum) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B) return;
  float acc = 0.f;
  for (int m = lane; m < M; m += 64) acc += fabsf(St[(long)row * M + m]);
  acc = wave_sum(acc);
  if (lane == 0) rowsum[row] = acc;
}

// one workgroup: SSE from the partials, the largest row sum and its (first) row, the loss
__global__ __launch_bounds__(256) void rica_finish_kernel(const double* __restrict__ part, int nparts,
                                                          const float* __restrict__ rowsum, int B,
                                                          double inv_bl, double l1_scale,
                                                          double* __restrict__ loss, int* __restrict__ argmax) {
  __shared__ double red[16];
  __shared__ float bestv[256];
  __shared__ int besti[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nparts; i += blockDim.x) acc += part[i];
  const double sse = block_sum<double>(acc, red);
  float v = -1.f;
  int bi = 0;
  for (int b = threadIdx.x; b < B; b += blockDim.x)
    if (rowsum[b] > v) { v = rowsum[b]; bi = b; }
  bestv[threadIdx.x] = v;
  besti[threadIdx.x] = bi;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < (int)blockDim.x; ++i)
      if (bestv[i] > v || (bestv[i] == v && besti[i] < bi)) { v = bestv[i]; bi = besti[i]; }
    loss[0] = sse * inv_bl + l1_scale * (double)v;
    loss[1] = sse;
    loss[2] = (double)v;
    argmax[0] = bi;
  }
}

// dSt = c G, plus the subgradient of the matrix 1-norm on its arg-max row
__global__ __launch_bounds__(256) void rica_grad_kernel(const float* __restrict__ G, const float* __restrict__ St,
                                                        const int* __restrict__ argmax, float c, float l1, int B,
                                                        int M, float* __restrict__ dSt) {
  const long n = (long)B * M;
  const int row = argmax[0];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float g = c * G[i];
    if (i / M == row) {
      const float s = St[i];
      g += l1 * (s > 0.f ? 1.f : s < 0.f ? -1.f : 0.f);
    }
    dSt[i] = g;
  }
}

#define RICA_PARTS 1024

static size_t rica_gemm_ws(int B, int L, int M) {
  size_t a = igemm_workspace_floats(B, L, M, 1);
  const size_t b = igemm_workspace_floats(B, M, L, 1), c = igemm_workspace_floats(L, M, B, 1);
  if (b > a) a = b;
  if (c > a) a = c;
  return (a + 16 + 63) & ~(size_t)63;
}
static size_t align64(size_t n) { return (n + 63) & ~(size_t)63; }

// workspace layout (floats, every piece on a 256-byte boundary):
//   E (B L) | G (B M) | dA (L M) | rowsum (B) | partial sums (doubles) + loss scratch | arg-max | GEMM scratch
struct RicaWs {
  float *E, *G, *dA, *rowsum, *gemm;
  double* part;
  int* argmax;
  size_t gemm_floats, total;
};
static RicaWs rica_carve(float* ws, int B, int L, int M) {
  RicaWs w;
  size_t o = 0;
  auto take = [&](size_t n) { float* p = ws ? ws + o : nullptr; o += align64(n); return p; };
  w.E = take((size_t)B * L);
  w.G = take((size_t)B * M);
  w.dA = take((size_t)L * M);
  w.rowsum = take((size_t)B);
  w.part = reinterpret_cast<double*>(take(2 * (RICA_PARTS + 8)));
  w.argmax = reinterpret_cast<int*>(take(16));
  w.gemm_floats = rica_gemm_ws(B, L, M);
  w.gemm = take(w.gemm_floats);
  w.total = o;
  return w;
}
size_t rica_workspace_floats(int B, int L, int M) { return rica_carve(nullptr, B, L, M).total; }

// E = Xt - St A^T into the workspace, sum of squares partials behind it
static int rica_residual(const float* Xt, const float* A, const float* St, int B, int L, int M, const RicaWs& w,
                         hipStream_t st) {
  int rc = linear_fwd(LinFwdIO{St, A, nullptr, w.E}, M, L, B, M, L, 0, w.gemm, w.gemm_floats, st);
  if (rc) return rc;
  const long n = (long)B * L;
  hipLaunchKernelGGL(rica_residual_kernel, dim3(RICA_PARTS), dim3(256), 0, st, Xt, w.E, n >> 2, n, w.part);
  return check_launch("rica_residual");
}

int rica_loss_grad(const float* Xt, const float* A, const float* St, int B, int L, int M, float lambda1,
                   double* loss, float* dSt, float* ws, size_t wsf, hipStream_t st) {
  const RicaWs w = rica_carve(ws, B, L, M);
  if (wsf < w.total) { set_last_error("rica: workspace too small"); return LSHM_ERR_WORKSPACE; }
  int rc = rica_residual(Xt, A, St, B, L, M, w, st);
  if (rc) return rc;
  hipLaunchKernelGGL(rica_rowabs_kernel, dim3(cdiv(B, 4)), dim3(256), 0, st, St, B, M, w.rowsum);
  if ((rc = check_launch("rica_rowabs"))) return rc;
  const double inv_bl = 1.0 / ((double)B * L), l1 = (double)lambda1 / ((double)M * B);
  double* scratch = w.part + RICA_PARTS;  // [loss, sse, norm1]
  hipLaunchKernelGGL(rica_finish_kernel, dim3(1), dim3(256), 0, st, w.part, RICA_PARTS, w.rowsum, B, inv_bl, l1,
                     scratch, w.argmax);
  if ((rc = check_launch("rica_finish"))) return rc;
  if (hipMemcpyAsync(loss, scratch, sizeof(double), hipMemcpyDeviceToDevice, st) != hipSuccess) {
    set_last_error("rica: loss copy failed");
    return LSHM_ERR_ARG;
  }
  if (!dSt) return LSHM_OK;
  // d/dSt of ||Xt - St A^T||^2 / (B L) = -2 E A / (B L)
  rc = linear_dgrad(LinDgradIO{w.E, A, w.G, nullptr, nullptr}, L, M, 0, 0, 0, B, M, L, w.gemm, w.gemm_floats, st);
  if (rc) return rc;
  hipLaunchKernelGGL(rica_grad_kernel, dim3(cdiv((long)B * M, 256 * 4)), dim3(256), 0, st, w.G, St, w.argmax,
                     (float)(-2.0 * inv_bl), (float)l1, B, M, dSt);
  return check_launch("rica_grad");
}

int rica_update_dictionary(const float* Xt, float* A, const float* St, int B, int L, int M, float eta,
                           double* dA_norm_sq, float* ws, size_t wsf, hipStream_t st) {
  const RicaWs w = rica_carve(ws, B, L, M);
  if (wsf < w.total) { set_last_error("rica: workspace too small"); return LSHM_ERR_WORKSPACE; }
  int rc = rica_residual(Xt, A, St, B, L, M, w, st);
  if (rc) return rc;
  // dA B = E S^T = Et^T St: the weight gradient of the dense layer
  rc = linear_wgrad(LinWgradIO{St, w.E, w.dA, nullptr}, M, L, B, M, L, w.gemm, w.gemm_floats, st);
  if (rc) return rc;
  if (dA_norm_sq && (rc = dot_flat(w.dA, w.dA, (long)L * M, dA_norm_sq, reinterpret_cast<float*>(w.part), st)))
    return rc;  // ||E S^T||_F^2; the caller divides by B^2
  return axpy_flat(A, w.dA, eta / (float)B, (long)L * M, st);
}

}  // namespace lshm
