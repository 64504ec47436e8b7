// Layer-level dispatch: maps the four convolution flavours of the autoencoders
// (src/lofar_models.py:31-57, 115-142) and the dense layers onto the implicit
// GEMM problems of igemm.hip.  A transposed convolution's forward is the
// data-gradient problem of the matching strided convolution and vice versa, so
// three GEMM problems per dimensionality cover forward, dgrad and wgrad of both.
#include "kernels.h"

namespace lshm {

void conv_out_dims(const ConvLayer& L, int& Ho, int& Wo) {
  switch (L.kind) {
    case 0: Ho = L.Hin / 2; Wo = L.Win / 2; break;
    case 1: Ho = L.Hin * 2; Wo = L.Win * 2; break;
    case 2: Ho = 1; Wo = (L.Win - 2) / 4 + 1; break;
    default: Ho = 1; Wo = L.Win * 4; break;
  }
}

static bool transposed(const ConvLayer& L) { return L.kind == 1 || L.kind == 3; }

// split-K plan for the weight gradient: (M, N) = weight matrix, K = B * small spatial size
struct WgradPlan { int M, N; long K; int S, ksplit, Sb; };
static WgradPlan wgrad_plan(const ConvLayer& L) {
  WgradPlan w;
  int Ho, Wo;
  conv_out_dims(L, Ho, Wo);
  const int taps = (L.kind < 2) ? 16 : 4;
  const long small_sp = transposed(L) ? (long)L.Hin * L.Win : (long)Ho * Wo;
  w.M = transposed(L) ? L.Cin : L.Cout;                 // small-tensor channels
  w.N = (transposed(L) ? L.Cout : L.Cin) * taps;        // big-tensor channels x taps
  w.K = (long)L.B * small_sp;
  const long tiles = (long)cdiv(w.M, 64) * cdiv(w.N, 64);
  long want = 1024 / tiles;
  if (want < 1) want = 1;
  long maxs = cdiv(w.K, 256);
  if (maxs < 1) maxs = 1;
  long S = want < maxs ? want : maxs;
  long ks = cdiv(w.K, S);
  ks = (ks + 15) / 16 * 16;
  w.ksplit = (int)ks;
  w.S = cdiv(w.K, ks);
  w.Sb = L.B < 32 ? L.B : 32;
  return w;
}
size_t conv_wgrad_workspace_floats(const ConvLayer& L) {
  const WgradPlan w = wgrad_plan(L);
  return (size_t)w.S * w.M * w.N + (size_t)w.Sb * L.Cout + 16;
}

int conv_layer_fwd(const ConvLayer& L, const float* x, const float* w, const float* b, float* y,
                   int act, hipStream_t st) {
  int Ho, Wo;
  conv_out_dims(L, Ho, Wo);
  switch (L.kind) {
    case 0: {
      Conv2dFwdParams p{x, w, b, y, nullptr, L.B, L.Cin, L.Hin, L.Win, L.Cout, Ho, Wo,
                        L.in_bs, L.out_bs, act, L.B * Ho * Wo, L.Cout, L.Cin * 16};
      return conv2d_fwd(p, st);
    }
    case 1: {
      Conv2dDgradParams p{x, w, b, y, nullptr, L.B, L.Cin, L.Hin, L.Win, L.Cout,
                          L.in_bs, L.out_bs, act, L.B * L.Hin * L.Win, L.Cout, L.Cin * 4};
      return conv2d_dgrad(p, st);
    }
    case 2: {
      Conv1dFwdParams p{x, w, b, y, nullptr, L.B, L.Cin, L.Win, L.Cout, Wo, 1,
                        L.in_bs, L.out_bs, act, L.B * Wo, L.Cout, L.Cin * 4};
      return conv1d_fwd(p, st);
    }
    default: {
      Conv1dDgradParams p{x, w, b, y, nullptr, L.B, L.Cin, L.Win, L.Cout, Wo, 0,
                          L.in_bs, L.out_bs, act, L.B * L.Win, L.Cout * 4, L.Cin};
      return conv1d_dgrad(p, st);
    }
  }
}

int conv_layer_dgrad(const ConvLayer& L, const float* dz, const float* w, float* dx,
                     const float* dact_in, hipStream_t st) {
  int Ho, Wo;
  conv_out_dims(L, Ho, Wo);
  switch (L.kind) {
    case 0: {  // dx (big) from dz (small)
      Conv2dDgradParams p{dz, w, nullptr, dx, dact_in, L.B, L.Cout, Ho, Wo, L.Cin,
                          L.out_bs, L.in_bs, 0, L.B * Ho * Wo, L.Cin, L.Cout * 4};
      return conv2d_dgrad(p, st);
    }
    case 1: {  // dx (small) = strided conv of dz (big) with the same weight tensor
      Conv2dFwdParams p{dz, w, nullptr, dx, dact_in, L.B, L.Cout, Ho, Wo, L.Cin, L.Hin, L.Win,
                        L.out_bs, L.in_bs, 0, L.B * L.Hin * L.Win, L.Cin, L.Cout * 16};
      return conv2d_fwd(p, st);
    }
    case 2: {
      Conv1dDgradParams p{dz, w, nullptr, dx, dact_in, L.B, L.Cout, Wo, L.Cin, L.Win, 1,
                          L.out_bs, L.in_bs, 0, L.B * Wo, L.Cin * 4, L.Cout};
      return conv1d_dgrad(p, st);
    }
    default: {
      Conv1dFwdParams p{dz, w, nullptr, dx, dact_in, L.B, L.Cout, Wo, L.Cin, L.Win, 0,
                        L.out_bs, L.in_bs, 0, L.B * L.Win, L.Cin, L.Cout * 4};
      return conv1d_fwd(p, st);
    }
  }
}

int conv_layer_wgrad(const ConvLayer& L, const float* x, const float* dz, float* dw, float* db,
                     float* ws, size_t ws_floats, int accumulate, hipStream_t st) {
  if (ws_floats < conv_wgrad_workspace_floats(L)) {
    set_last_error("conv wgrad: workspace too small");
    return LSHM_ERR_WORKSPACE;
  }
  int Ho, Wo;
  conv_out_dims(L, Ho, Wo);
  const WgradPlan wp = wgrad_plan(L);
  const bool tr = transposed(L);
  // small / big tensors of the underlying strided-conv geometry
  const float* small = tr ? x : dz;
  const float* big = tr ? dz : x;
  const long s_bs = tr ? L.in_bs : L.out_bs;
  const long big_bs = tr ? L.out_bs : L.in_bs;
  const int Cs = wp.M, Cb = tr ? L.Cout : L.Cin;
  int rc;
  if (L.kind < 2) {
    const int Hs = tr ? L.Hin : Ho, Ws = tr ? L.Win : Wo;
    Conv2dWgradParams p{small, big, ws, L.B, Cs, Hs, Ws, Cb, s_bs, big_bs,
                        wp.M, wp.N, (int)wp.K, wp.ksplit};
    rc = conv2d_wgrad(p, wp.S, st);
  } else {
    const int Ls = tr ? L.Win : Wo, Lb = tr ? Wo : L.Win;
    Conv1dWgradParams p{small, big, ws, L.B, Cs, Ls, Cb, Lb, tr ? 0 : 1, s_bs, big_bs,
                        wp.M, wp.N, (int)wp.K, wp.ksplit};
    rc = conv1d_wgrad(p, wp.S, st);
  }
  if (rc) return rc;
  rc = reduce_partials(ws, dw, (long)wp.M * wp.N, wp.S, accumulate, st);
  if (rc || !db) return rc;
  float* bpart = ws + (size_t)wp.S * wp.M * wp.N;
  rc = channel_sum_partials(dz, L.out_bs, L.B, L.Cout, (long)Ho * Wo, bpart, wp.Sb, st);
  if (rc) return rc;
  return reduce_partials(bpart, db, L.Cout, wp.Sb, accumulate, st);
}

// --------------------------------------------------------------------------
// dense layers
// --------------------------------------------------------------------------
int linear_fwd(const float* x, long ldx, const float* w, const float* b, float* y, long ldy, int B,
               int K, int N, int act, hipStream_t st) {
  StridedGemmParams p{x, w, b, y, nullptr, ldx, 1, 1, K, ldy, 1, 0, 0, act, B, N, K, nullptr, 0, 0};
  return strided_gemm(p, false, false, st);
}
int linear_dgrad(const float* dz, long lddz, const float* w, float* dx, long lddx,
                 const float* xsaved, long ldxs, int B, int K, int N, hipStream_t st,
                 const float* add, long ldadd, int add_n) {
  StridedGemmParams p{dz, w, nullptr, dx, xsaved, lddz, 1, K, 1, lddx, 1, ldxs, 1, 0, B, K, N,
                      add, ldadd, add_n};
  return strided_gemm(p, false, true, st);
}
__global__ void copy2d_kernel(const float* __restrict__ src, long lds_, float* __restrict__ dst,
                              long ldd, int rows, int cols) {
  const long n = (long)rows * cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long r = i / cols, c = i - r * cols;
    dst[r * ldd + c] = src[r * lds_ + c];
  }
}
int copy2d(const float* src, long lds_, float* dst, long ldd, int rows, int cols, hipStream_t st) {
  hipLaunchKernelGGL(copy2d_kernel, dim3(min(cdiv((long)rows * cols, 256), 1024)), dim3(256), 0, st,
                     src, lds_, dst, ldd, rows, cols);
  return check_launch("copy2d");
}
__global__ void colsum_kernel(const float* __restrict__ dz, long lddz, int B, int N,
                              float* __restrict__ db, int accumulate) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float acc = 0.f;
  for (int b = 0; b < B; ++b) acc += dz[(long)b * lddz + n];
  db[n] = accumulate ? db[n] + acc : acc;
}
int linear_wgrad(const float* x, long ldx, const float* dz, long lddz, float* dw, float* db, int B,
                 int K, int N, int accumulate, hipStream_t st) {
  if (accumulate) { set_last_error("linear_wgrad: accumulate not supported"); return LSHM_ERR_UNSUPPORTED; }
  StridedGemmParams p{dz, x, nullptr, dw, nullptr, 1, lddz, ldx, 1, K, 1, 0, 0, 0, N, K, B,
                      nullptr, 0, 0};
  int rc = strided_gemm(p, true, true, st);
  if (rc || !db) return rc;
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(N, 64)), dim3(64), 0, st, dz, lddz, B, N, db, 0);
  return check_launch("colsum");
}

}  // namespace lshm
