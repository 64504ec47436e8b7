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

// GEMM shapes of the three problems of a layer
struct LayerGemm { int M, N, K, Z; };
static LayerGemm fwd_gemm(const ConvLayer& L) {
  int Ho, Wo;
  conv_out_dims(L, Ho, Wo);
  switch (L.kind) {
    case 0: return {L.B * Ho * Wo, L.Cout, L.Cin * 16, 1};
    case 1: return {L.B * L.Hin * L.Win, L.Cout, L.Cin * 4, 4};
    case 2: return {L.B * Wo, L.Cout, L.Cin * 4, 1};
    default: return {L.B * L.Win, L.Cout * 4, L.Cin, 1};
  }
}
static LayerGemm dgrad_gemm(const ConvLayer& L) {
  int Ho, Wo;
  conv_out_dims(L, Ho, Wo);
  switch (L.kind) {
    case 0: return {L.B * Ho * Wo, L.Cin, L.Cout * 4, 4};
    case 1: return {L.B * L.Hin * L.Win, L.Cin, L.Cout * 16, 1};
    case 2: return {L.B * Wo, L.Cin * 4, L.Cout, 1};
    default: return {L.B * L.Win, L.Cin, L.Cout * 4, 1};
  }
}
static LayerGemm wgrad_gemm(const ConvLayer& L) {
  int Ho, Wo;
  conv_out_dims(L, Ho, Wo);
  const int taps = (L.kind < 2) ? 16 : 4;
  const long small_sp = transposed(L) ? (long)L.Hin * L.Win : (long)Ho * Wo;
  return {transposed(L) ? L.Cin : L.Cout, (transposed(L) ? L.Cout : L.Cin) * taps,
          (int)((long)L.B * small_sp), 1};
}

#define BIAS_WS_FLOATS (128 * 256)
static int bias_slices(const ConvLayer& L) {
  int s = 512 / (L.Cout > 0 ? L.Cout : 1);
  if (s < 1) s = 1;
  if (s > 128) s = 128;
  if (s > L.B) s = L.B;
  return s;
}

size_t conv_workspace_floats(const ConvLayer& L) {
  const LayerGemm f = fwd_gemm(L), d = dgrad_gemm(L), w = wgrad_gemm(L);
  size_t a = igemm_workspace_floats(f.M, f.N, f.K, f.Z);
  const size_t b = igemm_workspace_floats(d.M, d.N, d.K, d.Z);
  const size_t c = igemm_workspace_floats(w.M, w.N, w.K, w.Z);
  if (b > a) a = b;
  if (c > a) a = c;
  if (L.kind < 2) {
    const size_t d2 = conv2d_wgrad_direct_workspace_floats(w.M, w.N / 16);
    if (d2 > a) a = d2;
  } else {
    const size_t d1 = conv1d_wgrad_direct_workspace_floats(w.M, w.N / 4);
    if (d1 > a) a = d1;
  }
  return a + BIAS_WS_FLOATS + 16;
}

int conv_layer_fwd(const ConvLayer& L, const float* x, const float* w, const float* b, float* y,
                   int act, float* ws, size_t wsf, hipStream_t st) {
  int Ho, Wo;
  conv_out_dims(L, Ho, Wo);
  switch (L.kind) {
    case 0: {
      if (conv2d_direct_supported(L.Cin, L.Cout, Ho, Wo))
        return conv2d_direct(x, L.in_bs, w, b, y, L.out_bs, nullptr, L.B, L.Cin, L.Cout, Ho, Wo, act, st);
      Conv2dFwdParams p{x, w, b, y, nullptr, L.B, L.Cin, L.Hin, L.Win, L.Cout, Ho, Wo,
                        L.in_bs, L.out_bs, act, L.B * Ho * Wo, L.Cout, L.Cin * 16, {}};
      return conv2d_fwd(p, ws, wsf, st);
    }
    case 1: {
      if (tconv2d_direct_supported(L.Cin, L.Cout, L.Hin, L.Win))
        return tconv2d_direct(x, L.in_bs, w, b, y, L.out_bs, nullptr, L.B, L.Cin, L.Cout, L.Hin, L.Win, act, st);
      Conv2dDgradParams p{x, w, b, y, nullptr, L.B, L.Cin, L.Hin, L.Win, L.Cout,
                          L.in_bs, L.out_bs, act, L.B * L.Hin * L.Win, L.Cout, L.Cin * 4, {}};
      return conv2d_dgrad(p, ws, wsf, st);
    }
    case 2: {
      Conv1dFwdParams p{x, w, b, y, nullptr, L.B, L.Cin, L.Win, L.Cout, Wo, 1,
                        L.in_bs, L.out_bs, act, L.B * Wo, L.Cout, L.Cin * 4, {}};
      return conv1d_fwd(p, ws, wsf, st);
    }
    default: {
      Conv1dDgradParams p{x, w, b, y, nullptr, L.B, L.Cin, L.Win, L.Cout, Wo, 0,
                          L.in_bs, L.out_bs, act, L.B * L.Win, L.Cout * 4, L.Cin, {}};
      return conv1d_dgrad(p, ws, wsf, st);
    }
  }
}

int conv_layer_dgrad(const ConvLayer& L, const float* dz, const float* w, float* dx,
                     const float* dact_in, float* ws, size_t wsf, hipStream_t st) {
  int Ho, Wo;
  conv_out_dims(L, Ho, Wo);
  switch (L.kind) {
    case 0: {  // dx (big) from dz (small)
      if (tconv2d_direct_supported(L.Cout, L.Cin, Ho, Wo))
        return tconv2d_direct(dz, L.out_bs, w, nullptr, dx, L.in_bs, dact_in, L.B, L.Cout, L.Cin, Ho, Wo, 0, st);
      Conv2dDgradParams p{dz, w, nullptr, dx, dact_in, L.B, L.Cout, Ho, Wo, L.Cin,
                          L.out_bs, L.in_bs, 0, L.B * Ho * Wo, L.Cin, L.Cout * 4, {}};
      return conv2d_dgrad(p, ws, wsf, st);
    }
    case 1: {  // dx (small) = strided conv of dz (big) with the same weight tensor
      if (conv2d_direct_supported(L.Cout, L.Cin, L.Hin, L.Win))
        return conv2d_direct(dz, L.out_bs, w, nullptr, dx, L.in_bs, dact_in, L.B, L.Cout, L.Cin, L.Hin, L.Win, 0, st);
      Conv2dFwdParams p{dz, w, nullptr, dx, dact_in, L.B, L.Cout, Ho, Wo, L.Cin, L.Hin, L.Win,
                        L.out_bs, L.in_bs, 0, L.B * L.Hin * L.Win, L.Cin, L.Cout * 16, {}};
      return conv2d_fwd(p, ws, wsf, st);
    }
    case 2: {
      Conv1dDgradParams p{dz, w, nullptr, dx, dact_in, L.B, L.Cout, Wo, L.Cin, L.Win, 1,
                          L.out_bs, L.in_bs, 0, L.B * Wo, L.Cin * 4, L.Cout, {}};
      return conv1d_dgrad(p, ws, wsf, st);
    }
    default: {
      Conv1dFwdParams p{dz, w, nullptr, dx, dact_in, L.B, L.Cout, Wo, L.Cin, L.Win, 0,
                        L.out_bs, L.in_bs, 0, L.B * L.Win, L.Cin, L.Cout * 4, {}};
      return conv1d_fwd(p, ws, wsf, st);
    }
  }
}

int conv_layer_wgrad(const ConvLayer& L, const float* x, const float* dz, float* dw, float* db,
                     float* ws, size_t ws_floats, int accumulate, hipStream_t st) {
  if (!ws || ws_floats < BIAS_WS_FLOATS + 16) {
    set_last_error("conv wgrad: workspace too small");
    return LSHM_ERR_WORKSPACE;
  }
  int Ho, Wo;
  conv_out_dims(L, Ho, Wo);
  const LayerGemm g = wgrad_gemm(L);
  const bool tr = transposed(L);
  // small / big tensors of the underlying strided-conv geometry
  const float* small = tr ? x : dz;
  const float* big = tr ? dz : x;
  const long s_bs = tr ? L.in_bs : L.out_bs;
  const long big_bs = tr ? L.out_bs : L.in_bs;
  const int Cs = g.M, Cb = tr ? L.Cout : L.Cin;
  float* bias_ws = ws;
  float* gemm_ws = ws + BIAS_WS_FLOATS;
  const size_t gemm_wsf = ws_floats - BIAS_WS_FLOATS;
  int rc;
  if (L.kind < 2) {
    const int Hs = tr ? L.Hin : Ho, Ws = tr ? L.Win : Wo;
    if (conv2d_wgrad_direct_supported(Cs, Cb, Hs, Ws) && gemm_wsf >= conv2d_wgrad_direct_workspace_floats(Cs, Cb)) {
      rc = conv2d_wgrad_direct(small, s_bs, big, big_bs, dw, L.B, Cs, Cb, Hs, Ws, gemm_ws, gemm_wsf, accumulate, st);
      goto bias;
    }
    Conv2dWgradParams p{small, big, dw, L.B, Cs, Hs, Ws, Cb, s_bs, big_bs, g.M, g.N, g.K, accumulate, {}};
    rc = conv2d_wgrad(p, gemm_ws, gemm_wsf, st);
  } else {
    const int Ls = tr ? L.Win : Wo, Lb = tr ? Wo : L.Win;
    if (conv1d_wgrad_direct_supported(Cs, Cb, Ls) && gemm_wsf >= conv1d_wgrad_direct_workspace_floats(Cs, Cb))
      // weight and bias gradient in one pass (dz is `big` for the transposed conv, `small` otherwise)
      return conv1d_wgrad_direct(small, s_bs, big, big_bs, dw, db, tr ? 2 : 1, L.Cout, L.B, Cs, Cb, Ls, Lb,
                                 tr ? 0 : 1, gemm_ws, gemm_wsf, accumulate, st);
    Conv1dWgradParams p{small, big, dw, L.B, Cs, Ls, Cb, Lb, tr ? 0 : 1, s_bs, big_bs,
                        g.M, g.N, g.K, accumulate, {}};
    rc = conv1d_wgrad(p, gemm_ws, gemm_wsf, st);
  }
bias:
  if (rc || !db) return rc;
  if ((long)L.B * Ho * Wo <= 65536)  // small tensor: one workgroup per channel, no second stage
    return channel_sum_direct(dz, L.out_bs, L.B, L.Cout, (long)Ho * Wo, db, accumulate, st);
  const int Sb = bias_slices(L);
  rc = channel_sum_partials(dz, L.out_bs, L.B, L.Cout, (long)Ho * Wo, bias_ws, Sb, st);
  if (rc) return rc;
  return reduce_partials(bias_ws, db, L.Cout, Sb, accumulate, st);
}

// --------------------------------------------------------------------------
// dense layers
// --------------------------------------------------------------------------
int linear_fwd(const float* x, long ldx, const float* w, const float* b, float* y, long ldy, int B,
               int K, int N, int act, float* ws, size_t wsf, hipStream_t st) {
  StridedGemmParams p{x, w, b, y, nullptr, ldx, 1, 1, K, ldy, 1, 0, 0, act, B, N, K, {}, nullptr, 0, 0};
  return strided_gemm(p, false, false, ws, wsf, st);
}
int linear_dgrad(const float* dz, long lddz, const float* w, float* dx, long lddx,
                 const float* xsaved, long ldxs, int B, int K, int N, float* ws, size_t wsf,
                 hipStream_t st, const float* add, long ldadd, int add_n) {
  StridedGemmParams p{dz, w, nullptr, dx, xsaved, lddz, 1, K, 1, lddx, 1, ldxs, 1, 0, B, K, N, {},
                      add, ldadd, add_n};
  return strided_gemm(p, false, true, ws, wsf, st);
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
// db[n] = sum_b dz[b, n]: one wave per 64 columns would leave the chip idle for B x 768
// problems, so rows are split over the 4 waves of a block and combined through LDS.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ dz, long lddz, int B, int N,
                                                     float* __restrict__ db) {
  __shared__ float red[256];
  const int nl = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + nl;
  float acc = 0.f;
  if (n < N)
    for (int b = part; b < B; b += 4) acc += dz[(long)b * lddz + n];
  red[threadIdx.x] = acc;
  __syncthreads();
  if (part == 0 && n < N) db[n] = (red[nl] + red[64 + nl]) + (red[128 + nl] + red[192 + nl]);
}
int linear_wgrad(const float* x, long ldx, const float* dz, long lddz, float* dw, float* db, int B,
                 int K, int N, float* ws, size_t wsf, hipStream_t st) {
  StridedGemmParams p{dz, x, nullptr, dw, nullptr, 1, lddz, ldx, 1, K, 1, 0, 0, 0, N, K, B, {},
                      nullptr, 0, 0};
  int rc = strided_gemm(p, true, true, ws, wsf, st);
  if (rc || !db) return rc;
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(N, 64)), dim3(256), 0, st, dz, lddz, B, N, db);
  return check_launch("colsum");
}

}  // namespace lshm
