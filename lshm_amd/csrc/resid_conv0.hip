// conv0 of netT AND netF straight from the minibatch and the 2-D reconstruction, for the forward whose activations nobody
// reads afterwards (the no-grad forward that closes an ADMM iteration, src/kharmonic_lofar.py:187-196):
//   r = (x - x1) / 2                                              (src/kharmonic_lofar.py:142-143)
//   netT.conv0(row-vectorised r), netF.conv0(column-vectorised r)   (:144-147, src/lofar_models.py:115: Conv1d(4, 8, 4, stride=4, padding=1) + ELU)
// The closure forward materialises both vectorisations (`residual_split`: read x and x1, write two images; the backward's
// weight gradients read them).  Here neither is written: the row-vectorised sequence IS the image (a thread reads its 4
// inputs per channel from x and x1 directly), the column-vectorised one is read as 128 x 32 image tiles that a workgroup
// holds in LDS (pitch 33) and walks column-wise.  0.27 GB of writes and 0.13 GB of reads less per forward, one launch less.
// The arithmetic is that of residual_split_kernel followed by conv1d_stream_kernel<4, 8, true>, operation for operation
// (same products in the same fused-multiply-add order): results are bitwise the same.
#include <stdlib.h>

#include "kernels.h"

namespace lshm {

namespace {
typedef const __attribute__((address_space(4))) float* cfloat_ptr;
typedef const __attribute__((address_space(4))) f32x4* cf32x4_ptr;
__device__ __forceinline__ f32x4 uload4(const float* q) { return *(cf32x4_ptr)(q); }
__device__ __forceinline__ float uload(const float* q) { return *(cfloat_ptr)(q); }
constexpr int P = 128, CI = 4, CO = 8, L = P * P, LO = L / 4;
constexpr int TC = 32;        // image columns per netF tile
constexpr int PITCH = TC + 1;
}  // namespace

struct ResidConv0Args {
  const float* x;    // (B, 4, 128, 128)
  const float* x1;   // same shape: the 2-D reconstruction
  const float* w[2]; const float* bias[2];  // netT, netF: (8, 4, 4), (8)
  float* y[2];       // (B, 8, 4096) each, batch stride y_bs
  long y_bs;
  int B;
  int nF;            // workgroups of the netF part (they come first: they are the long ones)
};

// one output position: taps 0..3 at sequence positions 4j-1 .. 4j+2; xm[ci] = the element before the quad (0 at j == 0)
template <class T>
__device__ __forceinline__ void conv0_point(const float (&xm)[CI], const float (&v)[CI][3], const float* __restrict__ w,
                                            const float* __restrict__ bias, T* __restrict__ y) {
#pragma unroll
  for (int co = 0; co < CO; ++co) {
    float acc = uload(bias + co);
#pragma unroll
    for (int ci = 0; ci < CI; ++ci) {
      const f32x4 w4 = uload4(w + (co * CI + ci) * 4);
      acc = fmaf(xm[ci], w4[0], acc);
      acc = fmaf(v[ci][0], w4[1], acc);
      acc = fmaf(v[ci][1], w4[2], acc);
      acc = fmaf(v[ci][2], w4[3], acc);
    }
    Elem<T>::st(y + (long)co * LO, elu(acc));
  }
}

// T: element type of x1 and of the outputs (bf16 storage): the residual is then rounded to bf16 as
// residual_split_kernel<bf16> stores it, before it enters the products
template <class T>
__device__ __forceinline__ float resid(float xa, float xb) {
  const float v = (xa - xb) * 0.5f;
  if constexpr (sizeof(T) == 2) return (float)(T)v;
  return v;
}
template <class T>
__global__ __launch_bounds__(256) void resid_conv0_kernel(const ResidConv0Args a) {
  const T* __restrict__ x1 = reinterpret_cast<const T*>(a.x1);
  __shared__ float res[CI * P * PITCH];  // netF: the residual of a 128 x 32 tile of all four channels
  __shared__ float prev[CI];             // ... and the last element of the column before the tile
  const int t = threadIdx.x;
  if ((int)blockIdx.x >= a.nF) {
    // ---- netT: the row-vectorised sequence is the image itself
    const long idx = (long)(blockIdx.x - a.nF) * 256 + t;
    if (idx >= (long)a.B * LO) return;
    const int b = (int)(idx / LO), j = (int)(idx - (long)b * LO);
    const long base = (long)b * CI * L + 4L * j;
    float xm[CI], v[CI][3];
#pragma unroll
    for (int ci = 0; ci < CI; ++ci) {
      const f32x4 xa = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.x + base + (long)ci * L));
      const f32x4 xb = Elem<T>::ld4(x1 + base + (long)ci * L);
#pragma unroll
      for (int k = 0; k < 3; ++k) v[ci][k] = resid<T>(xa[k], xb[k]);
      xm[ci] = j > 0 ? resid<T>(a.x[base + (long)ci * L - 1], Elem<T>::ld(x1 + base + (long)ci * L - 1)) : 0.f;
    }
    conv0_point(xm, v, a.w[0], a.bias[0], reinterpret_cast<T*>(a.y[0]) + (long)b * a.y_bs + j);
    return;
  }
  // ---- netF: sequence position s = 128 c + r (column c, row r); output j = 32 c + g reads rows 4g-1 .. 4g+2 of column c
  // (row -1 = the last row of column c - 1; nothing before j == 0)
  const int b = blockIdx.x / (P / TC), c0 = (blockIdx.x - b * (P / TC)) * TC;
  const float* xb = a.x + (long)b * CI * L;
  const T* x1b = x1 + (long)b * CI * L;
  for (int i = t; i < CI * P * (TC / 4); i += 256) {
    const int c4 = i % (TC / 4), rr = i / (TC / 4);  // rr = ci * 128 + r
    const long g = (long)rr * P + c0 + 4 * c4;
    const f32x4 xa = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xb + g));
    const f32x4 xc = Elem<T>::ld4(x1b + g);
    float* d = &res[rr * PITCH + 4 * c4];
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] = resid<T>(xa[k], xc[k]);
  }
  if (t < CI) {
    const long g = ((long)t * P + (P - 1)) * P + c0 - 1;
    prev[t] = c0 > 0 ? resid<T>(xb[g], Elem<T>::ld(x1b + g)) : 0.f;
  }
  __syncthreads();
  const int g = t & 31, cg = t >> 5;  // 32 row groups x 8 column phases; a thread walks columns cg, cg + 8, ...
#pragma unroll
  for (int k = 0; k < TC / 8; ++k) {
    const int c = cg + 8 * k;
    float xm[CI], v[CI][3];
#pragma unroll
    for (int ci = 0; ci < CI; ++ci) {
      const float* col = &res[(ci * P + 4 * g) * PITCH + c];
      v[ci][0] = col[0]; v[ci][1] = col[PITCH]; v[ci][2] = col[2 * PITCH];
      xm[ci] = g > 0 ? col[-PITCH] : (c > 0 ? res[(ci * P + P - 1) * PITCH + c - 1] : prev[ci]);
    }
    conv0_point(xm, v, a.w[1], a.bias[1], reinterpret_cast<T*>(a.y[1]) + (long)b * a.y_bs + (long)(c0 + c) * (P / 4) + g);
  }
}

bool resid_conv0_supported(int C, int Pp, int Cin, int Cout, int L1d) {
  return !sched(LSHM_SCHED_NO_RESID_CONV0) && C == CI && Pp == P && Cin == CI && Cout == CO && L1d == L;
}

int resid_conv0(const float* x, const float* x1, const float* wT, const float* bT, float* yT, const float* wF, const float* bF,
                float* yF, long y_bs, int B, hipStream_t st, int bf) {
  if (!x || !x1 || !wT || !bT || !yT || !wF || !bF || !yF || B < 1 || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(x1) |
      reinterpret_cast<uintptr_t>(wT) | reinterpret_cast<uintptr_t>(wF)) & 15)) {
    set_last_error("resid_conv0: null or unaligned pointer");
    return LSHM_ERR_ARG;
  }
  ResidConv0Args a;
  a.x = x; a.x1 = x1;
  a.w[0] = wT; a.bias[0] = bT; a.y[0] = yT;
  a.w[1] = wF; a.bias[1] = bF; a.y[1] = yF;
  a.y_bs = y_bs; a.B = B;
  a.nF = B * (P / TC);
  const long nT = ((long)B * LO + 255) / 256;
  int rc = kernel_budget_ok(bf ? reinterpret_cast<const void*>(&resid_conv0_kernel<bf16>) : reinterpret_cast<const void*>(&resid_conv0_kernel<float>),
                            256, 0, "resid_conv0");
  if (rc) return rc;
  if (bf) hipLaunchKernelGGL(resid_conv0_kernel<bf16>, dim3((unsigned)(a.nF + nT)), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(resid_conv0_kernel<float>, dim3((unsigned)(a.nF + nT)), dim3(256), 0, st, a);
  return check_launch("resid_conv0");
}

}  // namespace lshm
