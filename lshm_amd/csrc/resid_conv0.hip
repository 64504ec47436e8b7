// conv0 of netT AND netF straight from the minibatch and the 2-D reconstruction (src/kharmonic_lofar.py:142-147):
//   r = (x - x1) / 2                                              (src/kharmonic_lofar.py:142-143)
//   netT.conv0(row-vectorised r), netF.conv0(column-vectorised r)   (:144-147, src/lofar_models.py:115: Conv1d(4, 8, 4, stride=4, padding=1) + ELU)
// One workgroup holds the residual of a 64 x 64 tile of all four channels in LDS (pitch 65) and serves BOTH layers from it:
// the row-vectorised sequence is the image (16 output positions per tile row), the column-vectorised one is the tile walked
// column-wise (16 output positions per tile column).  x and x1 are read once (the round-3 form read them once per layer:
// 267 MB fetched for 134 MB of input), every store is a 64-byte run, and the four tiles of an image are consecutive
// workgroups of ONE XCD (block index -> (image, tile) below), so the halves of a 128-byte line meet in that XCD's L2 and the
// one-element halos (the column left of the tile for netT, the row above it for netF) are L2 hits.
//   KEEP = 0: the forward whose activations nobody reads afterwards (the no-grad forward that closes an ADMM
//     iteration, :187-196) -- neither vectorisation is written.
//   KEEP = 2: the closure forward -- the two vectorisations the backward's weight gradients read are written from the
//     same tile (`residual_split` + `conv1d_stream<4,8>` read x, x1 and then both residual images again: 495 MB -> 330 MB).
//   KEEP = 1: ... only the row image: conv0's backward as one tile kernel (conv0_bwd_tile.hip) reads nothing else.
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
constexpr int TS = 64;         // tile side
constexpr int PITCH = TS + 1;
constexpr int NT = 512;        // two workgroups per CU (LDS): 16 wavefronts
constexpr int TILES = (P / TS) * (P / TS);
}  // namespace

struct ResidConv0Args {
  const float* x;    // (B, 4, 128, 128)
  const float* x1;   // same shape: the 2-D reconstruction
  const float* w[2]; const float* bias[2];  // netT, netF: (8, 4, 4), (8)
  float* y[2];       // (B, 8, 4096) each, batch stride y_bs
  float* out_row; float* out_col;  // KEEP: the residual as the image and as its per-plane transpose, (B, 4, 128 * 128) each
  long y_bs;
  int B;
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

// T: element type of x1, of the outputs and of the kept residual images (bf16 storage): the residual is then
// rounded to bf16 as residual_split_kernel<bf16> stores it, before it enters the products
template <class T>
__device__ __forceinline__ float resid(float xa, float xb) {
  const float v = (xa - xb) * 0.5f;
  if constexpr (sizeof(T) == 2) return (float)(T)v;
  return v;
}
template <class T, int KEEP>  // 0: nothing kept, 1: the row image, 2: the row image and its per-plane transpose
__global__ __launch_bounds__(NT) void resid_conv0_kernel(const ResidConv0Args a) {
  const T* __restrict__ x1 = reinterpret_cast<const T*>(a.x1);
  __shared__ float res[CI * TS * PITCH];  // the residual of the tile, res[(ci * 64 + r) * 65 + c]
  __shared__ float left[CI * TS];         // netT: the element before each tile row (the previous row's end at column 0)
  __shared__ float top[CI * TS];          // netF: the element above each tile column (the previous column's end at row 0)
  const int t = threadIdx.x;
  // blocks 32 q + 8 k + i  ->  image 8 q + i, tile k: the four tiles of an image follow each other on one XCD
  const int n = blockIdx.x, b = 8 * (n >> 5) + (n & 7), tile = (n >> 3) & 3;
  if (b >= a.B) return;
  const int r0 = (tile >> 1) * TS, c0 = (tile & 1) * TS;
  const float* xb = a.x + (long)b * CI * L;
  const T* x1b = x1 + (long)b * CI * L;
  if (t < CI * TS) {
    const int ci = t >> 6, i = t & (TS - 1);
    {  // row r0 + i: the element at column c0 - 1, or the end of the row above
      const int r = r0 + i;
      const long g = c0 > 0 ? ((long)ci * P + r) * P + c0 - 1 : ((long)ci * P + r - 1) * P + P - 1;
      left[t] = (c0 > 0 || r > 0) ? resid<T>(xb[g], Elem<T>::ld(x1b + g)) : 0.f;
    }
    {  // column c0 + i: the element at row r0 - 1, or the end of the column before
      const int c = c0 + i;
      const long g = r0 > 0 ? ((long)ci * P + r0 - 1) * P + c : ((long)ci * P + P - 1) * P + c - 1;
      top[t] = (r0 > 0 || c > 0) ? resid<T>(xb[g], Elem<T>::ld(x1b + g)) : 0.f;
    }
  }
  T* orow = reinterpret_cast<T*>(a.out_row) + (long)b * CI * L;
#pragma unroll 4
  for (int i = t; i < CI * TS * (TS / 4); i += NT) {
    const int c4 = i & (TS / 4 - 1), rr = i >> 4;  // rr = ci * 64 + r
    const int ci = rr >> 6, r = rr & (TS - 1);
    const long g = ((long)ci * P + r0 + r) * P + c0 + 4 * c4;
    const f32x4 xa = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xb + g));
    const f32x4 xc = Elem<T>::ld4(x1b + g);
    f32x4 v;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = resid<T>(xa[k], xc[k]);
    if constexpr (KEEP > 0) Elem<T>::st4(orow + g, v);
    float* d = &res[rr * PITCH + 4 * c4];
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] = v[k];
  }
  __syncthreads();
  const int q = t & 15, s = t >> 4;  // 16 output positions along the sequence x NT / 16 rows (netT) / columns (netF) per pass
  // ---- netF: sequence position 128 c + r; output 32 c + g reads rows 4g-1 .. 4g+2 of column c
  {
    T* yF = reinterpret_cast<T*>(a.y[1]) + (long)b * a.y_bs + r0 / 4 + q;
#pragma unroll
    for (int k = 0; k < TS / (NT / 16); ++k) {
      const int c = s + (NT / 16) * k;
      float xm[CI], v[CI][3];
#pragma unroll
      for (int ci = 0; ci < CI; ++ci) {
        const float* col = &res[(ci * TS + 4 * q) * PITCH + c];
        v[ci][0] = col[0]; v[ci][1] = col[PITCH]; v[ci][2] = col[2 * PITCH];
        xm[ci] = q > 0 ? col[-PITCH] : top[ci * TS + c];
      }
      conv0_point(xm, v, a.w[1], a.bias[1], yF + (long)(c0 + c) * (P / 4));
    }
  }
  // ---- netT: the row-vectorised sequence is the image itself; output 32 r + p reads columns 4p-1 .. 4p+2 of row r
  {
    T* yT = reinterpret_cast<T*>(a.y[0]) + (long)b * a.y_bs + c0 / 4 + q;
#pragma unroll
    for (int k = 0; k < TS / (NT / 16); ++k) {
      const int r = s + (NT / 16) * k;
      float xm[CI], v[CI][3];
#pragma unroll
      for (int ci = 0; ci < CI; ++ci) {
        const float* row = &res[(ci * TS + r) * PITCH + 4 * q];
        v[ci][0] = row[0]; v[ci][1] = row[1]; v[ci][2] = row[2];
        xm[ci] = q > 0 ? row[-1] : left[ci * TS + r];
      }
      conv0_point(xm, v, a.w[0], a.bias[0], yT + (long)(r0 + r) * (P / 4));
    }
  }
  if constexpr (KEEP == 2) {  // the column-vectorised residual: plane (ci) transposed, 64-element runs along the tile's rows
    T* ocol = reinterpret_cast<T*>(a.out_col) + (long)b * CI * L;
#pragma unroll 4
    for (int i = t; i < CI * TS * (TS / 4); i += NT) {
      const int r4 = i & (TS / 4 - 1), cc = i >> 4;  // cc = ci * 64 + c
      const int ci = cc >> 6, c = cc & (TS - 1);
      const float* src = &res[(ci * TS + 4 * r4) * PITCH + c];
      f32x4 v;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = src[k * PITCH];
      Elem<T>::st4(ocol + ((long)ci * P + c0 + c) * P + r0 + 4 * r4, v);
    }
  }
}

bool resid_conv0_supported(int C, int Pp, int Cin, int Cout, int L1d) {
  return !sched(LSHM_SCHED_NO_RESID_CONV0) && C == CI && Pp == P && Cin == CI && Cout == CO && L1d == L &&
         device_lds_fits(sizeof(float) * (CI * TS * PITCH + 2 * CI * TS));
}

template <class T, int KEEP>
static int launch_resid_conv0(const ResidConv0Args& a, hipStream_t st) {
  int rc = kernel_budget_ok(reinterpret_cast<const void*>(&resid_conv0_kernel<T, KEEP>), NT, 0, "resid_conv0");
  if (rc) return rc;
  const unsigned blocks = (unsigned)((a.B + 7) / 8) * 8 * TILES;
  hipLaunchKernelGGL((resid_conv0_kernel<T, KEEP>), dim3(blocks), dim3(NT), 0, st, a);
  return check_launch("resid_conv0");
}

// out_row / out_col: both null (the no-grad form), both given (the closure form: the residual images are kept), or out_row
// alone (the closure form when the backward of conv0 reads netF's windows from the row image: conv0_bwd_tile.hip)
int resid_conv0(const float* x, const float* x1, const float* wT, const float* bT, float* yT, const float* wF, const float* bF,
                float* yF, long y_bs, int B, hipStream_t st, int bf, float* out_row, float* out_col) {
  if (!x || !x1 || !wT || !bT || !yT || !wF || !bF || !yF || B < 1 || (out_col && !out_row) ||
      ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(x1) | reinterpret_cast<uintptr_t>(wT) | reinterpret_cast<uintptr_t>(wF) |
        reinterpret_cast<uintptr_t>(out_row) | reinterpret_cast<uintptr_t>(out_col)) & 15)) {
    set_last_error("resid_conv0: null or unaligned pointer");
    return LSHM_ERR_ARG;
  }
  ResidConv0Args a;
  a.x = x; a.x1 = x1;
  a.w[0] = wT; a.bias[0] = bT; a.y[0] = yT;
  a.w[1] = wF; a.bias[1] = bF; a.y[1] = yF;
  a.out_row = out_row; a.out_col = out_col;
  a.y_bs = y_bs; a.B = B;
  if (out_col) return bf ? launch_resid_conv0<bf16, 2>(a, st) : launch_resid_conv0<float, 2>(a, st);
  if (out_row) return bf ? launch_resid_conv0<bf16, 1>(a, st) : launch_resid_conv0<float, 1>(a, st);
  return bf ? launch_resid_conv0<bf16, 0>(a, st) : launch_resid_conv0<float, 0>(a, st);
}

}  // namespace lshm
