// Memory-bound glue of the training step: harmonic features, residual/transposes,
// the fused reconstruction losses + their gradients, multiplier update, Adam and
// flat-vector algebra.  Reference call sites: src/lofar_models.py:60-62 and
// src/kharmonic_lofar.py:137-158,187-202,92.  All kernels are coalesced, float4
// where the layout allows, transposes go through padded LDS tiles, and every
// reduction is deterministic (fixed-order partials, no float atomics).
#include "kernels.h"

namespace lshm {

// --------------------------------------------------------------------------
// uv harmonics: out[b, 2h+c] = sin(scales[h]*uv[b,c]); out[b, 2H+2h+c] = cos(..)
// --------------------------------------------------------------------------
__global__ void uv_harmonics_kernel(const float* __restrict__ uv, const float* __restrict__ sc,
                                    int H, int B, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int w = 2 * H;
  if (i >= B * w) return;
  const int b = i / w, j = i - b * w;
  const float a = sc[j >> 1] * uv[2 * b + (j & 1)];
  out[(long)b * 2 * w + j] = sinf(a);
  out[(long)b * 2 * w + w + j] = cosf(a);
}
int uv_harmonics(const float* uv, const float* scales, int H, int B, float* out, hipStream_t st) {
  const int n = B * 2 * H;
  hipLaunchKernelGGL(uv_harmonics_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, uv, scales, H, B, out);
  return check_launch("uv_harmonics");
}

// same, with the (<= 8) scales passed by value: no device copy of the scales is needed, which
// keeps the engine's launch sequence free of host->device copies (graph capture friendly)
struct ScalesArg { float s[8]; };
__global__ void uv_harmonics_val_kernel(const float* __restrict__ uv, ScalesArg sc, int H, int B,
                                        float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int w = 2 * H;
  if (i >= B * w) return;
  const int b = i / w, j = i - b * w;
  const float a = sc.s[j >> 1] * uv[2 * b + (j & 1)];
  out[(long)b * 2 * w + j] = sinf(a);
  out[(long)b * 2 * w + w + j] = cosf(a);
}
int uv_harmonics_host_scales(const float* uv, const float* scales_host, int H, int B, float* out,
                             hipStream_t st) {
  if (H > 8) { set_last_error("uv_harmonics: at most 8 scales"); return LSHM_ERR_UNSUPPORTED; }
  ScalesArg sc;
  for (int i = 0; i < 8; ++i) sc.s[i] = i < H ? scales_host[i] : 0.f;
  const int n = B * 2 * H;
  hipLaunchKernelGGL(uv_harmonics_val_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, uv, sc, H, B, out);
  return check_launch("uv_harmonics");
}

// Harmonic features and every layer that sees nothing else: uvh (B, 4H) as above, then for each listed
// dense layer out[b, n] = elu(bias[n] + sum_k w[n, k] uvh[b, k]) written straight into its slot of the
// autoencoder's concatenation buffers (fcuv1 / fcuv3 of the three networks: src/lofar_models.py:80,89,
// :165,174).  fmaf chain over k in ascending order, bias added last: bit-for-bit the implicit-GEMM result.
__global__ __launch_bounds__(256) void uv_features_kernel(const float* __restrict__ uv, ScalesArg sc, int H, int B,
                                                          float* __restrict__ uvh_out, UvLayers layers) {
  constexpr int R = 8;  // rows per workgroup
  __shared__ float f[R][32];
  const int hd = 4 * H, w = 2 * H;
  const int r0 = blockIdx.x * R;
  for (int i = threadIdx.x; i < R * w; i += blockDim.x) {
    const int r = i / w, j = i - r * w, b = r0 + r;
    if (b < B) {
      const float a = sc.s[j >> 1] * uv[2 * b + (j & 1)];
      const float sn = sinf(a), cs = cosf(a);
      f[r][j] = sn;
      f[r][w + j] = cs;
      uvh_out[(long)b * hd + j] = sn;
      uvh_out[(long)b * hd + w + j] = cs;
    }
  }
  __syncthreads();
  const int per = R * hd;
  for (int i = threadIdx.x; i < layers.n * per; i += blockDim.x) {
    const int l = i / per, rem = i - l * per;
    const int r = rem / hd, n = rem - r * hd, b = r0 + r;
    if (b >= B) continue;
    const float* wrow = layers.w[l] + (long)n * hd;
    float acc = 0.f;
    for (int k = 0; k < hd; ++k) acc = fmaf(f[r][k], wrow[k], acc);
    layers.out[l][(long)b * layers.ld[l] + n] = elu(acc + layers.bias[l][n]);
  }
}
int uv_features(const float* uv, const float* scales_host, int H, int B, float* uvh, const UvLayers& layers,
                hipStream_t st) {
  if (H > 8) { set_last_error("uv_features: at most 8 scales"); return LSHM_ERR_UNSUPPORTED; }
  ScalesArg sc;
  for (int i = 0; i < 8; ++i) sc.s[i] = i < H ? scales_host[i] : 0.f;
  hipLaunchKernelGGL(uv_features_kernel, dim3(cdiv(B, 8)), dim3(256), 0, st, uv, sc, H, B, uvh, layers);
  return check_launch("uv_features");
}

__global__ void elu_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                               float* __restrict__ dz, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dz[i] = gy[i] * elu_grad_from_out(y[i]);
}
int elu_bwd(const float* gy, const float* y, float* dz, long n, hipStream_t st) {
  hipLaunchKernelGGL(elu_bwd_kernel, dim3(min(cdiv(n, 256), 4096)), dim3(256), 0, st, gy, y, dz, n);
  return check_launch("elu_bwd");
}

// --------------------------------------------------------------------------
// 32x32 tile helpers: block (32,8); thread handles rows ty, ty+8, ty+16, ty+24
// --------------------------------------------------------------------------
#define TILE 32
struct TileIdx {
  long row_off[4];  // offsets of (h0+ty+8i, w0+tx) in the row-major plane
  long col_off[4];  // offsets of (w0+ty+8i, h0+tx) in the transposed plane
};
__device__ __forceinline__ TileIdx tile_idx_at(int P, int bx, int by) {
  TileIdx t;
  const long plane = (long)blockIdx.z * P * P;
  const int h0 = by * TILE, w0 = bx * TILE;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    t.row_off[i] = plane + (long)(h0 + threadIdx.y + 8 * i) * P + w0 + threadIdx.x;
    t.col_off[i] = plane + (long)(w0 + threadIdx.y + 8 * i) * P + h0 + threadIdx.x;
  }
  return t;
}
__device__ __forceinline__ TileIdx tile_idx(int P) { return tile_idx_at(P, blockIdx.x, blockIdx.y); }

// T: element type of x1 and of the two outputs (x itself is the caller's fp32 minibatch)
template <class T>
__global__ __launch_bounds__(256) void residual_split_kernel(const float* __restrict__ x,
                                                             const T* __restrict__ x1,
                                                             T* __restrict__ out_row,
                                                             T* __restrict__ out_col, int P) {
  __shared__ float tile[TILE][TILE + 1];
  const TileIdx t = tile_idx(P);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float v = (__builtin_nontemporal_load(x + t.row_off[i]) - Elem<T>::ld(x1 + t.row_off[i])) * 0.5f;
    if (out_row) Elem<T>::st(out_row + t.row_off[i], v);
    tile[threadIdx.y + 8 * i][threadIdx.x] = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) Elem<T>::st(out_col + t.col_off[i], tile[threadIdx.x][threadIdx.y + 8 * i]);
}
int residual_split(const float* x, const float* x1, float* out_row, float* out_col, int planes,
                   int P, hipStream_t st, int bf) {
  if (P % TILE) { set_last_error("residual_split: patch size must be a multiple of 32"); return LSHM_ERR_ARG; }
  if (bf)
    hipLaunchKernelGGL(residual_split_kernel<bf16>, dim3(P / TILE, P / TILE, planes), dim3(TILE, 8), 0, st, x,
                       reinterpret_cast<const bf16*>(x1), reinterpret_cast<bf16*>(out_row), reinterpret_cast<bf16*>(out_col), P);
  else
    hipLaunchKernelGGL(residual_split_kernel<float>, dim3(P / TILE, P / TILE, planes), dim3(TILE, 8), 0, st,
                       x, x1, out_row, out_col, P);
  return check_launch("residual_split");
}

template <class T>
__global__ __launch_bounds__(256) void plane_transpose_kernel(const T* __restrict__ in,
                                                              float* __restrict__ out, int P) {
  __shared__ float tile[TILE][TILE + 1];
  const TileIdx t = tile_idx(P);
#pragma unroll
  for (int i = 0; i < 4; ++i) tile[threadIdx.y + 8 * i][threadIdx.x] = Elem<T>::ld(in + t.row_off[i]);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) out[t.col_off[i]] = tile[threadIdx.x][threadIdx.y + 8 * i];
}
int plane_transpose(const float* in, float* out, int planes, int P, hipStream_t st, int in_bf) {
  if (P % TILE) { set_last_error("plane_transpose: size must be a multiple of 32"); return LSHM_ERR_ARG; }
  if (in_bf)
    hipLaunchKernelGGL(plane_transpose_kernel<bf16>, dim3(P / TILE, P / TILE, planes), dim3(TILE, 8), 0, st,
                       reinterpret_cast<const bf16*>(in), out, P);
  else
    hipLaunchKernelGGL(plane_transpose_kernel<float>, dim3(P / TILE, P / TILE, planes), dim3(TILE, 8), 0, st, in, out, P);
  return check_launch("plane_transpose");
}
// out (fp32) = in (bf16), n elements: the reconstructions handed back by lshm_engine_encode
__global__ void widen_kernel(const bf16* __restrict__ in, float* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = (float)in[i];
}
int widen_bf16(const float* in_bf16, float* out, long n, hipStream_t st) {
  hipLaunchKernelGGL(widen_kernel, dim3(min(cdiv(n, 256), 4096)), dim3(256), 0, st, reinterpret_cast<const bf16*>(in_bf16), out, n);
  return check_launch("widen_bf16");
}

// --------------------------------------------------------------------------
// reduce_partials / channel sums
// --------------------------------------------------------------------------
// same with an explicit slab stride (slabs that carry more than one output array) and an optional
// second (partial, out) pair handled by blockIdx.y == 1
__global__ __launch_bounds__(256) void reduce_partials_strided_kernel(const float* __restrict__ partial0,
                                                                      const float* __restrict__ partial1,
                                                                      long stride, float* __restrict__ out0,
                                                                      float* __restrict__ out1, long n, int S,
                                                                      int accumulate) {
  __shared__ float red[256];
  const float* partial = blockIdx.y ? partial1 : partial0;
  float* out = blockIdx.y ? out1 : out0;
  const int ol = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const long i = (long)blockIdx.x * 16 + ol;
  float acc = 0.f;
  if (i < n)
    for (int s = sl; s < S; s += 16) acc += partial[(long)s * stride + i];
  red[threadIdx.x] = acc;
  __syncthreads();
  if (sl == 0 && i < n) {
    for (int s = 1; s < 16; ++s) acc += red[s * 16 + ol];
    out[i] = accumulate ? out[i] + acc : acc;
  }
}
int reduce_partials_strided(const float* partial, long stride, float* out, long n, int S, int accumulate,
                            hipStream_t st, const float* partial2, float* out2) {
  hipLaunchKernelGGL(reduce_partials_strided_kernel, dim3(cdiv(n, 16), partial2 ? 2 : 1), dim3(256), 0, st,
                     partial, partial2, stride, out, out2, n, S, accumulate);
  return check_launch("reduce_partials");
}
int reduce_partials(const float* partial, float* out, long n, int S, int accumulate,
                    hipStream_t st, const float* partial2, float* out2) {
  return reduce_partials_strided(partial, n, out, n, S, accumulate, st, partial2, out2);
}

// partial[s*C + c] = sum over images b in slice s, all HW positions, of dz[b*bs + c*HW + r]
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ dz0,
                                                          const float* __restrict__ dz1, long bs, int B,
                                                          int C, long HW, float* __restrict__ partial0,
                                                          float* __restrict__ partial1, int S, int accumulate) {
  __shared__ float red[16];
  const float* dz = blockIdx.z ? dz1 : dz0;
  float* partial = blockIdx.z ? partial1 : partial0;
  const int c = blockIdx.x, s = blockIdx.y;
  const int b0 = (int)((long)B * s / S), b1 = (int)((long)B * (s + 1) / S);
  float acc = 0.f;
  // one flat index over (image, position) so small feature maps still use every lane
  if ((HW & 3) == 0) {
    const long hw4 = HW >> 2, n4 = (long)(b1 - b0) * hw4;
    for (long i = threadIdx.x; i < n4; i += blockDim.x) {
      const long bi = i / hw4, r = i - bi * hw4;
      const f32x4 v = reinterpret_cast<const f32x4*>(dz + (b0 + bi) * bs + (long)c * HW)[r];
      acc += (v[0] + v[1]) + (v[2] + v[3]);
    }
  } else {
    const long n = (long)(b1 - b0) * HW;
    for (long i = threadIdx.x; i < n; i += blockDim.x) {
      const long bi = i / HW, r = i - bi * HW;
      acc += dz[(b0 + bi) * bs + (long)c * HW + r];
    }
  }
  const float tot = block_sum<float>(acc, red);
  if (threadIdx.x == 0) {
    float* d = partial + (long)s * C + c;
    *d = accumulate ? *d + tot : tot;
  }
}
int channel_sum_partials(const float* dz, long bs, int B, int C, long HW, float* partial, int S,
                         hipStream_t st, const float* dz2, float* partial2) {
  hipLaunchKernelGGL(channel_sum_kernel, dim3(C, S, dz2 ? 2 : 1), dim3(HW >= 1024 ? 256 : 64), 0, st, dz, dz2,
                     bs, B, C, HW, partial, partial2, S, 0);
  return check_launch("channel_sum");
}
// single-pass variant for small tensors: db[c] (=|+=) sum_{b,r} dz[b,c,r], one workgroup per channel
int channel_sum_direct(const float* dz, long bs, int B, int C, long HW, float* db, int accumulate,
                       hipStream_t st, const float* dz2, float* db2) {
  hipLaunchKernelGGL(channel_sum_kernel, dim3(C, 1, dz2 ? 2 : 1), dim3(256), 0, st, dz, dz2, bs, B, C, HW, db,
                     db2, 1, accumulate);
  return check_launch("channel_sum");
}

// --------------------------------------------------------------------------
// reconstruction losses (src/kharmonic_lofar.py:150-158) and their gradients.
//   r1 = x-x1, h = r1/2, r2 = h-x2, r3 = h-x3, e = x1+x2+x3-x       (x3 = x3c^T per plane)
//   sums = [sum e^2, y1.r1, sum r1^2, y2.r2, sum r2^2, y3.r3, sum r3^2]
//   gx2  = (2e - y2 - rho r2)/n           gx3c = ((2e - y3 - rho r3)/n)^T
//   gx1p = (2e - y1 - rho r1)/n - (y2 + rho r2 + y3 + rho r3)/(2n)
// --------------------------------------------------------------------------
// UPD: first advance the multipliers, y_k += rho r_k (src/kharmonic_lofar.py:200-202), write them back, and
// evaluate everything else with the new values: the multiplier update of one ADMM iteration and the
// reconstruction terms of the next one read the same seven arrays, so they can share one pass.
// GRAD = false: only the seven sums (the gradient-free closures of a line search).
// T: element type of the three reconstructions and of the three gradient images (x and the multipliers are fp32)
// FROMA: the reconstructions of netT / netF are not read but formed here from the inputs of their last layer
// (ReconFromA: x2 = netT.tconv5(aT), x3c = netF.tconv5(aF), ConvTranspose1d(8, C, 4, stride=4), src/lofar_models.py:142, no
// activation): every image element is one tap of one position, eight multiply-adds over cached values -- in the order of
// tconv1d_stream_kernel, so the values are bitwise the ones that kernel would have written.  The forward that feeds this
// pass then stops one layer early: 0.13 GB less written, 0.2 GB less read per iteration.
struct ReconFromA {
  const float* aT; const float* aF;                                    // (B, 8, P*P/4), batch stride a_bs
  const float* wT; const float* bT; const float* wF; const float* bF;  // (8, C, 4), (C)
  long a_bs;
  int C;                                                               // image channels = planes per sample
};
template <bool UPD, bool GRAD = true, class T = float, bool FROMA = false>
__global__ __launch_bounds__(256) void recon_kernel(
    const float* __restrict__ x, const T* __restrict__ x1, const T* __restrict__ x2,
    const T* __restrict__ x3c, float* y1, float* y2, float* y3, float rho, float inv_n, int P,
    double* __restrict__ partials, T* __restrict__ gx1p, T* __restrict__ gx2,
    T* __restrict__ gx3c, const ReconFromA fa) {
  __shared__ float tile[TILE][TILE + 1];
  __shared__ float red[4][8];
  const TileIdx t = tile_idx(P);
  // FROMA: this thread's tap (its image column and its transposed-tile column are both == threadIdx.x mod 4) of the eight
  // input channels of both layers and the biases of this plane's channel.  The activations a 32 x 32 tile needs are
  // 32 rows x 8 positions x 8 channels per layer: staged in LDS with two float4 loads per thread and layer (reading them
  // per element through the vector cache -- 64 more load instructions per thread -- made the pass 0.11 ms slower)
  __shared__ __attribute__((aligned(16))) float stage[FROMA ? 2 * 8 * TILE * 8 : 4];
  float wt[FROMA ? 8 : 1], wf[FROMA ? 8 : 1], bt = 0.f, bfv = 0.f;
  if constexpr (FROMA) {
    const int b = blockIdx.z / fa.C, ch = blockIdx.z - b * fa.C, tap = threadIdx.x & 3;
    const long LA = (long)P * P / 4;
    const int h0 = blockIdx.y * TILE, w0 = blockIdx.x * TILE;
    {  // sT[cs][rr][k] = aT[cs][(h0 + rr) * P / 4 + w0 / 4 + k],  sF[cs][q][k] = aF[cs][(w0 + q) * P / 4 + h0 / 4 + k]
      // (T = bf16: the layer's input is a bf16 tensor too, and its output would have been rounded to bf16 on its way here)
      const int tid = threadIdx.y * TILE + threadIdx.x, cs = tid >> 5, rr = tid & 31;
      const T* pT = reinterpret_cast<const T*>(fa.aT) + (long)b * fa.a_bs + cs * LA + (long)(h0 + rr) * (P / 4) + w0 / 4;
      const T* pF = reinterpret_cast<const T*>(fa.aF) + (long)b * fa.a_bs + cs * LA + (long)(w0 + rr) * (P / 4) + h0 / 4;
      f32x4* dT = reinterpret_cast<f32x4*>(&stage[(cs * TILE + rr) * 8]);
      f32x4* dF = reinterpret_cast<f32x4*>(&stage[8 * TILE * 8 + (cs * TILE + rr) * 8]);
      dT[0] = Elem<T>::ld4(pT); dT[1] = Elem<T>::ld4(pT + 4);
      dF[0] = Elem<T>::ld4(pF); dF[1] = Elem<T>::ld4(pF + 4);
    }
#pragma unroll
    for (int cs = 0; cs < 8; ++cs) {
      wt[cs] = fa.wT[(cs * fa.C + ch) * 4 + tap];
      wf[cs] = fa.wF[(cs * fa.C + ch) * 4 + tap];
    }
    bt = fa.bT[ch]; bfv = fa.bF[ch];
    __syncthreads();
    const float* sF = stage + 8 * TILE * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // x3c at (column w0 + ty + 8 i, row h0 + tx): position (c * P + r) / 4 of the column-vectorised sequence
      float v = bfv;
#pragma unroll
      for (int cs = 0; cs < 8; ++cs) v = fmaf(sF[(cs * TILE + threadIdx.y + 8 * i) * 8 + (threadIdx.x >> 2)], wf[cs], v);
      if constexpr (sizeof(T) == 2) v = (float)(T)v;
      tile[threadIdx.y + 8 * i][threadIdx.x] = v;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) tile[threadIdx.y + 8 * i][threadIdx.x] = Elem<T>::ld(x3c + t.col_off[i]);
  }
  __syncthreads();
  float s[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float g3[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long o = t.row_off[i];
    // x and the multipliers are not needed again soon (streamed); x2 / x3 were just written and the
    // three gradients are read next by the backward: leave those to the cache
    const float xv = __builtin_nontemporal_load(x + o), a1 = Elem<T>::ld(x1 + o);
    float a2;
    if constexpr (FROMA) {
      a2 = bt;
#pragma unroll
      for (int cs = 0; cs < 8; ++cs) a2 = fmaf(stage[(cs * TILE + threadIdx.y + 8 * i) * 8 + (threadIdx.x >> 2)], wt[cs], a2);
      if constexpr (sizeof(T) == 2) a2 = (float)(T)a2;
    } else {
      a2 = Elem<T>::ld(x2 + o);
    }
    const float a3 = tile[threadIdx.x][threadIdx.y + 8 * i];
    const float m1 = __builtin_nontemporal_load(y1 + o), m2 = __builtin_nontemporal_load(y2 + o), m3 = __builtin_nontemporal_load(y3 + o);
    const ReconElem q = recon_elem<UPD, GRAD>(xv, a1, a2, a3, m1, m2, m3, rho, inv_n, s);
    if (UPD) { y1[o] = q.m1; y2[o] = q.m2; y3[o] = q.m3; }
    if (GRAD) {
      Elem<T>::st(gx2 + o, q.g2);
      g3[i] = q.g3;
      Elem<T>::st(gx1p + o, q.g1p);
    }
  }
  if (GRAD) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) tile[threadIdx.y + 8 * i][threadIdx.x] = g3[i];
  }
  // per-wavefront sums in fp32 (256 terms each), combined across the 4 waves and all tiles in fp64
  const int tid = threadIdx.y * TILE + threadIdx.x;
  const int lane = tid & 63, w = tid >> 6;
#pragma unroll
  for (int q = 0; q < 7; ++q) {
    const float v = wave_sum(s[q]);
    if (lane == 0) red[w][q] = v;
  }
  __syncthreads();
  if (GRAD) {
#pragma unroll
    for (int i = 0; i < 4; ++i) Elem<T>::st(gx3c + t.col_off[i], tile[threadIdx.x][threadIdx.y + 8 * i]);
  }
  const long blk = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  if (tid < 7)
    partials[blk * 7 + tid] = ((double)red[0][tid] + (double)red[1][tid]) + ((double)red[2][tid] + (double)red[3][tid]);
}
__global__ __launch_bounds__(1024) void sum7_kernel(const double* __restrict__ partials, long nblk,
                                                    double* __restrict__ sums7) {
  __shared__ double red[16];
  const int q = blockIdx.x;  // one workgroup per reduced quantity
  double acc = 0.0;
  for (long i = threadIdx.x; i < nblk; i += blockDim.x) acc += partials[i * 7 + q];
  const double tot = block_sum<double>(acc, red);
  if (threadIdx.x == 0) sums7[q] = tot;
}
size_t recon_partials_floats(int planes, int P) {
  return (size_t)planes * (P / TILE) * (P / TILE) * 7 * 2;
}
template <class T>
static void recon_launch_t(bool upd, const float* x, const T* x1, const T* x2, const T* x3c, float* y1, float* y2, float* y3,
                           float rho, float inv_n, int P, double* part, T* gx1p, T* gx2, T* gx3c, dim3 grid,
                           hipStream_t st) {
  if (!gx1p && !upd)  // no gradient buffers: the sums alone
    hipLaunchKernelGGL((recon_kernel<false, false, T>), grid, dim3(TILE, 8), 0, st, x, x1, x2, x3c, y1, y2, y3, rho, inv_n,
                       P, part, gx1p, gx2, gx3c, ReconFromA{});
  else if (upd)
    hipLaunchKernelGGL((recon_kernel<true, true, T>), grid, dim3(TILE, 8), 0, st, x, x1, x2, x3c, y1, y2, y3, rho, inv_n, P,
                       part, gx1p, gx2, gx3c, ReconFromA{});
  else
    hipLaunchKernelGGL((recon_kernel<false, true, T>), grid, dim3(TILE, 8), 0, st, x, x1, x2, x3c, y1, y2, y3, rho, inv_n, P,
                       part, gx1p, gx2, gx3c, ReconFromA{});
}
int recon_sum7(const float* block_partials, int planes, int P, double* sums7, hipStream_t st) {
  hipLaunchKernelGGL(sum7_kernel, dim3(7), dim3(1024), 0, st, reinterpret_cast<const double*>(block_partials),
                     (long)planes * (P / TILE) * (P / TILE), sums7);
  return check_launch("recon_sum7");
}
// sums7 == nullptr: the per-block partials are left in block_partials and the caller runs recon_sum7 later (the sums are
// only needed where the loss terms are assembled, which is not on the critical path of the step)
static int recon_launch(bool upd, const float* x, const float* x1, const float* x2, const float* x3c, float* y1,
                        float* y2, float* y3, float rho, int planes, int P, double* sums7, float* gx1p, float* gx2,
                        float* gx3c, float* block_partials, hipStream_t st, float grad_scale, int bf = 0) {
  if (P % TILE) { set_last_error("recon_losses: patch size must be a multiple of 32"); return LSHM_ERR_ARG; }
  const double n = (double)planes * P * P;
  double* part = reinterpret_cast<double*>(block_partials);
  dim3 grid(P / TILE, P / TILE, planes);
  const float inv_n = (float)(grad_scale / n);
  if (bf) {
    auto B = [](const float* q) { return reinterpret_cast<const bf16*>(q); };
    auto Bw = [](float* q) { return reinterpret_cast<bf16*>(q); };
    recon_launch_t<bf16>(upd, x, B(x1), B(x2), B(x3c), y1, y2, y3, rho, inv_n, P, part, Bw(gx1p), Bw(gx2), Bw(gx3c), grid, st);
  } else {
    recon_launch_t<float>(upd, x, x1, x2, x3c, y1, y2, y3, rho, inv_n, P, part, gx1p, gx2, gx3c, grid, st);
  }
  int rc = check_launch("recon_losses");
  if (rc || !sums7) return rc;
  hipLaunchKernelGGL(sum7_kernel, dim3(7), dim3(1024), 0, st, part,
                     (long)grid.x * grid.y * grid.z, sums7);
  return check_launch("recon_sum7");
}
int recon_losses_fwd_bwd(const float* x, const float* x1, const float* x2, const float* x3c,
                         const float* y1, const float* y2, const float* y3, float rho, int planes,
                         int P, double* sums7, float* gx1p, float* gx2, float* gx3c,
                         float* block_partials, hipStream_t st, float grad_scale, int bf) {
  return recon_launch(false, x, x1, x2, x3c, const_cast<float*>(y1), const_cast<float*>(y2), const_cast<float*>(y3),
                      rho, planes, P, sums7, gx1p, gx2, gx3c, block_partials, st, grad_scale, bf);
}
// multiplier_update_recon with the reconstructions of netT / netF formed from the inputs of their last layer (fp32 storage;
// sums left as per-block partials: recon_sum7 finishes them)
bool recon_from_a_supported(int C, int P, int Cin, int Cout, int Ls) {
  return !sched(LSHM_SCHED_NO_RECON_FROM_A) && Cin == 8 && Cout == C && C >= 1 && C <= 8 && P % TILE == 0 && (long)Ls * 4 == (long)P * P;
}
int multiplier_update_recon_from_a(const float* x, const float* x1, const float* aT, const float* aF, long a_bs, const float* wT,
                                   const float* bT, const float* wF, const float* bF, int C, float* y1, float* y2, float* y3,
                                   float rho, int planes, int P, float* gx1p, float* gx2, float* gx3c, float* block_partials,
                                   hipStream_t st, float grad_scale, int bf) {
  if (P % TILE || planes % C) { set_last_error("recon_losses: patch size must be a multiple of 32"); return LSHM_ERR_ARG; }
  if (!x || !x1 || !aT || !aF || !wT || !bT || !wF || !bF || !gx1p || !gx2 || !gx3c || !block_partials) {
    set_last_error("recon_losses: null pointer");
    return LSHM_ERR_ARG;
  }
  const double n = (double)planes * P * P;
  const ReconFromA fa{aT, aF, wT, bT, wF, bF, a_bs, C};
  if (bf)  // x1, the two activation tensors and the three gradient images are bf16 tensors
    hipLaunchKernelGGL((recon_kernel<true, true, bf16, true>), dim3(P / TILE, P / TILE, planes), dim3(TILE, 8), 0, st, x,
                       reinterpret_cast<const bf16*>(x1), (const bf16*)nullptr, (const bf16*)nullptr, y1, y2, y3, rho,
                       (float)(grad_scale / n), P, reinterpret_cast<double*>(block_partials), reinterpret_cast<bf16*>(gx1p),
                       reinterpret_cast<bf16*>(gx2), reinterpret_cast<bf16*>(gx3c), fa);
  else
    hipLaunchKernelGGL((recon_kernel<true, true, float, true>), dim3(P / TILE, P / TILE, planes), dim3(TILE, 8), 0, st, x, x1,
                       (const float*)nullptr, (const float*)nullptr, y1, y2, y3, rho, (float)(grad_scale / n), P,
                       reinterpret_cast<double*>(block_partials), gx1p, gx2, gx3c, fa);
  return check_launch("recon_losses");
}
// the same pass without the multiplier update (the C entry lshm_recon_losses_from_a: per-kernel parity of the FROMA form)
int recon_losses_from_a(const float* x, const float* x1, const float* aT, const float* aF, long a_bs, const float* wT, const float* bT,
                        const float* wF, const float* bF, int C, const float* y1, const float* y2, const float* y3, float rho,
                        int planes, int P, double* sums7, float* gx1p, float* gx2, float* gx3c, float* block_partials,
                        hipStream_t st) {
  if (P % TILE || planes % C) { set_last_error("recon_losses: patch size must be a multiple of 32"); return LSHM_ERR_ARG; }
  const double n = (double)planes * P * P;
  const ReconFromA fa{aT, aF, wT, bT, wF, bF, a_bs, C};
  const dim3 grid(P / TILE, P / TILE, planes);
  double* part = reinterpret_cast<double*>(block_partials);
  hipLaunchKernelGGL((recon_kernel<false, true, float, true>), grid, dim3(TILE, 8), 0, st, x, x1, (const float*)nullptr,
                     (const float*)nullptr, const_cast<float*>(y1), const_cast<float*>(y2), const_cast<float*>(y3), rho,
                     (float)(1.0 / n), P, part, gx1p, gx2, gx3c, fa);
  int rc = check_launch("recon_losses");
  if (rc) return rc;
  hipLaunchKernelGGL(sum7_kernel, dim3(7), dim3(1024), 0, st, part, (long)grid.x * grid.y * grid.z, sums7);
  return check_launch("recon_sum7");
}
// y_k += rho r_k, then the reconstruction terms of the next closure with the updated multipliers
int multiplier_update_recon(const float* x, const float* x1, const float* x2, const float* x3c, float* y1, float* y2,
                            float* y3, float rho, int planes, int P, double* sums7, float* gx1p, float* gx2,
                            float* gx3c, float* block_partials, hipStream_t st, float grad_scale, int bf) {
  return recon_launch(true, x, x1, x2, x3c, y1, y2, y3, rho, planes, P, sums7, gx1p, gx2, gx3c, block_partials, st,
                      grad_scale, bf);
}

// gx1 = gx1p - 0.5*(gT + gFc^T)
template <class T>
__global__ __launch_bounds__(256) void combine_dx1_kernel(const T* __restrict__ gx1p,
                                                          const T* __restrict__ gT,
                                                          const T* __restrict__ gFc,
                                                          T* __restrict__ gx1, int P) {
  __shared__ float tile[TILE][TILE + 1];
  const TileIdx t = tile_idx(P);
#pragma unroll
  for (int i = 0; i < 4; ++i) tile[threadIdx.y + 8 * i][threadIdx.x] = Elem<T>::ld(gFc + t.col_off[i]);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long o = t.row_off[i];
    Elem<T>::st(gx1 + o, Elem<T>::ld_nt(gx1p + o) - 0.5f * (Elem<T>::ld(gT + o) + tile[threadIdx.x][threadIdx.y + 8 * i]));
  }
}
int combine_dx1(const float* gx1p, const float* gT, const float* gFc, float* gx1, int planes, int P,
                hipStream_t st, int bf) {
  if (bf)
    hipLaunchKernelGGL(combine_dx1_kernel<bf16>, dim3(P / TILE, P / TILE, planes), dim3(TILE, 8), 0, st,
                       reinterpret_cast<const bf16*>(gx1p), reinterpret_cast<const bf16*>(gT),
                       reinterpret_cast<const bf16*>(gFc), reinterpret_cast<bf16*>(gx1), P);
  else
    hipLaunchKernelGGL(combine_dx1_kernel<float>, dim3(P / TILE, P / TILE, planes), dim3(TILE, 8), 0, st,
                       gx1p, gT, gFc, gx1, P);
  return check_launch("combine_dx1");
}

// y1 += rho (x-x1); y2 += rho (h-x2); y3 += rho (h-x3)   (src/kharmonic_lofar.py:200-202)
template <class T>
__global__ __launch_bounds__(256) void multiplier_update_kernel(
    const float* __restrict__ x, const T* __restrict__ x1, const T* __restrict__ x2,
    const T* __restrict__ x3c, float* __restrict__ y1, float* __restrict__ y2,
    float* __restrict__ y3, float rho, int P) {
  __shared__ float tile[TILE][TILE + 1];
  const TileIdx t = tile_idx(P);
#pragma unroll
  for (int i = 0; i < 4; ++i) tile[threadIdx.y + 8 * i][threadIdx.x] = Elem<T>::ld(x3c + t.col_off[i]);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long o = t.row_off[i];
    // the multipliers are touched once per iteration: stream them past the caches
    const float r1 = __builtin_nontemporal_load(x + o) - Elem<T>::ld(x1 + o), h = 0.5f * r1;
    __builtin_nontemporal_store(fmaf(rho, r1, __builtin_nontemporal_load(y1 + o)), y1 + o);
    __builtin_nontemporal_store(fmaf(rho, h - Elem<T>::ld(x2 + o), __builtin_nontemporal_load(y2 + o)), y2 + o);
    __builtin_nontemporal_store(fmaf(rho, h - tile[threadIdx.x][threadIdx.y + 8 * i], __builtin_nontemporal_load(y3 + o)),
                                y3 + o);
  }
}
int multiplier_update(const float* x, const float* x1, const float* x2, const float* x3c, float* y1,
                      float* y2, float* y3, float rho, int planes, int P, hipStream_t st, int bf) {
  if (P % TILE) { set_last_error("multiplier_update: patch size must be a multiple of 32"); return LSHM_ERR_ARG; }
  if (bf)
    hipLaunchKernelGGL(multiplier_update_kernel<bf16>, dim3(P / TILE, P / TILE, planes), dim3(TILE, 8), 0, st, x,
                       reinterpret_cast<const bf16*>(x1), reinterpret_cast<const bf16*>(x2),
                       reinterpret_cast<const bf16*>(x3c), y1, y2, y3, rho, P);
  else
    hipLaunchKernelGGL(multiplier_update_kernel<float>, dim3(P / TILE, P / TILE, planes), dim3(TILE, 8), 0,
                       st, x, x1, x2, x3c, y1, y2, y3, rho, P);
  return check_launch("multiplier_update");
}

// --------------------------------------------------------------------------
// Adam over one flat arena (torch.optim.Adam defaults; src/kharmonic_lofar.py:92)
// --------------------------------------------------------------------------
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                            const int* __restrict__ step_dev, int step_host, float gscale) {
  const int tstep = step_dev ? *step_dev : step_host;
  const double c1 = 1.0 - pow((double)b1, (double)tstep);
  const double c2 = 1.0 - pow((double)b2, (double)tstep);
  const float step_size = (float)((double)lr / c1);
  const float inv_sqrt_c2 = (float)(1.0 / sqrt(c2));
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const float gi = g[i] * gscale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= step_size * mi / (sqrtf(vi) * inv_sqrt_c2 + eps);
  }
}
int adam_step_flat(float* p, const float* g, float* m, float* v, long n, float lr, float b1,
                   float b2, float eps, const int* step_dev, int step_host, float gscale,
                   hipStream_t st) {
  hipLaunchKernelGGL(adam_kernel, dim3(min(cdiv(n, 256), 2048)), dim3(256), 0, st, p, g, m, v, n, lr,
                     b1, b2, eps, step_dev, step_host, gscale);
  return check_launch("adam");
}

__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float a, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) y[i] += a * x[i];
}
int axpy_flat(float* y, const float* x, float alpha, long n, hipStream_t st) {
  hipLaunchKernelGGL(axpy_kernel, dim3(min(cdiv(n, 256), 2048)), dim3(256), 0, st, y, x, alpha, n);
  return check_launch("axpy");
}
__global__ void scale_kernel(float* __restrict__ x, float a, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) x[i] *= a;
}
int scale_flat(float* x, float alpha, long n, hipStream_t st) {
  hipLaunchKernelGGL(scale_kernel, dim3(min(cdiv(n, 256), 2048)), dim3(256), 0, st, x, alpha, n);
  return check_launch("scale");
}

// two-stage deterministic dot product; ws >= 2*DOT_BLOCKS floats (holds doubles)
#define DOT_BLOCKS 256
__global__ __launch_bounds__(256) void dot_stage1(const float* __restrict__ a, const float* __restrict__ b,
                                                  long n, double* __restrict__ part, int vec) {
  __shared__ double red[16];
  double acc = 0.0;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
  if (vec) {  // both pointers 16-byte aligned: float4 loads, the n % 4 tail goes to the first threads
    const long n4 = n >> 2;
    const f32x4* a4 = reinterpret_cast<const f32x4*>(a);
    const f32x4* b4 = reinterpret_cast<const f32x4*>(b);
    for (long i = gid; i < n4; i += stride) {
      const f32x4 x = a4[i], y = b4[i];
      acc += ((double)x[0] * (double)y[0] + (double)x[1] * (double)y[1]) +
             ((double)x[2] * (double)y[2] + (double)x[3] * (double)y[3]);
    }
    if (gid < (n & 3)) acc += (double)a[4 * n4 + gid] * (double)b[4 * n4 + gid];
  } else {
    for (long i = gid; i < n; i += stride) acc += (double)a[i] * (double)b[i];
  }
  const double tot = block_sum<double>(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = tot;
}
__global__ __launch_bounds__(256) void dot_stage2(const double* __restrict__ part, int nb,
                                                  double* __restrict__ out) {
  __shared__ double red[16];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) acc += part[i];
  const double tot = block_sum<double>(acc, red);
  if (threadIdx.x == 0) out[0] = tot;
}
__global__ void dot_stage2(const double* __restrict__ part, int nb, double* __restrict__ out);
__global__ __launch_bounds__(256) void asum_stage1(const float* __restrict__ a, long n, double* __restrict__ part) {
  __shared__ double red[16];
  double acc = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    acc += (double)fabsf(a[i]);
  const double tot = block_sum<double>(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = tot;
}
int asum_flat(const float* a, long n, double* out, float* ws, hipStream_t st) {
  double* part = reinterpret_cast<double*>(ws);
  hipLaunchKernelGGL(asum_stage1, dim3(DOT_BLOCKS), dim3(256), 0, st, a, n, part);
  int rc = check_launch("asum1");
  if (rc) return rc;
  hipLaunchKernelGGL(dot_stage2, dim3(1), dim3(256), 0, st, part, DOT_BLOCKS, out);
  return check_launch("asum2");
}
int dot_flat(const float* a, const float* b, long n, double* out, float* ws, hipStream_t st) {
  double* part = reinterpret_cast<double*>(ws);
  const int vec = (((uintptr_t)a | (uintptr_t)b) & 15) == 0;
  hipLaunchKernelGGL(dot_stage1, dim3(DOT_BLOCKS), dim3(256), 0, st, a, b, n, part, vec);
  int rc = check_launch("dot1");
  if (rc) return rc;
  hipLaunchKernelGGL(dot_stage2, dim3(1), dim3(256), 0, st, part, DOT_BLOCKS, out);
  return check_launch("dot2");
}

// --------------------------------------------------------------------------
// Several inner products in one launch pair: out[i] = <a_i, b_i>, i < count <= kMaxDots (grid.y = i).
// --------------------------------------------------------------------------
struct DotList { const float* a[kMaxDots]; const float* b[kMaxDots]; };
__device__ __forceinline__ double dot_chunk(const float* __restrict__ a, const float* __restrict__ b, long n, int vec) {
  double acc = 0.0;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
  if (vec) {
    const long n4 = n >> 2;
    const f32x4* a4 = reinterpret_cast<const f32x4*>(a);
    const f32x4* b4 = reinterpret_cast<const f32x4*>(b);
    for (long i = gid; i < n4; i += stride) {
      const f32x4 x = a4[i], y = b4[i];
      acc += ((double)x[0] * (double)y[0] + (double)x[1] * (double)y[1]) +
             ((double)x[2] * (double)y[2] + (double)x[3] * (double)y[3]);
    }
    if (gid < (n & 3)) acc += (double)a[4 * n4 + gid] * (double)b[4 * n4 + gid];
  } else {
    for (long i = gid; i < n; i += stride) acc += (double)a[i] * (double)b[i];
  }
  return acc;
}
__global__ __launch_bounds__(256) void multi_dot_stage1(const DotList L, long n, double* __restrict__ part, int vec) {
  __shared__ double red[16];
  const int j = blockIdx.y;
  const double tot = block_sum<double>(dot_chunk(L.a[j], L.b[j], n, vec), red);
  if (threadIdx.x == 0) part[(long)j * DOT_BLOCKS + blockIdx.x] = tot;
}
__global__ __launch_bounds__(256) void multi_dot_stage2(const double* __restrict__ part, double* __restrict__ out) {
  __shared__ double red[16];
  const double tot = block_sum<double>(part[(long)blockIdx.x * DOT_BLOCKS + threadIdx.x], red);
  if (threadIdx.x == 0) out[blockIdx.x] = tot;
}
static int all_aligned16(const float* const* p, int n) {
  uintptr_t m = 0;
  for (int i = 0; i < n; ++i) m |= reinterpret_cast<uintptr_t>(p[i]);
  return (m & 15) == 0;
}
size_t multi_dot_workspace_doubles(int count) { return (size_t)count * DOT_BLOCKS; }
int multi_dot_flat(const float* const* a, const float* const* b, int count, long n, double* out, double* ws,
                   hipStream_t st) {
  DotList L;
  for (int i = 0; i < kMaxDots; ++i) { L.a[i] = a[i < count ? i : 0]; L.b[i] = b[i < count ? i : 0]; }
  const int vec = all_aligned16(a, count) && all_aligned16(b, count);
  hipLaunchKernelGGL(multi_dot_stage1, dim3(DOT_BLOCKS, count), dim3(256), 0, st, L, n, ws, vec);
  int rc = check_launch("multi_dot1");
  if (rc) return rc;
  hipLaunchKernelGGL(multi_dot_stage2, dim3(count), dim3(DOT_BLOCKS), 0, st, ws, out);
  return check_launch("multi_dot2");
}

// --------------------------------------------------------------------------
// L-BFGS two-loop recursion (src/lbfgsnew.py:632-651) without host round trips.  The host version needs the value
// of every inner product before it can enqueue the next update: 3 m synchronisations per search direction.  Here
// each link of the chain is ONE launch: every workgroup first adds up the DOT_BLOCKS stage-1 partials the previous
// link left behind (all workgroups in the same order: the same coefficient, bitwise), updates its share of the
// vector and leaves the stage-1 partial of the NEXT inner product over the updated share.
//   first loop  (i = m-1 .. 0):  al_i = <s_i, q> / <y_i, s_i>;  q -= al_i y_i          [after i = 0: q *= H_diag]
//   second loop (i = 0 .. m-1):  be_i = <y_i, r> / <y_i, s_i>;  r += (al_i - be_i) s_i
// Coefficients are formed in double and rounded to float for the update, as the host version's axpy does.
// --------------------------------------------------------------------------
struct LbfgsLink {
  float* q;                 // updated in place
  const float* upd;         // q += coef * upd
  const float* nxt;         // stage-1 partial of <nxt, q> is left in part_out (null: none)
  const double* prev_part;  // partials of the inner product that defines coef
  const double* ys_part;    // partials of <y_i, s_i>
  double* al;               // al[i]: written by the first loop (workgroup 0), read by the second
  double* part_out;
  float post_scale;         // q *= post_scale after the update (H_diag behind the first loop, else 1)
  int second;               // 0: coef = -al_i;  1: coef = al_i - be_i
};
__global__ __launch_bounds__(256) void lbfgs_link_kernel(const LbfgsLink k, long n, int vec) {
  __shared__ double red[16];
  __shared__ double sc[2];
  const double dot = block_sum<double>(k.prev_part[threadIdx.x], red);
  if (threadIdx.x == 0) sc[0] = dot;
  const double ys = block_sum<double>(k.ys_part[threadIdx.x], red);
  if (threadIdx.x == 0) sc[1] = ys;
  __syncthreads();
  const double ratio = sc[0] * (1.0 / sc[1]);  // dot * ro_i, ro_i = 1 / <y_i, s_i>
  double coef;
  if (k.second) {
    coef = *k.al - ratio;
  } else {
    coef = -ratio;
    if (blockIdx.x == 0 && threadIdx.x == 0) *k.al = ratio;
  }
  const float c = (float)coef, ps = k.post_scale;
  double acc = 0.0;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
  if (vec) {
    const long n4 = n >> 2;
    f32x4* q4 = reinterpret_cast<f32x4*>(k.q);
    const f32x4* u4 = reinterpret_cast<const f32x4*>(k.upd);
    const f32x4* x4 = reinterpret_cast<const f32x4*>(k.nxt);
    for (long i = gid; i < n4; i += stride) {
      f32x4 v = q4[i];
      const f32x4 u = u4[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (v[e] + c * u[e]) * ps;
      q4[i] = v;
      if (x4) {
        const f32x4 x = x4[i];
        acc += ((double)x[0] * (double)v[0] + (double)x[1] * (double)v[1]) +
               ((double)x[2] * (double)v[2] + (double)x[3] * (double)v[3]);
      }
    }
    if (gid < (n & 3)) {
      const long i = 4 * n4 + gid;
      const float v = (k.q[i] + c * k.upd[i]) * ps;
      k.q[i] = v;
      if (k.nxt) acc += (double)k.nxt[i] * (double)v;
    }
  } else {
    for (long i = gid; i < n; i += stride) {
      const float v = (k.q[i] + c * k.upd[i]) * ps;
      k.q[i] = v;
      if (k.nxt) acc += (double)k.nxt[i] * (double)v;
    }
  }
  if (k.nxt) {
    const double tot = block_sum<double>(acc, red);
    if (threadIdx.x == 0) k.part_out[blockIdx.x] = tot;
  }
}
// q = -g (times post_scale) and the stage-1 partial of <nxt, q>
__global__ __launch_bounds__(256) void lbfgs_start_kernel(const float* __restrict__ g, float* __restrict__ q,
                                                          const float* __restrict__ nxt, double* __restrict__ part_out,
                                                          float post_scale, long n) {
  __shared__ double red[16];
  double acc = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = -g[i] * post_scale;
    q[i] = v;
    if (nxt) acc += (double)nxt[i] * (double)v;
  }
  if (nxt) {
    const double tot = block_sum<double>(acc, red);
    if (threadIdx.x == 0) part_out[blockIdx.x] = tot;
  }
}
size_t lbfgs_direction_workspace_doubles(int m) { return (size_t)(m + 2) * DOT_BLOCKS + (size_t)m; }
int lbfgs_direction(const float* const* y, const float* const* s, int m, const float* g, double h_diag, float* d, long n,
                    double* ws, hipStream_t st) {
  if (m == 0) {
    hipLaunchKernelGGL(lbfgs_start_kernel, dim3(DOT_BLOCKS), dim3(256), 0, st, g, d, (const float*)nullptr,
                       (double*)nullptr, (float)h_diag, n);
    return check_launch("lbfgs_start");
  }
  double* ys_part = ws;                              // [m][DOT_BLOCKS]
  double* ping = ws + (size_t)m * DOT_BLOCKS;        // [2][DOT_BLOCKS]
  double* al = ping + 2 * DOT_BLOCKS;                // [m]
  DotList L;
  for (int i = 0; i < kMaxDots; ++i) { L.a[i] = y[i < m ? i : 0]; L.b[i] = s[i < m ? i : 0]; }
  const int vec = all_aligned16(y, m) && all_aligned16(s, m) && (reinterpret_cast<uintptr_t>(d) & 15) == 0;
  hipLaunchKernelGGL(multi_dot_stage1, dim3(DOT_BLOCKS, m), dim3(256), 0, st, L, n, ys_part, vec);
  int rc = check_launch("lbfgs_ys");
  if (rc) return rc;
  hipLaunchKernelGGL(lbfgs_start_kernel, dim3(DOT_BLOCKS), dim3(256), 0, st, g, d, s[m - 1], ping, 1.f, n);
  if ((rc = check_launch("lbfgs_start"))) return rc;
  int cur = 0;
  for (int i = m - 1; i >= 0; --i) {  // q -= al_i y_i; next: <s_{i-1}, q>, or <y_0, H q> behind the last link
    LbfgsLink k{d, y[i], i > 0 ? s[i - 1] : y[0], ping + cur * DOT_BLOCKS, ys_part + (size_t)i * DOT_BLOCKS, al + i,
                ping + (cur ^ 1) * DOT_BLOCKS, i == 0 ? (float)h_diag : 1.f, 0};
    hipLaunchKernelGGL(lbfgs_link_kernel, dim3(DOT_BLOCKS), dim3(256), 0, st, k, n, vec);
    if ((rc = check_launch("lbfgs_link"))) return rc;
    cur ^= 1;
  }
  for (int i = 0; i < m; ++i) {       // r += (al_i - be_i) s_i; next: <y_{i+1}, r>
    LbfgsLink k{d, s[i], i + 1 < m ? y[i + 1] : nullptr, ping + cur * DOT_BLOCKS, ys_part + (size_t)i * DOT_BLOCKS,
                al + i, ping + (cur ^ 1) * DOT_BLOCKS, 1.f, 1};
    hipLaunchKernelGGL(lbfgs_link_kernel, dim3(DOT_BLOCKS), dim3(256), 0, st, k, n, vec);
    if ((rc = check_launch("lbfgs_link"))) return rc;
    cur ^= 1;
  }
  return LSHM_OK;
}

// --------------------------------------------------------------------------
// RICA penalty: loss = scale * sum log cosh(z);  dz (+)= scale * tanh(z)
// (src/kharmonic_lofar.py:169-171).  One workgroup: the latents are tiny.
// --------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void logcosh_kernel(const float* __restrict__ z, long ldz, int rows,
                                                       int cols, float scale, double* __restrict__ loss,
                                                       float* __restrict__ dz, long lddz, int accumulate) {
  __shared__ double red[16];
  double acc = 0.0;
  const long n = (long)rows * cols;
  for (long i = threadIdx.x; i < n; i += blockDim.x) {
    const int r = (int)(i / cols), c = (int)(i - (long)r * cols);
    const float v = z[(long)r * ldz + c];
    const float a = fabsf(v);
    // log cosh v = |v| + log1p(exp(-2|v|)) - log 2   (no overflow)
    acc += (double)(a + log1pf(expf(-2.f * a)) - 0.69314718056f);
    if (dz) {
      const float gz = scale * tanhf(v);
      float* d = dz + (long)r * lddz + c;
      *d = accumulate ? *d + gz : gz;
    }
  }
  const double tot = block_sum<double>(acc, red);
  if (threadIdx.x == 0 && loss) loss[0] = tot * (double)scale;
}
// All three RICA penalties of the closure in one launch: Mu = [mu | muT | muF] (rows x (c0+c1+c2)),
// per-segment scale; per-workgroup partial sums [grid][3] are combined by the caller's finalize kernel.
struct Seg3 { int end[3]; float scale[3]; };
__global__ __launch_bounds__(256) void logcosh3_kernel(const float* __restrict__ z, long ldz, int rows, int cols,
                                                       Seg3 sg, double* __restrict__ partial,
                                                       float* __restrict__ dz, long lddz) {
  __shared__ double red[16];
  double acc[3] = {0.0, 0.0, 0.0};
  const long n = (long)rows * cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols), c = (int)(i - (long)r * cols);
    const int seg = c < sg.end[0] ? 0 : (c < sg.end[1] ? 1 : 2);
    const float v = z[(long)r * ldz + c];
    const float a = fabsf(v);
    acc[seg] += (double)(a + log1pf(expf(-2.f * a)) - 0.69314718056f);
    dz[(long)r * lddz + c] += sg.scale[seg] * tanhf(v);
  }
  for (int q = 0; q < 3; ++q) {
    const double tot = block_sum<double>(acc[q], red);
    if (threadIdx.x == 0) partial[blockIdx.x * 3 + q] = tot * (double)sg.scale[q];
  }
}
int logcosh3_fwd_bwd(const float* z, long ldz, int rows, const int* seg_cols, const float* seg_scale,
                     double* partial, int nblocks, float* dz, long lddz, hipStream_t st) {
  Seg3 sg;
  int e = 0;
  for (int i = 0; i < 3; ++i) { e += seg_cols[i]; sg.end[i] = e; sg.scale[i] = seg_scale[i]; }
  hipLaunchKernelGGL(logcosh3_kernel, dim3(nblocks), dim3(256), 0, st, z, ldz, rows, e, sg, partial, dz, lddz);
  return check_launch("logcosh3");
}

int logcosh_mean_fwd_bwd(const float* z, long ldz, int rows, int cols, float scale, double* loss,
                         float* dz, long lddz, int accumulate, hipStream_t st) {
  hipLaunchKernelGGL(logcosh_kernel, dim3(1), dim3(1024), 0, st, z, ldz, rows, cols, scale, loss, dz,
                     lddz, accumulate);
  return check_launch("logcosh");
}

}  // namespace lshm
