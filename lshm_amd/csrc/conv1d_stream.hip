// Streaming kernels for the two outermost layers of the 1-D autoencoders (src/lofar_models.py:115-116
// conv0/conv1 and :141-142 tconv4/tconv5, and the data gradients with the same geometry).
// Kernel size == stride == 4, so every input position feeds exactly one group of 4 output positions
// (or the reverse) and the op is a pure stream: K is 4..12 and there is nothing to tile.  A thread owns
// one position of the short tensor for all channels: loads are one float (coalesced across lanes) or
// one float4 per channel, stores one float4 or one float per channel, weights are uniform and come
// through the scalar cache.  fp32 FMA chains in a fixed order: bitwise reproducible.
#include "kernels.h"

namespace lshm {

// Weights and biases are read-only for the whole launch and their addresses are wave-uniform: viewing
// them through the constant address space lets the compiler fetch them with scalar loads (s_load_dwordx4
// into SGPRs) instead of 64-lane vector loads of one address.
typedef const __attribute__((address_space(4))) float* cfloat_ptr;
typedef const __attribute__((address_space(4))) f32x4* cf32x4_ptr;
__device__ __forceinline__ f32x4 uniform_load4(const float* q) { return *(cf32x4_ptr)(q); }
__device__ __forceinline__ float uniform_load(const float* q) { return *(cfloat_ptr)(q); }

// upsampling direction: big[b, cb, 4j + t - pad] = bias[cb] + sum_cs small[b, cs, j] * w[cs, cb, t]
//   pad = 0: forward of ConvTranspose1d(k4, s4);  pad = 1: data gradient of Conv1d(k4, s4, p1)
template <int CS, int CB, bool PAD>
__global__ __launch_bounds__(256) void tconv1d_stream_kernel(const Conv1dDgradParams p0, const Conv1dDgradParams p1) {
  const Conv1dDgradParams& p = blockIdx.y ? p1 : p0;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)p.B * p.Ls) return;
  const int b = (int)(idx / p.Ls), j = (int)(idx - (long)b * p.Ls);
  const float* xs = p.s + (long)b * p.s_bs + j;
  float xv[CS], xn[CS];
#pragma unroll
  for (int cs = 0; cs < CS; ++cs) {
    xv[cs] = xs[(long)cs * p.Ls];
    xn[cs] = (PAD && j + 1 < p.Ls) ? xs[(long)cs * p.Ls + 1] : 0.f;
  }
  const long obase = (long)b * p.big_bs + 4 * (long)j;
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    const float bv = p.bias ? uniform_load(p.bias + cb) : 0.f;
    f32x4 acc = {bv, bv, bv, bv};
#pragma unroll
    for (int cs = 0; cs < CS; ++cs) {
      const f32x4 w4 = uniform_load4(p.w + ((long)cs * CB + cb) * 4);
      if (PAD) {  // output 4j+r comes from tap r+1 of position j (r < 3) and tap 0 of position j+1
        acc[0] = fmaf(xv[cs], w4[1], acc[0]);
        acc[1] = fmaf(xv[cs], w4[2], acc[1]);
        acc[2] = fmaf(xv[cs], w4[3], acc[2]);
        acc[3] = fmaf(xn[cs], w4[0], acc[3]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = fmaf(xv[cs], w4[r], acc[r]);
      }
    }
    const long g = obase + (long)cb * p.Lb;
    if (p.act) {
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = elu(acc[r]);
    }
    if (p.dact) {
      const f32x4 sv = *reinterpret_cast<const f32x4*>(p.dact + g);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] *= elu_grad_from_out(sv[r]);
    }
    *reinterpret_cast<f32x4*>(p.big + g) = acc;
  }
}

// downsampling direction: y[b, co, j] = bias[co] + sum_{ci,t} w[co, ci, t] * x[b, ci, 4j - pad + t]
//   pad = 1: forward of Conv1d(k4, s4, p1);  pad = 0: data gradient of ConvTranspose1d(k4, s4)
template <int CIN, int COUT, bool PAD>
__global__ __launch_bounds__(256) void conv1d_stream_kernel(const Conv1dFwdParams p0, const Conv1dFwdParams p1) {
  const Conv1dFwdParams& p = blockIdx.y ? p1 : p0;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)p.B * p.Lo) return;
  const int b = (int)(idx / p.Lo), j = (int)(idx - (long)b * p.Lo);
  const float* xb = p.x + (long)b * p.x_bs + 4 * (long)j;
  f32x4 v[CIN];
  float xm[CIN];
#pragma unroll
  for (int ci = 0; ci < CIN; ++ci) {
    v[ci] = *reinterpret_cast<const f32x4*>(xb + (long)ci * p.L);
    xm[ci] = (PAD && j > 0) ? xb[(long)ci * p.L - 1] : 0.f;
  }
  const long obase = (long)b * p.y_bs + j;
#pragma unroll
  for (int co = 0; co < COUT; ++co) {
    float acc = p.bias ? uniform_load(p.bias + co) : 0.f;
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
      const f32x4 w4 = uniform_load4(p.w + ((long)co * CIN + ci) * 4);
      if (PAD) {  // taps 0..3 sit at 4j-1 .. 4j+2
        acc = fmaf(xm[ci], w4[0], acc);
        acc = fmaf(v[ci][0], w4[1], acc);
        acc = fmaf(v[ci][1], w4[2], acc);
        acc = fmaf(v[ci][2], w4[3], acc);
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = fmaf(v[ci][t], w4[t], acc);
      }
    }
    const long g = obase + (long)co * p.Lo;
    if (p.act) acc = elu(acc);
    if (p.dact) acc *= elu_grad_from_out(p.dact[g]);
    p.y[g] = acc;
  }
}

static bool aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

bool tconv1d_stream_supported(const Conv1dDgradParams& p) {
  const bool shape = (p.Cs == 8 && p.Cb == 4) || (p.Cs == 12 && p.Cb == 8);
  return shape && (p.pad == 0 || p.pad == 1) && p.Lb == 4 * p.Ls && p.big_bs % 4 == 0 && aligned16(p.big) &&
         aligned16(p.w) && (!p.dact || aligned16(p.dact));
}
bool conv1d_stream_supported(const Conv1dFwdParams& p) {
  const bool shape = (p.Cin == 4 && p.Cout == 8) || (p.Cin == 8 && p.Cout == 12);
  return shape && (p.pad == 0 || p.pad == 1) && p.L == 4 * p.Lo && p.x_bs % 4 == 0 && aligned16(p.x) && aligned16(p.w);
}

int tconv1d_stream(const Conv1dDgradParams& p, const Conv1dDgradParams* p1, hipStream_t st) {
  const dim3 grid(cdiv((long)p.B * p.Ls, 256), p1 ? 2 : 1);
  const Conv1dDgradParams& q = p1 ? *p1 : p;
#define LSHM_LAUNCH(CS, CB)                                                                                   \
  if (p.pad) hipLaunchKernelGGL((tconv1d_stream_kernel<CS, CB, true>), grid, dim3(256), 0, st, p, q);          \
  else hipLaunchKernelGGL((tconv1d_stream_kernel<CS, CB, false>), grid, dim3(256), 0, st, p, q)
  if (p.Cs == 8) { LSHM_LAUNCH(8, 4); } else { LSHM_LAUNCH(12, 8); }
#undef LSHM_LAUNCH
  return check_launch("tconv1d_stream");
}
int conv1d_stream(const Conv1dFwdParams& p, const Conv1dFwdParams* p1, hipStream_t st) {
  const dim3 grid(cdiv((long)p.B * p.Lo, 256), p1 ? 2 : 1);
  const Conv1dFwdParams& q = p1 ? *p1 : p;
#define LSHM_LAUNCH(CI, CO)                                                                                   \
  if (p.pad) hipLaunchKernelGGL((conv1d_stream_kernel<CI, CO, true>), grid, dim3(256), 0, st, p, q);           \
  else hipLaunchKernelGGL((conv1d_stream_kernel<CI, CO, false>), grid, dim3(256), 0, st, p, q)
  if (p.Cin == 4) { LSHM_LAUNCH(4, 8); } else { LSHM_LAUNCH(8, 12); }
#undef LSHM_LAUNCH
  return check_launch("conv1d_stream");
}

}  // namespace lshm
