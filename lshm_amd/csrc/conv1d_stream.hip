// Streaming kernels for the two outermost layers of the 1-D autoencoders (src/lofar_models.py:115-116
// conv0/conv1 and :141-142 tconv4/tconv5, and the data gradients with the same geometry).
// Kernel size == stride == 4, so every input position feeds exactly one group of 4 output positions
// (or the reverse) and the op is a pure stream: K is 4..12 and there is nothing to tile.  A thread owns
// one position of the short tensor for all channels: loads are one float (coalesced across lanes) or
// one float4 per channel, stores one float4 or one float per channel, weights are uniform and come
// through the scalar cache.  fp32 FMA chains in a fixed order: bitwise reproducible.
#include <stdlib.h>

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
// TO: element type of `big` (bf16 storage of the image-sized tensors, see common.h)
// TS: element type of `small`; `dact` has big's shape and element type
template <int CS, int CB, bool PAD, class TO = float, class TS = float>
__global__ __launch_bounds__(256) void tconv1d_stream_kernel(const Conv1dDgradParams p0, const Conv1dDgradParams p1) {
  const Conv1dDgradParams& p = blockIdx.y ? p1 : p0;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)p.B * p.Ls) return;
  const int b = (int)(idx / p.Ls), j = (int)(idx - (long)b * p.Ls);
  const TS* xs = reinterpret_cast<const TS*>(p.s) + (long)b * p.s_bs + j;
  float xv[CS], xn[CS];
#pragma unroll
  for (int cs = 0; cs < CS; ++cs) {
    xv[cs] = Elem<TS>::ld(xs + (long)cs * p.Ls);
    xn[cs] = (PAD && j + 1 < p.Ls) ? Elem<TS>::ld(xs + (long)cs * p.Ls + 1) : 0.f;
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
      const f32x4 sv = Elem<TO>::ld4(reinterpret_cast<const TO*>(p.dact) + g);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] *= elu_grad_from_out(sv[r]);
    }
    Elem<TO>::st4(reinterpret_cast<TO*>(p.big) + g, acc);
  }
}

// downsampling direction: y[b, co, j] = bias[co] + sum_{ci,t} w[co, ci, t] * x[b, ci, 4j - pad + t]
//   pad = 1: forward of Conv1d(k4, s4, p1);  pad = 0: data gradient of ConvTranspose1d(k4, s4)
// TI: element type of x; TO: of y (and of dact, which has y's shape)
template <int CIN, int COUT, bool PAD, class TI = float, class TO = float>
__global__ __launch_bounds__(256) void conv1d_stream_kernel(const Conv1dFwdParams p0, const Conv1dFwdParams p1) {
  const Conv1dFwdParams& p = blockIdx.y ? p1 : p0;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)p.B * p.Lo) return;
  const int b = (int)(idx / p.Lo), j = (int)(idx - (long)b * p.Lo);
  const TI* xb = reinterpret_cast<const TI*>(p.x) + (long)b * p.x_bs + 4 * (long)j;
  f32x4 v[CIN];
  float xm[CIN];
#pragma unroll
  for (int ci = 0; ci < CIN; ++ci) {
    v[ci] = Elem<TI>::ld4(xb + (long)ci * p.L);
    xm[ci] = (PAD && j > 0) ? Elem<TI>::ld(xb + (long)ci * p.L - 1) : 0.f;
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
    if (p.dact) acc *= elu_grad_from_out(Elem<TO>::ld(reinterpret_cast<const TO*>(p.dact) + g));
    Elem<TO>::st(reinterpret_cast<TO*>(p.y) + g, acc);
  }
}

// ----------------------------------------------------------------------------------------------
// k4 s4 weight gradient of the same layers, bias gradient fused, no LDS staging:
//   dW[cs, cb, t] = sum_{b,j} small[b,cs,j] * big[b,cb,4j-pad+t]
// v_mfma_f32_4x4x1_16b_f32 runs 16 independent 4x4 outer products per instruction (measured layout:
// D[lane l][reg r] += A[lane 4*(l/4)+r] * B[lane l]).  Slot b = l/4 carries one position, the A quad
// of a slot 4 small channels, the B quad 4 big channels at one tap: every lane of every instruction is
// useful for 8/12 and 4/8 channels, where the 16x16 tile would be half empty.  A wavefront walks tiles
// of 64 positions: a lane loads one float4 of `small` (4 positions of its channel -> 4 MFMA steps) and
// four float4 of `big` (the 16 taps of those positions) straight from global memory into registers,
// the next tile's loads in flight while the current tile's MFMAs run.  Each slot accumulates its own
// positions; slots, wavefronts and workgroups are combined at the end in a fixed order.
// ----------------------------------------------------------------------------------------------
// DG (fused data gradient; transposed conv only, pad 0, CB == 4): the registers that feed the weight gradient of
// a transposed layer -- its input `small` (an ELU output) and the gradient `big` of its output -- are everything
// the layer's data gradient needs too:
//   dsmall[cs, j] = ELU'(small[cs, j]) * sum_{cb,t} w[cs, cb, t] * big[cb, 4j + t]
// A lane holds one big channel (cb = q) of its 4 positions: it forms the partial sums of all CS channels over its
// taps on the vector ALU, the four lanes of a quad exchange them reduce-scatter fashion (two DPP quad
// permutations: 24 moves for 32 values), and every lane ends with exactly the (channel, positions) float4s it
// loaded from `small` -- multiplied by ELU' and stored where they came from in the gradient tensor.  One pass over
// `big` and `small` instead of two (src/lofar_models.py:141-142 backward: 469 -> 268 MB for the pair at B = 256).
struct DgradArgs { const float* w0; const float* w1; float* d0; float* d1; long d_bs; };
template <int CTRL>
__device__ __forceinline__ float quad_swap(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CS, int CB, class TB, bool DG, class TS = float>  // TS: element type of `small` (and of the fused data gradient)
__device__ __forceinline__ void conv1d_wgrad_stream_body(const float* __restrict__ small0,
                                                         const float* __restrict__ small1, long s_bs,
                                                         const float* __restrict__ big0_,
                                                         const float* __restrict__ big1_, long big_bs,
                                                         float* __restrict__ partial0,
                                                         float* __restrict__ partial1, int Ls, int Lb,
                                                         int pad, int bias_from, int ntiles, DgradArgs dg) {
  static_assert(!DG || (CB == 4 && CS == 8), "fused data gradient: 8 -> 4 channel layers");
  const TS* small = reinterpret_cast<const TS*>(blockIdx.y ? small1 : small0);
  const TB* big = reinterpret_cast<const TB*>(blockIdx.y ? big1_ : big0_);  // TB: element type of `big`
  float* partial = blockIdx.y ? partial1 : partial0;
  constexpr int GA = CS / 4, GB = CB / 4;
  constexpr int NW = CS * CB * 4, SLAB = NW + 16;
  __shared__ float comb[4][SLAB];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int slot = lane >> 2, q = lane & 3;
  f32x4 acc[GA][GB][4];
#pragma unroll
  for (int a = 0; a < GA; ++a)
#pragma unroll
    for (int g = 0; g < GB; ++g)
#pragma unroll
      for (int tp = 0; tp < 4; ++tp) acc[a][g][tp] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum[GA > GB ? GA : GB];
#pragma unroll
  for (int i = 0; i < (GA > GB ? GA : GB); ++i) bsum[i] = 0.f;

  const int tiles_per = Ls / 64;
  const int nwaves = gridDim.x * 4, w0 = blockIdx.x * 4 + wave;
  f32x4 wq[DG ? CS : 1];  // w[cs][cb = q][0..3]
  TS* dsmall = nullptr;
  if constexpr (DG) {
    const float* w = blockIdx.y ? dg.w1 : dg.w0;
    dsmall = reinterpret_cast<TS*>(blockIdx.y ? dg.d1 : dg.d0);
#pragma unroll
    for (int cs = 0; cs < CS; ++cs) wq[cs] = *reinterpret_cast<const f32x4*>(w + ((long)cs * CB + q) * 4);
  }
  f32x4 ra[GA], rb[GB][4];
  auto load_tile = [&](int tile) {
    const int b = tile / tiles_per, j0 = (tile - b * tiles_per) * 64;
    const TS* sb = small + (long)b * s_bs + j0 + 4 * slot;
    const TB* bb = big + (long)b * big_bs + 4L * (j0 + 4 * slot) - pad;
#pragma unroll
    for (int a = 0; a < GA; ++a) ra[a] = Elem<TS>::ld4(sb + (long)(4 * a + q) * Ls);
    if (pad && j0 == 0 && slot == 0) {  // the window of position 0 starts one element before the row
#pragma unroll
      for (int g = 0; g < GB; ++g) {
        const TB* row = bb + pad + (long)(4 * g + q) * Lb;
        float e[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f32x4 v = Elem<TB>::ld4(row + 4 * i);
          e[4 * i] = v[0]; e[4 * i + 1] = v[1]; e[4 * i + 2] = v[2]; e[4 * i + 3] = v[3];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
          rb[g][i] = (f32x4){i == 0 ? 0.f : e[4 * i - 1], e[4 * i], e[4 * i + 1], e[4 * i + 2]};
      }
    } else {
#pragma unroll
      for (int g = 0; g < GB; ++g) {
        if constexpr (sizeof(TB) == 4) {
#pragma unroll
          for (int i = 0; i < 4; ++i) rb[g][i] = Elem<TB>::ld4(bb + (long)(4 * g + q) * Lb + 4 * i);
        } else if (!pad) {
#pragma unroll
          for (int i = 0; i < 4; ++i) rb[g][i] = Elem<TB>::ld4(bb + (long)(4 * g + q) * Lb + 4 * i);
        } else {
          // pad = 1: the windows start one element before an 8-byte boundary; two-byte elements cannot be loaded
          // four at a time from there, so the five aligned quads around them are loaded and shifted in registers
          const TB* al = bb + 1 + (long)(4 * g + q) * Lb;  // aligned: element 4 (j0 + 4 slot) of the row
          f32x4 e[5];
#pragma unroll
          for (int i = 0; i < 5; ++i) e[i] = Elem<TB>::ld4(al + 4 * (i - 1));
#pragma unroll
          for (int i = 0; i < 4; ++i) rb[g][i] = (f32x4){e[i][3], e[i + 1][0], e[i + 1][1], e[i + 1][2]};
        }
      }
    }
  };
  if (w0 < ntiles) load_tile(w0);
  for (int tile = w0; tile < ntiles; tile += nwaves) {
    f32x4 ca[GA], cb_[GB][4];
#pragma unroll
    for (int a = 0; a < GA; ++a) ca[a] = ra[a];
#pragma unroll
    for (int g = 0; g < GB; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) cb_[g][i] = rb[g][i];
    if (tile + nwaves < ntiles) load_tile(tile + nwaves);
    if (bias_from == 1) {
#pragma unroll
      for (int a = 0; a < GA; ++a) bsum[a] += (ca[a][0] + ca[a][1]) + (ca[a][2] + ca[a][3]);
    } else if (bias_from == 2) {
#pragma unroll
      for (int g = 0; g < GB; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) bsum[g] += (cb_[g][i][0] + cb_[g][i][1]) + (cb_[g][i][2] + cb_[g][i][3]);
    }
#pragma unroll
    for (int st = 0; st < 4; ++st)      // position 4*slot + st of the tile
#pragma unroll
      for (int tp = 0; tp < 4; ++tp)    // tap
#pragma unroll
        for (int a = 0; a < GA; ++a)
#pragma unroll
          for (int g = 0; g < GB; ++g)
            acc[a][g][tp] = __builtin_amdgcn_mfma_f32_4x4x1f32(ca[a][st], cb_[g][st][tp], acc[a][g][tp], 0, 0, 0);
    if constexpr (DG) {
      const bool odd = q & 1, hi = q & 2;
      f32x4 r2[2];  // channel 4m + q, this lane's 4 positions
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        float pc[CS];
#pragma unroll
        for (int cs = 0; cs < CS; ++cs) {
          const f32x4 gq = cb_[0][st];
          pc[cs] = fmaf(wq[cs][3], gq[3], fmaf(wq[cs][2], gq[2], fmaf(wq[cs][1], gq[1], wq[cs][0] * gq[0])));
        }
        float r1[4];  // after the pair exchange: channel 2k + (q & 1), summed over two big channels
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float send = odd ? pc[2 * k] : pc[2 * k + 1], keep = odd ? pc[2 * k + 1] : pc[2 * k];
          r1[k] = keep + quad_swap<0xB1>(send);  // quad_perm [1,0,3,2]
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const float send = hi ? r1[2 * m] : r1[2 * m + 1], keep = hi ? r1[2 * m + 1] : r1[2 * m];
          r2[m][st] = keep + quad_swap<0x4E>(send);  // quad_perm [2,3,0,1]
        }
      }
      const int b = tile / tiles_per, j0 = (tile - b * tiles_per) * 64;
      TS* db_ = dsmall + (long)b * dg.d_bs + j0 + 4 * slot;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        f32x4 o;
#pragma unroll
        for (int st = 0; st < 4; ++st) o[st] = r2[m][st] * elu_grad_from_out(ca[m][st]);
        Elem<TS>::st4(db_ + (long)(4 * m + q) * Ls, o);
      }
    }
  }
  // ---- 16 slots -> lane q of slot 0 (butterflies over lane bits 2..5), then the 4 waves through LDS
#pragma unroll
  for (int a = 0; a < GA; ++a)
#pragma unroll
    for (int g = 0; g < GB; ++g)
#pragma unroll
      for (int tp = 0; tp < 4; ++tp)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[a][g][tp][r];
          v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
          // lane (slot 0, q): dW[cs = 4a + r][cb = 4g + q][tap tp]
          if (slot == 0) comb[wave][((4 * a + r) * CB + 4 * g + q) * 4 + tp] = v;
        }
#pragma unroll
  for (int i = 0; i < (GA > GB ? GA : GB); ++i) {
    float v = bsum[i];
    v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
    if (slot == 0) comb[wave][NW + 4 * i + q] = v;  // channel 4i + q of small (bias_from 1) or big (2)
  }
  __syncthreads();
  float* out = partial + (size_t)blockIdx.x * SLAB;
  for (int i = t; i < SLAB; i += 256) {
    const bool live = i < NW || (bias_from == 1 && i - NW < CS) || (bias_from == 2 && i - NW < CB);
    out[i] = live ? (comb[0][i] + comb[1][i]) + (comb[2][i] + comb[3][i]) : 0.f;
  }
}

template <int CS, int CB, class TB = float, class TS = float>
__global__ __launch_bounds__(256) void conv1d_wgrad_stream_kernel(const float* __restrict__ small0,
                                                                  const float* __restrict__ small1, long s_bs,
                                                                  const float* __restrict__ big0,
                                                                  const float* __restrict__ big1, long big_bs,
                                                                  float* __restrict__ partial0,
                                                                  float* __restrict__ partial1, int Ls, int Lb,
                                                                  int pad, int bias_from, int ntiles) {
  conv1d_wgrad_stream_body<CS, CB, TB, false, TS>(small0, small1, s_bs, big0, big1, big_bs, partial0, partial1, Ls, Lb, pad,
                                                  bias_from, ntiles, DgradArgs{});
}
// weight + bias + data gradient of the 8 -> 4 channel transposed layer (204 registers: two wavefronts per SIMD; capped
// at 168 for three it spills 10 and runs 81 us instead of 61)
template <class TB, class TS = float>
__global__ __launch_bounds__(256) void conv1d_bwd_fused_kernel(const float* __restrict__ small0,
                                                                  const float* __restrict__ small1, long s_bs,
                                                                  const float* __restrict__ big0,
                                                                  const float* __restrict__ big1, long big_bs,
                                                                  float* __restrict__ partial0,
                                                                  float* __restrict__ partial1, int Ls, int Lb,
                                                                  int bias_from, int ntiles, DgradArgs dg) {
  conv1d_wgrad_stream_body<8, 4, TB, true, TS>(small0, small1, s_bs, big0, big1, big_bs, partial0, partial1, Ls, Lb, 0,
                                               bias_from, ntiles, dg);
}

bool conv1d_wgrad_stream_supported(int Cs, int Cb, int Ls, int Lb, int pad, int bias_from, long s_bs, long big_bs,
                                   const float* small, const float* big) {
  const bool shape = (Cs == 8 && Cb == 4) || (Cs == 12 && Cb == 8);
  return shape && Ls % 64 == 0 && Lb == 4 * Ls && (pad == 0 || pad == 1) && !(bias_from == 2 && pad != 0) &&
         s_bs % 4 == 0 && big_bs % 4 == 0 && (reinterpret_cast<uintptr_t>(small) & 15) == 0 &&
         (reinterpret_cast<uintptr_t>(big) & 15) == 0;
}
bool conv1d_bwd_fused_supported(int Cs, int Cb, int pad) {
  return (Cs == 8 && Cb == 4 && pad == 0) || conv1d_bwd_fused2_supported(Cs, Cb, pad);
}
// one slab of Cs*Cb*4 + 16 floats per workgroup at ws (and ws2 for the second problem); returns the grid size
int conv1d_wgrad_stream(const float* small, const float* small2, long s_bs, const float* big, const float* big2,
                        long big_bs, float* ws, float* ws2, int B, int Cs, int Cb, int Ls, int Lb, int pad,
                        int bias_from, int max_blocks, hipStream_t st, int* grid_out, int big_bf16,
                        const FusedDgrad* fd, int small_bf16) {
  // every fused layer but the original 8 -> 4 transposed one (which keeps its own kernel)
  if (fd && fd->dx && (!(Cs == 8 && Cb == 4 && pad == 0) || !fd->dact))
    return conv1d_bwd_fused2(small, small2, s_bs, big, big2, big_bs, ws, ws2, B, Cs, Cb, Ls, Lb, pad, max_blocks, st,
                             grid_out, big_bf16, *fd, small_bf16);
  const float* w = fd ? fd->w : nullptr;
  const float* w2 = fd ? fd->w2 : nullptr;
  float* dsmall = fd ? fd->dx : nullptr;
  float* dsmall2 = fd ? fd->dx2 : nullptr;
  const long d_bs = fd ? fd->dx_bs : 0;
  const DgradArgs dg{w, w2, dsmall, dsmall2, d_bs};
  if (dsmall && !(Cs == 8 && Cb == 4 && pad == 0 && w && (!small2 || (w2 && dsmall2)) && d_bs % 4 == 0 &&
                  (reinterpret_cast<uintptr_t>(dsmall) & 15) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0 &&
                  (!small2 || ((reinterpret_cast<uintptr_t>(dsmall2) & 15) == 0 && (reinterpret_cast<uintptr_t>(w2) & 15) == 0)))) {
    set_last_error("conv1d_wgrad_stream: fused data gradient needs the 8 -> 4 channel transposed layer, 16-byte aligned");
    return LSHM_ERR_UNSUPPORTED;
  }
  const int ntiles = (Ls / 64) * B;
  // measured at B=256: 4..8 tiles per wavefront and at most 512 workgroups per problem (more workgroups
  // only add closing butterflies and partial slabs)
  int grid = ntiles / 16;
  if (grid > 512) grid = 512;
  if (grid < 1) grid = 1;
  if (grid > max_blocks) grid = max_blocks;
  *grid_out = grid;
  const dim3 g(grid, small2 ? 2 : 1);
  // bf16 storage: `big` for both outer layer shapes, `small` (and the fused data gradient) for the outermost one
  if (small_bf16 && !(Cs == 8 && big_bf16)) { set_last_error("conv1d_wgrad_stream: a bf16 `small` needs the outermost layer with bf16 `big`"); return LSHM_ERR_UNSUPPORTED; }
#define LSHM_WGS(CS_, CB_, TB_, TS_) \
  hipLaunchKernelGGL((conv1d_wgrad_stream_kernel<CS_, CB_, TB_, TS_>), g, dim3(256), 0, st, small, small2, s_bs, big, big2, big_bs, ws, ws2, Ls, Lb, pad, bias_from, ntiles)
#define LSHM_BFK(TB_, TS_) \
  hipLaunchKernelGGL((conv1d_bwd_fused_kernel<TB_, TS_>), g, dim3(256), 0, st, small, small2, s_bs, big, big2, big_bs, ws, ws2, Ls, Lb, bias_from, ntiles, dg)
  if (dsmall) {
    if (big_bf16 && small_bf16) LSHM_BFK(bf16, bf16);
    else if (big_bf16) LSHM_BFK(bf16, float);
    else LSHM_BFK(float, float);
  } else if (Cs == 8) {
    if (big_bf16 && small_bf16) LSHM_WGS(8, 4, bf16, bf16);
    else if (big_bf16) LSHM_WGS(8, 4, bf16, float);
    else LSHM_WGS(8, 4, float, float);
  } else {
    if (big_bf16) LSHM_WGS(12, 8, bf16, float);
    else LSHM_WGS(12, 8, float, float);
  }
#undef LSHM_WGS
#undef LSHM_BFK
  return check_launch("conv1d_wgrad_stream");
}

static bool aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

bool tconv1d_stream_supported(const Conv1dDgradParams& p) {
  const bool shape = (p.Cs == 8 && p.Cb == 4 && (!p.s_bf16 || p.big_bf16)) || (p.Cs == 12 && p.Cb == 8 && !p.s_bf16);
  return shape && (p.pad == 0 || p.pad == 1) && p.Lb == 4 * p.Ls && p.big_bs % 4 == 0 && aligned16(p.big) &&
         aligned16(p.w) && (!p.dact || aligned16(p.dact));
}
bool conv1d_stream_supported(const Conv1dFwdParams& p) {
  const bool shape = (p.Cin == 4 && p.Cout == 8 && (!p.y_bf16 || p.x_bf16)) || (p.Cin == 8 && p.Cout == 12 && !p.y_bf16);
  return shape && (p.pad == 0 || p.pad == 1) && p.L == 4 * p.Lo && p.x_bs % 4 == 0 && aligned16(p.x) && aligned16(p.w);
}

int tconv1d_stream(const Conv1dDgradParams& p, const Conv1dDgradParams* p1, hipStream_t st) {
  const dim3 grid(cdiv((long)p.B * p.Ls, 256), p1 ? 2 : 1);
  const Conv1dDgradParams& q = p1 ? *p1 : p;
#define LSHM_LAUNCH(CS, CB, T, TS_)                                                                           \
  if (p.pad) hipLaunchKernelGGL((tconv1d_stream_kernel<CS, CB, true, T, TS_>), grid, dim3(256), 0, st, p, q);  \
  else hipLaunchKernelGGL((tconv1d_stream_kernel<CS, CB, false, T, TS_>), grid, dim3(256), 0, st, p, q)
  if (p.Cs == 8 && p.big_bf16 && p.s_bf16) { LSHM_LAUNCH(8, 4, bf16, bf16); }
  else if (p.Cs == 8 && p.big_bf16) { LSHM_LAUNCH(8, 4, bf16, float); }
  else if (p.Cs == 8) { LSHM_LAUNCH(8, 4, float, float); }
  else if (p.big_bf16) { LSHM_LAUNCH(12, 8, bf16, float); }
  else { LSHM_LAUNCH(12, 8, float, float); }
#undef LSHM_LAUNCH
  return check_launch("tconv1d_stream");
}
int conv1d_stream(const Conv1dFwdParams& p, const Conv1dFwdParams* p1, hipStream_t st) {
  const dim3 grid(cdiv((long)p.B * p.Lo, 256), p1 ? 2 : 1);
  const Conv1dFwdParams& q = p1 ? *p1 : p;
#define LSHM_LAUNCH(CI, CO, T, TO_)                                                                           \
  if (p.pad) hipLaunchKernelGGL((conv1d_stream_kernel<CI, CO, true, T, TO_>), grid, dim3(256), 0, st, p, q);   \
  else hipLaunchKernelGGL((conv1d_stream_kernel<CI, CO, false, T, TO_>), grid, dim3(256), 0, st, p, q)
  if (p.Cin == 4 && p.x_bf16 && p.y_bf16) { LSHM_LAUNCH(4, 8, bf16, bf16); }
  else if (p.Cin == 4 && p.x_bf16) { LSHM_LAUNCH(4, 8, bf16, float); }
  else if (p.Cin == 4) { LSHM_LAUNCH(4, 8, float, float); }
  else if (p.x_bf16) { LSHM_LAUNCH(8, 12, bf16, float); }
  else { LSHM_LAUNCH(8, 12, float, float); }
#undef LSHM_LAUNCH
  return check_launch("conv1d_stream");
}

}  // namespace lshm
