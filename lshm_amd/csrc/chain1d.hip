// LDS-resident chains of three k4 s4 1-D layers (the mid layers of AutoEncoder1DCNN, src/lofar_models.py:119-123
// conv2 -> conv3 -> conv4 and :137-140 tconv1 -> tconv2 -> tconv3, and the data-gradient passes through the same
// layers in the opposite direction): one workgroup per (patch, problem), the activations of the patch stay in LDS
// from layer to layer, weights stream from L2 (97 KB per chain), every layer's output also goes to HBM once
// (coalesced float4 rows; the backward pass and the weight gradients need it).  One launch instead of three --
// these layers move < 10 MB each and were bound by the fixed cost of a launch (~5 us of dispatch, prologue and
// drain for 10-14 us kernels).
//
// Both directions are per-position GEMMs on v_mfma_f32_16x16x4_f32 (exact fp32):
//   down (stride-4 conv; forward of conv2..4, data gradient of tconv3..1):
//       Y[n][j] = epi( sum_{ci,t} X[ci][4j - pad + t] W[n][ci*4 + t] )          M = positions, N = Cout, K = 4 Cin
//     A fragments are single LDS words of the input image (row pitch == 1 mod 4: conflict-free), B fragments one
//     float4 of the weight row per four matrix instructions (k = 16 s + 4 lk + e <-> channel 4 s + lk, tap e);
//   up (stride-4 transposed conv; forward of tconv1..3, data gradient of conv4..2):
//       Y[co][4i + t - pad] = epi( sum_ci X[ci][i] W[ci][co*4 + t] )            M = positions, N = 4 Cout, K = Cin
//     the four taps of an output channel are four accumulators of one lane (n-tile e <-> tap e, lane <-> channel),
//     so one float4 of the weights feeds four matrix instructions and a lane leaves whole float4s of the output row.
// Epilogue per stage: bias + ELU (forward) or the ELU' multiply by the saved activation of that tensor (data
// gradient; applied in the copy-out pass, which reads the saved tensor with the same coalesced float4s).
#include <stdlib.h>

#include "kernels.h"
#include "chain1d_dev.h"

namespace lshm {

// conv2 -> conv3 -> conv4 geometry: C0 x L0 -> C1 x L0/4 -> C2 x L0/16 -> C3 x L0/64
template <int C0, int C1, int C2, int C3, int L0>
__global__ __launch_bounds__(kChainThreads) void conv1d_chain_down_kernel(const Chain1dArgs a) {
  constexpr int P0 = pitch_down(L0), P1 = pitch_down(L0 / 4), P2 = pitch_down(L0 / 16), P3 = pitch_down(L0 / 64);
  static_assert(P0 % 4 == 1 && P1 % 4 == 1 && P2 % 4 == 1 && P3 % 4 == 1, "conflict-free A fragments");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // x0 is dead once stage 1 has run: the images of stages 2 and 3 take its place (74 KB per workgroup: two per CU)
  static_assert(C2 * P2 + C3 * P3 <= C0 * P0, "stage-2 / stage-3 images fit in the input image");
  float* x0 = smem;
  float* y1 = x0 + C0 * P0;
  float* y2 = x0;
  float* y3 = y2 + C2 * P2;
  const int b = blockIdx.x, pr = blockIdx.y, t = threadIdx.x;
  // front slots (position -1: the left padding) of every image read by a stage
  for (int c = t; c < C0; c += kChainThreads) x0[c * P0] = 0.f;
  for (int c = t; c < C1; c += kChainThreads) y1[c * P1] = 0.f;
  const float* in = a.in[pr] + (long)b * a.in_bs;
#pragma unroll 4
  for (int i = t; i < C0 * L0; i += kChainThreads) {  // one element per lane: coalesced loads, conflict-free LDS stores
    const int c = i / L0, q = i - c * L0;
    x0[c * P0 + q + 1] = in[(long)c * L0 + q];
  }
  const float* da0 = a.st[0].dact[pr] ? a.st[0].dact[pr] + (long)b * a.st[0].out_bs : nullptr;
  const float* da1 = a.st[1].dact[pr] ? a.st[1].dact[pr] + (long)b * a.st[1].out_bs : nullptr;
  const float* da2 = a.st[2].dact[pr] ? a.st[2].dact[pr] + (long)b * a.st[2].out_bs : nullptr;
  DactRegs<C1, L0 / 4> r1;
  DactRegs<C2, L0 / 16> r2;
  DactRegs<C3, L0 / 64> r3;
  load_dact(r1, da0);
  __syncthreads();
  down_stage<C0, C1, L0, P0, P1>(x0, y1, a.st[0].w[pr], a.st[0].bias[pr], a.st[0].act, a.pad);
  __syncthreads();
  load_dact(r2, da1);
  copy_out<C1, L0 / 4, P1>(y1, 1, a.st[0].out[pr] + (long)b * a.st[0].out_bs, da0, false, r1);
  for (int c = t; c < C2; c += kChainThreads) y2[c * P2] = 0.f;  // (x0 is dead: the barrier above followed its last read)
  __syncthreads();
  down_stage<C1, C2, L0 / 4, P1, P2>(y1, y2, a.st[1].w[pr], a.st[1].bias[pr], a.st[1].act, a.pad);
  __syncthreads();
  load_dact(r3, da2);
  copy_out<C2, L0 / 16, P2>(y2, 1, a.st[1].out[pr] + (long)b * a.st[1].out_bs, da1, false, r2);
  __syncthreads();
  down_stage<C2, C3, L0 / 16, P2, P3>(y2, y3, a.st[2].w[pr], a.st[2].bias[pr], a.st[2].act, a.pad);
  __syncthreads();
  copy_out<C3, L0 / 64, P3>(y3, 1, a.st[2].out[pr] + (long)b * a.st[2].out_bs, da2, false, r3);
}

// tconv1 -> tconv2 -> tconv3 geometry: C0 x L0 -> C1 x 4 L0 -> C2 x 16 L0 -> C3 x 64 L0
template <int C0, int C1, int C2, int C3, int L0>
__global__ __launch_bounds__(kChainThreads) void conv1d_chain_up_kernel(const Chain1dArgs a) {
  constexpr int P0 = pitch_up(L0), P1 = pitch_up(4 * L0), P2 = pitch_up(16 * L0), P3 = pitch_up(64 * L0);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // the input image and the stage-1 image are dead when stage 3 writes: they live inside y3's space (76 KB per workgroup)
  static_assert(C0 * P0 + C1 * P1 <= C3 * P3, "input + stage-1 images fit in the stage-3 image");
  float* y3 = smem;
  float* x0 = y3;
  float* y1 = x0 + C0 * P0;
  float* y2 = y3 + C3 * P3;
  const int b = blockIdx.x, pr = blockIdx.y, t = threadIdx.x;
  const int pad = a.pad;  // pad 1: logical position p of an output image sits at index p + 1
  const float* in = a.in[pr] + (long)b * a.in_bs;
  for (int i = t; i < C0 * (L0 / 4); i += kChainThreads) {
    const int c = i / (L0 / 4), c4 = i - c * (L0 / 4);
    *reinterpret_cast<f32x4*>(x0 + c * P0 + 4 * c4) = *reinterpret_cast<const f32x4*>(in + (long)c * L0 + 4 * c4);
  }
  const float* da0 = a.st[0].dact[pr] ? a.st[0].dact[pr] + (long)b * a.st[0].out_bs : nullptr;
  const float* da1 = a.st[1].dact[pr] ? a.st[1].dact[pr] + (long)b * a.st[1].out_bs : nullptr;
  const float* da2 = a.st[2].dact[pr] ? a.st[2].dact[pr] + (long)b * a.st[2].out_bs : nullptr;
  DactRegs<C1, 4 * L0> r1;
  DactRegs<C2, 16 * L0> r2;
  DactRegs<C3, 64 * L0> r3;
  load_dact(r1, da0);
  __syncthreads();
  up_stage<C0, C1, L0, P0, P1>(x0, 0, y1, a.st[0].w[pr], a.st[0].bias[pr], a.st[0].act);
  __syncthreads();
  load_dact(r2, da1);
  copy_out<C1, 4 * L0, P1>(y1, pad, a.st[0].out[pr] + (long)b * a.st[0].out_bs, da0, pad != 0, r1);
  __syncthreads();
  up_stage<C1, C2, 4 * L0, P1, P2>(y1, pad, y2, a.st[1].w[pr], a.st[1].bias[pr], a.st[1].act);
  __syncthreads();
  load_dact(r3, da2);
  copy_out<C2, 16 * L0, P2>(y2, pad, a.st[1].out[pr] + (long)b * a.st[1].out_bs, da1, pad != 0, r2);
  __syncthreads();
  up_stage<C2, C3, 16 * L0, P2, P3>(y2, pad, y3, a.st[2].w[pr], a.st[2].bias[pr], a.st[2].act);
  __syncthreads();
  copy_out<C3, 64 * L0, P3>(y3, pad, a.st[2].out[pr] + (long)b * a.st[2].out_bs, da2, pad != 0, r3);
}

static size_t lds_down(int c0, int c1, int c2, int c3, int l0) {
  return sizeof(float) * ((size_t)c0 * pitch_down(l0) + (size_t)c1 * pitch_down(l0 / 4));  // stages 2, 3 alias the input image
}
static size_t lds_up(int c0, int c1, int c2, int c3, int l0) {
  return sizeof(float) * ((size_t)c2 * pitch_up(16 * l0) + (size_t)c3 * pitch_up(64 * l0));  // input + stage 1 alias the stage-3 image
}

// the two chains of AutoEncoder1DCNN's mid layers: down 12 -> 24 -> 48 -> 96 channels from 1024 positions, up
// 96 -> 48 -> 24 -> 12 channels from 16 positions
bool conv1d_chain_supported(bool up, const int* ch, int L0) {
  if (sched(LSHM_SCHED_NO_CHAIN1D)) return false;
  // (the LDS a chain needs is part of the answer: on a device that cannot hold it the plan keeps the three launches)
  if (!up) return ch[0] == 12 && ch[1] == 24 && ch[2] == 48 && ch[3] == 96 && L0 == 1024 && device_lds_fits(lds_down(12, 24, 48, 96, 1024));
  return ch[0] == 96 && ch[1] == 48 && ch[2] == 24 && ch[3] == 12 && L0 == 16 && device_lds_fits(lds_up(96, 48, 24, 12, 16));
}

int conv1d_chain(bool up, const Chain1dStage* st, const float* in0, const float* in1, long in_bs, int pad, int B, hipStream_t s) {
  Chain1dArgs a;
  const bool two = in1 != nullptr;
  a.in[0] = in0; a.in[1] = two ? in1 : in0;
  a.in_bs = in_bs;
  a.pad = pad;
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  bool ok = al16(in0) && (!two || al16(in1)) && in_bs % 4 == 0 && (pad == 0 || pad == 1);
  for (int i = 0; i < 3; ++i) {
    for (int g = 0; g < 2; ++g) {
      const int q = two ? g : 0;
      a.st[i].w[g] = st[i].w[q]; a.st[i].bias[g] = st[i].bias[q]; a.st[i].out[g] = st[i].out[q]; a.st[i].dact[g] = st[i].dact[q];
      ok = ok && st[i].w[q] && st[i].out[q] && al16(st[i].w[q]) && al16(st[i].out[q]) && al16(st[i].dact[q]);
    }
    a.st[i].out_bs = st[i].out_bs;
    a.st[i].act = st[i].act;
    ok = ok && st[i].out_bs % 4 == 0;
  }
  if (!ok) { set_last_error("conv1d_chain: null / unaligned pointer or stride"); return LSHM_ERR_ARG; }
  const dim3 grid(B, two ? 2 : 1);
  int rc;
  if (!up) {
    const size_t lds = lds_down(12, 24, 48, 96, 1024);
    auto kern = conv1d_chain_down_kernel<12, 24, 48, 96, 1024>;
    if ((rc = kernel_budget_ok(reinterpret_cast<const void*>(kern), kChainThreads, lds, "conv1d chain (down)"))) return rc;
    if ((rc = raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds, "conv1d chain (down)"))) return rc;
    hipLaunchKernelGGL(kern, grid, dim3(kChainThreads), lds, s, a);
  } else {
    const size_t lds = lds_up(96, 48, 24, 12, 16);
    auto kern = conv1d_chain_up_kernel<96, 48, 24, 12, 16>;
    if ((rc = kernel_budget_ok(reinterpret_cast<const void*>(kern), kChainThreads, lds, "conv1d chain (up)"))) return rc;
    if ((rc = raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds, "conv1d chain (up)"))) return rc;
    hipLaunchKernelGGL(kern, grid, dim3(kChainThreads), lds, s, a);
  }
  return check_launch("conv1d_chain");
}

}  // namespace lshm
