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

namespace lshm {

// threads per workgroup: 16 wavefronts share the tiles of a stage (32 / 12 / 6 of them), so a wavefront's serial chain of
// weight fetch -> matrix instructions is 1-2 tiles long; two workgroups (74-76 KB of LDS each) per CU
constexpr int kChainThreads = 1024;

struct Chain1dArgs {
  const float* in[2];
  long in_bs;
  Chain1dStage st[3];
  int pad;  // down: left padding of the windows (1: forward of conv, 0: data gradient of the transposed conv)
            // up: 1 shifts the output one position to the left (data gradient of conv), 0: forward of the transposed conv
};

// ---- down stage: X (CIN x LIN, LDS image `xs`, pitch PIN, element p at xs[c*PIN + p + 1]) -> Y (COUT x LIN/4) into
// the LDS image `ys` (pitch POUT, same +1 convention); bias / ELU applied
template <int CIN, int COUT, int LIN, int PIN, int POUT>
__device__ __forceinline__ void down_stage(const float* __restrict__ xs, float* __restrict__ ys, const float* __restrict__ w,
                                           const float* __restrict__ bias, int act, int pad) {
  constexpr int K = CIN * 4, LOUT = LIN / 4;
  constexpr int MT = LOUT / 16, NT = (COUT + 15) / 16;
  static_assert(K % 16 == 0 && LOUT % 16 == 0, "whole k-blocks and m-tiles");
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lm = lane & 15, lk = lane >> 4;
  for (int f = wave; f < MT * NT; f += kChainThreads / 64) {
    const int nt = f % NT, mt = f / NT;  // consecutive tiles of a wavefront keep their weight rows in L1
    const int n = 16 * nt + lm;
    const bool nok = n < COUT;
    const float* wrow = w + (long)(nok ? n : 0) * K + 4 * lk;
    const float* arow = xs + lk * PIN + 4 * (16 * mt + lm) + 1 - pad;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // all B fragments of the tile first (independent loads, one L2 round trip), then the matrix instructions
    f32x4 bq[K / 16];
#pragma unroll
    for (int s = 0; s < K / 16; ++s) bq[s] = nok ? *reinterpret_cast<const f32x4*>(wrow + 16 * s) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < K / 16; ++s) {
      const float* ap = arow + 4 * s * PIN;
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[e], bq[s][e], acc, 0, 0, 0);
    }
    if (nok) {
      const float bv = bias ? bias[n] : 0.f;
      float* yp = ys + n * POUT + 16 * mt + 4 * lk + 1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = acc[r] + bv;
        yp[r] = act ? elu(v) : v;
      }
    }
  }
}

// ---- up stage: X (CIN x LIN, LDS image, element p at xs[c*PIN + p + xoff]) -> Y (COUT x 4 LIN) into `ys`
// (pitch POUT, multiple of 4): the value for logical position 4i + t - pad is stored at ys[co*POUT + 4i + t]
template <int CIN, int COUT, int LIN, int PIN, int POUT>
__device__ __forceinline__ void up_stage(const float* __restrict__ xs, int xoff, float* __restrict__ ys,
                                         const float* __restrict__ w, const float* __restrict__ bias, int act) {
  constexpr int MT = LIN / 16, CBT = (COUT + 15) / 16;
  static_assert(CIN % 4 == 0 && LIN % 16 == 0 && POUT % 4 == 0, "whole k-steps and m-tiles, float4 rows");
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lm = lane & 15, lk = lane >> 4;
  for (int f = wave; f < MT * CBT; f += kChainThreads / 64) {
    const int cb = f % CBT, mt = f / CBT;
    const int co = 16 * cb + lm;
    const bool cok = co < COUT;
    const float* wp = w + ((long)lk * COUT + (cok ? co : 0)) * 4;
    const float* ap = xs + lk * PIN + 16 * mt + lm + xoff;
    f32x4 acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // w[ci = 4s + lk][co][0..3]: the tile's B fragments in flight at once, at most 8 k-steps (32 registers) per batch so that
    // the kernel fits 64 VGPRs = two 16-wavefront workgroups per CU
    constexpr int KS = CIN / 4, KC = KS % 8 == 0 ? 8 : (KS % 6 == 0 ? 6 : KS);
    static_assert(KS % KC == 0, "whole batches of k-steps");
#pragma unroll
    for (int s0 = 0; s0 < KS; s0 += KC) {
      f32x4 bq[KC];
#pragma unroll
      for (int s = 0; s < KC; ++s) bq[s] = cok ? *reinterpret_cast<const f32x4*>(wp + (long)16 * (s0 + s) * COUT) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KC; ++s) {
        const float a = ap[4 * (s0 + s) * PIN];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bq[s][e], acc[e], 0, 0, 0);
      }
    }
    if (cok) {
      const float bv = bias ? bias[co] : 0.f;
      float* yp = ys + co * POUT + 4 * (16 * mt + 4 * lk);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        f32x4 o = {acc[0][r] + bv, acc[1][r] + bv, acc[2][r] + bv, acc[3][r] + bv};
        if (act) {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = elu(o[e]);
        }
        *reinterpret_cast<f32x4*>(yp + 4 * r) = o;
      }
    }
  }
}

// copy-out pass: rows of the LDS image (C x L, logical element p at img[c*P + p + off]) -> global (c-major rows of L),
// with the optional ELU' multiply written back to the image for the next stage.  `zero_last`: the last logical
// element has no source (pad-1 up stage) and is written as zero.
// The ELU' inputs of a data-gradient stage are requested BEFORE the stage's matrix instructions (load_dact) and used in
// its copy-out: their HBM round trip (~2 us, three times per sample) hides behind the stage instead of following it.
template <int C, int L>
struct DactRegs {
  static constexpr int N = (C * L + kChainThreads - 1) / kChainThreads;
  float v[N];
};
template <int C, int L>
__device__ __forceinline__ void load_dact(DactRegs<C, L>& r, const float* __restrict__ dact) {
  if (!dact) return;
#pragma unroll
  for (int k = 0; k < DactRegs<C, L>::N; ++k) {
    const int i = threadIdx.x + k * kChainThreads;
    r.v[k] = i < C * L ? dact[i] : 0.f;  // the same flat order as copy_out
  }
}
template <int C, int L, int P>
__device__ __forceinline__ void copy_out(float* __restrict__ img, int off, float* __restrict__ out, const float* __restrict__ dact,
                                         bool zero_last, const DactRegs<C, L>& r) {
  // one element per lane: consecutive lanes read consecutive LDS words (no bank conflicts; float4 rows of the
  // odd-pitched images were 4-way conflicts, 57-65 % of the LDS cycles of the first version) and store 256-byte runs
#pragma unroll
  for (int k = 0; k < DactRegs<C, L>::N; ++k) {
    const int i = threadIdx.x + k * kChainThreads;
    if (i < C * L) {
      const int c = i / L, q = i - c * L;
      float* p = img + c * P + q + off;
      float v = (zero_last && q == L - 1) ? 0.f : *p;
      if (dact) v *= elu_grad_from_out(r.v[k]);
      if (dact || (zero_last && q == L - 1)) *p = v;
      out[(long)c * L + q] = v;
    }
  }
}

constexpr int pitch_down(int L) { return L + 1 + ((4 - (L + 1) % 4) % 4 + 1) % 4; }  // >= L + 1, == 1 (mod 4)
constexpr int pitch_up(int L) {  // smallest pitch >= L + 1 that is 16 or 48 (mod 64): A fragments of the four k-lanes hit four bank groups
  int p = L + 1;
  while (p % 64 != 16 && p % 64 != 48) ++p;
  return p;
}

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
  static const bool off = getenv("LSHM_CHAIN_OFF") != nullptr;
  if (off) return false;
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
