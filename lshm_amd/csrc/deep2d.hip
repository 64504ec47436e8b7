// The deep section of AutoEncoderCNN2's forward as ONE launch (round 4): conv3 -> conv4 -> conv5 -> fc1 -> fc2in ->
// fc2out -> fc3 -> tconv0 -> tconv1 -> tconv2 -> tconv3 (src/lofar_models.py:36-41,43-51,52-55 and the forward :66-69,73-98).
//
// Eleven launches of 6-35 us for < 10 MB each were the longest latency-bound stretch of a forward (~230 us of the 2-D
// autoencoder's ~420, profiles/r03/step_timeline.txt).  Here a workgroup takes G patches through all eleven layers:
// activations never leave LDS between layers (every layer's output is also written to HBM once -- the backward pass
// and the weight gradients read it), the 4.9 MB of weights stream from L2.  DESIGN.md (round 3) rejected this with
// "302 MB of L2 traffic per layer"; measured (profiles/r04/l2_broadcast_probe.txt), 256 workgroups streaming the same
// 4.9 MB take 46 us (27 TB/s of L2 reads, ~50 bytes per clock and CU) -- a quarter of the launches they replace.
//
// Weight stream.  The layers' own layouts are not streamable at small M: with lanes along output channels a lane's
// float4 along K sits 3-6 KB from its neighbour's (64 cache lines per wave instruction, an eighth of each used).
// deep2d_pack_kernel therefore re-orders the weights once per forward into FRAGMENT ORDER: for every (stage, unit,
// step) the 64 lanes' float4s are 1 KiB of consecutive memory, in the order the consumer's matrix instructions take
// them (5.1 MB, ~5 us, at the head of the forward's stream; pack_source is the one definition of the order).  A "unit"
// is the work of one wavefront in one stage: an output-channel tile x a slice of K, all rows.  A wavefront keeps NB
// steps in flight beside the NB it consumes, and requests the first NB steps of its next stage BEFORE the barrier and
// the reduction that end the current one.
//
// Arithmetic.  Stages with >= 16 rows per patch (conv3, conv4, tconv1..3) run on v_mfma_f32_16x16x4_f32, the 4-row
// stages (conv5: 2x2 positions; tconv0: 2x2 positions per output parity) on v_mfma_f32_4x4x1_16b_f32 -- 16 independent
// 4x4 blocks per instruction, so every lane is useful at M = 4 --, the dense layers (one row per patch) on the vector
// ALU (an fmaf chain per lane = output column).  K is split over wavefronts; the partial tiles meet in LDS slabs that a
// second phase adds in a FIXED order (the bias rides in the first slice's accumulators), then ELU: bitwise
// reproducible, no float atomics, and the same bits whatever G / the workgroup size.
// Transposed convolutions are computed per output parity (2x2 taps each, no structural zeros); taps are taken in
// POSITIONAL order (lower input row / column first) so that the two column taps of a lane are one 8-byte LDS word.
//
// LDS plan per patch (floats; regions are reused as their contents die):
//   RA 7776: conv2 output 24x(16+2)x(16+2) -> every stage's slab -> tconv2's output in the same padded form
//   RS 6144: conv3 slab -> conv4 output (96 x 6 rows of two 4-wide windows) -> fc3 output (192 x 4 rows of column
//            pairs) -> tconv0 output 96x(6x8+4)
//   RX 4800: conv3 output 48x10x10 -> tconv1 output 48x10x10
//   RV 1536: cat1 (784) | z1 (224) | mu (224) | cat3 (240)
// 20,256 floats = 79 KB per patch: G = 2 fits the 160 KB of a CU.
#include <stdlib.h>

#include "kernels.h"
#include "deep2d.h"

namespace lshm {
namespace {

constexpr int kL = 224;   // latent width of the 2-D autoencoder (src/kharmonic_lofar.py:37: 256 - 2 x 16)
constexpr int kHd = 16;   // harmonic features (4 scales)

// ---- packed weights: stage bases (floats); one step of one region = 64 lanes x float4 = 256 floats
constexpr long kStep = 256;
constexpr long P_CONV3 = 0;                         // 6 regions (n-tile, k-half) x 12 steps (channel)
constexpr long P_CONV4 = P_CONV3 + 6 * 12 * kStep;  // 12 regions (n-tile, k-half) x 24 steps
constexpr long P_CONV5 = P_CONV4 + 12 * 24 * kStep; // 12 regions (64 channels, k-quarter) x 96 steps (channel, kernel row)
constexpr long P_FC1 = P_CONV5 + 12 * 96 * kStep;   // 16 regions (64 columns, k-quarter) x 49 steps
constexpr long P_FC2IN = P_FC1 + 16 * 49 * kStep;   // 16 x 14
constexpr long P_FC2OUT = P_FC2IN + 16 * 14 * kStep;
constexpr long P_FC3 = P_FC2OUT + 16 * 14 * kStep;  // 48 regions (64 columns of 768, k-quarter) x 15 steps
constexpr long P_TCONV0 = P_FC3 + 48 * 15 * kStep;  // 12 regions (16 channels, k-half) x 96 steps (input channel)
constexpr long P_TCONV1 = P_TCONV0 + 12 * 96 * kStep;  // 12 regions (16 channels, row parity, k-half) x 24 steps (4 input channels, row tap)
constexpr long P_TCONV2 = P_TCONV1 + 12 * 24 * kStep;  // 8 regions (16 channels, parity) x 12 steps (4 input channels)
constexpr long P_TCONV3 = P_TCONV2 + 8 * 12 * kStep;   // 2 regions (row parity) x 12 steps (4 input channels, row tap)
constexpr long P_TOTAL = P_TCONV3 + 2 * 12 * kStep;

// transposed conv k4 s2 p1, taps in positional order: output row 2a + p takes input rows a + p - 1 + j (j = 0, 1) through
// kernel rows {3, 1} (p = 0) / {2, 0} (p = 1)  [oy = 2 iy - 1 + ky]
__device__ __forceinline__ int tap_k(int p, int j) { return p ? (j ? 0 : 2) : (j ? 1 : 3); }

struct PackSrc {
  const float *c3, *c4, *c5, *fc1, *fc2in, *fc2out, *fc3, *t0, *t1, *t2, *t3;
};

// source element of packed float `off`; nullptr: padding (zero)
__device__ __forceinline__ const float* pack_source(const PackSrc& w, long off) {
  const int e = (int)(off & 3), lane = (int)((off >> 2) & 63);
  const int lm = lane & 15, lk = lane >> 4;
  if (off < P_CONV4) {            // conv3: W (48, 24, 4, 4)
    const int q = (int)((off - P_CONV3) / kStep), r = q / 12, s = q % 12;
    const int nt = r >> 1, kg = r & 1, ci = 12 * kg + s, n = 16 * nt + lm;
    return w.c3 + ((n * 24 + ci) * 16 + lk * 4 + e);
  }
  if (off < P_CONV5) {            // conv4: W (96, 48, 4, 4)
    const int q = (int)((off - P_CONV4) / kStep), r = q / 24, s = q % 24;
    const int nt = r >> 1, kg = r & 1, ci = 24 * kg + s, n = 16 * nt + lm;
    return w.c4 + ((n * 48 + ci) * 16 + lk * 4 + e);
  }
  if (off < P_FC1) {              // conv5: W (192, 96, 4, 4)
    const int q = (int)((off - P_CONV5) / kStep), r = q / 96, s = q % 96;
    const int ng = r >> 2, kg = r & 3, ci = 24 * kg + (s >> 2), ky = s & 3, n = 64 * ng + lane;
    return w.c5 + ((n * 96 + ci) * 16 + ky * 4 + e);
  }
  if (off < P_TCONV0) {           // the dense layers: W (N, K) row-major
    const float* base;
    long o;
    int N, K, SPK;
    if (off < P_FC2IN) { base = w.fc1; o = off - P_FC1; N = kL; K = 768 + kHd; SPK = 49; }
    else if (off < P_FC2OUT) { base = w.fc2in; o = off - P_FC2IN; N = kL; K = kL; SPK = 14; }
    else if (off < P_FC3) { base = w.fc2out; o = off - P_FC2OUT; N = kL; K = kL; SPK = 14; }
    else { base = w.fc3; o = off - P_FC3; N = 768; K = kL + kHd; SPK = 15; }
    const int q = (int)(o / kStep), r = q / SPK, s = q % SPK;
    const int ng = r >> 2, kg = r & 3, n = 64 * ng + lane, k = 4 * (kg * SPK + s) + e;
    return n < N ? base + ((long)n * K + k) : nullptr;
  }
  if (off < P_TCONV1) {           // tconv0: W (192, 96, 4, 4); lane = (parity, 16 channels), e = (row tap, column tap)
    const int q = (int)((off - P_TCONV0) / kStep), r = q / 96, s = q % 96;
    const int c = r >> 1, kg = r & 1, ci = 96 * kg + s;
    const int p = lane >> 4, co = 16 * c + (lane & 15);
    const int ky = tap_k(p >> 1, e >> 1), kx = tap_k(p & 1, e & 1);
    return w.t0 + ((ci * 96 + co) * 16 + ky * 4 + kx);
  }
  if (off < P_TCONV2) {           // tconv1: W (96, 48, 4, 4); region = (channel tile, row parity, k-half), step = (4 input channels, row tap)
    const int q = (int)((off - P_TCONV1) / kStep), r = q / 24, s = q % 24;
    const int kg = r & 1, py = (r >> 1) & 1, c = r >> 2;
    const int ci = 4 * (12 * kg + (s >> 1)) + lk, jy = s & 1, co = 16 * c + lm;
    return w.t1 + ((ci * 48 + co) * 16 + tap_k(py, jy) * 4 + e);
  }
  if (off < P_TCONV3) {           // tconv2: W (48, 24, 4, 4); region = (channel tile, parity), step = 4 input channels
    const int q = (int)((off - P_TCONV2) / kStep), r = q / 12, s = q % 12;
    const int c = r >> 2, p = r & 3, ci = 4 * s + lk, co = 16 * c + lm;
    const int ky = tap_k(p >> 1, e >> 1), kx = tap_k(p & 1, e & 1);
    return co < 24 ? w.t2 + ((ci * 24 + co) * 16 + ky * 4 + kx) : nullptr;
  }
  {                               // tconv3: W (24, 12, 4, 4); region = row parity, step = (4 input channels, row tap)
    const int q = (int)((off - P_TCONV3) / kStep), py = q / 12, s = q % 12;
    const int ci = 4 * (s >> 1) + lk, jy = s & 1, co = lm;
    return co < 12 ? w.t3 + ((ci * 12 + co) * 16 + tap_k(py, jy) * 4 + e) : nullptr;
  }
}

__global__ __launch_bounds__(256) void deep2d_pack_kernel(const PackSrc w, float* __restrict__ packed) {
  const long i4 = (long)blockIdx.x * 256 + threadIdx.x;  // one float4 of the packed image per thread
  if (i4 * 4 >= P_TOTAL) return;
  f32x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float* s = pack_source(w, i4 * 4 + e);
    v[e] = s ? *s : 0.f;
  }
  *reinterpret_cast<f32x4*>(packed + i4 * 4) = v;
}

struct Deep2dArgs {
  const float* wp;                 // packed weights (P_TOTAL floats)
  const float* x2;                 // conv2 output (B, 24, 16, 16)
  const float *b3, *b4, *b5, *bfc1, *bfc2in, *bfc2out, *bfc3, *bt0, *bt1, *bt2, *bt3;
  float *a3, *a4;                  // conv3 / conv4 outputs (B, 48, 8, 8), (B, 96, 4, 4)
  float* cat1;                     // (B, 784): [0, 768) conv5 output, written; [768, 784) elu(fcuv1), read
  float* z1;                       // (B, 224)
  float* mu; long mu_ld;           // (B, 224) inside Mu
  float* cat3;                     // (B, 240): [0, 224) elu(fc2out), written; [224, 240) elu(fcuv3), read
  float* d0;                       // (B, 768) fc3 output (no activation)
  float *t0, *t1, *t2, *t3;        // tconv0..3 outputs (B, 96, 4, 4), (B, 48, 8, 8), (B, 24, 16, 16), (B, 12, 32, 32)
  int B;
  long long* stamps;               // diagnostics (or null): shader-clock readings of workgroup 0 at the stage boundaries
};

// LDS regions (floats per patch)
constexpr int RA_F = 7776, RS_F = 6144, RX_F = 4800, RV_F = 1536;
constexpr int V_CAT1 = 0, V_Z1 = 784, V_MU = 1008, V_CAT3 = 1232;
constexpr int X2_CP = 324, X2_RP = 18;   // conv2 / tconv2 output image: channel / row pitch (16 + 2 border)
constexpr int X3_CP = 100, X3_RP = 10;   // 8 + 2
constexpr int X4_CP = 48;                // 6 padded rows x [cols -1..2 | cols 1..4]
constexpr int D0_CP = 32;                // 4 padded rows x [cols (0,1) | (1,2) | (2,3) | -]
constexpr int T0_CP = 52, T0_RP = 8;     // 4 + 2 rows of 8 (+4): k-lanes land on disjoint bank groups
constexpr int T1_CP = 100, T1_RP = 10;
constexpr int kPF = 8;                   // float4 registers of the cross-barrier weight prefetch

template <int NB>
__device__ __forceinline__ void load_batch(f32x4 (&q)[NB], const f32x4* __restrict__ p, int s0, int nsteps) {
#pragma unroll
  for (int k = 0; k < NB; ++k) q[k] = s0 + k < nsteps ? p[(long)(s0 + k) * 64] : (f32x4){0.f, 0.f, 0.f, 0.f};
}
// the first NB steps of a wavefront's first unit of a stage, requested ahead of the barrier in front of the stage
template <int NB>
__device__ __forceinline__ void prefetch(f32x4 (&pf)[kPF], const f32x4* __restrict__ p, bool on) {
  static_assert(NB <= kPF, "prefetch registers");
  if (on) {
#pragma unroll
    for (int k = 0; k < NB; ++k) pf[k] = p[(long)k * 64];
  }
}
// NS steps of one unit: body(step, float4); `first`: the steps 0..NB-1 are already in pf
template <int NS, int NB, class Body>
__device__ __forceinline__ void stream_unit(const f32x4* __restrict__ wq, const f32x4 (&pf)[kPF], bool first, Body body) {
  static_assert(NS % NB == 0, "whole batches");
  f32x4 cur[NB], nxt[NB];
  if (first) {
#pragma unroll
    for (int k = 0; k < NB; ++k) cur[k] = pf[k];
  } else {
    load_batch<NB>(cur, wq, 0, NS);
  }
  for (int s0 = 0; s0 < NS; s0 += NB) {
    load_batch<NB>(nxt, wq, s0 + NB, NS);
#pragma unroll
    for (int k = 0; k < NB; ++k) body(s0 + k, cur[k]);
#pragma unroll
    for (int k = 0; k < NB; ++k) cur[k] = nxt[k];
  }
}

// One dense layer (one row per patch): out[n] = act(bias[n] + sum_k x[k] W[n][k]); unit = (64 columns, quarter of K).
// The reduction + epilogue is the caller's (it differs per layer); returns after the barrier behind the slab writes.
template <int G, int NW, int N, int NG, int SPK, int NB>
__device__ __forceinline__ void dense_units(const float* __restrict__ xv /* RV + x offset, patch stride RV_F */,
                                            const f32x4* __restrict__ wp4, float* __restrict__ slab, const float* __restrict__ bias,
                                            const f32x4 (&pf)[kPF]) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  constexpr int NGP = NG * 64;
  for (int u = wave; u < NG * 4; u += NW) {
    const int kg = u & 3, ng = u >> 2, n = 64 * ng + lane;
    const f32x4* wq = wp4 + (long)((ng * 4 + kg) * SPK) * 64 + lane;
    float acc[G];
    const float bv = (kg == 0 && n < N) ? bias[n] : 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = bv;
    stream_unit<SPK, NB>(wq, pf, u == wave, [&](int s, const f32x4& w4) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(xv + g * RV_F + 4 * (kg * SPK + s));
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[g] = fmaf(x[e], w4[e], acc[g]);
      }
    });
#pragma unroll
    for (int g = 0; g < G; ++g) slab[g * 4 * NGP + kg * NGP + n] = acc[g];
  }
}
template <int NG>
__device__ __forceinline__ float dense_sum(const float* __restrict__ slab, int g, int n) {
  constexpr int NGP = NG * 64;
  const float* sp = slab + g * 4 * NGP + n;
  return (sp[0] + sp[NGP]) + (sp[2 * NGP] + sp[3 * NGP]);
}
template <int NW, int NG, int SPK>
__device__ __forceinline__ const f32x4* dense_wq(const f32x4* wp4, int u, int lane) {
  return wp4 + (long)(((u >> 2) * 4 + (u & 3)) * SPK) * 64 + lane;
}

template <int G, int THREADS>
__global__ __launch_bounds__(THREADS) void deep2d_fwd_kernel(const Deep2dArgs a) {
  constexpr int NW = THREADS / 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const RA = smem;
  float* const RS = RA + G * RA_F;
  float* const RX = RS + G * RS_F;
  float* const RV = RX + G * RX_F;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lm = lane & 15, lk = lane >> 4;
  const int b0 = blockIdx.x * G;
  const f32x4* const wp4 = reinterpret_cast<const f32x4*>(a.wp);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  int nstamp = 0;
  auto stamp = [&]() {
    if (a.stamps && blockIdx.x == 0 && t == 0) a.stamps[nstamp] = (long long)__builtin_amdgcn_s_memtime();
    ++nstamp;
  };
  auto splat = [](float v) { return (f32x4){v, v, v, v}; };
  f32x4 pf[kPF];
  stamp();

  // first units of the stages, by wavefront (for the cross-barrier prefetch)
  auto wq_conv3 = [&](int u) { return wp4 + P_CONV3 / 4 + (long)(((u >> 2) * 2 + (u & 1)) * 12) * 64 + lane; };
  auto wq_conv4 = [&](int u) { return wp4 + P_CONV4 / 4 + (long)(((u >> 1) * 2 + (u & 1)) * 24) * 64 + lane; };
  auto wq_conv5 = [&](int u) { return wp4 + P_CONV5 / 4 + (long)(((u >> 2) * 4 + (u & 3)) * 96) * 64 + lane; };
  auto wq_tconv0 = [&](int u) { return wp4 + P_TCONV0 / 4 + (long)(((u >> 1) * 2 + (u & 1)) * 96) * 64 + lane; };
  auto wq_tconv1 = [&](int u) { return wp4 + P_TCONV1 / 4 + (long)((((u >> 2) * 2 + ((u >> 1) & 1)) * 2 + (u & 1)) * 24) * 64 + lane; };
  auto wq_tconv2 = [&](int u) { return wp4 + P_TCONV2 / 4 + (long)(((u >> 3) * 4 + ((u >> 1) & 3)) * 12) * 64 + lane; };
  auto wq_tconv3 = [&](int u) { return wp4 + P_TCONV3 / 4 + (long)((u & 1) * 12) * 64 + lane; };

  // ---- the conv2 output of the G patches into the padded image; harmonic-feature tails of cat1 / cat3
  prefetch<4>(pf, wq_conv3(wave), wave < 12);
  for (int i = t; i < G * RA_F / 4; i += THREADS) reinterpret_cast<f32x4*>(RA)[i] = zero4;
  __syncthreads();
  stamp();
  for (int i = t; i < G * 1536; i += THREADS) {
    const int g = i / 1536, j = i - g * 1536;
    const int ci = j >> 6, y = (j >> 2) & 15, x4 = (j & 3) * 4;
    if (b0 + g < a.B) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(a.x2 + (long)(b0 + g) * 6144 + 4 * j);
      float* d = RA + g * RA_F + ci * X2_CP + (y + 1) * X2_RP + x4 + 1;
      d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
  }
  for (int i = t; i < G * 2 * kHd; i += THREADS) {
    const int g = i / (2 * kHd), j = i - g * 2 * kHd;
    const bool ok = b0 + g < a.B;
    if (j < kHd) RV[g * RV_F + V_CAT1 + 768 + j] = ok ? a.cat1[(long)(b0 + g) * (768 + kHd) + 768 + j] : 0.f;
    else RV[g * RV_F + V_CAT3 + kL + j - kHd] = ok ? a.cat3[(long)(b0 + g) * (kL + kHd) + kL + j - kHd] : 0.f;
  }
  __syncthreads();
  stamp();

  // ==== conv3: 24 x 16 x 16 -> 48 x 8 x 8.  M = 64 positions (4 m-tiles of two output rows), N = 48, K = 24 x 16.
  // unit = (n-tile, half of the m-tiles, half of the channels): 12 units x 96 matrix instructions
  for (int u = wave; u < 12; u += NW) {
    const int kg = u & 1, mh = (u >> 1) & 1, nt = u >> 2;
    f32x4 acc[G][2];
    const f32x4 b4v = splat(kg == 0 ? a.b3[16 * nt + lm] : 0.f);
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g][0] = acc[g][1] = b4v;
    stream_unit<12, 4>(wq_conv3(u), pf, u == wave, [&](int s, const f32x4& w4) {
      const int ci = 12 * kg + s;
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const int oy = 2 * (2 * mh + mi) + (lm >> 3), ox = lm & 7;
          const float* ap = RA + g * RA_F + ci * X2_CP + (2 * oy + lk) * X2_RP + 2 * ox;
          const float2 a01 = *reinterpret_cast<const float2*>(ap), a23 = *reinterpret_cast<const float2*>(ap + 2);
          acc[g][mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a01.x, w4[0], acc[g][mi], 0, 0, 0);
          acc[g][mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a01.y, w4[1], acc[g][mi], 0, 0, 0);
          acc[g][mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a23.x, w4[2], acc[g][mi], 0, 0, 0);
          acc[g][mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a23.y, w4[3], acc[g][mi], 0, 0, 0);
        }
    });
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)  // D: lane (n = lm, lk) holds rows m = 4 lk + r of the tile
        *reinterpret_cast<f32x4*>(RS + g * RS_F + kg * 3072 + (16 * nt + lm) * 64 + 16 * (2 * mh + mi) + 4 * lk) = acc[g][mi];
  }
  prefetch<8>(pf, wq_conv4(wave), wave < 12);
  __syncthreads();
  stamp();
  for (int i = t; i < G * RX_F; i += THREADS) {  // the padded 10 x 10 images, border included
    const int g = i / RX_F, j = i - g * RX_F, n = j / X3_CP, r = j - n * X3_CP, pr = r / X3_RP, pc = r - pr * X3_RP;
    float v = 0.f;
    if (pr >= 1 && pr <= 8 && pc >= 1 && pc <= 8) {
      const int o = n * 64 + (pr - 1) * 8 + pc - 1;
      const float* sp = RS + g * RS_F + o;
      v = elu(sp[0] + sp[3072]);
      if (b0 + g < a.B) a.a3[(long)(b0 + g) * 3072 + o] = v;
    }
    RX[i] = v;
  }
  __syncthreads();
  stamp();

  // ==== conv4: 48 x 8 x 8 -> 96 x 4 x 4.  M = 16, N = 96 (6 n-tiles), K = 48 x 16; unit = (n-tile, half of the channels)
  for (int u = wave; u < 12; u += NW) {
    const int kg = u & 1, nt = u >> 1;
    f32x4 acc[G];
    const f32x4 b4v = splat(kg == 0 ? a.b4[16 * nt + lm] : 0.f);
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = b4v;
    stream_unit<24, 8>(wq_conv4(u), pf, u == wave, [&](int s, const f32x4& w4) {
      const int ci = 24 * kg + s;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int oy = lm >> 2, ox = lm & 3;
        const float* ap = RX + g * RX_F + ci * X3_CP + (2 * oy + lk) * X3_RP + 2 * ox;
        const float2 a01 = *reinterpret_cast<const float2*>(ap), a23 = *reinterpret_cast<const float2*>(ap + 2);
        acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a01.x, w4[0], acc[g], 0, 0, 0);
        acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a01.y, w4[1], acc[g], 0, 0, 0);
        acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a23.x, w4[2], acc[g], 0, 0, 0);
        acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a23.y, w4[3], acc[g], 0, 0, 0);
      }
    });
#pragma unroll
    for (int g = 0; g < G; ++g) *reinterpret_cast<f32x4*>(RA + g * RA_F + kg * 1536 + (16 * nt + lm) * 16 + 4 * lk) = acc[g];
  }
  prefetch<8>(pf, wq_conv5(wave), wave < 12);
  __syncthreads();
  stamp();
  // conv4's output as rows of two 4-wide windows: record (channel, padded row) = [cols -1..2 | cols 1..4], so that the
  // 4 x 4 window of conv5's output position (oy, ox) is four aligned ds_read_b128 at column 4 ox
  for (int i = t; i < G * 96 * X4_CP; i += THREADS) {
    const int g = i / (96 * X4_CP), j = i - g * (96 * X4_CP), n = j / X4_CP, r = j - n * X4_CP, pr = r >> 3, k = r & 7;
    const int pc = k < 4 ? k : k - 2;  // padded column of this slot
    float v = 0.f;
    if (pr >= 1 && pr <= 4 && pc >= 1 && pc <= 4) {
      const int o = n * 16 + (pr - 1) * 4 + pc - 1;
      const float* sp = RA + g * RA_F + o;
      v = elu(sp[0] + sp[1536]);
      if ((k <= 3 || k == 6) && b0 + g < a.B) a.a4[(long)(b0 + g) * 1536 + o] = v;  // (columns 2, 3 have two slots)
    }
    RS[g * RS_F + j] = v;
  }
  __syncthreads();
  stamp();

  // ==== conv5: 96 x 4 x 4 -> 192 x 2 x 2.  M = 4 positions: v_mfma_f32_4x4x1 (16 blocks = 64 output channels, one lane
  // each; block row r = position r).  unit = (64 channels, quarter of the input channels): 12 units x 384 instructions
  for (int u = wave; u < 12; u += NW) {
    const int kg = u & 3, ng = u >> 2;
    f32x4 acc[G];
    const f32x4 b4v = splat(kg == 0 ? a.b5[64 * ng + lane] : 0.f);
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = b4v;
    const int oy = (lane >> 1) & 1, ox = lane & 1;
    stream_unit<96, 8>(wq_conv5(u), pf, u == wave, [&](int s, const f32x4& w4) {
      const int ci = 24 * kg + (s >> 2), ky = s & 3;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(RS + g * RS_F + ci * X4_CP + (2 * oy + ky) * 8 + 4 * ox);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[e], w4[e], acc[g], 0, 0, 0);
      }
    });
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) RA[g * RA_F + kg * 768 + r * 192 + 64 * ng + lane] = acc[g][r];
  }
  prefetch<7>(pf, dense_wq<NW, 4, 49>(wp4 + P_FC1 / 4, wave, lane), wave < 16);
  __syncthreads();
  stamp();
  for (int i = t; i < G * 768; i += THREADS) {
    const int g = i / 768, j = i - g * 768, n = j >> 2, m = j & 3;
    const float* sp = RA + g * RA_F + m * 192 + n;
    const float v = elu((sp[0] + sp[768]) + (sp[1536] + sp[2304]));
    RV[g * RV_F + V_CAT1 + j] = v;
    if (b0 + g < a.B) a.cat1[(long)(b0 + g) * (768 + kHd) + j] = v;
  }
  __syncthreads();
  stamp();

  // ==== fc1 -> fc2in (the latent code) -> fc2out -> fc3
  dense_units<G, NW, kL, 4, 49, 7>(RV + V_CAT1, wp4 + P_FC1 / 4, RA, a.bfc1, pf);
  prefetch<7>(pf, dense_wq<NW, 4, 14>(wp4 + P_FC2IN / 4, wave, lane), wave < 16);
  __syncthreads();
  for (int i = t; i < G * kL; i += THREADS) {
    const int g = i / kL, n = i - g * kL;
    const float v = elu(dense_sum<4>(RA, g, n));
    RV[g * RV_F + V_Z1 + n] = v;
    if (b0 + g < a.B) a.z1[(long)(b0 + g) * kL + n] = v;
  }
  __syncthreads();
  stamp();
  dense_units<G, NW, kL, 4, 14, 7>(RV + V_Z1, wp4 + P_FC2IN / 4, RA, a.bfc2in, pf);
  prefetch<7>(pf, dense_wq<NW, 4, 14>(wp4 + P_FC2OUT / 4, wave, lane), wave < 16);
  __syncthreads();
  for (int i = t; i < G * kL; i += THREADS) {
    const int g = i / kL, n = i - g * kL;
    const float v = elu(dense_sum<4>(RA, g, n));
    RV[g * RV_F + V_MU + n] = v;
    if (b0 + g < a.B) a.mu[(long)(b0 + g) * a.mu_ld + n] = v;
  }
  __syncthreads();
  stamp();
  dense_units<G, NW, kL, 4, 14, 7>(RV + V_MU, wp4 + P_FC2OUT / 4, RA, a.bfc2out, pf);
  prefetch<5>(pf, dense_wq<NW, 12, 15>(wp4 + P_FC3 / 4, wave, lane), wave < 48);
  __syncthreads();
  for (int i = t; i < G * kL; i += THREADS) {
    const int g = i / kL, n = i - g * kL;
    const float v = elu(dense_sum<4>(RA, g, n));
    RV[g * RV_F + V_CAT3 + n] = v;
    if (b0 + g < a.B) a.cat3[(long)(b0 + g) * (kL + kHd) + n] = v;
  }
  __syncthreads();
  stamp();
  dense_units<G, NW, 768, 12, 15, 5>(RV + V_CAT3, wp4 + P_FC3 / 4, RA, a.bfc3, pf);
  prefetch<8>(pf, wq_tconv0(wave), wave < 12);
  __syncthreads();
  // fc3's output (192 x 2 x 2, no activation: src/lofar_models.py:91) as padded rows of column PAIRS: record (channel, padded
  // row) = [cols (0,1) | (1,2) | (2,3) | unused]: the two column taps of a transposed-conv lane are one ds_read_b64
  for (int i = t; i < G * 192 * D0_CP; i += THREADS) {
    const int g = i / (192 * D0_CP), j = i - g * (192 * D0_CP), ci = j >> 5, r = j & 31, pr = r >> 3, k = r & 7;
    const int pc = (k + 1) >> 1;  // padded column of this slot (k < 6)
    float v = 0.f;
    if (k < 6 && (pr == 1 || pr == 2) && (pc == 1 || pc == 2)) {
      const int n = ci * 4 + (pr - 1) * 2 + pc - 1;
      v = dense_sum<12>(RA, g, n);
      if ((k & 1) && b0 + g < a.B) a.d0[(long)(b0 + g) * 768 + n] = v;  // (slots 1 and 3 are the first of each column)
    }
    RS[g * RS_F + j] = v;
  }
  __syncthreads();
  stamp();

  // ==== tconv0: 192 x 2 x 2 -> 96 x 4 x 4 on v_mfma_f32_4x4x1: block = (output parity, channel quad), block row r = input
  // position r, k = (input channel, tap).  unit = (16 channels, half of the input channels): 12 units x 384 instructions
  for (int u = wave; u < 12; u += NW) {
    const int kg = u & 1, c = u >> 1;
    const int p = lane >> 4, py = p >> 1, px = p & 1, iy = (lane >> 1) & 1, ix = lane & 1;
    const int co = 16 * c + (lane & 15);
    f32x4 acc[G];
    const f32x4 b4v = splat(kg == 0 ? a.bt0[co] : 0.f);
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = b4v;
    const int aoff = (iy + py) * 8 + 2 * (ix + px);  // rows iy + py - 1 + j, column pair (ix + px - 1, ix + px), padded by one
    stream_unit<96, 8>(wq_tconv0(u), pf, u == wave, [&](int s, const f32x4& w4) {
      const int ci = 96 * kg + s;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float* ap = RS + g * RS_F + ci * D0_CP + aoff;
        const float2 r0 = *reinterpret_cast<const float2*>(ap), r1 = *reinterpret_cast<const float2*>(ap + 8);
        acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(r0.x, w4[0], acc[g], 0, 0, 0);
        acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(r0.y, w4[1], acc[g], 0, 0, 0);
        acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(r1.x, w4[2], acc[g], 0, 0, 0);
        acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(r1.y, w4[3], acc[g], 0, 0, 0);
      }
    });
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r)  // row r = input position (r >> 1, r & 1) -> output (2 iy + py, 2 ix + px)
        RA[g * RA_F + kg * 1536 + co * 16 + (2 * (r >> 1) + py) * 4 + 2 * (r & 1) + px] = acc[g][r];
  }
  prefetch<8>(pf, wq_tconv1(wave), wave < 12);
  __syncthreads();
  stamp();
  for (int i = t; i < G * 96 * T0_CP; i += THREADS) {  // padded 6 x 8 (+4) images, border included
    const int g = i / (96 * T0_CP), j = i - g * (96 * T0_CP), co = j / T0_CP, r = j - co * T0_CP, pr = r >> 3, pc = r & 7;
    float v = 0.f;
    if (r < 48 && pr >= 1 && pr <= 4 && pc >= 1 && pc <= 4) {
      const int o = co * 16 + (pr - 1) * 4 + pc - 1;
      const float* sp = RA + g * RA_F + o;
      v = elu(sp[0] + sp[1536]);
      if (b0 + g < a.B) a.t0[(long)(b0 + g) * 1536 + o] = v;
    }
    RS[g * RS_F + j] = v;
  }
  __syncthreads();
  stamp();

  // ==== tconv1: 96 x 4 x 4 -> 48 x 8 x 8, per output parity M = 16 input positions, N = 48, K = 96 x 4 taps (k-step = four
  // input channels at one tap).  unit = (16 channels, row parity, half of the input channels): 12 units x 96 instructions
  for (int u = wave; u < 12; u += NW) {
    const int kg = u & 1, py = (u >> 1) & 1, c = u >> 2;
    const int iy = lm >> 2, ix = lm & 3, co = 16 * c + lm;
    f32x4 acc[G][2];
    const f32x4 b4v = splat(kg == 0 ? a.bt1[co] : 0.f);
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g][0] = acc[g][1] = b4v;
    stream_unit<24, 8>(wq_tconv1(u), pf, u == wave, [&](int s, const f32x4& w4) {
      const int ci = 4 * (12 * kg + (s >> 1)) + lk, jy = s & 1;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float* ap = RS + g * RS_F + ci * T0_CP + (iy + py + jy) * T0_RP + ix;  // padded columns ix, ix + 1, ix + 2
        const float am = ap[0], a0 = ap[1], a1 = ap[2];
        // column parity 0: columns ix - 1, ix through kernel columns 3, 1; parity 1: columns ix, ix + 1 through 2, 0
        acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(am, w4[3], acc[g][0], 0, 0, 0);
        acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, w4[1], acc[g][0], 0, 0, 0);
        acc[g][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, w4[2], acc[g][1], 0, 0, 0);
        acc[g][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, w4[0], acc[g][1], 0, 0, 0);
      }
    });
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) {  // row m = 4 lk + r = input position (lk, r) -> outputs (2 lk + py, 2 r), (2 lk + py, 2 r + 1)
        const float2 o = {acc[g][0][r], acc[g][1][r]};
        *reinterpret_cast<float2*>(RA + g * RA_F + kg * 3072 + co * 64 + (2 * lk + py) * 8 + 2 * r) = o;
      }
  }
  prefetch<4>(pf, wq_tconv2(wave), wave < 16);
  __syncthreads();
  stamp();
  for (int i = t; i < G * RX_F; i += THREADS) {
    const int g = i / RX_F, j = i - g * RX_F, co = j / T1_CP, r = j - co * T1_CP, pr = r / T1_RP, pc = r - pr * T1_RP;
    float v = 0.f;
    if (pr >= 1 && pr <= 8 && pc >= 1 && pc <= 8) {
      const int o = co * 64 + (pr - 1) * 8 + pc - 1;
      const float* sp = RA + g * RA_F + o;
      v = elu(sp[0] + sp[3072]);
      if (b0 + g < a.B) a.t1[(long)(b0 + g) * 3072 + o] = v;
    }
    RX[i] = v;
  }
  __syncthreads();
  stamp();

  // ==== tconv2: 48 x 8 x 8 -> 24 x 16 x 16, per output parity M = 64 (4 m-tiles of two input rows), N = 24 (two n-tiles, the
  // second half empty), K = 48 x 4 taps.  unit = (n-tile, parity, half of the m-tiles), whole K: 16 units x 96 instructions.
  // The output goes to RA in the padded form tconv3 reads (the slab of tconv1 is dead); its border is cleared here.
  for (int i = t; i < G * 24 * 68; i += THREADS) {
    const int g = i / (24 * 68), j = i - g * (24 * 68), ch = j / 68, k = j - ch * 68;
    const int o = k < 18 ? k : k < 36 ? 17 * X2_RP + k - 18 : (1 + ((k - 36) >> 1)) * X2_RP + ((k & 1) ? 17 : 0);
    RA[g * RA_F + ch * X2_CP + o] = 0.f;
  }
  for (int u = wave; u < 16; u += NW) {
    const int mh = u & 1, p = (u >> 1) & 3, c = u >> 3, py = p >> 1, px = p & 1;
    const int co = 16 * c + lm;
    f32x4 acc[G][2];
    const f32x4 b4v = splat(co < 24 ? a.bt2[co] : 0.f);
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g][0] = acc[g][1] = b4v;
    stream_unit<12, 4>(wq_tconv2(u), pf, u == wave, [&](int s, const f32x4& w4) {
      const int ci = 4 * s + lk;
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const int iy = 2 * (2 * mh + mi) + (lm >> 3), ix = lm & 7;
          const float* ap = RX + g * RX_F + ci * T1_CP + (iy + py) * T1_RP + ix + px;  // rows iy + py - 1 + j, columns ix + px - 1 + j
          acc[g][mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[0], w4[0], acc[g][mi], 0, 0, 0);
          acc[g][mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[1], w4[1], acc[g][mi], 0, 0, 0);
          acc[g][mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[T1_RP], w4[2], acc[g][mi], 0, 0, 0);
          acc[g][mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[T1_RP + 1], w4[3], acc[g][mi], 0, 0, 0);
        }
    });
    if (co < 24) {
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = 16 * (2 * mh + mi) + 4 * lk + r, iy = m >> 3, ix = m & 7;
            RA[g * RA_F + co * X2_CP + (2 * iy + py + 1) * X2_RP + 2 * ix + px + 1] = elu(acc[g][mi][r]);
          }
    }
  }
  prefetch<4>(pf, wq_tconv3(wave), wave < 16);
  __syncthreads();
  stamp();
  for (int i = t; i < G * 6144; i += THREADS) {
    const int g = i / 6144, j = i - g * 6144, ch = j >> 8, y = (j >> 4) & 15, x = j & 15;
    if (b0 + g < a.B) a.t2[(long)(b0 + g) * 6144 + j] = RA[g * RA_F + ch * X2_CP + (y + 1) * X2_RP + x + 1];
  }

  // ==== tconv3: 24 x 16 x 16 -> 12 x 32 x 32, per output parity M = 256 (16 m-tiles = input rows), N = 12 (one n-tile), K = 24 x 4
  // taps.  unit = (row parity, two input rows), both column parities: 16 units x 96 instructions; a lane ends with eight
  // consecutive outputs of a row (two float4 stores, straight to HBM)
  for (int u = wave; u < 16; u += NW) {
    const int py = u & 1, mq = u >> 1;
    f32x4 acc[G][2][2];
    const f32x4 b4v = splat(lm < 12 ? a.bt3[lm] : 0.f);
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g][0][0] = acc[g][0][1] = acc[g][1][0] = acc[g][1][1] = b4v;
    stream_unit<12, 4>(wq_tconv3(u), pf, u == wave, [&](int s, const f32x4& w4) {
      const int ci = 4 * (s >> 1) + lk, jy = s & 1;
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const int iy = 2 * mq + mi;
          const float* ap = RA + g * RA_F + ci * X2_CP + (iy + py + jy) * X2_RP + lm;  // padded columns ix, ix + 1, ix + 2 (ix = lm)
          const float am = ap[0], a0 = ap[1], a1 = ap[2];
          acc[g][mi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(am, w4[3], acc[g][mi][0], 0, 0, 0);
          acc[g][mi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, w4[1], acc[g][mi][0], 0, 0, 0);
          acc[g][mi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, w4[2], acc[g][mi][1], 0, 0, 0);
          acc[g][mi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, w4[0], acc[g][mi][1], 0, 0, 0);
        }
    });
    if (lm < 12) {
#pragma unroll
      for (int g = 0; g < G; ++g)
        if (b0 + g < a.B) {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) {  // rows m = 4 lk + r = input columns -> output columns 8 lk + 2 r + px of row 2 iy + py
            float* op = a.t3 + (long)(b0 + g) * 12288 + lm * 1024 + (2 * (2 * mq + mi) + py) * 32 + 8 * lk;
            const f32x4 lo = {elu(acc[g][mi][0][0]), elu(acc[g][mi][1][0]), elu(acc[g][mi][0][1]), elu(acc[g][mi][1][1])};
            const f32x4 hi = {elu(acc[g][mi][0][2]), elu(acc[g][mi][1][2]), elu(acc[g][mi][0][3]), elu(acc[g][mi][1][3])};
            *reinterpret_cast<f32x4*>(op) = lo;
            *reinterpret_cast<f32x4*>(op + 4) = hi;
          }
        }
    }
  }
  stamp();
}

}  // namespace

size_t deep2d_packed_floats() { return (size_t)P_TOTAL; }

bool deep2d_supported(int L, int hd, int rica, const int* enc_ch /* the output channels of conv1 .. conv5 */, int H2 /* conv2's output height */) {
  return L == kL && hd == kHd && rica && enc_ch[0] == 12 && enc_ch[1] == 24 && enc_ch[2] == 48 && enc_ch[3] == 96 && enc_ch[4] == 192 && H2 == 16;
}

int deep2d_pack(const Deep2dWeights& w, float* packed, hipStream_t st) {
  if (!packed || !w.c3 || !w.c4 || !w.c5 || !w.fc1 || !w.fc2in || !w.fc2out || !w.fc3 || !w.t0 || !w.t1 || !w.t2 || !w.t3) {
    set_last_error("deep2d_pack: null pointer");
    return LSHM_ERR_ARG;
  }
  if (reinterpret_cast<uintptr_t>(packed) & 15) { set_last_error("deep2d_pack: unaligned destination"); return LSHM_ERR_ARG; }
  const PackSrc s{w.c3, w.c4, w.c5, w.fc1, w.fc2in, w.fc2out, w.fc3, w.t0, w.t1, w.t2, w.t3};
  hipLaunchKernelGGL(deep2d_pack_kernel, dim3(cdiv(P_TOTAL / 4, 256)), dim3(256), 0, st, s, packed);
  return check_launch("deep2d_pack");
}

template <int G, int THREADS>
static int launch_fwd(const Deep2dArgs& a, hipStream_t st) {
  const size_t lds = sizeof(float) * (size_t)G * (RA_F + RS_F + RX_F + RV_F);
  auto kern = deep2d_fwd_kernel<G, THREADS>;
  int rc;
  if ((rc = kernel_budget_ok(reinterpret_cast<const void*>(kern), THREADS, lds, "deep2d forward"))) return rc;
  static thread_local const void* raised = nullptr;  // the dynamic-LDS limit of a kernel is raised once per thread and kernel
  if (raised != reinterpret_cast<const void*>(kern)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      (void)hipGetLastError();
      set_last_error("deep2d_fwd: cannot raise the dynamic LDS limit");
      return LSHM_ERR_UNSUPPORTED;
    }
    raised = reinterpret_cast<const void*>(kern);
  }
  hipLaunchKernelGGL(kern, dim3(cdiv(a.B, G)), dim3(THREADS), lds, st, a);
  return check_launch("deep2d_fwd");
}

int deep2d_fwd(const Deep2dIO& io, const float* packed, int B, int variant, hipStream_t st) {
  auto al16 = [](const void* q) { return q && (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const bool ok = al16(packed) && al16(io.x2) && al16(io.t3) && io.t2 && io.a3 && io.a4 && io.cat1 && io.z1 && io.mu && io.cat3 && io.d0 &&
                  io.t0 && io.t1 && io.b3 && io.b4 && io.b5 && io.bfc1 && io.bfc2in && io.bfc2out && io.bfc3 && io.bt0 && io.bt1 &&
                  io.bt2 && io.bt3 && B > 0;
  if (!ok) { set_last_error("deep2d_fwd: null / unaligned pointer"); return LSHM_ERR_ARG; }
  Deep2dArgs a;
  a.wp = packed; a.x2 = io.x2;
  a.b3 = io.b3; a.b4 = io.b4; a.b5 = io.b5; a.bfc1 = io.bfc1; a.bfc2in = io.bfc2in; a.bfc2out = io.bfc2out; a.bfc3 = io.bfc3;
  a.bt0 = io.bt0; a.bt1 = io.bt1; a.bt2 = io.bt2; a.bt3 = io.bt3;
  a.a3 = io.a3; a.a4 = io.a4; a.cat1 = io.cat1; a.z1 = io.z1; a.mu = io.mu; a.mu_ld = io.mu_ld; a.cat3 = io.cat3; a.d0 = io.d0;
  a.t0 = io.t0; a.t1 = io.t1; a.t2 = io.t2; a.t3 = io.t3;
  a.B = B;
  a.stamps = io.stamps;
  switch (variant) {
    case 0: return launch_fwd<1, 1024>(a, st);
    case 1: return launch_fwd<2, 1024>(a, st);
    case 2: return launch_fwd<1, 512>(a, st);
    default: set_last_error("deep2d_fwd: unknown variant"); return LSHM_ERR_ARG;
  }
}

}  // namespace lshm
