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
//
// BACKWARD (BWD = true).  The data-gradient pass through the same layers is the same eleven-stage pipeline with other
// weights: the data gradient of a k4 s2 p1 transposed conv IS the conv with the same weight tensor read as
// (out, in, ky, kx) and vice versa, so tconv2' tconv1' tconv0' run as the conv3 conv4 conv5 stages and conv5' conv4' conv3'
// conv2' as the tconv0..3 stages, on the layers' own tensors; only the four dense stages read their weights transposed
// (fc3' 768 -> 240, fc2out' and fc2in' 224 -> 224, fc1' 224 -> 784).  Epilogues: no bias; instead of ELU every stage
// multiplies by ELU'(saved activation of the tensor it differentiates) and the latent-term gradient joins in front of
// fc2out's ELU'.  Every stage's output (the dz the weight gradients read) goes to HBM once, as in the forward.
#include <stdlib.h>

#include <type_traits>

#include "kernels.h"
#include "deep2d.h"

namespace lshm {
namespace {

constexpr int kL = 224;   // latent width of the 2-D autoencoder (src/kharmonic_lofar.py:37: 256 - 2 x 16)
constexpr int kHd = 16;   // harmonic features (4 scales)

// ---- packed weights: stage bases (floats); one step of one region = 64 lanes x float4 = 256 floats
constexpr long kStep = 256;
template <bool BWD>
struct Lay {
  // dense stages: (outputs N, 64-column groups NG, float4 steps per K quarter SPK, steps in flight NB)
  static constexpr int N1 = BWD ? kL + kHd : kL, SPK1 = BWD ? 48 : 49, NB1 = BWD ? 6 : 7;             // fc1 784 -> 224 | fc3' 768 -> 240
  static constexpr int N4 = BWD ? 768 + kHd : 768, NG4 = BWD ? 13 : 12, SPK4 = BWD ? 14 : 15, NB4 = BWD ? 7 : 5;  // fc3 240 -> 768 | fc1' 224 -> 784
  static constexpr long P_CONV3 = 0;                         // 6 regions (n-tile, k-half) x 12 steps (channel)
  static constexpr long P_CONV4 = P_CONV3 + 6 * 12 * kStep;  // 12 regions (n-tile, k-half) x 24 steps
  static constexpr long P_CONV5 = P_CONV4 + 12 * 24 * kStep; // 12 regions (64 channels, k-quarter) x 96 steps (channel, kernel row)
  static constexpr long P_FC1 = P_CONV5 + 12 * 96 * kStep;   // 16 regions (64 columns, k-quarter) x SPK1 steps
  static constexpr long P_FC2IN = P_FC1 + 16 * SPK1 * kStep; // 16 x 14
  static constexpr long P_FC2OUT = P_FC2IN + 16 * 14 * kStep;
  static constexpr long P_FC3 = P_FC2OUT + 16 * 14 * kStep;  // 4 NG4 regions (64 columns, k-quarter) x SPK4 steps
  static constexpr long P_TCONV0 = P_FC3 + 4 * NG4 * SPK4 * kStep;  // 12 regions (16 channels, k-half) x 96 steps (input channel)
  static constexpr long P_TCONV1 = P_TCONV0 + 12 * 96 * kStep;  // 12 regions (16 channels, row parity, k-half) x 24 steps (4 input channels, row tap)
  static constexpr long P_TCONV2 = P_TCONV1 + 12 * 24 * kStep;  // 8 regions (16 channels, parity) x 12 steps (4 input channels)
  static constexpr long P_TCONV3 = P_TCONV2 + 8 * 12 * kStep;   // 2 regions (row parity) x 12 steps (4 input channels, row tap)
  static constexpr long P_TOTAL = P_TCONV3 + 2 * 12 * kStep;
};
constexpr long kPackedMax = Lay<true>::P_TOTAL > Lay<false>::P_TOTAL ? Lay<true>::P_TOTAL : Lay<false>::P_TOTAL;

// transposed conv k4 s2 p1, taps in positional order: output row 2a + p takes input rows a + p - 1 + j (j = 0, 1) through
// kernel rows {3, 1} (p = 0) / {2, 0} (p = 1)  [oy = 2 iy - 1 + ky]
__device__ __forceinline__ int tap_k(int p, int j) { return p ? (j ? 0 : 2) : (j ? 1 : 3); }

struct PackSrc {
  const float *c3, *c4, *c5, *fc1, *fc2in, *fc2out, *fc3, *t0, *t1, *t2, *t3;
};

// source element of packed float `off`; nullptr: padding (zero)
template <bool BWD>
__device__ __forceinline__ const float* pack_source(const PackSrc& w, long off) {
  using Y = Lay<BWD>;
  const int e = (int)(off & 3), lane = (int)((off >> 2) & 63);
  const int lm = lane & 15, lk = lane >> 4;
  if (off < Y::P_CONV4) {            // conv3: W (48, 24, 4, 4)
    const int q = (int)((off - Y::P_CONV3) / kStep), r = q / 12, s = q % 12;
    const int nt = r >> 1, kg = r & 1, ci = 12 * kg + s, n = 16 * nt + lm;
    return w.c3 + ((n * 24 + ci) * 16 + lk * 4 + e);
  }
  if (off < Y::P_CONV5) {            // conv4: W (96, 48, 4, 4)
    const int q = (int)((off - Y::P_CONV4) / kStep), r = q / 24, s = q % 24;
    const int nt = r >> 1, kg = r & 1, ci = 24 * kg + s, n = 16 * nt + lm;
    return w.c4 + ((n * 48 + ci) * 16 + lk * 4 + e);
  }
  if (off < Y::P_FC1) {              // conv5: W (192, 96, 4, 4)
    const int q = (int)((off - Y::P_CONV5) / kStep), r = q / 96, s = q % 96;
    const int ng = r >> 2, kg = r & 3, ci = 24 * kg + (s >> 2), ky = s & 3, n = 64 * ng + lane;
    return w.c5 + ((n * 96 + ci) * 16 + ky * 4 + e);
  }
  if (off < Y::P_TCONV0) {        // the dense layers: W (N, K) row-major; BWD: the transposed product, W (K, N) row-major
    const float* base;
    long o;
    int N, K, SPK;
    if (off < Y::P_FC2IN) { base = w.fc1; o = off - Y::P_FC1; N = Y::N1; K = BWD ? 768 : 768 + kHd; SPK = Y::SPK1; }
    else if (off < Y::P_FC2OUT) { base = w.fc2in; o = off - Y::P_FC2IN; N = kL; K = kL; SPK = 14; }
    else if (off < Y::P_FC3) { base = w.fc2out; o = off - Y::P_FC2OUT; N = kL; K = kL; SPK = 14; }
    else { base = w.fc3; o = off - Y::P_FC3; N = Y::N4; K = BWD ? kL : kL + kHd; SPK = Y::SPK4; }
    const int q = (int)(o / kStep), r = q / SPK, s = q % SPK;
    const int ng = r >> 2, kg = r & 3, n = 64 * ng + lane, k = 4 * (kg * SPK + s) + e;
    if (n >= N) return nullptr;
    return BWD ? base + ((long)k * N + n) : base + ((long)n * K + k);
  }
  if (off < Y::P_TCONV1) {           // tconv0: W (192, 96, 4, 4); lane = (parity, 16 channels), e = (row tap, column tap)
    const int q = (int)((off - Y::P_TCONV0) / kStep), r = q / 96, s = q % 96;
    const int c = r >> 1, kg = r & 1, ci = 96 * kg + s;
    const int p = lane >> 4, co = 16 * c + (lane & 15);
    const int ky = tap_k(p >> 1, e >> 1), kx = tap_k(p & 1, e & 1);
    return w.t0 + ((ci * 96 + co) * 16 + ky * 4 + kx);
  }
  if (off < Y::P_TCONV2) {           // tconv1: W (96, 48, 4, 4); region = (channel tile, row parity, k-half), step = (4 input channels, row tap)
    const int q = (int)((off - Y::P_TCONV1) / kStep), r = q / 24, s = q % 24;
    const int kg = r & 1, py = (r >> 1) & 1, c = r >> 2;
    const int ci = 4 * (12 * kg + (s >> 1)) + lk, jy = s & 1, co = 16 * c + lm;
    return w.t1 + ((ci * 48 + co) * 16 + tap_k(py, jy) * 4 + e);
  }
  if (off < Y::P_TCONV3) {           // tconv2: W (48, 24, 4, 4); region = (channel tile, parity), step = 4 input channels
    const int q = (int)((off - Y::P_TCONV2) / kStep), r = q / 12, s = q % 12;
    const int c = r >> 2, p = r & 3, ci = 4 * s + lk, co = 16 * c + lm;
    const int ky = tap_k(p >> 1, e >> 1), kx = tap_k(p & 1, e & 1);
    return co < 24 ? w.t2 + ((ci * 24 + co) * 16 + ky * 4 + kx) : nullptr;
  }
  {                               // tconv3: W (24, 12, 4, 4); region = row parity, step = (4 input channels, row tap)
    const int q = (int)((off - Y::P_TCONV3) / kStep), py = q / 12, s = q % 12;
    const int ci = 4 * (s >> 1) + lk, jy = s & 1, co = lm;
    return co < 12 ? w.t3 + ((ci * 12 + co) * 16 + tap_k(py, jy) * 4 + e) : nullptr;
  }
}

template <bool BWD, bool BFW>
__global__ __launch_bounds__(256) void deep2d_pack_kernel(const PackSrc w, float* __restrict__ packed) {
  const long i4 = (long)blockIdx.x * 256 + threadIdx.x;  // one float4 of the packed image per thread
  if (i4 * 4 >= Lay<BWD>::P_TOTAL) return;
  f32x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float* s = pack_source<BWD>(w, i4 * 4 + e);
    v[e] = s ? *s : 0.f;
  }
  if (BFW) reinterpret_cast<bf16x4*>(packed)[i4] = __builtin_convertvector(v, bf16x4);  // round to nearest even (v_cvt_pk_bf16_f32)
  else *reinterpret_cast<f32x4*>(packed + i4 * 4) = v;
}

struct Deep2dArgs {
  const float* wp;                 // packed weights
  const float* x2;                 // stage-0 input (B, 24, 16, 16): conv2's output | gradient w.r.t. tconv2's output
  // forward: biases.  backward: null
  const float *b3, *b4, *b5, *bfc1, *bfc2in, *bfc2out, *bfc3, *bt0, *bt1, *bt2, *bt3;
  // backward: the saved activation each stage's output is multiplied by ELU' of (null: no multiply); forward: null
  const float *s3, *s4, *sd1, *sd2, *sd3, *sd4, *st0, *st1, *st2, *st3;
  long sd2_ld;                     // row pitch of sd2 (the latent code inside Mu)
  const float* gmu; long gmu_ld;   // backward: latent-term gradient added in front of the second dense stage's ELU'
  float *a3, *a4;                  // outputs of stages 0, 1: (B, 48, 8, 8), (B, 96, 4, 4)
  float* cat1; long cat1_ld;       // stage 2: (B, 768) with row pitch cat1_ld; forward: columns 768..783 = elu(fcuv1) are READ
  float* z1; long z1_ld;           // dense 1: forward (B, 224), backward (B, 240)
  float* mu; long mu_ld;           // dense 2: (B, 224)
  float* cat3; long cat3_ld;       // dense 3: (B, 224); forward: columns 224..239 = elu(fcuv3) are READ
  float* d0; long d0_ld;           // dense 4: forward (B, 768), backward (B, 784)
  float *t0, *t1, *t2, *t3;        // stages 7..10: (B, 96, 4, 4), (B, 48, 8, 8), (B, 24, 16, 16), (B, 12, 32, 32)
  int B;
  long long* stamps;               // diagnostics (or null): shader-clock readings of workgroup 0 at the stage boundaries
};

// LDS regions (floats per patch)
constexpr int RA_F = 7776, RS_F = 6144, RX_F = 4800, RV_F = 1536;
constexpr int V_0 = 0, V_1 = 784, V_2 = 1024, V_3 = 1248;  // vectors of the dense stages: 784 | 240 | 224 | 240
constexpr int X2_CP = 324, X2_RP = 18;   // conv2 / tconv2 output image: channel / row pitch (16 + 2 border)
constexpr int X3_CP = 100, X3_RP = 10;   // 8 + 2
constexpr int X4_CP = 48;                // 6 padded rows x [cols -1..2 | cols 1..4]
constexpr int D0_CP = 32;                // 4 padded rows x [cols (0,1) | (1,2) | (2,3) | -]
constexpr int T0_CP = 52, T0_RP = 8;     // 4 + 2 rows of 8 (+4): k-lanes land on disjoint bank groups
constexpr int T1_CP = 100, T1_RP = 10;
constexpr int kPF = 8;                   // float4 registers of the cross-barrier weight prefetch

// A weight slot is the four values a lane consumes in one step: a float4, or (bf16 operand precision, BASELINE configs[2]) four
// bf16 in 8 bytes -- half the L2 stream; widened to fp32 registers on arrival (a shift), fp32 products and accumulation.
// (round 4, second half: the bf16 slot stays bf16 in registers and goes into v_mfma_f32_16x16x16_bf16 / 4x4x4_bf16 as it is; the four
// activations it meets are rounded to bf16 -- the operand precision of BASELINE configs[2], as in the implicit GEMMs' bf16 form --
// one matrix instruction instead of four.  fp32 slots: the instruction sequence of before, bit for bit.)
typedef short s16x4 __attribute__((ext_vector_type(4)));  // operand registers of the bf16 matrix instructions
__device__ __forceinline__ f32x4 wload(const f32x4* __restrict__ p) { return *p; }
__device__ __forceinline__ bf16x4 wload(const bf16x4* __restrict__ p) { return *p; }
__device__ __forceinline__ f32x4 wide(const f32x4& w) { return w; }
__device__ __forceinline__ f32x4 wide(const bf16x4& w) { return __builtin_convertvector(w, f32x4); }
__device__ __forceinline__ s16x4 op_bits(const bf16x4& v) { return __builtin_bit_cast(s16x4, v); }
__device__ __forceinline__ s16x4 op_pack(float a0, float a1, float a2, float a3) {
  return op_bits(__builtin_convertvector((f32x4){a0, a1, a2, a3}, bf16x4));
}
// four k-steps of one 16 x 16 accumulator tile: activations a0..a3 against the slot's four values
__device__ __forceinline__ f32x4 mma16(float a0, float a1, float a2, float a3, const f32x4& w, f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, w[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, w[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, w[2], acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a3, w[3], acc, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma16(float a0, float a1, float a2, float a3, const bf16x4& w, f32x4 acc) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(op_pack(a0, a1, a2, a3), op_bits(w), acc, 0, 0, 0);
}
// two k-steps: activations a0, a1 against values IA, IB of the slot (the transposed convolutions' column parities)
template <int IA, int IB>
__device__ __forceinline__ f32x4 mma16_2(float a0, float a1, const f32x4& w, f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, w[IA], acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a1, w[IB], acc, 0, 0, 0);
}
template <int IA, int IB>
__device__ __forceinline__ f32x4 mma16_2(float a0, float a1, const bf16x4& w, f32x4 acc) {
  const bf16x4 wv = {w[IA], w[IB], (bf16)0.f, (bf16)0.f};
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(op_pack(a0, a1, 0.f, 0.f), op_bits(wv), acc, 0, 0, 0);
}
// four k-steps of the 16 independent 4 x 4 blocks
__device__ __forceinline__ f32x4 mma4(float a0, float a1, float a2, float a3, const f32x4& w, f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, w[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a1, w[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a2, w[2], acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_4x4x1f32(a3, w[3], acc, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma4(float a0, float a1, float a2, float a3, const bf16x4& w, f32x4 acc) {
  return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(op_pack(a0, a1, a2, a3), op_bits(w), acc, 0, 0, 0);
}
template <int NB, class W4>
__device__ __forceinline__ void load_batch(W4 (&q)[NB], const W4* __restrict__ p, int s0, int nsteps) {
#pragma unroll
  for (int k = 0; k < NB; ++k) q[k] = s0 + k < nsteps ? wload(p + (long)(s0 + k) * 64) : W4{};
}
// the first NB steps of a wavefront's first unit of a stage, requested ahead of the barrier in front of the stage
template <int NB, class W4>
__device__ __forceinline__ void prefetch(W4 (&pf)[kPF], const W4* __restrict__ p, bool on) {
  static_assert(NB <= kPF, "prefetch registers");
  if (on) {
#pragma unroll
    for (int k = 0; k < NB; ++k) pf[k] = wload(p + (long)k * 64);
  }
}
// NS steps of one unit: body(step, float4); `first`: the steps 0..NB-1 are already in pf
template <int NS, int NB, class W4, class Body>
__device__ __forceinline__ void stream_unit(const W4* __restrict__ wq, const W4 (&pf)[kPF], bool first, Body body) {
  static_assert(NS % NB == 0, "whole batches");
  W4 cur[NB], nxt[NB];
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

// One dense layer (one row per patch): out[n] = bias[n] + sum_k x[k] W[n][k]; unit = (64 columns, quarter of K).
// The reduction + epilogue is the caller's (it differs per layer).
template <int G, int NW, int N, int NG, int SPK, int NB, class W4>
__device__ __forceinline__ void dense_units(const float* __restrict__ xv /* RV + x offset, patch stride RV_F */,
                                            const W4* __restrict__ wp4, float* __restrict__ slab, const float* __restrict__ bias,
                                            const W4 (&pf)[kPF]) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  constexpr int NGP = NG * 64;
  for (int u = wave; u < NG * 4; u += NW) {
    const int kg = u & 3, ng = u >> 2, n = 64 * ng + lane;
    const W4* wq = wp4 + (long)((ng * 4 + kg) * SPK) * 64 + lane;
    float acc[G];
    const float bv = (bias && kg == 0 && n < N) ? bias[n] : 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = bv;
    stream_unit<SPK, NB>(wq, pf, u == wave, [&](int s, const W4& wraw) {
      const f32x4 w4 = wide(wraw);  // (the dense stages multiply in fp32 in both forms)
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
template <int SPK, class W4>
__device__ __forceinline__ const W4* dense_wq(const W4* wp4, int u, int lane) {
  return wp4 + (long)(((u >> 2) * 4 + (u & 3)) * SPK) * 64 + lane;
}

template <int G, int THREADS, bool BWD, bool BFW>
__global__ __launch_bounds__(THREADS) void deep2d_kernel(const Deep2dArgs a) {
  using Y = Lay<BWD>;
  using W4 = typename std::conditional<BFW, bf16x4, f32x4>::type;
  constexpr int NW = THREADS / 64;
  static_assert(G == 1 || G == 2, "the reductions tell the patches apart with one compare");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const RA = smem;
  float* const RS = RA + G * RA_F;
  float* const RX = RS + G * RS_F;
  float* const RV = RX + G * RX_F;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lm = lane & 15, lk = lane >> 4;
  const int b0 = blockIdx.x * G;
  const W4* const wp4 = reinterpret_cast<const W4*>(a.wp);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  int nstamp = 0;
  auto stamp = [&]() {
    if (a.stamps && blockIdx.x == 0 && t == 0) a.stamps[nstamp] = (long long)__builtin_amdgcn_s_memtime();
    ++nstamp;
  };
  auto splat = [](float v) { return (f32x4){v, v, v, v}; };
  // epilogue of a summed pre-activation: forward ELU (the bias is already in), backward the ELU' multiply by the saved activation
  auto epi = [](float v, const float* __restrict__ saved, long idx) -> float {
    if (BWD) return saved ? v * elu_grad_from_out(saved[idx]) : v;
    return elu(v);
  };
  W4 pf[kPF];
  stamp();

  // first units of the stages, by wavefront (for the cross-barrier prefetch)
  auto wq_conv3 = [&](int u) { return wp4 + Y::P_CONV3 / 4 + (long)(((u >> 2) * 2 + (u & 1)) * 12) * 64 + lane; };
  auto wq_conv4 = [&](int u) { return wp4 + Y::P_CONV4 / 4 + (long)(((u >> 1) * 2 + (u & 1)) * 24) * 64 + lane; };
  auto wq_conv5 = [&](int u) { return wp4 + Y::P_CONV5 / 4 + (long)(((u >> 2) * 4 + (u & 3)) * 96) * 64 + lane; };
  auto wq_tconv0 = [&](int u) { return wp4 + Y::P_TCONV0 / 4 + (long)(((u >> 1) * 2 + (u & 1)) * 96) * 64 + lane; };
  auto wq_tconv1 = [&](int u) { return wp4 + Y::P_TCONV1 / 4 + (long)((((u >> 2) * 2 + ((u >> 1) & 1)) * 2 + (u & 1)) * 24) * 64 + lane; };
  auto wq_tconv2 = [&](int u) { return wp4 + Y::P_TCONV2 / 4 + (long)(((u >> 3) * 4 + ((u >> 1) & 3)) * 12) * 64 + lane; };
  auto wq_tconv3 = [&](int u) { return wp4 + Y::P_TCONV3 / 4 + (long)((u & 1) * 12) * 64 + lane; };

  // ---- the stage-0 input of the G patches into the padded image; forward: the harmonic-feature tails of cat1 / cat3
  prefetch<4>(pf, wq_conv3(wave), wave < 12);
  for (int i = t; i < G * RA_F / 4; i += THREADS) reinterpret_cast<f32x4*>(RA)[i] = zero4;
  for (int i = t; i < G * RX_F / 4; i += THREADS) reinterpret_cast<f32x4*>(RX)[i] = zero4;  // (RX only ever holds the two 48 x 10 x 10 images: their border is written once)
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
  if (!BWD) {
    for (int i = t; i < G * 2 * kHd; i += THREADS) {
      const int g = i / (2 * kHd), j = i - g * 2 * kHd;
      const bool ok = b0 + g < a.B;
      if (j < kHd) RV[g * RV_F + V_0 + 768 + j] = ok ? a.cat1[(long)(b0 + g) * a.cat1_ld + 768 + j] : 0.f;
      else RV[g * RV_F + V_3 + kL + j - kHd] = ok ? a.cat3[(long)(b0 + g) * a.cat3_ld + kL + j - kHd] : 0.f;
    }
  }
  __syncthreads();
  stamp();

  // ==== stage 0 (conv3 | tconv2'): 24 x 16 x 16 -> 48 x 8 x 8.  M = 64 positions (4 m-tiles of two output rows), N = 48,
  // K = 24 x 16.  unit = (n-tile, half of the m-tiles, half of the channels): 12 units x 96 matrix instructions
  for (int u = wave; u < 12; u += NW) {
    const int kg = u & 1, mh = (u >> 1) & 1, nt = u >> 2;
    f32x4 acc[G][2];
    const f32x4 b4v = splat((!BWD && kg == 0) ? a.b3[16 * nt + lm] : 0.f);
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g][0] = acc[g][1] = b4v;
    stream_unit<12, 4>(wq_conv3(u), pf, u == wave, [&](int s, const W4& w4) {
      const int ci = 12 * kg + s;
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const int oy = 2 * (2 * mh + mi) + (lm >> 3), ox = lm & 7;
          const float* ap = RA + g * RA_F + ci * X2_CP + (2 * oy + lk) * X2_RP + 2 * ox;
          const float2 a01 = *reinterpret_cast<const float2*>(ap), a23 = *reinterpret_cast<const float2*>(ap + 2);
          acc[g][mi] = mma16(a01.x, a01.y, a23.x, a23.y, w4, acc[g][mi]);
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
  for (int i = t; i < G * 3072; i += THREADS) {  // interior of the padded 10 x 10 images (the border was cleared at the start)
    const int g = i >= 3072 ? 1 : 0, o = i - g * 3072, n = o >> 6, m = o & 63;
    float v = 0.f;
    if (b0 + g < a.B) {
      const float* sp = RS + g * RS_F + o;
      v = epi(sp[0] + sp[3072], a.s3, (long)(b0 + g) * 3072 + o);
      a.a3[(long)(b0 + g) * 3072 + o] = v;
    }
    RX[g * RX_F + n * X3_CP + ((m >> 3) + 1) * X3_RP + (m & 7) + 1] = v;
  }
  __syncthreads();
  stamp();

  // ==== stage 1 (conv4 | tconv1'): 48 x 8 x 8 -> 96 x 4 x 4.  M = 16, N = 96 (6 n-tiles), K = 48 x 16; unit = (n-tile, half of the channels)
  for (int u = wave; u < 12; u += NW) {
    const int kg = u & 1, nt = u >> 1;
    f32x4 acc[G];
    const f32x4 b4v = splat((!BWD && kg == 0) ? a.b4[16 * nt + lm] : 0.f);
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = b4v;
    stream_unit<24, 8>(wq_conv4(u), pf, u == wave, [&](int s, const W4& w4) {
      const int ci = 24 * kg + s;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int oy = lm >> 2, ox = lm & 3;
        const float* ap = RX + g * RX_F + ci * X3_CP + (2 * oy + lk) * X3_RP + 2 * ox;
        const float2 a01 = *reinterpret_cast<const float2*>(ap), a23 = *reinterpret_cast<const float2*>(ap + 2);
        acc[g] = mma16(a01.x, a01.y, a23.x, a23.y, w4, acc[g]);
      }
    });
#pragma unroll
    for (int g = 0; g < G; ++g) *reinterpret_cast<f32x4*>(RA + g * RA_F + kg * 1536 + (16 * nt + lm) * 16 + 4 * lk) = acc[g];
  }
  prefetch<8>(pf, wq_conv5(wave), wave < 12);
  __syncthreads();
  stamp();
  // stage 1's output as rows of two 4-wide windows: record (channel, padded row) = [cols -1..2 | cols 1..4], so that the
  // 4 x 4 window of stage 2's output position (oy, ox) is four aligned ds_read_b128 at column 4 ox
  for (int i = t; i < G * 96 * 24; i += THREADS) {  // the 24 border slots of a channel: rows 0 and 5, slots 0 and 7 of rows 1..4
    const int g = i >= 96 * 24 ? 1 : 0, j = i - g * 96 * 24, n = j / 24, k = j - n * 24;
    const int o = k < 8 ? k : k < 16 ? 40 + k - 8 : 8 * (1 + ((k - 16) >> 1)) + ((k & 1) ? 7 : 0);
    RS[g * RS_F + n * X4_CP + o] = 0.f;
  }
  for (int i = t; i < G * 1536; i += THREADS) {
    const int g = i >= 1536 ? 1 : 0, o = i - g * 1536, n = o >> 4, y = (o >> 2) & 3, pc = (o & 3) + 1;
    float v = 0.f;
    if (b0 + g < a.B) {
      const float* sp = RA + g * RA_F + o;
      v = epi(sp[0] + sp[1536], a.s4, (long)(b0 + g) * 1536 + o);
      a.a4[(long)(b0 + g) * 1536 + o] = v;
    }
    float* rec = RS + g * RS_F + n * X4_CP + (y + 1) * 8;
    if (pc <= 3) rec[pc] = v;       // window of ox = 0: columns -1..2
    if (pc >= 2) rec[2 + pc] = v;   // window of ox = 1: columns 1..4
  }
  __syncthreads();
  stamp();

  // ==== stage 2 (conv5 | tconv0'): 96 x 4 x 4 -> 192 x 2 x 2.  M = 4 positions: v_mfma_f32_4x4x1 (16 blocks = 64 output channels,
  // one lane each; block row r = position r).  unit = (64 channels, quarter of the input channels): 12 units x 384 instructions
  for (int u = wave; u < 12; u += NW) {
    const int kg = u & 3, ng = u >> 2;
    f32x4 acc[G];
    const f32x4 b4v = splat((!BWD && kg == 0) ? a.b5[64 * ng + lane] : 0.f);
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = b4v;
    const int oy = (lane >> 1) & 1, ox = lane & 1;
    stream_unit<96, 8>(wq_conv5(u), pf, u == wave, [&](int s, const W4& w4) {
      const int ci = 24 * kg + (s >> 2), ky = s & 3;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(RS + g * RS_F + ci * X4_CP + (2 * oy + ky) * 8 + 4 * ox);
        acc[g] = mma4(av[0], av[1], av[2], av[3], w4, acc[g]);
      }
    });
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) RA[g * RA_F + kg * 768 + r * 192 + 64 * ng + lane] = acc[g][r];
  }
  prefetch<Y::NB1>(pf, dense_wq<Y::SPK1>(wp4 + Y::P_FC1 / 4, wave, lane), wave < 16);
  __syncthreads();
  stamp();
  for (int i = t; i < G * 768; i += THREADS) {
    const int g = i / 768, j = i - g * 768, n = j >> 2, m = j & 3;
    const float* sp = RA + g * RA_F + m * 192 + n;
    float v = (sp[0] + sp[768]) + (sp[1536] + sp[2304]);
    if (!BWD) v = elu(v);  // (backward: the gradient w.r.t. fc3's output, which has no activation)
    RV[g * RV_F + V_0 + j] = v;
    if (b0 + g < a.B) a.cat1[(long)(b0 + g) * a.cat1_ld + j] = v;
  }
  __syncthreads();
  stamp();

  // ==== the four dense stages: fc1 -> fc2in (the latent code) -> fc2out -> fc3 | fc3' -> fc2out' -> fc2in' -> fc1'
  dense_units<G, NW, Y::N1, 4, Y::SPK1, Y::NB1>(RV + V_0, wp4 + Y::P_FC1 / 4, RA, a.bfc1, pf);
  prefetch<7>(pf, dense_wq<14>(wp4 + Y::P_FC2IN / 4, wave, lane), wave < 16);
  __syncthreads();
  for (int i = t; i < G * Y::N1; i += THREADS) {
    const int g = i / Y::N1, n = i - g * Y::N1;
    if (b0 + g < a.B) {
      const float v = epi(dense_sum<4>(RA, g, n), a.sd1, (long)(b0 + g) * a.z1_ld + n);
      RV[g * RV_F + V_1 + n] = v;
      a.z1[(long)(b0 + g) * a.z1_ld + n] = v;
    } else {
      RV[g * RV_F + V_1 + n] = 0.f;
    }
  }
  __syncthreads();
  stamp();
  dense_units<G, NW, kL, 4, 14, 7>(RV + V_1, wp4 + Y::P_FC2IN / 4, RA, a.bfc2in, pf);
  prefetch<7>(pf, dense_wq<14>(wp4 + Y::P_FC2OUT / 4, wave, lane), wave < 16);
  __syncthreads();
  for (int i = t; i < G * kL; i += THREADS) {
    const int g = i / kL, n = i - g * kL;
    if (b0 + g < a.B) {
      float v = dense_sum<4>(RA, g, n);
      if (BWD && a.gmu) v += a.gmu[(long)(b0 + g) * a.gmu_ld + n];  // the latent-space terms' gradient w.r.t. the code
      v = epi(v, a.sd2, (long)(b0 + g) * a.sd2_ld + n);
      RV[g * RV_F + V_2 + n] = v;
      a.mu[(long)(b0 + g) * a.mu_ld + n] = v;
    } else {
      RV[g * RV_F + V_2 + n] = 0.f;
    }
  }
  __syncthreads();
  stamp();
  dense_units<G, NW, kL, 4, 14, 7>(RV + V_2, wp4 + Y::P_FC2OUT / 4, RA, a.bfc2out, pf);
  prefetch<Y::NB4>(pf, dense_wq<Y::SPK4>(wp4 + Y::P_FC3 / 4, wave, lane), wave < 4 * Y::NG4);
  __syncthreads();
  for (int i = t; i < G * kL; i += THREADS) {
    const int g = i / kL, n = i - g * kL;
    if (b0 + g < a.B) {
      const float v = epi(dense_sum<4>(RA, g, n), a.sd3, (long)(b0 + g) * a.cat3_ld + n);
      RV[g * RV_F + V_3 + n] = v;
      a.cat3[(long)(b0 + g) * a.cat3_ld + n] = v;
    } else {
      RV[g * RV_F + V_3 + n] = 0.f;
    }
  }
  __syncthreads();
  stamp();
  dense_units<G, NW, Y::N4, Y::NG4, Y::SPK4, Y::NB4>(RV + V_3, wp4 + Y::P_FC3 / 4, RA, a.bfc3, pf);
  prefetch<8>(pf, wq_tconv0(wave), wave < 12);
  __syncthreads();
  // dense 4's first 768 outputs (192 x 2 x 2; forward: fc3 has no activation, src/lofar_models.py:91) as padded rows of
  // column PAIRS: record (channel, padded row) = [cols (0,1) | (1,2) | (2,3) | unused]: the two column taps of a
  // transposed-conv lane are one ds_read_b64
  for (int i = t; i < G * 192 * 16; i += THREADS) {  // the 16 border slots of a channel that are read: rows 0 and 3 (slots 0..5), slots 0 and 5 of rows 1, 2
    const int g = i >= 192 * 16 ? 1 : 0, j = i - g * 192 * 16, ci = j >> 4, k = j & 15;
    const int o = k < 6 ? k : k < 12 ? 24 + k - 6 : 8 * (1 + ((k - 12) >> 1)) + ((k & 1) ? 5 : 0);
    RS[g * RS_F + ci * D0_CP + o] = 0.f;
  }
  for (int i = t; i < G * 768; i += THREADS) {
    const int g = i >= 768 ? 1 : 0, n = i - g * 768, ci = n >> 2, pr = ((n >> 1) & 1) + 1, pc = (n & 1) + 1;
    float v = 0.f;
    if (b0 + g < a.B) {
      v = dense_sum<Y::NG4>(RA, g, n);
      if (BWD) v = epi(v, a.sd4, (long)(b0 + g) * a.d0_ld + n);
      a.d0[(long)(b0 + g) * a.d0_ld + n] = v;
    }
    float* rec = RS + g * RS_F + ci * D0_CP + pr * 8;  // column pc sits in the pairs (pc - 1, pc) and (pc, pc + 1): slots 2 pc - 1, 2 pc
    rec[2 * pc - 1] = v;
    rec[2 * pc] = v;
  }
  if (BWD) {  // fc1' has 16 more outputs: the gradient w.r.t. elu(fcuv1) (the weight gradient of fcuv1 reads it)
    for (int i = t; i < G * kHd; i += THREADS) {
      const int g = i / kHd, n = 768 + i - g * kHd;
      if (b0 + g < a.B) a.d0[(long)(b0 + g) * a.d0_ld + n] = epi(dense_sum<Y::NG4>(RA, g, n), a.sd4, (long)(b0 + g) * a.d0_ld + n);
    }
  }
  __syncthreads();
  stamp();

  // ==== stage 7 (tconv0 | conv5'): 192 x 2 x 2 -> 96 x 4 x 4 on v_mfma_f32_4x4x1: block = (output parity, channel quad), block
  // row r = input position r, k = (input channel, tap).  unit = (16 channels, half of the input channels): 12 units x 384
  for (int u = wave; u < 12; u += NW) {
    const int kg = u & 1, c = u >> 1;
    const int p = lane >> 4, py = p >> 1, px = p & 1, iy = (lane >> 1) & 1, ix = lane & 1;
    const int co = 16 * c + (lane & 15);
    f32x4 acc[G];
    const f32x4 b4v = splat((!BWD && kg == 0) ? a.bt0[co] : 0.f);
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = b4v;
    const int aoff = (iy + py) * 8 + 2 * (ix + px);  // rows iy + py - 1 + j, column pair (ix + px - 1, ix + px), padded by one
    stream_unit<96, 8>(wq_tconv0(u), pf, u == wave, [&](int s, const W4& w4) {
      const int ci = 96 * kg + s;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float* ap = RS + g * RS_F + ci * D0_CP + aoff;
        const float2 r0 = *reinterpret_cast<const float2*>(ap), r1 = *reinterpret_cast<const float2*>(ap + 8);
        acc[g] = mma4(r0.x, r0.y, r1.x, r1.y, w4, acc[g]);
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
  for (int i = t; i < G * 96 * 20; i += THREADS) {  // the 20 border cells of a channel that are read (6 x 6 around the 4 x 4 interior)
    const int g = i >= 96 * 20 ? 1 : 0, j = i - g * 96 * 20, co = j / 20, k = j - co * 20;
    const int o = k < 6 ? k : k < 12 ? 5 * T0_RP + k - 6 : (1 + ((k - 12) >> 1)) * T0_RP + ((k & 1) ? 5 : 0);
    RS[g * RS_F + co * T0_CP + o] = 0.f;
  }
  for (int i = t; i < G * 1536; i += THREADS) {
    const int g = i >= 1536 ? 1 : 0, o = i - g * 1536, co = o >> 4, pos = o & 15;
    float v = 0.f;
    if (b0 + g < a.B) {
      const float* sp = RA + g * RA_F + o;
      v = epi(sp[0] + sp[1536], a.st0, (long)(b0 + g) * 1536 + o);
      a.t0[(long)(b0 + g) * 1536 + o] = v;
    }
    RS[g * RS_F + co * T0_CP + ((pos >> 2) + 1) * T0_RP + (pos & 3) + 1] = v;
  }
  __syncthreads();
  stamp();

  // ==== stage 8 (tconv1 | conv4'): 96 x 4 x 4 -> 48 x 8 x 8, per output parity M = 16 input positions, N = 48, K = 96 x 4 taps (k-step
  // = four input channels at one tap).  unit = (16 channels, row parity, half of the input channels): 12 units x 96 instructions
  for (int u = wave; u < 12; u += NW) {
    const int kg = u & 1, py = (u >> 1) & 1, c = u >> 2;
    const int iy = lm >> 2, ix = lm & 3, co = 16 * c + lm;
    f32x4 acc[G][2];
    const f32x4 b4v = splat((!BWD && kg == 0) ? a.bt1[co] : 0.f);
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g][0] = acc[g][1] = b4v;
    stream_unit<24, 8>(wq_tconv1(u), pf, u == wave, [&](int s, const W4& w4) {
      const int ci = 4 * (12 * kg + (s >> 1)) + lk, jy = s & 1;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float* ap = RS + g * RS_F + ci * T0_CP + (iy + py + jy) * T0_RP + ix;  // padded columns ix, ix + 1, ix + 2
        const float am = ap[0], a0 = ap[1], a1 = ap[2];
        // column parity 0: columns ix - 1, ix through kernel columns 3, 1; parity 1: columns ix, ix + 1 through 2, 0
        acc[g][0] = mma16_2<3, 1>(am, a0, w4, acc[g][0]);
        acc[g][1] = mma16_2<2, 0>(a0, a1, w4, acc[g][1]);
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
  for (int i = t; i < G * 3072; i += THREADS) {
    const int g = i >= 3072 ? 1 : 0, o = i - g * 3072, co = o >> 6, pos = o & 63;
    float v = 0.f;
    if (b0 + g < a.B) {
      const float* sp = RA + g * RA_F + o;
      v = epi(sp[0] + sp[3072], a.st1, (long)(b0 + g) * 3072 + o);
      a.t1[(long)(b0 + g) * 3072 + o] = v;
    }
    RX[g * RX_F + co * T1_CP + ((pos >> 3) + 1) * T1_RP + (pos & 7) + 1] = v;
  }
  __syncthreads();
  stamp();

  // ==== stage 9 (tconv2 | conv3'): 48 x 8 x 8 -> 24 x 16 x 16, per output parity M = 64 (4 m-tiles of two input rows), N = 24 (two
  // n-tiles, the second half empty), K = 48 x 4 taps.  unit = (n-tile, parity, half of the m-tiles), whole K: 16 units x 96.
  // The output goes to RA in the padded form the last stage reads (the slab of stage 8 is dead); its border is cleared here.
  for (int i = t; i < G * 24 * 68; i += THREADS) {
    const int g = i / (24 * 68), j = i - g * (24 * 68), ch = j / 68, k = j - ch * 68;
    const int o = k < 18 ? k : k < 36 ? 17 * X2_RP + k - 18 : (1 + ((k - 36) >> 1)) * X2_RP + ((k & 1) ? 17 : 0);
    RA[g * RA_F + ch * X2_CP + o] = 0.f;
  }
  for (int u = wave; u < 16; u += NW) {
    const int mh = u & 1, p = (u >> 1) & 3, c = u >> 3, py = p >> 1, px = p & 1;
    const int co = 16 * c + lm;
    f32x4 acc[G][2];
    const f32x4 b4v = splat((!BWD && co < 24) ? a.bt2[co] : 0.f);
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g][0] = acc[g][1] = b4v;
    stream_unit<12, 4>(wq_tconv2(u), pf, u == wave, [&](int s, const W4& w4) {
      const int ci = 4 * s + lk;
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const int iy = 2 * (2 * mh + mi) + (lm >> 3), ix = lm & 7;
          const float* ap = RX + g * RX_F + ci * T1_CP + (iy + py) * T1_RP + ix + px;  // rows iy + py - 1 + j, columns ix + px - 1 + j
          acc[g][mi] = mma16(ap[0], ap[1], ap[T1_RP], ap[T1_RP + 1], w4, acc[g][mi]);
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
            const float v = acc[g][mi][r];
            RA[g * RA_F + co * X2_CP + (2 * iy + py + 1) * X2_RP + 2 * ix + px + 1] = BWD ? v : elu(v);
          }
    }
  }
  prefetch<4>(pf, wq_tconv3(wave), wave < 16);
  __syncthreads();
  stamp();
  if (BWD) {  // the ELU' multiply in place (coalesced reads of the saved activation), then the copy to HBM
    for (int i = t; i < G * 6144; i += THREADS) {
      const int g = i / 6144, j = i - g * 6144, ch = j >> 8, y = (j >> 4) & 15, x = j & 15;
      float* q = RA + g * RA_F + ch * X2_CP + (y + 1) * X2_RP + x + 1;
      if (b0 + g < a.B) {
        const float v = epi(*q, a.st2, (long)(b0 + g) * 6144 + j);
        *q = v;
        a.t2[(long)(b0 + g) * 6144 + j] = v;
      }
    }
    __syncthreads();
  } else {
    for (int i = t; i < G * 6144; i += THREADS) {
      const int g = i / 6144, j = i - g * 6144, ch = j >> 8, y = (j >> 4) & 15, x = j & 15;
      if (b0 + g < a.B) a.t2[(long)(b0 + g) * 6144 + j] = RA[g * RA_F + ch * X2_CP + (y + 1) * X2_RP + x + 1];
    }
  }

  // ==== stage 10 (tconv3 | conv2'): 24 x 16 x 16 -> 12 x 32 x 32, per output parity M = 256 (16 m-tiles = input rows), N = 12 (one
  // n-tile), K = 24 x 4 taps.  unit = (row parity, two input rows), both column parities: 16 units x 96 instructions; a lane
  // ends with eight consecutive outputs of a row (two float4 stores, straight to HBM)
  for (int u = wave; u < 16; u += NW) {
    const int py = u & 1, mq = u >> 1;
    f32x4 acc[G][2][2];
    const f32x4 b4v = splat((!BWD && lm < 12) ? a.bt3[lm] : 0.f);
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g][0][0] = acc[g][0][1] = acc[g][1][0] = acc[g][1][1] = b4v;
    stream_unit<12, 4>(wq_tconv3(u), pf, u == wave, [&](int s, const W4& w4) {
      const int ci = 4 * (s >> 1) + lk, jy = s & 1;
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const int iy = 2 * mq + mi;
          const float* ap = RA + g * RA_F + ci * X2_CP + (iy + py + jy) * X2_RP + lm;  // padded columns ix, ix + 1, ix + 2 (ix = lm)
          const float am = ap[0], a0 = ap[1], a1 = ap[2];
          acc[g][mi][0] = mma16_2<3, 1>(am, a0, w4, acc[g][mi][0]);
          acc[g][mi][1] = mma16_2<2, 0>(a0, a1, w4, acc[g][mi][1]);
        }
    });
    f32x4 sv[G][2][2];  // backward: the saved activation at the lane's eight outputs per row, (requested after the matrix work: ahead of it the registers spill)
    if (BWD && a.st3 && lm < 12) {
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const float* sp = a.st3 + (long)(b0 + g < a.B ? b0 + g : 0) * 12288 + lm * 1024 + (2 * (2 * mq + mi) + py) * 32 + 8 * lk;
          sv[g][mi][0] = *reinterpret_cast<const f32x4*>(sp);
          sv[g][mi][1] = *reinterpret_cast<const f32x4*>(sp + 4);
        }
    }
    if (lm < 12) {
#pragma unroll
      for (int g = 0; g < G; ++g)
        if (b0 + g < a.B) {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) {  // rows m = 4 lk + r = input columns -> output columns 8 lk + 2 r + px of row 2 iy + py
            float* op = a.t3 + (long)(b0 + g) * 12288 + lm * 1024 + (2 * (2 * mq + mi) + py) * 32 + 8 * lk;
            f32x4 lo = {acc[g][mi][0][0], acc[g][mi][1][0], acc[g][mi][0][1], acc[g][mi][1][1]};
            f32x4 hi = {acc[g][mi][0][2], acc[g][mi][1][2], acc[g][mi][0][3], acc[g][mi][1][3]};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (BWD) {
                if (a.st3) { lo[e] *= elu_grad_from_out(sv[g][mi][0][e]); hi[e] *= elu_grad_from_out(sv[g][mi][1][e]); }
              } else {
                lo[e] = elu(lo[e]); hi[e] = elu(hi[e]);
              }
            }
            *reinterpret_cast<f32x4*>(op) = lo;
            *reinterpret_cast<f32x4*>(op + 4) = hi;
          }
        }
    }
  }
  stamp();
}

}  // namespace

size_t deep2d_packed_floats() { return (size_t)kPackedMax; }

bool deep2d_supported(int L, int hd, int rica, const int* enc_ch /* the output channels of conv1 .. conv5 */, int H2 /* conv2's output height */) {
  return L == kL && hd == kHd && rica && enc_ch[0] == 12 && enc_ch[1] == 24 && enc_ch[2] == 48 && enc_ch[3] == 96 && enc_ch[4] == 192 && H2 == 16 &&
         device_lds_fits(sizeof(float) * 2 * (RA_F + RS_F + RX_F + RV_F));  // (two patches per workgroup: 158 KB)
}

int deep2d_pack(const Deep2dWeights& w, float* packed, int backward, int bf16_weights, hipStream_t st) {
  if (!packed || (backward && !w.c2) || !w.c3 || !w.c4 || !w.c5 || !w.fc1 || !w.fc2in || !w.fc2out || !w.fc3 || !w.t0 || !w.t1 || !w.t2 || !w.t3) {
    set_last_error("deep2d_pack: null pointer");
    return LSHM_ERR_ARG;
  }
  if (reinterpret_cast<uintptr_t>(packed) & 15) { set_last_error("deep2d_pack: unaligned destination"); return LSHM_ERR_ARG; }
  // backward: the data-gradient pipeline -- every conv-shaped stage reads the tensor of the layer it differentiates as it stands
  // (tconv2' = conv with tconv2.weight (48, 24, 4, 4) ... conv2' = transposed conv with conv2.weight (24, 12, 4, 4)), the
  // dense stages read theirs transposed: fc3' | fc2out' | fc2in' | fc1'
  const PackSrc sb{w.t2, w.t1, w.t0, w.fc3, w.fc2out, w.fc2in, w.fc1, w.c5, w.c4, w.c3, w.c2};
  const PackSrc sf{w.c3, w.c4, w.c5, w.fc1, w.fc2in, w.fc2out, w.fc3, w.t0, w.t1, w.t2, w.t3};
  const dim3 gb(cdiv(Lay<true>::P_TOTAL / 4, 256)), gf(cdiv(Lay<false>::P_TOTAL / 4, 256));
  if (backward && bf16_weights) hipLaunchKernelGGL((deep2d_pack_kernel<true, true>), gb, dim3(256), 0, st, sb, packed);
  else if (backward) hipLaunchKernelGGL((deep2d_pack_kernel<true, false>), gb, dim3(256), 0, st, sb, packed);
  else if (bf16_weights) hipLaunchKernelGGL((deep2d_pack_kernel<false, true>), gf, dim3(256), 0, st, sf, packed);
  else hipLaunchKernelGGL((deep2d_pack_kernel<false, false>), gf, dim3(256), 0, st, sf, packed);
  return check_launch("deep2d_pack");
}

template <int G, int THREADS, bool BWD, bool BFW>
static int launch_deep(const Deep2dArgs& a, hipStream_t st) {
  const size_t lds = sizeof(float) * (size_t)G * (RA_F + RS_F + RX_F + RV_F);
  auto kern = deep2d_kernel<G, THREADS, BWD, BFW>;
  int rc;
  if ((rc = kernel_budget_ok(reinterpret_cast<const void*>(kern), THREADS, lds, "deep2d"))) return rc;
  if ((rc = raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds, "deep2d"))) return rc;
  hipLaunchKernelGGL(kern, dim3(cdiv(a.B, G)), dim3(THREADS), lds, st, a);
  return check_launch("deep2d");
}

int deep2d_fwd(const Deep2dIO& io, const float* packed, int B, int variant, hipStream_t st) {
  auto al16 = [](const void* q) { return q && (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const bool ok = al16(packed) && al16(io.x2) && al16(io.t3) && io.t2 && io.a3 && io.a4 && io.cat1 && io.z1 && io.mu && io.cat3 && io.d0 &&
                  io.t0 && io.t1 && io.b3 && io.b4 && io.b5 && io.bfc1 && io.bfc2in && io.bfc2out && io.bfc3 && io.bt0 && io.bt1 &&
                  io.bt2 && io.bt3 && B > 0;
  if (!ok) { set_last_error("deep2d_fwd: null / unaligned pointer"); return LSHM_ERR_ARG; }
  Deep2dArgs a{};
  a.wp = packed; a.x2 = io.x2;
  a.b3 = io.b3; a.b4 = io.b4; a.b5 = io.b5; a.bfc1 = io.bfc1; a.bfc2in = io.bfc2in; a.bfc2out = io.bfc2out; a.bfc3 = io.bfc3;
  a.bt0 = io.bt0; a.bt1 = io.bt1; a.bt2 = io.bt2; a.bt3 = io.bt3;
  a.a3 = io.a3; a.a4 = io.a4; a.cat1 = io.cat1; a.cat1_ld = 768 + kHd; a.z1 = io.z1; a.z1_ld = kL; a.mu = io.mu; a.mu_ld = io.mu_ld;
  a.cat3 = io.cat3; a.cat3_ld = kL + kHd; a.d0 = io.d0; a.d0_ld = 768;
  a.t0 = io.t0; a.t1 = io.t1; a.t2 = io.t2; a.t3 = io.t3;
  a.B = B;
  a.stamps = io.stamps;
  switch (variant) {  // bit 2 (+4): the packed weights are bf16
    case 0: return launch_deep<1, 1024, false, false>(a, st);
    case 1: return launch_deep<2, 1024, false, false>(a, st);
    case 2: return launch_deep<1, 512, false, false>(a, st);
    case 4: return launch_deep<1, 1024, false, true>(a, st);
    case 5: return launch_deep<2, 1024, false, true>(a, st);
    default: set_last_error("deep2d_fwd: unknown variant"); return LSHM_ERR_ARG;
  }
}

int deep2d_bwd(const Deep2dBwdIO& io, const float* packed, int B, int variant, hipStream_t st) {
  auto al16 = [](const void* q) { return q && (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const bool ok = al16(packed) && al16(io.g_t2) && al16(io.g_c1) && al16(io.s_c1) && io.s_t1 && io.s_t0 && io.s_cat3 && io.s_mu && io.s_z1 &&
                  io.s_cat1 && io.s_c4 && io.s_c3 && io.s_c2 && io.g_t1 && io.g_t0 && io.g_d0 && io.g_cat3 && io.g_mu && io.g_z1 &&
                  io.g_cat1 && io.g_c4 && io.g_c3 && io.g_c2 && B > 0;
  if (!ok) { set_last_error("deep2d_bwd: null / unaligned pointer"); return LSHM_ERR_ARG; }
  Deep2dArgs a{};
  a.wp = packed; a.x2 = io.g_t2;
  a.s3 = io.s_t1; a.s4 = io.s_t0; a.sd1 = io.s_cat3; a.sd2 = io.s_mu; a.sd2_ld = io.s_mu_ld; a.sd3 = io.s_z1; a.sd4 = io.s_cat1;
  a.st0 = io.s_c4; a.st1 = io.s_c3; a.st2 = io.s_c2; a.st3 = io.s_c1;
  a.gmu = io.gmu; a.gmu_ld = io.gmu_ld;
  a.a3 = io.g_t1; a.a4 = io.g_t0; a.cat1 = io.g_d0; a.cat1_ld = 768; a.z1 = io.g_cat3; a.z1_ld = kL + kHd; a.mu = io.g_mu; a.mu_ld = io.g_mu_ld;
  a.cat3 = io.g_z1; a.cat3_ld = kL; a.d0 = io.g_cat1; a.d0_ld = 768 + kHd;
  a.t0 = io.g_c4; a.t1 = io.g_c3; a.t2 = io.g_c2; a.t3 = io.g_c1;
  a.B = B;
  a.stamps = io.stamps;
  switch (variant) {
    case 0: return launch_deep<1, 1024, true, false>(a, st);
    case 1: return launch_deep<2, 1024, true, false>(a, st);
    case 4: return launch_deep<1, 1024, true, true>(a, st);
    case 5: return launch_deep<2, 1024, true, true>(a, st);
    default: set_last_error("deep2d_bwd: unknown variant"); return LSHM_ERR_ARG;
  }
}

}  // namespace lshm
