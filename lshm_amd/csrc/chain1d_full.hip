// The whole mid + deep section of AutoEncoder1DCNN as ONE launch per direction (round 4; src/lofar_models.py:119-135,
// 137-140 and the forward :158-183): conv2 -> conv3 -> conv4 -> conv5 -> fc1 -> fc2in -> fc2out -> fc3 -> tconv0 -> tconv1 ->
// tconv2 -> tconv3 for one (patch, network) per workgroup, activations resident in LDS.  It joins what chain1d.hip
// (two three-layer chains), two implicit-GEMM launches (conv5, tconv0) and dense1d.hip (the dense middle, 16 patches
// per workgroup) did in five launches of 10-45 us per forward: ~115 us of kernel time per forward for 2.4 MMAC and
// 0.9 MB of weights per patch -- latency, not work.  The three-layer stages are chain1d's (chain1d_dev.h); new here are
// the two 4-position stages (conv5: 96 x 16 -> 192 x 4, tconv0: 192 x 4 -> 96 x 16; one 16-row MFMA tile with 4 rows in
// use) and the four dense layers of ONE patch on the vector ALU (a wavefront per output for the reductions over 784
// inputs, a thread per output for fc3).  Weights are read in the layers' own layouts (0.9 MB per network: L2-resident,
// every stage requests all of a tile's fragments before its matrix instructions).
//
// LDS: 19,008 floats (76 KB, two workgroups per CU) -- the down chain's images, then the vectors of the dense middle
// and fc3's output image, then the up chain's images, each aliasing what died before it (layout in the kernel).
#include <stdlib.h>

#include "kernels.h"
#include "chain1d_dev.h"
#include "chain1d_full.h"

namespace lshm {
namespace {

constexpr int kLt = 16, kHd1 = 16, kCat = 768 + kHd1;

// stride-4 conv from 16 positions to 4 (one m-tile, rows >= 4 unused): Y[n][j] = sum_{ci,t} X[ci][4j - pad + t] W[n][ci*4 + t].
// X: LDS image (pitch PIN, position p at xs[c*PIN + p + 1]); Y: vector yv[n*4 + j] (the flattened (COUT, 4) tensor).
template <int CIN, int COUT, int PIN>
__device__ __forceinline__ void down_small(const float* __restrict__ xs, float* __restrict__ yv, const float* __restrict__ w,
                                           const float* __restrict__ bias, int act, int pad) {
  constexpr int K = CIN * 4, NT = COUT / 16;
  static_assert(K % 16 == 0 && COUT % 16 == 0, "whole k-blocks and n-tiles");
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lm = lane & 15, lk = lane >> 4;
  for (int f = wave; f < NT; f += kChainThreads / 64) {
    const int n = 16 * f + lm;
    const float* wrow = w + (long)n * K + 4 * lk;
    const float* arow = xs + lk * PIN + 4 * (lm & 3) + 1 - pad;  // rows 4..15 of the tile repeat rows 0..3 (unused)
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    constexpr int KC = 12;  // k-blocks per batch of weight fragments in flight
    static_assert((K / 16) % KC == 0, "whole batches");
#pragma unroll
    for (int s0 = 0; s0 < K / 16; s0 += KC) {
      f32x4 bq[KC];
#pragma unroll
      for (int s = 0; s < KC; ++s) bq[s] = *reinterpret_cast<const f32x4*>(wrow + 16 * (s0 + s));
#pragma unroll
      for (int s = 0; s < KC; ++s) {
        const float* ap = arow + 4 * (s0 + s) * PIN;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[e], bq[s][e], acc, 0, 0, 0);
      }
    }
    if (lk == 0) {  // D rows m = 4 lk + r: the four positions
      const float bv = bias ? bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = acc[r] + bv;
        yv[n * 4 + r] = act ? elu(v) : v;
      }
    }
  }
}

// stride-4 transposed conv from 4 positions to 16: Y[co][4i + t - pad] = sum_ci X[ci][i] W[ci][co*4 + t].
// X: vector xv[ci*4 + i]; Y: LDS image, the value for logical position 4i + t - pad at ys[co*POUT + 4i + t] (up_stage's rule).
template <int CIN, int COUT, int POUT>
__device__ __forceinline__ void up_small(const float* __restrict__ xv, float* __restrict__ ys, const float* __restrict__ w,
                                         const float* __restrict__ bias, int act) {
  constexpr int CBT = COUT / 16, KS = CIN / 4;
  static_assert(CIN % 4 == 0 && COUT % 16 == 0 && POUT % 4 == 0, "whole k-steps and tiles, float4 rows");
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lm = lane & 15, lk = lane >> 4;
  for (int f = wave; f < CBT; f += kChainThreads / 64) {
    const int co = 16 * f + lm;
    const float* wp = w + ((long)lk * COUT + co) * 4;
    const float* ap = xv + lk * 4 + (lm & 3);  // rows 4..15 of the tile repeat rows 0..3 (unused)
    f32x4 acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int KC = 8;
    static_assert(KS % KC == 0, "whole batches of k-steps");
#pragma unroll
    for (int s0 = 0; s0 < KS; s0 += KC) {
      f32x4 bq[KC];
#pragma unroll
      for (int s = 0; s < KC; ++s) bq[s] = *reinterpret_cast<const f32x4*>(wp + (long)16 * (s0 + s) * COUT);
#pragma unroll
      for (int s = 0; s < KC; ++s) {
        const float a = ap[16 * (s0 + s)];  // channel 4 (s0 + s) + lk, position lm & 3
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bq[s][e], acc[e], 0, 0, 0);
      }
    }
    if (lk == 0) {  // D rows m = r: input positions 0..3
      const float bv = bias ? bias[co] : 0.f;
      float* yp = ys + co * POUT;
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

// LDS offsets (floats).  Down chain: x0 12 x 1025 | y1 24 x 257, y2 / y3 inside x0.  Middle: vectors + fc3's output at
// kVec.. (inside the dead x0).  Up chain: y3 12 x 1040 | y2 24 x 272, its input 96 x 48 and stage 1 48 x 80 inside y3.
constexpr int kP0d = pitch_down(1024), kP1d = pitch_down(256), kP2d = pitch_down(64), kP3d = pitch_down(16);
constexpr int kP0u = pitch_up(16), kP1u = pitch_up(64), kP2u = pitch_up(256), kP3u = pitch_up(1024);
constexpr int kVec = 4800;                  // cat1 (784) | z1 (16) | mu (16) | cat3 (32)
constexpr int kVz1 = kVec + kCat, kVmu = kVz1 + kLt, kVc3 = kVmu + kLt;
constexpr int kD0 = 5700;                   // fc3's output, the flattened (192, 4) tensor
constexpr int kFullLds = 12 * kP3u + 24 * kP2u;  // 19,008
static_assert(48 * kP2d + 96 * kP3d <= kVec && kVc3 + 2 * kLt <= kD0, "middle vectors behind the down chain's last images");
static_assert(96 * kP0u <= kVec && kD0 + 768 <= 12 * kP0d, "tconv0 writes its output beside the vectors it reads");
static_assert(96 * kP0u + 48 * kP1u <= 12 * kP3u && 12 * kP0d + 24 * kP1d <= kFullLds, "every phase fits");

__global__ __launch_bounds__(kChainThreads) void chain1d_full_fwd_kernel(const Chain1dFullArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x, pr = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int nstamp = 0;
  auto stamp = [&]() {
    if (a.stamps && b == 0 && pr == 0 && t == 0) a.stamps[nstamp] = (long long)__builtin_amdgcn_s_memtime();
    ++nstamp;
  };
  stamp();
  // ---- down chain (conv2 -> conv3 -> conv4), as conv1d_chain_down_kernel
  float* x0 = smem;
  float* y1 = x0 + 12 * kP0d;
  float* y2 = x0;
  float* y3 = y2 + 48 * kP2d;
  for (int c = t; c < 12; c += kChainThreads) x0[c * kP0d] = 0.f;
  for (int c = t; c < 24; c += kChainThreads) y1[c * kP1d] = 0.f;
  const float* in = a.in[pr] + (long)b * a.in_bs;
#pragma unroll 4
  for (int i = t; i < 12 * 1024; i += kChainThreads) {
    const int c = i >> 10, q = i & 1023;
    x0[c * kP0d + q + 1] = in[i];
  }
  const DactRegs<24, 256> r1{};
  const DactRegs<48, 64> r2{};
  const DactRegs<96, 16> r3{};
  __syncthreads();
  stamp();
  down_stage<12, 24, 1024, kP0d, kP1d>(x0, y1, a.dn[0].w[pr], a.dn[0].bias[pr], 1, 1);
  __syncthreads();
  stamp();
  copy_out<24, 256, kP1d>(y1, 1, a.dn[0].out[pr] + (long)b * a.dn[0].out_bs, nullptr, false, r1);
  for (int c = t; c < 48; c += kChainThreads) y2[c * kP2d] = 0.f;
  __syncthreads();
  stamp();
  down_stage<24, 48, 256, kP1d, kP2d>(y1, y2, a.dn[1].w[pr], a.dn[1].bias[pr], 1, 1);
  __syncthreads();
  stamp();
  copy_out<48, 64, kP2d>(y2, 1, a.dn[1].out[pr] + (long)b * a.dn[1].out_bs, nullptr, false, r2);
  for (int c = t; c < 96; c += kChainThreads) y3[c * kP3d] = 0.f;
  __syncthreads();
  stamp();
  down_stage<48, 96, 64, kP2d, kP3d>(y2, y3, a.dn[2].w[pr], a.dn[2].bias[pr], 1, 1);
  __syncthreads();
  stamp();
  copy_out<96, 16, kP3d>(y3, 1, a.dn[2].out[pr] + (long)b * a.dn[2].out_bs, nullptr, false, r3);
  // ---- conv5: 96 x 16 -> 192 x 4 = cat1[0..767]; cat1[768..783] = elu(fcuv1(uvh)) and cat3[16..31] = elu(fcuv3(uvh)) come from HBM
  float* cat1 = smem + kVec;
  float* z1 = smem + kVz1;
  float* mu = smem + kVmu;
  float* cat3 = smem + kVc3;
  float* gcat1 = a.cat1[pr] + (long)b * kCat;
  float* gcat3 = a.cat3[pr] + (long)b * (kLt + kHd1);
  if (t < kHd1) cat1[768 + t] = gcat1[768 + t];
  else if (t < 2 * kHd1) cat3[kLt + t - kHd1] = gcat3[kLt + t - kHd1];
  down_small<96, 192, kP3d>(y3, cat1, a.w5[pr], a.b5[pr], 1, 1);
  __syncthreads();
  stamp();
  if (t < 768) gcat1[t] = cat1[t];
  // ---- fc1 (784 -> 16): a wavefront per output; fc2in, fc2out (16 -> 16): sixteen lanes of wavefront w per output w
  {
    const float* wr = a.fc1w[pr] + (long)wave * kCat;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 13; ++j) {
      const int k = lane + 64 * j;
      if (k < kCat) acc = fmaf(cat1[k], wr[k], acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) {
      const float v = elu(acc + a.fc1b[pr][wave]);
      z1[wave] = v;
      a.z1[pr][(long)b * kLt + wave] = v;
    }
  }
  __syncthreads();
  stamp();
  {
    float acc = lane < kLt ? z1[lane] * a.fc2inw[pr][wave * kLt + lane] : 0.f;
    acc = wave_sum(acc);
    if (lane == 0) {
      const float v = elu(acc + a.fc2inb[pr][wave]);
      mu[wave] = v;
      a.mu[pr][(long)b * a.mu_ld + wave] = v;
    }
  }
  __syncthreads();
  stamp();
  {
    float acc = lane < kLt ? mu[lane] * a.fc2outw[pr][wave * kLt + lane] : 0.f;
    acc = wave_sum(acc);
    if (lane == 0) {
      const float v = elu(acc + a.fc2outb[pr][wave]);
      cat3[wave] = v;
      gcat3[wave] = v;
    }
  }
  __syncthreads();
  stamp();
  // ---- fc3 (32 -> 768, no activation): a thread per output
  float* d0 = smem + kD0;
  if (t < 768) {
    const f32x4* wr = reinterpret_cast<const f32x4*>(a.fc3w[pr] + (long)t * (kLt + kHd1));
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < (kLt + kHd1) / 4; ++q) {
      const f32x4 wv = wr[q];
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = fmaf(cat3[4 * q + e], wv[e], acc);
    }
    const float v = acc + a.fc3b[pr][t];
    d0[t] = v;
    a.d0[pr][(long)b * 768 + t] = v;
  }
  __syncthreads();
  stamp();
  // ---- tconv0: 192 x 4 -> 96 x 16 into the up chain's input image
  float* u3 = smem;                  // 12 x kP3u, written last
  float* u0 = u3;                    // 96 x kP0u
  float* u1 = u0 + 96 * kP0u;        // 48 x kP1u
  float* u2 = u3 + 12 * kP3u;        // 24 x kP2u
  up_small<192, 96, kP0u>(d0, u0, a.wt0[pr], a.bt0[pr], 1);
  __syncthreads();
  stamp();
  const DactRegs<96, 16> q0{};
  const DactRegs<48, 64> q1{};
  const DactRegs<24, 256> q2{};
  const DactRegs<12, 1024> q3{};
  copy_out<96, 16, kP0u>(u0, 0, a.t0[pr] + (long)b * 96 * 16, nullptr, false, q0);
  // ---- up chain (tconv1 -> tconv2 -> tconv3), as conv1d_chain_up_kernel
  up_stage<96, 48, 16, kP0u, kP1u>(u0, 0, u1, a.up[0].w[pr], a.up[0].bias[pr], 1);
  __syncthreads();
  stamp();
  copy_out<48, 64, kP1u>(u1, 0, a.up[0].out[pr] + (long)b * a.up[0].out_bs, nullptr, false, q1);
  up_stage<48, 24, 64, kP1u, kP2u>(u1, 0, u2, a.up[1].w[pr], a.up[1].bias[pr], 1);
  __syncthreads();
  stamp();
  copy_out<24, 256, kP2u>(u2, 0, a.up[1].out[pr] + (long)b * a.up[1].out_bs, nullptr, false, q2);
  up_stage<24, 12, 256, kP2u, kP3u>(u2, 0, u3, a.up[2].w[pr], a.up[2].bias[pr], 1);
  __syncthreads();
  stamp();
  copy_out<12, 1024, kP3u>(u3, 0, a.up[2].out[pr] + (long)b * a.up[2].out_bs, nullptr, false, q3);
  stamp();
}

}  // namespace

bool chain1d_full_supported(int L, int hd, int rica, const int* ch /* conv1's .. conv5's output channels */, int L1 /* conv1's output length */) {
  return L == kLt && hd == kHd1 && rica && ch[0] == 12 && ch[1] == 24 && ch[2] == 48 && ch[3] == 96 && ch[4] == 192 && L1 == 1024 &&
         device_lds_fits(sizeof(float) * kFullLds);
}

int chain1d_full_fwd(const Chain1dFullArgs& a, int B, int nproblems, hipStream_t s) {
  auto al16 = [](const void* q) { return q && (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  bool ok = B > 0 && (nproblems == 1 || nproblems == 2) && a.in_bs % 4 == 0;
  for (int g = 0; g < nproblems && ok; ++g) {
    ok = al16(a.in[g]) && al16(a.w5[g]) && a.b5[g] && al16(a.wt0[g]) && a.bt0[g] && al16(a.fc1w[g]) && a.fc1b[g] && al16(a.fc2inw[g]) &&
         a.fc2inb[g] && al16(a.fc2outw[g]) && a.fc2outb[g] && al16(a.fc3w[g]) && a.fc3b[g] && al16(a.cat1[g]) && a.z1[g] && a.mu[g] &&
         al16(a.cat3[g]) && al16(a.d0[g]) && al16(a.t0[g]);
    for (int i = 0; i < 3 && ok; ++i)
      ok = al16(a.dn[i].w[g]) && a.dn[i].bias[g] && al16(a.dn[i].out[g]) && al16(a.up[i].w[g]) && a.up[i].bias[g] && al16(a.up[i].out[g]) &&
           a.dn[i].out_bs % 4 == 0 && a.up[i].out_bs % 4 == 0;
  }
  if (!ok) { set_last_error("chain1d_full_fwd: null / unaligned pointer or stride"); return LSHM_ERR_ARG; }
  const size_t lds = sizeof(float) * kFullLds;
  auto kern = chain1d_full_fwd_kernel;
  int rc;
  if ((rc = kernel_budget_ok(reinterpret_cast<const void*>(kern), kChainThreads, lds, "chain1d (full forward)"))) return rc;
  if ((rc = raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds, "chain1d (full forward)"))) return rc;
  hipLaunchKernelGGL(kern, dim3(B, nproblems), dim3(kChainThreads), lds, s, a);
  return check_launch("chain1d_full_fwd");
}

}  // namespace lshm
