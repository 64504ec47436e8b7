// Device-side building blocks of the LDS-resident 1-D chains (chain1d.hip: three layers; chain1d_full.hip: the whole
// mid + deep section of AutoEncoder1DCNN): per-position GEMM stages on v_mfma_f32_16x16x4_f32 over LDS images, and the
// copy-out pass that writes a stage's output to HBM.
#pragma once
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

}  // namespace lshm
