// Direct (LDS-patch) kernels for the bandwidth-heavy outer layers of the 2-D autoencoder
// (src/lofar_models.py:31-33 conv0/conv1 and :56-57 tconv4/tconv5, plus the data gradients with
// the same geometry).  The generic implicit GEMM re-gathers every input element once per tap and,
// for the transposed conv, writes each output parity with stride-2 scalar stores; here
//   * a workgroup stages the raw input patch of its tile in LDS once (coalesced, halo included),
//   * MFMA A-fragments are read straight out of that patch (no im2col copy),
//   * the transposed conv computes all four output parities at once: the GEMM N dimension is
//     (parity_y, parity_x, channel), so the 16-wide fp32 MFMA tile is full even for 4 output
//     channels, K runs over the 3x3 input neighbourhood (taps a parity does not use have zero weight),
//   * results go through an LDS output tile and leave as full-row float4 stores, with bias, ELU and
//     the ELU' multiply of the backward pass fused.
// Arithmetic is v_mfma_f32_16x16x4_f32 (exact fp32).
#include <stdlib.h>

#include "kernels.h"

namespace lshm {

// ----------------------------------------------------------------------------------------------
// transposed conv k4 s2 p1 forward == conv k4 s2 p1 data gradient (tconv4: 12 -> 8 channels).
//   big[b, co, 2m+py, 2n+px] = bias[co] + sum_{cs,dy,dx} small[b, cs, m+dy, n+dx] * w[cs, co, py-2dy+1, px-2dx+1]
//   (dy in {py-1, py}, dx in {px-1, px})
// One GEMM per output ROW parity py, the two column parities side by side in the 16-wide tile: M = 16 consecutive n,
// N = (px, co) = 2 CB = 16, K = (cs, dyi in {0,1}, dxp in {0,1,2}) = 6 CS with dy = py - 1 + dyi, dx = dxp - 1 (a column
// parity uses two of the three dxp: 2/3 of the products are useful; the first form of this kernel put all four parities
// into N = 32 over the whole 3 x 3 neighbourhood, K = 9 CS with 4/9 useful -- 3/2 of the matrix instructions and of the
// weight-fragment registers).  Tile: TH small rows x TW small columns; 4 wavefronts, each (TH*TW/16)/4 m-tiles.
// ----------------------------------------------------------------------------------------------
// bf16 storage (BASELINE configs[2]): the four taps of a kernel row meet the four values of a weight quad in ONE
// v_mfma_f32_4x4x4_16B_bf16 (a lane's operand = its four k) instead of four v_mfma_f32_4x4x1_f32; the activations come from bf16
// tensors (exact), or are rounded to bf16 on the way (the fp32 minibatch under conv0: the operand precision of that configuration),
// the weights are rounded once per launch.  fp32 storage: the instruction sequence of before.
typedef short q4_s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ q4_s16x4 q4_bits(const bf16x4& v) { return __builtin_bit_cast(q4_s16x4, v); }
__device__ __forceinline__ f32x4 q4_mma(float a0, float a1, float a2, float a3, const bf16x4& w, f32x4 acc) {
  return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(q4_bits(__builtin_convertvector((f32x4){a0, a1, a2, a3}, bf16x4)), q4_bits(w), acc, 0, 0, 0);
}
__device__ __forceinline__ f32x4 q16_mma(float a0, float a1, float a2, float a3, const bf16x4& w, f32x4 acc) {  // ... and v_mfma_f32_16x16x16_bf16 for four k-steps of a 16 x 16 tile
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(q4_bits(__builtin_convertvector((f32x4){a0, a1, a2, a3}, bf16x4)), q4_bits(w), acc, 0, 0, 0);
}

template <int CS, int CB, int TH, int TW, class TO = float>  // TO: element type of `big` and of `dact` (bf16 storage, common.h)
__global__ __launch_bounds__(256) void tconv2d_direct_kernel(const float* __restrict__ small, long s_bs,
                                                             const float* __restrict__ w,
                                                             const float* __restrict__ bias,
                                                             float* __restrict__ big_, long big_bs,
                                                             const float* __restrict__ dact_, int Hs, int Ws,
                                                             int act, int ntiles) {
  static_assert(2 * CB == 16 && (CS * 6) % 4 == 0, "(px, co) fills one 16-wide tile; whole k-steps");
  TO* __restrict__ big = reinterpret_cast<TO*>(big_);
  const TO* __restrict__ dact = reinterpret_cast<const TO*>(dact_);
  constexpr int KS = CS * 6 / 4;                    // k-steps of 4
  constexpr int PH = TH + 2, PW = TW + 2;           // input patch with halo
  constexpr int MT = TH * TW / 16;                  // m-tiles per workgroup
  constexpr int MW = MT / 4;                        // m-tiles per wave
  constexpr int OW = 2 * TW;                        // output tile width
  static_assert(MT % 4 == 0, "tile must give every wave the same number of m-tiles");
  __shared__ float patch[CS * PH * PW];
  __shared__ __attribute__((aligned(16))) float otile[CB * 2 * TH * OW];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lm = lane & 15, lk = lane >> 4;
  const int px = lm >> 3, co = lm & 7;
  // ---- B fragments: lane (kk = lk, n = lm = (px, co)) of step s, row parity py: ky = 3 - py - 2 dyi, kx = px - 2 dxp + 3
  float bf[KS][2];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int k = 4 * s + lk;
    const int cs = k / 6, r = k - cs * 6;
    const int dyi = r / 3, dxp = r - dyi * 3;
    const int kx = px - 2 * dxp + 3;
#pragma unroll
    for (int py = 0; py < 2; ++py) bf[s][py] = (kx >= 0 && kx <= 3) ? w[(((long)cs * CB + co) * 4 + 3 - py - 2 * dyi) * 4 + kx] : 0.f;
  }
  // bf16 storage of the output: four k-steps per v_mfma_f32_16x16x16_bf16 (the last group padded with zeros)
  constexpr bool BF16_MMA = sizeof(TO) == 2;
  constexpr int KG = (KS + 3) / 4;
  bf16x4 bfp[KG][2];
#pragma unroll
  for (int g = 0; g < KG; ++g)
#pragma unroll
    for (int py = 0; py < 2; ++py) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = 4 * g + e < KS ? bf[4 * g + e < KS ? 4 * g + e : 0][py] : 0.f;
      bfp[g][py] = __builtin_convertvector(v, bf16x4);
    }
  const float bv = bias ? bias[co] : 0.f;
  // persistent over tiles: the weight fragments above are loaded once per workgroup; the next
  // tile's patch is fetched into registers while the current tile computes (software pipeline)
  const int tiles_x = Ws / TW, tiles_y = Hs / TH;
  constexpr int NV4 = (CS * PH * (TW / 4) + 255) / 256, NH = (CS * PH * 2 + 255) / 256;
  f32x4 rv[NV4];
  float rh[NH];
  auto load_tile = [&](int tile) {
    const int b = tile / (tiles_x * tiles_y);
    const int tr_ = tile - b * (tiles_x * tiles_y);
    const int m0 = (tr_ / tiles_x) * TH, n0 = (tr_ % tiles_x) * TW;
    const float* sb = small + (long)b * s_bs;
#pragma unroll
    for (int q = 0; q < NV4; ++q) {
      const int i = q * 256 + t;
      const int rowi = i / (TW / 4), c4 = i - rowi * (TW / 4);
      const int cs = rowi / PH, py = rowi - cs * PH;
      const int iy = m0 + py - 1;
      rv[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (i < CS * PH * (TW / 4) && (unsigned)iy < (unsigned)Hs)
        rv[q] = *reinterpret_cast<const f32x4*>(sb + ((long)cs * Hs + iy) * Ws + n0 + 4 * c4);
    }
#pragma unroll
    for (int q = 0; q < NH; ++q) {
      const int i = q * 256 + t;
      const int rowi = i >> 1, side = i & 1;
      const int cs = rowi / PH, py = rowi - cs * PH;
      const int iy = m0 + py - 1, ix = side ? n0 + TW : n0 - 1;
      rh[q] = 0.f;
      if (i < CS * PH * 2 && (unsigned)iy < (unsigned)Hs && (unsigned)ix < (unsigned)Ws)
        rh[q] = sb[((long)cs * Hs + iy) * Ws + ix];
    }
  };
  if ((int)blockIdx.x < ntiles) load_tile(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
  const int b = tile / (tiles_x * tiles_y);
  const int tr_ = tile - b * (tiles_x * tiles_y);
  const int m0 = (tr_ / tiles_x) * TH, n0 = (tr_ % tiles_x) * TW;
  __syncthreads();  // previous tile's output pass is done with the LDS buffers
#pragma unroll
  for (int q = 0; q < NV4; ++q) {
    const int i = q * 256 + t;
    if (i < CS * PH * (TW / 4)) {
      const int rowi = i / (TW / 4), c4 = i - rowi * (TW / 4);
      float* d = &patch[rowi * PW + 1 + 4 * c4];
      d[0] = rv[q][0]; d[1] = rv[q][1]; d[2] = rv[q][2]; d[3] = rv[q][3];
    }
  }
#pragma unroll
  for (int q = 0; q < NH; ++q) {
    const int i = q * 256 + t;
    if (i < CS * PH * 2) patch[(i >> 1) * PW + ((i & 1) ? PW - 1 : 0)] = rh[q];
  }
  __syncthreads();
  if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);

  f32x4 acc[MW][2];  // [m-tile][py]
#pragma unroll
  for (int i = 0; i < MW; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // m-tile index -> (row in tile, first column in tile)
  constexpr int TPR = TW / 16;  // m-tiles per tile row
  if constexpr (BF16_MMA) {
#pragma unroll
    for (int g = 0; g < KG; ++g) {
      int koff[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = 4 * (4 * g + e < KS ? 4 * g + e : KS - 1) + lk;
        const int cs = k / 6, r = k - cs * 6;
        const int dyi = r / 3, dxp = r - dyi * 3;
        koff[e] = (cs * PH + dyi) * PW + dxp;
      }
#pragma unroll
      for (int i = 0; i < MW; ++i) {
        const int mt = wave * MW + i;
        const int row = mt / TPR, col = (mt - row * TPR) * 16;
        const float* ap = &patch[row * PW + col + lm];
#pragma unroll
        for (int py = 0; py < 2; ++py)  // (a step past the last one meets a zero weight)
          acc[i][py] = q16_mma(ap[koff[0] + py * PW], ap[koff[1] + py * PW], ap[koff[2] + py * PW], ap[koff[3] + py * PW], bfp[g][py], acc[i][py]);
      }
    }
  } else {
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int k = 4 * s + lk;
    const int cs = k / 6, r = k - cs * 6;
    const int dyi = r / 3, dxp = r - dyi * 3;
    const int koff = (cs * PH + dyi) * PW + dxp;  // patch row = row + py + dyi, column = col + dxp
#pragma unroll
    for (int i = 0; i < MW; ++i) {
      const int mt = wave * MW + i;
      const int row = mt / TPR, col = (mt - row * TPR) * 16;
      const float* ap = &patch[koff + row * PW + col + lm];
#pragma unroll
      for (int py = 0; py < 2; ++py)
        acc[i][py] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[py * PW], bf[s][py], acc[i][py], 0, 0, 0);
    }
  }
  }
  // ---- accumulators -> LDS output tile [co][2*TH][2*TW] (bias + activation applied here)
#pragma unroll
  for (int i = 0; i < MW; ++i) {
    const int mt = wave * MW + i;
    const int row = mt / TPR, col = (mt - row * TPR) * 16 + 4 * lk;  // 4 consecutive small columns
#pragma unroll
    for (int py = 0; py < 2; ++py) {
      float* o = &otile[(co * 2 * TH + 2 * row + py) * OW + 2 * col + px];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = acc[i][py][r] + bv;
        o[2 * r] = act ? elu(v) : v;
      }
    }
  }
  __syncthreads();
  // ---- coalesced float4 stores (and the ELU' multiply for the backward use)
  const int Hb = 2 * Hs, Wb = 2 * Ws;
  TO* bb = big + (long)b * big_bs;
  const TO* db = dact ? dact + (long)b * big_bs : nullptr;
  for (int i = t; i < CB * 2 * TH * OW / 4; i += 256) {
    const int e = 4 * i;
    const int c = e / (2 * TH * OW), r = e - c * (2 * TH * OW);
    const int oy = r / OW, ox = r - oy * OW;
    const long g = ((long)c * Hb + 2 * m0 + oy) * Wb + 2 * n0 + ox;
    f32x4 v = *reinterpret_cast<const f32x4*>(&otile[e]);
    if (db) {
      const f32x4 sv = Elem<TO>::ld4(db + g);
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] *= elu_grad_from_out(sv[q]);
    }
    Elem<TO>::st4(bb + g, v);
  }
  }  // tile loop
}

// ----------------------------------------------------------------------------------------------
// The same transposed conv for tconv5 (8 -> 4 channels; as a data gradient: conv0) on
// v_mfma_f32_4x4x1_16b_f32, one output parity at a time with only the four taps that parity uses:
// no structurally-zero taps (the all-parity 16x16 formulation spends 9/4 of the useful MACs) and no
// padded channel tile.   lane l <-> small column n = l of one small row;  A = small[cs][m+dy][n+dx]
// from the LDS patch (one ds_read_b32, shared by every parity that uses that neighbour);  B =
// w[cs][q][ky][kx] for q = l%4;  D[py][px] = 4 consecutive n of channel q.  The two px parities of
// a lane interleave into 8 consecutive output columns -> two float4 stores per output row.
// Wavefront (rp, ch) takes small rows {2rp, 2rp+1} and input channels 4ch..4ch+3 (64 weight
// registers, resident); the two channel halves of a row meet in LDS and are added half 0 + half 1.
// ----------------------------------------------------------------------------------------------
template <int TH, class TO = float, class TS = float>  // TO: element type of `big` and `dact`, TS: of `small` (bf16 storage, common.h)
__global__ __launch_bounds__(256, 3) void tconv2d_q4_kernel(const float* __restrict__ small_, long s_bs,
                                                            const float* __restrict__ w,
                                                            const float* __restrict__ bias,
                                                            float* __restrict__ big_, long big_bs,
                                                            const float* __restrict__ dact_, int Hs, int Ws,
                                                            int act, int ntiles) {
  TO* __restrict__ big = reinterpret_cast<TO*>(big_);
  const TO* __restrict__ dact = reinterpret_cast<const TO*>(dact_);
  const TS* __restrict__ small = reinterpret_cast<const TS*>(small_);
  constexpr int CS = 8, CB = 4, TW = 64;
  constexpr int PH = TH + 2, PW = TW + 2;
  static_assert(TH == 4, "two small rows per wavefront pair");
  __shared__ float patch[CS * PH * PW];
  __shared__ f32x4 red[4][2][2][64];  // [wave][py][px][lane]: the row this wavefront does not finish
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int q = lane & 3, rp = wave & 1, ch = wave >> 1;
  // B fragments: bw[c][ky] = taps kx 0..3 of w[4ch + c][q][ky][:]
  f32x4 bw[4][4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int ky = 0; ky < 4; ++ky)
      bw[c][ky] = *reinterpret_cast<const f32x4*>(w + (((long)(4 * ch + c) * CB + q) * 4 + ky) * 4);
  // bf16 storage of the output: the four taps of an output parity in one v_mfma_f32_4x4x4_16B_bf16 (bwp[c][py][px] = the quad
  // the loop below meets in its (dyi, dxi) order)
  constexpr bool BF16_MMA = sizeof(TO) == 2;
  bf16x4 bwp[4][2][2];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
      for (int px = 0; px < 2; ++px) {
        f32x4 v;
#pragma unroll
        for (int dyi = 0; dyi < 2; ++dyi)
#pragma unroll
          for (int dxi = 0; dxi < 2; ++dxi) {
            const int dy = py - 1 + dyi, dx = px - 1 + dxi;
            v[2 * dyi + dxi] = bw[c][py - 2 * dy + 1][px - 2 * dx + 1];
          }
        bwp[c][py][px] = __builtin_convertvector(v, bf16x4);
      }
  const float bv = bias ? bias[q] : 0.f;

  const int tiles_x = Ws / TW, tiles_y = Hs / TH;
  constexpr int NV4 = (CS * PH * (TW / 4) + 255) / 256, NHL = (CS * PH * 2 + 255) / 256;
  f32x4 rv[NV4];
  float rh[NHL];
  auto load_tile = [&](int tile) {
    const int b = tile / (tiles_x * tiles_y);
    const int tr_ = tile - b * (tiles_x * tiles_y);
    const int m0 = (tr_ / tiles_x) * TH, n0 = (tr_ % tiles_x) * TW;
    const TS* sb = small + (long)b * s_bs;
#pragma unroll
    for (int k = 0; k < NV4; ++k) {
      const int i = k * 256 + t;
      const int rowi = i / (TW / 4), c4 = i - rowi * (TW / 4);
      const int cs = rowi / PH, py = rowi - cs * PH;
      const int iy = m0 + py - 1;
      rv[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (i < CS * PH * (TW / 4) && (unsigned)iy < (unsigned)Hs)
        rv[k] = Elem<TS>::ld4(sb + ((long)cs * Hs + iy) * Ws + n0 + 4 * c4);
    }
#pragma unroll
    for (int k = 0; k < NHL; ++k) {
      const int i = k * 256 + t;
      const int rowi = i >> 1, side = i & 1;
      const int cs = rowi / PH, py = rowi - cs * PH;
      const int iy = m0 + py - 1, ix = side ? n0 + TW : n0 - 1;
      rh[k] = 0.f;
      if (i < CS * PH * 2 && (unsigned)iy < (unsigned)Hs && (unsigned)ix < (unsigned)Ws)
        rh[k] = Elem<TS>::ld(sb + ((long)cs * Hs + iy) * Ws + ix);
    }
  };
  if ((int)blockIdx.x < ntiles) load_tile(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int b = tile / (tiles_x * tiles_y);
    const int tr_ = tile - b * (tiles_x * tiles_y);
    const int m0 = (tr_ / tiles_x) * TH, n0 = (tr_ % tiles_x) * TW;
    __syncthreads();  // the previous tile's reads of patch / red are done
#pragma unroll
    for (int k = 0; k < NV4; ++k) {
      const int i = k * 256 + t;
      if (i < CS * PH * (TW / 4)) {
        const int rowi = i / (TW / 4), c4 = i - rowi * (TW / 4);
        float* d = &patch[rowi * PW + 1 + 4 * c4];
        d[0] = rv[k][0]; d[1] = rv[k][1]; d[2] = rv[k][2]; d[3] = rv[k][3];
      }
    }
#pragma unroll
    for (int k = 0; k < NHL; ++k) {
      const int i = k * 256 + t;
      if (i < CS * PH * 2) patch[(i >> 1) * PW + ((i & 1) ? PW - 1 : 0)] = rh[k];
    }
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);

    // acc[rr][py][px]: small row 2rp + rr, output parity (py, px)
    f32x4 acc[2][2][2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
      for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int px = 0; px < 2; ++px) acc[rr][py][px] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      // neighbourhood of both rows: patch rows 2rp .. 2rp+3 (= small rows 2rp-1 .. 2rp+2), columns n-1, n, n+1
      float a[4][3];
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) {
        const float* prow = &patch[((4 * ch + c) * PH + 2 * rp + pr) * PW + lane];
        a[pr][0] = prow[0]; a[pr][1] = prow[1]; a[pr][2] = prow[2];
      }
#pragma unroll
      for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
          for (int px = 0; px < 2; ++px) {
            if constexpr (BF16_MMA) {  // (dy, dx) = (py - 1, px - 1), (py - 1, px), (py, px - 1), (py, px)
              acc[rr][py][px] = q4_mma(a[rr + py][px], a[rr + py][px + 1], a[rr + py + 1][px], a[rr + py + 1][px + 1], bwp[c][py][px],
                                       acc[rr][py][px]);
            } else {
#pragma unroll
              for (int dyi = 0; dyi < 2; ++dyi)
#pragma unroll
                for (int dxi = 0; dxi < 2; ++dxi) {
                  const int dy = py - 1 + dyi, dx = px - 1 + dxi;  // dy in {py-1, py}, dx in {px-1, px}
                  const int ky = py - 2 * dy + 1, kx = px - 2 * dx + 1;
                  acc[rr][py][px] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[rr + dy + 1][dx + 1], bw[c][ky][kx],
                                                                      acc[rr][py][px], 0, 0, 0);
                }
            }
          }
    }
    // ---- exchange: this wavefront finishes small row 2rp + ch, its partner (other ch) the other one
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
      for (int px = 0; px < 2; ++px) red[wave][py][px][lane] = ch == 0 ? acc[1][py][px] : acc[0][py][px];
    __syncthreads();
    const int partner = wave ^ 2;
    const int m = m0 + 2 * rp + ch;
    const int Hb = 2 * Hs, Wb = 2 * Ws;
#pragma unroll
    for (int py = 0; py < 2; ++py) {
      f32x4 s0 = ch == 0 ? acc[0][py][0] : red[partner][py][0][lane];  // channel half 0
      f32x4 s1 = ch == 0 ? acc[0][py][1] : red[partner][py][1][lane];
      const f32x4 h0 = ch == 0 ? red[partner][py][0][lane] : acc[1][py][0];  // channel half 1
      const f32x4 h1 = ch == 0 ? red[partner][py][1][lane] : acc[1][py][1];
      s0 += h0;
      s1 += h1;
      // lane (slot, q): output row 2m+py, columns 2*(n0 + 4 slot) .. +7 of channel q: o[2r + px] = D[py][px][r]
      const long g = (long)b * big_bs + ((long)q * Hb + 2 * m + py) * Wb + 2 * (n0 + 4 * (lane >> 2));
      f32x4 o0 = {s0[0] + bv, s1[0] + bv, s0[1] + bv, s1[1] + bv};
      f32x4 o1 = {s0[2] + bv, s1[2] + bv, s0[3] + bv, s1[3] + bv};
      if (act) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { o0[r] = elu(o0[r]); o1[r] = elu(o1[r]); }
      }
      if (dact) {
        const f32x4 v0 = Elem<TO>::ld4(dact + g);
        const f32x4 v1 = Elem<TO>::ld4(dact + g + 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) { o0[r] *= elu_grad_from_out(v0[r]); o1[r] *= elu_grad_from_out(v1[r]); }
      }
      Elem<TO>::st4(big + g, o0);
      Elem<TO>::st4(big + g + 4, o1);
    }
  }
}

bool tconv2d_direct_supported(int Cs, int Cb, int Hs, int Ws) {
  if (Cs == 8 && Cb == 4) return Hs % 4 == 0 && Ws % 64 == 0;
  if (Cs == 12 && Cb == 8) return Hs % 8 == 0 && Ws % 32 == 0;
  return false;
}

int tconv2d_direct(const float* small, long s_bs, const float* w, const float* bias, float* big,
                   long big_bs, const float* dact, int B, int Cs, int Cb, int Hs, int Ws, int act,
                   hipStream_t st, int big_bf16, int small_bf16) {
  // bf16 storage: `big` (and `dact`, which has its shape) for both outer layers, `small` for the outermost one
  if (small_bf16 && !(Cs == 8 && Cb == 4 && big_bf16)) { set_last_error("tconv2d_direct: a bf16 `small` needs the outermost layer with bf16 `big`"); return LSHM_ERR_UNSUPPORTED; }
  if (Cs == 8 && Cb == 4) {
    const int ntiles = (Ws / 64) * (Hs / 4) * B;
    const dim3 grid(ntiles < 1024 ? ntiles : 1024);
    if (big_bf16 && small_bf16)
      hipLaunchKernelGGL((tconv2d_q4_kernel<4, bf16, bf16>), grid, dim3(256), 0, st, small, s_bs, w, bias, big, big_bs, dact, Hs, Ws, act, ntiles);
    else if (big_bf16)
      hipLaunchKernelGGL((tconv2d_q4_kernel<4, bf16>), grid, dim3(256), 0, st, small, s_bs, w, bias, big, big_bs, dact, Hs, Ws, act, ntiles);
    else
      hipLaunchKernelGGL((tconv2d_q4_kernel<4>), grid, dim3(256), 0, st, small, s_bs, w, bias, big, big_bs, dact, Hs, Ws, act, ntiles);
  } else if (Cs == 12 && Cb == 8) {
    constexpr int cap = 512;  // persistent workgroups (768 / 1024: 25.1 / 27.5 us against 25.6, profiles/r03)
    if (Hs % 4) {  // (8-row tiles only where the height asks for them: measured equal at best, profiles/r03/README.md)
      const int ntiles = (Ws / 32) * (Hs / 8) * B;
      const dim3 grid(ntiles < cap ? ntiles : cap);
      if (big_bf16)
        hipLaunchKernelGGL((tconv2d_direct_kernel<12, 8, 8, 32, bf16>), grid, dim3(256), 0, st, small, s_bs, w, bias, big, big_bs, dact, Hs, Ws, act, ntiles);
      else
        hipLaunchKernelGGL((tconv2d_direct_kernel<12, 8, 8, 32>), grid, dim3(256), 0, st, small, s_bs, w, bias, big, big_bs, dact, Hs, Ws, act, ntiles);
    } else {
      const int ntiles = (Ws / 32) * (Hs / 4) * B;
      const dim3 grid(ntiles < cap ? ntiles : cap);
      if (big_bf16)
        hipLaunchKernelGGL((tconv2d_direct_kernel<12, 8, 4, 32, bf16>), grid, dim3(256), 0, st, small, s_bs, w, bias, big, big_bs, dact, Hs, Ws, act, ntiles);
      else
        hipLaunchKernelGGL((tconv2d_direct_kernel<12, 8, 4, 32>), grid, dim3(256), 0, st, small, s_bs, w, bias, big, big_bs, dact, Hs, Ws, act, ntiles);
    }
  } else {
    set_last_error("tconv2d_direct: unsupported shape");
    return LSHM_ERR_UNSUPPORTED;
  }
  return check_launch("tconv2d_direct");
}

}  // namespace lshm

namespace lshm {

// ----------------------------------------------------------------------------------------------
// conv k4 s2 p1 weight gradient for the outer layers (also the transposed conv's, with the roles
// of the tensors swapped):   dW[cs, cb, ky, kx] = sum_{b,oy,ox} small[b,cs,oy,ox] big[b,cb,2oy-1+ky,2ox-1+kx]
// GEMM view: M = Cs (<= 16), N = Cb*16 (one 16-wide MFMA tile per big channel), K = all positions.
// A workgroup walks tiles of TH x TW small positions: the small tile and the matching big patch
// are staged once in LDS, each wavefront takes every 4th group of 4 positions (K split over the
// waves), fragments are read straight from the patch.  Accumulators stay in registers across all
// tiles of the (persistent) workgroup; one slab per workgroup is combined by reduce_partials.
// ----------------------------------------------------------------------------------------------
template <int CS, int CB, int TH, int TW, class TB = float, class TS = float>  // TB / TS: element types of `big` / `small`
__global__ __launch_bounds__(256) void conv2d_wgrad_direct_kernel(const float* __restrict__ small_, long s_bs,
                                                                  const float* __restrict__ big_, long big_bs,
                                                                  float* __restrict__ partial, int Hs, int Ws,
                                                                  int ntiles, int bias_from) {
  const TB* __restrict__ big = reinterpret_cast<const TB*>(big_);
  const TS* __restrict__ small = reinterpret_cast<const TS*>(small_);
  // bias_from: 0 none, 1 bias gradient = sum of `small` (conv layer), 2 = sum of `big` (transposed conv);
  // every element passes through this thread's registers on its way to LDS, and a thread always stages
  // the same channel, so the sums cost one add per float4
  constexpr int MT = (CS + 15) / 16;      // m-tiles: 24 small channels (conv2 / tconv3) take two
  constexpr int BPAD = ((CS > CB ? CS : CB) + 15) / 16 * 16;  // bias slots behind the weight slab
  constexpr int SLAB = CS * CB * 16 + BPAD;
  constexpr int TP = TH * TW;             // positions per tile
  constexpr int LDS_S = TP + 2;           // small-tile row stride: == 2 (mod 32) -> conflict-free A reads
  constexpr int PH = 2 * TH + 2, PW = 2 * TW + 2;
  constexpr int NT = CB;                  // one n-tile (16 taps) per big channel
  // one buffer: [small tile | big patch] while the tiles are walked, then the combine image(s)
  constexpr int CPITCH = CB * 16 + 4;  // combine row pitch: the 4 row groups of a lane quad land 16 banks apart
  constexpr int TILE_FLOATS = 16 * MT * LDS_S + CB * PH * PW;
  constexpr int COMB_FLOATS = MT > 1 ? 4 * CS * CPITCH : 0;
  __shared__ float smem_w[TILE_FLOATS > COMB_FLOATS ? TILE_FLOATS : COMB_FLOATS];
  float* stile = smem_w;
  float* patch = smem_w + 16 * MT * LDS_S;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lm = lane & 15, lk = lane >> 4;
  for (int i = t; i < 16 * MT * LDS_S; i += 256) stile[i] = 0.f;  // rows >= CS stay zero

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int NQS = (CS * TH * (TW / 4) + 255) / 256;      // float4 of the small tile per thread
  constexpr int NQB = (CB * PH * (2 * TW / 4) + 255) / 256;  // float4 of the big patch per thread
  float bs_small[NQS], bs_big[NQB];
#pragma unroll
  for (int qq = 0; qq < NQS; ++qq) bs_small[qq] = 0.f;
#pragma unroll
  for (int qq = 0; qq < NQB; ++qq) bs_big[qq] = 0.f;

  const int tiles_x = Ws / TW, tiles_y = Hs / TH;
  const int Hb = 2 * Hs, Wb = 2 * Ws;
  const int ky = lm >> 2, kx = lm & 3;
  // Software pipeline over the tiles of this (persistent) workgroup: the next tile's global loads are issued
  // into registers before the MFMA loop of the current one and written to LDS after it, so the HBM latency
  // of a tile hides behind the matrix work of its predecessor.
  constexpr int NHL = (CB * PH * 2 + 255) / 256;  // halo-column scalars per thread
  f32x4 rs[NQS], rb[NQB];
  float rh[NHL];
  auto fetch = [&](int tile) {
    const int b = tile / (tiles_x * tiles_y);
    const int tr_ = tile - b * (tiles_x * tiles_y);
    const int m0 = (tr_ / tiles_x) * TH, n0 = (tr_ % tiles_x) * TW;
    const TS* sb = small + (long)b * s_bs;
    const TB* bb = big + (long)b * big_bs;
    // small tile: [cs][TH*TW] as float4 rows of TW
#pragma unroll
    for (int qq = 0; qq < NQS; ++qq) {
      const int i = t + 256 * qq;
      rs[qq] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (i < CS * TH * (TW / 4)) {
        const int c4 = i % (TW / 4), rr = i / (TW / 4);
        const int row = rr % TH, cs = rr / TH;
        rs[qq] = Elem<TS>::ld4(sb + ((long)cs * Hs + m0 + row) * Ws + n0 + 4 * c4);
      }
    }
    // big patch: rows 2*m0-1 .. 2*m0+2*TH, cols 2*n0-1 .. 2*n0+2*TW (zero outside the image)
#pragma unroll
    for (int qq = 0; qq < NQB; ++qq) {
      const int i = t + 256 * qq;
      rb[qq] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (i < CB * PH * (2 * TW / 4)) {
        const int c4 = i % (2 * TW / 4), rr = i / (2 * TW / 4);
        const int prow = rr % PH, cb = rr / PH;
        const int iy = 2 * m0 - 1 + prow;
        if ((unsigned)iy < (unsigned)Hb) rb[qq] = Elem<TB>::ld4(bb + ((long)cb * Hb + iy) * Wb + 2 * n0 + 4 * c4);
      }
    }
#pragma unroll
    for (int qq = 0; qq < NHL; ++qq) {
      const int i = t + 256 * qq;
      rh[qq] = 0.f;
      if (i < CB * PH * 2) {
        const int side = i & 1, rr = i >> 1;
        const int prow = rr % PH, cb = rr / PH;
        const int iy = 2 * m0 - 1 + prow, ix = side ? 2 * n0 + 2 * TW : 2 * n0 - 1;
        if ((unsigned)iy < (unsigned)Hb && (unsigned)ix < (unsigned)Wb) rh[qq] = Elem<TB>::ld(bb + ((long)cb * Hb + iy) * Wb + ix);
      }
    }
  };
  if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    __syncthreads();  // the previous tile's MFMA loop has finished reading LDS
#pragma unroll
    for (int qq = 0; qq < NQS; ++qq) {
      const int i = t + 256 * qq;
      if (i < CS * TH * (TW / 4)) {
        const int c4 = i % (TW / 4), rr = i / (TW / 4);
        const int row = rr % TH, cs = rr / TH;
        const f32x4 v = rs[qq];
        float* d = &stile[cs * LDS_S + row * TW + 4 * c4];
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
        bs_small[qq] += (v[0] + v[1]) + (v[2] + v[3]);
      }
    }
#pragma unroll
    for (int qq = 0; qq < NQB; ++qq) {
      const int i = t + 256 * qq;
      if (i < CB * PH * (2 * TW / 4)) {
        const int c4 = i % (2 * TW / 4), rr = i / (2 * TW / 4);
        const int prow = rr % PH, cb = rr / PH;
        const f32x4 v = rb[qq];
        float* d = &patch[(cb * PH + prow) * PW + 1 + 4 * c4];
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
        // halo rows belong to the neighbouring tiles: only the 2*TH interior rows count towards the bias sum
        if (prow >= 1 && prow <= 2 * TH) bs_big[qq] += (v[0] + v[1]) + (v[2] + v[3]);
      }
    }
#pragma unroll
    for (int qq = 0; qq < NHL; ++qq) {
      const int i = t + 256 * qq;
      if (i < CB * PH * 2) {
        const int side = i & 1, rr = i >> 1;
        const int prow = rr % PH, cb = rr / PH;
        patch[(cb * PH + prow) * PW + (side ? PW - 1 : 0)] = rh[qq];
      }
    }
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);
    // K loop: groups of 4 consecutive positions, every 4th group per wave
#pragma unroll 4
    for (int it = 0; it < TP / 16; ++it) {  // (a constant trip count: `s = wave; s < TP / 4; s += 4` is refused by the unroller)
      const int s = wave + 4 * it;
      const int p = 4 * s + lk;
      const int oy = p / TW, ox = p - oy * TW;
      float a[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a[mt] = stile[(16 * mt + lm) * LDS_S + p];
      const int boff = (2 * oy + ky) * PW + 2 * ox + kx;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const float bv = patch[j * PH * PW + boff];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], bv, acc[mt][j], 0, 0, 0);
      }
    }
  }
  // ---- combine the 4 waves (fixed order) and write this workgroup's slab [CS][CB*16]
  __syncthreads();
  float* out = partial + (size_t)blockIdx.x * SLAB;
  if constexpr (MT == 1) {
    // the patch is dead now: reuse it as the 4 x 16 x (N+1) combine buffer
    static_assert(MT > 1 || CB * PH * PW >= 4 * 16 * (CB * 16 + 1), "combine buffer must fit in the patch");
    float (*comb)[16][CB * 16 + 1] = reinterpret_cast<float (*)[16][CB * 16 + 1]>(patch);
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) comb[wave][4 * lk + r][16 * j + lm] = acc[0][j][r];
    __syncthreads();
    for (int i = t; i < CS * CB * 16; i += 256) {
      const int m = i / (CB * 16), n = i - m * (CB * 16);
      out[i] = (comb[0][m][n] + comb[1][m][n]) + (comb[2][m][n] + comb[3][m][n]);
    }
  } else {
    // every wavefront leaves its partial image [CS][CB*16] (independent stores: a read-modify-write chain through
    // LDS costs ~150 cycles per element), then all threads add the four in wavefront order
    float* comb = smem_w + wave * CS * CPITCH;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * mt + 4 * lk + r;
          if (row < CS) comb[row * CPITCH + 16 * j + lm] = acc[mt][j][r];
        }
    __syncthreads();
    for (int i = t; i < CS * CB * 16; i += 256) {
      const int m = i / (CB * 16), n = i - m * (CB * 16);
      const float* c0 = smem_w + m * CPITCH + n;
      out[i] = (c0[0] + c0[CS * CPITCH]) + (c0[2 * CS * CPITCH] + c0[3 * CS * CPITCH]);
    }
  }
  // ---- bias partials: per-thread sums -> per-channel workgroup sums (fixed order), slab[CS*CB*16 + c]
  __syncthreads();
  float* bred = patch;  // [BPAD channels][4 waves]
  if (bias_from) {
    const int nch = bias_from == 1 ? CS : CB;
    for (int c = 0; c < nch; ++c) {
      float v = 0.f;
      if (bias_from == 1) {
#pragma unroll
        for (int qq = 0; qq < NQS; ++qq) {
          const int i = t + 256 * qq;
          if (i < CS * TH * (TW / 4) && (i / (TW / 4)) / TH == c) v += bs_small[qq];
        }
      } else {
#pragma unroll
        for (int qq = 0; qq < NQB; ++qq) {
          const int i = t + 256 * qq;
          if (i < CB * PH * (2 * TW / 4) && (i / (2 * TW / 4)) / PH == c) v += bs_big[qq];
        }
      }
      v = wave_sum(v);
      if (lane == 0) bred[c * 4 + wave] = v;
    }
  }
  __syncthreads();
  if (t < BPAD) {
    const int nch = bias_from == 1 ? CS : bias_from == 2 ? CB : 0;
    out[CS * CB * 16 + t] = t < nch ? (bred[t * 4] + bred[t * 4 + 1]) + (bred[t * 4 + 2] + bred[t * 4 + 3]) : 0.f;
  }
}

bool conv2d_wgrad_direct_supported(int Cs, int Cb, int Hs, int Ws) {
  if (Cs == 8 && Cb == 4) return Hs % 4 == 0 && Ws % 64 == 0;
  if (Cs == 12 && Cb == 8) return Hs % 8 == 0 && Ws % 32 == 0;
  if (Cs == 24 && Cb == 12) return Hs % 8 == 0 && Ws % 16 == 0;  // conv2 / tconv3: K = B*Hs*Ws >> M, N
  return false;
}
static int wgrad_bias_pad(int Cs, int Cb) { return ((Cs > Cb ? Cs : Cb) + 15) / 16 * 16; }
size_t conv2d_wgrad_direct_workspace_floats(int Cs, int Cb) {
  return (size_t)1024 * (Cs * Cb * 16 + wgrad_bias_pad(Cs, Cb));
}

int conv2d_wgrad_direct(const float* small, long s_bs, const float* big, long big_bs, float* dw, float* db,
                        int bias_from, int B, int Cs, int Cb, int Hs, int Ws, float* ws, size_t wsf, int accumulate,
                        hipStream_t st, GradJobs* defer, int big_bf16, int small_bf16) {
  if (!db) bias_from = 0;
  // bf16 storage: `big` for the two outer layer shapes, `small` for the outermost one (8 / 4 channels)
  if ((big_bf16 && Cs == 24) || (small_bf16 && !(Cs == 8 && Cb == 4))) { set_last_error("conv2d_wgrad_direct: bf16 storage only for the outer layers"); return LSHM_ERR_UNSUPPORTED; }
  if (wsf < conv2d_wgrad_direct_workspace_floats(Cs, Cb)) { set_last_error("conv2d_wgrad_direct: workspace too small"); return LSHM_ERR_WORKSPACE; }
  int grid;
#define LSHM_WG2D(CS_, CB_, TH_, TW_, TB_, TS_) \
  hipLaunchKernelGGL((conv2d_wgrad_direct_kernel<CS_, CB_, TH_, TW_, TB_, TS_>), dim3(grid), dim3(256), 0, st, small, s_bs, big, big_bs, ws, Hs, Ws, ntiles, bias_from)
  if (Cs == 8 && Cb == 4) {
    const int ntiles = (Ws / 64) * (Hs / 4) * B;
    grid = ntiles < 1024 ? ntiles : 1024;  // 112 VGPRs, 37 KB of LDS: four workgroups per CU; 4096 tiles at B = 256 -> 4 full rounds
    if (big_bf16 && small_bf16) LSHM_WG2D(8, 4, 4, 64, bf16, bf16);
    else if (big_bf16) LSHM_WG2D(8, 4, 4, 64, bf16, float);
    else if (small_bf16) LSHM_WG2D(8, 4, 4, 64, float, bf16);
    else LSHM_WG2D(8, 4, 4, 64, float, float);
  } else if (Cs == 12 && Cb == 8) {
    const int ntiles = (Ws / 32) * (Hs / 8) * B;
    grid = ntiles < 512 ? ntiles : 512;
    if (big_bf16) LSHM_WG2D(12, 8, 8, 32, bf16, float);
    else LSHM_WG2D(12, 8, 8, 32, float, float);
  } else if (Cs == 24 && Cb == 12) {
    const int ntiles = (Ws / 16) * (Hs / 8) * B;
    constexpr int cap2412 = 768;
    grid = ntiles < cap2412 ? ntiles : cap2412;
    LSHM_WG2D(24, 12, 8, 16, float, float);
  } else {
    set_last_error("conv2d_wgrad_direct: unsupported shape");
    return LSHM_ERR_UNSUPPORTED;
  }
#undef LSHM_WG2D
  int rc = check_launch("conv2d_wgrad_direct");
  if (rc) return rc;
  const int nw = Cs * Cb * 16, slab = nw + wgrad_bias_pad(Cs, Cb);
  const int nbias = bias_from == 1 ? Cs : Cb;
  if (defer) {
    defer->sums.push_back(SumJob{ws, dw, slab, nw, grid, 0, 0, 0, 0, accumulate, 0});
    if (bias_from) defer->sums.push_back(SumJob{ws + nw, db, slab, nbias, grid, 0, 0, 0, 0, accumulate, 0});
    return LSHM_OK;
  }
  rc = reduce_partials_strided(ws, slab, dw, nw, grid, accumulate, st);
  if (rc || !bias_from) return rc;
  return reduce_partials_strided(ws + nw, slab, db, nbias, grid, accumulate, st);
}

}  // namespace lshm

namespace lshm {

// ----------------------------------------------------------------------------------------------
// conv k4 s2 p1 forward for the outer layers (also the transposed conv's data gradient):
//   y[b,co,oy,ox] = bias[co] + sum_{ci,ky,kx} w[co,ci,ky,kx] x[b,ci,2oy-1+ky,2ox-1+kx]
// M = 16 consecutive ox per MFMA tile, N = Cout (<= 16), K steps = (ci, ky) with the 4 kx taps.
// The input patch of a TH x TW output tile sits in LDS once; A fragments are read at
// patch[ci][2oy+ky][2ox+kx] (bank = 2*lane + kx: conflict-free); weights live in registers.
// A PERSISTENT, software-pipelined workgroup over small tiles: with one 8 x 32 tile per workgroup and every workgroup
// of the launch resident at once (the first form of this kernel), all of them load, then all multiply, then all
// store -- an ablation (profiles/r03/README.md) shows the phases of conv1's forward simply adding up (6.5 fixed + 8
// loads + 8 matrix + 4.5 stores = 27 us for 46 MB).  Here a workgroup walks four 4 x 32 tiles (512 workgroups at
// B = 256); the float4 loads of tile i+1 are issued into registers before the matrix instructions of tile i and
// written to LDS after them, and the stores of tile i leave while tile i+1 computes: 26.8 -> 22.3 us (conv1 forward),
// 25.7 -> 22.3 us (tconv4's data gradient); 256 / 1024 / 2048 workgroups: 26.3 / 22.8 / 29.0 us.
// ----------------------------------------------------------------------------------------------
template <int CIN, int COUT, int TH, int TW, class TI = float>
__global__ __launch_bounds__(256, 4) void conv2d_direct_kernel(const float* __restrict__ x_, long x_bs,
                                                                 const float* __restrict__ w,
                                                                 const float* __restrict__ bias,
                                                                 float* __restrict__ y, long y_bs,
                                                                 const float* __restrict__ dact, int Ho, int Wo,
                                                                 int act, int ntiles) {
  const TI* __restrict__ x = reinterpret_cast<const TI*>(x_);
  constexpr int KS = CIN * 4;  // k-steps: (ci, ky), 4 kx taps each
  constexpr int PH = 2 * TH + 2, PW = 2 * TW + 2;
  constexpr int MT = TH * TW / 16, MW = MT / 4, TPR = TW / 16;
  static_assert(MT % 4 == 0 && COUT <= 16, "tile / channel limits");
  __shared__ float patch[CIN * PH * PW];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lm = lane & 15, lk = lane >> 4;
  float bf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) bf[s] = lm < COUT ? w[((long)lm * CIN * 4 + s) * 4 + lk] : 0.f;
  // bf16 storage of the input: the four kernel rows of an input channel in one v_mfma_f32_16x16x16_bf16
  constexpr bool BF16_MMA = sizeof(TI) == 2;
  bf16x4 bfp[CIN];
#pragma unroll
  for (int ci = 0; ci < CIN; ++ci) bfp[ci] = __builtin_convertvector((f32x4){bf[4 * ci], bf[4 * ci + 1], bf[4 * ci + 2], bf[4 * ci + 3]}, bf16x4);
  const float bv = (bias && lm < COUT) ? bias[lm] : 0.f;
  const int tiles_x = Wo / TW, tiles_y = Ho / TH;
  const int H = 2 * Ho, W = 2 * Wo;
  constexpr int NV4 = (CIN * PH * (2 * TW / 4) + 255) / 256, NH = (CIN * PH * 2 + 255) / 256;
  f32x4 rv[NV4];
  float rh[NH];
  auto load_tile = [&](int tile) {
    const int b = tile / (tiles_x * tiles_y);
    const int tr_ = tile - b * (tiles_x * tiles_y);
    const int m0 = (tr_ / tiles_x) * TH, n0 = (tr_ % tiles_x) * TW;
    const TI* xb = x + (long)b * x_bs;
#pragma unroll
    for (int q = 0; q < NV4; ++q) {
      const int i = q * 256 + t;
      const int c4 = i % (2 * TW / 4), rr = i / (2 * TW / 4);
      const int prow = rr % PH, ci = rr / PH;
      const int iy = 2 * m0 - 1 + prow;
      rv[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (i < CIN * PH * (2 * TW / 4) && (unsigned)iy < (unsigned)H) rv[q] = Elem<TI>::ld4(xb + ((long)ci * H + iy) * W + 2 * n0 + 4 * c4);
    }
#pragma unroll
    for (int q = 0; q < NH; ++q) {
      const int i = q * 256 + t;
      const int side = i & 1, rr = i >> 1;
      const int prow = rr % PH, ci = rr / PH;
      const int iy = 2 * m0 - 1 + prow, ix = side ? 2 * n0 + 2 * TW : 2 * n0 - 1;
      rh[q] = 0.f;
      if (i < CIN * PH * 2 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) rh[q] = Elem<TI>::ld(xb + ((long)ci * H + iy) * W + ix);
    }
  };
  if ((int)blockIdx.x < ntiles) load_tile(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int b = tile / (tiles_x * tiles_y);
    const int tr_ = tile - b * (tiles_x * tiles_y);
    const int m0 = (tr_ / tiles_x) * TH, n0 = (tr_ % tiles_x) * TW;
    __syncthreads();  // the previous tile's A fragments have been read
#pragma unroll
    for (int q = 0; q < NV4; ++q) {
      const int i = q * 256 + t;
      if (i < CIN * PH * (2 * TW / 4)) {
        const int c4 = i % (2 * TW / 4), rr = i / (2 * TW / 4);
        float* d = &patch[rr * PW + 1 + 4 * c4];
        d[0] = rv[q][0]; d[1] = rv[q][1]; d[2] = rv[q][2]; d[3] = rv[q][3];
      }
    }
#pragma unroll
    for (int q = 0; q < NH; ++q) {
      const int i = q * 256 + t;
      if (i < CIN * PH * 2) patch[(i >> 1) * PW + ((i & 1) ? PW - 1 : 0)] = rh[q];
    }
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);
    f32x4 acc[MW];
#pragma unroll
    for (int i = 0; i < MW; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (BF16_MMA) {
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int i = 0; i < MW; ++i) {
          const int mt = wave * MW + i;
          const int row = mt / TPR, col = (mt - row * TPR) * 16;
          const float* ap = &patch[(ci * PH + 2 * row) * PW + 2 * (col + lm) + lk];
          acc[i] = q16_mma(ap[0], ap[PW], ap[2 * PW], ap[3 * PW], bfp[ci], acc[i]);
        }
    } else {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int ci = s >> 2, ky = s & 3;
#pragma unroll
      for (int i = 0; i < MW; ++i) {
        const int mt = wave * MW + i;
        const int row = mt / TPR, col = (mt - row * TPR) * 16;
        const float a = patch[(ci * PH + 2 * row + ky) * PW + 2 * (col + lm) + lk];
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bf[s], acc[i], 0, 0, 0);
      }
    }
    }
    if (lm < COUT) {
#pragma unroll
      for (int i = 0; i < MW; ++i) {
        const int mt = wave * MW + i;
        const int row = mt / TPR, col = (mt - row * TPR) * 16 + 4 * lk;
        const long g = (long)b * y_bs + ((long)lm * Ho + m0 + row) * Wo + n0 + col;
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = acc[i][r] + bv;
          o[r] = act ? elu(v) : v;
        }
        if (dact) {
          const f32x4 sv = *reinterpret_cast<const f32x4*>(dact + g);
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] *= elu_grad_from_out(sv[r]);
        }
        *reinterpret_cast<f32x4*>(y + g) = o;
      }
    }
  }
}

// ----------------------------------------------------------------------------------------------
// Same op on v_mfma_f32_4x4x1_16b_f32 for conv0 (4 -> 8 channels; as a data gradient: tconv5).  The
// 4x4x1 form runs 16 independent 4x4 outer products per instruction, D[lane l][r] += A[lane 4*(l/4)+r]
// * B[lane l], at the same FLOP/clk as 16x16x4 - so 8 output channels are two full instructions instead
// of one half-empty 16-wide tile, and no lane idles in the ELU epilogue.
//   lane l <-> output column ox = l of one output row;  A = input value under tap (ci,ky,kx) at that
//   column, read straight from the LDS patch (two taps per ds_read_b64);  B = w[4h + l%4][ci][ky][kx];
//   D = 4 consecutive ox of channel 4h + l%4 -> float4 stores.
// Wavefront w takes input channel ci = w for all TH rows of the tile (its 16 taps x 2 channel groups =
// 32 weight registers stay resident for the whole launch); the four per-channel partial tiles meet in
// LDS and are added in channel order (fixed order: bitwise reproducible) by the wavefront that owns the row.
// ----------------------------------------------------------------------------------------------
template <int COUT, int TH, class TI = float, class TO = float>  // TI: element type of x, TO: of y and dact
__global__ __launch_bounds__(256, 3) void conv2d_q4_kernel(const float* __restrict__ x_, long x_bs,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y_,
                                                        long y_bs, const float* __restrict__ dact_, int Ho, int Wo,
                                                        int act, int ntiles) {
  const TI* __restrict__ x = reinterpret_cast<const TI*>(x_);
  TO* __restrict__ y = reinterpret_cast<TO*>(y_);
  const TO* __restrict__ dact = reinterpret_cast<const TO*>(dact_);
  constexpr int CIN = 4, TW = 64;
  constexpr int NH = COUT / 4;
  constexpr int PH = 2 * TH + 2, PW = 2 * TW + 2;  // even row stride keeps the ds_read_b64 at column 2*ox aligned
  static_assert(COUT % 4 == 0 && TH == 4, "one output row per wavefront in the combine");
  __shared__ __attribute__((aligned(16))) float patch[CIN * PH * PW];
  __shared__ f32x4 red[CIN][TH - 1][NH][64];  // partial rows of the other wavefronts (the own row stays in registers)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int q = lane & 3;
  // B fragments of this wavefront's input channel: bw[ky][h] = taps kx 0..3 of w[4h+q][wave][ky][:]
  f32x4 bw[4][NH];
#pragma unroll
  for (int ky = 0; ky < 4; ++ky)
#pragma unroll
    for (int h = 0; h < NH; ++h)
      bw[ky][h] = *reinterpret_cast<const f32x4*>(w + (((long)(4 * h + q) * CIN + wave) * 4 + ky) * 4);
  constexpr bool BF16_MMA = sizeof(TO) == 2;  // bf16 storage of the output: the bf16 configuration
  bf16x4 bwh[4][NH];
#pragma unroll
  for (int ky = 0; ky < 4; ++ky)
#pragma unroll
    for (int h = 0; h < NH; ++h) bwh[ky][h] = __builtin_convertvector(bw[ky][h], bf16x4);
  float bv[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) bv[h] = bias ? bias[4 * h + q] : 0.f;

  const int tiles_x = Wo / TW, tiles_y = Ho / TH;
  const int H = 2 * Ho, W = 2 * Wo;
  // software pipeline: the next tile's patch is fetched into registers while this one computes
  constexpr int NV4 = (CIN * PH * (2 * TW / 4) + 255) / 256, NHL = (CIN * PH * 2 + 255) / 256;
  f32x4 rv[NV4];
  float rh[NHL];
  auto load_tile = [&](int tile) {
    const int b = tile / (tiles_x * tiles_y);
    const int tr_ = tile - b * (tiles_x * tiles_y);
    const int m0 = (tr_ / tiles_x) * TH, n0 = (tr_ % tiles_x) * TW;
    const TI* xb = x + (long)b * x_bs;
#pragma unroll
    for (int k = 0; k < NV4; ++k) {
      const int i = k * 256 + t;
      const int c4 = i % (2 * TW / 4), rr = i / (2 * TW / 4);
      const int prow = rr % PH, ci = rr / PH;
      const int iy = 2 * m0 - 1 + prow;
      rv[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (i < CIN * PH * (2 * TW / 4) && (unsigned)iy < (unsigned)H)
        rv[k] = Elem<TI>::ld4(xb + ((long)ci * H + iy) * W + 2 * n0 + 4 * c4);
    }
#pragma unroll
    for (int k = 0; k < NHL; ++k) {
      const int i = k * 256 + t;
      const int side = i & 1, rr = i >> 1;
      const int prow = rr % PH, ci = rr / PH;
      const int iy = 2 * m0 - 1 + prow, ix = side ? 2 * n0 + 2 * TW : 2 * n0 - 1;
      rh[k] = 0.f;
      if (i < CIN * PH * 2 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) rh[k] = Elem<TI>::ld(xb + ((long)ci * H + iy) * W + ix);
    }
  };
  if ((int)blockIdx.x < ntiles) load_tile(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int b = tile / (tiles_x * tiles_y);
    const int tr_ = tile - b * (tiles_x * tiles_y);
    const int m0 = (tr_ / tiles_x) * TH, n0 = (tr_ % tiles_x) * TW;
    __syncthreads();  // the previous tile's A reads and combine reads are done
#pragma unroll
    for (int k = 0; k < NV4; ++k) {
      const int i = k * 256 + t;
      if (i < CIN * PH * (2 * TW / 4)) {
        const int c4 = i % (2 * TW / 4), rr = i / (2 * TW / 4);
        float* d = &patch[rr * PW + 1 + 4 * c4];  // odd offset: b32 + b64 + b32
        d[0] = rv[k][0];
        *reinterpret_cast<float2*>(d + 1) = make_float2(rv[k][1], rv[k][2]);
        d[3] = rv[k][3];
      }
    }
#pragma unroll
    for (int k = 0; k < NHL; ++k) {
      const int i = k * 256 + t;
      if (i < CIN * PH * 2) patch[(i >> 1) * PW + ((i & 1) ? PW - 1 : 0)] = rh[k];
    }
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);

    // ---- this wavefront's input channel, all rows: TH*NH independent accumulator chains
    f32x4 acc[TH][NH];
#pragma unroll
    for (int r = 0; r < TH; ++r)
#pragma unroll
      for (int h = 0; h < NH; ++h) acc[r][h] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
      float2 a01[TH], a23[TH];
#pragma unroll
      for (int r = 0; r < TH; ++r) {
        const float* pr = &patch[(wave * PH + 2 * r + ky) * PW + 2 * lane];
        a01[r] = *reinterpret_cast<const float2*>(pr);
        a23[r] = *reinterpret_cast<const float2*>(pr + 2);
      }
      if constexpr (BF16_MMA) {
#pragma unroll
        for (int r = 0; r < TH; ++r)
#pragma unroll
          for (int h = 0; h < NH; ++h) acc[r][h] = q4_mma(a01[r].x, a01[r].y, a23[r].x, a23[r].y, bwh[ky][h], acc[r][h]);
      } else {
#pragma unroll
      for (int r = 0; r < TH; ++r)
#pragma unroll
        for (int h = 0; h < NH; ++h) acc[r][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(a01[r].x, bw[ky][h][0], acc[r][h], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < TH; ++r)
#pragma unroll
        for (int h = 0; h < NH; ++h) acc[r][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(a01[r].y, bw[ky][h][1], acc[r][h], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < TH; ++r)
#pragma unroll
        for (int h = 0; h < NH; ++h) acc[r][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(a23[r].x, bw[ky][h][2], acc[r][h], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < TH; ++r)
#pragma unroll
        for (int h = 0; h < NH; ++h) acc[r][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(a23[r].y, bw[ky][h][3], acc[r][h], 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < TH; ++r)
      if (r != wave) {
#pragma unroll
        for (int h = 0; h < NH; ++h) red[wave][r < wave ? r : r - 1][h][lane] = acc[r][h];
      }
    __syncthreads();
    // ---- row `wave`: add the four channel partials in order, epilogue, float4 stores
    // lane (slot, q): 4 consecutive ox = 4*slot .. 4*slot+3 of channel 4h + q
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const f32x4 own = wave == 0 ? acc[0][h] : wave == 1 ? acc[1][h] : wave == 2 ? acc[2][h] : acc[3][h];
      f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < CIN; ++c)  // channel order 0..3 whichever wavefront adds
        s4 += (c == wave) ? own : red[c][wave < c ? wave : wave - 1][h][lane];
      const long g = (long)b * y_bs + ((long)(4 * h + q) * Ho + m0 + wave) * Wo + n0 + 4 * (lane >> 2);
      f32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = s4[r] + bv[h];
        o[r] = act ? elu(v) : v;
      }
      if (dact) {
        const f32x4 sv = Elem<TO>::ld4(dact + g);
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] *= elu_grad_from_out(sv[r]);
      }
      Elem<TO>::st4(y + g, o);
    }
  }
}

// ----------------------------------------------------------------------------------------------
// One-pass backward of the outermost 2-D decoder layer, tconv5 (8 -> 4 channels, src/lofar_models.py:57):
// data gradient (conv2d_q4_kernel above, with the ELU' multiply), weight gradient and bias gradient
// (conv2d_wgrad_direct_kernel<8, 4, 4, 64>) from ONE staging of the gradient image's patch in LDS and one read of
// the saved input: the two kernels read the same 67 MB gradient image and the same 33 MB saved input, and stage
// the same 4 x 10 x 130 patch.  Per tile: patch -> LDS; data gradient on v_mfma_f32_4x4x1 (wavefront = input
// channel, partial rows combined through LDS in channel order, exactly as conv2d_q4_kernel does: bitwise the same
// data gradient); the epilogue's float4s of the saved input go to the LDS small tile on their way through the ELU'
// multiply; weight gradient on v_mfma_f32_16x16x4 from the small tile and the patch, accumulators kept in
// registers across the tiles of the persistent workgroup (the same MFMA sequence as the stand-alone kernel).
// ----------------------------------------------------------------------------------------------
template <class TB, class TS = float>  // TB: element type of `big` (the gradient image), TS: of `small` and `dsmall`
__global__ __launch_bounds__(256, 2) void tconv2d_bwd_fused_kernel(const float* __restrict__ big_, long big_bs,
                                                                   const float* __restrict__ small_, long s_bs,
                                                                   const float* __restrict__ w,
                                                                   float* __restrict__ dsmall_, float* __restrict__ partial,
                                                                   int Hs, int Ws, int ntiles, int dact) {
  const TB* __restrict__ big = reinterpret_cast<const TB*>(big_);
  const TS* __restrict__ small = reinterpret_cast<const TS*>(small_);
  TS* __restrict__ dsmall = reinterpret_cast<TS*>(dsmall_);
  constexpr int CS = 8, CB = 4, TH = 4, TW = 64, NH = CS / 4;
  constexpr int PH = 2 * TH + 2, PW = 2 * TW + 2;
  constexpr int TP = TH * TW, LDS_S = TP + 2;
  constexpr int SLAB = CS * CB * 16 + 16;
  __shared__ __attribute__((aligned(16))) float patch[CB * PH * PW];
  __shared__ f32x4 red[CB][TH - 1][NH][64];
  __shared__ float stile[16 * LDS_S];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int q = lane & 3, lm = lane & 15, lk = lane >> 4;
  for (int i = t; i < 16 * LDS_S; i += 256) stile[i] = 0.f;  // rows >= CS stay zero
  // data-gradient B fragments: bw[ky][h] = taps kx 0..3 of w[4h + q][wave][ky][:]  (conv view: out = small ch, in = big ch)
  f32x4 bw[4][NH];
#pragma unroll
  for (int ky = 0; ky < 4; ++ky)
#pragma unroll
    for (int h = 0; h < NH; ++h)
      bw[ky][h] = *reinterpret_cast<const f32x4*>(w + (((long)(4 * h + q) * CB + wave) * 4 + ky) * 4);
  constexpr bool BF16_MMA = sizeof(TB) == 2;  // the gradient image is a bf16 tensor: the bf16 configuration
  bf16x4 bwh[4][NH];
#pragma unroll
  for (int ky = 0; ky < 4; ++ky)
#pragma unroll
    for (int h = 0; h < NH; ++h) bwh[ky][h] = __builtin_convertvector(bw[ky][h], bf16x4);
  f32x4 wacc[CB];
#pragma unroll
  for (int j = 0; j < CB; ++j) wacc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int NV4 = (CB * PH * (2 * TW / 4) + 255) / 256, NHL = (CB * PH * 2 + 255) / 256;
  float bs_big[NV4];
#pragma unroll
  for (int k = 0; k < NV4; ++k) bs_big[k] = 0.f;

  const int tiles_x = Ws / TW, tiles_y = Hs / TH;
  const int Hb = 2 * Hs, Wb = 2 * Ws;
  f32x4 rv[NV4], rs[NH];
  float rh[NHL];
  auto load_tile = [&](int tile) {
    const int b = tile / (tiles_x * tiles_y);
    const int tr_ = tile - b * (tiles_x * tiles_y);
    const int m0 = (tr_ / tiles_x) * TH, n0 = (tr_ % tiles_x) * TW;
    const TB* xb = big + (long)b * big_bs;
#pragma unroll
    for (int k = 0; k < NV4; ++k) {
      const int i = k * 256 + t;
      const int c4 = i % (2 * TW / 4), rr = i / (2 * TW / 4);
      const int prow = rr % PH, ci = rr / PH;
      const int iy = 2 * m0 - 1 + prow;
      rv[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (i < CB * PH * (2 * TW / 4) && (unsigned)iy < (unsigned)Hb)
        rv[k] = Elem<TB>::ld4(xb + ((long)ci * Hb + iy) * Wb + 2 * n0 + 4 * c4);
    }
#pragma unroll
    for (int k = 0; k < NHL; ++k) {
      const int i = k * 256 + t;
      const int side = i & 1, rr = i >> 1;
      const int prow = rr % PH, ci = rr / PH;
      const int iy = 2 * m0 - 1 + prow, ix = side ? 2 * n0 + 2 * TW : 2 * n0 - 1;
      rh[k] = 0.f;
      if (i < CB * PH * 2 && (unsigned)iy < (unsigned)Hb && (unsigned)ix < (unsigned)Wb) rh[k] = Elem<TB>::ld(xb + ((long)ci * Hb + iy) * Wb + ix);
    }
    // the saved input this thread finishes in the epilogue: row m0 + wave, columns n0 + 4 (lane / 4) .., channel 4h + q
#pragma unroll
    for (int h = 0; h < NH; ++h)
      rs[h] = Elem<TS>::ld4(small + (long)b * s_bs + ((long)(4 * h + q) * Hs + m0 + wave) * Ws + n0 + 4 * (lane >> 2));
  };
  if ((int)blockIdx.x < ntiles) load_tile(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int b = tile / (tiles_x * tiles_y);
    const int tr_ = tile - b * (tiles_x * tiles_y);
    const int m0 = (tr_ / tiles_x) * TH, n0 = (tr_ % tiles_x) * TW;
    __syncthreads();  // the previous tile's weight-gradient loop has finished reading the patch and the small tile
#pragma unroll
    for (int k = 0; k < NV4; ++k) {
      const int i = k * 256 + t;
      if (i < CB * PH * (2 * TW / 4)) {
        const int c4 = i % (2 * TW / 4), rr = i / (2 * TW / 4);
        const int prow = rr % PH;
        float* d = &patch[rr * PW + 1 + 4 * c4];  // odd offset: b32 + b64 + b32
        d[0] = rv[k][0];
        *reinterpret_cast<float2*>(d + 1) = make_float2(rv[k][1], rv[k][2]);
        d[3] = rv[k][3];
        // halo rows belong to the neighbouring tiles: only the 2*TH interior rows count towards the bias gradient
        if (prow >= 1 && prow <= 2 * TH) bs_big[k] += (rv[k][0] + rv[k][1]) + (rv[k][2] + rv[k][3]);
      }
    }
#pragma unroll
    for (int k = 0; k < NHL; ++k) {
      const int i = k * 256 + t;
      if (i < CB * PH * 2) patch[(i >> 1) * PW + ((i & 1) ? PW - 1 : 0)] = rh[k];
    }
    f32x4 sv[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) sv[h] = rs[h];
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);

    // ---- data gradient: this wavefront's big channel, all rows (conv2d_q4_kernel)
    f32x4 acc[TH][NH];
#pragma unroll
    for (int r = 0; r < TH; ++r)
#pragma unroll
      for (int h = 0; h < NH; ++h) acc[r][h] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
      float2 a01[TH], a23[TH];
#pragma unroll
      for (int r = 0; r < TH; ++r) {
        const float* pr = &patch[(wave * PH + 2 * r + ky) * PW + 2 * lane];
        a01[r] = *reinterpret_cast<const float2*>(pr);
        a23[r] = *reinterpret_cast<const float2*>(pr + 2);
      }
      if constexpr (BF16_MMA) {
#pragma unroll
        for (int r = 0; r < TH; ++r)
#pragma unroll
          for (int h = 0; h < NH; ++h) acc[r][h] = q4_mma(a01[r].x, a01[r].y, a23[r].x, a23[r].y, bwh[ky][h], acc[r][h]);
      } else {
#pragma unroll
      for (int r = 0; r < TH; ++r)
#pragma unroll
        for (int h = 0; h < NH; ++h) acc[r][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(a01[r].x, bw[ky][h][0], acc[r][h], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < TH; ++r)
#pragma unroll
        for (int h = 0; h < NH; ++h) acc[r][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(a01[r].y, bw[ky][h][1], acc[r][h], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < TH; ++r)
#pragma unroll
        for (int h = 0; h < NH; ++h) acc[r][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(a23[r].x, bw[ky][h][2], acc[r][h], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < TH; ++r)
#pragma unroll
        for (int h = 0; h < NH; ++h) acc[r][h] = __builtin_amdgcn_mfma_f32_4x4x1f32(a23[r].y, bw[ky][h][3], acc[r][h], 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < TH; ++r)
      if (r != wave) {
#pragma unroll
        for (int h = 0; h < NH; ++h) red[wave][r < wave ? r : r - 1][h][lane] = acc[r][h];
      }
    __syncthreads();
    // ---- row `wave`: add the four channel partials in order, ELU' multiply, float4 stores; the saved input goes to the small tile
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const f32x4 own = wave == 0 ? acc[0][h] : wave == 1 ? acc[1][h] : wave == 2 ? acc[2][h] : acc[3][h];
      f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < CB; ++c) s4 += (c == wave) ? own : red[c][wave < c ? wave : wave - 1][h][lane];
      const long g = (long)b * s_bs + ((long)(4 * h + q) * Hs + m0 + wave) * Ws + n0 + 4 * (lane >> 2);
      f32x4 o = s4;
      if (dact) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] *= elu_grad_from_out(sv[h][r]);
      }
      Elem<TS>::st4(dsmall + g, o);
      float* d = &stile[(4 * h + q) * LDS_S + wave * TW + 4 * (lane >> 2)];
      d[0] = sv[h][0]; d[1] = sv[h][1]; d[2] = sv[h][2]; d[3] = sv[h][3];
    }
    __syncthreads();
    // ---- weight gradient: groups of 4 consecutive positions, every 4th group per wave (conv2d_wgrad_direct_kernel)
    const int ky = lm >> 2, kx = lm & 3;
#pragma unroll 4
    for (int it = 0; it < TP / 16; ++it) {  // (a constant trip count: `s = wave; s < TP / 4; s += 4` is refused by the unroller)
      const int s = wave + 4 * it;
      const int p = 4 * s + lk;
      const int oy = p / TW, ox = p - oy * TW;
      const float av = stile[lm * LDS_S + p];
      const int boff = (2 * oy + ky) * PW + 2 * ox + kx;
#pragma unroll
      for (int j = 0; j < CB; ++j) wacc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, patch[j * PH * PW + boff], wacc[j], 0, 0, 0);
    }
  }
  // ---- combine the 4 waves (fixed order) and write this workgroup's slab [CS][CB*16] + bias sums
  __syncthreads();
  float* out = partial + (size_t)blockIdx.x * SLAB;
  {
    static_assert(CB * PH * PW >= 4 * 16 * (CB * 16 + 1), "combine buffer must fit in the patch");
    float (*comb)[16][CB * 16 + 1] = reinterpret_cast<float (*)[16][CB * 16 + 1]>(patch);
#pragma unroll
    for (int j = 0; j < CB; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) comb[wave][4 * lk + r][16 * j + lm] = wacc[j][r];
    __syncthreads();
    for (int i = t; i < CS * CB * 16; i += 256) {
      const int m = i / (CB * 16), n = i - m * (CB * 16);
      out[i] = (comb[0][m][n] + comb[1][m][n]) + (comb[2][m][n] + comb[3][m][n]);
    }
  }
  __syncthreads();
  float* bred = patch;  // [16 channels][4 waves]
  for (int c = 0; c < CB; ++c) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < NV4; ++k) {
      const int i = k * 256 + t;
      if (i < CB * PH * (2 * TW / 4) && (i / (2 * TW / 4)) / PH == c) v += bs_big[k];
    }
    v = wave_sum(v);
    if (lane == 0) bred[c * 4 + wave] = v;
  }
  __syncthreads();
  if (t < 16) out[CS * CB * 16 + t] = t < CB ? (bred[t * 4] + bred[t * 4 + 1]) + (bred[t * 4 + 2] + bred[t * 4 + 3]) : 0.f;
}

bool tconv2d_bwd_fused_supported(int Cs, int Cb, int Hs, int Ws) { return Cs == 8 && Cb == 4 && Hs % 4 == 0 && Ws % 64 == 0; }
// weight + bias + data gradient of the 8 -> 4 transposed layer; one slab per workgroup in ws (conv2d_wgrad_direct_workspace_floats)
int tconv2d_bwd_fused(const float* small, long s_bs, const float* big, long big_bs, const float* w, float* dsmall, int dact,
                      float* dw, float* db, int B, int Hs, int Ws, float* ws, size_t wsf, int accumulate, hipStream_t st,
                      GradJobs* defer, int big_bf16, int small_bf16) {
  constexpr int Cs = 8, Cb = 4;
  if (wsf < conv2d_wgrad_direct_workspace_floats(Cs, Cb)) { set_last_error("tconv2d_bwd_fused: workspace too small"); return LSHM_ERR_WORKSPACE; }
  if (!tconv2d_bwd_fused_supported(Cs, Cb, Hs, Ws) || s_bs % 4 || big_bs % 4 || (reinterpret_cast<uintptr_t>(small) & 15) ||
      (reinterpret_cast<uintptr_t>(big) & 15) || (reinterpret_cast<uintptr_t>(dsmall) & 15) || (reinterpret_cast<uintptr_t>(w) & 15) ||
      (small_bf16 && !big_bf16)) {
    set_last_error("tconv2d_bwd_fused: unsupported shape, alignment or storage combination");
    return LSHM_ERR_UNSUPPORTED;
  }
  const int ntiles = (Ws / 64) * (Hs / 4) * B;
  const int grid = ntiles < 512 ? ntiles : 512;  // 62 KB of LDS: two workgroups per CU
  int rc;
#define LSHM_F2D(TB_, TS_)                                                                                                       \
  do {                                                                                                                           \
    if ((rc = kernel_budget_ok(reinterpret_cast<const void*>(&tconv2d_bwd_fused_kernel<TB_, TS_>), 256, 0, "tconv2d_bwd_fused"))) return rc; \
    hipLaunchKernelGGL((tconv2d_bwd_fused_kernel<TB_, TS_>), dim3(grid), dim3(256), 0, st, big, big_bs, small, s_bs, w, dsmall, ws, Hs, \
                       Ws, ntiles, dact);                                                                                        \
  } while (0)
  if (big_bf16 && small_bf16) LSHM_F2D(bf16, bf16);
  else if (big_bf16) LSHM_F2D(bf16, float);
  else LSHM_F2D(float, float);
#undef LSHM_F2D
  if ((rc = check_launch("tconv2d_bwd_fused"))) return rc;
  const int nw = Cs * Cb * 16, slab = nw + 16;
  if (defer) {
    defer->sums.push_back(SumJob{ws, dw, slab, nw, grid, 0, 0, 0, 0, accumulate, 0});
    if (db) defer->sums.push_back(SumJob{ws + nw, db, slab, Cb, grid, 0, 0, 0, 0, accumulate, 0});
    return LSHM_OK;
  }
  rc = reduce_partials_strided(ws, slab, dw, nw, grid, accumulate, st);
  if (rc || !db) return rc;
  return reduce_partials_strided(ws + nw, slab, db, Cb, grid, accumulate, st);
}

bool conv2d_direct_supported(int Cin, int Cout, int Ho, int Wo) {
  if (Cin == 4 && Cout == 8) return Ho % 4 == 0 && Wo % 64 == 0;
  if (Cin == 8 && Cout == 12) return Ho % 8 == 0 && Wo % 32 == 0;
  return false;
}

int conv2d_direct(const float* x, long x_bs, const float* w, const float* bias, float* y, long y_bs,
                  const float* dact, int B, int Cin, int Cout, int Ho, int Wo, int act, hipStream_t st, int x_bf16, int y_bf16) {
  // bf16 storage: x for both outer layer shapes; y (and dact, which has its shape) for the outermost one
  if (y_bf16 && !(Cin == 4 && Cout == 8)) { set_last_error("conv2d_direct: a bf16 output only for the outermost layer"); return LSHM_ERR_UNSUPPORTED; }
  if (Cin == 4 && Cout == 8) {
    const int ntiles = (Wo / 64) * (Ho / 4) * B;
    const dim3 grid(ntiles < 768 ? ntiles : 768);
#define LSHM_Q4(TI_, TO_) hipLaunchKernelGGL((conv2d_q4_kernel<8, 4, TI_, TO_>), grid, dim3(256), 0, st, x, x_bs, w, bias, y, y_bs, dact, Ho, Wo, act, ntiles)
    if (x_bf16 && y_bf16) LSHM_Q4(bf16, bf16);
    else if (x_bf16) LSHM_Q4(bf16, float);
    else if (y_bf16) LSHM_Q4(float, bf16);
    else LSHM_Q4(float, float);
#undef LSHM_Q4
  } else if (Cin == 8 && Cout == 12) {
    const int ntiles = (Wo / 32) * (Ho / 4) * B;
    constexpr int cap = 512;  // persistent workgroups over four 4 x 32 tiles each (256 / 1024 / 2048: 26.3 / 22.8 / 29.0 us, profiles/r03)
    const dim3 grid(ntiles < cap ? ntiles : cap);
    if (x_bf16)
      hipLaunchKernelGGL((conv2d_direct_kernel<8, 12, 4, 32, bf16>), grid, dim3(256), 0, st, x, x_bs, w, bias, y, y_bs, dact, Ho, Wo, act, ntiles);
    else
      hipLaunchKernelGGL((conv2d_direct_kernel<8, 12, 4, 32>), grid, dim3(256), 0, st, x, x_bs, w, bias, y, y_bs, dact, Ho, Wo, act, ntiles);
  } else {
    set_last_error("conv2d_direct: unsupported shape");
    return LSHM_ERR_UNSUPPORTED;
  }
  return check_launch("conv2d_direct");
}

}  // namespace lshm

namespace lshm {

// ----------------------------------------------------------------------------------------------
// k4 s4 1-D weight gradient for the outer layers of the 1-D autoencoders (src/lofar_models.py:115-117,
// :141-142), bias gradient fused: launcher around conv1d_wgrad_stream_kernel (conv1d_stream.hip); one
// slab of Cs*Cb*4 + 16 floats per workgroup, summed here or queued on the backward's job list.
// ----------------------------------------------------------------------------------------------
// ----------------------------------------------------------------------------------------------
// k4 s4 weight gradient of the MID 1-D layers (conv2 / tconv3: 24 small, 12 big channels; conv3 / tconv2:
// 48 / 24), bias gradient fused:   dW[cs, cb, t] = sum_{b,j} small[b,cs,j] * big[b,cb,4j-pad+t]
// GEMM view: M = Cs, N = Cb*4 (an n-tile = 4 big channels x 4 taps), K = every position of the batch --
// a tiny output over a huge K, which the implicit-GEMM template serves badly (30-45 us per pair at B=256).
// A workgroup walks tiles of TP positions: the small tile [Cs][TP] and the matching big segment [Cb][4 TP]
// are staged once in LDS with float4 loads, each wavefront takes every 4th group of 4 positions, fragments
// are read straight out of the images; accumulators stay in registers across all tiles of the (persistent)
// workgroup; the four wavefronts then add into one image in a fixed order.  One slab per workgroup.
// ----------------------------------------------------------------------------------------------
template <int CS, int CB, int TP>
__global__ __launch_bounds__(256) void conv1d_wgrad_mid_kernel(const float* __restrict__ small0,
                                                               const float* __restrict__ small1, long s_bs,
                                                               const float* __restrict__ big0,
                                                               const float* __restrict__ big1, long big_bs,
                                                               float* __restrict__ partial0,
                                                               float* __restrict__ partial1, int Ls, int Lb, int pad,
                                                               int bias_from, int ntiles) {
  const float* small = blockIdx.y ? small1 : small0;
  const float* big = blockIdx.y ? big1 : big0;
  float* partial = blockIdx.y ? partial1 : partial0;
  constexpr int MT = (CS + 15) / 16, NT = CB / 4;
  constexpr int NW = CS * CB * 4;
  constexpr int BPAD = ((CS > CB ? CS : CB) + 15) / 16 * 16;
  constexpr int SLAB = NW + BPAD;
  constexpr int LDS_S = TP + 2;        // == 2 (mod 32): the 16 rows x 4 positions of an A fragment hit 32 banks
  constexpr int BP = 4 * TP + 8;       // big row pitch, == 8 (mod 32): 4 channels x 4 taps x 2 positions conflict-free
  static_assert(CB % 4 == 0 && TP % 64 == 0, "tile shape");
  constexpr int CPITCH = CB * 4 + 4;   // combine row pitch (see the 2-D kernel)
  constexpr int TILE_FLOATS = 16 * MT * LDS_S + CB * BP, COMB_FLOATS = 4 * CS * CPITCH;
  __shared__ float smem_w[TILE_FLOATS > COMB_FLOATS ? TILE_FLOATS : COMB_FLOATS];
  float* stile = smem_w;
  float* bimg = smem_w + 16 * MT * LDS_S;  // bimg[cb][1 + i] = big[cb][4 j0 + i]; [0] = element before
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lm = lane & 15, lk = lane >> 4;
  for (int i = t; i < 16 * MT * LDS_S; i += 256) stile[i] = 0.f;  // rows >= CS stay zero
  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int NQS = (CS * (TP / 4) + 255) / 256, NQB = (CB * TP + 255) / 256;
  float bs_small[NQS], bs_big[NQB];
#pragma unroll
  for (int q = 0; q < NQS; ++q) bs_small[q] = 0.f;
#pragma unroll
  for (int q = 0; q < NQB; ++q) bs_big[q] = 0.f;
  const int tiles_per = Ls / TP;
  // lane (lm, lk) of a B fragment: big channel 4 jt + (lm >> 2), tap lm & 3, position 4 s + lk
  const int bofs = (lm >> 2) * BP + (lm & 3) + 1 - pad;
  // software pipeline over the tiles (see conv2d_wgrad_direct_kernel): loads of tile i+1 fly during the MFMAs of tile i
  f32x4 rs[NQS], rb[NQB];
  float rh = 0.f;
  auto fetch = [&](int tile) {
    const int b = tile / tiles_per, j0 = (tile - b * tiles_per) * TP;
    const float* sb = small + (long)b * s_bs + j0;
    const float* bb = big + (long)b * big_bs + 4L * j0;
#pragma unroll
    for (int q = 0; q < NQS; ++q) {
      const int i = t + 256 * q;
      rs[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (i < CS * (TP / 4)) {
        const int c4 = i % (TP / 4), cs = i / (TP / 4);
        rs[q] = *reinterpret_cast<const f32x4*>(sb + (long)cs * Ls + 4 * c4);
      }
    }
#pragma unroll
    for (int q = 0; q < NQB; ++q) {
      const int i = t + 256 * q;
      rb[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (i < CB * TP) {
        const int c4 = i % TP, cb = i / TP;
        rb[q] = *reinterpret_cast<const f32x4*>(bb + (long)cb * Lb + 4 * c4);
      }
    }
    // the element before the segment (pad = 1: tap 0 of the first position; zero at the row start)
    rh = (t < CB && pad && j0 > 0) ? bb[(long)t * Lb - 1] : 0.f;
  };
  if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQS; ++q) {
      const int i = t + 256 * q;
      if (i < CS * (TP / 4)) {
        const int c4 = i % (TP / 4), cs = i / (TP / 4);
        const f32x4 v = rs[q];
        float* d = &stile[cs * LDS_S + 4 * c4];
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
        bs_small[q] += (v[0] + v[1]) + (v[2] + v[3]);
      }
    }
#pragma unroll
    for (int q = 0; q < NQB; ++q) {
      const int i = t + 256 * q;
      if (i < CB * TP) {
        const int c4 = i % TP, cb = i / TP;
        const f32x4 v = rb[q];
        float* d = &bimg[cb * BP + 1 + 4 * c4];
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
        bs_big[q] += (v[0] + v[1]) + (v[2] + v[3]);
      }
    }
    if (t < CB) bimg[t * BP] = rh;
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);
#pragma unroll 4
    for (int it = 0; it < TP / 16; ++it) {  // (a constant trip count: `s = wave; s < TP / 4; s += 4` is refused by the unroller)
      const int s = wave + 4 * it;
      const int p = 4 * s + lk;
      float a[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a[mt] = stile[(16 * mt + lm) * LDS_S + p];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const float bv = bimg[4 * j * BP + bofs + 4 * p];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], bv, acc[mt][j], 0, 0, 0);
      }
    }
  }
  // ---- every wavefront leaves its partial image [CS][CB*4] (lane: rows 4 lk .. + 3, column 16 j + lm =
  // (cb = 4 j + lm / 4) * 4 + tap), then all threads add the four in wavefront order
  __syncthreads();
  float* comb = smem_w + wave * CS * CPITCH;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * mt + 4 * lk + r;
        if (row < CS) comb[row * CPITCH + 16 * j + lm] = acc[mt][j][r];
      }
  __syncthreads();
  float* out = partial + (size_t)blockIdx.x * SLAB;
  for (int i = t; i < NW; i += 256) {
    const int m = i / (CB * 4), n = i - m * (CB * 4);
    const float* c0 = smem_w + m * CPITCH + n;
    out[i] = (c0[0] + c0[CS * CPITCH]) + (c0[2 * CS * CPITCH] + c0[3 * CS * CPITCH]);
  }
  __syncthreads();
  // ---- bias partials: a thread always stages the same channel (tile-invariant index -> channel map)
  float* bred = bimg;  // [BPAD channels][4 waves]
  if (bias_from) {
    const int nch = bias_from == 1 ? CS : CB;
    for (int c = 0; c < nch; ++c) {
      float v = 0.f;
      if (bias_from == 1) {
#pragma unroll
        for (int q = 0; q < NQS; ++q) {
          const int i = t + 256 * q;
          if (i < CS * (TP / 4) && i / (TP / 4) == c) v += bs_small[q];
        }
      } else {
#pragma unroll
        for (int q = 0; q < NQB; ++q) {
          const int i = t + 256 * q;
          if (i < CB * TP && i / TP == c) v += bs_big[q];
        }
      }
      v = wave_sum(v);
      if (lane == 0) bred[c * 4 + wave] = v;
    }
  }
  __syncthreads();
  if (t < BPAD) {
    const int nch = bias_from == 1 ? CS : bias_from == 2 ? CB : 0;
    out[NW + t] = t < nch ? (bred[t * 4] + bred[t * 4 + 1]) + (bred[t * 4 + 2] + bred[t * 4 + 3]) : 0.f;
  }
}

bool conv1d_wgrad_mid_supported(int Cs, int Cb, int Ls, int Lb, int pad, int bias_from, long s_bs, long big_bs,
                                const float* small, const float* big) {
  const bool shape = (Cs == 24 && Cb == 12 && Ls % 128 == 0) || (Cs == 48 && Cb == 24 && Ls % 64 == 0);
  return shape && Lb == 4 * Ls && (pad == 0 || pad == 1) && !(bias_from == 2 && pad != 0) && s_bs % 4 == 0 &&
         big_bs % 4 == 0 && (reinterpret_cast<uintptr_t>(small) & 15) == 0 && (reinterpret_cast<uintptr_t>(big) & 15) == 0;
}
size_t conv1d_wgrad_mid_workspace_floats(int Cs, int Cb) { return (size_t)1024 * (Cs * Cb * 4 + wgrad_bias_pad(Cs, Cb)); }

// second problem (small2, big2, dw2, db2) optional: both run in one launch; the closing sums are queued on `defer`
// (or run in place without it)
int conv1d_wgrad_mid(const float* small, long s_bs, const float* big, long big_bs, float* dw, float* db, int bias_from,
                     int B, int Cs, int Cb, int Ls, int Lb, int pad, float* ws, size_t wsf, int accumulate,
                     hipStream_t st, const float* small2, const float* big2, float* dw2, float* db2, GradJobs* defer) {
  const int G = small2 ? 2 : 1;
  if (wsf < G * conv1d_wgrad_mid_workspace_floats(Cs, Cb)) { set_last_error("conv1d_wgrad_mid: workspace too small"); return LSHM_ERR_WORKSPACE; }
  if (!db) bias_from = 0;
  float* ws2 = ws + conv1d_wgrad_mid_workspace_floats(Cs, Cb);
  const int nw = Cs * Cb * 4, slab = nw + wgrad_bias_pad(Cs, Cb);
  const int TP = Cs == 24 ? 128 : 64;
  const int ntiles = (Ls / TP) * B;
  constexpr int capmid = 256;
  // (workgroups over the pair = partial slabs for the closing sums; per iteration 1024 / 512 / 256 / 128: 2.037 / 2.033 / 2.019 / 2.022 ms)
  int grid = ntiles < capmid / G ? ntiles : capmid / G;
  if (grid < 1) grid = 1;
  const dim3 g(grid, G);
  if (Cs == 24)
    hipLaunchKernelGGL((conv1d_wgrad_mid_kernel<24, 12, 128>), g, dim3(256), 0, st, small, small2, s_bs, big, big2, big_bs,
                       ws, ws2, Ls, Lb, pad, bias_from, ntiles);
  else
    hipLaunchKernelGGL((conv1d_wgrad_mid_kernel<48, 24, 64>), g, dim3(256), 0, st, small, small2, s_bs, big, big2, big_bs,
                       ws, ws2, Ls, Lb, pad, bias_from, ntiles);
  int rc = check_launch("conv1d_wgrad_mid");
  if (rc) return rc;
  const int nbias = bias_from == 1 ? Cs : Cb;
  for (int q = 0; q < G; ++q) {
    const float* part = q ? ws2 : ws;
    float* dwq = q ? dw2 : dw;
    float* dbq = q ? db2 : db;
    if (defer) {
      defer->sums.push_back(SumJob{part, dwq, slab, nw, grid, 0, 0, 0, 0, accumulate, 0});
      if (bias_from) defer->sums.push_back(SumJob{part + nw, dbq, slab, nbias, grid, 0, 0, 0, 0, accumulate, 0});
    } else {
      if ((rc = reduce_partials_strided(part, slab, dwq, nw, grid, accumulate, st))) return rc;
      if (bias_from && (rc = reduce_partials_strided(part + nw, slab, dbq, nbias, grid, accumulate, st))) return rc;
    }
  }
  return LSHM_OK;
}

bool conv1d_wgrad_direct_supported(int Cs, int Cb, int Ls) {
  return ((Cs == 8 && Cb == 4) || (Cs == 12 && Cb == 8)) && Ls % 64 == 0;
}
size_t conv1d_wgrad_direct_workspace_floats(int Cs, int Cb) { return (size_t)2048 * (Cs * Cb * 4 + 16); }

// second problem (small2, big2, dw2, db2) optional: both run in one launch
int conv1d_wgrad_direct(const float* small, long s_bs, const float* big, long big_bs, float* dw, float* db,
                        int bias_from, int nbias, int B, int Cs, int Cb, int Ls, int Lb, int pad, float* ws,
                        size_t wsf, int accumulate, hipStream_t st, const float* small2, const float* big2,
                        float* dw2, float* db2, GradJobs* defer, int big_bf16, const FusedDgrad* fd, int small_bf16) {
  const int G = small2 ? 2 : 1;
  if (wsf < G * conv1d_wgrad_direct_workspace_floats(Cs, Cb)) { set_last_error("conv1d_wgrad_direct: workspace too small"); return LSHM_ERR_WORKSPACE; }
  if (!db) bias_from = 0;
  if (!conv1d_wgrad_stream_supported(Cs, Cb, Ls, Lb, pad, bias_from, s_bs, big_bs, small, big) ||
      (small2 && !conv1d_wgrad_stream_supported(Cs, Cb, Ls, Lb, pad, bias_from, s_bs, big_bs, small2, big2))) {
    set_last_error("conv1d_wgrad_direct: unsupported shape or alignment");
    return LSHM_ERR_UNSUPPORTED;
  }
  float* ws2 = ws + conv1d_wgrad_direct_workspace_floats(Cs, Cb);
  const int slab = Cs * Cb * 4 + 16;
  int grid = 0;
  int rc = conv1d_wgrad_stream(small, small2, s_bs, big, big2, big_bs, ws, ws2, B, Cs, Cb, Ls, Lb, pad, bias_from,
                               2048 / G, st, &grid, big_bf16, fd, small_bf16);

  if (rc) return rc;
  if (defer) {
    const int nw = Cs * Cb * 4;
    for (int g = 0; g < G; ++g) {
      const float* part = g ? ws2 : ws;
      defer->sums.push_back(SumJob{part, g ? dw2 : dw, slab, nw, grid, 0, 0, 0, 0, accumulate, 0});
      if (bias_from) defer->sums.push_back(SumJob{part + nw, g ? db2 : db, slab, nbias, grid, 0, 0, 0, 0, accumulate, 0});
    }
    return LSHM_OK;
  }
  // strided reduces over the slabs: weights, then (optionally) the bias entries
  rc = reduce_partials_strided(ws, slab, dw, (long)Cs * Cb * 4, grid, accumulate, st, small2 ? ws2 : nullptr, dw2);
  if (rc || !bias_from) return rc;
  return reduce_partials_strided(ws + (long)Cs * Cb * 4, slab, db, nbias, grid, accumulate, st,
                                 small2 ? ws2 + (long)Cs * Cb * 4 : nullptr, db2);
}

}  // namespace lshm
