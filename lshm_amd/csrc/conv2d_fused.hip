// One-pass backward of the 12 <-> 8 channel layers of the 2-D autoencoder (tconv4: src/lofar_models.py:56, conv1: :32):
// weight, bias AND data gradient from one staging of dz and the saved input.
//
// The weight-gradient kernel of these layers (conv2d_wgrad_direct_kernel<12,8,..>, conv_direct.hip) already stages, per
// tile of TH x TW small positions, the small tile and the matching big patch (halo included) in LDS, software-
// pipelined over the tiles of a persistent workgroup.  Everything the data gradient needs is in those two images:
//   transposed layer (tconv4; small = saved input a, big = dz):
//     dsmall[cs][m][n] = ELU'(a) * sum_{cb,ky,kx} dz[cb][2m-1+ky][2n-1+kx] w[cs][cb][ky][kx]
//     -- the stride-2 conv of conv2d_direct_kernel: M = 16 consecutive n, N = cs, k-steps (cb, ky) with the 4 kx taps,
//        A fragments straight from the big patch, ELU' from the small tile, float4 stores from the accumulators;
//   conv layer (conv1; small = dz, big = saved input x):
//     dbig[cb][2m+py][2n+px] = ELU'(x) * sum_{cs,dy,dx} dz[cs][m+dy][n+dx] w[cs][cb][py-2dy+1][px-2dx+1]
//     -- one GEMM per row parity py: M = 16 consecutive n, N = (px, cb) = 16, K = (cs, 2 x 3 neighbourhood) = 72 (the
//        all-parity form of tconv2d_direct_kernel has N = 32, K = 108: 3/2 of the matrix instructions and of the
//        weight registers); the small tile is staged WITH its halo ring for this (the weight gradient reads its
//        interior), results go through an LDS output tile, ELU' comes from the interior of the big patch.
// Tiles are 4 x 32 small positions (8 x 64 big), 2048 of them at B = 256 over 512 persistent workgroups; the
// weight-gradient accumulators (32 registers) live across all tiles of a workgroup and leave as one slab per
// workgroup for the deferred sums (the format of conv2d_wgrad_direct).  v_mfma_f32_16x16x4_f32 throughout (exact fp32).
#include <stdlib.h>

#include "kernels.h"

namespace lshm {

namespace {
constexpr int CS = 12, CB = 8, TH = 4, TW = 32;
constexpr int TP = TH * TW;                      // small positions per tile
constexpr int PH = 2 * TH + 2, PW = 2 * TW + 2;  // big patch with halo
constexpr int SPH = TH + 2, SPW = TW + 2;        // small patch with halo (conv layer)
constexpr int LDS_S = TP + 2;                    // flat small-tile row pitch (transposed layer): == 2 (mod 32)
constexpr int NW = CS * CB * 16, BPAD = 16, SLAB = NW + BPAD;
constexpr int CPITCH = CB * 16 + 4;              // combine row pitch
constexpr int OW = 2 * TW;                       // output tile width (conv layer)
}  // namespace

struct Bwd2dArgs {
  const float* small; long s_bs;
  const float* big; long big_bs;
  const float* w;
  float* dout; long d_bs;
  float* partial;
  int Hs, Ws, ntiles;
};

// TB: element type of the big tensor and, for the conv layer, of its gradient (bf16 storage: DESIGN 4.6)
template <bool CONV, bool DACT, class TB = float>
__global__ __launch_bounds__(256, 2) void conv2d_bwd_lds_kernel(const Bwd2dArgs a) {
  constexpr int SM_FLOATS = CONV ? 16 * SPH * SPW : 16 * LDS_S;  // 16 rows: the MFMA row tile of the weight gradient; rows >= CS stay zero
  constexpr int OT_FLOATS = CONV ? CB * 2 * TH * OW : 0;
  constexpr int TILE_FLOATS = SM_FLOATS + CB * PH * PW + OT_FLOATS;
  static_assert(TILE_FLOATS >= 4 * CS * CPITCH, "the combine images alias the tile images");
  __shared__ __attribute__((aligned(16))) float smem[TILE_FLOATS];
  float* simg = smem;                    // small image
  float* patch = smem + SM_FLOATS;       // big patch [cb][PH][PW]
  float* otile = patch + CB * PH * PW;   // CONV: data-gradient tile [cb][2 TH][2 TW]
  const float* __restrict__ small = a.small;
  const TB* __restrict__ big = reinterpret_cast<const TB*>(a.big);
  const float* __restrict__ w = a.w;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lm = lane & 15, lk = lane >> 4;
  const int Hs = a.Hs, Ws = a.Ws, Hb = 2 * Hs, Wb = 2 * Ws;
  for (int i = t; i < SM_FLOATS; i += 256) simg[i] = 0.f;

  // ---- data-gradient weight fragments of this lane
  // conv layer: one GEMM per output row parity py with the two column parities side by side in the 16-wide tile:
  //   N = (px, cb), K = (cs, dyi in {0, 1}, dxp in {0, 1, 2}) = 72 (18 k-steps, 2/3 of the products useful; the form with all
  //   four parities in N has K = 108 and 4/9), dy = py - 1 + dyi, dx = dxp - 1, ky = 3 - py - 2 dyi, kx = px - 2 dxp + 3
  constexpr int KS = CONV ? CS * 6 / 4 : CB * 4;
  constexpr int NTD = CONV ? 2 : 1;  // CONV: index = py
  float bf[KS][NTD];
  if constexpr (CONV) {
    const int px = lm >> 3, co = lm & 7;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k = 4 * s + lk;
      const int cs = k / 6, r = k - cs * 6;
      const int dyi = r / 3, dxp = r - dyi * 3;
      const int kx = px - 2 * dxp + 3;
#pragma unroll
      for (int py = 0; py < 2; ++py) {
        const int ky = 3 - py - 2 * dyi;
        bf[s][py] = (kx >= 0 && kx <= 3) ? w[(((long)cs * CB + co) * 4 + ky) * 4 + kx] : 0.f;
      }
    }
  } else {
#pragma unroll
    for (int s = 0; s < KS; ++s) bf[s][0] = lm < CS ? w[((long)lm * CB * 4 + s) * 4 + lk] : 0.f;  // (cb, ky) = s, kx = lk
  }

  f32x4 acc[CB];  // weight gradient: rows = small channel, one 16-tap tile per big channel
#pragma unroll
  for (int j = 0; j < CB; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int SROWS = CONV ? SPH : TH;                               // small rows staged per channel
  constexpr int NQS = (CS * SROWS * (TW / 4) + 255) / 256;             // float4 of the small image per thread
  constexpr int NHS = CONV ? (CS * SPH * 2 + 255) / 256 : 0;           // its halo-column scalars
  constexpr int NQB = (CB * PH * (2 * TW / 4) + 255) / 256;            // float4 of the big patch per thread
  constexpr int NHB = (CB * PH * 2 + 255) / 256;                       // its halo-column scalars
  float bsum[CONV ? NQS : NQB];
#pragma unroll
  for (int q = 0; q < (CONV ? NQS : NQB); ++q) bsum[q] = 0.f;
  const int tiles_x = Ws / TW, tiles_y = Hs / TH;
  const int ky_l = lm >> 2, kx_l = lm & 3;
  f32x4 rs[NQS], rb[NQB];
  float rhs[NHS ? NHS : 1], rhb[NHB];
  auto fetch = [&](int tile) {
    const int b = tile / (tiles_x * tiles_y);
    const int tr_ = tile - b * (tiles_x * tiles_y);
    const int m0 = (tr_ / tiles_x) * TH, n0 = (tr_ % tiles_x) * TW;
    const float* sb = small + (long)b * a.s_bs;
    const TB* bb = big + (long)b * a.big_bs;
#pragma unroll
    for (int q = 0; q < NQS; ++q) {
      const int i = t + 256 * q;
      rs[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (i < CS * SROWS * (TW / 4)) {
        const int c4 = i % (TW / 4), rr = i / (TW / 4);
        const int row = rr % SROWS, cs = rr / SROWS;
        const int iy = m0 + row - (CONV ? 1 : 0);
        if ((unsigned)iy < (unsigned)Hs) rs[q] = *reinterpret_cast<const f32x4*>(sb + ((long)cs * Hs + iy) * Ws + n0 + 4 * c4);
      }
    }
    if constexpr (CONV) {
#pragma unroll
      for (int q = 0; q < NHS; ++q) {
        const int i = t + 256 * q;
        rhs[q] = 0.f;
        if (i < CS * SPH * 2) {
          const int side = i & 1, rr = i >> 1;
          const int row = rr % SPH, cs = rr / SPH;
          const int iy = m0 + row - 1, ix = side ? n0 + TW : n0 - 1;
          if ((unsigned)iy < (unsigned)Hs && (unsigned)ix < (unsigned)Ws) rhs[q] = sb[((long)cs * Hs + iy) * Ws + ix];
        }
      }
    }
#pragma unroll
    for (int q = 0; q < NQB; ++q) {
      const int i = t + 256 * q;
      rb[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (i < CB * PH * (2 * TW / 4)) {
        const int c4 = i % (2 * TW / 4), rr = i / (2 * TW / 4);
        const int prow = rr % PH, cb = rr / PH;
        const int iy = 2 * m0 - 1 + prow;
        if ((unsigned)iy < (unsigned)Hb) rb[q] = Elem<TB>::ld4(bb + ((long)cb * Hb + iy) * Wb + 2 * n0 + 4 * c4);
      }
    }
#pragma unroll
    for (int q = 0; q < NHB; ++q) {
      const int i = t + 256 * q;
      rhb[q] = 0.f;
      if (i < CB * PH * 2) {
        const int side = i & 1, rr = i >> 1;
        const int prow = rr % PH, cb = rr / PH;
        const int iy = 2 * m0 - 1 + prow, ix = side ? 2 * n0 + 2 * TW : 2 * n0 - 1;
        if ((unsigned)iy < (unsigned)Hb && (unsigned)ix < (unsigned)Wb) rhb[q] = Elem<TB>::ld(bb + ((long)cb * Hb + iy) * Wb + ix);
      }
    }
  };
  if ((int)blockIdx.x < a.ntiles) fetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int b = tile / (tiles_x * tiles_y);
    const int tr_ = tile - b * (tiles_x * tiles_y);
    const int m0 = (tr_ / tiles_x) * TH, n0 = (tr_ % tiles_x) * TW;
    __syncthreads();  // the previous tile is done with the LDS images
#pragma unroll
    for (int q = 0; q < NQS; ++q) {
      const int i = t + 256 * q;
      if (i < CS * SROWS * (TW / 4)) {
        const int c4 = i % (TW / 4), rr = i / (TW / 4);
        const int row = rr % SROWS, cs = rr / SROWS;
        const f32x4 v = rs[q];
        float* d = CONV ? &simg[(cs * SPH + row) * SPW + 1 + 4 * c4] : &simg[cs * LDS_S + row * TW + 4 * c4];
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
        if constexpr (CONV) {  // bias gradient of a conv layer: the sum of dz over the tile's own rows
          if (row >= 1 && row <= TH) bsum[q] += (v[0] + v[1]) + (v[2] + v[3]);
        }
      }
    }
    if constexpr (CONV) {
#pragma unroll
      for (int q = 0; q < NHS; ++q) {
        const int i = t + 256 * q;
        if (i < CS * SPH * 2) simg[(i >> 1) * SPW + ((i & 1) ? SPW - 1 : 0)] = rhs[q];
      }
    }
#pragma unroll
    for (int q = 0; q < NQB; ++q) {
      const int i = t + 256 * q;
      if (i < CB * PH * (2 * TW / 4)) {
        const int c4 = i % (2 * TW / 4), rr = i / (2 * TW / 4);
        const int prow = rr % PH;
        const f32x4 v = rb[q];
        float* d = &patch[rr * PW + 1 + 4 * c4];
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
        if constexpr (!CONV) {  // bias gradient of a transposed layer: the sum of dz over the tile's own rows
          if (prow >= 1 && prow <= 2 * TH) bsum[q] += (v[0] + v[1]) + (v[2] + v[3]);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < NHB; ++q) {
      const int i = t + 256 * q;
      if (i < CB * PH * 2) patch[(i >> 1) * PW + ((i & 1) ? PW - 1 : 0)] = rhb[q];
    }
    __syncthreads();
    if (tile + (int)gridDim.x < a.ntiles) fetch(tile + gridDim.x);
    // ---- weight gradient: groups of 4 consecutive positions, every 4th group per wavefront
#pragma unroll 2
    for (int it = 0; it < TP / 16; ++it) {  // (a constant trip count: `s = wave; s < TP / 4; s += 4` is refused by the unroller)
      const int s = wave + 4 * it;
      const int p = 4 * s + lk;
      const int oy = p / TW, ox = p - oy * TW;
      const float av = CONV ? simg[(lm * SPH + oy + 1) * SPW + ox + 1] : simg[lm * LDS_S + p];
      const int boff = (2 * oy + ky_l) * PW + 2 * ox + kx_l;
#pragma unroll
      for (int j = 0; j < CB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, patch[j * PH * PW + boff], acc[j], 0, 0, 0);
    }
    // ---- data gradient: 16-column row tiles, two per wavefront
    constexpr int TPR = TW / 16, MW = TH * TPR / 4;
    if constexpr (!CONV) {
      f32x4 d[MW];
#pragma unroll
      for (int i = 0; i < MW; ++i) d[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int cb = s >> 2, ky = s & 3;
#pragma unroll
        for (int i = 0; i < MW; ++i) {
          const int mt = wave * MW + i;
          const int row = mt / TPR, col = (mt - row * TPR) * 16;
          d[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(patch[(cb * PH + 2 * row + ky) * PW + 2 * (col + lm) + lk], bf[s][0], d[i], 0, 0, 0);
        }
      }
      if (lm < CS) {  // lane: 4 consecutive columns of small channel lm
#pragma unroll
        for (int i = 0; i < MW; ++i) {
          const int mt = wave * MW + i;
          const int row = mt / TPR, col = (mt - row * TPR) * 16 + 4 * lk;
          f32x4 o = d[i];
          if constexpr (DACT) {
            const float* sv = &simg[lm * LDS_S + row * TW + col];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] *= elu_grad_from_out(sv[r]);
          }
          *reinterpret_cast<f32x4*>(a.dout + (long)b * a.d_bs + ((long)lm * Hs + m0 + row) * Ws + n0 + col) = o;
        }
      }
    } else {
      f32x4 d[MW][2];  // [m-tile][py]
#pragma unroll
      for (int i = 0; i < MW; ++i) { d[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; d[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int k = 4 * s + lk;
        const int cs = k / 6, r = k - cs * 6;
        const int dyi = r / 3, dxp = r - dyi * 3;
        const int koff = (cs * SPH + dyi) * SPW + dxp;  // haloed small image: row + py + dyi, column + dxp
#pragma unroll
        for (int i = 0; i < MW; ++i) {
          const int mt = wave * MW + i;
          const int row = mt / TPR, col = (mt - row * TPR) * 16;
          const float* ap = &simg[koff + row * SPW + col + lm];
#pragma unroll
          for (int py = 0; py < 2; ++py) d[i][py] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[py * SPW], bf[s][py], d[i][py], 0, 0, 0);
        }
      }
      {
        const int px = lm >> 3, co = lm & 7;
#pragma unroll
        for (int i = 0; i < MW; ++i) {
          const int mt = wave * MW + i;
          const int row = mt / TPR, col = (mt - row * TPR) * 16 + 4 * lk;
#pragma unroll
          for (int py = 0; py < 2; ++py) {
            float* o = &otile[(co * 2 * TH + 2 * row + py) * OW + 2 * col + px];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[2 * r] = d[i][py][r];
          }
        }
      }
      __syncthreads();
      TB* ob = reinterpret_cast<TB*>(a.dout) + (long)b * a.d_bs;
      for (int i = t; i < CB * 2 * TH * OW / 4; i += 256) {
        const int e = 4 * i;
        const int co = e / (2 * TH * OW), r = e - co * (2 * TH * OW);
        const int oy = r / OW, ox = r - oy * OW;
        f32x4 v = *reinterpret_cast<const f32x4*>(&otile[e]);
        if constexpr (DACT) {
          const float* xv = &patch[(co * PH + 1 + oy) * PW + 1 + ox];  // the saved input at the same place
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] *= elu_grad_from_out(xv[q]);
        }
        Elem<TB>::st4(ob + ((long)co * Hb + 2 * m0 + oy) * Wb + 2 * n0 + ox, v);
      }
    }
  }
  // ---- the four wavefronts' weight-gradient images -> one slab (fixed order), bias partials behind it
  __syncthreads();
  float* out = a.partial + (size_t)blockIdx.x * SLAB;
  {
    float* comb = smem + wave * CS * CPITCH;
#pragma unroll
    for (int j = 0; j < CB; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * lk + r;
        if (row < CS) comb[row * CPITCH + 16 * j + lm] = acc[j][r];
      }
    __syncthreads();
    for (int i = t; i < NW; i += 256) {
      const int m = i / (CB * 16), n = i - m * (CB * 16);
      const float* c0 = smem + m * CPITCH + n;
      out[i] = (c0[0] + c0[CS * CPITCH]) + (c0[2 * CS * CPITCH] + c0[3 * CS * CPITCH]);
    }
  }
  __syncthreads();
  float* bred = smem;  // [16 channels][4 wavefronts]
  constexpr int nch = CONV ? CS : CB;
  for (int c = 0; c < nch; ++c) {
    float v = 0.f;
    if constexpr (CONV) {
#pragma unroll
      for (int q = 0; q < NQS; ++q) {
        const int i = t + 256 * q;
        if (i < CS * SROWS * (TW / 4) && (i / (TW / 4)) / SROWS == c) v += bsum[q];
      }
    } else {
#pragma unroll
      for (int q = 0; q < NQB; ++q) {
        const int i = t + 256 * q;
        if (i < CB * PH * (2 * TW / 4) && (i / (2 * TW / 4)) / PH == c) v += bsum[q];
      }
    }
    v = wave_sum(v);
    if (lane == 0) bred[c * 4 + wave] = v;
  }
  __syncthreads();
  if (t < BPAD) out[NW + t] = t < nch ? (bred[t * 4] + bred[t * 4 + 1]) + (bred[t * 4 + 2] + bred[t * 4 + 3]) : 0.f;
}

bool conv2d_bwd_lds_supported(int Cs, int Cb, int Hs, int Ws) {
  return !sched(LSHM_SCHED_NO_BWD_LDS2D) && Cs == CS && Cb == CB && Hs % TH == 0 && Ws % TW == 0;
}

// conv != 0: conv layer (small = dz, big = saved input, dout = dbig); else transposed layer (small = saved input, big = dz,
// dout = dsmall).  dact != 0: dout is multiplied by ELU' of the saved tensor of its own shape.
int conv2d_bwd_lds(const float* small, long s_bs, const float* big, long big_bs, const float* w, float* dout, int conv, int dact,
                   float* dw, float* db, int B, int Cs, int Cb, int Hs, int Ws, float* ws, size_t wsf, int accumulate,
                   hipStream_t st, GradJobs* defer, int big_bf16) {
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (!conv2d_bwd_lds_supported(Cs, Cb, Hs, Ws) || !small || !big || !w || !dout || !dw || s_bs % 4 || big_bs % 4 || !al16(small) ||
      !al16(big) || !al16(dout)) {
    set_last_error("conv2d_bwd_lds: unsupported shape, stride or alignment");
    return LSHM_ERR_UNSUPPORTED;
  }
  if (wsf < conv2d_wgrad_direct_workspace_floats(Cs, Cb)) { set_last_error("conv2d_bwd_lds: workspace too small"); return LSHM_ERR_WORKSPACE; }
  Bwd2dArgs a;
  a.small = small; a.s_bs = s_bs; a.big = big; a.big_bs = big_bs; a.w = w;
  a.dout = dout; a.d_bs = conv ? big_bs : s_bs;
  a.partial = ws;
  a.Hs = Hs; a.Ws = Ws; a.ntiles = (Ws / TW) * (Hs / TH) * B;
  constexpr int cap = 512;  // persistent workgroups (two per CU)
  const int grid = a.ntiles < cap ? a.ntiles : cap;
  int rc;
#define LSHM_B2D(CONV_, DACT_, TB_) do { \
    if ((rc = kernel_budget_ok(reinterpret_cast<const void*>(&conv2d_bwd_lds_kernel<CONV_, DACT_, TB_>), 256, 0, "conv2d_bwd_lds"))) return rc; \
    hipLaunchKernelGGL((conv2d_bwd_lds_kernel<CONV_, DACT_, TB_>), dim3(grid), dim3(256), 0, st, a); } while (0)
  if (big_bf16) {
    if (conv) { if (dact) LSHM_B2D(true, true, bf16); else LSHM_B2D(true, false, bf16); }
    else { if (dact) LSHM_B2D(false, true, bf16); else LSHM_B2D(false, false, bf16); }
  } else {
    if (conv) { if (dact) LSHM_B2D(true, true, float); else LSHM_B2D(true, false, float); }
    else { if (dact) LSHM_B2D(false, true, float); else LSHM_B2D(false, false, float); }
  }
#undef LSHM_B2D
  if ((rc = check_launch("conv2d_bwd_lds"))) return rc;
  const int nbias = conv ? Cs : Cb;
  if (defer) {
    defer->sums.push_back(SumJob{ws, dw, SLAB, NW, grid, 0, 0, 0, 0, accumulate, 0});
    if (db) defer->sums.push_back(SumJob{ws + NW, db, SLAB, nbias, grid, 0, 0, 0, 0, accumulate, 0});
    return LSHM_OK;
  }
  rc = reduce_partials_strided(ws, SLAB, dw, NW, grid, accumulate, st);
  if (rc || !db) return rc;
  return reduce_partials_strided(ws + NW, SLAB, db, nbias, grid, accumulate, st);
}

}  // namespace lshm
