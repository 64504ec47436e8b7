// One-pass backward of the outer 1-D layers (k4 s4; src/lofar_models.py:115-117 conv0 / conv1 and :140-142
// tconv4 / tconv5 of AutoEncoder1DCNN): weight, bias AND data gradient from ONE read of the layer's output
// gradient and of its saved input.  The two-kernel form reads both tensors twice (once for the weight gradient,
// once for the data gradient); these layers carry 76 % of the 1-D bytes and are HBM-bound.
//
// Geometry shared with conv1d_wgrad_stream_kernel (conv1d_stream.hip): `small` is the short tensor (B, CS, Ls),
// `big` the long one (B, CB, 4 Ls); lane (slot, q) of a wavefront owns 4 consecutive positions of a 64-position
// tile, the small channels {4a + q} and the big channels {4g + q}; the weight gradient runs on
// v_mfma_f32_4x4x1_16b_f32 straight from those registers.  The data gradient re-uses the same registers:
//   transposed layer (pad 0; small = saved input, big = dz):
//       dsmall[cs, j] = ELU'(small[cs, j]) * sum_{cb,t} w[cs, cb, t] big[cb, 4j + t]
//     a lane sums over ITS big channels for every cs, the four lanes of a quad reduce-scatter the partial sums
//     (two DPP quad permutations) and each ends with exactly the channels it loaded;
//   conv layer (pad 1; small = dz, big = saved input):
//       dbig[cb, 4j - 1 + t] = ELU'(big[..]) * sum_cs small[cs, j] w[cs, cb, t]
//     the quad all-gathers the small channels (DPP broadcasts), a lane then forms ITS big channels; the aligned
//     output quad 4j .. 4j+3 takes taps 1..3 of position j and tap 0 of position j + 1, so a lane also loads the
//     small values of the position after its four.
// `big` is read as aligned quads plus, for pad 1, the one element before them; the pad-1 windows of the weight
// gradient are register selections (no shifted loads).  Weights of the data gradient: 8 x float4 per lane for the
// 8 / 4 layers (registers), 24 for 12 / 8 (LDS, one broadcast ds_read_b128 per use).
#include <stdlib.h>

#include "kernels.h"

namespace lshm {

struct FusedBwd1dArgs {
  const float* small[2];
  const void* big[2];
  const float* w[2];       // [CS][CB][4] (Conv1d: (Cout, Cin, 4); ConvTranspose1d: (Cin, Cout, 4))
  float* partial[2];       // one slab of CS*CB*4 + 16 floats per workgroup
  void* dout[2];           // conv layer: gradient w.r.t. big (element type of big); transposed: w.r.t. small (element type of small)
  long s_bs, big_bs, d_bs;
  int Ls, Lb, ntiles;
};

template <int CTRL>
__device__ __forceinline__ float quad_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// CONV: conv layer (pad 1, bias gradient = sum of small) / transposed layer (pad 0, bias gradient = sum of big)
// DACT: multiply the data gradient by ELU'(saved input) (the saved input is the layer's input: small for the
//       transposed layer, big for the conv layer)
// two wavefronts per SIMD (at most 256 registers, accumulators included): with one, nothing hides the latency of a
// tile's loads but the next tile's prefetch; the 12 / 8 kernels keep 96 accumulator registers and therefore do
// NOT hold a second tile in registers (PF == false) -- the other wavefront of the SIMD covers the wait instead
template <int CS, int CB, bool CONV, bool DACT, bool WLDS, class TB, class TS = float>  // TB / TS: element types of `big` / `small`
__global__ __launch_bounds__(256, 2) void conv1d_bwd_fused2_kernel(const FusedBwd1dArgs a) {
  constexpr bool PF = CS <= 8;
  constexpr int GA = CS / 4, GB = CB / 4;
  constexpr int NW = CS * CB * 4, SLAB = NW + 16;
  constexpr int WSTR = CS * GB + 1;  // float4 units; (4 WSTR) mod 64 banks spreads the four q rows
  static_assert(CS % 4 == 0 && CB % 4 == 0, "channel groups of four");
  __shared__ float comb[4][SLAB];
  __shared__ f32x4 wl[WLDS ? 4 * WSTR : 1];
  const int pr = blockIdx.y;
  const TS* __restrict__ small = reinterpret_cast<const TS*>(a.small[pr]);
  const TB* __restrict__ big = reinterpret_cast<const TB*>(a.big[pr]);
  const float* __restrict__ w = a.w[pr];
  float* __restrict__ partial = a.partial[pr];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int slot = lane >> 2, q = lane & 3;
  const int Ls = a.Ls, Lb = a.Lb;

  // w[cs][cb = 4g + q][0..3] of this lane
  f32x4 wr[WLDS ? 1 : CS * GB];
  if constexpr (WLDS) {
    for (int i = t; i < CS * CB; i += 256) {
      const int cs = i / CB, cb = i - cs * CB;
      wl[(cb & 3) * WSTR + cs * GB + (cb >> 2)] = *reinterpret_cast<const f32x4*>(w + 4 * i);
    }
    __syncthreads();
  } else {
#pragma unroll
    for (int cs = 0; cs < CS; ++cs)
#pragma unroll
      for (int g = 0; g < GB; ++g) wr[cs * GB + g] = *reinterpret_cast<const f32x4*>(w + ((long)cs * CB + 4 * g + q) * 4);
  }
  auto W = [&](int cs, int g) -> f32x4 {
    if constexpr (WLDS) return wl[q * WSTR + cs * GB + g];
    else return wr[cs * GB + g];
  };

  f32x4 acc[GA][GB][4];
#pragma unroll
  for (int x = 0; x < GA; ++x)
#pragma unroll
    for (int g = 0; g < GB; ++g)
#pragma unroll
      for (int tp = 0; tp < 4; ++tp) acc[x][g][tp] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int NBS = CONV ? GA : GB;
  float bsum[NBS];
#pragma unroll
  for (int i = 0; i < NBS; ++i) bsum[i] = 0.f;

  const int tiles_per = Ls / 64;
  const int nwaves = gridDim.x * 4, w0 = blockIdx.x * 4 + wave;
  f32x4 ra[GA], rq[GB][4];
  float rx[GA], rm[GB];
  auto load_tile = [&](int tile) {
    const int b = tile / tiles_per, p0 = (tile - b * tiles_per) * 64 + 4 * slot;
    const TS* sb = small + (long)b * a.s_bs + p0;
    const TB* bb = big + (long)b * a.big_bs + 4L * p0;
#pragma unroll
    for (int x = 0; x < GA; ++x) {
      ra[x] = Elem<TS>::ld4(sb + (long)(4 * x + q) * Ls);
      if constexpr (CONV) rx[x] = p0 + 4 < Ls ? Elem<TS>::ld(sb + (long)(4 * x + q) * Ls + 4) : 0.f;
    }
#pragma unroll
    for (int g = 0; g < GB; ++g) {
#pragma unroll
      for (int i = 0; i < 4; ++i) rq[g][i] = Elem<TB>::ld4(bb + (long)(4 * g + q) * Lb + 4 * i);
      if constexpr (CONV) rm[g] = p0 > 0 ? Elem<TB>::ld(bb + (long)(4 * g + q) * Lb - 1) : 0.f;
    }
  };
  if (PF && w0 < a.ntiles) load_tile(w0);
  for (int tile = w0; tile < a.ntiles; tile += nwaves) {
    if (!PF) load_tile(tile);
    f32x4 ca[GA], cq[GB][4];
    float cx[GA], cm[GB];
#pragma unroll
    for (int x = 0; x < GA; ++x) { ca[x] = ra[x]; if constexpr (CONV) cx[x] = rx[x]; }
#pragma unroll
    for (int g = 0; g < GB; ++g) {
#pragma unroll
      for (int i = 0; i < 4; ++i) cq[g][i] = rq[g][i];
      if constexpr (CONV) cm[g] = rm[g];
    }
    if (PF && tile + nwaves < a.ntiles) load_tile(tile + nwaves);
    if constexpr (CONV) {
#pragma unroll
      for (int x = 0; x < GA; ++x) bsum[x] += (ca[x][0] + ca[x][1]) + (ca[x][2] + ca[x][3]);
    } else {
#pragma unroll
      for (int g = 0; g < GB; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) bsum[g] += (cq[g][i][0] + cq[g][i][1]) + (cq[g][i][2] + cq[g][i][3]);
    }
    // ---- weight gradient: 16 slots x (4 small channels x 4 big channels) per instruction
#pragma unroll
    for (int st = 0; st < 4; ++st)      // position p0 + st
#pragma unroll
      for (int tp = 0; tp < 4; ++tp)    // tap: big[4 (p0 + st) - pad + tp]
#pragma unroll
        for (int x = 0; x < GA; ++x)
#pragma unroll
          for (int g = 0; g < GB; ++g) {
            float bv;
            if constexpr (CONV) bv = tp == 0 ? (st == 0 ? cm[g] : cq[g][st - 1][3]) : cq[g][st][tp - 1];
            else bv = cq[g][st][tp];
            acc[x][g][tp] = __builtin_amdgcn_mfma_f32_4x4x1f32(ca[x][st], bv, acc[x][g][tp], 0, 0, 0);
          }
    const int b = tile / tiles_per, p0 = (tile - b * tiles_per) * 64 + 4 * slot;
    // (position-outer loops below: short live ranges keep the 12 / 8 kernels inside 256 registers)
    if constexpr (!CONV) {
      // ---- data gradient of the transposed layer: partial sums over this lane's big channels, reduce-scatter
      const bool odd = q & 1, hi = q & 2;
      f32x4 r2[GA];  // channel 4m + q, this lane's 4 positions
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        // (LDS weights: re-read per position -- 24 broadcast ds_read_b128 -- instead of 96 registers held across the loop)
        if constexpr (WLDS) asm volatile("" ::: "memory");
        float pc[CS];
#pragma unroll
        for (int cs = 0; cs < CS; ++cs) {
          float s = 0.f;
#pragma unroll
          for (int g = 0; g < GB; ++g) {
            const f32x4 wv = W(cs, g);
            s = fmaf(wv[3], cq[g][st][3], fmaf(wv[2], cq[g][st][2], fmaf(wv[1], cq[g][st][1], fmaf(wv[0], cq[g][st][0], s))));
          }
          pc[cs] = s;
        }
        float r1[CS / 2];  // after the pair exchange: channel 2k + (q & 1), summed over two lanes
#pragma unroll
        for (int k = 0; k < CS / 2; ++k) {
          const float send = odd ? pc[2 * k] : pc[2 * k + 1], keep = odd ? pc[2 * k + 1] : pc[2 * k];
          r1[k] = keep + quad_dpp<0xB1>(send);  // quad_perm [1,0,3,2]
        }
#pragma unroll
        for (int m = 0; m < GA; ++m) {
          const float send = hi ? r1[2 * m] : r1[2 * m + 1], keep = hi ? r1[2 * m + 1] : r1[2 * m];
          r2[m][st] = keep + quad_dpp<0x4E>(send);  // quad_perm [2,3,0,1]
        }
      }
      TS* dsm = reinterpret_cast<TS*>(a.dout[pr]) + (long)b * a.d_bs + p0;
#pragma unroll
      for (int m = 0; m < GA; ++m) {
        f32x4 o = r2[m];
        if constexpr (DACT) {
#pragma unroll
          for (int st = 0; st < 4; ++st) o[st] *= elu_grad_from_out(ca[m][st]);
        }
        Elem<TS>::st4(dsm + (long)(4 * m + q) * Ls, o);
      }
    } else {
      // ---- data gradient of the conv layer: all-gather the small channels of one position over the quad, then this
      // lane's big channels; the values of the next position (tap 0 of the aligned quad's last element) carry over
      auto gather = [&](int i, float (&dst)[CS]) {
#pragma unroll
        for (int x = 0; x < GA; ++x) {
          const float v = i < 4 ? ca[x][i < 4 ? i : 0] : cx[x];
          dst[4 * x + 0] = quad_dpp<0x00>(v);
          dst[4 * x + 1] = quad_dpp<0x55>(v);
          dst[4 * x + 2] = quad_dpp<0xAA>(v);
          dst[4 * x + 3] = quad_dpp<0xFF>(v);
        }
      };
      TB* dbg = reinterpret_cast<TB*>(a.dout[pr]) + (long)b * a.d_bs + 4L * p0;
      float cur[CS], nxt[CS];
      gather(0, cur);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (WLDS) asm volatile("" ::: "memory");  // as above: the weights come from LDS again for every position
        gather(i + 1, nxt);
#pragma unroll
        for (int g = 0; g < GB; ++g) {
          f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int cs = 0; cs < CS; ++cs) {
            const f32x4 wv = W(cs, g);
            // big[4p + r] (r < 3) is tap r + 1 of position p; big[4p + 3] is tap 0 of position p + 1
            o[0] = fmaf(cur[cs], wv[1], o[0]);
            o[1] = fmaf(cur[cs], wv[2], o[1]);
            o[2] = fmaf(cur[cs], wv[3], o[2]);
            o[3] = fmaf(nxt[cs], wv[0], o[3]);
          }
          if constexpr (DACT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] *= elu_grad_from_out(cq[g][i][r]);
          }
          Elem<TB>::st4(dbg + (long)(4 * g + q) * Lb + 4 * i, o);
        }
#pragma unroll
        for (int cs = 0; cs < CS; ++cs) cur[cs] = nxt[cs];
      }
    }
  }
  // ---- 16 slots -> lane q of slot 0 (butterflies over lane bits 2..5), then the 4 waves through LDS
#pragma unroll
  for (int x = 0; x < GA; ++x)
#pragma unroll
    for (int g = 0; g < GB; ++g)
#pragma unroll
      for (int tp = 0; tp < 4; ++tp)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[x][g][tp][r];
          v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
          // lane (slot 0, q): dW[cs = 4x + r][cb = 4g + q][tap tp]
          if (slot == 0) comb[wave][((4 * x + r) * CB + 4 * g + q) * 4 + tp] = v;
        }
#pragma unroll
  for (int i = 0; i < NBS; ++i) {
    float v = bsum[i];
    v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
    if (slot == 0) comb[wave][NW + 4 * i + q] = v;  // channel 4i + q of small (conv) or big (transposed)
  }
  __syncthreads();
  float* out = partial + (size_t)blockIdx.x * SLAB;
  for (int i = t; i < SLAB; i += 256) {
    const bool live = i < NW || i - NW < 4 * NBS;
    out[i] = live ? (comb[0][i] + comb[1][i]) + (comb[2][i] + comb[3][i]) : 0.f;
  }
}

bool conv1d_bwd_lds_supported(int Cs, int Cb, int Ls, int pad);
int conv1d_bwd_lds(const float* small, const float* small2, long s_bs, const float* big, const float* big2, long big_bs, float* ws,
                   float* ws2, int B, int Cs, int Cb, int Ls, int Lb, int pad, int max_blocks, hipStream_t st, int* grid_out,
                   const FusedDgrad& fd, int big_bf16, int small_bf16);
bool conv1d_bwd_fused2_supported(int Cs, int Cb, int pad) {
  // 12 / 8 channels: only as the LDS-staged form below (conv1d_bwd_lds_kernel).  In registers 96 accumulators + the
  // data-gradient working set do not fit two wavefronts per SIMD (80-330 spilled registers, 67-129 us against 50 us for the two
  // separate kernels, profiles/r03/fused_bwd_probe.txt): that form was removed in round 4.
  const bool wide = !sched(LSHM_SCHED_NO_BWD_LDS);
  return (wide && Cs == 12 && Cb == 8 && (pad == 0 || pad == 1)) || (Cs == 8 && Cb == 4 && (pad == 0 || pad == 1));
}

// one slab of Cs*Cb*4 + 16 floats per workgroup at ws (and ws2 for the second problem); returns the grid size
int conv1d_bwd_fused2(const float* small, const float* small2, long s_bs, const float* big, const float* big2, long big_bs,
                      float* ws, float* ws2, int B, int Cs, int Cb, int Ls, int Lb, int pad, int max_blocks,
                      hipStream_t st, int* grid_out, int big_bf16, const FusedDgrad& fd, int small_bf16) {
  // (8 / 4 channels with bf16 storage stay on the register form below: the LDS form is 0.05 ms per iteration SLOWER there --
  //  half the bytes, the same matrix and LDS work -- while it is 0.03 ms faster in fp32)
  if (((Cs == 12 && !small_bf16) || (Cs == 8 && small_bf16 == big_bf16 && !big_bf16)) &&
      conv1d_bwd_lds_supported(Cs, Cb, Ls, pad))
    return conv1d_bwd_lds(small, small2, s_bs, big, big2, big_bs, ws, ws2, B, Cs, Cb, Ls, Lb, pad, max_blocks, st, grid_out, fd, big_bf16,
                          small_bf16);
  const bool two = small2 != nullptr;
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (!conv1d_bwd_fused2_supported(Cs, Cb, pad) || Ls % 64 || Lb != 4 * Ls || s_bs % 4 || big_bs % 4 || fd.dx_bs % 4 ||
      !fd.w || !fd.dx || !al16(small) || !al16(big) || !al16(fd.w) || !al16(fd.dx) ||
      (two && (!big2 || !fd.w2 || !fd.dx2 || !al16(small2) || !al16(big2) || !al16(fd.w2) || !al16(fd.dx2))) ||
      (big_bf16 && !(Cs == 8 && Cb == 4)) || (small_bf16 && !big_bf16)) {
    set_last_error("conv1d_bwd_fused: unsupported layer shape, stride or alignment");
    return LSHM_ERR_UNSUPPORTED;
  }
  FusedBwd1dArgs a;
  a.small[0] = small; a.small[1] = two ? small2 : small;
  a.big[0] = big; a.big[1] = two ? big2 : big;
  a.w[0] = fd.w; a.w[1] = two ? fd.w2 : fd.w;
  a.partial[0] = ws; a.partial[1] = two ? ws2 : ws;
  a.dout[0] = fd.dx; a.dout[1] = two ? fd.dx2 : fd.dx;
  a.s_bs = s_bs; a.big_bs = big_bs; a.d_bs = fd.dx_bs;
  a.Ls = Ls; a.Lb = Lb; a.ntiles = (Ls / 64) * B;
  constexpr int tpw = 4;
  int grid = a.ntiles / (4 * tpw);  // tiles per wavefront
  if (grid > 512) grid = 512;
  if (grid < 1) grid = 1;
  if (grid > max_blocks) grid = max_blocks;
  *grid_out = grid;
  const dim3 g(grid, two ? 2 : 1);
  const bool dact = fd.dact != 0;
#define LSHM_FUSED2(CS, CB, CONV, WL, T, TS_)                                                                          \
  do {                                                                                                                 \
    if (dact) hipLaunchKernelGGL((conv1d_bwd_fused2_kernel<CS, CB, CONV, true, WL, T, TS_>), g, dim3(256), 0, st, a);  \
    else hipLaunchKernelGGL((conv1d_bwd_fused2_kernel<CS, CB, CONV, false, WL, T, TS_>), g, dim3(256), 0, st, a);      \
  } while (0)
  if (Cs == 12) {  // (the register form of the 12 / 8 layers spilled 80-330 registers and was removed: LDS form or nothing)
    set_last_error("conv1d_bwd_fused: the 12 / 8 channel layers need a length the LDS-staged kernel takes (a multiple of 128)");
    return LSHM_ERR_UNSUPPORTED;
  }
  if (pad == 0 && big_bf16 && small_bf16) LSHM_FUSED2(8, 4, false, false, bf16, bf16);
  else if (pad == 0 && big_bf16) LSHM_FUSED2(8, 4, false, false, bf16, float);
  else if (pad == 0) LSHM_FUSED2(8, 4, false, false, float, float);
  else if (big_bf16 && small_bf16) LSHM_FUSED2(8, 4, true, false, bf16, bf16);
  else if (big_bf16) LSHM_FUSED2(8, 4, true, false, bf16, float);
  else LSHM_FUSED2(8, 4, true, false, float, float);
#undef LSHM_FUSED2
  return check_launch("conv1d_bwd_fused");
}

}  // namespace lshm

namespace lshm {

// ----------------------------------------------------------------------------------------------
// One-pass backward of the 12 <-> 8 channel 1-D layers (tconv4: src/lofar_models.py:141, conv1: :116), LDS-staged:
// the register form above keeps 96 accumulator registers for the 4x4x1 weight gradient and spills; here a tile of TP
// small positions and the matching big segment are staged in LDS once (float4 loads, next tile in flight in
// registers: the layout of conv1d_wgrad_mid_kernel) and BOTH products run on v_mfma_f32_16x16x4_f32 from it:
//   weight gradient   dW[cs][(cb,t)] += small[cs][p] big[cb][4p - pad + t]        M = cs, N = 32, K = positions
//   transposed layer  dsmall[p][cs]   = ELU'(small) * sum_(cb,t) big[cb][4p + t] w[cs][cb][t]     M = p, N = cs, K = 32
//   conv layer        dbig[p][(cb,t)] = ELU'(big)   * sum_cs small[cs][p] w[cs][cb][t]            M = p, N = 32, K = 12
// (a handful of accumulator registers; the weights of the data gradient are 6-8 registers per lane).  The conv
// layer's output window of position p is big[4p - 1 .. 4p + 2]: results go through an LDS image at offset +1, so the
// tile's aligned output quads need tap 0 of the position AFTER the tile -- one extra small column, a 12-term sum per
// big channel.
// ----------------------------------------------------------------------------------------------
struct BwdLds1dArgs {
  const float* small[2];
  const float* big[2];
  const float* w[2];
  float* partial[2];
  float* dout[2];
  long s_bs, big_bs, d_bs;
  int Ls, Lb, ntiles;
};

// TB: element type of the big tensor and (conv layer) of its gradient -- fp32, or bf16 storage
template <int CS, int CB, int TP, bool CONV, bool DACT, typename TB, typename TS = float>  // TS: element type of the small tensor (and of its gradient)
__global__ __launch_bounds__(256) void conv1d_bwd_lds_kernel(const BwdLds1dArgs a) {
  static_assert(CS <= 16 && CS % 4 == 0 && CB % 4 == 0 && TP % 64 == 0, "one 16-row tile of small channels");
  constexpr int NT = CB / 4, NW = CS * CB * 4, SLAB = NW + 16;
  constexpr int pad = CONV ? 1 : 0;
  constexpr int LDS_S = TP + 2;   // small row pitch (== 2 mod 32); column TP holds the position after the tile (CONV)
  constexpr int BP = 4 * TP + 8;  // big row pitch (== 8 mod 32); bimg[cb][1 + i] = big[cb][4 j0 + i], [0] = element before
  constexpr int CPITCH = CB * 4 + 4;
  constexpr int OB = CONV ? CB * BP : 0;
  __shared__ __attribute__((aligned(16))) float smem[16 * LDS_S + CB * BP + OB];
  float* stile = smem;
  float* bimg = smem + 16 * LDS_S;
  float* obuf = bimg + CB * BP;  // CONV: data-gradient image, logical element i of the tile at obuf[cb][1 + i]
  const int pr = blockIdx.y;
  const TS* __restrict__ small = reinterpret_cast<const TS*>(a.small[pr]);
  const TB* __restrict__ big = reinterpret_cast<const TB*>(a.big[pr]);
  const float* __restrict__ w = a.w[pr];
  float* __restrict__ dout = a.dout[pr];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lm = lane & 15, lk = lane >> 4;
  const int Ls = a.Ls, Lb = a.Lb;
  for (int i = t; i < 16 * LDS_S; i += 256) stile[i] = 0.f;  // rows >= CS stay zero
  // data-gradient weights of this lane
  float wd[CONV ? (CS / 4) * NT : CB];
  if constexpr (CONV) {
#pragma unroll
    for (int s = 0; s < CS / 4; ++s)
#pragma unroll
      for (int j = 0; j < NT; ++j) wd[s * NT + j] = w[(long)(4 * s + lk) * CB * 4 + 16 * j + lm];  // B[k = cs][n = (cb, t)]
  } else {
#pragma unroll
    for (int s = 0; s < CB; ++s) wd[s] = lm < CS ? w[((long)lm * CB + s) * 4 + lk] : 0.f;  // B[k = (cb = s, t = lk)][n = cs]
  }
  f32x4 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int NQS = (CS * (TP / 4) + 255) / 256, NQB = (CB * TP + 255) / 256;
  float bs_small[NQS], bs_big[NQB];
#pragma unroll
  for (int q = 0; q < NQS; ++q) bs_small[q] = 0.f;
#pragma unroll
  for (int q = 0; q < NQB; ++q) bs_big[q] = 0.f;
  const int tiles_per = Ls / TP;
  const int bofs = (lm >> 2) * BP + (lm & 3) + 1 - pad;  // weight-gradient B fragment: big channel 4 j + lm / 4, tap lm % 4
  f32x4 rs[NQS], rb[NQB];
  float rh = 0.f, rx = 0.f;
  auto fetch = [&](int tile) {
    const int b = tile / tiles_per, j0 = (tile - b * tiles_per) * TP;
    const TS* sb = small + (long)b * a.s_bs + j0;
    const TB* bb = big + (long)b * a.big_bs + 4L * j0;
#pragma unroll
    for (int q = 0; q < NQS; ++q) {
      const int i = t + 256 * q;
      rs[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (i < CS * (TP / 4)) rs[q] = Elem<TS>::ld4(sb + (long)(i / (TP / 4)) * Ls + 4 * (i % (TP / 4)));
    }
#pragma unroll
    for (int q = 0; q < NQB; ++q) {
      const int i = t + 256 * q;
      rb[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (i < CB * TP) rb[q] = Elem<TB>::ld4(bb + (long)(i / TP) * Lb + 4 * (i % TP));
    }
    rh = (t < CB && pad && j0 > 0) ? Elem<TB>::ld(bb + (long)t * Lb - 1) : 0.f;                      // the element before the segment
    if constexpr (CONV) rx = (t >= 64 && t < 64 + CS && j0 + TP < Ls) ? Elem<TS>::ld(sb + (long)(t - 64) * Ls + TP) : 0.f;  // the position after the tile
  };
  if ((int)blockIdx.x < a.ntiles) fetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int b = tile / tiles_per, j0 = (tile - b * tiles_per) * TP;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQS; ++q) {
      const int i = t + 256 * q;
      if (i < CS * (TP / 4)) {
        const f32x4 v = rs[q];
        float* d = &stile[(i / (TP / 4)) * LDS_S + 4 * (i % (TP / 4))];
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
        bs_small[q] += (v[0] + v[1]) + (v[2] + v[3]);
      }
    }
#pragma unroll
    for (int q = 0; q < NQB; ++q) {
      const int i = t + 256 * q;
      if (i < CB * TP) {
        const f32x4 v = rb[q];
        float* d = &bimg[(i / TP) * BP + 1 + 4 * (i % TP)];
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
        bs_big[q] += (v[0] + v[1]) + (v[2] + v[3]);
      }
    }
    if (t < CB) bimg[t * BP] = rh;
    if constexpr (CONV) { if (t >= 64 && t < 64 + CS) stile[(t - 64) * LDS_S + TP] = rx; }
    __syncthreads();
    if (tile + (int)gridDim.x < a.ntiles) fetch(tile + gridDim.x);
    // ---- weight gradient: groups of 4 positions, every 4th group per wavefront
#pragma unroll 4
    for (int it = 0; it < TP / 16; ++it) {  // (a constant trip count: `s = wave; s < TP / 4; s += 4` is refused by the unroller)
      const int s = wave + 4 * it;
      const int p = 4 * s + lk;
      const float av = stile[lm * LDS_S + p];
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bimg[4 * j * BP + bofs + 4 * p], acc[j], 0, 0, 0);
    }
    // ---- data gradient: 16-position row tiles, every 4th per wavefront
    if constexpr (!CONV) {
      TS* dsm = reinterpret_cast<TS*>(dout) + (long)b * a.d_bs + j0;
      for (int mt = wave; mt < TP / 16; mt += 4) {
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < CB; ++s) d = __builtin_amdgcn_mfma_f32_16x16x4f32(bimg[s * BP + 1 + 4 * (16 * mt + lm) + lk], wd[s], d, 0, 0, 0);
        if (lm < CS) {  // lane: positions 16 mt + 4 lk .. + 3 of small channel lm
          const int p = 16 * mt + 4 * lk;
          if constexpr (DACT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) d[r] *= elu_grad_from_out(stile[lm * LDS_S + p + r]);
          }
          Elem<TS>::st4(dsm + (long)lm * Ls + p, d);
        }
      }
    } else {
      for (int mt = wave; mt < TP / 16; mt += 4) {
        f32x4 d[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) d[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < CS / 4; ++s) {
          const float av = stile[(4 * s + lk) * LDS_S + 16 * mt + lm];
#pragma unroll
          for (int j = 0; j < NT; ++j) d[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wd[s * NT + j], d[j], 0, 0, 0);
        }
        // lane: positions 16 mt + 4 lk + r, column (cb = 4 j + lm / 4, t = lm % 4) -> logical element 4 p - 1 + t, kept at + 1
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) obuf[(4 * j + (lm >> 2)) * BP + 4 * (16 * mt + 4 * lk + r) + (lm & 3)] = d[j][r];
      }
      __syncthreads();
      TB* dbg = reinterpret_cast<TB*>(dout) + (long)b * a.d_bs + 4L * j0;
      for (int i = t; i < CB * TP; i += 256) {
        const int cb = i / TP, c4 = i - cb * TP;
        const float* o = obuf + cb * BP + 1 + 4 * c4;
        f32x4 v = {o[0], o[1], o[2], o[3]};
        if (c4 == TP - 1) {  // the tile's last element: tap 0 of the position after the tile
          float s = 0.f;
#pragma unroll
          for (int cs = 0; cs < CS; ++cs) s = fmaf(stile[cs * LDS_S + TP], w[((long)cs * CB + cb) * 4], s);
          v[3] = s;
        }
        if constexpr (DACT) {
          const float* xb = bimg + cb * BP + 1 + 4 * c4;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] *= elu_grad_from_out(xb[r]);
        }
        Elem<TB>::st4(dbg + (long)cb * Lb + 4 * c4, v);
      }
    }
  }
  // ---- weight-gradient images of the four wavefronts -> one slab (fixed order), bias partials
  __syncthreads();
  float* comb = smem + wave * CS * CPITCH;
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * lk + r;
      if (row < CS) comb[row * CPITCH + 16 * j + lm] = acc[j][r];
    }
  __syncthreads();
  float* out = a.partial[pr] + (size_t)blockIdx.x * SLAB;
  for (int i = t; i < NW; i += 256) {
    const int m = i / (CB * 4), n = i - m * (CB * 4);
    const float* c0 = smem + m * CPITCH + n;
    out[i] = (c0[0] + c0[CS * CPITCH]) + (c0[2 * CS * CPITCH] + c0[3 * CS * CPITCH]);
  }
  __syncthreads();
  float* bred = bimg;  // [16 channels][4 waves]
  constexpr int nch = CONV ? CS : CB;
  for (int c = 0; c < nch; ++c) {
    float v = 0.f;
    if constexpr (CONV) {
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
  __syncthreads();
  if (t < 16) out[NW + t] = t < nch ? (bred[t * 4] + bred[t * 4 + 1]) + (bred[t * 4 + 2] + bred[t * 4 + 3]) : 0.f;
}

bool conv1d_bwd_lds_supported(int Cs, int Cb, int Ls, int pad) {
  if (sched(LSHM_SCHED_NO_BWD_LDS)) return false;
  if (Cs == 12 && Cb == 8) return Ls % 128 == 0 && (pad == 0 || pad == 1);
  // conv0 of the 1-D autoencoders (4 -> 8 channels, pad 1): the register form runs at 3.0 TB/s with two wavefronts per SIMD
  return Cs == 8 && Cb == 4 && pad == 1 && Ls % 256 == 0 && !sched(LSHM_SCHED_NO_BWD_LDS_8_4);
}

int conv1d_bwd_lds(const float* small, const float* small2, long s_bs, const float* big, const float* big2, long big_bs, float* ws,
                   float* ws2, int B, int Cs, int Cb, int Ls, int Lb, int pad, int max_blocks, hipStream_t st, int* grid_out,
                   const FusedDgrad& fd, int big_bf16, int small_bf16) {
  const bool two = small2 != nullptr;
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (!conv1d_bwd_lds_supported(Cs, Cb, Ls, pad) || Lb != 4 * Ls || s_bs % 4 || big_bs % 4 || fd.dx_bs % 4 || !fd.w || !fd.dx ||
      !al16(small) || !al16(big) || !al16(fd.dx) ||
      (two && (!big2 || !fd.w2 || !fd.dx2 || !al16(small2) || !al16(big2) || !al16(fd.dx2)))) {
    set_last_error("conv1d_bwd_lds: unsupported layer shape, stride or alignment");
    return LSHM_ERR_UNSUPPORTED;
  }
  BwdLds1dArgs a;
  a.small[0] = small; a.small[1] = two ? small2 : small;
  a.big[0] = big; a.big[1] = two ? big2 : big;
  a.w[0] = fd.w; a.w[1] = two ? fd.w2 : fd.w;
  a.partial[0] = ws; a.partial[1] = two ? ws2 : ws;
  a.dout[0] = fd.dx; a.dout[1] = two ? fd.dx2 : fd.dx;
  a.s_bs = s_bs; a.big_bs = big_bs; a.d_bs = fd.dx_bs;
  const int TP = Cs == 8 ? 256 : 128;
  a.Ls = Ls; a.Lb = Lb; a.ntiles = (Ls / TP) * B;
  constexpr int cap_env = 256;
  // 256 workgroups per problem: 128 / 256 / 384 / 512 / 768 are within 0.01 ms of each other in the step; fewer slabs for the closing sums
  int grid = a.ntiles < cap_env ? a.ntiles : cap_env;
  if (grid > max_blocks) grid = max_blocks;
  if (grid < 1) grid = 1;
  *grid_out = grid;
  const dim3 g(grid, two ? 2 : 1);
  const bool dact = fd.dact != 0;
#define LSHM_BLDS(CONV_, DACT_, TB_) hipLaunchKernelGGL((conv1d_bwd_lds_kernel<12, 8, 128, CONV_, DACT_, TB_>), g, dim3(256), 0, st, a)
  if (Cs == 8) {
    // bf16 storage: the layer's input (big, and its gradient) and its output gradient (small) are both bf16 tensors, or neither
    if (big_bf16 != small_bf16) { set_last_error("conv1d_bwd_lds: 8 / 4 channels: both tensors bf16 or both fp32"); return LSHM_ERR_UNSUPPORTED; }
    if (big_bf16) {
      if (dact) hipLaunchKernelGGL((conv1d_bwd_lds_kernel<8, 4, 256, true, true, bf16, bf16>), g, dim3(256), 0, st, a);
      else hipLaunchKernelGGL((conv1d_bwd_lds_kernel<8, 4, 256, true, false, bf16, bf16>), g, dim3(256), 0, st, a);
    } else {
      if (dact) hipLaunchKernelGGL((conv1d_bwd_lds_kernel<8, 4, 256, true, true, float>), g, dim3(256), 0, st, a);
      else hipLaunchKernelGGL((conv1d_bwd_lds_kernel<8, 4, 256, true, false, float>), g, dim3(256), 0, st, a);
    }
  } else if (small_bf16) {
    set_last_error("conv1d_bwd_lds: 12 / 8 channels: the small tensor is fp32");
    return LSHM_ERR_UNSUPPORTED;
  } else if (big_bf16) {
    if (pad == 0) { if (dact) LSHM_BLDS(false, true, bf16); else LSHM_BLDS(false, false, bf16); }
    else { if (dact) LSHM_BLDS(true, true, bf16); else LSHM_BLDS(true, false, bf16); }
  } else {
    if (pad == 0) { if (dact) LSHM_BLDS(false, true, float); else LSHM_BLDS(false, false, float); }
    else { if (dact) LSHM_BLDS(true, true, float); else LSHM_BLDS(true, false, float); }
  }
#undef LSHM_BLDS
  return check_launch("conv1d_bwd_lds");
}

}  // namespace lshm
