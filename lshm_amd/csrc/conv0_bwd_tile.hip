// Backward of conv0 of netT AND netF (Conv1d(4, 8, 4, stride=4, padding=1), src/lofar_models.py:115) and the gradient that
// enters the 2-D autoencoder's backward, in one pass over 32 x 64 image tiles:
//   dW_T[cs][cb][t] += dzT[cs][j] r[cb][row-seq 4j - 1 + t]        dW_F likewise on the column-vectorised sequence
//   db_T[cs]        += dzT[cs][j]
//   gx1 = gx1p - (dT + dF^T) / 2,   dT[cb][4j - 1 + t] = sum_cs dzT[cs][j] wT[cs][cb][t]     (src/kharmonic_lofar.py:142-147:
//                                                                   both networks read the residual (x - x1) / 2)
// The residual is the layer input of both networks: its row-vectorised sequence is the image, its column-vectorised one the
// image walked column-wise, and stride == kernel size, so every image element belongs to exactly ONE (position, tap) of each
// network.  A workgroup stages the residual tile once (from the row image alone: the transposed copy is never read), the
// 8 x 32 x 16 positions of netT and 8 x 64 x 8 positions of netF that cover it, and forms
//   * both weight gradients on v_mfma_f32_16x16x4_f32 (M = output channel, N = (input channel, tap), K = positions; the
//     window of a position is read row-wise for netT and column-wise for netF from the same LDS image),
//   * both data gradients, eight multiply-adds per element and network, combined with the reconstruction term straight
//     into gx1.
// This replaces conv1d_bwd_lds_kernel<8, 4, 256> on the pair (reads dz and both residual copies, writes dT and dF) and
// combine_dx1_kernel (reads gx1p, dT, dF, writes gx1): 616 MB -> 335 MB of traffic and one launch less.  The four column/row
// neighbours of a tile run on the same XCD (block -> tile map below), so the halo elements are L2 hits.
#include <stdlib.h>

#include "kernels.h"

namespace lshm {

namespace {
typedef const __attribute__((address_space(4))) f32x4* cf32x4_ptr;
__device__ __forceinline__ f32x4 uload4(const float* q) { return *(cf32x4_ptr)(q); }
constexpr int P = 128, CI = 4, CO = 8, L = P * P, LO = L / 4;
constexpr int TR = 32, TC = 64;          // tile rows / columns
constexpr int NT = 512;
constexpr int TPI = (P / TR) * (P / TC);  // tiles per image
constexpr int RP = TC + 2;               // res[ci][1 + r][1 + c]: row 0 = the row above the tile, column 0 = the column before it
constexpr int RCH = (TR + 1) * RP;
constexpr int ZTR = TC / 4 + 1;          // zT[cs][r][p]: 16 positions of a tile row + the one after them
constexpr int ZTP = TR * ZTR + 4;        // (== 4 mod 32: the eight channels of an A fragment fall into different banks)
constexpr int ZFG = TR / 4 + 1;          // zF[cs][g][c]: 8 positions of a tile column + the one below them
constexpr int ZFP = ZFG * TC + 4;
constexpr int NW = CO * CI * 4;          // 128 weights per network
constexpr int SLAB = 2 * (NW + 16);      // per workgroup: [netT weights, 16 bias slots][netF weights, 16 bias slots]
constexpr int MAX_GRID = 1024;
}  // namespace

struct Conv0BwdTileArgs {
  const float* r;      // (B, 4, 128 * 128): the residual as the image (netT's input)
  const float* dz[2];  // (B, 8, 4096), batch stride z_bs: gradients w.r.t. conv0's pre-activations, netT / netF
  const float* w[2];   // (8, 4, 4)
  const float* gx1p;   // (B, 4, 128, 128): the reconstruction terms' share of the gradient
  float* gx1;          // (B, 4, 128, 128)
  float* partial;      // gridDim.x slabs
  long z_bs;
  int B, ntiles;
};

// T: element type of the residual, of the two dz tensors and of gx1p / gx1 (bf16 storage, DESIGN 4.6: the two data gradients are
// then rounded to bf16 before they are combined, as the kernels this replaces stored them)
template <class T>
__device__ __forceinline__ float stored(float v) {
  if constexpr (sizeof(T) == 2) return (float)(T)v;
  return v;
}
template <class T>
__global__ __launch_bounds__(NT, 4) void conv0_bwd_tile_kernel(const Conv0BwdTileArgs a) {
  static_assert(NT == 512 && TR == 32 && TC == 64, "the thread maps below are written for 512 threads and 32 x 64 tiles");
  __shared__ __attribute__((aligned(16))) float res[CI * RCH];
  __shared__ __attribute__((aligned(16))) float zT[CO * ZTP];
  __shared__ __attribute__((aligned(16))) float zF[CO * ZFP];
  __shared__ float wfl[NW];  // netF's weights: a thread's rows all have the same (row + 1) mod 4 = tap, but lanes differ
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lm = lane & 15, lk = lane >> 4;
  const int c4 = t & 15, rq = t >> 4;  // this thread's output quads: row rq, columns 4 c4 .. 4 c4 + 3 of every channel
  const int tap = (rq + 1) & 3;
  if (t < NW) wfl[t] = a.w[1][t];
  f32x4 accT = {0.f, 0.f, 0.f, 0.f}, accF = {0.f, 0.f, 0.f, 0.f};
  float bacc = 0.f;  // bias gradient: thread (net = t >> 8, cs = (t >> 5) & 7) sums 1/32 of the tile's positions
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    // tiles 64 q + 8 k + i  ->  image 8 q + i, tile k: the eight tiles of an image run on one XCD, close in time
    const int b = 8 * (tile >> 6) + (tile & 7), k = (tile >> 3) & 7;
    if (b >= a.B) continue;
    const int r0 = (k >> 1) * TR, c0 = (k & 1) * TC;
    const T* rb = reinterpret_cast<const T*>(a.r) + (long)b * CI * L;
    const T* zTb = reinterpret_cast<const T*>(a.dz[0]) + (long)b * a.z_bs;
    const T* zFb = reinterpret_cast<const T*>(a.dz[1]) + (long)b * a.z_bs;
    __syncthreads();  // the previous tile's readers are done
    // ---- stage the residual tile, the positions that cover it and the halos
    {
      constexpr int NV = CI * TR * (TC / 4) / NT, NZT = CO * TR * 4 / NT, NZF = CO * TC * 2 / NT;
      f32x4 v[NV], zt[NZT], zf[NZF];
#pragma unroll
      for (int q = 0; q < NV; ++q)  // i = t + NT q: channel q, row rq, quad c4
        v[q] = Elem<T>::ld4(rb + ((long)q * P + r0 + rq) * P + c0 + 4 * c4);
#pragma unroll
      for (int q = 0; q < NZT; ++q) {
        const int i = t + NT * q, p4 = i & 3, rr = i >> 2, cs = rr >> 5, r = rr & (TR - 1);
        zt[q] = Elem<T>::ld4(zTb + (long)cs * LO + (r0 + r) * (P / 4) + c0 / 4 + 4 * p4);
      }
#pragma unroll
      for (int q = 0; q < NZF; ++q) {
        const int i = t + NT * q, g4 = i & 1, cc = i >> 1, cs = cc >> 6, c = cc & (TC - 1);
        zf[q] = Elem<T>::ld4(zFb + (long)cs * LO + (c0 + c) * (P / 4) + r0 / 4 + 4 * g4);
      }
      float h = 0.f, ex = 0.f;
      if (t < CI * TR) {  // the element before each tile row: column c0 - 1, or the end of the row above
        const int ci = t >> 5, r = r0 + (t & (TR - 1));
        const long g = c0 > 0 ? ((long)ci * P + r) * P + c0 - 1 : ((long)ci * P + r - 1) * P + P - 1;
        if (c0 > 0 || r > 0) h = Elem<T>::ld(rb + g);
      } else if (t >= 256) {  // the element above each tile column: row r0 - 1, or the end of the column before
        const int u = t - 256, ci = u >> 6, c = c0 + (u & (TC - 1));
        const long g = r0 > 0 ? ((long)ci * P + r0 - 1) * P + c : ((long)ci * P + P - 1) * P + c - 1;
        if (r0 > 0 || c > 0) h = Elem<T>::ld(rb + g);
      }
      {  // netF: the position below each tile column's eight (its tap 0 is the column's last element of the tile)
        const int cs = t >> 6, c = t & (TC - 1);
        const int j = (c0 + c) * (P / 4) + r0 / 4 + TR / 4;
        if (j < LO) ex = Elem<T>::ld(zFb + (long)cs * LO + j);
      }
      float et = 0.f;
      if (t < CO * TR) {  // netT: the position after each tile row's sixteen
        const int cs = t >> 5, r = t & (TR - 1);
        const int j = (r0 + r) * (P / 4) + c0 / 4 + TC / 4;
        if (j < LO) et = Elem<T>::ld(zTb + (long)cs * LO + j);
      }
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        float* d = &res[q * RCH + (1 + rq) * RP + 1 + 4 * c4];
        d[0] = v[q][0]; d[1] = v[q][1]; d[2] = v[q][2]; d[3] = v[q][3];
      }
#pragma unroll
      for (int q = 0; q < NZT; ++q) {
        const int i = t + NT * q, p4 = i & 3, rr = i >> 2, cs = rr >> 5, r = rr & (TR - 1);
        float* d = &zT[cs * ZTP + r * ZTR + 4 * p4];
        d[0] = zt[q][0]; d[1] = zt[q][1]; d[2] = zt[q][2]; d[3] = zt[q][3];
      }
#pragma unroll
      for (int q = 0; q < NZF; ++q) {
        const int i = t + NT * q, g4 = i & 1, cc = i >> 1, cs = cc >> 6, c = cc & (TC - 1);
        float* d = &zF[cs * ZFP + 4 * g4 * TC + c];
        d[0] = zf[q][0]; d[TC] = zf[q][1]; d[2 * TC] = zf[q][2]; d[3 * TC] = zf[q][3];
      }
      if (t < CI * TR) res[(t >> 5) * RCH + (1 + (t & (TR - 1))) * RP] = h;
      else if (t >= 256) res[((t - 256) >> 6) * RCH + 1 + (t & (TC - 1))] = h;
      zF[(t >> 6) * ZFP + (TR / 4) * TC + (t & (TC - 1))] = ex;
      if (t < CO * TR) zT[(t >> 5) * ZTP + (t & (TR - 1)) * ZTR + TC / 4] = et;
    }
    __syncthreads();
    // the reconstruction terms' share of this thread's four output quads: in flight while the matrix cores work
    f32x4 gp[CI];
#pragma unroll
    for (int ci = 0; ci < CI; ++ci)
      gp[ci] = Elem<T>::ld4(reinterpret_cast<const T*>(a.gx1p) + (long)b * CI * L + ((long)ci * P + r0 + rq) * P + c0 + 4 * c4);
    // ---- weight gradients: four positions per matrix instruction, every eighth group per wavefront
    {
      const int cb = lm >> 2, tt = lm & 3;
#pragma unroll 4
      for (int it = 0; it < TR * (TC / 4) / 4 / (NT / 64); ++it) {  // netT: positions (r, p), four consecutive p
        const int s = wave + (NT / 64) * it;
        const int q = 4 * s + lk, r = q >> 4, p = q & 15;
        const float av = lm < CO ? zT[lm * ZTP + r * ZTR + p] : 0.f;
        accT = __builtin_amdgcn_mfma_f32_16x16x4f32(av, res[cb * RCH + (1 + r) * RP + 4 * p + tt], accT, 0, 0, 0);
      }
#pragma unroll 4
      for (int it = 0; it < TC * (TR / 4) / 4 / (NT / 64); ++it) {  // netF: positions (c, g), four consecutive c
        const int s = wave + (NT / 64) * it;
        const int q = 4 * s + lk, g = q >> 6, c = q & (TC - 1);
        const float av = lm < CO ? zF[lm * ZFP + g * TC + c] : 0.f;
        accF = __builtin_amdgcn_mfma_f32_16x16x4f32(av, res[cb * RCH + (4 * g + tt) * RP + 1 + c], accF, 0, 0, 0);
      }
    }
    {  // bias gradients
      const int net = t >> 8, cs = (t >> 5) & 7, part = t & 31;
      float sum = 0.f;
      if (net == 0) {
#pragma unroll
        for (int i = 0; i < TC / 4; ++i) sum += zT[cs * ZTP + part * ZTR + i];
      } else {
#pragma unroll
        for (int i = 0; i < 2 * (TR / 4); ++i) sum += zF[cs * ZFP + (i >> 1) * TC + 2 * part + (i & 1)];
      }
      bacc += sum;
    }
    // ---- data gradients and the combination: element (r, c) is tap (c + 1) % 4 of netT's position (r, (c + 1) / 4) and
    // tap (r + 1) % 4 of netF's position (c, (r + 1) / 4)
    T* gb = reinterpret_cast<T*>(a.gx1) + (long)b * CI * L;
    const int g = (rq + 1) >> 2;
#pragma unroll
    for (int ci = 0; ci < CI; ++ci) {
      f32x4 dT = {0.f, 0.f, 0.f, 0.f}, dF = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int cs = 0; cs < CO; ++cs) {
        const float z0 = zT[cs * ZTP + rq * ZTR + c4], z1 = zT[cs * ZTP + rq * ZTR + c4 + 1];
        const f32x4 w4 = uload4(a.w[0] + (cs * CI + ci) * 4);
        dT[0] = fmaf(z0, w4[1], dT[0]); dT[1] = fmaf(z0, w4[2], dT[1]); dT[2] = fmaf(z0, w4[3], dT[2]);
        dT[3] = fmaf(z1, w4[0], dT[3]);
        const f32x4 zf4 = *reinterpret_cast<const f32x4*>(&zF[cs * ZFP + g * TC + 4 * c4]);
        const float wfv = wfl[(cs * CI + ci) * 4 + tap];
#pragma unroll
        for (int e = 0; e < 4; ++e) dF[e] = fmaf(zf4[e], wfv, dF[e]);
      }
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = gp[ci][e] - 0.5f * (stored<T>(dT[e]) + stored<T>(dF[e]));
      Elem<T>::st4(gb + ((long)ci * P + r0 + rq) * P + c0 + 4 * c4, o);
    }
  }
  // ---- the eight wavefronts' weight-gradient images -> one slab (fixed order), bias partials
  __syncthreads();
  float* comb = zT;  // [wave][net][cs][16]
  if (lk < 2) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      comb[((wave * 2 + 0) * CO + 4 * lk + r) * 16 + lm] = accT[r];
      comb[((wave * 2 + 1) * CO + 4 * lk + r) * 16 + lm] = accF[r];
    }
  }
  // thirty-two lanes share one (network, channel): butterfly inside the half wavefront
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) bacc += __shfl_xor(bacc, off, 32);
  __syncthreads();
  float* out = a.partial + (size_t)blockIdx.x * SLAB;
  if (t < 2 * NW) {
    const int net = t >> 7, i = t & 127;  // 2 x 128 weights
    const float* cp = comb + net * NW + i;
    constexpr int WS = 2 * NW;            // one wavefront's image
    out[net * (NW + 16) + i] = ((cp[0] + cp[WS]) + (cp[2 * WS] + cp[3 * WS])) + ((cp[4 * WS] + cp[5 * WS]) + (cp[6 * WS] + cp[7 * WS]));
  }
  if ((t & 31) == 0) out[(t >> 8) * (NW + 16) + NW + ((t >> 5) & 7)] = bacc;
  if (t < 16) out[(t >> 3) * (NW + 16) + NW + 8 + (t & 7)] = 0.f;
}

bool conv0_bwd_tile_supported(int C, int Pp, int Cin, int Cout, int L1d, long in_bs) {
  // (the schedule is decided where the plan is built: a device whose workgroups cannot hold the tile takes the separate launches)
  return !sched(LSHM_SCHED_NO_CONV0_BWD_TILE) && C == CI && Pp == P && Cin == CI && Cout == CO && L1d == L && in_bs == (long)CI * L &&
         device_lds_fits(sizeof(float) * (CI * RCH + CO * ZTP + CO * ZFP + NW));
}
size_t conv0_bwd_tile_workspace_floats() { return (size_t)MAX_GRID * SLAB; }

// dw / db of both networks: closed here (defer == nullptr) or queued as closing sums
int conv0_bwd_tile(const float* r, const float* dzT, const float* dzF, long z_bs, const float* wT, const float* wF, const float* gx1p,
                   float* gx1, float* dwT, float* dbT, float* dwF, float* dbF, int B, float* ws, size_t wsf, int accumulate,
                   hipStream_t st, GradJobs* defer, int bf) {
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (!r || !dzT || !dzF || !wT || !wF || !gx1p || !gx1 || !dwT || !dwF || !ws || B < 1 || z_bs % 4 || !al16(r) || !al16(dzT) || !al16(dzF) ||
      !al16(wT) || !al16(wF) || !al16(gx1p) || !al16(gx1)) {
    set_last_error("conv0_bwd_tile: null or unaligned pointer");
    return LSHM_ERR_ARG;
  }
  if (wsf < conv0_bwd_tile_workspace_floats()) { set_last_error("conv0_bwd_tile: workspace too small"); return LSHM_ERR_WORKSPACE; }
  Conv0BwdTileArgs a;
  a.r = r; a.dz[0] = dzT; a.dz[1] = dzF; a.w[0] = wT; a.w[1] = wF; a.gx1p = gx1p; a.gx1 = gx1; a.partial = ws;
  a.z_bs = z_bs; a.B = B; a.ntiles = ((B + 7) / 8) * 8 * TPI;  // (strides and offsets in elements of the storage type)
  int rc = kernel_budget_ok(bf ? reinterpret_cast<const void*>(&conv0_bwd_tile_kernel<bf16>) : reinterpret_cast<const void*>(&conv0_bwd_tile_kernel<float>),
                            NT, 0, "conv0_bwd_tile");
  if (rc) return rc;
  // two workgroups fit a CU (LDS); each keeps its weight-gradient accumulators over its tiles.  256 / 512 / 768 / 1024 workgroups for
  // the 2048 tiles of B = 256: 1.820 / 1.816 / 1.793 / 1.802 ms per iteration -- the weight-gradient stream's kernels run beside this one
  const int grid = a.ntiles < 768 ? a.ntiles : 768;
  if (bf) hipLaunchKernelGGL(conv0_bwd_tile_kernel<bf16>, dim3(grid), dim3(NT), 0, st, a);
  else hipLaunchKernelGGL(conv0_bwd_tile_kernel<float>, dim3(grid), dim3(NT), 0, st, a);
  if ((rc = check_launch("conv0_bwd_tile"))) return rc;
  float* dw[2] = {dwT, dwF};
  float* db[2] = {dbT, dbF};
  for (int g = 0; g < 2; ++g) {
    const float* part = ws + g * (NW + 16);
    if (defer) {
      defer->sums.push_back(SumJob{part, dw[g], SLAB, NW, grid, 0, 0, 0, 0, accumulate, 0});
      if (db[g]) defer->sums.push_back(SumJob{part + NW, db[g], SLAB, CO, grid, 0, 0, 0, 0, accumulate, 0});
    } else {
      if ((rc = reduce_partials_strided(part, SLAB, dw[g], NW, grid, accumulate, st, nullptr, nullptr))) return rc;
      if (db[g] && (rc = reduce_partials_strided(part + NW, SLAB, db[g], CO, grid, accumulate, st, nullptr, nullptr))) return rc;
    }
  }
  return LSHM_OK;
}

}  // namespace lshm
