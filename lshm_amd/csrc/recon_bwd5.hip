// The reconstruction pass of the paired schedule (multiplier update of one ADMM iteration + reconstruction terms and their
// gradients for the next closure, src/kharmonic_lofar.py:150-158,200-202; recon_kernel<UPD, GRAD, float, FROMA> of
// elementwise.hip) TOGETHER with the backward of the last layer of netT and netF (ConvTranspose1d(8, 4, 4, stride=4),
// src/lofar_models.py:142) that consumes two of its three gradient images:
//   x2 = netT.tconv5(aT), x3 = netF.tconv5(aF)^T      (formed here from the layer's input, as in the FROMA pass)
//   y_k += rho r_k; seven sums; gx1p                   (element for element the arithmetic of recon_kernel: the same bits)
//   g2 = d/dx2, g3 = d/dx3                            (kept in LDS, never written)
//   dW_T[cs][co][t] += aT[cs][j] g2[co][4j + t],  db_T[co] += g2[co][.],  d aT[cs][j] = ELU'(aT[cs][j]) sum_(co,t) g2[co][4j + t] wT[cs][co][t]
//   ... and the same for netF on the column-vectorised sequence (read column-wise from the g3 tile)
// A workgroup takes a 32 x 32 tile of ALL four planes of a sample (512 threads: two planes at a time, each 256-thread half the
// thread map of recon_kernel), because the layer sums over the four output channels.  The two gradient images (134 MB written
// here, 134 MB + the layer's input 67 MB read by conv1d_bwd_fused_kernel) and that launch disappear from the iteration.
// Persistent workgroups keep the weight-gradient accumulators over their tiles; the sixteen tiles of a sample run on one XCD
// (tile map below), so the 32-byte runs of the two data gradients meet in that XCD's L2.
#include <stdlib.h>

#include "kernels.h"

namespace lshm {

namespace {
typedef const __attribute__((address_space(4))) f32x4* cf32x4_ptr;
__device__ __forceinline__ f32x4 uload4(const float* q) { return *(cf32x4_ptr)(q); }
constexpr int P = 128, C = 4, CA = 8, T32 = 32, LA = P * P / 4;
constexpr int NT = 512;
constexpr int GP = T32 * (T32 + 1) + 8;  // plane pitch of the two gradient tiles (8 mod 32: the four planes of a B fragment spread over the banks)
constexpr int SP = T32 * 8 + 4;          // channel pitch of the staged layer inputs (4 mod 32: the eight channels of an A fragment)
constexpr int NW = CA * C * 4;           // 128 weights per network
constexpr int SLAB = 2 * (NW + 16);
constexpr int MAX_GRID = 1024;
}  // namespace

struct ReconBwd5Args {
  const float* x; const float* x1;   // (B, 4, 128, 128)
  const float* aT; const float* aF;  // (B, 8, 4096), batch stride a_bs: the last layer's input (an ELU output)
  const float* wT; const float* bT; const float* wF; const float* bF;  // (8, 4, 4), (4)
  float* y1; float* y2; float* y3;
  double* partials;                  // [sample * 4 + plane][4][4][7]: what recon_sum7 reads
  float* gx1p;
  float* dT; float* dF;              // (B, 8, 4096), batch stride d_bs: gradients w.r.t. the layer's input, times ELU'
  float* slabs;                      // gridDim.x x SLAB
  const float* gx2; const float* gx3c;  // FUSED == false: the two gradient images (gx3c per plane transposed), read instead of formed
  long a_bs, d_bs;
  float rho, inv_n;
  int B, ntiles;
};

// FUSED == false: the layer's backward alone, from the two gradient images another pass wrote -- the same tiles on the same
// workgroups in the same order, so that a schedule without the fused pass gets the same bits (the closing sums are shared too)
// T: element type of x1, of the layers' inputs, of gx1p, of the two gradient images and of the data gradients (bf16 storage, DESIGN
// 4.6): the reconstructions and the two gradient images are then rounded to bf16 where the separate launches stored them
template <class T>
__device__ __forceinline__ float stored5(float v) {
  if constexpr (sizeof(T) == 2) return (float)(T)v;
  return v;
}
template <bool FUSED, class T>
__global__ __launch_bounds__(NT, 4) void recon_bwd5_kernel(const ReconBwd5Args a) {
  __shared__ __attribute__((aligned(16))) float g2[C * GP];     // d/dx2 of the tile, g2[co][r * 33 + c]
  __shared__ __attribute__((aligned(16))) float g3[C * GP];     // x3 (as read by the pass: [column][row]), then d/dx3 as g3[co][r * 33 + c]
  __shared__ __attribute__((aligned(16))) float stage[2 * CA * SP];  // aT: [cs][r * 8 + p];  aF: [cs][c * 8 + g]
  __shared__ float red[NT / 64][9];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lm = lane & 15, lk = lane >> 4;
  const int tx = t & 31, ty = (t >> 5) & 7, ph = t >> 8, tid = t & 255;
  const float rho = a.rho, inv_n = a.inv_n;
  f32x4 accT = {0.f, 0.f, 0.f, 0.f}, accF = {0.f, 0.f, 0.f, 0.f};
  float bias0 = 0.f, bias1 = 0.f;  // tid 7: db_T of planes ph and 2 + ph; tid 8: db_F of the same
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    // tiles 128 q + 8 k + i  ->  sample 8 q + i, tile k: the sixteen tiles of a sample run on one XCD, close in time
    const int b = 8 * (tile >> 7) + (tile & 7), kt = (tile >> 3) & 15;
    if (b >= a.B) continue;
    const int h0 = (kt >> 2) * T32, w0 = (kt & 3) * T32;
    __syncthreads();  // the previous tile's readers are done
#pragma unroll
    for (int u = 0; u < 2; ++u) {  // the layer inputs a 32 x 32 tile needs: 32 rows (columns) x 8 positions x 8 channels per network
      const int j = t + NT * u, net = j >> 9, idx = j & 511, cs = idx >> 6, rr = (idx >> 1) & 31, half = idx & 1;
      const T* src = net == 0 ? reinterpret_cast<const T*>(a.aT) + (long)b * a.a_bs + (long)cs * LA + (long)(h0 + rr) * (P / 4) + w0 / 4 + 4 * half
                              : reinterpret_cast<const T*>(a.aF) + (long)b * a.a_bs + (long)cs * LA + (long)(w0 + rr) * (P / 4) + h0 / 4 + 4 * half;
      *reinterpret_cast<f32x4*>(&stage[(net * CA + cs) * SP + rr * 8 + 4 * half]) = Elem<T>::ld4(src);
    }
    __syncthreads();
    const float* sT = stage;
    const float* sF = stage + CA * SP;
#pragma unroll 1
    for (int k = 0; k < 2; ++k) {
      const int ch = 2 * k + ph;
      float* g2p = g2 + ch * GP;
      float* g3p = g3 + ch * GP;
      float b2 = 0.f, b3 = 0.f;
      const long plane = ((long)b * C + ch) * P * P;
      if constexpr (!FUSED) {
#pragma unroll
        for (int i = 0; i < 4; ++i)  // d/dx3 arrives per plane transposed: read along the image's columns, kept in image orientation
          g3p[tx * (T32 + 1) + ty + 8 * i] = Elem<T>::ld(reinterpret_cast<const T*>(a.gx3c) + plane + (long)(w0 + ty + 8 * i) * P + h0 + tx);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float v2 = Elem<T>::ld(reinterpret_cast<const T*>(a.gx2) + plane + (long)(h0 + ty + 8 * i) * P + w0 + tx);
          g2p[(ty + 8 * i) * (T32 + 1) + tx] = v2;
          b2 += v2; b3 += g3p[(ty + 8 * i) * (T32 + 1) + tx];
        }
        __syncthreads();
      } else {
        const int tap = tx & 3;
        float wt[CA], wf[CA];
#pragma unroll
        for (int cs = 0; cs < CA; ++cs) {
          wt[cs] = a.wT[(cs * C + ch) * 4 + tap];
          wf[cs] = a.wF[(cs * C + ch) * 4 + tap];
        }
        const float bt = a.bT[ch], bfv = a.bF[ch];
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // x3 at (column w0 + ty + 8 i, row h0 + tx)
          float v = bfv;
#pragma unroll
          for (int cs = 0; cs < CA; ++cs) v = fmaf(sF[cs * SP + (ty + 8 * i) * 8 + (tx >> 2)], wf[cs], v);
          g3p[(ty + 8 * i) * (T32 + 1) + tx] = stored5<T>(v);
        }
        __syncthreads();
        float s[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float gv3[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const long o = plane + (long)(h0 + ty + 8 * i) * P + w0 + tx;
          const float xv = __builtin_nontemporal_load(a.x + o), a1 = Elem<T>::ld(reinterpret_cast<const T*>(a.x1) + o);
          float a2 = bt;
#pragma unroll
          for (int cs = 0; cs < CA; ++cs) a2 = fmaf(sT[cs * SP + (ty + 8 * i) * 8 + (tx >> 2)], wt[cs], a2);
          a2 = stored5<T>(a2);
          const float a3 = g3p[tx * (T32 + 1) + ty + 8 * i];
          const float m1 = __builtin_nontemporal_load(a.y1 + o), m2 = __builtin_nontemporal_load(a.y2 + o), m3 = __builtin_nontemporal_load(a.y3 + o);
          const ReconElem q = recon_elem<true, true>(xv, a1, a2, a3, m1, m2, m3, rho, inv_n, s);
          a.y1[o] = q.m1; a.y2[o] = q.m2; a.y3[o] = q.m3;
          const float v2 = stored5<T>(q.g2);
          g2p[(ty + 8 * i) * (T32 + 1) + tx] = v2;
          gv3[i] = stored5<T>(q.g3);
          b2 += v2; b3 += gv3[i];
          Elem<T>::st(reinterpret_cast<T*>(a.gx1p) + o, q.g1p);
        }
        __syncthreads();  // every x3 of the tile has been read: the plane's tile now takes d/dx3 in image orientation
#pragma unroll
        for (int i = 0; i < 4; ++i) g3p[(ty + 8 * i) * (T32 + 1) + tx] = gv3[i];
        // per-wavefront sums in fp32 (256 terms each), combined across the plane's 4 waves and all tiles in fp64 (recon_kernel's order)
#pragma unroll
        for (int q = 0; q < 7; ++q) {
          const float v = wave_sum(s[q]);
          if (lane == 0) red[wave][q] = v;
        }
      }
      {
        const float v2 = wave_sum(b2), v3 = wave_sum(b3);
        if (lane == 0) { red[wave][7] = v2; red[wave][8] = v3; }
      }
      __syncthreads();
      const float (*rp)[9] = red + 4 * ph;
      if (FUSED && tid < 7) {
        const long blk = (((long)b * C + ch) * (P / T32) + h0 / T32) * (P / T32) + w0 / T32;
        a.partials[blk * 7 + tid] = ((double)rp[0][tid] + (double)rp[1][tid]) + ((double)rp[2][tid] + (double)rp[3][tid]);
      } else if (tid == 7 || tid == 8) {
        const float v = (rp[0][tid] + rp[1][tid]) + (rp[2][tid] + rp[3][tid]);
        if (k == 0) bias0 += v; else bias1 += v;
      }
    }
    __syncthreads();  // both gradient tiles are complete for the four planes
    // ---- data gradients: one position of one network per thread, all eight input channels
    {
      const int net = __builtin_amdgcn_readfirstlane(t >> 8), q = t & 255;  // (wavefronts 0-3: netT, 4-7: netF)
      float gv[C][4];
      const float* w;
      const float* sa;
      T* dst;
      if (net == 0) {  // netT: position (row r, p): elements 4p .. 4p + 3 of the row
        const int r = q >> 3, p = q & 7;
#pragma unroll
        for (int co = 0; co < C; ++co)
#pragma unroll
          for (int e = 0; e < 4; ++e) gv[co][e] = g2[co * GP + r * (T32 + 1) + 4 * p + e];
        w = a.wT; sa = sT + r * 8 + p;
        dst = reinterpret_cast<T*>(a.dT) + (long)b * a.d_bs + (long)(h0 + r) * (P / 4) + w0 / 4 + p;
      } else {         // netF: position (column c, g): elements 4g .. 4g + 3 of the column
        const int c = q >> 3, g = q & 7;
#pragma unroll
        for (int co = 0; co < C; ++co)
#pragma unroll
          for (int e = 0; e < 4; ++e) gv[co][e] = g3[co * GP + (4 * g + e) * (T32 + 1) + c];
        w = a.wF; sa = sF + c * 8 + g;
        dst = reinterpret_cast<T*>(a.dF) + (long)b * a.d_bs + (long)(w0 + c) * (P / 4) + h0 / 4 + g;
      }
#pragma unroll
      for (int cs = 0; cs < CA; ++cs) {
        float acc = 0.f;
#pragma unroll
        for (int co = 0; co < C; ++co) {
          const f32x4 w4 = uload4(w + (cs * C + co) * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc = fmaf(gv[co][e], w4[e], acc);
        }
        Elem<T>::st(dst + (long)cs * LA, acc * elu_grad_from_out(sa[cs * SP]));
      }
    }
    // ---- weight gradients: four positions per matrix instruction, every eighth group per wavefront
    {
      const int co = lm >> 2, tt = lm & 3;
#pragma unroll 4
      for (int it = 0; it < T32 * 8 / 4 / (NT / 64); ++it) {
        const int s4 = wave + (NT / 64) * it;
        const int q = 4 * s4 + lk, r = q >> 3, p = q & 7;  // netT: (row r, position p); netF: (column r, position p)
        const float avT = lm < CA ? sT[lm * SP + q] : 0.f;
        accT = __builtin_amdgcn_mfma_f32_16x16x4f32(avT, g2[co * GP + r * (T32 + 1) + 4 * p + tt], accT, 0, 0, 0);
        const float avF = lm < CA ? sF[lm * SP + q] : 0.f;
        accF = __builtin_amdgcn_mfma_f32_16x16x4f32(avF, g3[co * GP + (4 * p + tt) * (T32 + 1) + r], accF, 0, 0, 0);
      }
    }
  }
  // ---- the eight wavefronts' weight-gradient images -> one slab (fixed order), bias partials
  __syncthreads();
  float* comb = g2;  // [wave][net][cs][16]
  if (lk < 2) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      comb[((wave * 2 + 0) * CA + 4 * lk + r) * 16 + lm] = accT[r];
      comb[((wave * 2 + 1) * CA + 4 * lk + r) * 16 + lm] = accF[r];
    }
  }
  __syncthreads();
  float* out = a.slabs + (size_t)blockIdx.x * SLAB;
  if (t < 2 * NW) {
    const int net = t >> 7, i = t & 127;
    const float* cp = comb + net * NW + i;
    constexpr int WS = 2 * NW;
    out[net * (NW + 16) + i] = ((cp[0] + cp[WS]) + (cp[2 * WS] + cp[3 * WS])) + ((cp[4 * WS] + cp[5 * WS]) + (cp[6 * WS] + cp[7 * WS]));
  }
  if (tid == 7 || tid == 8) {
    float* o = out + (tid - 7) * (NW + 16) + NW;
    o[ph] = bias0;
    o[2 + ph] = bias1;
  }
  if (t >= 2 * NW && t < 2 * NW + 24) {
    const int i = t - 2 * NW;
    out[(i / 12) * (NW + 16) + NW + 4 + i % 12] = 0.f;
  }
}

bool recon_bwd5_supported(int Cc, int Pp, int Cin, int Cout, int Ls) {
  return !sched(LSHM_SCHED_NO_RECON_BWD5) && Cc == C && Pp == P && Cin == CA && Cout == C && Ls == LA &&
         device_lds_fits(sizeof(float) * (2 * C * GP + 2 * CA * SP + (NT / 64) * 9));
}
size_t recon_bwd5_workspace_floats() { return (size_t)MAX_GRID * SLAB; }

// The closing sums of the two layers' weight / bias gradients from the slabs the pass left (deferred, or on `st`)
int recon_bwd5_close(const float* slabs, int grid, float* dwT, float* dbT, float* dwF, float* dbF, int accumulate, hipStream_t st,
                     GradJobs* defer) {
  float* dw[2] = {dwT, dwF};
  float* db[2] = {dbT, dbF};
  int rc;
  for (int g = 0; g < 2; ++g) {
    const float* part = slabs + g * (NW + 16);
    if (defer) {
      defer->sums.push_back(SumJob{part, dw[g], SLAB, NW, grid, 0, 0, 0, 0, accumulate, 0});
      if (db[g]) defer->sums.push_back(SumJob{part + NW, db[g], SLAB, C, grid, 0, 0, 0, 0, accumulate, 0});
    } else {
      if ((rc = reduce_partials_strided(part, SLAB, dw[g], NW, grid, accumulate, st, nullptr, nullptr))) return rc;
      if (db[g] && (rc = reduce_partials_strided(part + NW, SLAB, db[g], C, grid, accumulate, st, nullptr, nullptr))) return rc;
    }
  }
  return LSHM_OK;
}

// 768 workgroups for the 4096 tiles of B = 256 (two fit a CU: 512 threads x 128 registers): 256 / 384 / 512 / 768 / 1024 ->
// 1.860 / 1.828 / 1.803 / 1.790 / 1.799 ms per iteration (profiles/r04/README.md) -- with every CU held by persistent workgroups
// from start to end, the closure forward's last kernels, which run beside the pass, wait for all of it
int recon_bwd5_grid(int B) {
  const int ntiles = ((B + 7) / 8) * 8 * 16;
  return ntiles < 768 ? ntiles : 768;
}

int recon_bwd5(const float* x, const float* x1, const float* aT, const float* aF, long a_bs, const float* wT, const float* bT,
               const float* wF, const float* bF, float* y1, float* y2, float* y3, float rho, int B, float* gx1p, float* dT, float* dF,
               long d_bs, float* block_partials, float* slabs, size_t slab_floats, hipStream_t st, float grad_scale, int bf) {
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (!x || !x1 || !aT || !aF || !wT || !bT || !wF || !bF || !y1 || !y2 || !y3 || !gx1p || !dT || !dF || !block_partials || !slabs || B < 1 ||
      a_bs % 4 || !al16(aT) || !al16(aF) || !al16(wT) || !al16(wF)) {
    set_last_error("recon_bwd5: null or unaligned pointer");
    return LSHM_ERR_ARG;
  }
  if (slab_floats < recon_bwd5_workspace_floats()) { set_last_error("recon_bwd5: workspace too small"); return LSHM_ERR_WORKSPACE; }
  const double n = (double)B * C * P * P;
  ReconBwd5Args a;
  a.x = x; a.x1 = x1; a.aT = aT; a.aF = aF; a.wT = wT; a.bT = bT; a.wF = wF; a.bF = bF;
  a.y1 = y1; a.y2 = y2; a.y3 = y3; a.partials = reinterpret_cast<double*>(block_partials); a.gx1p = gx1p; a.dT = dT; a.dF = dF;
  a.slabs = slabs; a.a_bs = a_bs; a.d_bs = d_bs; a.rho = rho; a.inv_n = (float)(grad_scale / n);
  a.B = B; a.ntiles = ((B + 7) / 8) * 8 * 16;
  a.gx2 = a.gx3c = nullptr;
  int rc = kernel_budget_ok(bf ? reinterpret_cast<const void*>(&recon_bwd5_kernel<true, bf16>) : reinterpret_cast<const void*>(&recon_bwd5_kernel<true, float>),
                            NT, 0, "recon_bwd5");
  if (rc) return rc;
  if (bf) hipLaunchKernelGGL((recon_bwd5_kernel<true, bf16>), dim3(recon_bwd5_grid(B)), dim3(NT), 0, st, a);
  else hipLaunchKernelGGL((recon_bwd5_kernel<true, float>), dim3(recon_bwd5_grid(B)), dim3(NT), 0, st, a);
  return check_launch("recon_bwd5");
}

// The layer's backward alone (both networks), bit for bit what the fused pass leaves: data gradients times ELU' in dT / dF, slabs for
// recon_bwd5_close.  gx2: (B, 4, 128, 128); gx3c: the same per plane transposed.
int tconv5_pair_bwd(const float* gx2, const float* gx3c, const float* aT, const float* aF, long a_bs, const float* wT, const float* wF,
                    int B, float* dT, float* dF, long d_bs, float* slabs, size_t slab_floats, hipStream_t st, int bf) {
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (!gx2 || !gx3c || !aT || !aF || !wT || !wF || !dT || !dF || !slabs || B < 1 || a_bs % 4 || !al16(aT) || !al16(aF) || !al16(wT) || !al16(wF)) {
    set_last_error("tconv5_pair_bwd: null or unaligned pointer");
    return LSHM_ERR_ARG;
  }
  if (slab_floats < recon_bwd5_workspace_floats()) { set_last_error("tconv5_pair_bwd: workspace too small"); return LSHM_ERR_WORKSPACE; }
  ReconBwd5Args a{};
  a.aT = aT; a.aF = aF; a.wT = wT; a.wF = wF; a.dT = dT; a.dF = dF; a.slabs = slabs; a.gx2 = gx2; a.gx3c = gx3c;
  a.a_bs = a_bs; a.d_bs = d_bs; a.B = B; a.ntiles = ((B + 7) / 8) * 8 * 16;
  int rc = kernel_budget_ok(bf ? reinterpret_cast<const void*>(&recon_bwd5_kernel<false, bf16>) : reinterpret_cast<const void*>(&recon_bwd5_kernel<false, float>),
                            NT, 0, "tconv5_pair_bwd");
  if (rc) return rc;
  if (bf) hipLaunchKernelGGL((recon_bwd5_kernel<false, bf16>), dim3(recon_bwd5_grid(B)), dim3(NT), 0, st, a);
  else hipLaunchKernelGGL((recon_bwd5_kernel<false, float>), dim3(recon_bwd5_grid(B)), dim3(NT), 0, st, a);
  return check_launch("tconv5_pair_bwd");
}

}  // namespace lshm
