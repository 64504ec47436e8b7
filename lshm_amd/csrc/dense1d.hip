// The dense middle of AutoEncoder1DCNN as ONE launch per direction (src/lofar_models.py:127-135,165-176 with rica=True,
// latent_dim = 16):   fc1 (784 -> 16, ELU) -> fc2in (16 -> 16, ELU: the latent code) -> fc2out (16 -> 16, ELU) ->
// fc3 (32 -> 768), and the data-gradient pass back through the same four layers.  As separate launches these were
// five forward launches (fc1 with its split-K combine) and five backward ones, 4-10 us each for a few hundred
// kilobytes: pure launch latency on the critical chain of the step.  Here a workgroup takes 16 rows of the batch
// (one MFMA row tile), keeps them in LDS from layer to layer and streams the weights (51 KB + 98 KB) from L2; the two
// problems of a pair (netT, netF) are grid.y.
//   forward : X = cat1 rows [16 x 784] -> z1 = elu(X fc1w^T + b) -> mu = elu(z1 fc2in^T + b) -> c = elu(mu fc2out^T + b)
//             -> d0 = [c | elu(fcuv3(uvh))] fc3w^T + b        (the uv half of cat3 is already in place: uv_features)
//   backward: dd0 [16 x 768] -> dcat3 = (dd0 fc3w) * ELU'(cat3) -> dzmu = (dcat3[:, :16] fc2outw + gMu) * ELU'(mu)
//             -> dz1 = (dzmu fc2inw) * ELU'(z1) -> dcat1 = (dz1 fc1w) * ELU'(cat1)
// v_mfma_f32_16x16x4_f32 (exact fp32).  K-fast weight rows are read as float4 (k = 16 s + 4 lk + e), the matching A
// fragment is one ds_read_b128 of the row image (pitch == 20 (mod 64) floats: conflict-free for both b32 and b128).
#include <stdlib.h>

#include "kernels.h"

namespace lshm {

namespace {
constexpr int LT = 16, HD = 16, NIN = 768 + HD;  // latent width, harmonic width, fc1 input width
constexpr int XP = NIN + 4;                      // row pitch of the [16 x 784] image: 788 == 20 (mod 64)
constexpr int DP = 768 + 4;                      // row pitch of the [16 x 768] image: 772 == 4 (mod 64)
constexpr int SP = 36;                           // row pitch of the small [16 x <=32] images
constexpr int NTH = 1024, NWV = NTH / 64;        // 16 wavefronts per workgroup: the K of fc1 / fc3' and the column tiles of
                                                 // fc3 / fc1' are shared 16 ways -- only 16-32 workgroups exist, so a wavefront's serial chain of L2 round trips is what the launch costs

// C[16 x 16] += A[16 x K] (LDS rows, pitch AP, k-fast) * W^T, W = rows of length ldw (k-fast, global): this wavefront's
// share of the 16-wide k-blocks (blocks wave, wave + nw, ...); n0: first weight row.  Lane (lm, lk): D rows 4 lk.., col lm.
template <int AP>
__device__ __forceinline__ f32x4 gemm_kfast(const float* __restrict__ As, const float* __restrict__ W, long ldw, int n0, int nvalid,
                                            int kblocks, int first, int step, f32x4 acc) {
  const int lane = threadIdx.x & 63, lm = lane & 15, lk = lane >> 4;
  const bool nok = lm < nvalid;
  const float* wrow = W + (long)(n0 + (nok ? lm : 0)) * ldw + 4 * lk;
  const float* arow = As + lm * AP + 4 * lk;
  for (int s = first; s < kblocks; s += step) {  // (run-time bounds: the unroller refuses a factor here)
    f32x4 b = *reinterpret_cast<const f32x4*>(wrow + 16 * s);
    if (!nok) b = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 a = *reinterpret_cast<const f32x4*>(arow + 16 * s);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc, 0, 0, 0);
  }
  return acc;
}
// C[16 x 16] = A[16 x K] (LDS rows, pitch AP) * W, W[k][n] n-fast with leading dimension ldw (global), columns n0..:
// the whole K on this wavefront (K small: 16 or 32)
template <int AP, int K>
__device__ __forceinline__ f32x4 gemm_nfast(const float* __restrict__ As, const float* __restrict__ W, long ldw, int n0, int nvalid) {
  const int lane = threadIdx.x & 63, lm = lane & 15, lk = lane >> 4;
  const bool nok = lm < nvalid;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < K / 4; ++s) {
    const float b = nok ? W[(long)(4 * s + lk) * ldw + n0 + lm] : 0.f;
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(As[lm * AP + 4 * s + lk], b, acc, 0, 0, 0);
  }
  return acc;
}
}  // namespace

struct Dense1dFwdArgs {
  const float* cat1[2];   // (B, 784)  [conv5 output | elu(fcuv1)]
  const float* fc1w[2]; const float* fc1b[2];
  const float* fc2inw[2]; const float* fc2inb[2];
  const float* fc2outw[2]; const float* fc2outb[2];
  const float* fc3w[2]; const float* fc3b[2];
  float* z1[2];           // (B, 16)
  float* mu[2]; long ldmu;  // (B, 16) inside Mu (B, D)
  float* cat3[2];         // (B, 32): columns 0..15 written here, 16..31 (elu(fcuv3)) read
  float* d0[2];           // (B, 768)
  int B;
};

__global__ __launch_bounds__(NTH) void dense1d_fwd_kernel(const Dense1dFwdArgs a) {
  __shared__ __attribute__((aligned(16))) float xs[16 * XP];
  __shared__ __attribute__((aligned(16))) float red[NWV][64 * 4];
  __shared__ __attribute__((aligned(16))) float z1s[16 * SP], mus[16 * SP], c3s[16 * SP];
  const int pr = blockIdx.y, r0 = blockIdx.x * 16;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lm = lane & 15, lk = lane >> 4;
  const int nrow = a.B - r0 < 16 ? a.B - r0 : 16;
  // ---- stage the 16 input rows (zero rows past the batch)
  const float* x = a.cat1[pr] + (long)r0 * NIN;
  for (int i = t; i < 16 * (NIN / 4); i += NTH) {
    const int r = i / (NIN / 4), c4 = i - r * (NIN / 4);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r < nrow) v = *reinterpret_cast<const f32x4*>(x + (long)r * NIN + 4 * c4);
    *reinterpret_cast<f32x4*>(xs + r * XP + 4 * c4) = v;
  }
  // the uv half of cat3 (written by uv_features before this launch)
  for (int i = t; i < 16 * HD; i += NTH) {
    const int r = i / HD, c = i - r * HD;
    c3s[r * SP + LT + c] = r < nrow ? a.cat3[pr][(long)(r0 + r) * (LT + HD) + LT + c] : 0.f;
  }
  __syncthreads();
  // ---- fc1: K = 784 split over the wavefronts (49 k-blocks), partial tiles added in wavefront order
  f32x4 acc = gemm_kfast<XP>(xs, a.fc1w[pr], NIN, 0, LT, NIN / 16, wave, NWV, (f32x4){0.f, 0.f, 0.f, 0.f});
  *reinterpret_cast<f32x4*>(&red[wave][4 * lane]) = acc;
  __syncthreads();
  if (wave == 0) {
    f32x4 v = *reinterpret_cast<const f32x4*>(&red[0][4 * lane]);
#pragma unroll
    for (int w = 1; w < NWV; ++w) v += *reinterpret_cast<const f32x4*>(&red[w][4 * lane]);
    const float bv = a.fc1b[pr][lm];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float o = elu(v[r] + bv);
      z1s[(4 * lk + r) * SP + lm] = o;
      if (4 * lk + r < nrow) a.z1[pr][(long)(r0 + 4 * lk + r) * LT + lm] = o;
    }
  }
  __syncthreads();
  // ---- the three small layers are dependent 16 x 16 products: one wavefront, W[k][n] = w[n * 16 + k] (k-fast rows)
  if (wave == 0) {
    f32x4 m = gemm_kfast<SP>(z1s, a.fc2inw[pr], LT, 0, LT, 1, 0, 1, (f32x4){0.f, 0.f, 0.f, 0.f});
    const float b2 = a.fc2inb[pr][lm];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float o = elu(m[r] + b2);
      mus[(4 * lk + r) * SP + lm] = o;
      if (4 * lk + r < nrow) a.mu[pr][(long)(r0 + 4 * lk + r) * a.ldmu + lm] = o;
    }
  }
  __syncthreads();
  if (wave == 0) {
    f32x4 c = gemm_kfast<SP>(mus, a.fc2outw[pr], LT, 0, LT, 1, 0, 1, (f32x4){0.f, 0.f, 0.f, 0.f});
    const float b3 = a.fc2outb[pr][lm];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float o = elu(c[r] + b3);
      c3s[(4 * lk + r) * SP + lm] = o;
      if (4 * lk + r < nrow) a.cat3[pr][(long)(r0 + 4 * lk + r) * (LT + HD) + lm] = o;
    }
  }
  __syncthreads();
  // ---- fc3: 768 outputs = 48 column tiles, 3 per wavefront; K = 32 (two k-blocks)
#pragma unroll
  for (int it = 0; it < (768 / 16 + NWV - 1) / NWV; ++it) {
    const int nt = wave + it * NWV;
    if (nt >= 768 / 16) break;
    const f32x4 o = gemm_kfast<SP>(c3s, a.fc3w[pr], LT + HD, 16 * nt, 16, 2, 0, 1, (f32x4){0.f, 0.f, 0.f, 0.f});
    const float bv = a.fc3b[pr][16 * nt + lm];
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (4 * lk + r < nrow) a.d0[pr][(long)(r0 + 4 * lk + r) * 768 + 16 * nt + lm] = o[r] + bv;
  }
}

struct Dense1dBwdArgs {
  const float* dd0[2];    // (B, 768) gradient w.r.t. fc3's output
  const float* cat3[2];   // (B, 32) saved
  const float* mu[2]; long ldmu;    // saved latent (inside Mu)
  const float* gmu[2]; long ldgmu;  // gradient of the latent-space terms w.r.t. the latent (inside gMu)
  const float* z1[2];     // (B, 16) saved
  const float* cat1[2];   // (B, 784) saved
  const float* fc1w[2]; const float* fc2inw[2]; const float* fc2outw[2]; const float* fc3w[2];
  float* dcat3[2];        // (B, 32)
  float* dzmu[2];         // (B, 16)
  float* dz1[2];          // (B, 16)
  float* dcat1[2];        // (B, 784)
  int B;
};

__global__ __launch_bounds__(NTH) void dense1d_bwd_kernel(const Dense1dBwdArgs a) {
  __shared__ __attribute__((aligned(16))) float ds[16 * DP];
  __shared__ __attribute__((aligned(16))) float red[NWV][2][64 * 4];
  __shared__ __attribute__((aligned(16))) float g3s[16 * SP], gms[16 * SP], g1s[16 * SP];
  const int pr = blockIdx.y, r0 = blockIdx.x * 16;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lm = lane & 15, lk = lane >> 4;
  const int nrow = a.B - r0 < 16 ? a.B - r0 : 16;
  const float* d = a.dd0[pr] + (long)r0 * 768;
  for (int i = t; i < 16 * (768 / 4); i += NTH) {
    const int r = i / (768 / 4), c4 = i - r * (768 / 4);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r < nrow) v = *reinterpret_cast<const f32x4*>(d + (long)r * 768 + 4 * c4);
    *reinterpret_cast<f32x4*>(ds + r * DP + 4 * c4) = v;
  }
  __syncthreads();
  // ---- dcat3 = (dd0 fc3w) * ELU'(cat3): M = 16, N = 32 (two column tiles), K = 768 over the wavefronts (192 k-steps)
  {
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const float* w3 = a.fc3w[pr];
#pragma unroll 4
    for (int it = 0; it < 768 / 4 / NWV; ++it) {
      const int s = wave + NWV * it;
      const float av = ds[lm * DP + 4 * s + lk];
      const float* wr = w3 + (long)(4 * s + lk) * (LT + HD) + lm;  // W[k = out][n = in]
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wr[16 * j], acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) *reinterpret_cast<f32x4*>(&red[wave][j][4 * lane]) = acc[j];
  }
  __syncthreads();
  if (wave < 2) {  // wavefront j finishes column tile j
    const int j = wave;
    f32x4 v = *reinterpret_cast<const f32x4*>(&red[0][j][4 * lane]);
#pragma unroll
    for (int w = 1; w < NWV; ++w) v += *reinterpret_cast<const f32x4*>(&red[w][j][4 * lane]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * lk + r;
      float o = 0.f;
      if (row < nrow) {
        o = v[r] * elu_grad_from_out(a.cat3[pr][(long)(r0 + row) * (LT + HD) + 16 * j + lm]);
        a.dcat3[pr][(long)(r0 + row) * (LT + HD) + 16 * j + lm] = o;
      }
      if (j == 0) g3s[row * SP + lm] = o;  // only the latent half feeds fc2out's data gradient
    }
  }
  __syncthreads();
  if (wave == 0) {
    // ---- dzmu = (dcat3[:, :16] fc2outw + gMu) * ELU'(mu)
    f32x4 m = gemm_nfast<SP, LT>(g3s, a.fc2outw[pr], LT, 0, LT);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * lk + r;
      float o = 0.f;
      if (row < nrow) {
        o = (m[r] + a.gmu[pr][(long)(r0 + row) * a.ldgmu + lm]) * elu_grad_from_out(a.mu[pr][(long)(r0 + row) * a.ldmu + lm]);
        a.dzmu[pr][(long)(r0 + row) * LT + lm] = o;
      }
      gms[row * SP + lm] = o;
    }
  }
  __syncthreads();
  if (wave == 0) {
    // ---- dz1 = (dzmu fc2inw) * ELU'(z1)
    f32x4 m = gemm_nfast<SP, LT>(gms, a.fc2inw[pr], LT, 0, LT);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * lk + r;
      float o = 0.f;
      if (row < nrow) {
        o = m[r] * elu_grad_from_out(a.z1[pr][(long)(r0 + row) * LT + lm]);
        a.dz1[pr][(long)(r0 + row) * LT + lm] = o;
      }
      g1s[row * SP + lm] = o;
    }
  }
  __syncthreads();
  // ---- dcat1 = (dz1 fc1w) * ELU'(cat1): N = 784 = 49 column tiles over the wavefronts, K = 16
#pragma unroll
  for (int it = 0; it < (NIN / 16 + NWV - 1) / NWV; ++it) {
    const int nt = wave + it * NWV;
    if (nt >= NIN / 16) break;
    const f32x4 m = gemm_nfast<SP, LT>(g1s, a.fc1w[pr], NIN, 16 * nt, 16);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * lk + r;
      if (row < nrow) {
        const long g = (long)(r0 + row) * NIN + 16 * nt + lm;
        a.dcat1[pr][g] = m[r] * elu_grad_from_out(a.cat1[pr][g]);
      }
    }
  }
}

// (A 224 / 256-wide form of the same kernels for the 2-D autoencoder existed in round 3: correct, but a workgroup's 16 rows ran
// ~2 MB of weights through ONE CU's matrix pipe -- 50 / 81 us against ~45 / ~75 for the five launches -- and was removed; those
// layers now run inside lshm_deep2d_fwd / _bwd, a patch per workgroup.)
bool dense1d_supported(int L, int hd, int rica) {
  return !sched(LSHM_SCHED_NO_DENSE1D) && rica && hd == HD && L == LT;
}
bool dense1d_built(int L) { return L == LT; }

int dense1d_fwd(const Dense1dFwdIO& p, const Dense1dFwdIO* p1, long ldmu, int B, hipStream_t st, int L) {
  Dense1dFwdArgs a;
  for (int g = 0; g < 2; ++g) {
    const Dense1dFwdIO& q = (g && p1) ? *p1 : p;
    a.cat1[g] = q.cat1; a.fc1w[g] = q.fc1w; a.fc1b[g] = q.fc1b; a.fc2inw[g] = q.fc2inw; a.fc2inb[g] = q.fc2inb;
    a.fc2outw[g] = q.fc2outw; a.fc2outb[g] = q.fc2outb; a.fc3w[g] = q.fc3w; a.fc3b[g] = q.fc3b;
    a.z1[g] = q.z1; a.mu[g] = q.mu; a.cat3[g] = q.cat3; a.d0[g] = q.d0;
    if (!(q.cat1 && q.fc1w && q.fc1b && q.fc2inw && q.fc2inb && q.fc2outw && q.fc2outb && q.fc3w && q.fc3b && q.z1 && q.mu &&
          q.cat3 && q.d0) || ((reinterpret_cast<uintptr_t>(q.cat1) | reinterpret_cast<uintptr_t>(q.fc1w) |
                               reinterpret_cast<uintptr_t>(q.fc2inw) | reinterpret_cast<uintptr_t>(q.fc2outw) |
                               reinterpret_cast<uintptr_t>(q.fc3w)) & 15)) {
      set_last_error("dense1d_fwd: null or unaligned pointer");
      return LSHM_ERR_ARG;
    }
  }
  a.ldmu = ldmu;
  a.B = B;
  if (L == LT) {
    hipLaunchKernelGGL(dense1d_fwd_kernel, dim3((B + 15) / 16, p1 ? 2 : 1), dim3(NTH), 0, st, a);
  } else {
    set_last_error("dense1d_fwd: latent width must be 16");
    return LSHM_ERR_UNSUPPORTED;
  }
  return check_launch("dense1d_fwd");
}

int dense1d_bwd(const Dense1dBwdIO& p, const Dense1dBwdIO* p1, long ldmu, long ldgmu, int B, hipStream_t st, int L) {
  Dense1dBwdArgs a;
  for (int g = 0; g < 2; ++g) {
    const Dense1dBwdIO& q = (g && p1) ? *p1 : p;
    a.dd0[g] = q.dd0; a.cat3[g] = q.cat3; a.mu[g] = q.mu; a.gmu[g] = q.gmu; a.z1[g] = q.z1; a.cat1[g] = q.cat1;
    a.fc1w[g] = q.fc1w; a.fc2inw[g] = q.fc2inw; a.fc2outw[g] = q.fc2outw; a.fc3w[g] = q.fc3w;
    a.dcat3[g] = q.dcat3; a.dzmu[g] = q.dzmu; a.dz1[g] = q.dz1; a.dcat1[g] = q.dcat1;
    if (!(q.dd0 && q.cat3 && q.mu && q.gmu && q.z1 && q.cat1 && q.fc1w && q.fc2inw && q.fc2outw && q.fc3w && q.dcat3 && q.dzmu &&
          q.dz1 && q.dcat1) || (reinterpret_cast<uintptr_t>(q.dd0) & 15)) {
      set_last_error("dense1d_bwd: null or unaligned pointer");
      return LSHM_ERR_ARG;
    }
  }
  a.ldmu = ldmu; a.ldgmu = ldgmu;
  a.B = B;
  if (L == LT) {
    hipLaunchKernelGGL(dense1d_bwd_kernel, dim3((B + 15) / 16, p1 ? 2 : 1), dim3(NTH), 0, st, a);
  } else {
    set_last_error("dense1d_bwd: latent width must be 16");
    return LSHM_ERR_UNSUPPORTED;
  }
  return check_launch("dense1d_bwd");
}

}  // namespace lshm
