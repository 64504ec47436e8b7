// Implicit-GEMM family on the exact-fp32 matrix cores (v_mfma_f32_16x16x4_f32).
//
// One kernel template serves every GEMM-shaped op of the autoencoders
// (reference: src/lofar_models.py:31-57,73-98 for the 2D AE, :115-142,158-183
// for the 1D AEs): k4s2p1 conv2d forward / data-gradient / weight-gradient (the
// transposed conv is the data-gradient kernel with the roles of the tensors
// swapped), k4s4 conv1d forward / data-gradient / weight-gradient, and the
// small dense layers.  Operand gathers, tile->tensor index maps and fused
// epilogues (bias, ELU, multiply by ELU' of a saved activation) are policy
// classes; the mainloop stages A and B tiles through LDS in bank-conflict-free
// layouts and feeds 16x16x4 f32 MFMAs, which are bit-for-bit an fmaf chain, so
// results match the fp32 CPU reference to rounding.
//
// 64-wide wavefronts: a 256-thread workgroup is 4 waves stacked along M; each
// wave owns (BM/64) x (BN/16) accumulator tiles.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <map>
#include <mutex>
#include <tuple>
#include <type_traits>

#include "kernels.h"

namespace lshm {

// --------------------------------------------------------------------------
// mainloop
// --------------------------------------------------------------------------
// Two problems of identical shape (the row- and column-vectorised 1-D autoencoders) can share one
// launch: grid.z carries a group index and each group has its own parameter block.
template <class P>
struct Pair {
  typename P::Params p[2];
  int zper;  // z-blocks per group (zgroups * splits)
};

// KW = 4 or 2 ("K over the wavefronts", BM = 64 / KW): KW wavefronts share each 16-row output tile and
// split the k-steps of every staged chunk between them; their partial tiles meet in LDS at the end and
// are added in wavefront order.  For the deep layers (a few thousand output rows, K in the hundreds or
// thousands) this gives many small tiles with short main loops in ONE launch, where split-K over
// workgroups needs a second launch to combine the slabs.
// BF: the operands are rounded to bf16 (nearest even) on their way into LDS and multiplied with
// v_mfma_f32_16x16x16_bf16 (one instruction per k-block instead of four, 8-byte operand reads); the
// accumulators, the epilogues and everything in HBM stay fp32.  Opt-in (MatrixPrecisionScope).
typedef short s16x4 __attribute__((ext_vector_type(4)));  // operand registers of v_mfma_f32_16x16x16_bf16
__device__ __forceinline__ unsigned bf16_bits(float v) {  // round to nearest even; NaN stays NaN
  const unsigned u = __float_as_uint(v);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
// Operands are fetched with raw buffer loads: one uniform resource per tensor (base pointer in SGPRs), a
// 32-bit byte offset per element and a scalar offset that steps through K.  An element that does not
// exist (padding, rows / columns past the edge) gets the offset kNoElem, which is beyond the resource's
// extent: the hardware returns 0 for it -- no predicates, no exec-mask branches, no 64-bit address
// arithmetic in the K loop (they were ~70 of its ~85 instructions per 8 matrix instructions).  Tensors are
// therefore limited to 4 GiB each (checked where the problems are built).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr unsigned kNoElem = 0xFFFFFFFFu;
__device__ __forceinline__ rsrc_t make_rsrc(const float* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)0xFFFFFFFFu, 0x00020000);
}
__device__ __forceinline__ float buf_ld(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ unsigned elem_off(bool ok, long elems) { return ok ? (unsigned)(elems << 2) : kNoElem; }

// One output tile (block column bx, block row by, z-block zblk) of problem p.
// p by value: the fields live in SGPRs instead of being re-read from the kernarg segment inside the K loop
template <class P, int BM, int BN, int BK, int KW, bool BF>
__device__ __forceinline__ void igemm_tile(const typename P::Params p, const int bx, const int by, const int zblk) {
  constexpr int NT = 256;
  static_assert(KW == 0 || ((KW == 2 || KW == 4) && BM * KW == 64), "KW wavefronts per 16-row tile");
  static_assert(BK % 16 == 0 && (KW == 0 || BK / 16 >= KW), "K chunks are whole 16-wide k-blocks");
  constexpr int WM = KW ? 4 / KW : 4;  // wavefronts along M
  constexpr int TM = KW ? 1 : BM / 64, TN = BN / 16;
  // LDS images, both operands: [k-block of 16][row][24], the 16 k of a row stored as 4 slots of 4 with
  // k = slot + 4e at slot*4 + e.  Lane (lm, lk) of an MFMA tile then gets its operand for the four
  // k-steps of a k-block (k = 4e + lk, e = 0..3) with ONE ds_read_b128 at row lm, slot lk: a quarter of
  // the LDS instructions (and waits) of dword reads at twice their bytes per clock.  Row pitch 24
  // floats makes the four 16-lane groups of ds_read_b128 conflict-free (slot (6 row + lk) mod 16); the
  // k-block pitch is 16 (mod 32) so that the dword stores of a k-fast operand spread over all banks.
  constexpr int LDK = 24;
  constexpr int NKB = BK / 16;
  constexpr int A_KBS = BM * LDK + 16, B_KBS = BN * LDK + 16;
  constexpr int A_ELEMS = NKB * A_KBS, B_ELEMS = NKB * B_KBS;
  // an m-fast (n-fast) operand is staged in float4 groups (row, k-block, slot): group G = t + 256 j
  constexpr int A_GROUPS = BM * BK / 4, B_GROUPS = BN * BK / 4;
  constexpr int NGA = (A_GROUPS + NT - 1) / NT, NGB = (B_GROUPS + NT - 1) / NT;
  // a k-fast operand in single elements: k = t % BK, rows t / BK + i * (NT / BK)
  constexpr int NA = P::A_M_FAST ? 4 * NGA : BM * BK / NT;
  constexpr int NB = P::B_N_FAST ? 4 * NGB : (BN + NT / BK - 1) / (NT / BK);
  constexpr int RED_ELEMS = KW ? 4 * TN * 64 * 4 : 0;  // cross-wave combine buffer (reuses the tile images)
  __shared__ __attribute__((aligned(16))) float smem[(A_ELEMS + B_ELEMS) > RED_ELEMS ? (A_ELEMS + B_ELEMS) : RED_ELEMS];
  float* As = smem;
  float* Bs = smem + A_ELEMS;
  // bf16 images: same [k-block][row][24] element layout with the 16 k of a row in natural order
  // (lane (lm, lk) reads k = 4 lk .. 4 lk + 3 as one 8-byte word; 48-byte rows are conflict-free for ds_read_b64)
  unsigned short* Ah = reinterpret_cast<unsigned short*>(smem);
  unsigned short* Bh = Ah + A_ELEMS;

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int lm = lane & 15, lk = lane >> 4;
  const int m0 = bx * BM, n0 = by * BN;
  // grid.z = zgroup (e.g. output parity of the transposed conv) x K split
  const int splits = p.sk.splits;
  const int zg = zblk / splits, split = zblk - zg * splits;
  const int kbeg = split * p.sk.kchunk;
  const int kend = min(p.K, kbeg + p.sk.kchunk);
  // kchunk is a multiple of BK: lets the compiler hoist the tap decode / bounds tests of the
  // gathers out of the K loop (they depend on k only through k % 16)
  __builtin_assume((kbeg & (BK - 1)) == 0);

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int wm0 = KW ? (wave % WM) * 16 : wave * (BM / 4);
  const int wk = KW ? wave / WM : 0;  // which share of the k-blocks

  float ra[NA], rb[NB];
  // "fast" operands are affine in the K-chunk index: element i of this thread sits at byte offset
  // voff[i] + chunk * step of its tensor, with tap decode and bounds tests done once, before the K loop
  const rsrc_t a_rs = make_rsrc(P::a_tensor(p)), b_rs = make_rsrc(P::b_tensor(p));
  unsigned avoff[NA], bvoff[NB];
  unsigned asoff = 0, bsoff = 0, astep = 0, bstep = 0;  // scalar byte offsets of the current chunk / per chunk
  typename P::FastA fa;
  typename P::FastB fb;
  // k (within the chunk) of element e of group g: 16 (g / 4) + (g % 4) + 4 e   (bf16: ... + 4 (g % 4) + e)
  auto kmap = [](int g, int e) { return BF ? 16 * (g >> 2) + 4 * (g & 3) + e : 16 * (g >> 2) + (g & 3) + 4 * e; };
  if constexpr (P::A_M_FAST) {
    static_assert(NT % BM == 0 && A_GROUPS % NT == 0, "every thread stages whole groups of one row");
    fa = P::a_fast(p, m0 + t % BM, zg);
    const long arel = P::a_base(p, fa) - P::a_tensor(p);
#pragma unroll
    for (int j = 0; j < NGA; ++j) {
      const int g = t / BM + j * (NT / BM);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bool ok;
        int off;
        P::a_affine(p, fa, m0 + t % BM, kmap(g, e), zg, off, ok);
        avoff[4 * j + e] = elem_off(ok, arel + off);
      }
    }
    astep = (unsigned)(P::a_step(p, BK) << 2);
    asoff = (unsigned)(kbeg / BK) * astep;
  }
  if constexpr (P::B_N_FAST) {
    const long brel = P::b_base(p) - P::b_tensor(p);
#pragma unroll
    for (int j = 0; j < NGB; ++j) {
      const int G = t + j * NT;
      const int n = G % BN, g = G / BN;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bool ok;
        int off;
        P::b_affine(p, P::b_fast(p, n0 + n, zg), n0 + n, kmap(g, e), zg, off, ok);
        bvoff[4 * j + e] = elem_off(ok && G < B_GROUPS, brel + off);
      }
    }
    bstep = (unsigned)(P::b_step(p, BK) << 2);
    bsoff = (unsigned)(kbeg / BK) * bstep;
  }
  // global -> registers for the chunk starting at k0 (issued one chunk ahead of its use); TAIL: the chunk is
  // cut short by the end of K (only possible in the last chunk of a K that is not a multiple of BK)
  auto fetch_t = [&](int k0, auto tail) {
    constexpr bool TAIL = decltype(tail)::value;
    const int krem = kend - k0;  // elements of this chunk that exist
    if constexpr (P::A_M_FAST) {
#pragma unroll
      for (int j = 0; j < NGA; ++j) {
        const int g = t / BM + j * (NT / BM);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = 4 * j + e;
          // (an element past the end of K must not be touched at all: its address may lie outside the tensor)
          ra[i] = buf_ld(a_rs, (!TAIL || kmap(g, e) < krem) ? avoff[i] : kNoElem, asoff);
        }
      }
      asoff += astep;
    } else {
      const int k = k0 + t % BK;
      const bool kok = !TAIL || k < kend;
      fa = P::a_fast(p, kok ? k : kbeg, zg);
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const unsigned vo = P::a_voff(p, fa, m0 + t / BK + i * (NT / BK), k, zg);
        ra[i] = buf_ld(a_rs, kok ? vo : kNoElem, 0u);
      }
    }
    if constexpr (P::B_N_FAST) {
#pragma unroll
      for (int j = 0; j < NGB; ++j) {
        const int g = (t + j * NT) / BN;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = 4 * j + e;
          rb[i] = buf_ld(b_rs, (!TAIL || kmap(g, e) < krem) ? bvoff[i] : kNoElem, bsoff);
        }
      }
      bsoff += bstep;
    } else {
      const int k = k0 + t % BK;
      const bool kok = !TAIL || k < kend;
      fb = P::b_fast(p, kok ? k : kbeg, zg);
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int nl = t / BK + i * (NT / BK);
        const unsigned vo = P::b_voff(p, fb, k, n0 + nl, zg);
        rb[i] = buf_ld(b_rs, (nl < BN && kok) ? vo : kNoElem, 0u);
      }
    }
  };
  auto fetch = [&](int k0) {
    if (kend - k0 >= BK) fetch_t(k0, std::false_type{});
    else fetch_t(k0, std::true_type{});
  };
  // position of this thread's k (k-fast operands) inside the image of one row
  const int kf = t % BK;
  const int kf_pos = BF ? (kf & 15) : (kf & 3) * 4 + ((kf & 15) >> 2);
  auto pack = [](float lo, float hi) { return bf16_bits(lo) | (bf16_bits(hi) << 16); };
  auto stage = [&]() {
    if constexpr (P::A_M_FAST) {
#pragma unroll
      for (int j = 0; j < NGA; ++j) {
        const int g = t / BM + j * (NT / BM);
        const int o = (g >> 2) * A_KBS + (t % BM) * LDK + 4 * (g & 3);
        if constexpr (BF)
          *reinterpret_cast<uint2*>(Ah + o) = make_uint2(pack(ra[4 * j], ra[4 * j + 1]), pack(ra[4 * j + 2], ra[4 * j + 3]));
        else
          *reinterpret_cast<f32x4*>(As + o) = (f32x4){ra[4 * j], ra[4 * j + 1], ra[4 * j + 2], ra[4 * j + 3]};
      }
    } else {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int o = (kf >> 4) * A_KBS + (t / BK + i * (NT / BK)) * LDK + kf_pos;
        if constexpr (BF) Ah[o] = (unsigned short)bf16_bits(ra[i]);
        else As[o] = ra[i];
      }
    }
    if constexpr (P::B_N_FAST) {
#pragma unroll
      for (int j = 0; j < NGB; ++j) {
        const int G = t + j * NT;
        const int n = G % BN, g = G / BN;
        const int o = (g >> 2) * B_KBS + n * LDK + 4 * (g & 3);
        if (G < B_GROUPS) {
          if constexpr (BF)
            *reinterpret_cast<uint2*>(Bh + o) = make_uint2(pack(rb[4 * j], rb[4 * j + 1]), pack(rb[4 * j + 2], rb[4 * j + 3]));
          else
            *reinterpret_cast<f32x4*>(Bs + o) = (f32x4){rb[4 * j], rb[4 * j + 1], rb[4 * j + 2], rb[4 * j + 3]};
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int nl = t / BK + i * (NT / BK);
        const int o = (kf >> 4) * B_KBS + nl * LDK + kf_pos;
        if (nl < BN) {
          if constexpr (BF) Bh[o] = (unsigned short)bf16_bits(rb[i]);
          else Bs[o] = rb[i];
        }
      }
    }
  };

  if (kbeg < kend) fetch(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    stage();
    __syncthreads();
    if (k0 + BK < kend) fetch(k0 + BK);  // loads stay in flight under the MFMAs below
#pragma unroll
    for (int kb = wk; kb < NKB; kb += KW ? KW : 1) {
      if constexpr (BF) {
        s16x4 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          a[i] = *reinterpret_cast<const s16x4*>(Ah + kb * A_KBS + (wm0 + 16 * i + lm) * LDK + 4 * lk);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          b[j] = *reinterpret_cast<const s16x4*>(Bh + kb * B_KBS + (16 * j + lm) * LDK + 4 * lk);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[i], b[j], acc[i][j], 0, 0, 0);
      } else {
        f32x4 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          a[i] = *reinterpret_cast<const f32x4*>(As + kb * A_KBS + (wm0 + 16 * i + lm) * LDK + 4 * lk);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          b[j] = *reinterpret_cast<const f32x4*>(Bs + kb * B_KBS + (16 * j + lm) * LDK + 4 * lk);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // ---- epilogue: lane holds rows 4*lk..4*lk+3 of column lm in each 16x16 tile
  if constexpr (KW) {
    // the KW partial tiles of each m-tile -> LDS -> share wk of the wavefronts finishes n-tiles j == wk
    // (mod KW), summing in share order
    f32x4* red = reinterpret_cast<f32x4*>(smem);
#pragma unroll
    for (int j = 0; j < TN; ++j) red[(wave * TN + j) * 64 + lane] = acc[0][j];
    __syncthreads();
    const int wmi = wave % WM;
    for (int j = wk; j < TN; j += KW) {
      f32x4 v = red[(wmi * TN + j) * 64 + lane];
#pragma unroll
      for (int w = 1; w < KW; ++w) v += red[((w * WM + wmi) * TN + j) * 64 + lane];
      const int m = m0 + wm0 + 4 * lk, n = n0 + 16 * j + lm;
      if (splits > 1) {
        const int Mp = (p.M + 3) & ~3;
        if (m < Mp && n < p.N) *reinterpret_cast<f32x4*>(p.sk.partial + ((long)zblk * p.N + n) * Mp + m) = v;
      } else {
        P::store(p, m, n, v, zg);
      }
    }
    return;
  }
  if (splits > 1) {
    // raw partial sums, layout [z][n][Mp] (Mp = M rounded up to 4); combined by splitk_epilogue_kernel
    const int Mp = (p.M + 3) & ~3;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int m = m0 + wm0 + 16 * i + 4 * lk, n = n0 + 16 * j + lm;
        if (m < Mp && n < p.N)
          *reinterpret_cast<f32x4*>(p.sk.partial + ((long)zblk * p.N + n) * Mp + m) = acc[i][j];
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
      P::store(p, m0 + wm0 + 16 * i + 4 * lk, n0 + 16 * j + lm, acc[i][j], zg);
}

template <class P, int BM, int BN, int BK, int KW = 0, bool BF = false>
__global__ __launch_bounds__(256) void igemm_kernel(const Pair<P> pp) {
  const int grp = blockIdx.z / pp.zper;
  const int zblk = blockIdx.z - grp * pp.zper;
  igemm_tile<P, BM, BN, BK, KW, BF>(pp.p[grp], blockIdx.x, blockIdx.y, zblk);
}

// Several independent problems of one policy in ONE launch (the weight gradients of the six dense layers of an
// autoencoder, times two for the netT / netF pair): each is a 5-10 us launch on its own, latency-bound, and
// nothing orders them among themselves.  Tiles are numbered consecutively over the problems.
constexpr int kMaxBatch = 12;
template <class P>
struct Batch {
  typename P::Params p[kMaxBatch];
  int first[kMaxBatch + 1];  // first tile of problem g; first[n] = all tiles
  int mt[kMaxBatch], nt[kMaxBatch];
  int n;
};
template <class P, int BM, int BN, int BK, int KW = 0, bool BF = false>
__global__ __launch_bounds__(256) void igemm_batch_kernel(const Batch<P> bp) {
  const int bid = blockIdx.x;
  int g = 0;
  for (int i = 1; i < bp.n; ++i) g += bid >= bp.first[i] ? 1 : 0;
  const int local = bid - bp.first[g];
  const int mt = bp.mt[g], nt = bp.nt[g];
  const int bx = local % mt, r = local / mt;
  igemm_tile<P, BM, BN, BK, KW, BF>(bp.p[g], bx, r % nt, r / nt);
}

// Second stage of a split-K launch: sums the partial slabs in a fixed order (bitwise
// reproducible) and applies the problem's own epilogue.  Block = OL outputs (groups of 4 rows)
// x SL split lanes; lanes are combined through LDS in lane order.
template <class P>
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const Pair<P> pp, int zgroups, int OL) {
  const typename P::Params& p = pp.p[blockIdx.y];
  __shared__ f32x4 red[256];
  const int SL = 256 / OL;
  const int ol = threadIdx.x % OL, sl = threadIdx.x / OL;
  const int Mp = (p.M + 3) & ~3, M4 = Mp >> 2;
  const long nout = (long)M4 * p.N * zgroups;
  const long o = (long)blockIdx.x * OL + ol;
  const int splits = p.sk.splits;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  int m = 0, n = 0, zg = 0;
  if (o < nout) {
    const int m4 = (int)(o % M4);
    const long r = o / M4;
    n = (int)(r % p.N);
    zg = (int)(r / p.N);
    m = 4 * m4;
    for (int s = sl; s < splits; s += SL)
      v += *reinterpret_cast<const f32x4*>(p.sk.partial + ((long)(zg * splits + s) * p.N + n) * Mp + m);
  }
  red[threadIdx.x] = v;
  __syncthreads();
  if (sl == 0 && o < nout) {
    for (int s = 1; s < SL; ++s) v += red[s * OL + ol];
    P::store(p, m, n, v, zg);
  }
}

// Division of an index by a kernel-uniform extent (rows per image, outputs per row ...).  Every layer of
// the 128 x 128 configuration has power-of-two extents, where this is a shift and a mask on a scalar
// condition; the generic sequence (~30 vector instructions) stays for everything else.  The gathers of
// the weight gradients decode their position once per K chunk, the epilogues once per output row.
struct UDiv { int d, sh; };
__device__ __forceinline__ UDiv udiv(int d) { return UDiv{d, (d & (d - 1)) == 0 ? 31 - __clz(d) : -1}; }
__device__ __forceinline__ int divmod(const UDiv u, int x, int& r) {
  int q;
  if (u.sh >= 0) {
    q = x >> u.sh;
    r = x & (u.d - 1);
  } else {
    q = x / u.d;
    r = x - q * u.d;
  }
  return q;
}

__device__ __forceinline__ float epi(float v, float bias, int act) {
  v += bias;
  return act ? elu(v) : v;
}

// --------------------------------------------------------------------------
// conv2d k4 s2 p1 forward:  y[b,co,oy,ox] = sum_{ci,ky,kx} w[co,ci,ky,kx] x[b,ci,2oy-1+ky,2ox-1+kx]
// (also the data-gradient of the transposed conv)
// --------------------------------------------------------------------------
struct Conv2dFwd {
  static constexpr bool A_M_FAST = true, B_N_FAST = false;
  static constexpr int ID = 0;  // stable key of the tuning cache
  using Params = Conv2dFwdParams;
  struct FastA { const float* base; int iy0, ix0; };
  struct FastB { int k; };
  __device__ static FastA a_fast(const Params& p, int m, int) {
    FastA f;
    if (m >= p.M) { f.base = nullptr; f.iy0 = f.ix0 = 0; return f; }
    int r, ox;
    const int b = divmod(udiv(p.Ho * p.Wo), m, r);
    const int oy = divmod(udiv(p.Wo), r, ox);
    f.base = p.x + (long)b * p.x_bs;
    f.iy0 = 2 * oy - 1;
    f.ix0 = 2 * ox - 1;
    return f;
  }
  __device__ static const float* a_tensor(const Params& p) { return p.x; }
  __device__ static const float* b_tensor(const Params& p) { return p.w; }
  // element (m, k = kl + chunk*BK): offset = (ci*H + iy)*W + ix, ci = kl/16 + chunk*BK/16
  __device__ static void a_affine(const Params& p, const FastA& f, int, int kl, int, int& off, bool& ok) {
    const int iy = f.iy0 + ((kl >> 2) & 3), ix = f.ix0 + (kl & 3);
    ok = f.base && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    off = ((kl >> 4) * p.H + iy) * p.W + ix;
  }
  __device__ static long a_step(const Params& p, int bk) { return (long)(bk / 16) * p.H * p.W; }
  __device__ static const float* a_base(const Params& p, const FastA& f) { return f.base ? f.base : p.x; }
  __device__ static FastB b_fast(const Params&, int k, int) { return FastB{k}; }
  __device__ static unsigned b_voff(const Params& p, const FastB&, int k, int n, int) {
    return elem_off(n < p.N, (long)n * p.K + k);
  }
  __device__ static void store(const Params& p, int m, int n, f32x4 v, int) {
    if (m >= p.M || n >= p.N) return;
    const int hw = p.Ho * p.Wo;
    int r;
    const int b = divmod(udiv(hw), m, r);
    const long idx = (long)b * p.y_bs + (long)n * hw + r;
    const float bias = p.bias ? p.bias[n] : 0.f;
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = epi(v[i], bias, p.act);
    if (p.dact) {
      const f32x4 s = *reinterpret_cast<const f32x4*>(p.dact + idx);
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] *= elu_grad_from_out(s[i]);
    }
    *reinterpret_cast<f32x4*>(p.y + idx) = o;
  }
};

// --------------------------------------------------------------------------
// conv2d k4 s2 p1 data-gradient == transposed-conv forward, one output parity
// (py,px) = (z>>1, z&1) per grid.z slice:
//   big[b,cb,2m+py,2n+px] = sum_{cs,t,u} small[b,cs,m+py-t,n+px-u] w[cs,cb,2t+1-py,2u+1-px]
// --------------------------------------------------------------------------
struct Conv2dDgrad {
  static constexpr bool A_M_FAST = true, B_N_FAST = true;
  static constexpr int ID = 1;  // stable key of the tuning cache
  using Params = Conv2dDgradParams;
  struct FastA { const float* base; int mm, nn; };
  struct FastB { int n; };
  __device__ static FastA a_fast(const Params& p, int m, int) {
    FastA f;
    if (m >= p.M) { f.base = nullptr; f.mm = f.nn = 0; return f; }
    int r;
    const int b = divmod(udiv(p.Hs * p.Ws), m, r);
    f.mm = divmod(udiv(p.Ws), r, f.nn);
    f.base = p.s + (long)b * p.s_bs;
    return f;
  }
  __device__ static const float* a_tensor(const Params& p) { return p.s; }
  __device__ static const float* b_tensor(const Params& p) { return p.w; }
  __device__ static void a_affine(const Params& p, const FastA& f, int, int kl, int z, int& off, bool& ok) {
    const int iy = f.mm + (z >> 1) - ((kl >> 1) & 1), ix = f.nn + (z & 1) - (kl & 1);
    ok = f.base && (unsigned)iy < (unsigned)p.Hs && (unsigned)ix < (unsigned)p.Ws;
    off = ((kl >> 2) * p.Hs + iy) * p.Ws + ix;
  }
  __device__ static long a_step(const Params& p, int bk) { return (long)(bk / 4) * p.Hs * p.Ws; }
  __device__ static const float* a_base(const Params& p, const FastA& f) { return f.base ? f.base : p.s; }
  __device__ static void b_affine(const Params& p, const FastB&, int n, int kl, int z, int& off, bool& ok) {
    const int ky = 2 * ((kl >> 1) & 1) + 1 - (z >> 1), kx = 2 * (kl & 1) + 1 - (z & 1);
    ok = n < p.N;
    off = (((kl >> 2) * p.Cb + n) * 4 + ky) * 4 + kx;
  }
  __device__ static long b_step(const Params& p, int bk) { return (long)(bk / 4) * p.Cb * 16; }
  __device__ static const float* b_base(const Params& p) { return p.w; }
  __device__ static FastB b_fast(const Params&, int n, int) { return FastB{n}; }
  __device__ static void store(const Params& p, int m, int n, f32x4 v, int z) {
    if (n >= p.N) return;
    const int hw = p.Hs * p.Ws;
    const float bias = p.bias ? p.bias[n] : 0.f;
    const int Hb = 2 * p.Hs, Wb = 2 * p.Ws;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int mr = m + i;
      if (mr >= p.M) break;
      int r, nn;
      const int b = divmod(udiv(hw), mr, r);
      const int mm = divmod(udiv(p.Ws), r, nn);
      const long idx = (long)b * p.big_bs + ((long)n * Hb + 2 * mm + (z >> 1)) * Wb + 2 * nn + (z & 1);
      float o = epi(v[i], bias, p.act);
      if (p.dact) o *= elu_grad_from_out(p.dact[idx]);
      p.big[idx] = o;
    }
  }
};

// --------------------------------------------------------------------------
// conv2d k4 s2 p1 weight-gradient (split-K over grid.z, deterministic partials):
//   dW[cs,cb,ky,kx] = sum_{b,oy,ox} small[b,cs,oy,ox] big[b,cb,2oy-1+ky,2ox-1+kx]
// --------------------------------------------------------------------------
struct Conv2dWgrad {
  static constexpr bool A_M_FAST = false, B_N_FAST = false;
  static constexpr int ID = 2;  // stable key of the tuning cache
  using Params = Conv2dWgradParams;
  struct FastA { long off; };                 // element offset of (image, position) inside `s`
  struct FastB { long off; int iy0, ix0; };   // element offset of the image inside `big`, window origin
  __device__ static const float* a_tensor(const Params& p) { return p.s; }
  __device__ static const float* b_tensor(const Params& p) { return p.big; }
  __device__ static FastA a_fast(const Params& p, int k, int) {
    int r;
    const int b = divmod(udiv(p.Hs * p.Ws), k, r);
    return FastA{(long)b * p.s_bs + r};
  }
  __device__ static unsigned a_voff(const Params& p, const FastA& f, int m, int, int) {
    return elem_off(m < p.M, f.off + (long)m * p.Hs * p.Ws);
  }
  __device__ static FastB b_fast(const Params& p, int k, int) {
    int r, ox;
    const int b = divmod(udiv(p.Hs * p.Ws), k, r);
    const int oy = divmod(udiv(p.Ws), r, ox);
    return FastB{(long)b * p.big_bs, 2 * oy - 1, 2 * ox - 1};
  }
  __device__ static unsigned b_voff(const Params& p, const FastB& f, int, int n, int) {
    const int cb = n >> 4, iy = f.iy0 + ((n >> 2) & 3), ix = f.ix0 + (n & 3);
    const int Hb = 2 * p.Hs, Wb = 2 * p.Ws;
    const bool ok = n < p.N && (unsigned)iy < (unsigned)Hb && (unsigned)ix < (unsigned)Wb;
    return elem_off(ok, f.off + (long)(cb * Hb + iy) * Wb + ix);
  }
  __device__ static void store(const Params& p, int m, int n, f32x4 v, int) {
    if (n >= p.N) return;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (m + i < p.M) {
        float* d = p.dw + (long)(m + i) * p.N + n;
        *d = p.accumulate ? *d + v[i] : v[i];
      }
  }
};

// --------------------------------------------------------------------------
// conv1d k4 s4 forward with left padding `pad`:
//   y[b,co,j] = sum_{ci,t} w[co,ci,t] x[b,ci,4j-pad+t]
// (pad=1: the encoder conv; pad=0: data-gradient of the k4 s4 p0 transposed conv)
// --------------------------------------------------------------------------
struct Conv1dFwd {
  static constexpr bool A_M_FAST = true, B_N_FAST = false;
  static constexpr int ID = 3;  // stable key of the tuning cache
  using Params = Conv1dFwdParams;
  struct FastA { const float* base; int j0; };
  struct FastB { int k; };
  __device__ static FastA a_fast(const Params& p, int m, int) {
    if (m >= p.M) return FastA{nullptr, 0};
    int j;
    const int b = divmod(udiv(p.Lo), m, j);
    return FastA{p.x + (long)b * p.x_bs, 4 * j - p.pad};
  }
  __device__ static void a_affine(const Params& p, const FastA& f, int, int kl, int, int& off, bool& ok) {
    const int pos = f.j0 + (kl & 3);
    ok = f.base && (unsigned)pos < (unsigned)p.L;
    off = (kl >> 2) * p.L + pos;
  }
  __device__ static long a_step(const Params& p, int bk) { return (long)(bk / 4) * p.L; }
  __device__ static const float* a_base(const Params& p, const FastA& f) { return f.base ? f.base : p.x; }
  __device__ static const float* a_tensor(const Params& p) { return p.x; }
  __device__ static const float* b_tensor(const Params& p) { return p.w; }
  __device__ static FastB b_fast(const Params&, int k, int) { return FastB{k}; }
  __device__ static unsigned b_voff(const Params& p, const FastB&, int k, int n, int) {
    return elem_off(n < p.N, (long)n * p.K + k);
  }
  __device__ static void store(const Params& p, int m, int n, f32x4 v, int) {
    if (m >= p.M || n >= p.N) return;
    int j;
    const int b = divmod(udiv(p.Lo), m, j);
    const long idx = (long)b * p.y_bs + (long)n * p.Lo + j;
    const float bias = p.bias ? p.bias[n] : 0.f;
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = epi(v[i], bias, p.act);
    if (p.dact) {
      const f32x4 s = *reinterpret_cast<const f32x4*>(p.dact + idx);
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] *= elu_grad_from_out(s[i]);
    }
    *reinterpret_cast<f32x4*>(p.y + idx) = o;
  }
};

// --------------------------------------------------------------------------
// k4 s4 transposed conv1d forward / conv1d data-gradient:
//   big[b,cb,4i+t-pad] = sum_cs small[b,cs,i] w[cs,cb,t]      (N = Cb*4, K = Cs)
// positions of `big` no window reaches (the last `pad` ones) are written as zero.
// --------------------------------------------------------------------------
struct Conv1dDgrad {
  static constexpr bool A_M_FAST = true, B_N_FAST = true;
  static constexpr int ID = 4;  // stable key of the tuning cache
  using Params = Conv1dDgradParams;
  struct FastA { const float* base; };
  struct FastB { int n; };
  __device__ static FastA a_fast(const Params& p, int m, int) {
    if (m >= p.M) return FastA{nullptr};
    int i;
    const int b = divmod(udiv(p.Ls), m, i);
    return FastA{p.s + (long)b * p.s_bs + i};
  }
  __device__ static void a_affine(const Params& p, const FastA& f, int, int kl, int, int& off, bool& ok) {
    ok = f.base != nullptr;
    off = kl * p.Ls;
  }
  __device__ static long a_step(const Params& p, int bk) { return (long)bk * p.Ls; }
  __device__ static const float* a_base(const Params& p, const FastA& f) { return f.base ? f.base : p.s; }
  __device__ static void b_affine(const Params& p, const FastB&, int n, int kl, int, int& off, bool& ok) {
    ok = n < p.N;
    off = kl * p.N + n;
  }
  __device__ static long b_step(const Params& p, int bk) { return (long)bk * p.N; }
  __device__ static const float* b_base(const Params& p) { return p.w; }
  __device__ static const float* a_tensor(const Params& p) { return p.s; }
  __device__ static const float* b_tensor(const Params& p) { return p.w; }
  __device__ static FastB b_fast(const Params&, int n, int) { return FastB{n}; }
  __device__ static void store(const Params& p, int m, int n, f32x4 v, int) {
    if (n >= p.N) return;
    const int cb = n >> 2, tt = n & 3;
    const float bias = p.bias ? p.bias[cb] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int mr = m + r;
      if (mr >= p.M) break;
      int i;
      const int b = divmod(udiv(p.Ls), mr, i);
      const long row = (long)b * p.big_bs + (long)cb * p.Lb;
      const int pos = 4 * i + tt - p.pad;
      if ((unsigned)pos < (unsigned)p.Lb) {
        float o = epi(v[r], bias, p.act);
        if (p.dact) o *= elu_grad_from_out(p.dact[row + pos]);
        p.big[row + pos] = o;
      }
      if (i == p.Ls - 1)
        for (int q = 4 * p.Ls - p.pad + tt; q < p.Lb; q += 4) p.big[row + q] = 0.f;
    }
  }
};

// --------------------------------------------------------------------------
// k4 s4 conv1d weight-gradient (split-K):
//   dW[cs,cb,t] = sum_{b,i} small[b,cs,i] big[b,cb,4i+t-pad]
// --------------------------------------------------------------------------
struct Conv1dWgrad {
  static constexpr bool A_M_FAST = false, B_N_FAST = false;
  static constexpr int ID = 5;  // stable key of the tuning cache
  using Params = Conv1dWgradParams;
  struct FastA { long off; };
  struct FastB { long off; int pos0; };
  __device__ static const float* a_tensor(const Params& p) { return p.s; }
  __device__ static const float* b_tensor(const Params& p) { return p.big; }
  __device__ static FastA a_fast(const Params& p, int k, int) {
    int i;
    const int b = divmod(udiv(p.Ls), k, i);
    return FastA{(long)b * p.s_bs + i};
  }
  __device__ static unsigned a_voff(const Params& p, const FastA& f, int m, int, int) {
    return elem_off(m < p.M, f.off + (long)m * p.Ls);
  }
  __device__ static FastB b_fast(const Params& p, int k, int) {
    int i;
    const int b = divmod(udiv(p.Ls), k, i);
    return FastB{(long)b * p.big_bs, 4 * i - p.pad};
  }
  __device__ static unsigned b_voff(const Params& p, const FastB& f, int, int n, int) {
    const int cb = n >> 2, pos = f.pos0 + (n & 3);
    return elem_off(n < p.N && (unsigned)pos < (unsigned)p.Lb, f.off + (long)cb * p.Lb + pos);
  }
  __device__ static void store(const Params& p, int m, int n, f32x4 v, int) {
    if (n >= p.N) return;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (m + i < p.M) {
        float* d = p.dw + (long)(m + i) * p.N + n;
        *d = p.accumulate ? *d + v[i] : v[i];
      }
  }
};

// --------------------------------------------------------------------------
// generic strided GEMM (dense layers):  C[m,n] = act(sum_k A[m,k] B[k,n] + bias[n]) * dact'
// --------------------------------------------------------------------------
template <bool AMF, bool BNF>
struct Strided {
  static constexpr bool A_M_FAST = AMF, B_N_FAST = BNF;
  static constexpr int ID = 6 + 2 * (AMF ? 1 : 0) + (BNF ? 1 : 0);
  using Params = StridedGemmParams;
  struct FastA { int i; };
  struct FastB { int i; };
  __device__ static FastA a_fast(const Params&, int i, int) { return FastA{i}; }
  __device__ static FastB b_fast(const Params&, int i, int) { return FastB{i}; }
  __device__ static void a_affine(const Params& p, const FastA&, int m, int kl, int, int& off, bool& ok) {
    ok = m < p.M;
    off = (int)(m * p.sam + kl * p.sak);
  }
  __device__ static long a_step(const Params& p, int bk) { return (long)bk * p.sak; }
  __device__ static const float* a_base(const Params& p, const FastA&) { return p.a; }
  // column index N (one past the real columns) is a virtual column of ones when rowsum is requested:
  // C[m][N] = sum_k A[m][k], the bias gradient of a dense layer for free
  __device__ static void b_affine(const Params& p, const FastB&, int n, int kl, int, int& off, bool& ok) {
    ok = n < p.N;
    off = (int)(kl * p.sbk + n * p.sbn);
  }
  __device__ static long b_step(const Params& p, int bk) { return (long)bk * p.sbk; }
  __device__ static const float* b_base(const Params& p) { return p.b; }
  __device__ static const float* a_tensor(const Params& p) { return p.a; }
  __device__ static const float* b_tensor(const Params& p) { return p.b; }
  __device__ static unsigned a_voff(const Params& p, const FastA&, int m, int k, int) {
    return elem_off(m < p.M, (long)m * p.sam + (long)k * p.sak);
  }
  __device__ static unsigned b_voff(const Params& p, const FastB&, int k, int n, int) {
    return elem_off(n < p.N, (long)k * p.sbk + (long)n * p.sbn);
  }
  __device__ static void store(const Params& p, int m, int n, f32x4 v, int) {
    if (n >= p.N) return;
    const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (m + i >= p.M) break;
      const long idx = (long)(m + i) * p.scm + (long)n * p.scn;
      float o = epi(v[i], bias, p.act);
      if (p.add && n < p.add_n) o += p.add[(long)(m + i) * p.sxm + n];
      if (p.dact) o *= elu_grad_from_out(p.dact[(long)(m + i) * p.sdm + (long)n * p.sdn]);
      p.c[idx] = o;
    }
  }
};

// Output description of a weight-gradient problem for a deferred split-K combine (see GradJobs);
// false: this problem's epilogue cannot be expressed as a plain sum and must run in place.
static bool describe_output(const Conv2dWgradParams& p, SumJob& J) {
  J.dst = p.dw; J.rs = p.N; J.cs = 1; J.accumulate = p.accumulate;
  return true;
}
static bool describe_output(const Conv1dWgradParams& p, SumJob& J) {
  J.dst = p.dw; J.rs = p.N; J.cs = 1; J.accumulate = p.accumulate;
  return true;
}
static bool describe_output(const StridedGemmParams& p, SumJob& J) {
  if (p.bias || p.act || p.add || p.dact) return false;
  J.dst = p.c; J.rs = (int)p.scm; J.cs = (int)p.scn; J.accumulate = 0;
  return true;
}
template <class T>
static bool describe_output(const T&, SumJob&) { return false; }

// --------------------------------------------------------------------------
// launch helpers
// --------------------------------------------------------------------------
// Split-K plan: when the output tiling alone cannot fill the chip (deep layers: few output
// positions, long K; weight gradients: tiny outputs, K = B*H*W), grid.z also splits K and a
// second kernel combines the slabs.  Needs workspace; without one the launch is unsplit.
struct SplitPlan { int splits, kchunk; };
// target = workgroups the split aims for; 0 disables splitting
static SplitPlan plan_split(long tiles, int K, int M, int N, int zgroups, size_t ws_floats, int BK, long target) {
  SplitPlan sp{1, (K + BK - 1) / BK * BK};
  // measured at B=256: with >= ~200 output tiles the second launch costs more than the split gains
  if (target == 0 || tiles >= 200 || K <= 64 || ws_floats == 0) return sp;
  long want = (target + tiles - 1) / tiles;
  const long maxs = K / 64;  // at least 64 K elements per split
  if (want > maxs) want = maxs;
  const long Mp = (M + 3) & ~3;
  const long per = Mp * (long)N * zgroups;
  if (per * want > (long)ws_floats) want = (long)ws_floats / per;
  if (want <= 1) return sp;
  int kc = (int)((K + want - 1) / want);
  kc = (kc + BK - 1) / BK * BK;
  sp.kchunk = kc;
  sp.splits = (K + kc - 1) / kc;
  if (sp.splits <= 1) { sp.splits = 1; sp.kchunk = (K + BK - 1) / BK * BK; }
  return sp;
}

static const long kSplitTargets[3] = {768, 0, 1536};
static thread_local int t_matrix_bf16 = 0;
MatrixPrecisionScope::MatrixPrecisionScope(int bf16) : prev(t_matrix_bf16) { t_matrix_bf16 = bf16 ? 1 : 0; }
MatrixPrecisionScope::~MatrixPrecisionScope() { t_matrix_bf16 = prev; }
int igemm_matrix_precision() { return t_matrix_bf16; }

template <class P, int BM, int BN, int BK, int KW = 0>
static int launch_cfg(const typename P::Params& p0, const typename P::Params* p1, int M, int N, int Z,
                      float* ws, size_t wsf, int smode, hipStream_t st, GradJobs* defer) {
  const int G = p1 ? 2 : 1;
  const size_t wsg = ws ? (wsf / G) & ~(size_t)63 : 0;  // split-K scratch per group (256-byte aligned)
  const long tiles = (long)cdiv(M, BM) * cdiv(N, BN) * Z * G;
  const SplitPlan sp = plan_split(tiles, p0.K, M, N, Z, wsg, BK, kSplitTargets[smode]);
  Pair<P> pp;
  pp.p[0] = p0;
  pp.p[1] = p1 ? *p1 : p0;
  for (int g = 0; g < 2; ++g) {
    pp.p[g].sk.partial = ws ? ws + (size_t)g * wsg : nullptr;
    pp.p[g].sk.splits = sp.splits;
    pp.p[g].sk.kchunk = sp.kchunk;
  }
  pp.zper = Z * sp.splits;
  dim3 grid(cdiv(M, BM), cdiv(N, BN), Z * sp.splits * G);
  // a tile configuration whose LDS image (up to 74 KB for 128-row tiles with 64-deep K chunks) or register
  // budget does not fit this device is an error code here, not a failed dispatch
  int rc = t_matrix_bf16 ? kernel_budget_ok(reinterpret_cast<const void*>(&igemm_kernel<P, BM, BN, BK, KW, true>), 256, 0, "igemm tile configuration")
                         : kernel_budget_ok(reinterpret_cast<const void*>(&igemm_kernel<P, BM, BN, BK, KW, false>), 256, 0, "igemm tile configuration");
  if (rc) return rc;
  if (t_matrix_bf16)
    hipLaunchKernelGGL((igemm_kernel<P, BM, BN, BK, KW, true>), grid, dim3(256), 0, st, pp);
  else
    hipLaunchKernelGGL((igemm_kernel<P, BM, BN, BK, KW, false>), grid, dim3(256), 0, st, pp);
  rc = check_launch("igemm");
  if (rc || sp.splits == 1) return rc;
  if (defer && Z == 1) {  // leave the slabs where they are; the combine joins the backward's job list
    SumJob J[2];
    bool ok = true;
    for (int g = 0; g < G; ++g) {
      const long Mp = (M + 3) & ~3;
      J[g] = SumJob{pp.p[g].sk.partial, nullptr, (long)N * Mp, (int)(N * Mp), sp.splits, (int)Mp, M, 0, 0, 0, 0};
      ok = ok && describe_output(pp.p[g], J[g]);
    }
    if (ok) {
      for (int g = 0; g < G; ++g) defer->sums.push_back(J[g]);
      return rc;
    }
  }
  const long nout = (long)((M + 3) / 4) * N * Z;
  const int OL = nout >= 16384 ? 64 : 16;
  hipLaunchKernelGGL((splitk_epilogue_kernel<P>), dim3(cdiv(nout, OL), G), dim3(256), 0, st, pp, Z, OL);
  return check_launch("splitk_epilogue");
}

// choose the N tile from the real N so padding waste stays small
template <class P, int BM, int BK, int KW = 0>
static int launch_by_n(const typename P::Params& p, const typename P::Params* p1, int M, int N, int Z,
                       float* ws, size_t wsf, int smode, hipStream_t st, GradJobs* defer) {
  if (N <= 16) return launch_cfg<P, BM, 16, BK, KW>(p, p1, M, N, Z, ws, wsf, smode, st, defer);
  if (N <= 32) return launch_cfg<P, BM, 32, BK, KW>(p, p1, M, N, Z, ws, wsf, smode, st, defer);
  if (N <= 48 || (N % 48 == 0 && N % 64 != 0))
    return launch_cfg<P, BM, 48, BK, KW>(p, p1, M, N, Z, ws, wsf, smode, st, defer);
  return launch_cfg<P, BM, 64, BK, KW>(p, p1, M, N, Z, ws, wsf, smode, st, defer);
}

// A launch configuration is (M tile 64|128) x (K chunk 16|32) x (split-K mode): index bits 0, 1, 2..3;
// 12, 13: the K-over-wavefronts form with 16-row tiles (four wavefronts per tile, K chunk 64) without /
// with split-K; 14, 15: the same with 32-row tiles (two wavefronts per 16 rows).
// 16..21: deep K chunks without the K-over-wavefronts split (64 / 128 / 64 elements of K per barrier for
// 64- / 64- / 128-row tiles; even: unsplit, odd: split-K towards 768 workgroups): the mid layers are bound by
// the chain of dependent global loads, one per K chunk, so fewer, fatter chunks keep more loads in flight.
constexpr int kNumConfigs = 22;
template <class P>
static int launch_idx(int c, const typename P::Params& p, const typename P::Params* p1, int M, int N, int Z,
                      float* ws, size_t wsf, hipStream_t st, GradJobs* defer = nullptr) {
  if (c == 12) return launch_by_n<P, 16, 64, 4>(p, p1, M, N, Z, ws, wsf, 1, st, defer);
  if (c == 13) return launch_by_n<P, 16, 64, 4>(p, p1, M, N, Z, ws, wsf, 0, st, defer);
  if (c == 14) return launch_by_n<P, 32, 64, 2>(p, p1, M, N, Z, ws, wsf, 1, st, defer);
  if (c == 15) return launch_by_n<P, 32, 64, 2>(p, p1, M, N, Z, ws, wsf, 0, st, defer);
  if (c == 16) return launch_by_n<P, 64, 64>(p, p1, M, N, Z, ws, wsf, 1, st, defer);
  if (c == 17) return launch_by_n<P, 64, 64>(p, p1, M, N, Z, ws, wsf, 0, st, defer);
  if (c == 18) return launch_by_n<P, 64, 128>(p, p1, M, N, Z, ws, wsf, 1, st, defer);
  if (c == 19) return launch_by_n<P, 64, 128>(p, p1, M, N, Z, ws, wsf, 0, st, defer);
  if (c == 20) return launch_by_n<P, 128, 64>(p, p1, M, N, Z, ws, wsf, 1, st, defer);
  if (c == 21) return launch_by_n<P, 128, 64>(p, p1, M, N, Z, ws, wsf, 0, st, defer);
  const int smode = c >> 2;
  switch (c & 3) {
    case 0: return launch_by_n<P, 64, 16>(p, p1, M, N, Z, ws, wsf, smode, st, defer);
    case 1: return launch_by_n<P, 128, 16>(p, p1, M, N, Z, ws, wsf, smode, st, defer);
    case 2: return launch_by_n<P, 64, 32>(p, p1, M, N, Z, ws, wsf, smode, st, defer);
    default: return launch_by_n<P, 128, 32>(p, p1, M, N, Z, ws, wsf, smode, st, defer);
  }
}

// --------------------------------------------------------------------------
// per-shape autotuning: the first eager call of a (policy, M, N, K, Z, groups) shape times every
// configuration on the caller's own buffers (outputs are overwritten, so re-running is harmless)
// and caches the winner for the life of the process.  Calls that accumulate into their output or
// arrive while the stream is being captured take the static heuristic and leave the cache alone.
// --------------------------------------------------------------------------
struct TuneKey {
  int pol, M, N, K, Z, G, prec;  // prec: operand precision the entry was measured with (0 fp32, 1 bf16)
  bool operator<(const TuneKey& o) const {
    return std::tie(pol, M, N, K, Z, G, prec) < std::tie(o.pol, o.M, o.N, o.K, o.Z, o.G, o.prec);
  }
};
static std::mutex g_tune_mu;
static std::map<TuneKey, int> g_tuned;
static int g_tune_mode = -1;   // -1: read LSHM_TUNE on first use; 0: table, else static heuristic (default); 1: table, else time
static int g_tune_force = -1;  // >= 0: every launch uses this configuration (tests)
template <class P>
static int policy_id() { return P::ID; }
template <class T>
static auto accumulates(const T& p, int) -> decltype(p.accumulate, true) { return p.accumulate != 0; }
template <class T>
static bool accumulates(const T&, long) { return false; }

// Cache <-> text, one "policy M N K Z groups config [precision]" line per shape: lets a process start with
// the configurations measured earlier (same results bit for bit from run to run, no timing launches).
size_t igemm_tuning_export(char* buf, size_t cap) {
  std::lock_guard<std::mutex> lk(g_tune_mu);
  std::string out;
  char line[96];
  for (const auto& kv : g_tuned) {
    const TuneKey& k = kv.first;
    if (k.prec) snprintf(line, sizeof line, "%d %d %d %d %d %d %d %d\n", k.pol, k.M, k.N, k.K, k.Z, k.G, kv.second, k.prec);
    else snprintf(line, sizeof line, "%d %d %d %d %d %d %d\n", k.pol, k.M, k.N, k.K, k.Z, k.G, kv.second);
    out += line;
  }
  if (buf && cap > 0) {
    const size_t n = out.size() < cap - 1 ? out.size() : cap - 1;
    memcpy(buf, out.data(), n);
    buf[n] = 0;
  }
  return out.size() + 1;
}
int igemm_tuning_import(const char* text) {
  if (!text) return 0;
  std::lock_guard<std::mutex> lk(g_tune_mu);
  int n = 0;
  const char* p = text;
  while (*p) {
    TuneKey k;
    k.prec = 0;
    int cfg = 0, used = 0;
    if (sscanf(p, "%d %d %d %d %d %d %d%n", &k.pol, &k.M, &k.N, &k.K, &k.Z, &k.G, &cfg, &used) == 7 && cfg >= 0 &&
        cfg < kNumConfigs) {
      p += used;
      int prec = 0, used2 = 0;  // optional 8th field on the same line
      const char* q = p;
      while (*q == ' ' || *q == '\t') ++q;
      if (*q >= '0' && *q <= '9' && sscanf(q, "%d%n", &prec, &used2) == 1) { k.prec = prec ? 1 : 0; p = q + used2; }
      g_tuned[k] = cfg;
      ++n;
    }
    while (*p && *p != '\n') ++p;
    if (*p == '\n') ++p;
  }
  return n;
}

void igemm_set_tuning(int mode, int force) {
  std::lock_guard<std::mutex> lk(g_tune_mu);
  g_tune_mode = mode;
  g_tune_force = force;
  if (mode == 1) g_tuned.clear();  // re-measure; modes 0 / force keep the imported table
}

template <class P>
static int launch_auto(const typename P::Params& p, const typename P::Params* p1, int M, int N, int Z,
                       float* ws, size_t wsf, hipStream_t st, GradJobs* defer = nullptr) {
  // static heuristic: small-M problems (deep layers, weight gradients) get 64-row tiles for more workgroups
  const int heuristic = (M <= 64 || (long)cdiv(M, 128) * cdiv(N, 64) * Z < 256) ? 0 : 1;
  int mode, force;
  {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    if (g_tune_mode < 0) {
      // timing launches synchronise inside the call and make the choice (hence the summation order) depend on
      // wall time: opt-in only (LSHM_TUNE=1, lshm_set_tuning(1, -1), bench.py --save-tuning)
      const char* e = getenv("LSHM_TUNE");
      g_tune_mode = e ? atoi(e) != 0 : 0;
    }
    mode = g_tune_mode;
    force = g_tune_force;
  }
  if (force >= 0) return launch_idx<P>(force % kNumConfigs, p, p1, M, N, Z, ws, wsf, st, defer);
  const TuneKey key{policy_id<P>(), M, N, p.K, Z, p1 ? 2 : 1, t_matrix_bf16};
  {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    auto it = g_tuned.find(key);
    if (it != g_tuned.end()) return launch_idx<P>(it->second, p, p1, M, N, Z, ws, wsf, st, defer);
  }
  if (!mode) return launch_idx<P>(heuristic, p, p1, M, N, Z, ws, wsf, st, defer);
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(st, &cap);
  if (cap != hipStreamCaptureStatusNone || accumulates(p, 0) || (p1 && accumulates(*p1, 0)))
    return launch_idx<P>(heuristic, p, p1, M, N, Z, ws, wsf, st, defer);
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)
    return launch_idx<P>(heuristic, p, p1, M, N, Z, ws, wsf, st, defer);
  int best = heuristic;
  float best_ms = 1e30f;
  static const int reps = getenv("LSHM_TUNE_REPS") ? std::max(1, atoi(getenv("LSHM_TUNE_REPS"))) : 3;
  static const bool verbose = getenv("LSHM_TUNE_LOG") && atoi(getenv("LSHM_TUNE_LOG")) > 1;
  for (int c = 0; c < kNumConfigs; ++c) {
    if (c < 12 && (c & 1) && M <= 64) continue;
    if (c >= 20 && M <= 64) continue;
    if (c >= 12 && c < 16 && (long)cdiv(M, 16) * Z > 65535) continue;  // 16-row tiles: keep grid.x sane, these are for small M
    int rc = launch_idx<P>(c, p, p1, M, N, Z, ws, wsf, st);  // warm
    if (rc) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return rc; }
    (void)hipEventRecord(e0, st);
    for (int r = 0; r < reps; ++r) launch_idx<P>(c, p, p1, M, N, Z, ws, wsf, st);
    (void)hipEventRecord(e1, st);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best_ms) { best_ms = ms; best = c; }
    if (verbose) fprintf(stderr, "[lshm tune]   pol %d M=%d N=%d K=%d Z=%d G=%d cfg %2d: %.1f us\n", key.pol, M, N, p.K, Z,
                         key.G, c, ms * 1000.f / reps);
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    g_tuned[key] = best;
  }
  if (getenv("LSHM_TUNE_LOG"))
    fprintf(stderr, "[lshm tune] policy %d M=%d N=%d K=%d Z=%d G=%d -> cfg %d (BM %d BK %d split-mode %d) %.1f us\n",
            key.pol, M, N, p.K, Z, key.G, best, best >= 14 ? 32 : best >= 12 ? 16 : (best & 1) ? 128 : 64,
            best >= 12 ? 64 : (best & 2) ? 32 : 16, best >= 12 ? 1 - (best & 1) : best >> 2,
            best_ms * 1000.f / reps);
  return launch_idx<P>(best, p, p1, M, N, Z, ws, wsf, st, defer);  // candidates ran their combine in place
}

// Weight gradients of dense layers, dW[N_out, K_in] = dz^T x, as one launch (16-row tiles, K over the four
// wavefronts, no split over workgroups: K is the batch size).
static bool fits32(long elems_a, long elems_b);
int strided_gemm_batch_mm(const StridedGemmParams* probs, int n, hipStream_t st) {
  using P = Strided<true, true>;
  for (int g = 0; g < n; ++g) {
    const StridedGemmParams& q = probs[g];
    if (!fits32(labs(q.sam) * (q.M - 1) + labs(q.sak) * (q.K - 1) + 1, labs(q.sbk) * (q.K - 1) + labs(q.sbn) * (q.N - 1) + 1))
      return LSHM_ERR_UNSUPPORTED;
  }
  constexpr int BM = 16, BN = 64, BK = 64, KW = 4;
  for (int base = 0; base < n; base += kMaxBatch) {
    const int cnt = n - base < kMaxBatch ? n - base : kMaxBatch;
    Batch<P> bp;
    int tiles = 0;
    for (int g = 0; g < kMaxBatch; ++g) {
      bp.p[g] = probs[base + (g < cnt ? g : 0)];
      bp.p[g].sk.partial = nullptr;
      bp.p[g].sk.splits = 1;
      bp.p[g].sk.kchunk = (bp.p[g].K + BK - 1) / BK * BK;
      bp.first[g] = tiles;
      bp.mt[g] = cdiv(bp.p[g].M, BM);
      bp.nt[g] = cdiv(bp.p[g].N, BN);
      if (g < cnt) tiles += bp.mt[g] * bp.nt[g];
    }
    bp.first[kMaxBatch] = tiles;
    for (int g = cnt; g < kMaxBatch; ++g) bp.first[g] = tiles;
    bp.n = cnt;
    if (t_matrix_bf16)
      hipLaunchKernelGGL((igemm_batch_kernel<P, BM, BN, BK, KW, true>), dim3(tiles), dim3(256), 0, st, bp);
    else
      hipLaunchKernelGGL((igemm_batch_kernel<P, BM, BN, BK, KW, false>), dim3(tiles), dim3(256), 0, st, bp);
    const int rc = check_launch("igemm_batch");
    if (rc) return rc;
  }
  return LSHM_OK;
}

// The weight gradients of the deep 2-D layers (tconv2..0, conv5..3: 15-30 us launches each, a few workgroup waves of
// latency-bound K loops, one after the other on the weight-gradient stream) as ONE launch: 64 x 64 tiles, K split so that
// every problem brings a few hundred workgroups, slabs summed by the backward's job list like any other split.
int grad_jobs_launch_conv(GradJobs& jobs, hipStream_t st) {
  using P = Conv2dWgrad;
  constexpr int BM = 64, BN = 64, BK = 16;
  const int n = (int)jobs.conv2d.size();
  if (n == 0) return LSHM_OK;
  const long target = 2304 / n > 192 ? 2304 / n : 192;  // workgroups per problem (~9 per CU in all)
  for (int base = 0; base < n; base += kMaxBatch) {
    const int cnt = n - base < kMaxBatch ? n - base : kMaxBatch;
    Batch<P> bp;
    int tiles = 0;
    for (int g = 0; g < kMaxBatch; ++g) {
      const GradJobs::ParkedConv2d& q = jobs.conv2d[base + (g < cnt ? g : 0)];
      bp.p[g] = q.p;
      const int M = q.p.M, N = q.p.N;
      bp.mt[g] = cdiv(M, BM);
      bp.nt[g] = cdiv(N, BN);
      const long tmn = (long)bp.mt[g] * bp.nt[g];
      SplitPlan sp{1, (q.p.K + BK - 1) / BK * BK};
      if (g < cnt && q.ws && q.wsf) {
        long want = (target + tmn - 1) / tmn;
        const long maxs = q.p.K / 64;
        if (want > maxs) want = maxs;
        const long per = (long)((M + 3) & ~3) * N;
        if (per * want > (long)q.wsf) want = (long)q.wsf / per;
        if (want > 1) {
          int kc = (int)((q.p.K + want - 1) / want);
          kc = (kc + BK - 1) / BK * BK;
          const int splits = (q.p.K + kc - 1) / kc;
          if (splits > 1) sp = SplitPlan{splits, kc};
        }
      }
      bp.p[g].sk.partial = sp.splits > 1 ? q.ws : nullptr;
      bp.p[g].sk.splits = sp.splits;
      bp.p[g].sk.kchunk = sp.kchunk;
      bp.first[g] = tiles;
      if (g < cnt) {
        tiles += (int)tmn * sp.splits;
        if (sp.splits > 1) {
          const long Mp = (M + 3) & ~3;
          SumJob J{q.ws, nullptr, (long)N * Mp, (int)(N * Mp), sp.splits, (int)Mp, M, 0, 0, 0, 0};
          describe_output(bp.p[g], J);
          jobs.sums.push_back(J);
        }
      }
    }
    bp.first[kMaxBatch] = tiles;
    for (int g = cnt; g < kMaxBatch; ++g) bp.first[g] = tiles;
    bp.n = cnt;
    int rc = kernel_budget_ok(reinterpret_cast<const void*>(&igemm_batch_kernel<P, BM, BN, BK, 0, false>), 256, 0, "igemm batch (conv2d wgrad)");
    if (rc) return rc;
    hipLaunchKernelGGL((igemm_batch_kernel<P, BM, BN, BK, 0, false>), dim3(tiles), dim3(256), 0, st, bp);
    if ((rc = check_launch("igemm_batch_conv"))) return rc;
  }
  jobs.conv2d.clear();
  return LSHM_OK;
}

size_t igemm_workspace_floats(int M, int N, int K, int zgroups) {
  // enough for the largest split the planner may choose (768 tiles' worth of slabs)
  const long Mp = (M + 3) & ~3;
  const long tiles = (long)cdiv(M, 128) * cdiv(N, 64) * zgroups;
  if (tiles >= 200 || K <= 64) return 0;
  long want = 1536;  // upper bound on splits (tiles >= 1)
  const long maxs = K / 64;
  if (want > maxs) want = maxs;
  return (size_t)(Mp * (long)N * zgroups * want);
}

// The operand fetches address a tensor through a buffer resource with 32-bit byte offsets: every element an
// operand tensor holds must lie below 4 GiB from its base pointer (1 Gi floats; the arena of BASELINE.json's
// largest configuration holds 0.2 Gi in its biggest tensor).  A larger tensor is refused, never wrapped.
static bool fits32(long elems_a, long elems_b) {
  const long lim = 1L << 30;
  if (elems_a < lim && elems_b < lim) return true;
  set_last_error("implicit GEMM: an operand tensor spans 4 GiB or more; split the batch");
  return false;
}
static long span(long bs, int B, long inner) { return (B > 0 ? (long)(B - 1) * bs : 0) + inner; }

int conv2d_fwd(const Conv2dFwdParams& p, float* ws, size_t wsf, hipStream_t st, const Conv2dFwdParams* p1) {
  if (!fits32(span(p.x_bs, p.B, (long)p.Cin * p.H * p.W), (long)p.Cout * p.Cin * 16)) return LSHM_ERR_UNSUPPORTED;
  return launch_auto<Conv2dFwd>(p, p1, p.M, p.N, 1, ws, wsf, st);
}
int conv2d_dgrad(const Conv2dDgradParams& p, float* ws, size_t wsf, hipStream_t st, const Conv2dDgradParams* p1) {
  if (!fits32(span(p.s_bs, p.B, (long)p.Cs * p.Hs * p.Ws), (long)p.Cs * p.Cb * 16)) return LSHM_ERR_UNSUPPORTED;
  return launch_auto<Conv2dDgrad>(p, p1, p.M, p.N, 4, ws, wsf, st);
}
int conv2d_wgrad(const Conv2dWgradParams& p, float* ws, size_t wsf, hipStream_t st, const Conv2dWgradParams* p1,
                 GradJobs* defer) {
  if (!fits32(span(p.s_bs, p.B, (long)p.Cs * p.Hs * p.Ws), span(p.big_bs, p.B, (long)p.Cb * p.Hs * p.Ws * 4)))
    return LSHM_ERR_UNSUPPORTED;
  if (defer && defer->batch_conv && !p1 && !p.accumulate && !t_matrix_bf16 && g_tune_force < 0) {
    defer->conv2d.push_back(GradJobs::ParkedConv2d{p, ws, wsf});  // launched with the others: grad_jobs_launch_conv
    return LSHM_OK;
  }
  return launch_auto<Conv2dWgrad>(p, p1, p.M, p.N, 1, ws, wsf, st, defer);
}
int conv1d_fwd(const Conv1dFwdParams& p, float* ws, size_t wsf, hipStream_t st, const Conv1dFwdParams* p1) {
  if ((g_tune_force < 0 || p.x_bf16 || p.y_bf16) && conv1d_stream_supported(p) && (!p1 || conv1d_stream_supported(*p1))) return conv1d_stream(p, p1, st);
  if (p.x_bf16 || p.y_bf16) { set_last_error("conv1d: bf16 tensors need the streaming kernel"); return LSHM_ERR_UNSUPPORTED; }
  if (!fits32(span(p.x_bs, p.B, (long)p.Cin * p.L), (long)p.Cout * p.Cin * 4)) return LSHM_ERR_UNSUPPORTED;
  return launch_auto<Conv1dFwd>(p, p1, p.M, p.N, 1, ws, wsf, st);
}
int conv1d_dgrad(const Conv1dDgradParams& p, float* ws, size_t wsf, hipStream_t st, const Conv1dDgradParams* p1) {
  if ((g_tune_force < 0 || p.big_bf16 || p.s_bf16) && tconv1d_stream_supported(p) && (!p1 || tconv1d_stream_supported(*p1))) return tconv1d_stream(p, p1, st);
  if (p.big_bf16 || p.s_bf16) { set_last_error("tconv1d: bf16 tensors need the streaming kernel"); return LSHM_ERR_UNSUPPORTED; }
  if (!fits32(span(p.s_bs, p.B, (long)p.Cs * p.Ls), (long)p.Cs * p.Cb * 4)) return LSHM_ERR_UNSUPPORTED;
  return launch_auto<Conv1dDgrad>(p, p1, p.M, p.N, 1, ws, wsf, st);
}
int conv1d_wgrad(const Conv1dWgradParams& p, float* ws, size_t wsf, hipStream_t st, const Conv1dWgradParams* p1,
                 GradJobs* defer) {
  if (!fits32(span(p.s_bs, p.B, (long)p.Cs * p.Ls), span(p.big_bs, p.B, (long)p.Cb * p.Lb))) return LSHM_ERR_UNSUPPORTED;
  return launch_auto<Conv1dWgrad>(p, p1, p.M, p.N, 1, ws, wsf, st, defer);
}
int strided_gemm(const StridedGemmParams& p, bool a_m_fast, bool b_n_fast, float* ws, size_t wsf,
                 hipStream_t st, const StridedGemmParams* p1, GradJobs* defer) {
  {
    auto ext = [](long s0, int n0, long s1, int n1) { return labs(s0) * (n0 - 1) + labs(s1) * (n1 - 1) + 1; };
    if (!fits32(ext(p.sam, p.M, p.sak, p.K), ext(p.sbk, p.K, p.sbn, p.N))) return LSHM_ERR_UNSUPPORTED;
  }
  if (a_m_fast && b_n_fast) return launch_auto<Strided<true, true>>(p, p1, p.M, p.N, 1, ws, wsf, st, defer);
  if (a_m_fast) return launch_auto<Strided<true, false>>(p, p1, p.M, p.N, 1, ws, wsf, st, defer);
  if (b_n_fast) return launch_auto<Strided<false, true>>(p, p1, p.M, p.N, 1, ws, wsf, st, defer);
  return launch_auto<Strided<false, false>>(p, p1, p.M, p.N, 1, ws, wsf, st, defer);
}

}  // namespace lshm
