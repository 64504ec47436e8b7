// Internal launcher interface shared by the C-ABI layer (capi.hip) and the step
// engine (engine.hip).  All pointers are device pointers; every launcher only
// enqueues work on `st` (no allocation, no synchronisation: graph-capturable).
#pragma once
#include <vector>
#include "common.h"
#include "../../include/lshm.h"

namespace lshm {

// ---- collectives (comm.hip; RCCL bound at run time)
}  // namespace lshm
struct lshm_comm;
namespace lshm {
int comm_allreduce_segments(lshm_comm* c, float* const* bufs, const size_t* ns, int nseg, double* tail, size_t ntail,
                            hipStream_t st);
int comm_world(const lshm_comm* c);

// ---- implicit-GEMM problem descriptors (igemm.hip) -------------------------
struct SplitK { float* partial; int splits; int kchunk; };  // filled in by the launcher
struct Conv2dFwdParams {   // y = act(conv2d_k4s2p1(x, w) + bias) [* elu'(dact)]
  const float* x; const float* w; const float* bias; float* y; const float* dact;
  int B, Cin, H, W, Cout, Ho, Wo;
  long x_bs, y_bs;  // batch strides (elements)
  int act;
  int M, N, K;
  SplitK sk;
};
struct Conv2dDgradParams {  // big = act(tconv2d_k4s2p1(small, w) + bias) [* elu'(dact)]
  const float* s; const float* w; const float* bias; float* big; const float* dact;
  int B, Cs, Hs, Ws, Cb;
  long s_bs, big_bs;
  int act;
  int M, N, K;
  SplitK sk;
};
struct Conv2dWgradParams {  // partial[z] = small^T (x) im2col(big) over a K slice
  const float* s; const float* big; float* dw;
  int B, Cs, Hs, Ws, Cb;
  long s_bs, big_bs;
  int M, N, K, accumulate;
  SplitK sk;
};
struct Conv1dFwdParams {
  const float* x; const float* w; const float* bias; float* y; const float* dact;
  int B, Cin, L, Cout, Lo, pad;
  long x_bs, y_bs;
  int act;
  int M, N, K;
  SplitK sk;
  int x_bf16 = 0;  // x is a bf16 tensor (streaming kernels of the two outer layers only)
  int y_bf16 = 0;  // y (and dact, which has its shape) is a bf16 tensor (outermost layer only)
};
struct Conv1dDgradParams {
  const float* s; const float* w; const float* bias; float* big; const float* dact;
  int B, Cs, Ls, Cb, Lb, pad;
  long s_bs, big_bs;
  int act;
  int M, N, K;
  SplitK sk;
  int big_bf16 = 0;  // big (and dact, which has its shape) is a bf16 tensor (streaming kernels of the two outer layers only)
  int s_bf16 = 0;    // s is a bf16 tensor (outermost layer only)
};
struct Conv1dWgradParams {
  const float* s; const float* big; float* dw;
  int B, Cs, Ls, Cb, Lb, pad;
  long s_bs, big_bs;
  int M, N, K, accumulate;
  SplitK sk;
};
struct StridedGemmParams {
  const float* a; const float* b; const float* bias; float* c; const float* dact;
  long sam, sak, sbk, sbn, scm, scn, sdm, sdn;
  int act;
  int M, N, K;
  SplitK sk;
  // optional addend applied before the ELU' multiply: c = (acc + add[m,n]) * elu'(dact), n < add_n
  const float* add; long sxm; int add_n;
};

// ---- deferred reductions (deferred.hip) ---------------------------------------------------------
// The backward pass of an autoencoder ends in ~60 tiny sums (split-K slabs of the weight gradients,
// per-workgroup partials of the direct kernels, bias gradients).  Instead of one launch each, a
// backward can park them in a GradJobs list; grad_jobs_finish() runs them all in two launches, in a
// fixed order (bitwise reproducible).  Everything a job reads must stay untouched until then:
// scratch comes from the list's own bump allocator, never from shared split-K scratch.
struct SumJob {     // dst[map(j)] (=|+=) sum_{s<S} src[s*stride + j],  j < n
  const float* src; float* dst;
  long stride;      // floats between consecutive partials
  int n, S;
  int Mp, M;        // Mp == 0: map(j) = j;  else j = col*Mp + row (split-K slab), rows >= M dropped,
  int rs, cs;       //          map(j) = row*rs + col*cs
  int accumulate;
  int lpo;          // lanes cooperating on one output (1, 16 or 64); chosen by grad_jobs_finish
};
struct ChanJob {    // partial[chunk*C + c] = sum over a chunk of {dz[b, c, :]}  (bias gradient, stage 1)
  const float* dz; float* partial;
  long bs, HW;      // batch stride and spatial size (HW % 4 == 0)
  int B, C, chunks;
};
struct GradJobs {
  std::vector<ChanJob> chan;
  std::vector<SumJob> sums;
  std::vector<StridedGemmParams> dense;  // dense-layer weight gradients waiting for their one shared launch
  bool batch_dense = false;              // linear_wgrad parks its GEMM here instead of launching it
  struct ParkedConv2d { Conv2dWgradParams p; float* ws; size_t wsf; };
  std::vector<ParkedConv2d> conv2d;      // implicit-GEMM weight gradients of 2-D layers waiting for their one shared launch
  bool batch_conv = false;               // conv2d_wgrad parks its problem here instead of launching it (grad_jobs_launch_conv)
  float* scratch = nullptr;
  size_t cap = 0, used = 0;
  float* take(size_t n) {
    n = (n + 63) & ~(size_t)63;  // 256-byte boundaries: slabs are written in full cache lines
    if (used + n > cap) return nullptr;
    float* p = scratch + used;
    used += n;
    return p;
  }
  // bias gradient of a conv layer's dz (B, C, HW): stage-1 job + final sum; false if it cannot be deferred
  bool add_channel_sum(const float* dz, long bs, int B, int C, long HW, float* db, int accumulate);
};
int grad_jobs_finish(GradJobs& jobs, hipStream_t st);
int grad_jobs_launch_dense(GradJobs& jobs, hipStream_t st);  // the parked dense weight gradients, one launch
int grad_jobs_launch_conv(GradJobs& jobs, hipStream_t st);   // the parked 2-D conv weight gradients, one launch (split-K slabs join the job list)

// ws / wsf: optional split-K scratch (null: never split)
// p1 (optional): a second problem of identical shape run in the same launch (ws is split in two)
int conv2d_fwd(const Conv2dFwdParams& p, float* ws, size_t wsf, hipStream_t st, const Conv2dFwdParams* p1 = nullptr);
int conv2d_dgrad(const Conv2dDgradParams& p, float* ws, size_t wsf, hipStream_t st, const Conv2dDgradParams* p1 = nullptr);
int conv2d_wgrad(const Conv2dWgradParams& p, float* ws, size_t wsf, hipStream_t st, const Conv2dWgradParams* p1 = nullptr,
                 GradJobs* defer = nullptr);
int conv1d_fwd(const Conv1dFwdParams& p, float* ws, size_t wsf, hipStream_t st, const Conv1dFwdParams* p1 = nullptr);
int conv1d_dgrad(const Conv1dDgradParams& p, float* ws, size_t wsf, hipStream_t st, const Conv1dDgradParams* p1 = nullptr);
int conv1d_wgrad(const Conv1dWgradParams& p, float* ws, size_t wsf, hipStream_t st, const Conv1dWgradParams* p1 = nullptr,
                 GradJobs* defer = nullptr);
// defer (weight-gradient problems only): a split launch leaves its slabs in ws and queues the combine
int strided_gemm(const StridedGemmParams& p, bool a_m_fast, bool b_n_fast, float* ws, size_t wsf,
                 hipStream_t st, const StridedGemmParams* p1 = nullptr, GradJobs* defer = nullptr);
// n independent m-fast x n-fast problems without bias / activation / split-K (dense weight gradients) in one launch
int strided_gemm_batch_mm(const StridedGemmParams* probs, int n, hipStream_t st);
size_t igemm_workspace_floats(int M, int N, int K, int zgroups);
// streaming kernels for the outer 1-D layers (conv1d_stream.hip); conv1d_fwd / conv1d_dgrad use them when they apply
bool conv1d_stream_supported(const Conv1dFwdParams& p);
int conv1d_stream(const Conv1dFwdParams& p, const Conv1dFwdParams* p1, hipStream_t st);
bool conv1d_wgrad_stream_supported(int Cs, int Cb, int Ls, int Lb, int pad, int bias_from, long s_bs, long big_bs,
                                   const float* small, const float* big);
// fd (optional; the 8 -> 4 channel transposed layer only): the layer's data gradient in the same pass over `big`
// and `small`: dx = ELU'(small) * conv(big, w), written to dx / dx2 (batch stride dx_bs)
// dact: multiply by ELU'(saved input of the layer)
struct FusedDgrad { const float* w; const float* w2; float* dx; float* dx2; long dx_bs; int dact = 1; };
bool conv1d_bwd_fused_supported(int Cs, int Cb, int pad);
// the general one-pass backward (conv1d_fused.hip): 12 / 8 and 8 / 4 channels, conv (pad 1) and transposed (pad 0)
bool conv1d_bwd_fused2_supported(int Cs, int Cb, int pad);
int conv1d_bwd_fused2(const float* small, const float* small2, long s_bs, const float* big, const float* big2, long big_bs,
                      float* ws, float* ws2, int B, int Cs, int Cb, int Ls, int Lb, int pad, int max_blocks,
                      hipStream_t st, int* grid_out, int big_bf16, const FusedDgrad& fd, int small_bf16 = 0);
int conv1d_wgrad_stream(const float* small, const float* small2, long s_bs, const float* big, const float* big2,
                        long big_bs, float* ws, float* ws2, int B, int Cs, int Cb, int Ls, int Lb, int pad,
                        int bias_from, int max_blocks, hipStream_t st, int* grid_out, int big_bf16 = 0,
                        const FusedDgrad* fd = nullptr, int small_bf16 = 0);
bool tconv1d_stream_supported(const Conv1dDgradParams& p);
int tconv1d_stream(const Conv1dDgradParams& p, const Conv1dDgradParams* p1, hipStream_t st);
// mode: 0 static heuristic, 1 time the tile configurations on first use of a shape; force >= 0 pins one configuration
void igemm_set_tuning(int mode, int force);
// Operand precision of the GEMM-shaped launches made by the calling thread while the scope lives
// (0: fp32 operands, 1: operands rounded to bf16 at LDS staging).  Engines carry theirs in
// lshm_step_config.precision, the per-op C ABI in the `_bf16` suffix: no process-wide switch.
struct MatrixPrecisionScope {
  int prev;
  explicit MatrixPrecisionScope(int bf16);
  ~MatrixPrecisionScope();
  MatrixPrecisionScope(const MatrixPrecisionScope&) = delete;
  MatrixPrecisionScope& operator=(const MatrixPrecisionScope&) = delete;
};
int igemm_matrix_precision();  // of the calling thread's innermost scope
// Schedule word (LSHM_SCHED_*, include/lshm.h) of the calling thread's innermost scope: an engine call runs under its
// engine's lshm_step_config.schedule, the per-op C ABI under 0 (the shipped choices) unless its `_ex` form is given a word.
// No environment variable chooses a kernel.
struct ScheduleScope {
  unsigned prev;
  explicit ScheduleScope(unsigned word);
  ~ScheduleScope();
  ScheduleScope(const ScheduleScope&) = delete;
  ScheduleScope& operator=(const ScheduleScope&) = delete;
};
unsigned schedule_word();
inline bool sched(unsigned bit) { return (schedule_word() & bit) != 0; }
size_t igemm_tuning_export(char* buf, size_t cap);  // returns the bytes needed (with the terminating 0)
int igemm_tuning_import(const char* text);          // returns the entries read

// ---- direct LDS-patch kernels for the outer 2-D layers (conv_direct.hip) ----
bool tconv2d_direct_supported(int Cs, int Cb, int Hs, int Ws);
int tconv2d_direct(const float* small, long s_bs, const float* w, const float* bias, float* big,
                   long big_bs, const float* dact, int B, int Cs, int Cb, int Hs, int Ws, int act,
                   hipStream_t st, int big_bf16 = 0, int small_bf16 = 0);

bool conv2d_direct_supported(int Cin, int Cout, int Ho, int Wo);
int conv2d_direct(const float* x, long x_bs, const float* w, const float* bias, float* y, long y_bs,
                  const float* dact, int B, int Cin, int Cout, int Ho, int Wo, int act, hipStream_t st, int x_bf16 = 0,
                  int y_bf16 = 0);
bool conv2d_wgrad_direct_supported(int Cs, int Cb, int Hs, int Ws);
size_t conv2d_wgrad_direct_workspace_floats(int Cs, int Cb);
// db (optional): bias gradient fused; bias_from 1: dz is `small` (conv), 2: dz is `big` (transposed conv)
int conv2d_wgrad_direct(const float* small, long s_bs, const float* big, long big_bs, float* dw, float* db,
                        int bias_from, int B, int Cs, int Cb, int Hs, int Ws, float* ws, size_t wsf, int accumulate,
                        hipStream_t st, GradJobs* defer = nullptr, int big_bf16 = 0, int small_bf16 = 0);

// one-pass backward (data + weight + bias gradient) of the outermost 2-D decoder layer (8 -> 4 channels)
bool tconv2d_bwd_fused_supported(int Cs, int Cb, int Hs, int Ws);
int tconv2d_bwd_fused(const float* small, long s_bs, const float* big, long big_bs, const float* w, float* dsmall, int dact,
                      float* dw, float* db, int B, int Hs, int Ws, float* ws, size_t wsf, int accumulate, hipStream_t st,
                      GradJobs* defer = nullptr, int big_bf16 = 0, int small_bf16 = 0);

// one-pass backward of the 12 <-> 8 channel 2-D layers (tconv4, conv1), conv2d_fused.hip; conv != 0: conv layer
bool conv2d_bwd_lds_supported(int Cs, int Cb, int Hs, int Ws);
int conv2d_bwd_lds(const float* small, long s_bs, const float* big, long big_bs, const float* w, float* dout, int conv, int dact,
                   float* dw, float* db, int B, int Cs, int Cb, int Hs, int Ws, float* ws, size_t wsf, int accumulate,
                   hipStream_t st, GradJobs* defer = nullptr, int big_bf16 = 0);

// conv0 of netT and netF from x and the 2-D reconstruction in one launch, no materialised residual (resid_conv0.hip)
bool resid_conv0_supported(int C, int P, int Cin, int Cout, int L1d);
int resid_conv0(const float* x, const float* x1, const float* wT, const float* bT, float* yT, const float* wF, const float* bF,
                float* yF, long y_bs, int B, hipStream_t st, int bf = 0,   // bf: x1, the outputs and the kept images are bf16 tensors
                float* out_row = nullptr, float* out_col = nullptr);       // both given: the residual images are written too

// backward of conv0 of netT / netF + the combination into the 2-D autoencoder's output gradient, one pass (conv0_bwd_tile.hip)
bool conv0_bwd_tile_supported(int C, int P, int Cin, int Cout, int L1d, long in_bs);
size_t conv0_bwd_tile_workspace_floats();
int conv0_bwd_tile(const float* r, const float* dzT, const float* dzF, long z_bs, const float* wT, const float* wF, const float* gx1p,
                   float* gx1, float* dwT, float* dbT, float* dwF, float* dbF, int B, float* ws, size_t wsf, int accumulate,
                   hipStream_t st, GradJobs* defer, int bf = 0);  // bf: the residual, both dz tensors, gx1p and gx1 are bf16 tensors

// One element of the reconstruction pass (src/kharmonic_lofar.py:150-158,200-202), shared by recon_kernel (elementwise.hip) and
// recon_bwd5_kernel so that both give the same bits whatever the compiler would have contracted in either context: every
// fused multiply-add is written out, nothing else is fused.
//   r1 = x - x1, h = r1 / 2, r2 = h - x2, r3 = h - x3, e = x1 + x2 + x3 - x;  UPD: m_k += rho r_k first
//   s += [e^2, m1 r1, r1^2, m2 r2, r2^2, m3 r3, r3^2];  g2 = (2e - m2 - rho r2) / n, g3 likewise,
//   g1p = (2e - m1 - rho r1) / n - (m2 + rho r2 + m3 + rho r3) / (2n)
struct ReconElem { float m1, m2, m3, g1p, g2, g3; };
template <bool UPD, bool GRAD>
__device__ __forceinline__ ReconElem recon_elem(float xv, float a1, float a2, float a3, float m1, float m2, float m3, float rho, float inv_n,
                                                float (&s)[7]) {
#pragma clang fp contract(off)
  const float r1 = xv - a1, h = 0.5f * r1, r2 = h - a2, r3 = h - a3;
  const float e = a1 + a2 + a3 - xv;
  if (UPD) { m1 = fmaf(rho, r1, m1); m2 = fmaf(rho, r2, m2); m3 = fmaf(rho, r3, m3); }
  s[0] = fmaf(e, e, s[0]);
  s[1] = fmaf(m1, r1, s[1]); s[2] = fmaf(r1, r1, s[2]);
  s[3] = fmaf(m2, r2, s[3]); s[4] = fmaf(r2, r2, s[4]);
  s[5] = fmaf(m3, r3, s[5]); s[6] = fmaf(r3, r3, s[6]);
  ReconElem o{m1, m2, m3, 0.f, 0.f, 0.f};
  if (GRAD) {
    const float t2 = fmaf(rho, r2, m2), t3 = fmaf(rho, r3, m3), e2 = 2.f * e;
    o.g2 = (e2 - t2) * inv_n;
    o.g3 = (e2 - t3) * inv_n;
    o.g1p = fmaf(-rho, r1, e2 - m1) * inv_n - (0.5f * (t2 + t3)) * inv_n;
  }
  return o;
}

// reconstruction pass + backward of the last layer of netT / netF (recon_bwd5.hip)
bool recon_bwd5_supported(int C, int P, int Cin, int Cout, int Ls);
size_t recon_bwd5_workspace_floats();
int recon_bwd5_grid(int B);
int recon_bwd5(const float* x, const float* x1, const float* aT, const float* aF, long a_bs, const float* wT, const float* bT,
               const float* wF, const float* bF, float* y1, float* y2, float* y3, float rho, int B, float* gx1p, float* dT, float* dF,
               long d_bs, float* block_partials, float* slabs, size_t slab_floats, hipStream_t st, float grad_scale, int bf = 0);
// (bf: x1, the layers' inputs, gx1p, the gradient images and the data gradients are bf16 tensors)
int tconv5_pair_bwd(const float* gx2, const float* gx3c, const float* aT, const float* aF, long a_bs, const float* wT, const float* wF,
                    int B, float* dT, float* dF, long d_bs, float* slabs, size_t slab_floats, hipStream_t st, int bf = 0);
int recon_bwd5_close(const float* slabs, int grid, float* dwT, float* dbT, float* dwF, float* dbF, int accumulate, hipStream_t st,
                     GradJobs* defer);

// LDS-staged weight gradient of the mid 1-D layers (24/12 and 48/24 channels), see conv_direct.hip
bool conv1d_wgrad_mid_supported(int Cs, int Cb, int Ls, int Lb, int pad, int bias_from, long s_bs, long big_bs,
                                const float* small, const float* big);
size_t conv1d_wgrad_mid_workspace_floats(int Cs, int Cb);
int conv1d_wgrad_mid(const float* small, long s_bs, const float* big, long big_bs, float* dw, float* db, int bias_from,
                     int B, int Cs, int Cb, int Ls, int Lb, int pad, float* ws, size_t wsf, int accumulate,
                     hipStream_t st, const float* small2 = nullptr, const float* big2 = nullptr, float* dw2 = nullptr,
                     float* db2 = nullptr, GradJobs* defer = nullptr);
bool conv1d_wgrad_direct_supported(int Cs, int Cb, int Ls);
size_t conv1d_wgrad_direct_workspace_floats(int Cs, int Cb);
int conv1d_wgrad_direct(const float* small, long s_bs, const float* big, long big_bs, float* dw, float* db,
                        int bias_from, int nbias, int B, int Cs, int Cb, int Ls, int Lb, int pad, float* ws,
                        size_t wsf, int accumulate, hipStream_t st, const float* small2 = nullptr,
                        const float* big2 = nullptr, float* dw2 = nullptr, float* db2 = nullptr,
                        GradJobs* defer = nullptr, int big_bf16 = 0, const FusedDgrad* fd = nullptr, int small_bf16 = 0);

// ---- LDS-resident chains of three k4 s4 1-D layers (chain1d.hip) -------------------------------
// one stage: weights, optional bias, global output (B, Cout, Lout) and either ELU (act) or the ELU' multiply by the
// saved activation of the output's shape (dact); [2]: the two problems of a paired launch
struct Chain1dStage {
  const float* w[2];
  const float* bias[2];
  float* out[2];
  const float* dact[2];
  long out_bs;
  int act;
};
// up == false: stride-4 conv direction, channels ch[0..3] from L0 positions (forward of conv2..4: pad 1; data gradient
// of tconv3..1: pad 0); up == true: transposed direction (forward of tconv1..3: pad 0; data gradient of conv4..2: pad 1)
bool conv1d_chain_supported(bool up, const int* ch, int L0);
int conv1d_chain(bool up, const Chain1dStage* st, const float* in0, const float* in1, long in_bs, int pad, int B, hipStream_t s);

// ---- the dense middle of AutoEncoder1DCNN (latent_dim 16, rica) as one launch per direction (dense1d.hip) ----------
struct Dense1dFwdIO {
  const float* cat1;                                   // (B, 784) [conv5 output | elu(fcuv1)]
  const float *fc1w, *fc1b, *fc2inw, *fc2inb, *fc2outw, *fc2outb, *fc3w, *fc3b;
  float *z1, *mu, *cat3, *d0;                          // (B,16), (B,16) inside Mu (ld ldmu), (B,32), (B,768)
};
struct Dense1dBwdIO {
  const float *dd0, *cat3, *mu, *gmu, *z1, *cat1;      // gradient of fc3's output; saved activations; latent-term gradient
  const float *fc1w, *fc2inw, *fc2outw, *fc3w;
  float *dcat3, *dzmu, *dz1, *dcat1;                   // (B,32), (B,16), (B,16), (B,784)
};
bool dense1d_supported(int L, int hd, int rica);  // the engine's choice
bool dense1d_built(int L);                          // latent widths the kernels exist for
int dense1d_fwd(const Dense1dFwdIO& p, const Dense1dFwdIO* p1, long ldmu, int B, hipStream_t st, int L = 16);  // L: 16 | 256
int dense1d_bwd(const Dense1dBwdIO& p, const Dense1dBwdIO* p1, long ldmu, long ldgmu, int B, hipStream_t st, int L = 16);

// ---- layer-level helpers (layers.hip): conv / tconv / linear, fwd + bwd -----
// kind: 0 conv2d k4s2p1, 1 tconv2d k4s2p1, 2 conv1d k4s4p1, 3 tconv1d k4s4p0
struct ConvLayer {
  int kind;
  int B, Cin, Cout;
  int Hin, Win;       // 1D: Hin = 1, Win = L
  long in_bs, out_bs; // batch strides of input / output tensors (elements)
  // bf16 storage (engine precision LSHM_PRECISION_BF16_STORAGE): the layer's input / output tensor -- and with it the
  // gradient w.r.t. that tensor -- is bf16; only the outermost layers (4 <-> 8 channels) have kernels for it
  int in_bf16 = 0, out_bf16 = 0;
};
void conv_out_dims(const ConvLayer& L, int& Hout, int& Wout);
// scratch (floats) that lets every problem of the layer use split-K + the bias partial sums
// (twice that when two problems share a launch)
size_t conv_workspace_floats(const ConvLayer& L);
// pointer bundles; every layer function takes one bundle and, optionally, a second one for an
// independent problem of identical shape that rides in the same launches (netT / netF)
struct ConvFwdIO { const float* x; const float* w; const float* b; float* y; };
struct ConvDgradIO { const float* dz; const float* w; float* dx; const float* dact_in; };
struct ConvWgradIO { const float* x; const float* dz; float* dw; float* db; };
// y = act(op(x,w)+b); ws optional (null: no split-K)
int conv_layer_fwd(const ConvLayer& L, const ConvFwdIO& io, int act, float* ws, size_t wsf, hipStream_t st,
                   const ConvFwdIO* io2 = nullptr);
// dx = op^T(dz, w) [* elu'(x_saved) if dact_in]
int conv_layer_dgrad(const ConvLayer& L, const ConvDgradIO& io, float* ws, size_t wsf, hipStream_t st,
                     const ConvDgradIO* io2 = nullptr);
// dw, db overwritten (accumulate=0) or accumulated (accumulate=1); ws required
// defer: scratch is taken from the job list instead of ws and the closing sums are queued on it
// fuse / fuse2 (only when conv_layer_bwd_fusable): the data gradient of the same layer is computed in the same pass
int conv_layer_wgrad(const ConvLayer& L, const ConvWgradIO& io, float* ws, size_t ws_floats, int accumulate,
                     hipStream_t st, const ConvWgradIO* io2 = nullptr, GradJobs* defer = nullptr,
                     const ConvDgradIO* fuse = nullptr, const ConvDgradIO* fuse2 = nullptr);
// one kernel can produce the weight, bias AND data gradient of this layer (outermost 1-D transposed layer)
bool conv_layer_bwd_fusable(const ConvLayer& L, const ConvWgradIO& io, const ConvDgradIO& dio);

size_t conv_wgrad_defer_floats(const ConvLayer& L, int G);
size_t linear_wgrad_defer_floats(int B, int K, int N, int G);

struct LinFwdIO { const float* x; const float* w; const float* b; float* y; };
struct LinDgradIO { const float* dz; const float* w; float* dx; const float* xsaved; const float* add; };
struct LinWgradIO { const float* x; const float* dz; float* dw; float* db; };
// y[B,N] (ld ldy) = act(x[B,K] (ld ldx) @ w[N,K]^T + b)
int linear_fwd(const LinFwdIO& io, long ldx, long ldy, int B, int K, int N, int act, float* ws, size_t wsf,
               hipStream_t st, const LinFwdIO* io2 = nullptr);
// dx[B,K] (ld lddx) = (dz[B,N] (ld lddz) @ w[N,K] + add[:, :add_n] (ld ldadd)) [* elu'(xsaved (ld ldxs))]
int linear_dgrad(const LinDgradIO& io, long lddz, long lddx, long ldxs, long ldadd, int add_n, int B, int K,
                 int N, float* ws, size_t wsf, hipStream_t st, const LinDgradIO* io2 = nullptr);
int copy2d(const float* src, long lds, float* dst, long ldd, int rows, int cols, hipStream_t st);
// dw[N,K] = dz^T x ; db[N] = colsum(dz)
int linear_wgrad(const LinWgradIO& io, long ldx, long lddz, int B, int K, int N, float* ws, size_t wsf,
                 hipStream_t st, const LinWgradIO* io2 = nullptr, GradJobs* defer = nullptr);

// ---- dictionary learning (rica.hip; src/rica_lofar.py) ------------------------------------------
size_t rica_workspace_floats(int B, int L, int M);
int rica_loss_grad(const float* Xt, const float* A, const float* St, int B, int L, int M, float lambda1,
                   double* loss, float* dSt, float* ws, size_t wsf, hipStream_t st);
int rica_update_dictionary(const float* Xt, float* A, const float* St, int B, int L, int M, float eta,
                           double* dA_norm_sq, float* ws, size_t wsf, hipStream_t st);

// ---- elementwise / reductions (elementwise.hip) -----------------------------
int uv_harmonics(const float* uv, const float* scales, int H, int B, float* out, hipStream_t st);
// uvh plus the dense layers fed by uvh alone (weights (4H, 4H) row-major, ELU), see uv_features_kernel
struct UvLayers {
  int n;
  const float* w[6];
  const float* bias[6];
  float* out[6];
  long ld[6];
};
int uv_features(const float* uv, const float* scales_host, int H, int B, float* uvh, const UvLayers& layers,
                hipStream_t st);
int uv_harmonics_host_scales(const float* uv, const float* scales_host, int H, int B, float* out,
                             hipStream_t st);
int elu_bwd(const float* gy, const float* y, float* dz, long n, hipStream_t st);
// out_row = (x-x1)/2 ; out_col = per-plane transpose of out_row (planes of P x P)
// bf != 0: x1 and the outputs are bf16 tensors behind the float pointers (bf16 storage, see common.h)
int residual_split(const float* x, const float* x1, float* out_row, float* out_col, int planes,
                   int P, hipStream_t st, int bf = 0);
int plane_transpose(const float* in, float* out, int planes, int P, hipStream_t st, int in_bf = 0);
int widen_bf16(const float* in_bf16, float* out, long n, hipStream_t st);
// reduce_partials: out[i] (=|+=) sum_{s<S} partial[s*n + i]
int reduce_partials(const float* partial, float* out, long n, int S, int accumulate,
                    hipStream_t st, const float* partial2 = nullptr, float* out2 = nullptr);
int reduce_partials_strided(const float* partial, long stride, float* out, long n, int S, int accumulate,
                            hipStream_t st, const float* partial2 = nullptr, float* out2 = nullptr);
int channel_sum_partials(const float* dz, long bs, int B, int C, long HW, float* partial, int S,
                         hipStream_t st, const float* dz2 = nullptr, float* partial2 = nullptr);
int channel_sum_direct(const float* dz, long bs, int B, int C, long HW, float* db, int accumulate,
                       hipStream_t st, const float* dz2 = nullptr, float* db2 = nullptr);
#define LOGCOSH3_BLOCKS 32
int logcosh3_fwd_bwd(const float* z, long ldz, int rows, const int* seg_cols, const float* seg_scale,
                     double* partial, int nblocks, float* dz, long lddz, hipStream_t st);
// the seven reductions + gradients of src/kharmonic_lofar.py:150-158 (see elementwise.hip)
int recon_losses_fwd_bwd(const float* x, const float* x1, const float* x2, const float* x3c,
                         const float* y1, const float* y2, const float* y3, float rho, int planes,
                         int P, double* sums7, float* gx1p, float* gx2, float* gx3c,
                         float* block_partials, hipStream_t st,
                         float grad_scale = 1.f, int bf = 0)  /* gradients (not the sums) are multiplied by grad_scale;
                                                                 bf: x1..x3c and the gradient images are bf16 */;
size_t recon_partials_floats(int planes, int P);
// second stage of the reconstruction pass on its own (multiplier_update_recon with sums7 == nullptr leaves it to the caller)
int recon_sum7(const float* block_partials, int planes, int P, double* sums7, hipStream_t st);
bool recon_from_a_supported(int C, int P, int Cin, int Cout, int Ls);
int multiplier_update_recon_from_a(const float* x, const float* x1, const float* aT, const float* aF, long a_bs, const float* wT,
                                   const float* bT, const float* wF, const float* bF, int C, float* y1, float* y2, float* y3,
                                   float rho, int planes, int P, float* gx1p, float* gx2, float* gx3c, float* block_partials,
                                   hipStream_t st, float grad_scale, int bf = 0);
int recon_losses_from_a(const float* x, const float* x1, const float* aT, const float* aF, long a_bs, const float* wT, const float* bT,
                        const float* wF, const float* bF, int C, const float* y1, const float* y2, const float* y3, float rho,
                        int planes, int P, double* sums7, float* gx1p, float* gx2, float* gx3c, float* block_partials,
                        hipStream_t st);
int multiplier_update_recon(const float* x, const float* x1, const float* x2, const float* x3c, float* y1, float* y2,
                            float* y3, float rho, int planes, int P, double* sums7, float* gx1p, float* gx2,
                            float* gx3c, float* block_partials, hipStream_t st, float grad_scale = 1.f, int bf = 0);
int combine_dx1(const float* gx1p, const float* gT, const float* gFc, float* gx1, int planes,
                int P, hipStream_t st, int bf = 0);
int multiplier_update(const float* x, const float* x1, const float* x2, const float* x3c,
                      float* y1, float* y2, float* y3, float rho, int planes, int P,
                      hipStream_t st, int bf = 0);
int adam_step_flat(float* p, const float* g, float* m, float* v, long n, float lr, float b1,
                   float b2, float eps, const int* step_dev, int step_host, float gscale,
                   hipStream_t st);
int axpy_flat(float* y, const float* x, float alpha, long n, hipStream_t st);
int dot_flat(const float* a, const float* b, long n, double* out, float* ws, hipStream_t st);
int scale_flat(float* x, float alpha, long n, hipStream_t st);
constexpr int kMaxDots = 16;  // pairs per multi_dot_flat launch; stored (y, s) pairs per lbfgs_direction call
size_t multi_dot_workspace_doubles(int count);
int multi_dot_flat(const float* const* a, const float* const* b, int count, long n, double* out, double* ws, hipStream_t st);
size_t lbfgs_direction_workspace_doubles(int m);
int lbfgs_direction(const float* const* y, const float* const* s, int m, const float* g, double h_diag, float* d, long n,
                    double* ws, hipStream_t st);
int asum_flat(const float* a, long n, double* out, float* ws, hipStream_t st);
int logcosh_mean_fwd_bwd(const float* z, long ldz, int rows, int cols, float scale, double* loss,
                         float* dz, long lddz, int accumulate, hipStream_t st);

// ---- k-harmonic means (khm.hip) ---------------------------------------------
size_t khm_workspace_floats(int N, int D, int K);
// loss_sum[0] = sum_i K/(e_i+eps) (NOT yet divided by N*K*D);
// dX = gscale * d(loss_mean)/dX, dM likewise, with loss_mean = loss_sum/(Ntot*K*D)
int khm_fwd_bwd(const float* X, long ldx, const float* M, int N, int D, int K, float p, float eps,
                double inv_count, float gscale, double* loss_sum, float* dX, long lddx, float* dM,
                int accumulate_dx, float* ws, size_t ws_floats, hipStream_t st, int accumulate_dm = 0);
int khm_offline_partials(const float* X, long ldx, const float* M, int N, int D, int K, float p,
                         float eps, float* num, float* den, float* ws, size_t ws_floats,
                         hipStream_t st);
int khm_mean_distances(const float* X, long ldx, const float* M, int N, int D, int K, float p,
                       float* dist, float* ws, size_t ws_floats, hipStream_t st);

// ---- small latent-space losses (latent.hip) ---------------------------------
int cluster_sim_fwd_bwd(const float* M, int K, int D, float eps, float gscale, double* loss,
                        float* dM, int accumulate, hipStream_t st);
int aug_loss_fwd_bwd(const float* Z, long ldz, int rows, int D, int bpb, int batch_size,
                     float gscale, double* loss, float* dZ, long lddz, int accumulate,
                     hipStream_t st);

int dist_epilogue(const float* dist, int K, int* argmin, float* prob, hipStream_t st);

// ---- LOFAR minibatch patch pipeline (patches.hip) -----------------------------
size_t patches_workspace_floats();
int patches_from_vis(const int8_t* vis, const float* scale, int nb, int ntime, int nfreq, int P, int NC, float clampv,
                     int normalize, float* y, double* mean_std, double* moments, float* ws, hipStream_t st);
int patches_normalize_moments(float* y, long n, const double* moments, hipStream_t st);

// ---- batched 2D FFT feature op (fft.hip) ------------------------------------
int fft2_ortho_shift_cat_clamp(const float* x, float* out, int B, int C, float clampv,
                               hipStream_t st);
size_t fft2_backward_workspace_floats(int B, int C);
int fft2_feature_backward(const float* g, const float* y, float* dx, int B, int C, float clampv, float* ws, size_t wsf,
                          hipStream_t st);

}  // namespace lshm
