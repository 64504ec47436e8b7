// Native step engine: one ADMM iteration of src/kharmonic_lofar.py:131-202
// (closure forward, backward, and the no-grad forward + multiplier update) as a
// fixed sequence of kernel launches on one HIP stream, over a flat parameter /
// gradient arena and a caller-provided activation workspace.  The engine owns no
// device memory and never synchronises, so a whole iteration can be captured in
// a HIP graph by the host.
//
// Data layout in HBM (all fp32):
//   params / grads : one arena; tensors in the reference's state_dict order for
//                    net, netT, netF (torch layouts) followed by mod.M; offsets
//                    are multiples of 4 floats.
//   workspace      : per autoencoder the saved activations
//                    (conv0..4 outputs, cat1 = [conv5 out | elu(fcuv1)], z1, cat3 =
//                    [elu(fc2out) | elu(fcuv3)], fc3 out, tconv0..4 outputs, output),
//                    shared: uv harmonics, Mu = [mu | muT | muF] (B, L+2Lt), the row- and
//                    column-vectorised residuals, gradient ping-pong buffers, split-K partials.
#include "../../include/lshm.h"
#include "kernels.h"
#include "deep2d.h"
#include "chain1d_full.h"

#include <stdlib.h>
#include <string.h>
#include <functional>
#include <string>
#include <vector>

namespace lshm {

static const int CHL[7] = {4, 8, 12, 24, 48, 96, 192};  // src/lofar_models.py:31-41 (index 0 = input channels)

struct ParamInfo {
  std::string name;
  long offset, numel;
  int ndim;
  long shape[4];
};

struct AEPlan {
  int ndim;  // 2 or 1
  int L;     // latent dim
  int mu_col;  // column offset of this AE's latent inside Mu
  ConvLayer enc[6], dec[6];
  // parameter offsets
  long cw[6], cb[6], tw[6], tb[6];
  long fcuv1w, fcuv1b, fcuv3w, fcuv3b, fc1w, fc1b, fc2inw, fc2inb, fc2outw, fc2outb, fc3w, fc3b;
  // workspace offsets (floats)
  size_t act[5];   // conv0..conv4 outputs
  size_t cat1, z1, cat3, d0;
  size_t dact[5];  // tconv0..tconv4 outputs
  size_t out;      // (B, C, P*P)
};

}  // namespace lshm

using namespace lshm;

namespace lshm {
using FwdStep = std::function<int(float* ws, hipStream_t st)>;
}

struct lshm_engine {
  lshm_step_config cfg;
  struct FwdPlan {  // step list of the forward for one (params, x, uv) triple, see forward_plan
    const float* prm = nullptr;
    const float* x = nullptr;
    const float* uv = nullptr;
    std::vector<lshm::FwdStep> steps;
    size_t latent_mark = 0, output1d_mark = 0;
    size_t resid_mark = 0;        // steps[resid_mark] = residual_split, steps[resid_mark + 1] = conv0 of netT / netF
    lshm::FwdStep resid_conv0;    // both in one launch, neither vectorisation written (resid_conv0.hip); empty: not available
    lshm::FwdStep resid_conv0_keep;  // ... with both vectorisations written: what a forward with kept activations runs
  } plan;
  int D;        // L + 2 Lt
  int hdim;     // 4 H
  std::vector<ParamInfo> params;
  long nparams;
  long Moff;
  AEPlan ae[3];
  // shared workspace offsets
  size_t o_scales, o_uvh, o_Mu, o_gMu, o_row, o_col, o_gx1p, o_gx2, o_gx3c, o_gT, o_gFc, o_gx1,
      o_scal, o_dMscratch;
  // Everything a forward pass writes (activations, harmonics, latents, residuals, its split-K scratch) sits in
  // the first fwd_floats of the layout, and a second copy of that prefix follows the layout at alt_base: two
  // forwards that do not depend on each other -- the no-grad forward that closes iteration k and the closure
  // forward that opens iteration k+1 -- run side by side on two streams, each with `ws` or `ws + alt_base` as base.
  size_t o_fpart, fwd_floats, alt_base;
  size_t o_pack2d = 0;  // fragment-ordered copy of the 2-D autoencoder's deep weights (deep2d.hip), inside the forward prefix
  bool deep2d = false;  // conv3 .. tconv3 of the 2-D autoencoder's forward as one launch
  bool deep2d_bwd = false;  // ... and the data gradients of tconv2 .. conv2 as one launch
  bool full1d = false;  // conv2 .. tconv3 of the 1-D autoencoders' forward as one launch (chain1d_full.hip)
  size_t o_pack2d_bwd = 0;  // the backward's fragment-ordered weight copy
  unsigned wgrad_on_main = 0;  // which of the deep layers' weight gradients follow the data-gradient chain on ITS stream (ae_backward)
  bool pack_bwd_done = false;  // this closure's backward weight copy was already made (on the latent-space stream)
  int deep_bf16 = 0;    // the deep chains stream bf16 copies of the weights (precision != LSHM_PRECISION_F32)
  int deep_variant = 0; // 0: one patch per workgroup (a forward alone), 1: two (two forwards side by side: each fills half of the CUs)
  size_t o_recon_w5;    // weight-gradient slabs of netT / netF's last layer, left by the reconstruction pass (recon_bwd5.hip)
  bool recon_bwd5_done = false;  // ... and whether the pass that made the workspace's reconstruction terms was that one
  size_t o_recon_part;  // per-block partial sums of the reconstruction pass (its own buffer: their seven sums may be made later, beside the backward)
  hipStream_t fstream = nullptr;  // the no-grad forward + shared reconstruction pass, beside the next closure forward
  // Two "lanes" of backward scratch: netT and netF (independent given AE1's output, identical
  // shapes) run as pairs inside the same launches, each with its own lane; o_part/o_wpart of a
  // lane are adjacent and together form the split-K scratch of a paired launch.
  struct Lane {
    size_t o_gdec[6], o_genc[6];  // input gradients of decoder / encoder layer i (kept until the deferred sums ran)
    size_t o_dcat1, o_dz1, o_dzmu, o_dcat3, o_dd0, o_part, o_wpart;
    size_t o_defer;               // scratch of the backward's deferred reductions (GradJobs)
  } lane[3];  // 0, 1: netT, netF; 2: the 2-D autoencoder (its backward may start while 0/1 are still being read)
  size_t defer_floats;
  // optional side stream: netF beside netT (LSHM_FORK=1)
  hipStream_t wstream;
  hipStream_t lstream = nullptr;  // latent-space terms (K-harmonic, similarity, augmentation, RICA): beside everything
  std::vector<hipEvent_t> events;
  mutable size_t next_event;
  // paired forwards: the two chains run the same parameters, so the no-grad chain reads the closure chain's fragment-ordered
  // copy of the deep weights instead of making its own at the head of the critical chain (two_forwards)
  mutable const float* pack_override = nullptr;
  mutable bool skip_pack = false;
  mutable hipEvent_t pack_event = nullptr;
  mutable bool col_written = true;  // false: the last closure forward kept only the row image of the residual (conv0_bwd_tile reads nothing else)
  hipEvent_t latent_event;  // set while the latent-space terms of the current forward are in flight
  bool sim_started = false;  // cluster_similarity of the current forward already launched (side stream)
  bool recon_ready;         // the workspace already holds the reconstruction terms of the next closure
  bool latent_early = false;  // the latent-space terms of the next closure are already in flight (started beside the paired forwards)
  bool sum7_pending = false;  // ... as per-block partials: their seven sums are made on the latent-space stream of the next closure
  size_t o_latent_ws, latent_ws_floats;
  bool side_ok;
  bool pair_mode;  // netT/netF share launches (default) instead of running on two streams (LSHM_FORK=1)
  bool side_wgrad; // weight-gradient chain on the side stream (default; LSHM_WGRAD_INLINE=1 turns it off)
  hipEvent_t take_event() const { return events[next_event++ % events.size()]; }
  // LSHM_PHASE_EVENTS=1 (diagnostic, lshm_engine_phase_times): timestamps at the phase boundaries of an iteration, unprofiled.
  // Each is a marker packet (~6 us of idle queue on its stream), so the instrumented iteration is ~0.05 ms slower.
  enum { PH_CLOSURE = 0, PH_BWD1D, PH_BWD_MAIN_END, PH_BWD_SIDE_END, PH_CLOSURE_END, PH_UPDATE, PH_FWD_CLOSURE_END, PH_FWD_NOGRAD_END,
         PH_RECON_END, PH_UPDATE_END, PH_COUNT };
  std::vector<hipEvent_t> phase;
  void mark(int which, hipStream_t s) const { if (!phase.empty() && !in_capture) (void)hipEventRecord(phase[which], s); }
  size_t part_floats;
  size_t ws_floats;
  int device;      // HIP device current at creation (-1: none); the side stream and the events live there
  int bf;          // bf16 storage of the image-sized activations / gradients (LSHM_PRECISION_BF16_STORAGE)
  // data parallelism inside the engine (lshm_engine_set_comm): the closures all-reduce their own results
  lshm_comm* comm = nullptr;
  hipStream_t cstream = nullptr;  // the early bucket (netT / netF gradients) runs here, beside the 2-D backward
  long off1d = 0;                 // arena offset of the first netT tensor: [0, off1d) net, [off1d, Moff) netT+netF
  bool early_ok = true;           // ranks agree on the early bucket (lshm_engine_set_early_bucket); else one group at the end
  bool in_capture = false;        // the stream of the current call is being captured (set by ENGINE_ENTER)
  unsigned last_flags = 0;        // LSHM_ENGINE_USED_*: what the last call actually did (tests)
  const void* seen_ptr[16] = {};  // device pointers already checked to live on this engine's device (ring: old entries are re-checked)
  int nseen = 0;
  unsigned seen_total = 0;
};

namespace lshm {

static long add_param(lshm_engine* e, const std::string& name, std::initializer_list<long> shape) {
  ParamInfo p;
  p.name = name;
  p.ndim = (int)shape.size();
  p.numel = 1;
  int i = 0;
  for (long s : shape) { p.shape[i++] = s; p.numel *= s; }
  for (; i < 4; ++i) p.shape[i] = 1;
  p.offset = e->nparams;
  e->nparams += (p.numel + 3) / 4 * 4;
  e->params.push_back(p);
  return p.offset;
}

// Every buffer of the workspace starts on a 256-byte boundary: image rows are 128-byte (32-float) tile
// segments, and a buffer that starts 32 bytes into a cache line makes each of them straddle two lines
// (measured on the reconstruction pass: 207 us against 140 us for the same launch on aligned buffers).
static size_t take(size_t& cur, size_t n) {
  const size_t o = cur;
  cur += (n + 63) / 64 * 64;
  return o;
}

static void plan_ae(lshm_engine* e, int idx, const char* prefix, int ndim, int L, int mu_col,
                    size_t& cur) {
  const lshm_step_config& c = e->cfg;
  AEPlan& a = e->ae[idx];
  a.ndim = ndim;
  a.L = L;
  a.mu_col = mu_col;
  const int B = c.B, P = c.P, hd = e->hdim;
  int ch[7];
  for (int i = 0; i < 7; ++i) ch[i] = CHL[i];
  ch[0] = c.C;
  const std::string pre(prefix);
  for (int i = 0; i < 6; ++i) {
    const std::string n = pre + ".conv" + std::to_string(i);
    if (ndim == 2) a.cw[i] = add_param(e, n + ".weight", {ch[i + 1], ch[i], 4, 4});
    else a.cw[i] = add_param(e, n + ".weight", {ch[i + 1], ch[i], 4});
    a.cb[i] = add_param(e, n + ".bias", {ch[i + 1]});
  }
  a.fcuv1w = add_param(e, pre + ".fcuv1.weight", {hd, hd});
  a.fcuv1b = add_param(e, pre + ".fcuv1.bias", {hd});
  a.fcuv3w = add_param(e, pre + ".fcuv3.weight", {hd, hd});
  a.fcuv3b = add_param(e, pre + ".fcuv3.bias", {hd});
  a.fc1w = add_param(e, pre + ".fc1.weight", {L, 768 + hd});
  a.fc1b = add_param(e, pre + ".fc1.bias", {L});
  if (c.rica) {
    a.fc2inw = add_param(e, pre + ".fc2in.weight", {L, L});
    a.fc2inb = add_param(e, pre + ".fc2in.bias", {L});
    a.fc2outw = add_param(e, pre + ".fc2out.weight", {L, L});
    a.fc2outb = add_param(e, pre + ".fc2out.bias", {L});
  }
  a.fc3w = add_param(e, pre + ".fc3.weight", {768, L + hd});
  a.fc3b = add_param(e, pre + ".fc3.bias", {768});
  for (int i = 0; i < 6; ++i) {
    const std::string n = pre + ".tconv" + std::to_string(i);
    if (ndim == 2) a.tw[i] = add_param(e, n + ".weight", {ch[6 - i], ch[5 - i], 4, 4});
    else a.tw[i] = add_param(e, n + ".weight", {ch[6 - i], ch[5 - i], 4});
    a.tb[i] = add_param(e, n + ".bias", {ch[5 - i]});
  }
  // layers + activation workspace
  const long PP = (long)P * P;
  for (int i = 0; i < 6; ++i) {
    ConvLayer& Le = a.enc[i];
    Le.kind = ndim == 2 ? 0 : 2;
    Le.B = B; Le.Cin = ch[i]; Le.Cout = ch[i + 1];
    if (ndim == 2) { Le.Hin = P >> i; Le.Win = P >> i; }
    else { Le.Hin = 1; Le.Win = (int)(PP >> (2 * i)); }
    int Ho, Wo;
    conv_out_dims(Le, Ho, Wo);
    Le.in_bs = (long)Le.Cin * Le.Hin * Le.Win;
    Le.out_bs = (i == 5) ? (768 + hd) : (long)Le.Cout * Ho * Wo;
    if (i < 5) a.act[i] = take(cur, (size_t)B * Le.out_bs);
    ConvLayer& Ld = a.dec[i];
    Ld.kind = ndim == 2 ? 1 : 3;
    Ld.B = B; Ld.Cin = ch[6 - i]; Ld.Cout = ch[5 - i];
    if (ndim == 2) { Ld.Hin = 2 << i; Ld.Win = 2 << i; }
    else { Ld.Hin = 1; Ld.Win = 4 << (2 * i); }
    conv_out_dims(Ld, Ho, Wo);
    Ld.in_bs = (long)Ld.Cin * Ld.Hin * Ld.Win;
    Ld.out_bs = (long)Ld.Cout * Ho * Wo;
    if (i < 5) a.dact[i] = take(cur, (size_t)B * Ld.out_bs);
  }
  a.cat1 = take(cur, (size_t)B * (768 + hd));
  a.z1 = take(cur, (size_t)B * L);
  a.cat3 = take(cur, (size_t)B * (L + hd));
  a.d0 = take(cur, (size_t)B * 768);
  a.out = take(cur, (size_t)B * c.C * PP);
}

// A forward pass is built as a list of steps, one launch each, every step a function of (workspace base, stream):
// the same list can then be enqueued once (the usual forward) or twice in lock step on two streams with two
// workspace bases (lshm_engine_multiplier_update_next_ex: two independent forwards side by side -- enqueueing one
// whole chain before the other would leave the second stream empty for the ~0.2 ms the host needs per chain).
struct Src {  // an input tensor: absolute (the caller's x) or relative to the workspace base
  const float* abs;
  size_t off;
  const float* at(const float* ws) const { return abs ? abs : ws + off; }
};

// Forward of one autoencoder (G == 1) or of two autoencoders of identical shape that share every
// launch (G == 2: netT and netF).  idx[] = AE indices, input[] = their input tensors.
// latent_mark: index of the first step after the latents are complete; output_mark: index of the step that writes
// the reconstruction itself (the last decoder layer)
static void ae_forward_steps(const lshm_engine* e, int G, const int* idx, const float* prm, const Src* input,
                             int ln,  // ln: scratch slot of problem 0
                             std::vector<FwdStep>& steps, size_t* latent_mark, size_t* output_mark) {
  const lshm_step_config& c = e->cfg;
  const int i0 = idx[0], i1 = G > 1 ? idx[1] : idx[0];
  const Src s0 = input[0], s1 = G > 1 ? input[1] : input[0];
  const int B = c.B, hd = e->hdim, L = e->ae[i0].L, D = e->D;
  const size_t pf = e->part_floats * (G > 1 ? 2 : 1);
  const size_t o_part = e->o_fpart + (size_t)ln * 2 * e->part_floats;  // a pair uses two adjacent scratch regions
  auto A = [e, i0, i1](int g) -> const AEPlan& { return e->ae[g ? i1 : i0]; };
  // mid layers of the 1-D autoencoders as LDS-resident chains (chain1d.hip): conv2 -> conv3 -> conv4 and
  // tconv1 -> tconv2 -> tconv3, one launch each instead of three
  const AEPlan& a0 = e->ae[i0];
  const int chd[4] = {a0.enc[2].Cin, a0.enc[2].Cout, a0.enc[3].Cout, a0.enc[4].Cout};
  const int chu[4] = {a0.dec[1].Cin, a0.dec[1].Cout, a0.dec[2].Cout, a0.dec[3].Cout};
  const bool chain_dn = a0.ndim == 1 && conv1d_chain_supported(false, chd, a0.enc[2].Win);
  const bool chain_up = a0.ndim == 1 && conv1d_chain_supported(true, chu, a0.dec[1].Win);
  // 2-D autoencoder: conv3 .. tconv3 (eleven layers) as one launch over a fragment-ordered copy of their weights (deep2d.hip);
  // the copy is made at the head of the forward (the parameters may have changed since the last one)
  const bool deep = G == 1 && a0.ndim == 2 && e->deep2d;
  if (deep) {
    steps.push_back([=](float* ws, hipStream_t st) -> int {
      const AEPlan& a = A(0);
      const Deep2dWeights w{prm + a.cw[2], prm + a.cw[3], prm + a.cw[4], prm + a.cw[5], prm + a.fc1w, prm + a.fc2inw, prm + a.fc2outw,
                            prm + a.fc3w, prm + a.tw[0], prm + a.tw[1], prm + a.tw[2], prm + a.tw[3]};
      if (e->skip_pack) return LSHM_OK;
      const int rc = deep2d_pack(w, ws + e->o_pack2d, 0, e->deep_bf16, st);
      if (rc == LSHM_OK && e->pack_event && hipEventRecord(e->pack_event, st) != hipSuccess) {
        set_last_error("engine: event record failed");
        return LSHM_ERR_ARG;
      }
      return rc;
    });
  }
  // 1-D autoencoders: conv2 .. tconv3 (twelve layers) as one launch (chain1d_full.hip)
  const bool full1d = a0.ndim == 1 && e->full1d;
  for (int i = 0; i < 6; ++i) {
    if (full1d && i == 2) {
      steps.push_back([=](float* ws, hipStream_t st) -> int {
        Chain1dFullArgs q{};
        for (int g = 0; g < 2; ++g) {
          const AEPlan& a = A(g < G ? g : 0);
          q.in[g] = ws + a.act[1];
          for (int k = 0; k < 3; ++k) {
            q.dn[k].w[g] = prm + a.cw[2 + k]; q.dn[k].bias[g] = prm + a.cb[2 + k]; q.dn[k].out[g] = ws + a.act[2 + k]; q.dn[k].dact[g] = nullptr;
            q.up[k].w[g] = prm + a.tw[1 + k]; q.up[k].bias[g] = prm + a.tb[1 + k]; q.up[k].out[g] = ws + a.dact[1 + k]; q.up[k].dact[g] = nullptr;
          }
          q.w5[g] = prm + a.cw[5]; q.b5[g] = prm + a.cb[5]; q.cat1[g] = ws + a.cat1;
          q.fc1w[g] = prm + a.fc1w; q.fc1b[g] = prm + a.fc1b; q.fc2inw[g] = prm + a.fc2inw; q.fc2inb[g] = prm + a.fc2inb;
          q.fc2outw[g] = prm + a.fc2outw; q.fc2outb[g] = prm + a.fc2outb; q.fc3w[g] = prm + a.fc3w; q.fc3b[g] = prm + a.fc3b;
          q.z1[g] = ws + a.z1; q.mu[g] = ws + e->o_Mu + a.mu_col; q.cat3[g] = ws + a.cat3; q.d0[g] = ws + a.d0;
          q.wt0[g] = prm + a.tw[0]; q.bt0[g] = prm + a.tb[0]; q.t0[g] = ws + a.dact[0];
        }
        q.in_bs = A(0).enc[2].in_bs;
        q.mu_ld = D;
        for (int k = 0; k < 3; ++k) {
          q.dn[k].out_bs = A(0).enc[2 + k].out_bs; q.dn[k].act = 1;
          q.up[k].out_bs = A(0).dec[1 + k].out_bs; q.up[k].act = 1;
        }
        return chain1d_full_fwd(q, B, G, st);
      });
      break;
    }
    if (deep && i == 3) {
      steps.push_back([=](float* ws, hipStream_t st) -> int {
        const AEPlan& a = A(0);
        Deep2dIO io;
        io.x2 = ws + a.act[2];
        io.b3 = prm + a.cb[3]; io.b4 = prm + a.cb[4]; io.b5 = prm + a.cb[5];
        io.bfc1 = prm + a.fc1b; io.bfc2in = prm + a.fc2inb; io.bfc2out = prm + a.fc2outb; io.bfc3 = prm + a.fc3b;
        io.bt0 = prm + a.tb[0]; io.bt1 = prm + a.tb[1]; io.bt2 = prm + a.tb[2]; io.bt3 = prm + a.tb[3];
        io.a3 = ws + a.act[3]; io.a4 = ws + a.act[4]; io.cat1 = ws + a.cat1; io.z1 = ws + a.z1;
        io.mu = ws + e->o_Mu + a.mu_col; io.mu_ld = D; io.cat3 = ws + a.cat3; io.d0 = ws + a.d0;
        io.t0 = ws + a.dact[0]; io.t1 = ws + a.dact[1]; io.t2 = ws + a.dact[2]; io.t3 = ws + a.dact[3];
        if (e->pack_override && e->pack_event && hipStreamWaitEvent(st, e->pack_event, 0) != hipSuccess) {
          set_last_error("engine: stream join failed");
          return LSHM_ERR_ARG;
        }
        return deep2d_fwd(io, e->pack_override ? e->pack_override : ws + e->o_pack2d, B, e->deep_variant + 4 * e->deep_bf16, st);
      });
      break;
    }
    if (chain_dn && i == 2) {
      steps.push_back([=](float* ws, hipStream_t st) -> int {
        Chain1dStage cs[3];
        for (int k = 0; k < 3; ++k) {
          for (int g = 0; g < 2; ++g) {
            const AEPlan& a = A(g < G ? g : 0);
            cs[k].w[g] = prm + a.cw[2 + k]; cs[k].bias[g] = prm + a.cb[2 + k];
            cs[k].out[g] = ws + a.act[2 + k]; cs[k].dact[g] = nullptr;
          }
          cs[k].out_bs = A(0).enc[2 + k].out_bs;
          cs[k].act = 1;
        }
        return conv1d_chain(false, cs, ws + A(0).act[1], G > 1 ? ws + A(1).act[1] : nullptr, A(0).enc[2].in_bs, 1, B, st);
      });
      i = 4;
      continue;
    }
    steps.push_back([=](float* ws, hipStream_t st) -> int {
      ConvFwdIO io[2];
      for (int g = 0; g < G; ++g) {
        const float* in = i == 0 ? (g ? s1 : s0).at(ws) : ws + A(g).act[i - 1];
        float* out = (i < 5) ? ws + A(g).act[i] : ws + A(g).cat1;
        io[g] = ConvFwdIO{in, prm + A(g).cw[i], prm + A(g).cb[i], out};
      }
      return conv_layer_fwd(A(0).enc[i], io[0], 1, ws + o_part, pf, st, G > 1 ? &io[1] : nullptr);
    });
  }
  // dense layers: (input offset, weight, bias, output offset) per problem; offsets relative to the workspace
  struct Lin { size_t x[2], y[2]; long w[2], b[2]; long ldx, ldy; int K, N, act; };
  auto lin = [&](const Lin& q) {
    steps.push_back([=](float* ws, hipStream_t st) -> int {
      LinFwdIO l[2];
      for (int g = 0; g < G; ++g) l[g] = LinFwdIO{ws + q.x[g], prm + q.w[g], prm + q.b[g], ws + q.y[g]};
      return linear_fwd(l[0], q.ldx, q.ldy, B, q.K, q.N, q.act, ws + o_part, pf, st, G > 1 ? &l[1] : nullptr);
    });
  };
  auto both = [&](auto f) { Lin q{}; for (int g = 0; g < 2; ++g) f(q, g, A(g)); return q; };
  // (elu(fcuv1(uvh)) and elu(fcuv3(uvh)) are already in cat1 / cat3: the uv_features step)
  const bool dense_chain = dense1d_supported(L, hd, c.rica);  // latent width 16 (netT / netF) or 256 (the 2-D autoencoder)
  if (deep || full1d) {
    if (latent_mark) *latent_mark = steps.size();
  } else if (dense_chain) {
    // fc1 -> fc2in -> fc2out -> fc3 of an autoencoder as one launch (dense1d.hip); the latents are complete inside it,
    // so the latent-space terms start right after it
    steps.push_back([=](float* ws, hipStream_t st) -> int {
      Dense1dFwdIO io[2];
      for (int g = 0; g < G; ++g) {
        const AEPlan& a = A(g);
        io[g] = Dense1dFwdIO{ws + a.cat1, prm + a.fc1w, prm + a.fc1b, prm + a.fc2inw, prm + a.fc2inb, prm + a.fc2outw,
                             prm + a.fc2outb, prm + a.fc3w, prm + a.fc3b, ws + a.z1, ws + e->o_Mu + a.mu_col, ws + a.cat3,
                             ws + a.d0};
      }
      return dense1d_fwd(io[0], G > 1 ? &io[1] : nullptr, D, B, st, L);
    });
    if (latent_mark) *latent_mark = steps.size();
  } else if (c.rica) {
    Lin q = both([&](Lin& q, int g, const AEPlan& a) { q.x[g] = a.cat1; q.w[g] = a.fc1w; q.b[g] = a.fc1b; q.y[g] = a.z1; });
    q.ldx = 768 + hd; q.ldy = L; q.K = 768 + hd; q.N = L; q.act = 1;
    lin(q);
    q = both([&](Lin& q, int g, const AEPlan& a) { q.x[g] = a.z1; q.w[g] = a.fc2inw; q.b[g] = a.fc2inb; q.y[g] = e->o_Mu + a.mu_col; });
    q.ldx = L; q.ldy = D; q.K = L; q.N = L; q.act = 1;
    lin(q);
    q = both([&](Lin& q, int g, const AEPlan& a) { q.x[g] = e->o_Mu + a.mu_col; q.w[g] = a.fc2outw; q.b[g] = a.fc2outb; q.y[g] = a.cat3; });
    q.ldx = D; q.ldy = L + hd; q.K = L; q.N = L; q.act = 1;
    lin(q);
  } else {
    Lin q = both([&](Lin& q, int g, const AEPlan& a) { q.x[g] = a.cat1; q.w[g] = a.fc1w; q.b[g] = a.fc1b; q.y[g] = e->o_Mu + a.mu_col; });
    q.ldx = 768 + hd; q.ldy = D; q.K = 768 + hd; q.N = L; q.act = 1;
    lin(q);
    for (int g = 0; g < G; ++g) {
      const size_t src = e->o_Mu + A(g).mu_col, dst = A(g).cat3;
      steps.push_back([=](float* ws, hipStream_t st) -> int { return copy2d(ws + src, D, ws + dst, L + hd, B, L, st); });
    }
  }
  if (!dense_chain && !deep && !full1d) {
    if (latent_mark) *latent_mark = steps.size();
    Lin q = both([&](Lin& q, int g, const AEPlan& a) { q.x[g] = a.cat3; q.w[g] = a.fc3w; q.b[g] = a.fc3b; q.y[g] = a.d0; });
    q.ldx = L + hd; q.ldy = 768; q.K = L + hd; q.N = 768; q.act = 0;
    lin(q);
  }
  for (int i = (deep || full1d) ? 4 : 0; i < 6; ++i) {
    if (i == 5 && output_mark) *output_mark = steps.size();
    if (chain_up && i == 1) {
      steps.push_back([=](float* ws, hipStream_t st) -> int {
        Chain1dStage cs[3];
        for (int k = 0; k < 3; ++k) {
          for (int g = 0; g < 2; ++g) {
            const AEPlan& a = A(g < G ? g : 0);
            cs[k].w[g] = prm + a.tw[1 + k]; cs[k].bias[g] = prm + a.tb[1 + k];
            cs[k].out[g] = ws + a.dact[1 + k]; cs[k].dact[g] = nullptr;
          }
          cs[k].out_bs = A(0).dec[1 + k].out_bs;
          cs[k].act = 1;
        }
        return conv1d_chain(true, cs, ws + A(0).dact[0], G > 1 ? ws + A(1).dact[0] : nullptr, A(0).dec[1].in_bs, 0, B, st);
      });
      i = 3;
      continue;
    }
    steps.push_back([=](float* ws, hipStream_t st) -> int {
      ConvFwdIO io[2];
      for (int g = 0; g < G; ++g) {
        const float* in = i == 0 ? ws + A(g).d0 : ws + A(g).dact[i - 1];
        float* out = (i < 5) ? ws + A(g).dact[i] : ws + A(g).out;
        io[g] = ConvFwdIO{in, prm + A(g).tw[i], prm + A(g).tb[i], out};
      }
      return conv_layer_fwd(A(0).dec[i], io[0], i < 5, ws + o_part, pf, st, G > 1 ? &io[1] : nullptr);
    });
  }
}

static int ae_forward(const lshm_engine* e, int G, const int* idx, const float* prm, const float* const* input,
                      float* ws, int ln, hipStream_t st,  // ln: scratch slot of problem 0
                      const std::function<int()>* after_latent = nullptr,  // called once the latents are enqueued
                      bool skip_output = false) {  // the reconstruction itself is not needed (only the saved activations)
  std::vector<FwdStep> steps;
  size_t lm = 0, om = 0;
  Src in[2];
  for (int g = 0; g < G; ++g) in[g] = Src{input[g], 0};
  ae_forward_steps(e, G, idx, prm, in, ln, steps, &lm, &om);
  for (size_t i = 0; i < steps.size(); ++i) {
    int rc;
    if (i == lm && after_latent && (rc = (*after_latent)())) return rc;
    if (i == om && skip_output) continue;
    if ((rc = steps[i](ws, st))) return rc;
  }
  return LSHM_OK;
}

// bf16 storage: the tile kernels of the 1-D pair's outermost layers take every image-sized tensor they touch as bf16, or none
static bool conv0_tile_bf_ok(const lshm_engine* e) {
  const ConvLayer& a = e->ae[1].enc[0];
  const ConvLayer& b = e->ae[2].enc[0];
  return !e->bf || (a.in_bf16 && a.out_bf16 && b.in_bf16 && b.out_bf16);
}
static bool dec5_bf_ok(const lshm_engine* e) {
  const ConvLayer& a = e->ae[1].dec[5];
  const ConvLayer& b = e->ae[2].dec[5];
  return !e->bf || (a.in_bf16 && a.out_bf16 && b.in_bf16 && b.out_bf16);
}

// The three forwards of the default (paired) schedule as one step list: harmonic features + the six layers fed by
// them alone, the 2-D autoencoder, the residual split, netT and netF as paired launches.
static void three_forward_steps(const lshm_engine* e, const float* prm, const float* x, const float* uv,
                                std::vector<FwdStep>& steps, size_t* latent_mark, size_t* output1d_mark,
                                size_t* resid_mark = nullptr, FwdStep* resid_conv0_step = nullptr,
                                FwdStep* resid_conv0_keep_step = nullptr) {
  const lshm_step_config& c = e->cfg;
  steps.push_back([=](float* ws, hipStream_t st) -> int {
    UvLayers ul;
    ul.n = 0;
    for (int a = 0; a < 3; ++a) {
      const AEPlan& A = e->ae[a];
      ul.w[ul.n] = prm + A.fcuv1w; ul.bias[ul.n] = prm + A.fcuv1b;
      ul.out[ul.n] = ws + A.cat1 + 768; ul.ld[ul.n] = 768 + e->hdim; ++ul.n;
      ul.w[ul.n] = prm + A.fcuv3w; ul.bias[ul.n] = prm + A.fcuv3b;
      ul.out[ul.n] = ws + A.cat3 + A.L; ul.ld[ul.n] = A.L + e->hdim; ++ul.n;
    }
    return uv_features(uv, c.scales, c.H, c.B, ws + e->o_uvh, ul, st);
  });
  const int i0[1] = {0};
  const Src in0[1] = {Src{x, 0}};
  ae_forward_steps(e, 1, i0, prm, in0, 0, steps, nullptr, nullptr);
  if (resid_mark) *resid_mark = steps.size();
  if (resid_conv0_step) {
    // a forward whose 1-D activations nobody reads afterwards may run conv0 of netT / netF straight from x and the 2-D
    // reconstruction (fp32 storage, the layer shapes of kharmonic_lofar.py)
    const AEPlan& aT = e->ae[1];
    const AEPlan& aF = e->ae[2];
    *resid_conv0_step = nullptr;
    // (bf16 storage: the reconstruction, the residual and conv0's outputs are bf16 tensors in both forms)
    const bool bf_ok = !e->bf || (aT.enc[0].in_bf16 && aT.enc[0].out_bf16 && aF.enc[0].in_bf16 && aF.enc[0].out_bf16);
    if (bf_ok && e->pair_mode && resid_conv0_supported(c.C, c.P, aT.enc[0].Cin, aT.enc[0].Cout, aT.enc[0].Win) &&
        aT.enc[0].out_bs == aF.enc[0].out_bs)
      *resid_conv0_step = [=](float* ws, hipStream_t st) -> int {
        return resid_conv0(x, ws + e->ae[0].out, prm + aT.cw[0], prm + aT.cb[0], ws + aT.act[0], prm + aF.cw[0], prm + aF.cb[0],
                           ws + aF.act[0], aT.enc[0].out_bs, c.B, st, e->bf);
      };
    if (resid_conv0_keep_step) {
      *resid_conv0_keep_step = nullptr;
      if (*resid_conv0_step && !sched(LSHM_SCHED_NO_RESID_CONV0_KEEP))
      {
        // the column-vectorised residual is only read by conv0's weight gradient: not written when that is the tile kernel
        const bool row_only = conv0_tile_bf_ok(e) && (e->pair_mode || !e->side_ok) && aT.enc[0].out_bs == aF.enc[0].out_bs &&
                              conv0_bwd_tile_supported(c.C, c.P, aT.enc[0].Cin, aT.enc[0].Cout, aT.enc[0].Win, aT.enc[0].in_bs);
        *resid_conv0_keep_step = [=](float* ws, hipStream_t st) -> int {
          e->col_written = !row_only;
          return resid_conv0(x, ws + e->ae[0].out, prm + aT.cw[0], prm + aT.cb[0], ws + aT.act[0], prm + aF.cw[0], prm + aF.cb[0],
                             ws + aF.act[0], aT.enc[0].out_bs, c.B, st, e->bf, ws + e->o_row, row_only ? nullptr : ws + e->o_col);
        };
      }
    }
  }
  steps.push_back([=](float* ws, hipStream_t st) -> int {
    e->col_written = true;
    return residual_split(x, ws + e->ae[0].out, ws + e->o_row, ws + e->o_col, c.B * c.C, c.P, st, e->bf);
  });
  const int i12[2] = {1, 2};
  const Src in12[2] = {Src{nullptr, e->o_row}, Src{nullptr, e->o_col}};
  ae_forward_steps(e, 2, i12, prm, in12, 0, steps, latent_mark, output1d_mark);
}

// Backward of one (G == 1) or two same-shape (G == 2) autoencoders.  dz_out[g]: gradient w.r.t. the
// AE output (B,C,P*P); gMu (B,D) holds the gradient w.r.t. the latents; dinput[g]: gradient w.r.t.
// the AE input, or null.  With G == 2, lane 0 holds problem 0's scratch and lane 1 problem 1's.
// Weight-gradient launches leave their closing sums (split-K slabs, workgroup partials, bias
// gradients) on a job list that two launches finish at the end; every dz therefore has its own buffer.
// netT / netF as a pair: where their input gradients go (gx1 = gx1p - (dT + dF^T) / 2, src/kharmonic_lofar.py:142-147).  Given, the
// backward of their first layer forms gx1 itself (conv0_bwd_tile.hip) and sets `done`; otherwise the caller combines dinput[].
struct CombineInto { const float* gx1p; float* gx1; bool done; bool dec5_done; };  // dec5_done: the reconstruction pass already made the last decoder layer's backward (recon_bwd5.hip)
static int ae_backward(const lshm_engine* e, int G, const int* idx, const float* prm, float* grd,
                       const float* const* input, const float* const* dz_out, float* const* dinput, float* ws,
                       int ln, hipStream_t st, hipStream_t wgrad_stream,
                       const std::function<int()>* before_dense = nullptr, const CombineInto* combine = nullptr) {
  const lshm_step_config& c = e->cfg;
  const AEPlan& a0 = e->ae[idx[0]];
  const int B = c.B, hd = e->hdim, L = a0.L, D = e->D;
  const float* uvh = ws + e->o_uvh;
  float* part = ws + e->lane[ln].o_part;  // a lane's two scratch regions are adjacent: 2x for a pair
  const size_t pf = e->part_floats * (G > 1 ? 2 : 1);
  auto A = [&](int g) -> const AEPlan& { return e->ae[idx[g]]; };
  auto LA = [&](int g) -> const lshm_engine::Lane& { return e->lane[ln + g]; };
  int rc;
  GradJobs jobs;
  jobs.scratch = ws + e->lane[ln].o_defer;
  jobs.cap = e->defer_floats;
  // the six dense layers' weight gradients (twelve for a pair) go as one launch once the last of their
  // gradients exists, instead of six latency-bound 5-10 us launches
  jobs.batch_dense = true;
  // Weight gradients need only dz and the saved input of their layer, and nothing waits for them
  // before the optimizer: with a side stream they form a second chain beside the data-gradient
  // chain, each link waiting for "dz is ready" on `st`; `st` itself never waits (the caller joins
  // once, after all backward passes).
  const bool side = wgrad_stream != nullptr && wgrad_stream != st;
  hipStream_t wst = side ? wgrad_stream : st;
  // "dz ready" without a marker packet on `st` (common.h: launch_stop_event): the launches of one layer on `st` carry an
  // event of their own as stop event; as long as nothing else has been put on `st` since, that event IS "everything on
  // `st` so far", and the side stream waits for it directly.  Otherwise (first release of a pass, after callbacks,
  // under capture, LSHM_SCHED_NO_STOP_EVENTS) an event is recorded as before.
  const bool stop_events = !(c.schedule & LSHM_SCHED_NO_STOP_EVENTS);
  const bool use_stop = side && stop_events && !e->in_capture;
  hipEvent_t last_ev = nullptr;
  auto on_st = [&](auto&& f) -> int {
    if (!use_stop) { last_ev = nullptr; return f(); }
    hipEvent_t ev = e->take_event();
    const unsigned before = launch_stop_count;
    int r;
    { StopEventScope sc(ev); r = f(); }
    if (launch_stop_count != before) last_ev = ev;  // (a call that launched nothing leaves the state as it was)
    return r;
  };
  auto dz_ready = [&]() -> int {
    if (!side) return LSHM_OK;
    if (last_ev) {
      if (hipStreamWaitEvent(wst, last_ev, 0) == hipSuccess) return LSHM_OK;
      set_last_error("engine: stream fork failed");
      return LSHM_ERR_ARG;
    }
    hipEvent_t ev = e->take_event();
    if (hipEventRecord(ev, st) != hipSuccess || hipStreamWaitEvent(wst, ev, 0) != hipSuccess) {
      set_last_error("engine: stream fork failed");
      return LSHM_ERR_ARG;
    }
    if (use_stop) last_ev = ev;
    return LSHM_OK;
  };
  // The small layers' weight gradients are released in groups: every release costs a barrier packet
  // on both queues (a few microseconds each), which for 5-10 us kernels is most of their run time.
  // (groups of 1 / 2 / 3 / 4: 2.224 / 2.220 / 2.240 / 2.255 ms per iteration once a release cost `st` nothing; at the end of
  //  round 3, with fewer and fused weight-gradient launches: 1 / 2 / 3 = 2.024 / 2.045 / 2.055 -- every layer on its own)
  constexpr int group = 1;
  std::vector<std::function<int()>> pending;
  auto release = [&](bool force) -> int {
    if (pending.empty() || (!force && (int)pending.size() < group)) return LSHM_OK;
    int r = dz_ready();
    for (auto& f : pending) if (!r) r = f();
    pending.clear();
    return r;
  };
  const float* dz[2];
  for (int g = 0; g < G; ++g) dz[g] = dz_out[g];
  // the mid layers of the 1-D autoencoders: data gradients of three layers as one LDS-resident chain (chain1d.hip)
  const int chd[4] = {a0.dec[3].Cout, a0.dec[3].Cin, a0.dec[2].Cin, a0.dec[1].Cin};  // tconv3 <- tconv2 <- tconv1: stride-4 conv direction
  const int chu[4] = {a0.enc[4].Cout, a0.enc[4].Cin, a0.enc[3].Cin, a0.enc[2].Cin};  // conv4 -> conv3 -> conv2: transposed direction
  const bool chain_bwd = !(c.schedule & LSHM_SCHED_NO_CHAIN1D_BWD);
  const bool chain_dec = chain_bwd && a0.ndim == 1 && conv1d_chain_supported(false, chd, a0.dec[3].Win * 4);
  const bool chain_enc = chain_bwd && a0.ndim == 1 && conv1d_chain_supported(true, chu, a0.enc[4].Win / 4);
  // 2-D autoencoder: the data gradients of tconv2 .. conv2 (eleven layers) as one launch (deep2d.hip); their weight
  // gradients follow behind it
  const bool deepb = G == 1 && a0.ndim == 2 && e->deep2d_bwd && dinput[0] == nullptr;
  if (deepb && !e->pack_bwd_done) {  // the fragment-ordered weight copy of the data-gradient pipeline, unless the latent-space stream made it
    const AEPlan& a = A(0);
    const Deep2dWeights w{prm + a.cw[2], prm + a.cw[3], prm + a.cw[4], prm + a.cw[5], prm + a.fc1w, prm + a.fc2inw, prm + a.fc2outw,
                          prm + a.fc3w, prm + a.tw[0], prm + a.tw[1], prm + a.tw[2], prm + a.tw[3]};
    if ((rc = deep2d_pack(w, ws + e->o_pack2d_bwd, 1, e->deep_bf16, st))) return rc;
  }
  // ---- decoder, last layer first
  int dec_from = 5;
  if (G == 2 && combine && combine->dec5_done) {
    // the reconstruction pass left the data gradients of netT / netF's last layer in the lanes' buffers and its weight-gradient
    // slabs in the workspace: only their closing sums are left
    if ((rc = recon_bwd5_close(ws + e->o_recon_w5, recon_bwd5_grid(B), grd + A(0).tw[5], grd + A(0).tb[5], grd + A(1).tw[5], grd + A(1).tb[5],
                               0, wst, &jobs))) return rc;
    for (int g = 0; g < G; ++g) dz[g] = ws + LA(g).o_gdec[5];
    dec_from = 4;
  } else if (G == 2 && combine && dec5_bf_ok(e) && a0.dec[5].in_bs == A(1).dec[5].in_bs && a0.dec[5].out_bs == (long)c.C * c.P * c.P &&
             A(1).dec[5].out_bs == a0.dec[5].out_bs && recon_bwd5_supported(c.C, c.P, a0.dec[5].Cin, a0.dec[5].Cout, a0.dec[5].Win)) {
    // ... or, when another pass made the gradient images (sequential forwards, a captured iteration, a line search's closure):
    // the same tiles on the same workgroups in the same order from those images -- every schedule gets the same bits
    if ((rc = on_st([&] {
           return tconv5_pair_bwd(dz[0], dz[1], ws + A(0).dact[4], ws + A(1).dact[4], a0.dec[5].in_bs, prm + A(0).tw[5], prm + A(1).tw[5], B,
                                  ws + LA(0).o_gdec[5], ws + LA(1).o_gdec[5], a0.dec[5].in_bs, ws + e->o_recon_w5, recon_bwd5_workspace_floats(), st,
                                  e->bf);
         }))) return rc;
    if ((rc = recon_bwd5_close(ws + e->o_recon_w5, recon_bwd5_grid(B), grd + A(0).tw[5], grd + A(0).tb[5], grd + A(1).tw[5], grd + A(1).tb[5],
                               0, wst, &jobs))) return rc;  // (closed with the other sums, behind the next "dz ready")
    for (int g = 0; g < G; ++g) dz[g] = ws + LA(g).o_gdec[5];
    dec_from = 4;
  }
  for (int i = dec_from; i >= 0; --i) {
    if (deepb && i == 2) break;
    if (chain_dec && i == 3) {
      // dz of tconv3 (12 channels) -> gradients w.r.t. the inputs of tconv3, tconv2, tconv1, each multiplied by ELU' of
      // that (saved) input; the three weight gradients follow on the other stream once the chain has written their dz
      Chain1dStage cs[3];
      for (int k = 0; k < 3; ++k) {  // stage k: layer 3 - k
        for (int g = 0; g < 2; ++g) {
          const int gg = g < G ? g : 0;
          cs[k].w[g] = prm + A(gg).tw[3 - k]; cs[k].bias[g] = nullptr;
          cs[k].out[g] = ws + LA(gg).o_gdec[3 - k]; cs[k].dact[g] = ws + A(gg).dact[2 - k];
        }
        cs[k].out_bs = a0.dec[3 - k].in_bs;
        cs[k].act = 0;
      }
      const float* dzin[2] = {dz[0], G > 1 ? dz[1] : nullptr};
      for (int k = 0; k < 3; ++k) {
        const int li = 3 - k;
        ConvWgradIO w0{ws + A(0).dact[li - 1], k == 0 ? dzin[0] : ws + LA(0).o_gdec[li + 1], grd + A(0).tw[li], grd + A(0).tb[li]};
        ConvWgradIO w1 = w0;
        if (G > 1) w1 = ConvWgradIO{ws + A(1).dact[li - 1], k == 0 ? dzin[1] : ws + LA(1).o_gdec[li + 1], grd + A(1).tw[li], grd + A(1).tb[li]};
        pending.push_back([&, li, w0, w1]() {
          return conv_layer_wgrad(a0.dec[li], w0, nullptr, 0, 0, wst, G > 1 ? &w1 : nullptr, &jobs);
        });
      }
      if ((rc = on_st([&] { return conv1d_chain(false, cs, dzin[0], dzin[1], a0.dec[3].out_bs, 0, B, st); }))) return rc;
      if ((rc = release(true))) return rc;
      for (int g = 0; g < G; ++g) dz[g] = ws + LA(g).o_gdec[1];
      i = 1;
      continue;
    }
    ConvWgradIO wg[2];
    ConvDgradIO dg[2];
    float* dx[2];
    for (int g = 0; g < G; ++g) {
      const float* xin = (i == 0) ? ws + A(g).d0 : ws + A(g).dact[i - 1];
      dx[g] = (i == 0) ? ws + LA(g).o_dd0 : ws + LA(g).o_gdec[i];
      wg[g] = ConvWgradIO{xin, dz[g], grd + A(g).tw[i], grd + A(g).tb[i]};
      // previous activation is an ELU output (except fc3's output feeding tconv0)
      dg[g] = ConvDgradIO{dz[g], prm + A(g).tw[i], dx[g], i == 0 ? nullptr : xin};
    }
    if (conv_layer_bwd_fusable(a0.dec[i], wg[0], dg[0]) && (G < 2 || conv_layer_bwd_fusable(a0.dec[i], wg[1], dg[1]))) {
      // outermost 1-D decoder layer: weight, bias and data gradient from one pass over dz and the saved input, on
      // the data-gradient stream (the closing sums still run on the other one, behind the next "dz ready" event)
      if ((rc = on_st([&] { return conv_layer_wgrad(a0.dec[i], wg[0], nullptr, 0, 0, st, G > 1 ? &wg[1] : nullptr, &jobs, &dg[0],
                                                    G > 1 ? &dg[1] : nullptr); }))) return rc;
      for (int g = 0; g < G; ++g) dz[g] = dx[g];
      continue;
    }
    pending.push_back([&, i, w0 = wg[0], w1 = wg[1]]() {
      return conv_layer_wgrad(a0.dec[i], w0, nullptr, 0, 0, wst, G > 1 ? &w1 : nullptr, &jobs);
    });
    if ((rc = release(i >= 4))) return rc;
    if ((rc = on_st([&] { return conv_layer_dgrad(a0.dec[i], dg[0], part, pf, st, G > 1 ? &dg[1] : nullptr); }))) return rc;
    for (int g = 0; g < G; ++g) dz[g] = dx[g];
  }
  LinWgradIO lw[2];
  LinDgradIO ld[2];
  int enc_from = 5;  // first encoder layer the loop below still has to differentiate
  std::vector<std::function<int()>> main_wgrads;  // weight gradients that follow the last data-gradient kernel on `st`
  auto wgrad = [&](long ldx, long lddz, int K, int N) {
    if (jobs.batch_dense)  // only parked here (no launch, no event): grad_jobs_launch_dense runs them all
      return linear_wgrad(lw[0], ldx, lddz, B, K, N, nullptr, 0, wst, G > 1 ? &lw[1] : nullptr, &jobs);
    pending.push_back([&, ldx, lddz, K, N, w0 = lw[0], w1 = lw[1]]() {
      return linear_wgrad(w0, ldx, lddz, B, K, N, nullptr, 0, wst, G > 1 ? &w1 : nullptr, &jobs);
    });
    return release(false);
  };
  // 1-D autoencoders: the four data gradients of the dense layers are one launch (dense1d.hip); the weight gradients
  // below still read the buffers it fills
  const bool dense_bwd_on = !(c.schedule & LSHM_SCHED_NO_DENSE1D_BWD);
  const bool dense_chain = deepb || (dense_bwd_on && dense1d_supported(L, hd, c.rica));  // (deepb: the chain below has every dense data gradient)
  auto dgrad = [&](long lddz, long lddx, long ldxs, long ldadd, int add_n, int K, int N) {
    if (dense_chain) return (int)LSHM_OK;
    return on_st([&] { return linear_dgrad(ld[0], lddz, lddx, ldxs, ldadd, add_n, B, K, N, part, pf, st, G > 1 ? &ld[1] : nullptr); });
  };
  // the latent-space gradient enters at fc3: whoever produced it beside the decoders is joined here, not earlier
  if (before_dense) { last_ev = nullptr; if ((rc = (*before_dense)())) return rc; }
  if (deepb) {
    const AEPlan& a = A(0);
    const lshm_engine::Lane& la = LA(0);
    Deep2dBwdIO io;
    io.g_t2 = dz[0];  // = o_gdec[3]: tconv3's data gradient, the last separate launch of the decoder
    io.s_t1 = ws + a.dact[1]; io.s_t0 = ws + a.dact[0]; io.s_cat3 = ws + a.cat3; io.s_mu = ws + e->o_Mu + a.mu_col; io.s_mu_ld = D;
    io.gmu = ws + e->o_gMu + a.mu_col; io.gmu_ld = D; io.s_z1 = ws + a.z1; io.s_cat1 = ws + a.cat1;
    io.s_c4 = ws + a.act[4]; io.s_c3 = ws + a.act[3]; io.s_c2 = ws + a.act[2]; io.s_c1 = ws + a.act[1];
    io.g_t1 = ws + la.o_gdec[2]; io.g_t0 = ws + la.o_gdec[1]; io.g_d0 = ws + la.o_dd0; io.g_cat3 = ws + la.o_dcat3;
    io.g_mu = ws + la.o_dzmu; io.g_mu_ld = L; io.g_z1 = ws + la.o_dz1; io.g_cat1 = ws + la.o_dcat1;
    io.g_c4 = ws + la.o_genc[5]; io.g_c3 = ws + la.o_genc[4]; io.g_c2 = ws + la.o_genc[3]; io.g_c1 = ws + la.o_genc[2];
    if ((rc = on_st([&] { return deep2d_bwd(io, ws + e->o_pack2d_bwd, B, 4 * e->deep_bf16, st); }))) return rc;
    // every dz of the eleven layers exists now: their weight gradients, released together
    // ... between the two streams: the data-gradient stream has only conv1's one-pass backward left, so a share of the
    // weight gradients follows that kernel there (bit k of the placement word: item k on the data-gradient stream)
    const bool batch_conv = !(c.schedule & LSHM_SCHED_NO_WGRAD_BATCH);
    // (batched: the batch goes to the weight-gradient stream right behind the chain, beside conv1's kernel; conv2's direct kernel
    //  -- item 6 -- follows conv1's kernel and the dense batch on the data-gradient stream instead: 1.727 -> 1.715 ms per iteration
    //  against the other way round, profiles/r04/README.md)
    const unsigned on_main = (side && !batch_conv) ? e->wgrad_on_main : (side && batch_conv) ? (1u << 6) : 0u;
    int item = 0;
    auto conv_w = [&](const ConvLayer& Lr, const float* xin, const float* dzp, long wo, long bo) {
      const ConvWgradIO w0{xin, dzp, grd + wo, grd + bo};
      if ((on_main >> item++) & 1u) main_wgrads.push_back([&jobs, &Lr, w0, st]() { return conv_layer_wgrad(Lr, w0, nullptr, 0, 0, st, nullptr, &jobs); });
      else pending.push_back([&jobs, &Lr, w0, wst]() { return conv_layer_wgrad(Lr, w0, nullptr, 0, 0, wst, nullptr, &jobs); });
    };
    conv_w(a0.dec[2], ws + a.dact[1], dz[0], a.tw[2], a.tb[2]);
    conv_w(a0.dec[1], ws + a.dact[0], ws + la.o_gdec[2], a.tw[1], a.tb[1]);
    conv_w(a0.dec[0], ws + a.d0, ws + la.o_gdec[1], a.tw[0], a.tb[0]);
    conv_w(a0.enc[5], ws + a.act[4], ws + la.o_dcat1, a.cw[5], a.cb[5]);
    conv_w(a0.enc[4], ws + a.act[3], ws + la.o_genc[5], a.cw[4], a.cb[4]);
    conv_w(a0.enc[3], ws + a.act[2], ws + la.o_genc[4], a.cw[3], a.cb[3]);
    conv_w(a0.enc[2], ws + a.act[1], ws + la.o_genc[3], a.cw[2], a.cb[2]);
    // the six implicit-GEMM ones park their problems; one launch runs them all
    jobs.batch_conv = batch_conv;
    // ... on the weight-gradient stream right behind the chain (its closing sums join the mid-way round); the data-gradient stream,
    // which has only conv1's one-pass backward left, then takes the dense batch and conv2's direct kernel, the latent-space stream
    // conv0's (below): the three streams finish ~120 us after the chain instead of ~250
    if (jobs.batch_conv) pending.push_back([&jobs, wst]() { return grad_jobs_launch_conv(jobs, wst); });
    enc_from = 1;
  }
  if (dense_chain && !deepb) {
    Dense1dBwdIO io[2];
    for (int g = 0; g < G; ++g)
      io[g] = Dense1dBwdIO{ws + LA(g).o_dd0, ws + A(g).cat3, ws + e->o_Mu + A(g).mu_col, ws + e->o_gMu + A(g).mu_col, ws + A(g).z1,
                           ws + A(g).cat1, prm + A(g).fc1w, prm + A(g).fc2inw, prm + A(g).fc2outw, prm + A(g).fc3w,
                           ws + LA(g).o_dcat3, ws + LA(g).o_dzmu, ws + LA(g).o_dz1, ws + LA(g).o_dcat1};
    if ((rc = on_st([&] { return dense1d_bwd(io[0], G > 1 ? &io[1] : nullptr, D, D, B, st, L); }))) return rc;
  }
  // ---- fc3 (no activation on its output): dd0 is its pre-activation gradient
  for (int g = 0; g < G; ++g) lw[g] = LinWgradIO{ws + A(g).cat3, ws + LA(g).o_dd0, grd + A(g).fc3w, grd + A(g).fc3b};
  if ((rc = wgrad(L + hd, 768, L + hd, 768))) return rc;
  // (rica == 0: latent == decoder input, so the latent-loss gradient is added before the ELU' multiply)
  for (int g = 0; g < G; ++g)
    ld[g] = LinDgradIO{ws + LA(g).o_dd0, prm + A(g).fc3w, ws + LA(g).o_dcat3, ws + A(g).cat3,
                       c.rica ? nullptr : ws + e->o_gMu + A(g).mu_col};
  if ((rc = dgrad(768, L + hd, L + hd, D, L, L + hd, 768))) return rc;
  for (int g = 0; g < G; ++g) lw[g] = LinWgradIO{uvh, ws + LA(g).o_dcat3 + L, grd + A(g).fcuv3w, grd + A(g).fcuv3b};
  if ((rc = wgrad(hd, L + hd, hd, hd))) return rc;
  long ld_dzfc1;
  size_t LaneOff[2];  // offset of the pre-activation gradient of fc1's output inside the workspace
  if (c.rica) {
    for (int g = 0; g < G; ++g) lw[g] = LinWgradIO{ws + e->o_Mu + A(g).mu_col, ws + LA(g).o_dcat3, grd + A(g).fc2outw, grd + A(g).fc2outb};
    if ((rc = wgrad(D, L + hd, L, L))) return rc;
    for (int g = 0; g < G; ++g)
      ld[g] = LinDgradIO{ws + LA(g).o_dcat3, prm + A(g).fc2outw, ws + LA(g).o_dzmu, ws + e->o_Mu + A(g).mu_col,
                         ws + e->o_gMu + A(g).mu_col};
    if ((rc = dgrad(L + hd, L, D, D, L, L, L))) return rc;
    for (int g = 0; g < G; ++g) lw[g] = LinWgradIO{ws + A(g).z1, ws + LA(g).o_dzmu, grd + A(g).fc2inw, grd + A(g).fc2inb};
    if ((rc = wgrad(L, L, L, L))) return rc;
    for (int g = 0; g < G; ++g) ld[g] = LinDgradIO{ws + LA(g).o_dzmu, prm + A(g).fc2inw, ws + LA(g).o_dz1, ws + A(g).z1, nullptr};
    if ((rc = dgrad(L, L, L, 0, 0, L, L))) return rc;
    for (int g = 0; g < G; ++g) LaneOff[g] = LA(g).o_dz1;
    ld_dzfc1 = L;
  } else {
    for (int g = 0; g < G; ++g) LaneOff[g] = LA(g).o_dcat3;
    ld_dzfc1 = L + hd;
  }
  for (int g = 0; g < G; ++g) lw[g] = LinWgradIO{ws + A(g).cat1, ws + LaneOff[g], grd + A(g).fc1w, grd + A(g).fc1b};
  if ((rc = wgrad(768 + hd, ld_dzfc1, 768 + hd, L))) return rc;
  for (int g = 0; g < G; ++g) ld[g] = LinDgradIO{ws + LaneOff[g], prm + A(g).fc1w, ws + LA(g).o_dcat1, ws + A(g).cat1, nullptr};
  if ((rc = dgrad(ld_dzfc1, 768 + hd, 768 + hd, 0, 0, 768 + hd, L))) return rc;
  for (int g = 0; g < G; ++g) lw[g] = LinWgradIO{uvh, ws + LA(g).o_dcat1 + 768, grd + A(g).fcuv1w, grd + A(g).fcuv1b};
  if ((rc = wgrad(hd, 768 + hd, hd, hd))) return rc;
  // the decoder's and the dense layers' closing sums go now (side stream, behind their producers): the
  // tail after the last weight gradient then only has the encoder's
  // (deepb: the dense batch joins the batched conv launch on the data-gradient stream, which has only conv1's kernel left;
  //  the weight-gradient stream keeps the two direct kernels and the closing sums; the dense batch on the latent-space stream
  //  instead, beside conv1's kernel: +0.03 ms, profiles/r04/README.md)
  if (deepb && side && jobs.batch_conv) main_wgrads.insert(main_wgrads.begin(), [&jobs, st]() { return grad_jobs_launch_dense(jobs, st); });
  else pending.push_back([&]() { return grad_jobs_launch_dense(jobs, wst); });
  if ((rc = release(true))) return rc;
  // (deepb: every sum whose producer has been enqueued by now -- all but the batched launch's slabs and the last two layers' --
  //  runs here, beside the data-gradient stream's last kernels; the round at the end is then a few small jobs)
  if (side && (rc = grad_jobs_finish(jobs, wst))) return rc;
  // ---- encoder
  bool fused_tail = false;  // a fused kernel on `st` wrote partials that the closing sums on `wst` have not been ordered behind yet
  for (int g = 0; g < G; ++g) dz[g] = enc_from == 5 ? ws + LA(g).o_dcat1 : ws + LA(g).o_genc[enc_from + 1];
  for (int i = enc_from; i >= 0; --i) {
    if (chain_enc && i == 4) {
      // dz of conv4 (96 channels) -> gradients w.r.t. the inputs of conv4, conv3, conv2 (x ELU' of the saved inputs)
      Chain1dStage cs[3];
      for (int k = 0; k < 3; ++k) {  // stage k: layer 4 - k
        for (int g = 0; g < 2; ++g) {
          const int gg = g < G ? g : 0;
          cs[k].w[g] = prm + A(gg).cw[4 - k]; cs[k].bias[g] = nullptr;
          cs[k].out[g] = ws + LA(gg).o_genc[4 - k]; cs[k].dact[g] = ws + A(gg).act[3 - k];
        }
        cs[k].out_bs = a0.enc[4 - k].in_bs;
        cs[k].act = 0;
      }
      const float* dzin[2] = {dz[0], G > 1 ? dz[1] : nullptr};
      for (int k = 0; k < 3; ++k) {
        const int li = 4 - k;
        ConvWgradIO w0{ws + A(0).act[li - 1], k == 0 ? dzin[0] : ws + LA(0).o_genc[li + 1], grd + A(0).cw[li], grd + A(0).cb[li]};
        ConvWgradIO w1 = w0;
        if (G > 1) w1 = ConvWgradIO{ws + A(1).act[li - 1], k == 0 ? dzin[1] : ws + LA(1).o_genc[li + 1], grd + A(1).cw[li], grd + A(1).cb[li]};
        pending.push_back([&, li, w0, w1]() {
          return conv_layer_wgrad(a0.enc[li], w0, nullptr, 0, 0, wst, G > 1 ? &w1 : nullptr, &jobs);
        });
      }
      if ((rc = on_st([&] { return conv1d_chain(true, cs, dzin[0], dzin[1], a0.enc[4].out_bs, 1, B, st); }))) return rc;
      if ((rc = release(true))) return rc;
      fused_tail = false;
      for (int g = 0; g < G; ++g) dz[g] = ws + LA(g).o_genc[2];
      i = 2;
      continue;
    }
    if (i == 0 && G == 2 && combine && conv0_tile_bf_ok(e) && A(1).enc[0].out_bs == a0.enc[0].out_bs &&
        conv0_bwd_tile_supported(c.C, c.P, a0.enc[0].Cin, a0.enc[0].Cout, a0.enc[0].Win, a0.enc[0].in_bs)) {
      // first layer of netT and netF: both weight gradients, both data gradients and their combination with the
      // reconstruction terms' share into the 2-D autoencoder's output gradient, one pass over image tiles
      const size_t twf = conv0_bwd_tile_workspace_floats();
      float* tws = jobs.take(twf);
      if (!tws) { set_last_error("engine: deferred-sum scratch exhausted"); return LSHM_ERR_WORKSPACE; }
      if ((rc = on_st([&] {
             return conv0_bwd_tile(input[0], dz[0], dz[1], a0.enc[0].out_bs, prm + A(0).cw[0], prm + A(1).cw[0], combine->gx1p, combine->gx1,
                                   grd + A(0).cw[0], grd + A(0).cb[0], grd + A(1).cw[0], grd + A(1).cb[0], B, tws, twf, 0, st, &jobs, e->bf);
           }))) return rc;
      const_cast<CombineInto*>(combine)->done = true;
      fused_tail = true;
      break;
    }
    if (i == 0 && a0.ndim == 1 && !e->col_written) {
      set_last_error("engine: this workspace's forward kept only the row image of the residual (it ran under a schedule word with "
                     "the one-pass backward of conv0); run the forward again under the current word");
      return LSHM_ERR_UNSUPPORTED;
    }
    ConvWgradIO wg[2];
    ConvDgradIO dg[2];
    float* dx[2];
    for (int g = 0; g < G; ++g) {
      const float* xin = (i == 0) ? input[g] : ws + A(g).act[i - 1];
      dx[g] = (i == 0) ? dinput[g] : ws + LA(g).o_genc[i];
      wg[g] = ConvWgradIO{xin, dz[g], grd + A(g).cw[i], grd + A(g).cb[i]};
      dg[g] = ConvDgradIO{dz[g], prm + A(g).cw[i], dx[g], i == 0 ? nullptr : xin};
    }
    if (dx[0] && conv_layer_bwd_fusable(a0.enc[i], wg[0], dg[0]) && (G < 2 || conv_layer_bwd_fusable(a0.enc[i], wg[1], dg[1]))) {
      // outer 1-D encoder layers: weight, bias and data gradient from one pass over dz and the saved input, on the
      // data-gradient stream; their closing sums run on the other one, behind the "dz ready" event below
      if ((rc = on_st([&] { return conv_layer_wgrad(a0.enc[i], wg[0], nullptr, 0, 0, st, G > 1 ? &wg[1] : nullptr, &jobs, &dg[0],
                                                    G > 1 ? &dg[1] : nullptr); }))) return rc;
      fused_tail = true;
      for (int g = 0; g < G; ++g) dz[g] = dx[g];
      continue;
    }
    if (i == 0 && G == 1 && a0.ndim == 2 && side && e->lstream && !dinput[0] && !e->in_capture) {
      // the first layer's weight gradient -- the last kernel of the iteration's longer stream -- on the latent-space stream, idle by
      // now: it runs beside the weight-gradient stream's closing sums instead of behind them; the last round of sums waits for it
      if ((rc = release(true))) return rc;
      hipEvent_t ev = e->take_event();
      if (hipEventRecord(ev, st) != hipSuccess || hipStreamWaitEvent(e->lstream, ev, 0) != hipSuccess) {
        set_last_error("engine: stream fork failed");
        return LSHM_ERR_ARG;
      }
      if ((rc = conv_layer_wgrad(a0.enc[i], wg[0], nullptr, 0, 0, e->lstream, nullptr, &jobs))) return rc;
      hipEvent_t ev2 = e->take_event();
      if (hipEventRecord(ev2, e->lstream) != hipSuccess || hipStreamWaitEvent(wst, ev2, 0) != hipSuccess) {
        set_last_error("engine: stream join failed");
        return LSHM_ERR_ARG;
      }
      fused_tail = false;
      break;
    }
    pending.push_back([&, i, w0 = wg[0], w1 = wg[1]]() {
      return conv_layer_wgrad(a0.enc[i], w0, nullptr, 0, 0, wst, G > 1 ? &w1 : nullptr, &jobs);
    });
    if ((rc = release(i <= 2))) return rc;
    fused_tail = false;
    if (i == 0 && !dinput[0]) break;
    if ((rc = on_st([&] { return conv_layer_dgrad(a0.enc[i], dg[0], part, pf, st, G > 1 ? &dg[1] : nullptr); }))) return rc;
    for (int g = 0; g < G; ++g) dz[g] = dx[g];
  }
  if (!main_wgrads.empty()) {
    for (auto& f : main_wgrads) if ((rc = f())) return rc;
    fused_tail = true;  // their partials are on `st`: the closing sums on `wst` wait for an event recorded behind them
    last_ev = nullptr;
  }
  if (fused_tail && (rc = dz_ready())) return rc;
  return grad_jobs_finish(jobs, wst);
}

// scal layout (doubles): [0..6] sums7, [7] khm sum, [8] sim, [9..11] rica x3, [12] aug, [13..] aug partials
__global__ void finalize_terms_kernel(const double* __restrict__ scal, double* __restrict__ terms,
                                      double n_global, double rho, double khm_scale, int rica,
                                      const double* __restrict__ rica_part, int nrica) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double rsum = 0.0;
  if (rica) for (int i = 0; i < nrica; ++i) rsum += rica_part[i];
  const double l0 = scal[0] / n_global;
  const double l1 = (scal[1] + 0.5 * rho * scal[2]) / n_global;
  const double l2 = (scal[3] + 0.5 * rho * scal[4]) / n_global;
  const double l3 = (scal[5] + 0.5 * rho * scal[6]) / n_global;
  const double kd = scal[7] * khm_scale;
  const double sim = scal[8];
  const double rc = rica ? rsum : 0.0;
  const double aug = scal[12];
  terms[0] = l0; terms[1] = l1; terms[2] = l2; terms[3] = l3;
  terms[4] = kd; terms[5] = aug; terms[6] = sim; terms[7] = rc;
  terms[8] = l0 + l1 + l2 + l3 + kd + aug + sim + rc;
  // failure detection (the reference only guards LBFGS scalars, src/lbfgsnew.py:153,170,237,...): number of
  // logged terms that are NaN or infinite; under data parallelism the slot sums to the job-wide count
  int bad = 0;
  for (int i = 0; i < 8; ++i) bad += !(fabs(terms[i]) <= 1.79769313486231570e308);
  terms[9] = (double)bad;
}

// the step list of the paired schedule for these (params, x, uv), built once and kept: the pointers of a trainer do
// not change from call to call, and ~60 closures need not be re-made for every forward
static const lshm_engine::FwdPlan& forward_plan(lshm_engine* e, const float* prm, const float* x, const float* uv) {
  lshm_engine::FwdPlan& P = e->plan;
  if (P.prm != prm || P.x != x || P.uv != uv || P.steps.empty()) {
    P.steps.clear();
    P.prm = prm; P.x = x; P.uv = uv;
    three_forward_steps(e, prm, x, uv, P.steps, &P.latent_mark, &P.output1d_mark, &P.resid_mark, &P.resid_conv0, &P.resid_conv0_keep);
  }
  return P;
}

// skip_1d_output: the outputs of netT / netF are only consumed by the reconstruction pass; when that pass has
// already been made for this forward (LSHM_STEP_RECON_READY) their last decoder layer is not run
static int three_forward(lshm_engine* e, const float* prm, const float* x, const float* uv, float* ws,
                         hipStream_t st, const std::function<int()>* after_latents = nullptr,
                         bool skip_1d_output = false) {
  const lshm_step_config& c = e->cfg;
  int rc;
  e->recon_ready = false;  // a new forward: whatever reconstruction terms the workspace held are stale
  e->sum7_pending = false;
  e->latent_early = false;
  if (e->pair_mode || !e->side_ok) {  // every launch of netT / netF carries both problems
    const lshm_engine::FwdPlan& P = forward_plan(e, prm, x, uv);
    for (size_t i = 0; i < P.steps.size(); ++i) {
      if (i == P.latent_mark && after_latents && (rc = (*after_latents)())) return rc;
      if (i == P.output1d_mark && skip_1d_output) continue;
      if (P.resid_conv0_keep && (i == P.resid_mark || i == P.resid_mark + 1)) {  // residual split + conv0 pair: one launch
        if (i == P.resid_mark && (rc = P.resid_conv0_keep(ws, st))) return rc;
        continue;
      }
      if ((rc = P.steps[i](ws, st))) return rc;
    }
    return LSHM_OK;
  }
  // LSHM_FORK=1: netT and netF on two streams side by side (no faster than paired launches since the reductions are deferred)
  {  // harmonic features + the six layers that depend on them alone (fcuv1 / fcuv3 of net, netT, netF)
    UvLayers ul;
    ul.n = 0;
    for (int a = 0; a < 3; ++a) {
      const AEPlan& A = e->ae[a];
      ul.w[ul.n] = prm + A.fcuv1w; ul.bias[ul.n] = prm + A.fcuv1b;
      ul.out[ul.n] = ws + A.cat1 + 768; ul.ld[ul.n] = 768 + e->hdim; ++ul.n;
      ul.w[ul.n] = prm + A.fcuv3w; ul.bias[ul.n] = prm + A.fcuv3b;
      ul.out[ul.n] = ws + A.cat3 + A.L; ul.ld[ul.n] = A.L + e->hdim; ++ul.n;
    }
    if ((rc = uv_features(uv, c.scales, c.H, c.B, ws + e->o_uvh, ul, st))) return rc;
  }
  {
    const int i0[1] = {0};
    const float* in0[1] = {x};
    if ((rc = ae_forward(e, 1, i0, prm, in0, ws, 0, st))) return rc;
  }
  e->col_written = true;
  if ((rc = residual_split(x, ws + e->ae[0].out, ws + e->o_row, ws + e->o_col, c.B * c.C, c.P, st, e->bf))) return rc;
  const int i12[2] = {1, 2};
  const float* in12[2] = {ws + e->o_row, ws + e->o_col};
  hipEvent_t evf = e->take_event();
  if (hipEventRecord(evf, st) != hipSuccess || hipStreamWaitEvent(e->wstream, evf, 0) != hipSuccess) {
    set_last_error("engine: stream fork failed");
    return LSHM_ERR_ARG;
  }
  if ((rc = ae_forward(e, 1, i12, prm, in12, ws, 0, st))) return rc;
  if ((rc = ae_forward(e, 1, i12 + 1, prm, in12 + 1, ws, 1, e->wstream))) return rc;
  hipEvent_t evj = e->take_event();
  if (hipEventRecord(evj, e->wstream) != hipSuccess || hipStreamWaitEvent(st, evj, 0) != hipSuccess) {
    set_last_error("engine: stream join failed");
    return LSHM_ERR_ARG;
  }
  return after_latents ? (*after_latents)() : LSHM_OK;
}

// Two forwards of the same (params, x, uv) side by side: the host enqueues step k of the first on (ws_a, st_a), then
// step k of the second on (ws_b, st_b), so neither stream runs dry while the other chain is being enqueued.
// A forward alternates bandwidth-bound stretches (the outer layers, the residual split) with latency-bound ones
// (mid / deep / dense layers: a few microseconds per launch on a fraction of the machine); in lock step like meets
// like: the latency-bound stretches overlap almost for free, the bandwidth-bound ones share the HBM pipe.
// (Holding the second chain n steps behind the first, which pairs one chain's outer layers with the other's deep layers,
// was measured slower -- 0 / 6 / 12 / 24 steps: 2.28 / 2.32 / 2.39 / 2.49 ms, profiles/r03/README.md; again in round 4 with the
// deep section as one launch and a cross-stream gate: 0 / 2 / 3 / 4 / 5 / 6 / 8 steps: 1.88 / 1.89 / 1.90 / 1.90 / 1.92 / 2.02 /
// 2.04 ms, one patch per deep workgroup instead of two: +0.04, profiles/r04/README.md -- and removed.)
// The second forward does not need the reconstructions of netT / netF (skip_b_1d_output).
static int two_forwards(lshm_engine* e, const float* prm, const float* x, const float* uv, float* ws_a, hipStream_t st_a,
                        float* ws_b, hipStream_t st_b, bool skip_b_1d_output, bool skip_a_1d_output = false,
                        const std::function<int()>* after_latents_b = nullptr) {  // called once chain b's latents are enqueued
  e->recon_ready = false;
  e->sum7_pending = false;
  e->latent_early = false;
  const lshm_engine::FwdPlan& P = forward_plan(e, prm, x, uv);
  const int n = (int)P.steps.size();
  int rc;
  // two chains side by side: the closure chain's deep section takes two patches per workgroup (half of the CUs, every weight
  // fetched from L2 serves two patches; bit for bit the one-patch results), the no-grad chain's -- the one the reconstruction pass
  // and the backward wait for -- one patch per workgroup on all CUs (100 us alone against 145): no-grad / closure = 2/2, 1/2, 2/1,
  // 1/1 patches: 1.754 / 1.743 / 1.757 / 1.780 ms per iteration (profiles/r04/README.md)
  struct VariantScope {
    lshm_engine* e;
    explicit VariantScope(lshm_engine* en) : e(en) { e->deep_variant = 1; }
    ~VariantScope() { e->deep_variant = 0; }
  } variant_scope(e);
  // chain a is the forward whose activations are not kept (the no-grad forward): its residual split + conv0 pair is one launch
  const bool fused_a = skip_b_1d_output && P.resid_conv0 && P.resid_mark + 1 < (size_t)n;
  auto step_b = [&](int i) -> int {
    if ((size_t)i == P.latent_mark && after_latents_b && (rc = (*after_latents_b)())) return rc;
    if ((size_t)i == P.output1d_mark && skip_b_1d_output) return LSHM_OK;
    if (P.resid_conv0_keep && ((size_t)i == P.resid_mark || (size_t)i == P.resid_mark + 1))
      return (size_t)i == P.resid_mark ? P.resid_conv0_keep(ws_b, st_b) : LSHM_OK;
    return P.steps[i](ws_b, st_b);
  };
  // one fragment-ordered copy of the deep weights for both chains: chain b (the closure forward) makes it, chain a reads it
  // behind an event -- 8-30 us less at the head of the chain everything after the forwards waits for
  struct PackShare {
    const lshm_engine* e;
    PackShare(const lshm_engine* en, const float* packed, bool on) : e(en) {
      if (on) { e->pack_event = e->take_event(); share = packed; }
    }
    ~PackShare() { e->pack_override = nullptr; e->skip_pack = false; e->pack_event = nullptr; }
    void chain_a(bool in) const { e->pack_override = in ? share : nullptr; e->skip_pack = in && share; }
    const float* share = nullptr;
  } pack_share(e, ws_b + e->o_pack2d, e->deep2d && !sched(LSHM_SCHED_NO_SHARED_PACK));
  for (int i = 0; i < n; ++i) {
    {
      pack_share.chain_a(true);
      e->deep_variant = 0;  // (the no-grad chain: one patch per workgroup, see above)
      struct Leave { const PackShare& p; lshm_engine* e; ~Leave() { p.chain_a(false); e->deep_variant = 1; } } leave{pack_share, e};
      if (fused_a && (size_t)i == P.resid_mark) {
        // (nothing: the residual is formed on the fly by the next step)
      } else if (skip_a_1d_output && (size_t)i == P.output1d_mark) {
        // (nothing: the reconstruction pass forms x2 / x3c from this layer's input)
      } else if (fused_a && (size_t)i == P.resid_mark + 1) {
        if ((rc = P.resid_conv0(ws_a, st_a))) return rc;
      } else if ((rc = P.steps[i](ws_a, st_a))) {
        return rc;
      }
    }
    if ((rc = step_b(i))) return rc;
  }
  return LSHM_OK;
}

// Latent-space terms (src/kharmonic_lofar.py:160-172): they need only the three latent codes, so
// they can run on a side stream while the decoders and the reconstruction kernel run on the main one.
//   gMu = d/dMu (alpha*khm + gamma*aug + lambda*rica),  dM = alpha*khm' + beta*sim'
static int latent_losses(lshm_engine* e, const float* prm, float* grd, float* ws, hipStream_t st) {
  const lshm_step_config& c = e->cfg;
  const int B = c.B, D = e->D;
  const double world = c.world > 0 ? c.world : 1;
  double* scal = reinterpret_cast<double*>(ws + e->o_scal);
  float* gMu = ws + e->o_gMu;
  float* Mu = ws + e->o_Mu;
  float* dM = grd ? grd + e->Moff : ws + e->o_dMscratch;
  const float* M = prm + e->Moff;
  int rc;
  const double inv_count = 1.0 / (world * (double)B * c.K * D);
  // the centroid-similarity term depends on the parameters alone: with a side stream it was started at the
  // beginning of the forward (start_similarity) and already wrote dM; the K-harmonic term then adds to it
  const bool sim_done = e->sim_started;
  e->sim_started = false;
  if ((rc = khm_fwd_bwd(Mu, D, M, B, D, c.K, c.p, 1e-9f, inv_count, c.alpha, scal + 7, gMu, D, dM, 0,
                        ws + e->o_latent_ws, e->latent_ws_floats, st, sim_done ? 1 : 0))) return rc;
  if (!sim_done && (rc = cluster_sim_fwd_bwd(M, c.K, D, 1e-9f, (float)(c.beta / world), scal + 8, dM, 1, st))) return rc;
  const int bs_global = (int)(c.batch_size * world);
  {
    // augmented loss: local groups, global normalisation.  aug_loss_fwd_bwd normalises by
    // batch_size*bpb*bpb of the value passed: pass the global batch size, but only the local rows take part
    const int used = c.batch_size * c.bpb < B ? c.batch_size * c.bpb : B;
    if ((rc = aug_loss_fwd_bwd(Mu, D, used, D, c.bpb, bs_global, c.gamma, scal + 12, gMu, D, 1, st))) return rc;
  }
  if (c.rica) {
    double* rica_part = scal + 16 + (B + c.bpb - 1) / c.bpb;  // [LOGCOSH3_BLOCKS][3]
    const AEPlan* a = e->ae;
    int cols[3];
    float sc3[3];
    for (int i = 0; i < 3; ++i) {
      cols[i] = a[i].L;
      sc3[i] = (float)(c.rica_lambda / (world * (double)B * a[i].L));
    }
    if ((rc = logcosh3_fwd_bwd(Mu, D, B, cols, sc3, rica_part, LOGCOSH3_BLOCKS, gMu, D, st))) return rc;
  }
  return LSHM_OK;
}

// forward of the three autoencoders with the latent-space terms overlapped (side stream if there is one)
// latent-space terms on the side stream if there is one (joined in losses_and_backward), else in line
static int pending_sum7(lshm_engine* e, float* ws, hipStream_t st) {
  if (!e->sum7_pending) return LSHM_OK;
  e->sum7_pending = false;
  return recon_sum7(ws + e->o_recon_part, e->cfg.B * e->cfg.C, e->cfg.P, reinterpret_cast<double*>(ws + e->o_scal), st);
}
// The fragment-ordered weight copy of the backward's deep chain depends on the parameters alone: it is made on the
// latent-space stream, whose event the data-gradient stream waits for anyway, instead of in front of the 2-D backward.
static int pack_backward_weights(lshm_engine* e, const float* prm, float* ws, hipStream_t st) {
  e->pack_bwd_done = false;
  if (!e->deep2d_bwd) return LSHM_OK;
  const AEPlan& a = e->ae[0];
  const Deep2dWeights w{prm + a.cw[2], prm + a.cw[3], prm + a.cw[4], prm + a.cw[5], prm + a.fc1w, prm + a.fc2inw, prm + a.fc2outw,
                        prm + a.fc3w, prm + a.tw[0], prm + a.tw[1], prm + a.tw[2], prm + a.tw[3]};
  const int rc = deep2d_pack(w, ws + e->o_pack2d_bwd, 1, e->deep_bf16, st);
  e->pack_bwd_done = rc == LSHM_OK;
  return rc;
}
static int start_latent_losses(lshm_engine* e, const float* prm, float* grd, float* ws, hipStream_t st) {
  e->latent_event = nullptr;
  int rc;
  if (!(e->side_ok && e->side_wgrad)) {
    if ((rc = pending_sum7(e, ws, st))) return rc;
    return latent_losses(e, prm, grd, ws, st);
  }
  hipEvent_t ev = e->take_event();
  if (hipEventRecord(ev, st) != hipSuccess || hipStreamWaitEvent(e->lstream, ev, 0) != hipSuccess) {
    set_last_error("engine: stream fork failed");
    return LSHM_ERR_ARG;
  }
  if ((rc = pending_sum7(e, ws, e->lstream))) return rc;  // the sums of the reconstruction pass made with the previous forwards
  rc = latent_losses(e, prm, grd, ws, e->lstream);
  if (rc) return rc;
  if (grd && (rc = pack_backward_weights(e, prm, ws, e->lstream))) return rc;
  e->latent_event = e->take_event();
  if (hipEventRecord(e->latent_event, e->lstream) != hipSuccess) {
    set_last_error("engine: event record failed");
    return LSHM_ERR_ARG;
  }
  return LSHM_OK;
}
// cluster_similarity (src/lofar_models.py:214-229) needs only M: on the side stream it runs beside the whole
// forward instead of queueing behind the K-harmonic kernel (at K = 64 it is the longest latent-space kernel)
static int start_similarity(lshm_engine* e, const float* prm, float* grd, float* ws, hipStream_t st) {
  e->sim_started = false;
  if (!(e->side_ok && e->side_wgrad)) return LSHM_OK;
  const lshm_step_config& c = e->cfg;
  const double world = c.world > 0 ? c.world : 1;
  hipEvent_t ev = e->take_event();
  if (hipEventRecord(ev, st) != hipSuccess || hipStreamWaitEvent(e->lstream, ev, 0) != hipSuccess) {
    set_last_error("engine: stream fork failed");
    return LSHM_ERR_ARG;
  }
  double* scal = reinterpret_cast<double*>(ws + e->o_scal);
  float* dM = grd ? grd + e->Moff : ws + e->o_dMscratch;
  int rc = cluster_sim_fwd_bwd(prm + e->Moff, c.K, e->D, 1e-9f, (float)(c.beta / world), scal + 8, dM, 0, e->lstream);
  if (rc) return rc;
  e->sim_started = true;
  return LSHM_OK;
}
static int forward_with_latent_losses(lshm_engine* e, const float* prm, float* grd, const float* x, const float* uv,
                                      float* ws, hipStream_t st, bool skip_1d_output = false) {
  int rc0 = start_similarity(e, prm, grd, ws, st);
  if (rc0) return rc0;
  const std::function<int()> hook = [&]() -> int { return start_latent_losses(e, prm, grd, ws, st); };
  return three_forward(e, prm, x, uv, ws, st, &hook, skip_1d_output);
}

// reconstruction losses, loss terms and (when grd != null) every gradient, after forward_with_latent_losses
static int losses_and_backward(lshm_engine* e, const float* prm, float* grd, const float* x,
                               const float* y1, const float* y2, const float* y3, double* terms,
                               float* ws, hipStream_t st, bool recon_done = false) {
  const lshm_step_config& c = e->cfg;
  const int B = c.B, D = e->D, planes = c.B * c.C;
  const double world = c.world > 0 ? c.world : 1;
  const double n_global = world * (double)B * c.C * c.P * c.P;
  double* scal = reinterpret_cast<double*>(ws + e->o_scal);
  int rc;
  // reconstruction terms; the kernel's 1/n uses the local element count, rescale for world > 1 below
  // gradients carry this rank's share of the global mean (1/world folded into the kernel's 1/n)
  // (a gradient-free closure passes no gradient images: the kernel then only reads)
  if (!recon_done && (rc = recon_losses_fwd_bwd(x, ws + e->ae[0].out, ws + e->ae[1].out, ws + e->ae[2].out, y1, y2, y3,
                                 c.rho, planes, c.P, scal, grd ? ws + e->o_gx1p : nullptr, grd ? ws + e->o_gx2 : nullptr,
                                 grd ? ws + e->o_gx3c : nullptr, ws + e->o_recon_part, st, (float)(1.0 / world), e->bf))) return rc;
  const double inv_count = 1.0 / (world * (double)B * c.K * D);
  double* rica_part = scal + 16 + (B + c.bpb - 1) / c.bpb;  // [LOGCOSH3_BLOCKS][3]
  // The latent-space terms ran beside the decoders (their own stream).  Their scalars close the loss terms and
  // their gradient enters the backward pass at the dense layers: the 1-D decoders' convolution gradients,
  // which need neither, start before that chain has finished (it ends ~30 us after the decoders).
  const std::function<int()> join_latent = [&]() -> int {
    if (e->latent_event) {
      if (hipStreamWaitEvent(st, e->latent_event, 0) != hipSuccess) {
        set_last_error("engine: stream join failed");
        return LSHM_ERR_ARG;
      }
      e->latent_event = nullptr;
    }
    hipLaunchKernelGGL(finalize_terms_kernel, dim3(1), dim3(64), 0, st, scal, terms, n_global, (double)c.rho,
                       (double)c.alpha * inv_count, c.rica, rica_part, LOGCOSH3_BLOCKS * 3);
    return check_launch("finalize_terms");
  };
  const bool deferred = grd && (e->pair_mode || !e->side_ok);
  if (!deferred && (rc = join_latent())) return rc;
  if (!grd)  // gradient-free closure (line search): every rank needs the global loss to take the same branch
    return e->comm ? comm_allreduce_segments(e->comm, nullptr, nullptr, 0, terms, 10, st) : LSHM_OK;
  // backward: netT, netF (their input gradients feed AE1 through the residual), then AE1
  hipStream_t wgs = (e->pair_mode && e->side_ok && e->side_wgrad) ? e->wstream : nullptr;
  CombineInto combine{ws + e->o_gx1p, ws + e->o_gx1, false, recon_done && e->recon_bwd5_done};
  {
    const int i12[2] = {1, 2};
    const float* in12[2] = {ws + e->o_row, ws + e->o_col};
    const float* dz12[2] = {ws + e->o_gx2, ws + e->o_gx3c};
    float* di12[2] = {ws + e->o_gT, ws + e->o_gFc};
    if (e->pair_mode || !e->side_ok) {
      if ((rc = ae_backward(e, 2, i12, prm, grd, in12, dz12, di12, ws, 0, st, wgs, deferred ? &join_latent : nullptr, &combine))) return rc;
    } else {
      hipEvent_t evf = e->take_event();
      if (hipEventRecord(evf, st) != hipSuccess || hipStreamWaitEvent(e->wstream, evf, 0) != hipSuccess) {
        set_last_error("engine: stream fork failed");
        return LSHM_ERR_ARG;
      }
      if ((rc = ae_backward(e, 1, i12, prm, grd, in12, dz12, di12, ws, 0, st, nullptr))) return rc;
      if ((rc = ae_backward(e, 1, i12 + 1, prm, grd, in12 + 1, dz12 + 1, di12 + 1, ws, 1, e->wstream, nullptr))) return rc;
      hipEvent_t evj = e->take_event();
      if (hipEventRecord(evj, e->wstream) != hipSuccess || hipStreamWaitEvent(st, evj, 0) != hipSuccess) {
        set_last_error("engine: stream join failed");
        return LSHM_ERR_ARG;
      }
    }
  }
  bool early_bucket = false;
  if (e->comm && wgs && e->cstream && e->early_ok && !e->in_capture) {
    // netT / netF are done once their closing sums have run on the weight-gradient stream: their 1.9 MB go
    // now, on a stream of their own, while the 2-D autoencoder's backward runs
    hipEvent_t ev = e->take_event();
    if (hipEventRecord(ev, wgs) != hipSuccess || hipStreamWaitEvent(e->cstream, ev, 0) != hipSuccess) {
      set_last_error("engine: stream fork failed");
      return LSHM_ERR_ARG;
    }
    float* seg[1] = {grd + e->off1d};
    const size_t nseg[1] = {(size_t)(e->Moff - e->off1d)};
    if ((rc = comm_allreduce_segments(e->comm, seg, nseg, 1, nullptr, 0, e->cstream))) return rc;
    early_bucket = true;
    e->last_flags |= LSHM_ENGINE_USED_EARLY_BUCKET;
  }
  e->mark(lshm_engine::PH_BWD1D, st);
  if (!combine.done && (rc = combine_dx1(ws + e->o_gx1p, ws + e->o_gT, ws + e->o_gFc, ws + e->o_gx1, planes, c.P, st, e->bf))) return rc;
  {
    const int i0[1] = {0};
    const float* in0[1] = {x};
    const float* dz0[1] = {ws + e->o_gx1};
    float* di0[1] = {nullptr};
    if ((rc = ae_backward(e, 1, i0, prm, grd, in0, dz0, di0, ws, 2, st, wgs))) return rc;
  }
  e->pack_bwd_done = false;
  e->mark(lshm_engine::PH_BWD_MAIN_END, st);
  if (wgs) e->mark(lshm_engine::PH_BWD_SIDE_END, wgs);
  if (wgs) {  // the weight-gradient chain joins here, before anything consumes the gradients
    hipEvent_t evj = e->take_event();
    if (hipEventRecord(evj, wgs) != hipSuccess || hipStreamWaitEvent(st, evj, 0) != hipSuccess) {
      set_last_error("engine: stream join failed");
      return LSHM_ERR_ARG;
    }
  }
  if (e->comm) {
    // the rest of the arena (net, mod.M; everything if no early bucket went) and the ten loss doubles: one group
    float* seg[2];
    size_t nseg[2];
    int ns = 0;
    if (early_bucket) {
      seg[ns] = grd; nseg[ns++] = (size_t)e->off1d;
      seg[ns] = grd + e->Moff; nseg[ns++] = (size_t)(e->nparams - e->Moff);
    } else {
      seg[ns] = grd; nseg[ns++] = (size_t)e->nparams;
    }
    if ((rc = comm_allreduce_segments(e->comm, seg, nseg, ns, terms, 10, st))) return rc;
    if (early_bucket) {
      hipEvent_t evc = e->take_event();
      if (hipEventRecord(evc, e->cstream) != hipSuccess || hipStreamWaitEvent(st, evc, 0) != hipSuccess) {
        set_last_error("engine: stream join failed");
        return LSHM_ERR_ARG;
      }
    }
  }
  return LSHM_OK;
}

}  // namespace lshm

// Scope of one engine call: the engine's device is current, its operand precision applies to the launches
// of this thread, and a stream capture is refused where it is known to crash the runtime.
struct EngineCall {
  MatrixPrecisionScope prec;
  ScheduleScope sched_scope;
  int prev_dev = -1;
  bool switched = false;
  bool ok = true;  // false: the engine's device could not be made current -- nothing may be launched
  explicit EngineCall(const lshm_engine* e) : prec(e->cfg.precision != LSHM_PRECISION_F32), sched_scope(e->cfg.schedule) {
    if (e->device < 0) return;
    if (hipGetDevice(&prev_dev) != hipSuccess) { (void)hipGetLastError(); ok = false; return; }
    if (prev_dev != e->device) {
      switched = hipSetDevice(e->device) == hipSuccess;
      if (!switched) { (void)hipGetLastError(); ok = false; }
    }
  }
  ~EngineCall() {
    if (switched) (void)hipSetDevice(prev_dev);
  }
};
// The stream of an engine call must belong to the engine's device, and so must the arena / workspace / input
// pointers: a launch on the wrong device faults, or crosses the fabric silently.  Pointers are looked up once
// (hipPointerGetAttributes) and remembered; host memory -- pinned or not -- is refused.
static int device_fence(lshm_engine* e, hipStream_t st, std::initializer_list<const void*> ptrs) {
  if (e->device < 0) return LSHM_OK;
  if (st) {
    hipDevice_t sd = -1;
    if (hipStreamGetDevice(st, &sd) != hipSuccess) { (void)hipGetLastError(); }
    else if ((int)sd != e->device) {
      set_last_error("engine: the stream belongs to another device than the engine (lshm_engine_device)");
      return LSHM_ERR_ARG;
    }
  }
  for (const void* q : ptrs) {
    if (!q) continue;
    bool seen = false;
    for (int i = 0; i < e->nseen; ++i) seen = seen || e->seen_ptr[i] == q;
    if (seen) continue;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, q) != hipSuccess) {
      (void)hipGetLastError();
      set_last_error("engine: an argument is not a device pointer (host memory?)");
      return LSHM_ERR_ARG;
    }
    if (at.type != hipMemoryTypeDevice || at.device != e->device) {
      set_last_error(at.type != hipMemoryTypeDevice ? "engine: an argument points to host / managed memory, not device memory"
                                                    : "engine: an argument lives on another device than the engine");
      return LSHM_ERR_ARG;
    }
    e->seen_ptr[e->seen_total++ % 16] = q;
    if (e->nseen < 16) ++e->nseen;
  }
  return LSHM_OK;
}
// Nested stream forks crash hipStreamEndCapture (ROCm 7.2): fork mode (LSHM_FORK=1: netT and netF on two
// streams, each with its own weight-gradient fork) is refused under capture; the early all-reduce bucket of an
// attached communicator (a fork off the forked weight-gradient stream, the same topology) is switched off for
// the captured call -- its gradients then travel in the closing group on the capturing stream.
static int capture_fence(lshm_engine* e, hipStream_t st) {
  e->in_capture = false;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); return LSHM_OK; }
  if (cap == hipStreamCaptureStatusNone) return LSHM_OK;
  e->in_capture = true;
  if (e->pair_mode || !e->side_ok) return LSHM_OK;
  set_last_error("engine: stream capture is not supported in fork mode (LSHM_FORK=1); use the default paired launches");
  return LSHM_ERR_UNSUPPORTED;
}
#define ENGINE_ENTER(e, st, ...)                                                          \
  EngineCall engine_call_scope(e);                                                        \
  do {                                                                                    \
    if (!engine_call_scope.ok) {                                                          \
      set_last_error("engine: cannot make the engine's device current");                  \
      return LSHM_ERR_ARG;                                                                \
    }                                                                                     \
    const int rc_fence = capture_fence(e, st);                                            \
    if (rc_fence) return rc_fence;                                                        \
    if (!e->in_capture) {                                                                 \
      const int rc_dev = device_fence(e, st, {__VA_ARGS__});                              \
      if (rc_dev) return rc_dev;                                                          \
    }                                                                                     \
    e->last_flags = 0;                                                                    \
  } while (0)

extern "C" {

int lshm_engine_create(const lshm_step_config* cfg, lshm_engine** out) {
  if (!cfg || !out) { set_last_error("engine_create: null argument"); return LSHM_ERR_ARG; }
  if (cfg->P != 128) { set_last_error("engine: patch size must be 128 (fc1 is 768+16 wide, src/lofar_models.py:45-46)"); return LSHM_ERR_UNSUPPORTED; }
  if (cfg->B < 1 || cfg->C < 1 || cfg->L < 1 || cfg->Lt < 1 || cfg->K < 1 || cfg->H < 1 || cfg->H > 8 ||
      cfg->bpb < 1 || cfg->batch_size < 1) {
    set_last_error("engine_create: bad configuration");
    return LSHM_ERR_ARG;
  }
  if (cfg->precision != LSHM_PRECISION_F32 && cfg->precision != LSHM_PRECISION_BF16_OPERANDS &&
      cfg->precision != LSHM_PRECISION_BF16_STORAGE) {
    set_last_error("engine_create: unknown precision");
    return LSHM_ERR_ARG;
  }
  if (cfg->K > 64 || cfg->L + 2 * cfg->Lt > 512 || cfg->bpb > 32) {
    set_last_error("engine: supports K <= 64, L+2Lt <= 512, bpb <= 32");
    return LSHM_ERR_UNSUPPORTED;
  }
  lshm_engine* e = new lshm_engine();
  e->cfg = *cfg;
  if (e->cfg.world < 1) e->cfg.world = 1;
  e->D = cfg->L + 2 * cfg->Lt;
  e->hdim = 4 * cfg->H;
  e->nparams = 0;
  size_t cur = 0;
  const int B = cfg->B;
  const long img = (long)cfg->C * cfg->P * cfg->P;
  plan_ae(e, 0, "net", 2, cfg->L, 0, cur);
  plan_ae(e, 1, "netT", 1, cfg->Lt, cfg->L, cur);
  plan_ae(e, 2, "netF", 1, cfg->Lt, cfg->L + cfg->Lt, cur);
  e->off1d = e->ae[1].cw[0];
  e->bf = cfg->precision == LSHM_PRECISION_BF16_STORAGE;
  if (e->bf) {
    // bf16 tensors: the three reconstructions, the row / column residuals, every image-sized gradient, and (round 3) the
    // 8-channel half-resolution tensors -- conv0's output and tconv4's output of each autoencoder with their gradients.
    // A layer flag covers the tensor AND the gradient with respect to it.
    for (int a = 0; a < 3; ++a) {
      e->ae[a].dec[5].out_bf16 = 1;                                // x1 / x2 / x3 and their gradients
      e->ae[a].enc[0].out_bf16 = e->ae[a].enc[1].in_bf16 = 1;      // conv0's output (B, 8, half resolution)
      e->ae[a].dec[4].out_bf16 = e->ae[a].dec[5].in_bf16 = 1;      // tconv4's output
    }
    for (int a = 1; a < 3; ++a) e->ae[a].enc[0].in_bf16 = 1;       // the row / column residuals
  }
  e->Moff = add_param(e, "mod.M", {cfg->K, e->D});
  // split-K / reduction scratch: the largest consumer among wgrads, KHM and the recon partials
  size_t pf = khm_workspace_floats(B, e->D, cfg->K);
  {
    const size_t rp = recon_partials_floats(B * cfg->C, cfg->P);
    if (rp > pf) pf = rp;
    for (int a = 0; a < 3; ++a)
      for (int i = 0; i < 6; ++i) {
        size_t w = conv_workspace_floats(e->ae[a].enc[i]);
        if (w > pf) pf = w;
        w = conv_workspace_floats(e->ae[a].dec[i]);
        if (w > pf) pf = w;
      }
    // dense layers: largest of the fc1 / fc3 problems
    const int Lm = cfg->L > cfg->Lt ? cfg->L : cfg->Lt;
    size_t w = igemm_workspace_floats(B, 768, Lm + e->hdim, 1);
    if (w > pf) pf = w;
    w = igemm_workspace_floats(Lm, 768 + e->hdim, B, 1);
    if (w > pf) pf = w;
    w = igemm_workspace_floats(768, Lm + e->hdim, B, 1);
    if (w > pf) pf = w;
    w = igemm_workspace_floats(B, 768 + e->hdim, Lm, 1);
    if (w > pf) pf = w;
  }
  e->part_floats = pf;
  // ---- the forward prefix: the activations above, then ...
  e->o_scales = take(cur, 8);
  e->o_uvh = take(cur, (size_t)B * e->hdim);
  e->o_Mu = take(cur, (size_t)B * e->D);
  e->o_row = take(cur, (size_t)B * img);
  e->o_col = take(cur, (size_t)B * img);
  e->o_fpart = take(cur, 4 * pf);  // split-K scratch of the forward launches (two pair-sized slots)
  {
    const int ech[5] = {e->ae[0].enc[1].Cout, e->ae[0].enc[2].Cout, e->ae[0].enc[3].Cout, e->ae[0].enc[4].Cout, e->ae[0].enc[5].Cout};
    e->deep_bf16 = cfg->precision != LSHM_PRECISION_F32;  // bf16 operand precision: the chains stream bf16 weights
    e->deep2d = !(cfg->schedule & LSHM_SCHED_NO_DEEP2D) && deep2d_supported(cfg->L, e->hdim, cfg->rica, ech, e->ae[0].enc[3].Hin);
    if (e->deep2d) e->o_pack2d = take(cur, deep2d_packed_floats());
  }
  e->fwd_floats = cur;
  // ---- backward / loss side
  e->o_gMu = take(cur, (size_t)B * e->D);
  e->o_gx1p = take(cur, (size_t)B * img);
  e->o_gx2 = take(cur, (size_t)B * img);
  e->o_gx3c = take(cur, (size_t)B * img);
  e->o_gT = take(cur, (size_t)B * img);
  e->o_gFc = take(cur, (size_t)B * img);
  e->o_gx1 = take(cur, (size_t)B * img);
  const int Lmax = cfg->L > cfg->Lt ? cfg->L : cfg->Lt;
  // deferred-reduction scratch of one backward: the 2-D autoencoder alone or the two 1-D ones as a pair
  e->defer_floats = 0;
  for (int a = 0; a < 2; ++a) {
    const int G = a == 0 ? 1 : 2;
    const int La = a == 0 ? cfg->L : cfg->Lt;
    size_t need = 0;
    for (int i = 0; i < 6; ++i)
      need += conv_wgrad_defer_floats(e->ae[a].enc[i], G) + conv_wgrad_defer_floats(e->ae[a].dec[i], G);
    need += linear_wgrad_defer_floats(B, La + e->hdim, 768, G) + linear_wgrad_defer_floats(B, 768 + e->hdim, La, G) +
            2 * linear_wgrad_defer_floats(B, e->hdim, e->hdim, G) + 2 * linear_wgrad_defer_floats(B, La, La, G);
    if (need > e->defer_floats) e->defer_floats = need;
  }
  for (int ln = 0; ln < 3; ++ln) {
    lshm_engine::Lane& la = e->lane[ln];
    for (int i = 1; i < 6; ++i) {  // same sizes for the 2-D and the 1-D autoencoders
      la.o_gdec[i] = take(cur, (size_t)B * e->ae[0].dec[i].in_bs);
      la.o_genc[i] = take(cur, (size_t)B * e->ae[0].enc[i].in_bs);
    }
    la.o_gdec[0] = la.o_genc[0] = 0;
    la.o_defer = take(cur, e->defer_floats);
    la.o_dcat1 = take(cur, (size_t)B * (768 + e->hdim));
    la.o_dz1 = take(cur, (size_t)B * Lmax);
    la.o_dzmu = take(cur, (size_t)B * Lmax);
    la.o_dcat3 = take(cur, (size_t)B * (Lmax + e->hdim));
    la.o_dd0 = take(cur, (size_t)B * 768);
  }
  e->o_dMscratch = take(cur, (size_t)cfg->K * e->D);
  e->deep2d_bwd = e->deep2d && !(cfg->schedule & LSHM_SCHED_NO_DEEP2D_BWD);
  {
    const int ech[5] = {e->ae[1].enc[1].Cout, e->ae[1].enc[2].Cout, e->ae[1].enc[3].Cout, e->ae[1].enc[4].Cout, e->ae[1].enc[5].Cout};
    e->full1d = (cfg->schedule & LSHM_SCHED_TRY_FULL1D) && cfg->precision == LSHM_PRECISION_F32 &&
                chain1d_full_supported(cfg->Lt, e->hdim, cfg->rica, ech, e->ae[1].enc[2].Win);
  }
  e->wgrad_on_main = (cfg->tune & 0xffffu) ? (cfg->tune & 0xffffu) - 1 : 0u;  // (experimental placement word: lshm_step_config.tune, 0 = shipped)
  if (e->deep2d_bwd) e->o_pack2d_bwd = take(cur, deep2d_packed_floats());
  e->o_recon_part = take(cur, recon_partials_floats(B * cfg->C, cfg->P));
  e->o_recon_w5 = take(cur, recon_bwd5_workspace_floats());
  e->latent_ws_floats = khm_workspace_floats(B, e->D, cfg->K);
  e->o_latent_ws = take(cur, e->latent_ws_floats);
  e->latent_event = nullptr;
  e->recon_ready = false;
  for (int ln = 0; ln < 3; ++ln) {
    e->lane[ln].o_part = take(cur, pf);
    e->lane[ln].o_wpart = take(cur, pf);
  }
  // side stream + events, fork mode only (host objects; absent on a machine without a HIP device)
  e->side_ok = false;
  e->next_event = 0;
  e->pair_mode = !(cfg->schedule & LSHM_SCHED_FORK);
  e->device = -1;
  {
    int ndev = 0;
    e->side_wgrad = !(cfg->schedule & LSHM_SCHED_WGRAD_INLINE);
    if (hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0 && hipGetDevice(&e->device) != hipSuccess) e->device = -1;
    if (e->device >= 0 && (!e->pair_mode || e->side_wgrad)) {
      // the weight-gradient stream sits BELOW the caller's stream in the dispatcher's order: when both have a kernel ready the
      // data-gradient chain (every link of which something waits for) gets the CUs first (2.145 against 2.155 ms per iteration;
      // above it: 2.153).  The no-grad forward's stream sits ABOVE: the reconstruction pass and with it the backward wait for
      // that chain, the closure forward beside it has slack (2.127 against 2.141 ms; below: 2.139; profiles/r03/README.md)
      int plo = 0, phi = 0;
      (void)hipDeviceGetStreamPriorityRange(&plo, &phi);
      bool ok = hipStreamCreateWithPriority(&e->wstream, hipStreamNonBlocking, plo) == hipSuccess;
      ok = ok && hipStreamCreateWithFlags(&e->lstream, hipStreamNonBlocking) == hipSuccess;
      ok = ok && hipStreamCreateWithPriority(&e->fstream, hipStreamNonBlocking, phi) == hipSuccess;
      e->events.resize(256);
      for (size_t i = 0; i < e->events.size() && ok; ++i)
        ok = ok && hipEventCreateWithFlags(&e->events[i], hipEventDisableTiming) == hipSuccess;
      e->side_ok = ok;
      if (ok && (cfg->schedule & LSHM_SCHED_PHASE_EVENTS)) {
        e->phase.resize(lshm_engine::PH_COUNT);
        for (auto& ev : e->phase) ok = ok && hipEventCreate(&ev) == hipSuccess;
        if (!ok) e->phase.clear();
      }
    }
    (void)hipGetLastError();
  }
  const size_t ngroups = (size_t)(B + cfg->bpb - 1) / cfg->bpb;
  e->o_scal = take(cur, 2 * (16 + ngroups + 3 * LOGCOSH3_BLOCKS));
  e->alt_base = cur;  // second forward prefix (see lshm_engine: fwd_floats)
  e->ws_floats = cur + e->fwd_floats;
  *out = e;
  return LSHM_OK;
}

void lshm_engine_destroy(lshm_engine* e) {
  if (!e) return;
  if (e->side_ok) {
    EngineCall scope(e);
    if (e->cstream) (void)hipStreamDestroy(e->cstream);
    (void)hipStreamDestroy(e->wstream);
    if (e->lstream) (void)hipStreamDestroy(e->lstream);
    if (e->fstream) (void)hipStreamDestroy(e->fstream);
    for (hipEvent_t ev : e->events) (void)hipEventDestroy(ev);
  }
  delete e;
}

long lshm_engine_param_count(const lshm_engine* e) { return e ? e->nparams : 0; }

int lshm_engine_param_lookup(const lshm_engine* e, const char* name, long* offset, long* numel) {
  if (!e || !name) { set_last_error("param_lookup: null argument"); return LSHM_ERR_ARG; }
  for (const ParamInfo& p : e->params)
    if (p.name == name) {
      if (offset) *offset = p.offset;
      if (numel) *numel = p.numel;
      return LSHM_OK;
    }
  set_last_error("param_lookup: unknown parameter name");
  return LSHM_ERR_ARG;
}

int lshm_engine_param_name(const lshm_engine* e, int index, char* buf, int buflen, long* offset,
                           long* numel, int* ndim, long* shape) {
  if (!e || index < 0 || index >= (int)e->params.size()) return LSHM_ERR_ARG;
  const ParamInfo& p = e->params[index];
  if (buf && buflen > 0) {
    strncpy(buf, p.name.c_str(), buflen - 1);
    buf[buflen - 1] = 0;
  }
  if (offset) *offset = p.offset;
  if (numel) *numel = p.numel;
  if (ndim) *ndim = p.ndim;
  if (shape) for (int i = 0; i < 4; ++i) shape[i] = p.shape[i];
  return LSHM_OK;
}

size_t lshm_engine_workspace_floats(const lshm_engine* e) { return e ? e->ws_floats : 0; }

int lshm_engine_set_comm(lshm_engine* e, lshm_comm* comm) {
  if (!e) { set_last_error("engine_set_comm: null engine"); return LSHM_ERR_ARG; }
  if (comm && comm_world(comm) != e->cfg.world) {
    set_last_error("engine_set_comm: the communicator's world size differs from lshm_step_config.world");
    return LSHM_ERR_ARG;
  }
  EngineCall scope(e);
  if (comm && !e->cstream && e->side_ok &&
      hipStreamCreateWithFlags(&e->cstream, hipStreamNonBlocking) != hipSuccess) {
    (void)hipGetLastError();
    e->cstream = nullptr;  // no early bucket then: one group after the last weight gradient
  }
  e->comm = comm;
  return LSHM_OK;
}

int lshm_engine_device(const lshm_engine* e) { return e ? e->device : -1; }
unsigned lshm_engine_set_schedule(lshm_engine* e, unsigned schedule) {
  if (!e) return 0u;
  const unsigned fixed = LSHM_SCHED_NO_DEEP2D | LSHM_SCHED_NO_DEEP2D_BWD | LSHM_SCHED_TRY_FULL1D | LSHM_SCHED_WGRAD_INLINE | LSHM_SCHED_FORK |
                         LSHM_SCHED_PHASE_EVENTS;
  e->cfg.schedule = (e->cfg.schedule & fixed) | (schedule & ~fixed);
  e->plan.steps.clear();  // the cached forward plan was built under the old word
  return e->cfg.schedule;
}
int lshm_engine_phase_times(const lshm_engine* e, float* ms, int n) {
  if (!e || !ms || n < 1) { set_last_error("engine_phase_times: bad argument"); return LSHM_ERR_ARG; }
  if (e->phase.empty()) { set_last_error("engine_phase_times: the engine was created without LSHM_SCHED_PHASE_EVENTS"); return LSHM_ERR_UNSUPPORTED; }
  EngineCall scope(e);  // the events live on the engine's device, whatever device is current in the caller
  if (!scope.ok) { set_last_error("engine: cannot make the engine's device current"); return LSHM_ERR_ARG; }
  if (hipDeviceSynchronize() != hipSuccess) { set_last_error("engine_phase_times: device synchronisation failed"); return LSHM_ERR_ARG; }
  for (int i = 0; i < n; ++i) {
    ms[i] = -1.f;  // an event that was never recorded (a phase this schedule does not have)
    if (i < lshm_engine::PH_COUNT && hipEventElapsedTime(&ms[i], e->phase[lshm_engine::PH_CLOSURE], e->phase[i]) != hipSuccess) {
      ms[i] = -1.f;
      (void)hipGetLastError();
    }
  }
  return LSHM_OK;
}
unsigned lshm_engine_last_flags(const lshm_engine* e) { return e ? e->last_flags : 0u; }
int lshm_engine_comm_early_bucket(const lshm_engine* e) {
  return (e && e->comm && e->cstream && e->side_ok && e->side_wgrad && e->pair_mode && e->early_ok) ? 1 : 0;
}
int lshm_engine_set_early_bucket(lshm_engine* e, int on) {
  if (!e) { set_last_error("engine_set_early_bucket: null engine"); return LSHM_ERR_ARG; }
  e->early_ok = on != 0;
  return LSHM_OK;
}

#define ENGINE_CHECK(cond, msg)     \
  do {                              \
    if (!(cond)) {                  \
      set_last_error(msg);          \
      return LSHM_ERR_ARG;          \
    }                               \
  } while (0)

int lshm_engine_forward_backward_ex(lshm_engine* e, const float* params, float* grads, const float* x,
                                    const float* uv, const float* y1, const float* y2, const float* y3,
                                    double* terms, float* ws, size_t wsf, unsigned flags, lshm_stream_t s) {
  ENGINE_CHECK(e && params && grads && x && uv && y1 && y2 && y3 && terms && ws, "engine_forward_backward: null pointer");
  if (wsf < e->ws_floats) { set_last_error("engine: workspace too small"); return LSHM_ERR_WORKSPACE; }
  hipStream_t st = reinterpret_cast<hipStream_t>(s);
  ENGINE_ENTER(e, st, params, grads, x, uv, y1, y2, y3, ws, terms);
  e->next_event = 0;
  // the forward below is recomputed either way; only the reconstruction pass can be the one the preceding
  // lshm_engine_multiplier_update_next already made with the same inputs
  const bool recon_done = (flags & LSHM_STEP_RECON_READY) && e->recon_ready;
  int rc;
  // (after lshm_engine_multiplier_update_next_ex(LSHM_NEXT_CONCURRENT_FORWARD) the seven sums of that pass exist only as
  //  per-block partials; the forward below forgets them, so they are closed first)
  if (recon_done && (rc = pending_sum7(e, ws, st))) return rc;
  // ... and with it done, nothing reads the reconstructions of netT / netF (the workspace keeps the identical
  // ones of the preceding no-grad forward)
  rc = forward_with_latent_losses(e, params, grads, x, uv, ws, st, recon_done);
  if (rc) return rc;
  return losses_and_backward(e, params, grads, x, y1, y2, y3, terms, ws, st, recon_done);
}

int lshm_engine_forward_backward(lshm_engine* e, const float* params, float* grads, const float* x,
                                 const float* uv, const float* y1, const float* y2, const float* y3,
                                 double* terms, float* ws, size_t wsf, lshm_stream_t s) {
  return lshm_engine_forward_backward_ex(e, params, grads, x, uv, y1, y2, y3, terms, ws, wsf, 0u, s);
}

int lshm_engine_backward_saved(lshm_engine* e, const float* params, float* grads, const float* x, const float* y1,
                               const float* y2, const float* y3, double* terms, float* ws, size_t wsf,
                               lshm_stream_t s) {
  ENGINE_CHECK(e && params && grads && x && y1 && y2 && y3 && terms && ws, "engine_backward_saved: null pointer");
  if (wsf < e->ws_floats) { set_last_error("engine: workspace too small"); return LSHM_ERR_WORKSPACE; }
  hipStream_t st = reinterpret_cast<hipStream_t>(s);
  ENGINE_ENTER(e, st, params, grads, x, y1, y2, y3, ws, terms);
  e->next_event = 0;
  const bool recon_done = e->recon_ready;
  e->recon_ready = false;
  e->mark(lshm_engine::PH_CLOSURE, st);
  int rc;
  if (e->latent_early) {
    // the terms were started beside the forwards (lshm_engine_multiplier_update_next_ex): what is left for their stream are
    // the seven sums of the reconstruction pass and the centroid gradient's move from scratch into the arena
    e->latent_early = false;
    e->latent_event = nullptr;
    hipEvent_t ev = e->take_event();
    if (hipEventRecord(ev, st) != hipSuccess || hipStreamWaitEvent(e->lstream, ev, 0) != hipSuccess) {
      set_last_error("engine: stream fork failed");
      return LSHM_ERR_ARG;
    }
    if ((rc = pending_sum7(e, ws, e->lstream))) return rc;
    if (hipMemcpyAsync(grads + e->Moff, ws + e->o_dMscratch, sizeof(float) * (size_t)e->cfg.K * e->D, hipMemcpyDeviceToDevice, e->lstream) != hipSuccess) {
      set_last_error("engine: centroid-gradient copy failed");
      return LSHM_ERR_ARG;
    }
    if ((rc = pack_backward_weights(e, params, ws, e->lstream))) return rc;
    e->latent_event = e->take_event();
    if (hipEventRecord(e->latent_event, e->lstream) != hipSuccess) {
      set_last_error("engine: event record failed");
      return LSHM_ERR_ARG;
    }
  } else if ((rc = start_latent_losses(e, params, grads, ws, st))) {
    return rc;
  }
  rc = losses_and_backward(e, params, grads, x, y1, y2, y3, terms, ws, st, recon_done);
  e->mark(lshm_engine::PH_CLOSURE_END, st);
  return rc;
}

int lshm_engine_multiplier_update_next_ex(lshm_engine* e, const float* params, const float* x, const float* uv,
                                          float* y1, float* y2, float* y3, float* ws, size_t wsf, unsigned flags,
                                          lshm_stream_t s) {
  ENGINE_CHECK(e && params && x && uv && y1 && y2 && y3 && ws, "engine_multiplier_update_next: null pointer");
  if (wsf < e->ws_floats) { set_last_error("engine: workspace too small"); return LSHM_ERR_WORKSPACE; }
  hipStream_t st = reinterpret_cast<hipStream_t>(s);
  ENGINE_ENTER(e, st, params, x, uv, y1, y2, y3, ws);
  e->next_event = 0;
  e->recon_ready = false;
  const lshm_step_config& c = e->cfg;
  const double world = c.world > 0 ? c.world : 1;
  // The no-grad forward that closes this iteration (src/kharmonic_lofar.py:187-196) and the closure forward that
  // opens the next one (:135-150) both depend on the updated parameters and on nothing else: with
  // LSHM_NEXT_CONCURRENT_FORWARD they are two chains on two streams, each with its own copy of the forward
  // buffers.  The reconstruction pass (multiplier update + the next closure's terms) follows the no-grad forward
  // on its stream; the caller's stream resumes when both are done.
  const bool concurrent = (flags & LSHM_NEXT_CONCURRENT_FORWARD) && e->side_ok && e->fstream && !e->in_capture &&
                          e->pair_mode;
  hipStream_t fst = concurrent ? e->fstream : st;
  float* fws = concurrent ? ws + e->alt_base : ws;
  e->mark(lshm_engine::PH_UPDATE, st);
  if (concurrent) {
    hipEvent_t ev = e->take_event();
    if (hipEventRecord(ev, st) != hipSuccess || hipStreamWaitEvent(fst, ev, 0) != hipSuccess) {
      set_last_error("engine: stream fork failed");
      return LSHM_ERR_ARG;
    }
  }
  // concurrent: the no-grad forward on (fws, fst) and the next closure's forward (activations saved in the primary
  // buffers; nothing reads its reconstructions of netT / netF -- the pass below takes them from the no-grad
  // forward -- so their last decoder layer is not run), enqueued in lock step
  // concurrent, fp32 storage: the no-grad forward stops in front of the last layer of netT / netF and the pass forms their
  // reconstructions from that layer's input (recon_kernel<.., FROMA>): one launch, 0.13 GB of writes and 0.2 GB of reads less
  const AEPlan& aT = e->ae[1];
  const AEPlan& aF = e->ae[2];
  const bool bf_ok = !e->bf || (aT.dec[5].in_bf16 && aT.dec[5].out_bf16 && aF.dec[5].in_bf16 && aF.dec[5].out_bf16);
  const bool from_a = concurrent && bf_ok && aT.dec[5].in_bs == aF.dec[5].in_bs &&
                      recon_from_a_supported(c.C, c.P, aT.dec[5].Cin, aT.dec[5].Cout, aT.dec[5].Win);
  // The latent-space terms of the NEXT closure (K-harmonic, similarity, augmentation, log-cosh: src/kharmonic_lofar.py:160-172)
  // need only the codes of the closure forward and the centroids: they start on their own stream as soon as that forward
  // has its three codes, beside the rest of the two forwards, instead of at the head of the backward -- where their ~150 us
  // chain ended ~40 us after the 1-D decoders' data gradients needed its result.  The centroid gradient waits in scratch;
  // lshm_engine_backward_saved moves it into the arena.  Same kernels on the same operands: the same bits.
  const bool early_latent = concurrent && e->side_wgrad && e->lstream && !(c.schedule & LSHM_SCHED_NO_EARLY_LATENT);
  const std::function<int()> latent_hook = [&]() -> int {
    hipEvent_t ev = e->take_event();
    if (hipEventRecord(ev, st) != hipSuccess || hipStreamWaitEvent(e->lstream, ev, 0) != hipSuccess) {
      set_last_error("engine: stream fork failed");
      return LSHM_ERR_ARG;
    }
    e->sim_started = false;
    return latent_losses(e, params, nullptr, ws, e->lstream);
  };
  int rc = concurrent ? two_forwards(e, params, x, uv, fws, fst, ws, st, true, from_a, early_latent ? &latent_hook : nullptr)
                      : three_forward(e, params, x, uv, fws, fst);
  if (rc) return rc;
  e->latent_early = early_latent;
  if (concurrent) e->mark(lshm_engine::PH_FWD_CLOSURE_END, st);
  e->mark(lshm_engine::PH_FWD_NOGRAD_END, fst);
  // concurrent: the seven sums of the pass (a 13 us launch the caller's stream would wait for) move to the latent-space
  // stream of the next closure: nothing needs them before the loss terms are assembled there
  // ... and, fp32 storage, the backward of that layer (both networks) inside the same pass: two of the three gradient images are
  // never written (recon_bwd5.hip); the backward finds the layer's data gradients in the lanes' buffers
  const bool with_bwd5 = from_a && dec5_bf_ok(e) && e->side_wgrad && aT.dec[5].out_bs == aF.dec[5].out_bs &&
                         recon_bwd5_supported(c.C, c.P, aT.dec[5].Cin, aT.dec[5].Cout, aT.dec[5].Win);
  e->recon_bwd5_done = with_bwd5;
  if (with_bwd5)
    rc = recon_bwd5(x, fws + e->ae[0].out, fws + aT.dact[4], fws + aF.dact[4], aT.dec[5].in_bs, params + aT.tw[5], params + aT.tb[5],
                    params + aF.tw[5], params + aF.tb[5], y1, y2, y3, c.rho, c.B, ws + e->o_gx1p, ws + e->lane[0].o_gdec[5],
                    ws + e->lane[1].o_gdec[5], aT.dec[5].in_bs, ws + e->o_recon_part, ws + e->o_recon_w5, recon_bwd5_workspace_floats(), fst,
                    (float)(1.0 / world), e->bf);
  else if (from_a)
    rc = multiplier_update_recon_from_a(x, fws + e->ae[0].out, fws + aT.dact[4], fws + aF.dact[4], aT.dec[5].in_bs, params + aT.tw[5],
                                        params + aT.tb[5], params + aF.tw[5], params + aF.tb[5], c.C, y1, y2, y3, c.rho, c.B * c.C,
                                        c.P, ws + e->o_gx1p, ws + e->o_gx2, ws + e->o_gx3c, ws + e->o_recon_part, fst,
                                        (float)(1.0 / world), e->bf);
  else
    rc = multiplier_update_recon(x, fws + e->ae[0].out, fws + e->ae[1].out, fws + e->ae[2].out, y1, y2, y3, c.rho,
                                 c.B * c.C, c.P, concurrent ? nullptr : reinterpret_cast<double*>(ws + e->o_scal), ws + e->o_gx1p,
                                 ws + e->o_gx2, ws + e->o_gx3c, ws + e->o_recon_part, fst, (float)(1.0 / world), e->bf);
  if (rc) return rc;
  e->mark(lshm_engine::PH_RECON_END, fst);
  e->sum7_pending = concurrent;
  if (concurrent) {
    hipEvent_t evj = e->take_event();
    if (hipEventRecord(evj, fst) != hipSuccess || hipStreamWaitEvent(st, evj, 0) != hipSuccess) {
      set_last_error("engine: stream join failed");
      return LSHM_ERR_ARG;
    }
    e->last_flags |= LSHM_ENGINE_USED_CONCURRENT_FORWARD;
  }
  e->mark(lshm_engine::PH_UPDATE_END, st);
  e->recon_ready = true;
  return LSHM_OK;
}

int lshm_engine_multiplier_update_next(lshm_engine* e, const float* params, const float* x, const float* uv,
                                       float* y1, float* y2, float* y3, float* ws, size_t wsf, lshm_stream_t s) {
  return lshm_engine_multiplier_update_next_ex(e, params, x, uv, y1, y2, y3, ws, wsf, 0u, s);
}

int lshm_engine_forward_loss(lshm_engine* e, const float* params, const float* x, const float* uv,
                             const float* y1, const float* y2, const float* y3, double* terms,
                             float* ws, size_t wsf, lshm_stream_t s) {
  ENGINE_CHECK(e && params && x && uv && y1 && y2 && y3 && terms && ws, "engine_forward_loss: null pointer");
  if (wsf < e->ws_floats) { set_last_error("engine: workspace too small"); return LSHM_ERR_WORKSPACE; }
  hipStream_t st = reinterpret_cast<hipStream_t>(s);
  ENGINE_ENTER(e, st, params, x, uv, y1, y2, y3, ws, terms);
  e->next_event = 0;
  int rc = forward_with_latent_losses(e, params, nullptr, x, uv, ws, st);
  if (rc) return rc;
  return losses_and_backward(e, params, nullptr, x, y1, y2, y3, terms, ws, st);
}

int lshm_engine_multiplier_update(lshm_engine* e, const float* params, const float* x, const float* uv,
                                  float* y1, float* y2, float* y3, float* ws, size_t wsf,
                                  lshm_stream_t s) {
  ENGINE_CHECK(e && params && x && uv && y1 && y2 && y3 && ws, "engine_multiplier_update: null pointer");
  if (wsf < e->ws_floats) { set_last_error("engine: workspace too small"); return LSHM_ERR_WORKSPACE; }
  hipStream_t st = reinterpret_cast<hipStream_t>(s);
  ENGINE_ENTER(e, st, params, x, uv, y1, y2, y3, ws);
  e->next_event = 0;
  int rc = three_forward(e, params, x, uv, ws, st);
  if (rc) return rc;
  const lshm_step_config& c = e->cfg;
  return multiplier_update(x, ws + e->ae[0].out, ws + e->ae[1].out, ws + e->ae[2].out, y1, y2, y3, c.rho,
                           c.B * c.C, c.P, st, e->bf);
}

int lshm_engine_encode(lshm_engine* e, const float* params, const float* x, const float* uv, float* Mu,
                       float* x1, float* x2, float* x3, float* ws, size_t wsf, lshm_stream_t s) {
  ENGINE_CHECK(e && params && x && uv && ws, "engine_encode: null pointer");
  if (wsf < e->ws_floats) { set_last_error("engine: workspace too small"); return LSHM_ERR_WORKSPACE; }
  hipStream_t st = reinterpret_cast<hipStream_t>(s);
  ENGINE_ENTER(e, st, params, x, uv, ws);
  e->next_event = 0;
  int rc = three_forward(e, params, x, uv, ws, st);
  if (rc) return rc;
  const lshm_step_config& c = e->cfg;
  const size_t img = (size_t)c.B * c.C * c.P * c.P;
  if (Mu && (rc = hipMemcpyAsync(Mu, ws + e->o_Mu, sizeof(float) * c.B * e->D, hipMemcpyDeviceToDevice, st))) return rc;
  if (e->bf) {  // the caller gets fp32 whatever the storage type
    if (x1 && (rc = widen_bf16(ws + e->ae[0].out, x1, (long)img, st))) return rc;
    if (x2 && (rc = widen_bf16(ws + e->ae[1].out, x2, (long)img, st))) return rc;
  } else {
    if (x1 && (rc = hipMemcpyAsync(x1, ws + e->ae[0].out, sizeof(float) * img, hipMemcpyDeviceToDevice, st))) return rc;
    if (x2 && (rc = hipMemcpyAsync(x2, ws + e->ae[1].out, sizeof(float) * img, hipMemcpyDeviceToDevice, st))) return rc;
  }
  if (x3 && (rc = plane_transpose(ws + e->ae[2].out, x3, c.B * c.C, c.P, st, e->bf))) return rc;
  return LSHM_OK;
}

}  // extern "C"
