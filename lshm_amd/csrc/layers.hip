// Layer-level dispatch: maps the four convolution flavours of the autoencoders
// (src/lofar_models.py:31-57, 115-142) and the dense layers onto the implicit
// GEMM problems of igemm.hip and the direct kernels of conv_direct.hip.  A
// transposed convolution's forward is the data-gradient problem of the matching
// strided convolution and vice versa, so three GEMM problems per dimensionality
// cover forward, dgrad and wgrad of both.  Every function accepts a second
// pointer bundle: two independent problems of identical shape (the row- and
// column-vectorised 1-D autoencoders) then share each launch.
#include <stdlib.h>

#include "kernels.h"

namespace lshm {

void conv_out_dims(const ConvLayer& L, int& Ho, int& Wo) {
  switch (L.kind) {
    case 0: Ho = L.Hin / 2; Wo = L.Win / 2; break;
    case 1: Ho = L.Hin * 2; Wo = L.Win * 2; break;
    case 2: Ho = 1; Wo = (L.Win - 2) / 4 + 1; break;
    default: Ho = 1; Wo = L.Win * 4; break;
  }
}

static bool transposed(const ConvLayer& L) { return L.kind == 1 || L.kind == 3; }

// GEMM shapes of the three problems of a layer
struct LayerGemm { int M, N, K, Z; };
static LayerGemm fwd_gemm(const ConvLayer& L) {
  int Ho, Wo;
  conv_out_dims(L, Ho, Wo);
  switch (L.kind) {
    case 0: return {L.B * Ho * Wo, L.Cout, L.Cin * 16, 1};
    case 1: return {L.B * L.Hin * L.Win, L.Cout, L.Cin * 4, 4};
    case 2: return {L.B * Wo, L.Cout, L.Cin * 4, 1};
    default: return {L.B * L.Win, L.Cout * 4, L.Cin, 1};
  }
}
static LayerGemm dgrad_gemm(const ConvLayer& L) {
  int Ho, Wo;
  conv_out_dims(L, Ho, Wo);
  switch (L.kind) {
    case 0: return {L.B * Ho * Wo, L.Cin, L.Cout * 4, 4};
    case 1: return {L.B * L.Hin * L.Win, L.Cin, L.Cout * 16, 1};
    case 2: return {L.B * Wo, L.Cin * 4, L.Cout, 1};
    default: return {L.B * L.Win, L.Cin, L.Cout * 4, 1};
  }
}
static LayerGemm wgrad_gemm(const ConvLayer& L) {
  int Ho, Wo;
  conv_out_dims(L, Ho, Wo);
  const int taps = (L.kind < 2) ? 16 : 4;
  const long small_sp = transposed(L) ? (long)L.Hin * L.Win : (long)Ho * Wo;
  return {transposed(L) ? L.Cin : L.Cout, (transposed(L) ? L.Cout : L.Cin) * taps,
          (int)((long)L.B * small_sp), 1};
}

#define BIAS_WS_FLOATS (2 * 128 * 256)
static int bias_slices(const ConvLayer& L) {
  int s = 512 / (L.Cout > 0 ? L.Cout : 1);
  if (s < 1) s = 1;
  if (s > 128) s = 128;
  if (s > L.B) s = L.B;
  return s;
}

size_t conv_workspace_floats(const ConvLayer& L) {
  const LayerGemm f = fwd_gemm(L), d = dgrad_gemm(L), w = wgrad_gemm(L);
  size_t a = igemm_workspace_floats(f.M, f.N, f.K, f.Z);
  const size_t b = igemm_workspace_floats(d.M, d.N, d.K, d.Z);
  const size_t c = igemm_workspace_floats(w.M, w.N, w.K, w.Z);
  if (b > a) a = b;
  if (c > a) a = c;
  if (L.kind < 2) {
    const size_t d2 = conv2d_wgrad_direct_workspace_floats(w.M, w.N / 16);
    if (d2 > a) a = d2;
  } else {
    const size_t d1 = conv1d_wgrad_direct_workspace_floats(w.M, w.N / 4);
    if (d1 > a) a = d1;
    const size_t d3 = conv1d_wgrad_mid_workspace_floats(w.M, w.N / 4);
    if (d3 > a) a = d3;
  }
  return a + BIAS_WS_FLOATS + 16;
}

// scratch a deferred weight-gradient call of this layer takes from its GradJobs list (G problems per launch)
size_t conv_wgrad_defer_floats(const ConvLayer& L, int G) {
  return (conv_workspace_floats(L) + 64) * G + (size_t)G * (256 * (size_t)L.Cout + 64);  // each take() rounds up to 64 floats
}
size_t linear_wgrad_defer_floats(int B, int K, int N, int G) {
  return (igemm_workspace_floats(N, K, B, 1) + 16 + 64) * G;
}

// ---- problem descriptors from pointer bundles ---------------------------------------------
static Conv2dFwdParams p_conv2d_fwd(const ConvLayer& L, const ConvFwdIO& io, int act, int Ho, int Wo) {
  return Conv2dFwdParams{io.x, io.w, io.b, io.y, nullptr, L.B, L.Cin, L.Hin, L.Win, L.Cout, Ho, Wo,
                         L.in_bs, L.out_bs, act, L.B * Ho * Wo, L.Cout, L.Cin * 16, {}};
}
static Conv2dDgradParams p_tconv2d_fwd(const ConvLayer& L, const ConvFwdIO& io, int act) {
  return Conv2dDgradParams{io.x, io.w, io.b, io.y, nullptr, L.B, L.Cin, L.Hin, L.Win, L.Cout,
                           L.in_bs, L.out_bs, act, L.B * L.Hin * L.Win, L.Cout, L.Cin * 4, {}};
}
static Conv1dFwdParams p_conv1d_fwd(const ConvLayer& L, const ConvFwdIO& io, int act, int Wo) {
  return Conv1dFwdParams{io.x, io.w, io.b, io.y, nullptr, L.B, L.Cin, L.Win, L.Cout, Wo, 1,
                         L.in_bs, L.out_bs, act, L.B * Wo, L.Cout, L.Cin * 4, {}};
}
static Conv1dDgradParams p_tconv1d_fwd(const ConvLayer& L, const ConvFwdIO& io, int act, int Wo) {
  return Conv1dDgradParams{io.x, io.w, io.b, io.y, nullptr, L.B, L.Cin, L.Win, L.Cout, Wo, 0,
                           L.in_bs, L.out_bs, act, L.B * L.Win, L.Cout * 4, L.Cin, {}};
}

static int bf16_unsupported() {
  set_last_error("conv layer: bf16 storage is only available on the outermost layers' own kernels");
  return LSHM_ERR_UNSUPPORTED;
}

int conv_layer_fwd(const ConvLayer& L, const ConvFwdIO& io, int act, float* ws, size_t wsf, hipStream_t st,
                   const ConvFwdIO* io2) {
  int Ho, Wo;
  conv_out_dims(L, Ho, Wo);
  switch (L.kind) {
    case 0: {
      if (!io2 && conv2d_direct_supported(L.Cin, L.Cout, Ho, Wo))  // (the launcher refuses storage combinations it has no kernel for)
        return conv2d_direct(io.x, L.in_bs, io.w, io.b, io.y, L.out_bs, nullptr, L.B, L.Cin, L.Cout, Ho, Wo, act, st,
                             L.in_bf16, L.out_bf16);
      if (L.in_bf16 || L.out_bf16) return bf16_unsupported();
      const Conv2dFwdParams p = p_conv2d_fwd(L, io, act, Ho, Wo);
      if (!io2) return conv2d_fwd(p, ws, wsf, st);
      const Conv2dFwdParams q = p_conv2d_fwd(L, *io2, act, Ho, Wo);
      return conv2d_fwd(p, ws, wsf, st, &q);
    }
    case 1: {
      if (!io2 && tconv2d_direct_supported(L.Cin, L.Cout, L.Hin, L.Win))
        return tconv2d_direct(io.x, L.in_bs, io.w, io.b, io.y, L.out_bs, nullptr, L.B, L.Cin, L.Cout, L.Hin, L.Win,
                              act, st, L.out_bf16, L.in_bf16);
      if (L.in_bf16 || L.out_bf16) return bf16_unsupported();
      const Conv2dDgradParams p = p_tconv2d_fwd(L, io, act);
      if (!io2) return conv2d_dgrad(p, ws, wsf, st);
      const Conv2dDgradParams q = p_tconv2d_fwd(L, *io2, act);
      return conv2d_dgrad(p, ws, wsf, st, &q);
    }
    case 2: {
      Conv1dFwdParams p = p_conv1d_fwd(L, io, act, Wo);
      p.x_bf16 = L.in_bf16;
      p.y_bf16 = L.out_bf16;
      if ((L.in_bf16 || L.out_bf16) && !conv1d_stream_supported(p)) return bf16_unsupported();
      if (!io2) return conv1d_fwd(p, ws, wsf, st);
      Conv1dFwdParams q = p_conv1d_fwd(L, *io2, act, Wo);
      q.x_bf16 = L.in_bf16;
      q.y_bf16 = L.out_bf16;
      return conv1d_fwd(p, ws, wsf, st, &q);
    }
    default: {
      Conv1dDgradParams p = p_tconv1d_fwd(L, io, act, Wo);
      p.big_bf16 = L.out_bf16;
      p.s_bf16 = L.in_bf16;
      if ((L.in_bf16 || L.out_bf16) && !tconv1d_stream_supported(p)) return bf16_unsupported();
      if (!io2) return conv1d_dgrad(p, ws, wsf, st);
      Conv1dDgradParams q = p_tconv1d_fwd(L, *io2, act, Wo);
      q.big_bf16 = L.out_bf16;
      q.s_bf16 = L.in_bf16;
      return conv1d_dgrad(p, ws, wsf, st, &q);
    }
  }
}

static Conv2dDgradParams p_conv2d_dgrad(const ConvLayer& L, const ConvDgradIO& io, int Ho, int Wo) {
  return Conv2dDgradParams{io.dz, io.w, nullptr, io.dx, io.dact_in, L.B, L.Cout, Ho, Wo, L.Cin,
                           L.out_bs, L.in_bs, 0, L.B * Ho * Wo, L.Cin, L.Cout * 4, {}};
}
static Conv2dFwdParams p_tconv2d_dgrad(const ConvLayer& L, const ConvDgradIO& io, int Ho, int Wo) {
  return Conv2dFwdParams{io.dz, io.w, nullptr, io.dx, io.dact_in, L.B, L.Cout, Ho, Wo, L.Cin, L.Hin, L.Win,
                         L.out_bs, L.in_bs, 0, L.B * L.Hin * L.Win, L.Cin, L.Cout * 16, {}};
}
static Conv1dDgradParams p_conv1d_dgrad(const ConvLayer& L, const ConvDgradIO& io, int Wo) {
  return Conv1dDgradParams{io.dz, io.w, nullptr, io.dx, io.dact_in, L.B, L.Cout, Wo, L.Cin, L.Win, 1,
                           L.out_bs, L.in_bs, 0, L.B * Wo, L.Cin * 4, L.Cout, {}};
}
static Conv1dFwdParams p_tconv1d_dgrad(const ConvLayer& L, const ConvDgradIO& io, int Wo) {
  return Conv1dFwdParams{io.dz, io.w, nullptr, io.dx, io.dact_in, L.B, L.Cout, Wo, L.Cin, L.Win, 0,
                         L.out_bs, L.in_bs, 0, L.B * L.Win, L.Cin, L.Cout * 4, {}};
}

int conv_layer_dgrad(const ConvLayer& L, const ConvDgradIO& io, float* ws, size_t wsf, hipStream_t st,
                     const ConvDgradIO* io2) {
  int Ho, Wo;
  conv_out_dims(L, Ho, Wo);
  switch (L.kind) {
    case 0: {  // dx (big) from dz (small); the ELU' reference (the layer's saved input) has dx's storage type
      if (!io2 && tconv2d_direct_supported(L.Cout, L.Cin, Ho, Wo))
        return tconv2d_direct(io.dz, L.out_bs, io.w, nullptr, io.dx, L.in_bs, io.dact_in, L.B, L.Cout, L.Cin, Ho, Wo,
                              0, st, L.in_bf16, L.out_bf16);
      if (L.in_bf16 || L.out_bf16) return bf16_unsupported();
      const Conv2dDgradParams p = p_conv2d_dgrad(L, io, Ho, Wo);
      if (!io2) return conv2d_dgrad(p, ws, wsf, st);
      const Conv2dDgradParams q = p_conv2d_dgrad(L, *io2, Ho, Wo);
      return conv2d_dgrad(p, ws, wsf, st, &q);
    }
    case 1: {  // dx (small) = strided conv of dz (big) with the same weight tensor
      if (!io2 && conv2d_direct_supported(L.Cout, L.Cin, L.Hin, L.Win))
        return conv2d_direct(io.dz, L.out_bs, io.w, nullptr, io.dx, L.in_bs, io.dact_in, L.B, L.Cout, L.Cin, L.Hin,
                             L.Win, 0, st, L.out_bf16, L.in_bf16);
      if (L.in_bf16 || L.out_bf16) return bf16_unsupported();
      const Conv2dFwdParams p = p_tconv2d_dgrad(L, io, Ho, Wo);
      if (!io2) return conv2d_fwd(p, ws, wsf, st);
      const Conv2dFwdParams q = p_tconv2d_dgrad(L, *io2, Ho, Wo);
      return conv2d_fwd(p, ws, wsf, st, &q);
    }
    case 2: {  // dx (big, the layer's input) from dz (small)
      Conv1dDgradParams p = p_conv1d_dgrad(L, io, Wo);
      p.big_bf16 = L.in_bf16;
      p.s_bf16 = L.out_bf16;
      if ((L.in_bf16 || L.out_bf16) && !tconv1d_stream_supported(p)) return bf16_unsupported();
      if (!io2) return conv1d_dgrad(p, ws, wsf, st);
      Conv1dDgradParams q = p_conv1d_dgrad(L, *io2, Wo);
      q.big_bf16 = L.in_bf16;
      q.s_bf16 = L.out_bf16;
      return conv1d_dgrad(p, ws, wsf, st, &q);
    }
    default: {  // dx (small) = strided conv of dz (big, the layer's output gradient)
      Conv1dFwdParams p = p_tconv1d_dgrad(L, io, Wo);
      p.x_bf16 = L.out_bf16;
      p.y_bf16 = L.in_bf16;
      if ((L.in_bf16 || L.out_bf16) && !conv1d_stream_supported(p)) return bf16_unsupported();
      if (!io2) return conv1d_fwd(p, ws, wsf, st);
      Conv1dFwdParams q = p_tconv1d_dgrad(L, *io2, Wo);
      q.x_bf16 = L.out_bf16;
      q.y_bf16 = L.in_bf16;
      return conv1d_fwd(p, ws, wsf, st, &q);
    }
  }
}

bool conv_layer_bwd_fusable(const ConvLayer& L, const ConvWgradIO& io, const ConvDgradIO& dio) {
  if (sched(LSHM_SCHED_NO_ONE_PASS_BWD) || !dio.dx || dio.dz != io.dz || L.in_bs % 4 != 0) return false;
  // the ELU' reference of the data gradient, if any, must be the layer's own input (it is, for every layer behind an ELU)
  if (dio.dact_in && dio.dact_in != io.x) return false;
  if (L.kind == 0) {  // 2-D conv: conv1 (8 -> 12 channels), fp32 storage
    int Ho2, Wo2;
    conv_out_dims(L, Ho2, Wo2);
    return !L.out_bf16 && L.out_bs % 4 == 0 && conv2d_bwd_lds_supported(L.Cout, L.Cin, Ho2, Wo2);  // (its input may be bf16)
  }
  if (L.kind == 1)  // 2-D transposed: the outermost decoder layer (8 -> 4 channels) and, fp32 storage, tconv4 (12 -> 8)
    return (!sched(LSHM_SCHED_NO_BWD_FUSED2D) && tconv2d_bwd_fused_supported(L.Cin, L.Cout, L.Hin, L.Win)) ||
           (!L.in_bf16 && L.out_bs % 4 == 0 && conv2d_bwd_lds_supported(L.Cin, L.Cout, L.Hin, L.Win));  // (its output may be bf16)
  if (L.kind == 3)  // transposed: small = the layer's input, big = dz
    return conv1d_bwd_fused_supported(L.Cin, L.Cout, 0) && (dio.dact_in || conv1d_bwd_fused2_supported(L.Cin, L.Cout, 0)) &&
           conv1d_wgrad_direct_supported(L.Cin, L.Cout, L.Win);
  int Ho, Wo;
  conv_out_dims(L, Ho, Wo);  // conv: small = dz, big = the layer's input
  return conv1d_bwd_fused2_supported(L.Cout, L.Cin, 1) && conv1d_wgrad_direct_supported(L.Cout, L.Cin, Wo);
}

int conv_layer_wgrad(const ConvLayer& L, const ConvWgradIO& io, float* ws, size_t ws_floats, int accumulate,
                     hipStream_t st, const ConvWgradIO* io2, GradJobs* defer, const ConvDgradIO* fuse,
                     const ConvDgradIO* fuse2) {
  const int G = io2 ? 2 : 1;
  if (defer) {  // private scratch that survives until grad_jobs_finish
    ws_floats = conv_workspace_floats(L) * G;
    ws = defer->take(ws_floats);
  }
  if (!ws || ws_floats < BIAS_WS_FLOATS + 16) {
    set_last_error("conv wgrad: workspace too small");
    return LSHM_ERR_WORKSPACE;
  }
  int Ho, Wo;
  conv_out_dims(L, Ho, Wo);
  const LayerGemm g = wgrad_gemm(L);
  const bool tr = transposed(L);
  // small / big tensors of the underlying strided-conv geometry
  const long s_bs = tr ? L.in_bs : L.out_bs;
  const long big_bs = tr ? L.out_bs : L.in_bs;
  const int Cs = g.M, Cb = tr ? L.Cout : L.Cin;
  auto small_of = [&](const ConvWgradIO& q) { return tr ? q.x : q.dz; };
  auto big_of = [&](const ConvWgradIO& q) { return tr ? q.dz : q.x; };
  float* bias_ws = ws;
  float* gemm_ws = ws + BIAS_WS_FLOATS;
  const size_t gemm_wsf = ws_floats - BIAS_WS_FLOATS;
  int rc;
  // bf16 storage: the `big` tensor (the layer's input for a conv, its output gradient for a transposed conv) of the two
  // outer layers and the `small` one of the outermost layer may be bf16; the kernels' launchers refuse the rest
  const int big_bf16 = tr ? L.out_bf16 : L.in_bf16;
  const int small_bf16 = tr ? L.in_bf16 : L.out_bf16;
  if (L.kind < 2) {
    const int Hs = tr ? L.Hin : Ho, Ws = tr ? L.Win : Wo;
    if (fuse) {
      if (!io2 && !small_bf16 && conv2d_bwd_lds_supported(Cs, Cb, Hs, Ws) &&
          gemm_wsf >= conv2d_wgrad_direct_workspace_floats(Cs, Cb))  // the 12 <-> 8 channel layers (conv2d_fused.hip)
        return conv2d_bwd_lds(small_of(io), s_bs, big_of(io), big_bs, fuse->w, fuse->dx, tr ? 0 : 1, fuse->dact_in ? 1 : 0, io.dw, io.db,
                              L.B, Cs, Cb, Hs, Ws, gemm_ws, gemm_wsf, accumulate, st, defer, big_bf16);
      if (!tr || io2 || !tconv2d_bwd_fused_supported(Cs, Cb, Hs, Ws) || gemm_wsf < conv2d_wgrad_direct_workspace_floats(Cs, Cb)) {
        set_last_error("conv wgrad: fused data gradient not available for this layer");
        return LSHM_ERR_UNSUPPORTED;
      }
      return tconv2d_bwd_fused(io.x, s_bs, io.dz, big_bs, fuse->w, fuse->dx, fuse->dact_in ? 1 : 0, io.dw, io.db, L.B, Hs, Ws,
                               gemm_ws, gemm_wsf, accumulate, st, defer, big_bf16, small_bf16);
    }
    if (!io2 && conv2d_wgrad_direct_supported(Cs, Cb, Hs, Ws) &&
        gemm_wsf >= conv2d_wgrad_direct_workspace_floats(Cs, Cb)) {
      // weight and bias gradient in one pass (dz is `big` for the transposed conv, `small` otherwise)
      return conv2d_wgrad_direct(small_of(io), s_bs, big_of(io), big_bs, io.dw, io.db, tr ? 2 : 1, L.B, Cs, Cb, Hs,
                                 Ws, gemm_ws, gemm_wsf, accumulate, st, defer, big_bf16, small_bf16);
    } else {
      if (big_bf16 || small_bf16) return bf16_unsupported();
      Conv2dWgradParams p{small_of(io), big_of(io), io.dw, L.B, Cs, Hs, Ws, Cb, s_bs, big_bs,
                          g.M, g.N, g.K, accumulate, {}};
      if (io2) {
        Conv2dWgradParams q = p;
        q.s = small_of(*io2); q.big = big_of(*io2); q.dw = io2->dw;
        rc = conv2d_wgrad(p, gemm_ws, gemm_wsf, st, &q, defer);
      } else {
        rc = conv2d_wgrad(p, gemm_ws, gemm_wsf, st, nullptr, defer);
      }
    }
  } else {
    const int Ls = tr ? L.Win : Wo, Lb = tr ? Wo : L.Win;
    const int bias_from = !io.db ? 0 : tr ? 2 : 1;
    const bool stream_ok =
        conv1d_wgrad_direct_supported(Cs, Cb, Ls) && gemm_wsf >= G * conv1d_wgrad_direct_workspace_floats(Cs, Cb) &&
        conv1d_wgrad_stream_supported(Cs, Cb, Ls, Lb, tr ? 0 : 1, bias_from, s_bs, big_bs, small_of(io), big_of(io)) &&
        (!io2 || conv1d_wgrad_stream_supported(Cs, Cb, Ls, Lb, tr ? 0 : 1, bias_from, s_bs, big_bs, small_of(*io2),
                                               big_of(*io2)));
    if (stream_ok) {
      // weight and bias gradient in one pass (dz is `big` for the transposed conv, `small` otherwise)
      FusedDgrad fd{};
      if (fuse) fd = FusedDgrad{fuse->w, fuse2 ? fuse2->w : nullptr, fuse->dx, fuse2 ? fuse2->dx : nullptr, L.in_bs,
                                fuse->dact_in ? 1 : 0};
      return conv1d_wgrad_direct(small_of(io), s_bs, big_of(io), big_bs, io.dw, io.db, tr ? 2 : 1, L.Cout, L.B, Cs,
                                 Cb, Ls, Lb, tr ? 0 : 1, gemm_ws, gemm_wsf, accumulate, st,
                                 io2 ? small_of(*io2) : nullptr, io2 ? big_of(*io2) : nullptr,
                                 io2 ? io2->dw : nullptr, io2 ? io2->db : nullptr, defer, big_bf16, fuse ? &fd : nullptr,
                                 small_bf16);
    }
    if (fuse) { set_last_error("conv wgrad: fused data gradient not available for this layer"); return LSHM_ERR_UNSUPPORTED; }
    if (big_bf16 || small_bf16) return bf16_unsupported();
    const bool use_mid = !sched(LSHM_SCHED_NO_WGRAD_MID);
    if (use_mid && gemm_wsf >= G * conv1d_wgrad_mid_workspace_floats(Cs, Cb) &&
        conv1d_wgrad_mid_supported(Cs, Cb, Ls, Lb, tr ? 0 : 1, bias_from, s_bs, big_bs, small_of(io), big_of(io)) &&
        (!io2 || conv1d_wgrad_mid_supported(Cs, Cb, Ls, Lb, tr ? 0 : 1, bias_from, s_bs, big_bs, small_of(*io2),
                                            big_of(*io2))))
      return conv1d_wgrad_mid(small_of(io), s_bs, big_of(io), big_bs, io.dw, io.db, tr ? 2 : 1, L.B, Cs, Cb, Ls, Lb,
                              tr ? 0 : 1, gemm_ws, gemm_wsf, accumulate, st, io2 ? small_of(*io2) : nullptr,
                              io2 ? big_of(*io2) : nullptr, io2 ? io2->dw : nullptr, io2 ? io2->db : nullptr, defer);
    Conv1dWgradParams p{small_of(io), big_of(io), io.dw, L.B, Cs, Ls, Cb, Lb, tr ? 0 : 1, s_bs, big_bs,
                        g.M, g.N, g.K, accumulate, {}};
    if (io2) {
      Conv1dWgradParams q = p;
      q.s = small_of(*io2); q.big = big_of(*io2); q.dw = io2->dw;
      rc = conv1d_wgrad(p, gemm_ws, gemm_wsf, st, &q, defer);
    } else {
      rc = conv1d_wgrad(p, gemm_ws, gemm_wsf, st, nullptr, defer);
    }
  }
  if (rc || !io.db) return rc;
  const float* dz2 = io2 ? io2->dz : nullptr;
  float* db2 = io2 ? io2->db : nullptr;
  const long HW = (long)Ho * Wo;
  if (defer) {
    const size_t mark_c = defer->chan.size(), mark_s = defer->sums.size();
    bool ok = defer->add_channel_sum(io.dz, L.out_bs, L.B, L.Cout, HW, io.db, accumulate);
    if (ok && io2) ok = defer->add_channel_sum(dz2, L.out_bs, L.B, L.Cout, HW, db2, accumulate);
    if (ok) return LSHM_OK;
    defer->chan.resize(mark_c);  // cannot be expressed as jobs: sum in place below
    defer->sums.resize(mark_s);
  }
  if ((long)L.B * HW <= 65536)  // small tensor: one workgroup per channel, no second stage
    return channel_sum_direct(io.dz, L.out_bs, L.B, L.Cout, HW, io.db, accumulate, st, dz2, db2);
  const int Sb = bias_slices(L);
  float* bias_ws2 = bias_ws + BIAS_WS_FLOATS / 2;
  rc = channel_sum_partials(io.dz, L.out_bs, L.B, L.Cout, HW, bias_ws, Sb, st, dz2, bias_ws2);
  if (rc) return rc;
  return reduce_partials(bias_ws, io.db, L.Cout, Sb, accumulate, st, io2 ? bias_ws2 : nullptr, db2);
}

// --------------------------------------------------------------------------
// dense layers
// --------------------------------------------------------------------------
int linear_fwd(const LinFwdIO& io, long ldx, long ldy, int B, int K, int N, int act, float* ws, size_t wsf,
               hipStream_t st, const LinFwdIO* io2) {
  StridedGemmParams p{io.x, io.w, io.b, io.y, nullptr, ldx, 1, 1, K, ldy, 1, 0, 0, act, B, N, K, {}, nullptr, 0, 0};
  if (!io2) return strided_gemm(p, false, false, ws, wsf, st);
  StridedGemmParams q = p;
  q.a = io2->x; q.b = io2->w; q.bias = io2->b; q.c = io2->y;
  return strided_gemm(p, false, false, ws, wsf, st, &q);
}
int linear_dgrad(const LinDgradIO& io, long lddz, long lddx, long ldxs, long ldadd, int add_n, int B, int K,
                 int N, float* ws, size_t wsf, hipStream_t st, const LinDgradIO* io2) {
  StridedGemmParams p{io.dz, io.w, nullptr, io.dx, io.xsaved, lddz, 1, K, 1, lddx, 1, ldxs, 1, 0, B, K, N, {},
                      io.add, ldadd, add_n};
  if (!io2) return strided_gemm(p, false, true, ws, wsf, st);
  StridedGemmParams q = p;
  q.a = io2->dz; q.b = io2->w; q.c = io2->dx; q.dact = io2->xsaved; q.add = io2->add;
  return strided_gemm(p, false, true, ws, wsf, st, &q);
}
__global__ void copy2d_kernel(const float* __restrict__ src, long lds_, float* __restrict__ dst,
                              long ldd, int rows, int cols) {
  const long n = (long)rows * cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long r = i / cols, c = i - r * cols;
    dst[r * ldd + c] = src[r * lds_ + c];
  }
}
int copy2d(const float* src, long lds_, float* dst, long ldd, int rows, int cols, hipStream_t st) {
  hipLaunchKernelGGL(copy2d_kernel, dim3(min(cdiv((long)rows * cols, 256), 1024)), dim3(256), 0, st,
                     src, lds_, dst, ldd, rows, cols);
  return check_launch("copy2d");
}
// db[n] = sum_b dz[b, n]: rows are split over the 4 waves of a block and combined through LDS;
// blockIdx.y selects the optional second problem.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ dz0, const float* __restrict__ dz1,
                                                     long lddz, int B, int N, float* __restrict__ db0,
                                                     float* __restrict__ db1) {
  __shared__ float red[256];
  const float* dz = blockIdx.y ? dz1 : dz0;
  float* db = blockIdx.y ? db1 : db0;
  const int nl = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + nl;
  float acc = 0.f;
  if (n < N)
    for (int b = part; b < B; b += 4) acc += dz[(long)b * lddz + n];
  red[threadIdx.x] = acc;
  __syncthreads();
  if (part == 0 && n < N) db[n] = (red[nl] + red[64 + nl]) + (red[128 + nl] + red[192 + nl]);
}
int grad_jobs_launch_dense(GradJobs& jobs, hipStream_t st) {
  if (jobs.dense.empty()) return LSHM_OK;
  const int rc = strided_gemm_batch_mm(jobs.dense.data(), (int)jobs.dense.size(), st);
  jobs.dense.clear();
  return rc;
}
int linear_wgrad(const LinWgradIO& io, long ldx, long lddz, int B, int K, int N, float* ws, size_t wsf,
                 hipStream_t st, const LinWgradIO* io2, GradJobs* defer) {
  if (defer && !defer->batch_dense) {
    wsf = (igemm_workspace_floats(N, K, B, 1) + 16) * (io2 ? 2 : 1);
    ws = defer->take(wsf);
    if (!ws) { set_last_error("linear wgrad: deferred scratch exhausted"); return LSHM_ERR_WORKSPACE; }
  }
  StridedGemmParams p{io.dz, io.x, nullptr, io.dw, nullptr, 1, lddz, ldx, 1, K, 1, 0, 0, 0, N, K, B, {},
                      nullptr, 0, 0};
  int rc;
  if (defer && defer->batch_dense) {  // launched together with the other dense layers (grad_jobs_launch_dense)
    defer->dense.push_back(p);
    if (io2) {
      StridedGemmParams q = p;
      q.a = io2->dz; q.b = io2->x; q.c = io2->dw;
      defer->dense.push_back(q);
    }
    rc = LSHM_OK;
  } else if (io2) {
    StridedGemmParams q = p;
    q.a = io2->dz; q.b = io2->x; q.c = io2->dw;
    rc = strided_gemm(p, true, true, ws, wsf, st, &q, defer);
  } else {
    rc = strided_gemm(p, true, true, ws, wsf, st, nullptr, defer);
  }
  if (rc || !io.db) return rc;
  if (defer) {  // column sums of dz: S = B partials of stride lddz
    defer->sums.push_back(SumJob{io.dz, io.db, lddz, N, B, 0, 0, 0, 0, 0, 0});
    if (io2) defer->sums.push_back(SumJob{io2->dz, io2->db, lddz, N, B, 0, 0, 0, 0, 0, 0});
    return LSHM_OK;
  }
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(N, 64), io2 ? 2 : 1), dim3(256), 0, st, io.dz,
                     io2 ? io2->dz : nullptr, lddz, B, N, io.db, io2 ? io2->db : nullptr);
  return check_launch("colsum");
}

}  // namespace lshm
