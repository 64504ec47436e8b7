// extern "C" surface of liblshm_hip (declared in include/lshm.h): argument
// checking + translation to the internal launchers.  No torch types, no
// allocation, no synchronisation.
#include "../../include/lshm.h"
#include "kernels.h"
#include "deep2d.h"
#include "chain1d_full.h"

#include <string.h>

#include <map>
#include <mutex>
#include <utility>
#include <vector>
#include <cxxabi.h>
#include <stdlib.h>

namespace lshm {
thread_local hipEvent_t launch_stop_event = nullptr;
thread_local unsigned launch_stop_count = 0;

static thread_local unsigned t_schedule = 0;
ScheduleScope::ScheduleScope(unsigned word) : prev(t_schedule) { t_schedule = word; }
ScheduleScope::~ScheduleScope() { t_schedule = prev; }
unsigned schedule_word() { return t_schedule; }

// ---- per-launch trace (diagnostics; see common.h)
thread_local bool launch_trace_on = false;
namespace {
struct TraceRec { const void* kernel; hipStream_t st; unsigned grid, block; hipEvent_t start, stop; };
struct Trace {
  std::vector<TraceRec> recs;
  std::vector<LaunchTraceSlot> pool;
  size_t used = 0;
  bool with_start = true;
};
thread_local Trace* g_trace = nullptr;
}  // namespace
bool launch_trace_take(const void* kernel, dim3 g, dim3 b, hipStream_t st, LaunchTraceSlot* slot) {
  Trace* t = g_trace;
  if (!t || t->used >= t->pool.size()) return false;
  *slot = t->pool[t->used++];
  t->recs.push_back(TraceRec{kernel, st, g.x * g.y * g.z, b.x * b.y * b.z, slot->start, slot->stop});
  return true;
}

static thread_local char g_err[256] = "";
void set_last_error(const char* msg) {
  strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}
int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) return LSHM_OK;
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
  return (int)e;
}
int device_lds_bytes() {
  int dev = 0, v = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return v;
}
int kernel_budget_ok(const void* kernel, int threads, size_t dyn_lds, const char* what) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, std::pair<size_t, int>> known;  // (kernel, device) -> (static LDS, max threads)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return LSHM_OK; }  // no device: the launch itself reports it
  std::pair<size_t, int> kv;
  {
    std::lock_guard<std::mutex> lk(mu);
    auto it = known.find({kernel, dev});
    if (it == known.end()) {
      hipFuncAttributes fa;
      if (hipFuncGetAttributes(&fa, kernel) != hipSuccess) { (void)hipGetLastError(); return LSHM_OK; }
      it = known.emplace(std::make_pair(kernel, dev), std::make_pair((size_t)fa.sharedSizeBytes, fa.maxThreadsPerBlock)).first;
    }
    kv = it->second;
  }
  const int lds_max = device_lds_bytes();
  if (lds_max > 0 && kv.first + dyn_lds > (size_t)lds_max) {
    snprintf(g_err, sizeof(g_err), "%s: needs %zu bytes of LDS per workgroup, the device has %d", what, kv.first + dyn_lds, lds_max);
    return LSHM_ERR_UNSUPPORTED;
  }
  if (kv.second > 0 && threads > kv.second) {
    snprintf(g_err, sizeof(g_err), "%s: register budget allows %d threads per workgroup, the launch needs %d", what, kv.second, threads);
    return LSHM_ERR_UNSUPPORTED;
  }
  return LSHM_OK;
}
int raise_dynamic_lds(const void* kernel, size_t dyn_lds, const char* what) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, size_t> raised;  // (kernel, device) -> limit already set
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return LSHM_OK; }
  std::lock_guard<std::mutex> lk(mu);
  auto it = raised.find({kernel, dev});
  if (it != raised.end() && it->second >= dyn_lds) return LSHM_OK;
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_lds) != hipSuccess) {
    (void)hipGetLastError();
    snprintf(g_err, sizeof(g_err), "%s: cannot raise the dynamic LDS limit to %zu bytes", what, dyn_lds);
    return LSHM_ERR_UNSUPPORTED;
  }
  raised[{kernel, dev}] = dyn_lds;
  return LSHM_OK;
}
}  // namespace lshm

using namespace lshm;

#define ST(s) reinterpret_cast<hipStream_t>(s)
#define REQUIRE(cond, msg)        \
  do {                            \
    if (!(cond)) {                \
      set_last_error(msg);        \
      return LSHM_ERR_ARG;        \
    }                             \
  } while (0)

static int make_layer(int kind, int B, int Cin, int Cout, int Hin, int Win, long in_bs, long out_bs,
                      ConvLayer* L) {
  REQUIRE(kind >= 0 && kind <= 3, "conv: unknown kind");
  REQUIRE(B > 0 && Cin > 0 && Cout > 0 && Hin > 0 && Win > 0, "conv: non-positive dimension");
  if (kind == 0) REQUIRE(Hin % 4 == 0 && Win % 4 == 0, "conv2d k4s2p1: H and W must be multiples of 4");
  if (kind == 1) REQUIRE(Hin >= 1 && Win >= 1, "tconv2d: bad size");
  if (kind >= 2) REQUIRE(Hin == 1, "1D conv: Hin must be 1");
  if (kind == 2) REQUIRE(Win % 16 == 0, "conv1d k4s4p1: L must be a multiple of 16");
  L->kind = kind; L->B = B; L->Cin = Cin; L->Cout = Cout; L->Hin = Hin; L->Win = Win;
  int Ho, Wo;
  conv_out_dims(*L, Ho, Wo);
  L->in_bs = in_bs ? in_bs : (long)Cin * Hin * Win;
  L->out_bs = out_bs ? out_bs : (long)Cout * Ho * Wo;
  REQUIRE(L->in_bs % 4 == 0 && L->out_bs % 4 == 0, "conv: batch strides must be multiples of 4");
  return LSHM_OK;
}

extern "C" {

int lshm_version(void) { return 100; }
const char* lshm_last_error_string(void) { return g_err; }
void lshm_set_tuning(int mode, int force) { igemm_set_tuning(mode, force); }
size_t lshm_tuning_export(char* buf, size_t cap) { return igemm_tuning_export(buf, cap); }
int lshm_tuning_import(const char* text) { return igemm_tuning_import(text); }

int lshm_uv_harmonics(const float* uv, const float* scales, int H, int B, float* out, lshm_stream_t s) {
  REQUIRE(uv && scales && out && H > 0 && B >= 0, "uv_harmonics: bad argument");
  if (B == 0) return LSHM_OK;
  return uv_harmonics(uv, scales, H, B, out, ST(s));
}

size_t lshm_conv_workspace_floats(int kind, int B, int Cin, int Cout, int Hin, int Win) {
  ConvLayer L;
  if (make_layer(kind, B, Cin, Cout, Hin, Win, 0, 0, &L)) return 0;
  return conv_workspace_floats(L);
}
int lshm_conv_fwd(int kind, const float* x, const float* w, const float* bias, float* y, int B, int Cin,
                  int Cout, int Hin, int Win, long in_bs, long out_bs, int act, float* ws, size_t wsf,
                  lshm_stream_t s) {
  REQUIRE(x && w && y, "conv_fwd: null pointer");
  ConvLayer L;
  int rc = make_layer(kind, B, Cin, Cout, Hin, Win, in_bs, out_bs, &L);
  if (rc) return rc;
  return conv_layer_fwd(L, ConvFwdIO{x, w, bias, y}, act, ws, ws ? wsf : 0, ST(s));
}
int lshm_conv_fwd_pair(int kind, const float* x0, const float* w0, const float* bias0, float* y0,
                       const float* x1, const float* w1, const float* bias1, float* y1, int B, int Cin, int Cout,
                       int Hin, int Win, long in_bs, long out_bs, int act, float* ws, size_t wsf, lshm_stream_t s) {
  REQUIRE(x0 && w0 && y0 && x1 && w1 && y1, "conv_fwd_pair: null pointer");
  ConvLayer L;
  int rc = make_layer(kind, B, Cin, Cout, Hin, Win, in_bs, out_bs, &L);
  if (rc) return rc;
  const ConvFwdIO io1{x1, w1, bias1, y1};
  return conv_layer_fwd(L, ConvFwdIO{x0, w0, bias0, y0}, act, ws, ws ? wsf : 0, ST(s), &io1);
}
int lshm_conv_dgrad(int kind, const float* dz, const float* w, float* dx, const float* y_in_saved, int B,
                    int Cin, int Cout, int Hin, int Win, long in_bs, long out_bs, float* ws, size_t wsf,
                    lshm_stream_t s) {
  REQUIRE(dz && w && dx, "conv_dgrad: null pointer");
  ConvLayer L;
  int rc = make_layer(kind, B, Cin, Cout, Hin, Win, in_bs, out_bs, &L);
  if (rc) return rc;
  return conv_layer_dgrad(L, ConvDgradIO{dz, w, dx, y_in_saved}, ws, ws ? wsf : 0, ST(s));
}
int lshm_conv_wgrad(int kind, const float* x, const float* dz, float* dw, float* db, int B, int Cin,
                    int Cout, int Hin, int Win, long in_bs, long out_bs, float* ws, size_t wsf,
                    int accumulate, lshm_stream_t s) {
  REQUIRE(x && dz && dw && ws, "conv_wgrad: null pointer");
  ConvLayer L;
  int rc = make_layer(kind, B, Cin, Cout, Hin, Win, in_bs, out_bs, &L);
  if (rc) return rc;
  return conv_layer_wgrad(L, ConvWgradIO{x, dz, dw, db}, ws, wsf, accumulate, ST(s));
}
int lshm_conv_bwd_fused_ex(int kind, const float* x, const float* dz, const float* w, float* dw, float* db, float* dx,
                           int elu_grad, int B, int Cin, int Cout, int Hin, int Win, float* ws, size_t wsf, unsigned schedule,
                           lshm_stream_t s) {
  ScheduleScope scope(schedule);
  return lshm_conv_bwd_fused(kind, x, dz, w, dw, db, dx, elu_grad, B, Cin, Cout, Hin, Win, ws, wsf, s);
}
int lshm_conv_bwd_fused(int kind, const float* x, const float* dz, const float* w, float* dw, float* db, float* dx,
                        int elu_grad, int B, int Cin, int Cout, int Hin, int Win, float* ws, size_t wsf,
                        lshm_stream_t s) {
  REQUIRE(x && dz && w && dw && dx && ws, "conv_bwd_fused: null pointer");
  ConvLayer L;
  int rc = make_layer(kind, B, Cin, Cout, Hin, Win, 0, 0, &L);
  if (rc) return rc;
  const ConvWgradIO io{x, dz, dw, db};
  const ConvDgradIO dio{dz, w, dx, elu_grad ? x : nullptr};
  if (!conv_layer_bwd_fusable(L, io, dio)) {
    set_last_error("conv_bwd_fused: no one-pass backward kernel for this layer");
    return LSHM_ERR_UNSUPPORTED;
  }
  return conv_layer_wgrad(L, io, ws, wsf, 0, ST(s), nullptr, nullptr, &dio);
}
int lshm_conv1d_chain3(int up, const float* x, const float* const* w, const float* const* bias, float* const* out,
                       const float* const* dact, int act, int pad, int B, lshm_stream_t s) {
  REQUIRE(x && w && out && B > 0 && (pad == 0 || pad == 1), "conv1d_chain3: bad argument");
  const int chd[4] = {12, 24, 48, 96}, chu[4] = {96, 48, 24, 12};
  if (!conv1d_chain_supported(up != 0, up ? chu : chd, up ? 16 : 1024)) {
    set_last_error("conv1d_chain3: not available on this device (LDS) or switched off by the calling scope's schedule word");
    return LSHM_ERR_UNSUPPORTED;
  }
  Chain1dStage st[3];
  long L = up ? 16 : 1024;
  for (int k = 0; k < 3; ++k) {
    REQUIRE(w[k] && out[k], "conv1d_chain3: null stage pointer");
    L = up ? 4 * L : L / 4;
    const int c = up ? chu[k + 1] : chd[k + 1];
    st[k].w[0] = st[k].w[1] = w[k];
    st[k].bias[0] = st[k].bias[1] = bias ? bias[k] : nullptr;
    st[k].out[0] = st[k].out[1] = out[k];
    st[k].dact[0] = st[k].dact[1] = dact ? dact[k] : nullptr;
    st[k].out_bs = (long)c * L;
    st[k].act = act;
  }
  return conv1d_chain(up != 0, st, x, nullptr, up ? 96L * 16 : 12L * 1024, pad, B, ST(s));
}
int lshm_trace_begin(int capacity) { return lshm_trace_begin_ex(capacity, 1); }
int lshm_trace_begin_ex(int capacity, int with_start) {
  if (capacity < 1 || capacity > 65536) { set_last_error("trace_begin: capacity must be 1..65536"); return LSHM_ERR_ARG; }
  if (g_trace) { set_last_error("trace_begin: this thread is already recording"); return LSHM_ERR_ARG; }
  Trace* t = new Trace();
  t->with_start = with_start != 0;
  t->pool.resize(capacity);
  for (auto& sl : t->pool) {
    sl.start = nullptr;
    if ((t->with_start && hipEventCreate(&sl.start) != hipSuccess) || hipEventCreate(&sl.stop) != hipSuccess) {
      (void)hipGetLastError();
      set_last_error("trace_begin: cannot create events");
      delete t;  // (events created so far are leaked: a diagnostic path on a device that is out of events)
      return LSHM_ERR_ARG;
    }
  }
  g_trace = t;
  launch_trace_on = true;
  return LSHM_OK;
}
int lshm_trace_end(void) {
  if (!g_trace) { set_last_error("trace_end: this thread is not recording"); return LSHM_ERR_ARG; }
  launch_trace_on = false;
  return (int)g_trace->recs.size();
}
int lshm_trace_read(int index, char* name, int name_cap, float* start_us, float* dur_us, int* stream_index, unsigned* grid_threads) {
  Trace* t = g_trace;
  if (!t || launch_trace_on) { set_last_error("trace_read: call lshm_trace_end first"); return LSHM_ERR_ARG; }
  if (index < 0 || index >= (int)t->recs.size()) { set_last_error("trace_read: index past the end"); return LSHM_ERR_ARG; }
  const TraceRec& r = t->recs[index];
  float a = 0.f, d = 0.f;
  if (t->with_start) {
    if (hipEventElapsedTime(&a, t->recs[0].start, r.start) != hipSuccess || hipEventElapsedTime(&d, r.start, r.stop) != hipSuccess) {
      (void)hipGetLastError();
      set_last_error("trace_read: an event has not completed (synchronise the device first)");
      return LSHM_ERR_ARG;
    }
  } else {  // completion times only: `start` = completion time after the first recorded launch's, duration unknown (-1)
    if (hipEventElapsedTime(&a, t->recs[0].stop, r.stop) != hipSuccess) {
      (void)hipGetLastError();
      set_last_error("trace_read: an event has not completed (synchronise the device first)");
      return LSHM_ERR_ARG;
    }
    d = -1e-3f;
  }
  if (start_us) *start_us = a * 1000.f;
  if (dur_us) *dur_us = d * 1000.f;
  if (stream_index) {
    int k = 0;
    std::vector<hipStream_t> seen;
    for (int i = 0; i <= index; ++i) {
      bool f = false;
      for (size_t j = 0; j < seen.size(); ++j) if (seen[j] == t->recs[i].st) { f = true; if (i == index) k = (int)j; }
      if (!f) { if (i == index) k = (int)seen.size(); seen.push_back(t->recs[i].st); }
    }
    *stream_index = k;
  }
  if (grid_threads) *grid_threads = r.grid * r.block;
  if (name && name_cap > 0) {
    const char* mangled = hipKernelNameRefByPtr(r.kernel, r.st);
    int status = 1;
    char* dem = mangled ? abi::__cxa_demangle(mangled, nullptr, nullptr, &status) : nullptr;
    strncpy(name, (status == 0 && dem) ? dem : (mangled ? mangled : "?"), name_cap - 1);
    name[name_cap - 1] = 0;
    free(dem);
  }
  return LSHM_OK;
}
int lshm_trace_free(void) {
  Trace* t = g_trace;
  if (!t) return LSHM_OK;
  launch_trace_on = false;
  for (auto& sl : t->pool) { if (sl.start) (void)hipEventDestroy(sl.start); (void)hipEventDestroy(sl.stop); }
  delete t;
  g_trace = nullptr;
  return LSHM_OK;
}
int lshm_chain1d_full_fwd(const float* x1, const float* const* w, const float* const* bias, float* const* out, long ldmu, int B,
                          long long* stamps, lshm_stream_t s) {
  REQUIRE(x1 && w && bias && out && B > 0 && ldmu >= 16, "chain1d_full_fwd: bad argument");
  for (int i = 0; i < 12; ++i) REQUIRE(w[i] && bias[i] && out[i], "chain1d_full_fwd: null layer pointer");
  const int ch[5] = {12, 24, 48, 96, 192};
  if (!chain1d_full_supported(16, 16, 1, ch, 1024)) { set_last_error("chain1d_full_fwd: not available on this device"); return LSHM_ERR_UNSUPPORTED; }
  Chain1dFullArgs q{};
  const long obs[3] = {24L * 256, 48L * 64, 96L * 16}, ubs[3] = {48L * 64, 24L * 256, 12L * 1024};
  for (int g = 0; g < 2; ++g) {
    q.in[g] = x1;
    for (int k = 0; k < 3; ++k) {
      q.dn[k].w[g] = w[k]; q.dn[k].bias[g] = bias[k]; q.dn[k].out[g] = out[k]; q.dn[k].dact[g] = nullptr;
      q.up[k].w[g] = w[9 + k]; q.up[k].bias[g] = bias[9 + k]; q.up[k].out[g] = out[9 + k]; q.up[k].dact[g] = nullptr;
    }
    q.w5[g] = w[3]; q.b5[g] = bias[3]; q.cat1[g] = out[3];
    q.fc1w[g] = w[4]; q.fc1b[g] = bias[4]; q.fc2inw[g] = w[5]; q.fc2inb[g] = bias[5]; q.fc2outw[g] = w[6]; q.fc2outb[g] = bias[6];
    q.fc3w[g] = w[7]; q.fc3b[g] = bias[7];
    q.z1[g] = out[4]; q.mu[g] = out[5]; q.cat3[g] = out[6]; q.d0[g] = out[7];
    q.wt0[g] = w[8]; q.bt0[g] = bias[8]; q.t0[g] = out[8];
  }
  q.in_bs = 12L * 1024;
  q.mu_ld = ldmu;
  for (int k = 0; k < 3; ++k) { q.dn[k].out_bs = obs[k]; q.dn[k].act = 1; q.up[k].out_bs = ubs[k]; q.up[k].act = 1; }
  q.stamps = stamps;
  return chain1d_full_fwd(q, B, 1, ST(s));
}
size_t lshm_deep2d_packed_floats(void) { return deep2d_packed_floats(); }
int lshm_deep2d_fwd(const float* x2, const float* const* w, const float* const* bias, float* const* out, long ldmu, float* packed,
                    int B, int variant, long long* stamps, lshm_stream_t s) {
  REQUIRE(x2 && w && bias && out && packed && B > 0 && ldmu >= 224, "deep2d_fwd: bad argument");
  for (int i = 0; i < 11; ++i) REQUIRE(w[i] && bias[i] && out[i], "deep2d_fwd: null layer pointer");
  const Deep2dWeights dw{nullptr, w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], w[8], w[9], w[10]};
  int rc = deep2d_pack(dw, packed, 0, (variant & 4) ? 1 : 0, ST(s));
  if (rc) return rc;
  Deep2dIO io;
  io.x2 = x2;
  io.b3 = bias[0]; io.b4 = bias[1]; io.b5 = bias[2]; io.bfc1 = bias[3]; io.bfc2in = bias[4]; io.bfc2out = bias[5]; io.bfc3 = bias[6];
  io.bt0 = bias[7]; io.bt1 = bias[8]; io.bt2 = bias[9]; io.bt3 = bias[10];
  io.a3 = out[0]; io.a4 = out[1]; io.cat1 = out[2]; io.z1 = out[3]; io.mu = out[4]; io.mu_ld = ldmu; io.cat3 = out[5]; io.d0 = out[6];
  io.t0 = out[7]; io.t1 = out[8]; io.t2 = out[9]; io.t3 = out[10];
  io.stamps = stamps;
  return deep2d_fwd(io, packed, B, variant, ST(s));
}
int lshm_deep2d_bwd(const float* g_t2, const float* const* w, const float* const* saved, long ldmu, const float* gmu, long ldgmu,
                    float* const* out, float* packed, int B, int variant, lshm_stream_t s) {
  REQUIRE(g_t2 && w && saved && out && packed && B > 0 && ldmu >= 224, "deep2d_bwd: bad argument");
  for (int i = 0; i < 12; ++i) REQUIRE(w[i], "deep2d_bwd: null weight pointer");
  for (int i = 0; i < 10; ++i) REQUIRE(saved[i], "deep2d_bwd: null saved-activation pointer");
  for (int i = 0; i < 11; ++i) REQUIRE(out[i], "deep2d_bwd: null output pointer");
  const Deep2dWeights dw{w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], w[8], w[9], w[10], w[11]};
  int rc = deep2d_pack(dw, packed, 1, (variant & 4) ? 1 : 0, ST(s));
  if (rc) return rc;
  Deep2dBwdIO io;
  io.g_t2 = g_t2;
  io.s_t1 = saved[0]; io.s_t0 = saved[1]; io.s_cat3 = saved[2]; io.s_mu = saved[3]; io.s_mu_ld = ldmu; io.s_z1 = saved[4]; io.s_cat1 = saved[5];
  io.s_c4 = saved[6]; io.s_c3 = saved[7]; io.s_c2 = saved[8]; io.s_c1 = saved[9];
  io.gmu = gmu; io.gmu_ld = ldgmu;
  io.g_t1 = out[0]; io.g_t0 = out[1]; io.g_d0 = out[2]; io.g_cat3 = out[3]; io.g_mu = out[4]; io.g_mu_ld = 224; io.g_z1 = out[5];
  io.g_cat1 = out[6]; io.g_c4 = out[7]; io.g_c3 = out[8]; io.g_c2 = out[9]; io.g_c1 = out[10];
  return deep2d_bwd(io, packed, B, variant, ST(s));
}
static int dense_mid_fwd(int L, const float* cat1, const float* const* wb, float* z1, float* mu, long ldmu, float* cat3, float* d0,
                         int B, lshm_stream_t s) {
  REQUIRE(cat1 && wb && z1 && mu && cat3 && d0 && B > 0 && L > 0 && ldmu >= L, "dense1d_fwd: bad argument");
  if (!dense1d_built(L)) { set_last_error("dense_fwd: latent width must be 16"); return LSHM_ERR_UNSUPPORTED; }
  const Dense1dFwdIO io{cat1, wb[0], wb[1], wb[2], wb[3], wb[4], wb[5], wb[6], wb[7], z1, mu, cat3, d0};
  return dense1d_fwd(io, nullptr, ldmu, B, ST(s), L);
}
static int dense_mid_bwd(int L, const float* dd0, const float* cat3, const float* mu, long ldmu, const float* gmu, long ldgmu,
                         const float* z1, const float* cat1, const float* const* w, float* dcat3, float* dzmu, float* dz1,
                         float* dcat1, int B, lshm_stream_t s) {
  REQUIRE(dd0 && cat3 && mu && gmu && z1 && cat1 && w && dcat3 && dzmu && dz1 && dcat1 && B > 0 && ldmu >= L && ldgmu >= L,
          "dense1d_bwd: bad argument");
  if (!dense1d_built(L)) { set_last_error("dense_bwd: latent width must be 16"); return LSHM_ERR_UNSUPPORTED; }
  const Dense1dBwdIO io{dd0, cat3, mu, gmu, z1, cat1, w[0], w[1], w[2], w[3], dcat3, dzmu, dz1, dcat1};
  return dense1d_bwd(io, nullptr, ldmu, ldgmu, B, ST(s), L);
}
int lshm_dense1d_fwd(const float* cat1, const float* const* wb, float* z1, float* mu, long ldmu, float* cat3, float* d0, int B,
                     lshm_stream_t s) {
  return dense_mid_fwd(16, cat1, wb, z1, mu, ldmu, cat3, d0, B, s);
}
int lshm_dense1d_bwd(const float* dd0, const float* cat3, const float* mu, long ldmu, const float* gmu, long ldgmu, const float* z1,
                     const float* cat1, const float* const* w, float* dcat3, float* dzmu, float* dz1, float* dcat1, int B,
                     lshm_stream_t s) {
  return dense_mid_bwd(16, dd0, cat3, mu, ldmu, gmu, ldgmu, z1, cat1, w, dcat3, dzmu, dz1, dcat1, B, s);
}
int lshm_elu_bwd(const float* gy, const float* y, float* dz, long n, lshm_stream_t s) {
  REQUIRE(gy && y && dz && n >= 0, "elu_bwd: bad argument");
  if (n == 0) return LSHM_OK;
  return elu_bwd(gy, y, dz, n, ST(s));
}

size_t lshm_linear_workspace_floats(int B, int K, int N) {
  size_t a = igemm_workspace_floats(B, N, K, 1);
  const size_t b = igemm_workspace_floats(B, K, N, 1), c = igemm_workspace_floats(N, K, B, 1);
  if (b > a) a = b;
  if (c > a) a = c;
  return a + 16;
}
int lshm_linear_fwd(const float* x, long ldx, const float* w, const float* bias, float* y, long ldy,
                    int B, int K, int N, int act, float* ws, size_t wsf, lshm_stream_t s) {
  REQUIRE(x && w && y && B > 0 && K > 0 && N > 0 && ldx >= K && ldy >= N, "linear_fwd: bad argument");
  return linear_fwd(LinFwdIO{x, w, bias, y}, ldx, ldy, B, K, N, act, ws, ws ? wsf : 0, ST(s));
}
int lshm_linear_dgrad(const float* dz, long lddz, const float* w, float* dx, long lddx,
                      const float* x_saved, long ldxs, int B, int K, int N, float* ws, size_t wsf,
                      lshm_stream_t s) {
  REQUIRE(dz && w && dx && B > 0 && K > 0 && N > 0 && lddz >= N && lddx >= K, "linear_dgrad: bad argument");
  return linear_dgrad(LinDgradIO{dz, w, dx, x_saved, nullptr}, lddz, lddx, ldxs, 0, 0, B, K, N, ws, ws ? wsf : 0, ST(s));
}
int lshm_linear_wgrad(const float* x, long ldx, const float* dz, long lddz, float* dw, float* db, int B,
                      int K, int N, float* ws, size_t wsf, lshm_stream_t s) {
  REQUIRE(x && dz && dw && B > 0 && K > 0 && N > 0, "linear_wgrad: bad argument");
  return linear_wgrad(LinWgradIO{x, dz, dw, db}, ldx, lddz, B, K, N, ws, ws ? wsf : 0, ST(s));
}

size_t lshm_khm_workspace_floats(int N, int D, int K) { return khm_workspace_floats(N, D, K); }
int lshm_khm_fwd_bwd(const float* X, long ldx, const float* M, int N, int D, int K, float p, float eps,
                     double inv_count, float gscale, double* loss_sum, float* dX, long lddx, float* dM,
                     int accumulate_dx, float* ws, size_t wsf, lshm_stream_t s) {
  REQUIRE(loss_sum && dM, "khm_fwd_bwd: null output");
  return khm_fwd_bwd(X, ldx, M, N, D, K, p, eps, inv_count, gscale, loss_sum, dX, lddx, dM,
                     accumulate_dx, ws, wsf, ST(s));
}
int lshm_khm_offline_partials(const float* X, long ldx, const float* M, int N, int D, int K, float p,
                              float eps, float* num, float* den, float* ws, size_t wsf, lshm_stream_t s) {
  REQUIRE(num && den, "khm_offline_partials: null output");
  return khm_offline_partials(X, ldx, M, N, D, K, p, eps, num, den, ws, wsf, ST(s));
}
int lshm_khm_mean_distances(const float* X, long ldx, const float* M, int N, int D, int K, float p,
                            float* dist, float* ws, size_t wsf, lshm_stream_t s) {
  REQUIRE(dist, "khm_mean_distances: null output");
  return khm_mean_distances(X, ldx, M, N, D, K, p, dist, ws, wsf, ST(s));
}
int lshm_khm_assign(const float* dist, int K, int* argmin, float* prob, lshm_stream_t s) {
  REQUIRE(dist && (argmin || prob), "khm_assign: null pointer");
  return dist_epilogue(dist, K, argmin, prob, ST(s));
}
int lshm_cluster_sim_fwd_bwd(const float* M, int K, int D, float eps, float gscale, double* loss,
                             float* dM, int accumulate, lshm_stream_t s) {
  return cluster_sim_fwd_bwd(M, K, D, eps, gscale, loss, dM, accumulate, ST(s));
}
int lshm_aug_loss_fwd_bwd(const float* Z, long ldz, int rows, int D, int bpb, int batch_size, float gscale,
                          double* loss, float* dZ, long lddz, int accumulate, lshm_stream_t s) {
  return aug_loss_fwd_bwd(Z, ldz, rows, D, bpb, batch_size, gscale, loss, dZ, lddz, accumulate, ST(s));
}
int lshm_logcosh_fwd_bwd(const float* z, long ldz, int rows, int cols, float scale, double* loss,
                         float* dz, long lddz, int accumulate, lshm_stream_t s) {
  REQUIRE(z && rows >= 0 && cols >= 0, "logcosh: bad argument");
  return logcosh_mean_fwd_bwd(z, ldz, rows, cols, scale, loss, dz, lddz, accumulate, ST(s));
}

int lshm_residual_split(const float* x, const float* x1, float* out_row, float* out_col, int planes, int P,
                        lshm_stream_t s) {
  REQUIRE(x && x1 && out_col && planes > 0, "residual_split: bad argument");
  return residual_split(x, x1, out_row, out_col, planes, P, ST(s));
}
int lshm_resid_conv0(const float* x, const float* x1, const float* wT, const float* bT, float* yT, const float* wF, const float* bF,
                     float* yF, int B, lshm_stream_t s) {
  REQUIRE(x && x1 && wT && bT && yT && wF && bF && yF && B > 0, "resid_conv0: bad argument");
  if (!resid_conv0_supported(4, 128, 4, 8, 128 * 128)) { set_last_error("resid_conv0: switched off by the schedule word (LSHM_SCHED_NO_RESID_CONV0)"); return LSHM_ERR_UNSUPPORTED; }
  return resid_conv0(x, x1, wT, bT, yT, wF, bF, yF, 8L * 4096, B, ST(s));
}
int lshm_resid_conv0_keep(const float* x, const float* x1, const float* wT, const float* bT, float* yT, const float* wF, const float* bF,
                          float* yF, float* out_row, float* out_col, int B, lshm_stream_t s) {
  REQUIRE(x && x1 && wT && bT && yT && wF && bF && yF && out_row && B > 0, "resid_conv0_keep: bad argument");
  return resid_conv0(x, x1, wT, bT, yT, wF, bF, yF, 8L * 4096, B, ST(s), 0, out_row, out_col);
}
size_t lshm_recon_bwd5_workspace_floats(int B) { return recon_partials_floats(B * 4, 128) + recon_bwd5_workspace_floats(); }
int lshm_recon_bwd5(const float* x, const float* x1, const float* aT, const float* aF, const float* wT, const float* bT, const float* wF,
                    const float* bF, float* y1, float* y2, float* y3, float rho, int B, double* sums7, float* gx1p, float* daT, float* daF,
                    float* dwT, float* dbT, float* dwF, float* dbF, float* ws, size_t wsf, int bf, lshm_stream_t s) {
  REQUIRE(x && x1 && aT && aF && wT && bT && wF && bF && y1 && y2 && y3 && sums7 && gx1p && daT && daF && dwT && dwF && ws && B > 0,
          "recon_bwd5: bad argument");
  if (wsf < lshm_recon_bwd5_workspace_floats(B)) { set_last_error("recon_bwd5: workspace too small"); return LSHM_ERR_WORKSPACE; }
  if (!recon_bwd5_supported(4, 128, 8, 4, 4096)) { set_last_error("recon_bwd5: switched off by the schedule word"); return LSHM_ERR_UNSUPPORTED; }
  float* slabs = ws + recon_partials_floats(B * 4, 128);
  int rc = recon_bwd5(x, x1, aT, aF, 8L * 4096, wT, bT, wF, bF, y1, y2, y3, rho, B, gx1p, daT, daF, 8L * 4096, ws, slabs,
                      recon_bwd5_workspace_floats(), ST(s), 1.f, bf);
  if (rc) return rc;
  if ((rc = recon_sum7(ws, B * 4, 128, sums7, ST(s)))) return rc;
  return recon_bwd5_close(slabs, recon_bwd5_grid(B), dwT, dbT, dwF, dbF, 0, ST(s), nullptr);
}
int lshm_tconv5_pair_bwd(const float* gx2, const float* gx3c, const float* aT, const float* aF, const float* wT, const float* wF, float* daT,
                         float* daF, float* dwT, float* dbT, float* dwF, float* dbF, int B, float* ws, size_t wsf, int bf, lshm_stream_t s) {
  REQUIRE(gx2 && gx3c && aT && aF && wT && wF && daT && daF && dwT && dwF && ws && B > 0, "tconv5_pair_bwd: bad argument");
  if (!recon_bwd5_supported(4, 128, 8, 4, 4096)) { set_last_error("tconv5_pair_bwd: switched off by the schedule word"); return LSHM_ERR_UNSUPPORTED; }
  int rc = tconv5_pair_bwd(gx2, gx3c, aT, aF, 8L * 4096, wT, wF, B, daT, daF, 8L * 4096, ws, wsf, ST(s), bf);
  if (rc) return rc;
  return recon_bwd5_close(ws, recon_bwd5_grid(B), dwT, dbT, dwF, dbF, 0, ST(s), nullptr);
}
size_t lshm_conv0_bwd_tile_workspace_floats(void) { return conv0_bwd_tile_workspace_floats(); }
int lshm_conv0_bwd_tile(const float* r, const float* dzT, const float* dzF, const float* wT, const float* wF, const float* gx1p,
                        float* gx1, float* dwT, float* dbT, float* dwF, float* dbF, int B, float* ws, size_t wsf, int accumulate,
                        int bf, lshm_stream_t s) {
  REQUIRE(r && dzT && dzF && wT && wF && gx1p && gx1 && dwT && dwF && ws && B > 0, "conv0_bwd_tile: bad argument");
  return conv0_bwd_tile(r, dzT, dzF, 8L * 4096, wT, wF, gx1p, gx1, dwT, dbT, dwF, dbF, B, ws, wsf, accumulate, ST(s), nullptr, bf);
}
int lshm_plane_transpose(const float* in, float* out, int planes, int P, lshm_stream_t s) {
  REQUIRE(in && out && planes > 0, "plane_transpose: bad argument");
  return plane_transpose(in, out, planes, P, ST(s));
}
size_t lshm_rica_workspace_floats(int B, int L, int M) {
  return (B > 0 && L > 0 && M > 0) ? rica_workspace_floats(B, L, M) : 0;
}
int lshm_rica_loss_grad(const float* Xt, const float* A, const float* St, int B, int L, int M, float lambda1,
                        double* loss, float* dSt, float* ws, size_t wsf, lshm_stream_t s) {
  REQUIRE(Xt && A && St && loss && ws && B > 0 && L > 0 && M > 0, "rica_loss_grad: bad argument");
  return rica_loss_grad(Xt, A, St, B, L, M, lambda1, loss, dSt, ws, wsf, ST(s));
}
int lshm_rica_update_dictionary(const float* Xt, float* A, const float* St, int B, int L, int M, float eta,
                                double* dA_norm_sq, float* ws, size_t wsf, lshm_stream_t s) {
  REQUIRE(Xt && A && St && ws && B > 0 && L > 0 && M > 0, "rica_update_dictionary: bad argument");
  return rica_update_dictionary(Xt, A, St, B, L, M, eta, dA_norm_sq, ws, wsf, ST(s));
}
size_t lshm_recon_workspace_floats(int planes, int P) { return recon_partials_floats(planes, P); }
int lshm_recon_losses_fwd_bwd(const float* x, const float* x1, const float* x2, const float* x3c,
                              const float* y1, const float* y2, const float* y3, float rho, int planes,
                              int P, double* sums7, float* gx1p, float* gx2, float* gx3c, float* ws,
                              lshm_stream_t s) {
  REQUIRE(x && x1 && x2 && x3c && y1 && y2 && y3 && sums7 && ws && planes > 0, "recon_losses: null pointer");
  REQUIRE((gx1p && gx2 && gx3c) || (!gx1p && !gx2 && !gx3c), "recon_losses: gradient images are all set or all NULL");
  return recon_losses_fwd_bwd(x, x1, x2, x3c, y1, y2, y3, rho, planes, P, sums7, gx1p, gx2, gx3c, ws, ST(s));
}
int lshm_recon_losses_from_a(const float* x, const float* x1, const float* aT, const float* aF, const float* wT, const float* bT,
                             const float* wF, const float* bF, const float* y1, const float* y2, const float* y3, float rho,
                             int planes, int P, int C, double* sums7, float* gx1p, float* gx2, float* gx3c, float* ws,
                             lshm_stream_t s) {
  REQUIRE(x && x1 && aT && aF && wT && bT && wF && bF && y1 && y2 && y3 && sums7 && gx1p && gx2 && gx3c && ws && planes > 0 && C > 0,
          "recon_losses_from_a: null pointer");
  if (!recon_from_a_supported(C, P, 8, C, P * P / 4)) {
    set_last_error("recon_losses_from_a: needs ConvTranspose1d(8, C, 4, stride=4) output layers and a patch size that is a multiple of 32");
    return LSHM_ERR_UNSUPPORTED;
  }
  return recon_losses_from_a(x, x1, aT, aF, 8L * (P * P / 4), wT, bT, wF, bF, C, y1, y2, y3, rho, planes, P, sums7, gx1p, gx2, gx3c, ws,
                             ST(s));
}
int lshm_combine_dx1(const float* gx1p, const float* gT, const float* gFc, float* gx1, int planes, int P,
                     lshm_stream_t s) {
  REQUIRE(gx1p && gT && gFc && gx1 && planes > 0 && P % 32 == 0, "combine_dx1: bad argument");
  return combine_dx1(gx1p, gT, gFc, gx1, planes, P, ST(s));
}
int lshm_multiplier_update(const float* x, const float* x1, const float* x2, const float* x3c, float* y1,
                           float* y2, float* y3, float rho, int planes, int P, lshm_stream_t s) {
  REQUIRE(x && x1 && x2 && x3c && y1 && y2 && y3 && planes > 0, "multiplier_update: bad argument");
  return multiplier_update(x, x1, x2, x3c, y1, y2, y3, rho, planes, P, ST(s));
}

int lshm_adam_step_flat(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2,
                        float eps, const int* step_dev, int step_host, float gscale, lshm_stream_t s) {
  REQUIRE(p && g && m && v && n >= 0, "adam: bad argument");
  if (n == 0) return LSHM_OK;
  return adam_step_flat(p, g, m, v, n, lr, b1, b2, eps, step_dev, step_host, gscale, ST(s));
}
int lshm_axpy_flat(float* y, const float* x, float alpha, long n, lshm_stream_t s) {
  REQUIRE(y && x && n >= 0, "axpy: bad argument");
  if (n == 0) return LSHM_OK;
  return axpy_flat(y, x, alpha, n, ST(s));
}
int lshm_scale_flat(float* x, float alpha, long n, lshm_stream_t s) {
  REQUIRE(x && n >= 0, "scale: bad argument");
  if (n == 0) return LSHM_OK;
  return scale_flat(x, alpha, n, ST(s));
}
int lshm_dot_flat(const float* a, const float* b, long n, double* out, float* ws, lshm_stream_t s) {
  REQUIRE(a && b && out && ws && n >= 0, "dot: bad argument");
  return dot_flat(a, b, n, out, ws, ST(s));
}

int lshm_asum_flat(const float* a, long n, double* out, float* ws, lshm_stream_t s) {
  REQUIRE(a && out && ws && n >= 0, "asum: bad argument");
  return asum_flat(a, n, out, ws, ST(s));
}

size_t lshm_multi_dot_workspace_doubles(int count) { return count > 0 ? multi_dot_workspace_doubles(count) : 0; }
int lshm_multi_dot_flat(const float* const* a, const float* const* b, int count, long n, double* out, double* ws,
                        size_t wsd, lshm_stream_t s) {
  REQUIRE(a && b && out && ws && n >= 0 && count >= 1 && count <= kMaxDots, "multi_dot: bad argument");
  for (int i = 0; i < count; ++i) REQUIRE(a[i] && b[i], "multi_dot: null vector");
  if (wsd < multi_dot_workspace_doubles(count)) { set_last_error("multi_dot: workspace too small"); return LSHM_ERR_WORKSPACE; }
  return multi_dot_flat(a, b, count, n, out, ws, ST(s));
}
size_t lshm_lbfgs_direction_workspace_doubles(int m) { return m >= 0 ? lbfgs_direction_workspace_doubles(m) : 0; }
int lshm_lbfgs_direction(const float* const* y, const float* const* s_, int m, const float* grad, double h_diag, float* d,
                         long n, double* ws, size_t wsd, lshm_stream_t s) {
  REQUIRE(grad && d && ws && n >= 0 && m >= 0 && m <= kMaxDots && (m == 0 || (y && s_)), "lbfgs_direction: bad argument");
  for (int i = 0; i < m; ++i) REQUIRE(y[i] && s_[i] && y[i] != d && s_[i] != d, "lbfgs_direction: null or aliased vector");
  if (wsd < lbfgs_direction_workspace_doubles(m)) { set_last_error("lbfgs_direction: workspace too small"); return LSHM_ERR_WORKSPACE; }
  if (n == 0) return LSHM_OK;
  return lbfgs_direction(y, s_, m, grad, h_diag, d, n, ws, ST(s));
}

size_t lshm_patches_workspace_floats(void) { return patches_workspace_floats(); }
int lshm_patches_from_vis(const int8_t* vis, const float* scale, int nb, int ntime, int nfreq, int patch,
                          float clampv, int normalize, float* y, double* mean_std, float* ws, lshm_stream_t s) {
  REQUIRE(vis && scale && y && mean_std && ws && nb > 0 && ntime > 0 && nfreq > 0, "patches_from_vis: bad argument");
  REQUIRE(patch > 0 && patch % 2 == 0, "patches_from_vis: patch size must be even");
  return patches_from_vis(vis, scale, nb, ntime, nfreq, patch, 4, clampv, normalize, y, mean_std, nullptr, ws, ST(s));
}
int lshm_patches_from_vis_ex(const int8_t* vis, const float* scale, int nb, int ntime, int nfreq, int patch,
                             int num_channels, float clampv, int normalize, float* y, double* mean_std,
                             double* moments, float* ws, lshm_stream_t s) {
  REQUIRE(vis && scale && y && mean_std && ws && nb > 0 && ntime > 0 && nfreq > 0, "patches_from_vis: bad argument");
  REQUIRE(patch > 0 && patch % 2 == 0, "patches_from_vis: patch size must be even");
  REQUIRE(num_channels == 4 || num_channels == 8, "patches_from_vis: num_channels is 4 or 8 (src/lofar_tools.py:70)");
  return patches_from_vis(vis, scale, nb, ntime, nfreq, patch, num_channels, clampv, normalize, y, mean_std, moments,
                          ws, ST(s));
}
int lshm_patches_normalize(float* y, long n, const double* moments, lshm_stream_t s) {
  REQUIRE(y && moments && n >= 0, "patches_normalize: bad argument");
  if (n == 0) return LSHM_OK;
  return patches_normalize_moments(y, n, moments, ST(s));
}

int lshm_fft2_ortho_shift_cat_clamp(const float* x, float* out, int B, int C, float clampv, lshm_stream_t s) {
  REQUIRE(x && out && B > 0 && C > 0, "fft2: bad argument");
  return fft2_ortho_shift_cat_clamp(x, out, B, C, clampv, ST(s));
}

size_t lshm_fft2_backward_workspace_floats(int B, int C) { return (B > 0 && C > 0) ? fft2_backward_workspace_floats(B, C) : 0; }
int lshm_fft2_backward(const float* grad_out, const float* out, float* grad_x, int B, int C, float clampv, float* ws,
                       size_t wsf, lshm_stream_t s) {
  REQUIRE(grad_out && out && grad_x && ws && B > 0 && C > 0, "fft2_backward: bad argument");
  return fft2_feature_backward(grad_out, out, grad_x, B, C, clampv, ws, wsf, ST(s));
}

/* ---- `_bf16` forms of the GEMM-shaped entry points: same arguments, operands rounded to bf16 at LDS
 * staging (v_mfma_f32_16x16x16_bf16), fp32 accumulation and storage.  The precision is a property of the
 * call (scope of the calling thread), never of the process. */
int lshm_conv_fwd_bf16(int kind, const float* x, const float* w, const float* bias, float* y, int B, int Cin,
                       int Cout, int Hin, int Win, long in_bs, long out_bs, int act, float* ws, size_t wsf,
                       lshm_stream_t s) {
  MatrixPrecisionScope sc(1);
  return lshm_conv_fwd(kind, x, w, bias, y, B, Cin, Cout, Hin, Win, in_bs, out_bs, act, ws, wsf, s);
}
int lshm_conv_dgrad_bf16(int kind, const float* dz, const float* w, float* dx, const float* y_in_saved, int B,
                         int Cin, int Cout, int Hin, int Win, long in_bs, long out_bs, float* ws, size_t wsf,
                         lshm_stream_t s) {
  MatrixPrecisionScope sc(1);
  return lshm_conv_dgrad(kind, dz, w, dx, y_in_saved, B, Cin, Cout, Hin, Win, in_bs, out_bs, ws, wsf, s);
}
int lshm_conv_wgrad_bf16(int kind, const float* x, const float* dz, float* dw, float* db, int B, int Cin,
                         int Cout, int Hin, int Win, long in_bs, long out_bs, float* ws, size_t wsf,
                         int accumulate, lshm_stream_t s) {
  MatrixPrecisionScope sc(1);
  return lshm_conv_wgrad(kind, x, dz, dw, db, B, Cin, Cout, Hin, Win, in_bs, out_bs, ws, wsf, accumulate, s);
}
int lshm_linear_fwd_bf16(const float* x, long ldx, const float* w, const float* bias, float* y, long ldy,
                         int B, int K, int N, int act, float* ws, size_t wsf, lshm_stream_t s) {
  MatrixPrecisionScope sc(1);
  return lshm_linear_fwd(x, ldx, w, bias, y, ldy, B, K, N, act, ws, wsf, s);
}
int lshm_linear_dgrad_bf16(const float* dz, long lddz, const float* w, float* dx, long lddx,
                           const float* x_saved, long ldxs, int B, int K, int N, float* ws, size_t wsf,
                           lshm_stream_t s) {
  MatrixPrecisionScope sc(1);
  return lshm_linear_dgrad(dz, lddz, w, dx, lddx, x_saved, ldxs, B, K, N, ws, wsf, s);
}
int lshm_linear_wgrad_bf16(const float* x, long ldx, const float* dz, long lddz, float* dw, float* db, int B,
                           int K, int N, float* ws, size_t wsf, lshm_stream_t s) {
  MatrixPrecisionScope sc(1);
  return lshm_linear_wgrad(x, ldx, dz, lddz, dw, db, B, K, N, ws, wsf, s);
}
int lshm_rica_loss_grad_bf16(const float* Xt, const float* A, const float* St, int B, int L, int M, float lambda1,
                             double* loss, float* dSt, float* ws, size_t wsf, lshm_stream_t s) {
  MatrixPrecisionScope sc(1);
  return lshm_rica_loss_grad(Xt, A, St, B, L, M, lambda1, loss, dSt, ws, wsf, s);
}
int lshm_rica_update_dictionary_bf16(const float* Xt, float* A, const float* St, int B, int L, int M, float eta,
                                     double* dA_norm_sq, float* ws, size_t wsf, lshm_stream_t s) {
  MatrixPrecisionScope sc(1);
  return lshm_rica_update_dictionary(Xt, A, St, B, L, M, eta, dA_norm_sq, ws, wsf, s);
}

}  // extern "C"
