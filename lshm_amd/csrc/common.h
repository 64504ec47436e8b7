// Shared device/host helpers for the LSHM gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <tuple>
#include <utility>

#define LSHM_OK 0
#define LSHM_ERR_ARG (-1)        // bad argument (null pointer, unsupported size)
#define LSHM_ERR_WORKSPACE (-2)  // workspace too small
#define LSHM_ERR_UNSUPPORTED (-3)

namespace lshm {

void set_last_error(const char* msg);
int check_launch(const char* what);  // hipGetLastError -> code + message
// LSHM_OK if `kernel` can be launched on the current device with `threads` per workgroup and `dyn_lds` bytes of
// dynamic LDS on top of its static allocation; LSHM_ERR_UNSUPPORTED (with a message) if its LDS or register
// budget does not fit the device -- asked once per (kernel, device) and remembered.  A tile configuration or
// kernel variant that cannot launch is thus a return code at the call site, never a failed or aborted dispatch.
int kernel_budget_ok(const void* kernel, int threads, size_t dyn_lds, const char* what);
int device_lds_bytes();  // LDS a workgroup may use on the current device (0: unknown)
// Raises the dynamic-LDS limit of `kernel` to `dyn_lds` bytes on the current device -- once per (kernel, device), remembered:
// hipFuncSetAttribute on every launch is host time on the enqueue path.  LSHM_ERR_UNSUPPORTED (with a message) if refused.
int raise_dynamic_lds(const void* kernel, size_t dyn_lds, const char* what);
// true if a workgroup of the current device may hold `bytes` of LDS (unknown device: true, the launch reports it)
inline bool device_lds_fits(size_t bytes) { const int v = device_lds_bytes(); return v <= 0 || bytes <= (size_t)v; }

// A kernel's own completion signal as an event.  hipEventRecord puts a marker packet behind the last kernel of a stream,
// and the NEXT kernel of that stream waits for the marker to retire: ~6 us of idle queue per record on the stream that
// records (profiles/r03/step_timeline.txt: every launch behind a "dz ready" record has gap = 6).  While
// `launch_stop_event` is set (StopEventScope), every launch of this thread goes through hipExtLaunchKernel with that
// event as its stop event instead: the event completes with the kernel, no marker.  Kernels of one stream complete in
// order, so after a scope the event stands for "everything the scope launched on that stream".
extern thread_local hipEvent_t launch_stop_event;
extern thread_local unsigned launch_stop_count;  // launches that carried a stop event (a scope that launched nothing must not be waited for)
struct StopEventScope {
  explicit StopEventScope(hipEvent_t ev) { launch_stop_event = ev; }
  ~StopEventScope() { launch_stop_event = nullptr; }
  StopEventScope(const StopEventScope&) = delete;
  StopEventScope& operator=(const StopEventScope&) = delete;
};
template <typename... KArgs, typename Tuple, size_t... I>
inline void launch_with_stop_event(void (*kernel)(KArgs...), dim3 g, dim3 b, unsigned lds, hipStream_t st, hipEvent_t ev,
                                   Tuple& t, std::index_sequence<I...>, hipEvent_t start = nullptr) {
  void* ptrs[sizeof...(KArgs) ? sizeof...(KArgs) : 1] = {static_cast<void*>(&std::get<I>(t))...};
  (void)hipExtLaunchKernel(reinterpret_cast<const void*>(kernel), g, b, ptrs, lds, st, start, ev, 0);
}
// Diagnostic per-launch trace WITHOUT a profiler (lshm_trace_begin / _end / _read, include/lshm.h): while a thread records,
// every launch it makes goes through hipExtLaunchKernel with a start and a stop event of the trace's own -- the kernel's
// dispatch and completion timestamps, no marker packets in the queue.  A launch inside a StopEventScope additionally records
// the scope's event behind the kernel (a marker: ~5 us of idle queue on that stream, ~25 times per iteration).
struct LaunchTraceSlot { hipEvent_t start, stop; };
extern thread_local bool launch_trace_on;
bool launch_trace_take(const void* kernel, dim3 g, dim3 b, hipStream_t st, LaunchTraceSlot* slot);  // false: trace full
template <typename... KArgs, typename... Args>
inline void launch(void (*kernel)(KArgs...), dim3 g, dim3 b, unsigned lds, hipStream_t st, Args&&... args) {
  static_assert(sizeof...(KArgs) == sizeof...(Args), "kernel argument count");
  LaunchTraceSlot slot;
  if (launch_trace_on && launch_trace_take(reinterpret_cast<const void*>(kernel), g, b, st, &slot)) {
    std::tuple<KArgs...> t{static_cast<KArgs>(args)...};
    launch_with_stop_event(kernel, g, b, lds, st, slot.stop, t, std::index_sequence_for<KArgs...>{}, slot.start);
    if (hipEvent_t ev = launch_stop_event) { (void)hipEventRecord(ev, st); ++launch_stop_count; }
    return;
  }
  if (hipEvent_t ev = launch_stop_event) {
    std::tuple<KArgs...> t{static_cast<KArgs>(args)...};  // the kernel's own parameter types, as <<<>>> would convert
    launch_with_stop_event(kernel, g, b, lds, st, ev, t, std::index_sequence_for<KArgs...>{});
    ++launch_stop_count;
  } else {
    kernel<<<g, b, lds, st>>>(static_cast<KArgs>(args)...);
  }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Element types of the tensors in HBM.  Everything is computed in fp32 registers; with bf16 STORAGE
// (LSHM_PRECISION_BF16_STORAGE, BASELINE.json configs[2]) the image-sized activations and gradients of the
// bandwidth-bound outer layers and of the glue passes are kept as bf16 (round to nearest even on the way out:
// v_cvt_pk_bf16_f32; a shift on the way in).  Kernels that touch such a tensor are templated on its type.
typedef __bf16 bf16;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
template <class T>
struct Elem;
template <>
struct Elem<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ float ld_nt(const float* p) { return __builtin_nontemporal_load(p); }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
  static __device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
  static __device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
};
template <>
struct Elem<bf16> {
  static __device__ __forceinline__ float ld(const bf16* p) { return (float)*p; }
  static __device__ __forceinline__ float ld_nt(const bf16* p) { return (float)__builtin_nontemporal_load(p); }
  static __device__ __forceinline__ void st(bf16* p, float v) { *p = (bf16)v; }
  static __device__ __forceinline__ f32x4 ld4(const bf16* p) {
    return __builtin_convertvector(*reinterpret_cast<const bf16x4*>(p), f32x4);
  }
  static __device__ __forceinline__ void st4(bf16* p, f32x4 v) {
    *reinterpret_cast<bf16x4*>(p) = __builtin_convertvector(v, bf16x4);
  }
};

// exp(v) - 1 for v <= 0 (the caller discards the value for v > 0).  Every activation of the path goes
// through this, and in the bandwidth-heavy layers it is most of the vector work, so it is written out:
// |v| < 1/4: six Taylor terms (truncation < 1.3e-8 relative); otherwise the hardware exp2 minus one,
// where the subtraction no longer cancels (relative error <= 3e-7 at v = -1/4, shrinking below).
// 11 instructions instead of the 22 of the library expm1f; agrees with it to <= 3e-7 relative.
__device__ __forceinline__ float expm1_nonpos(float v) {
  const float e = __builtin_amdgcn_exp2f(v * 1.44269504088896340736f);
  float p = fmaf(v, 1.f / 720.f, 1.f / 120.f);
  p = fmaf(v, p, 1.f / 24.f);
  p = fmaf(v, p, 1.f / 6.f);
  p = fmaf(v, p, 0.5f);
  p = fmaf(v, p, 1.f);
  return v > -0.25f ? p * v : e - 1.f;
}
__device__ __forceinline__ float elu(float v) { return v > 0.f ? v : expm1_nonpos(v); }
// derivative of ELU(alpha=1) from the saved *output*: 1 if y>0 else y+1
__device__ __forceinline__ float elu_grad_from_out(float y) { return y > 0.f ? 1.f : y + 1.f; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block-wide sum (blockDim.x multiple of 64, <= 1024); result valid in thread 0.
template <typename T>
__device__ __forceinline__ T block_sum(T v, T* smem /* >= 16 */) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if (lane == 0) smem[w] = v;
  __syncthreads();
  T r = 0;
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) r += smem[i];
  }
  __syncthreads();
  return r;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace lshm

// every launch of the library goes through lshm::launch (see launch_stop_event above)
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) ::lshm::launch(kernel, dim3(grid), dim3(block), lds, stream, ##__VA_ARGS__)
