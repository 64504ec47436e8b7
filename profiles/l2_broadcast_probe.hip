// Round 4, VERDICT item 1(a): how fast can EVERY workgroup of a launch stream the SAME weights out of L2?
// DESIGN.md section 8 (round 3) rejected per-patch chains through the deep layers with "302 MB of L2 traffic per layer";
// this measures that traffic instead of arguing about it.  Each workgroup reads the same `bytes` (1.18 MB = conv5 /
// tconv0 weights, 4.2-4.9 MB = all deep-layer weights of AutoEncoderCNN2) as coalesced float4s (a wavefront reads 1 KiB
// runs, wavefronts interleaved), `INFL` loads in flight per lane, into registers (summed so that nothing is dead) or
// through LDS (ds_write_b128 of every float4).
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/l2probe profiles/l2_broadcast_probe.hip && /tmp/l2probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int THREADS, int INFL, bool TO_LDS>
__global__ __launch_bounds__(THREADS) void stream_kernel(const f32x4* __restrict__ w, long n4, float* __restrict__ out) {
  __shared__ f32x4 stage[TO_LDS ? THREADS * INFL : 1];
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int t = threadIdx.x;
  for (long i = t; i < n4; i += (long)THREADS * INFL) {
    f32x4 v[INFL];
#pragma unroll
    for (int k = 0; k < INFL; ++k) {
      const long j = i + (long)k * THREADS;
      v[k] = j < n4 ? w[j] : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int k = 0; k < INFL; ++k) {
      if (TO_LDS) stage[k * THREADS + t] = v[k];
      acc += v[k];
    }
  }
  if (TO_LDS) {
    __syncthreads();
    acc += stage[(t * 7) % (THREADS * INFL)];
  }
  out[(long)blockIdx.x * THREADS + t] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int THREADS, int INFL, bool TO_LDS>
static void run(const char* name, const f32x4* w, long bytes, int grid, float* out, char* flush, size_t flush_bytes) {
  const long n4 = bytes / 16;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f, cold = 0.f;
  for (int rep = 0; rep < 6; ++rep) {
    if (rep == 0) hipMemsetAsync(flush, rep, flush_bytes, 0);  // first repetition: weights in HBM / Infinity Cache only
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((stream_kernel<THREADS, INFL, TO_LDS>), dim3(grid), dim3(THREADS), 0, 0, w, n4, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep == 0) cold = ms; else if (ms < best) best = ms;
  }
  const double tot = (double)bytes * grid;
  printf("%-28s %4d workgroups x %7.2f MB: %7.1f us warm (%5.2f TB/s, %5.1f GB/s per workgroup), %7.1f us after a 1 GiB memset\n", name, grid,
         bytes / 1e6, best * 1e3, tot / best / 1e9, bytes / best / 1e6, cold * 1e3);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
}

int main() {
  const size_t wbytes = 8u << 20;
  f32x4* w;
  float* out;
  char* flush;
  const size_t flush_bytes = 1u << 30;
  hipMalloc(&w, wbytes);
  hipMalloc(&out, 1024 * 1024 * sizeof(float));
  hipMalloc(&flush, flush_bytes);
  hipMemset(w, 0, wbytes);
  const long sizes[3] = {1179648, 4194304, 4915200};  // conv5 | "the same 4.2 MB" | every deep weight of the 2-D autoencoder
  for (long b : sizes) {
    for (int grid : {128, 256, 512}) {
      run<1024, 4, false>("regs 1024 thr, 4 in flight", w, b, grid, out, flush, flush_bytes);
      run<1024, 8, false>("regs 1024 thr, 8 in flight", w, b, grid, out, flush, flush_bytes);
      run<512, 8, false>("regs 512 thr, 8 in flight", w, b, grid, out, flush, flush_bytes);
      run<256, 8, false>("regs 256 thr, 8 in flight", w, b, grid, out, flush, flush_bytes);
      run<1024, 4, true>("LDS 1024 thr, 4 in flight", w, b, grid, out, flush, flush_bytes);
    }
  }
  return 0;
}
