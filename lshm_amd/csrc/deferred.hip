// Deferred reductions of a backward pass: every split-K combine, per-workgroup partial sum and bias
// gradient that the weight-gradient launches of one autoencoder leave behind is finished here by
// two multi-job launches instead of ~60 tiny ones (each of which costs a few microseconds of
// dispatch whatever its size).  Summation order inside a job is fixed: results are bitwise
// reproducible and independent of what else runs on the device.
#include <cstdio>
#include <cstdlib>

#include "kernels.h"

namespace lshm {

constexpr int kJobsPerLaunch = 40;  // keeps the by-value tables under the 4 KiB kernel-argument limit

struct ChanTable {
  ChanJob job[kJobsPerLaunch];
  int blk0[kJobsPerLaunch + 1];
  int njobs;
};
struct SumTable {
  SumJob job[kJobsPerLaunch];
  int blk0[kJobsPerLaunch + 1];
  int njobs;
};

template <class T>
__device__ __forceinline__ int find_job(const T& tab, int blk) {
  int j = 0;
  while (j + 1 < tab.njobs && blk >= tab.blk0[j + 1]) ++j;
  return j;
}

// stage 1 of the bias gradients: workgroup = (job, channel, chunk)
__global__ __launch_bounds__(256) void chan_partials_multi_kernel(const ChanTable tab) {
  __shared__ float red[16];
  const int jx = find_job(tab, blockIdx.x);
  const ChanJob& J = tab.job[jx];
  const int lb = blockIdx.x - tab.blk0[jx];
  const int c = lb / J.chunks, ch = lb - c * J.chunks;
  const long total4 = (long)J.B * J.HW / 4;
  const long len4 = (total4 + J.chunks - 1) / J.chunks;
  const long beg = ch * len4;
  const long end = beg + len4 < total4 ? beg + len4 : total4;
  const float* base = J.dz + (long)c * J.HW;
  float acc = 0.f;
  for (long i = beg + threadIdx.x; i < end; i += 256) {
    const long e = i * 4;
    const long b = e / J.HW, pos = e - b * J.HW;
    const float4 v = *reinterpret_cast<const float4*>(base + b * J.bs + pos);
    acc += (v.x + v.y) + (v.z + v.w);
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) J.partial[(long)ch * J.C + c] = acc;
}

// stage 2: every remaining sum.  A workgroup covers 256/lpo consecutive outputs (lanes of a wave read
// consecutive addresses); lpo groups of threads split the S partials and are combined through LDS
// in group order.
__global__ __launch_bounds__(256) void sum_jobs_multi_kernel(const SumTable tab) {
  __shared__ float red[256];
  // Consecutive workgroup ids are dealt round-robin to the 8 XCDs, each with its own L2.  With many lanes per
  // output a workgroup uses 16 bytes of every 128-byte line it touches; its neighbours use the rest, so they
  // must sit behind the same L2 or every XCD fetches the whole line again (measured: 3x the slab bytes from
  // HBM).  Give XCD x the x-th contiguous eighth of the blocks.
  const int per_xcd = (tab.blk0[tab.njobs] + 7) >> 3;
  const int blk = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (blk >= tab.blk0[tab.njobs]) return;
  const int jx = find_job(tab, blk);
  const SumJob& J = tab.job[jx];
  const int lb = blk - tab.blk0[jx];
  const int lpo = J.lpo;
  const int opb = 256 / lpo;
  const int sl = threadIdx.x / opb, ol = threadIdx.x - sl * opb;
  const long j = (long)lb * opb + ol;
  float acc = 0.f;
  if (j < J.n) {
    const float* src = J.src + j;
    const long step = (long)lpo * J.stride;
    int s = sl;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const float* q = src + (long)s * J.stride;
    for (; s + 3 * lpo < J.S; s += 4 * lpo, q += 4 * step) {
      a0 += q[0];
      a1 += q[step];
      a2 += q[2 * step];
      a3 += q[3 * step];
    }
    for (; s < J.S; s += lpo, q += step) a0 += q[0];
    acc = (a0 + a1) + (a2 + a3);
  }
  if (lpo > 1) {
    red[threadIdx.x] = acc;
    __syncthreads();
    if (sl == 0)
      for (int g = 1; g < lpo; ++g) acc += red[g * opb + ol];
  }
  if (sl == 0 && j < J.n) {
    long o = j;
    if (J.Mp) {
      const int col = (int)(j / J.Mp), row = (int)(j - (long)col * J.Mp);
      if (row >= J.M) return;
      o = (long)row * J.rs + (long)col * J.cs;
    }
    J.dst[o] = J.accumulate ? J.dst[o] + acc : acc;
  }
}

bool GradJobs::add_channel_sum(const float* dz, long bs, int B, int C, long HW, float* db, int accumulate) {
  if (HW % 4 != 0 || bs % 4 != 0 || (reinterpret_cast<uintptr_t>(dz) & 15)) return false;
  const long per_channel = (long)B * HW;
  int chunks = (int)((per_channel + 8191) / 8192);
  if (chunks < 1) chunks = 1;
  if (chunks > 256) chunks = 256;
  float* partial = take((size_t)chunks * C);
  if (!partial) return false;
  chan.push_back(ChanJob{dz, partial, bs, HW, B, C, chunks});
  sums.push_back(SumJob{partial, db, C, C, chunks, 0, 0, 0, 0, accumulate, 0});
  return true;
}

int grad_jobs_finish(GradJobs& jobs, hipStream_t st) {
  for (size_t j0 = 0; j0 < jobs.chan.size(); j0 += kJobsPerLaunch) {
    ChanTable tab;
    tab.njobs = (int)std::min<size_t>(kJobsPerLaunch, jobs.chan.size() - j0);
    int blk = 0;
    for (int j = 0; j < tab.njobs; ++j) {
      tab.job[j] = jobs.chan[j0 + j];
      tab.blk0[j] = blk;
      blk += tab.job[j].C * tab.job[j].chunks;
    }
    tab.blk0[tab.njobs] = blk;
    if (blk == 0) continue;
    hipLaunchKernelGGL(chan_partials_multi_kernel, dim3(blk), dim3(256), 0, st, tab);
    int rc = check_launch("chan_partials_multi");
    if (rc) return rc;
  }
  for (size_t j0 = 0; j0 < jobs.sums.size(); j0 += kJobsPerLaunch) {
    SumTable tab;
    tab.njobs = (int)std::min<size_t>(kJobsPerLaunch, jobs.sums.size() - j0);
    int blk = 0;
    for (int j = 0; j < tab.njobs; ++j) {
      SumJob& J = tab.job[j];
      J = jobs.sums[j0 + j];
      // aim for >= 128K threads per job: few outputs -> more thread groups share the S partials
      int lpo = 1;
      while (lpo < 64 && (long)J.n * lpo < 131072 && 4 * lpo <= J.S) lpo *= 2;
      J.lpo = lpo;
      if (getenv("LSHM_JOBS_LOG"))
        fprintf(stderr, "[lshm job] n=%d S=%d stride=%ld slab=%d lpo=%d blocks=%d MB=%.2f\n", J.n, J.S, J.stride, J.Mp,
                lpo, cdiv(J.n, 256 / lpo), 4e-6 * J.n * J.S);
      tab.blk0[j] = blk;
      blk += cdiv(J.n, 256 / J.lpo);
    }
    tab.blk0[tab.njobs] = blk;
    if (blk == 0) continue;
    hipLaunchKernelGGL(sum_jobs_multi_kernel, dim3((blk + 7) / 8 * 8), dim3(256), 0, st, tab);
    int rc = check_launch("sum_jobs_multi");
    if (rc) return rc;
  }
  jobs.chan.clear();
  jobs.sums.clear();
  return LSHM_OK;
}

}  // namespace lshm
