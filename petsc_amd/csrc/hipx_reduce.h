// hipx_reduce.h -- wave64 / workgroup / grid reduction building blocks shared by the Vec and Mat kernels.
#pragma once
#include "hipx_internal.h"

namespace hipx {

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_max_nan(double v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    double o = __shfl_down(v, off, 64);
    v        = (o > v || o != o) ? o : v;
  }
  return v;
}

enum { RED_SUM = 0, RED_MAXNAN = 1 };

// Publishes this block's NV partials and lets the last-arriving block produce the NV results.
// Inter-workgroup hand-off per the CDNA4 rules (MI355X_MICROARCH "Workgroup dispatch ..."): plain stores ->
// __syncthreads -> one-lane agent-scope release + drained vmcnt -> relaxed ticket; consumer: agent acquire ->
// __syncthreads -> plain loads.
template <int NV, int OP>
__device__ __forceinline__ void block_finish(double (&acc)[NV], RedOut out)
{
  double *partials = out.partials, *results = out.results;
  unsigned int *ticket = out.ticket;
  __shared__ double   s_w[NV][kRedThreads / 64];
  __shared__ unsigned s_last;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int v = 0; v < NV; v++) {
    double r = (OP == RED_SUM) ? wave_sum(acc[v]) : wave_max_nan(acc[v]);
    if (lane == 0) s_w[v][wid] = r;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    const int v = threadIdx.x;
    double    r = s_w[v][0];
#pragma unroll
    for (int w = 1; w < kRedThreads / 64; w++) {
      if (OP == RED_SUM) r += s_w[v][w];
      else r = (s_w[v][w] > r || s_w[v][w] != s_w[v][w]) ? s_w[v][w] : r;
    }
    partials[(size_t)v * kRedBlocks + blockIdx.x] = r;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last     = (t == gridDim.x - 1);
    if (s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  if (!s_last) return;
  // final fold: fixed order -- thread t takes partials t, t+256, ... then the same wave/LDS tree
#pragma unroll
  for (int v = 0; v < NV; v++) {
    double r = (OP == RED_SUM) ? 0.0 : -1.0;
    for (int b = threadIdx.x; b < (int)gridDim.x; b += kRedThreads) {
      double p = __hip_atomic_load(&partials[(size_t)v * kRedBlocks + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (OP == RED_SUM) r += p;
      else r = (p > r || p != p) ? p : r;
    }
    r = (OP == RED_SUM) ? wave_sum(r) : wave_max_nan(r);
    __syncthreads();
    if (lane == 0) s_w[v][wid] = r;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    const int v = threadIdx.x;
    double    r = s_w[v][0];
#pragma unroll
    for (int w = 1; w < kRedThreads / 64; w++) {
      if (OP == RED_SUM) r += s_w[v][w];
      else r = (s_w[v][w] > r || s_w[v][w] != s_w[v][w]) ? s_w[v][w] : r;
    }
    results[v] = r;  // pinned, fine-grained host memory
    if (out.dres) out.dres[v] = r;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    *ticket = 0u;  // re-arm for the next launch on this slot (stream-ordered)
    __threadfence_system();  // results before flag, visible to the host
    if (out.seq) __hip_atomic_store(out.flag, out.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}


}  // namespace hipx
