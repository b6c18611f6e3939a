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

// ---- compensated accumulation (opt-in "exact" reduction mode, hipxSetReductionMode): every sum is carried as an unevaluated pair
// (hi, lo) -- TwoProduct through fma, TwoSum for the running sum, the rounding errors collected in lo -- through the thread loop, the
// wave shuffle tree, the LDS fold, the last workgroup's fold of the 256 partials and, on several ranks, the all-reduce; hi + lo is
// formed ONCE at the very end.  This is Dot2 / Sum2 of Ogita, Rump & Oishi (SIAM J. Sci. Comput. 26(6), 2005): the result is the dot
// product evaluated in twice the working precision and rounded once, whatever the association of the partial sums -- so it equals
// what the reference computes under oracle/exactblas.c (its ddot/dnrm2/dasum/dgemv: bvec1.c:27, bvec2.c:202-223, dvec2.c:557) to the
// last bit unless the exact value lies within ~n eps^2 of a rounding boundary, independently of grid shape and rank count.
// ~25 flops per 16 bytes: free in kernels that wait for HBM.
template <bool COMP>
struct Acc;
template <>
struct Acc<false> {
  double s = 0.0;
  __device__ __forceinline__ void prod(double a, double b) { s += a * b; }
  __device__ __forceinline__ void add(double v) { s += v; }
};
template <>
struct Acc<true> {
  double s = 0.0, c = 0.0;  // hi, lo
  __device__ __forceinline__ void add(double v)
  {
    const double t = s + v, z = t - s;
    c += (s - (t - z)) + (v - z);  // TwoSum error term
    s = t;
  }
  __device__ __forceinline__ void prod(double a, double b)
  {
    const double h = a * b;
    const double r = __builtin_fma(a, b, -h);  // a*b = h + r exactly
    const double t = s + h, z = t - s;
    const double q = (s - (t - z)) + (h - z);
    s = t;
    c += q + r;
  }
  __device__ __forceinline__ void merge(double ohi, double olo)
  {
    const double t = s + ohi, z = t - s;
    const double q = (s - (t - z)) + (ohi - z);
    s = t;
    c += q + olo;
  }
};
__device__ __forceinline__ void wave_sum_dd(Acc<true> &a)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double ohi = __shfl_down(a.s, off, 64), olo = __shfl_down(a.c, off, 64);
    a.merge(ohi, olo);
  }
}

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

// block_finish for compensated sums: 2 NV partial rows per workgroup (hi rows 0..NV-1, lo rows NV..2NV-1; NV <= kMaxRedVals / 2), the
// same hand-off; the last workgroup folds pairs.  out.pairs: leave (hi, lo) unrounded in results[2v], results[2v+1] for a fold over
// ranks that follows on the stream (hipx_comm.hip); otherwise results[v] = hi + lo.
template <int NV>
__device__ __forceinline__ void block_finish_dd(Acc<true> (&acc)[NV], RedOut out)
{
  static_assert(2 * NV <= kMaxRedVals, "compensated reductions: at most kMaxRedVals / 2 sums per launch");
  double *partials = out.partials, *results = out.results;
  unsigned int *ticket = out.ticket;
  __shared__ double   s_hi[NV][kRedThreads / 64], s_lo[NV][kRedThreads / 64];
  __shared__ unsigned s_last;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int v = 0; v < NV; v++) {
    wave_sum_dd(acc[v]);
    if (lane == 0) {
      s_hi[v][wid] = acc[v].s;
      s_lo[v][wid] = acc[v].c;
    }
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    const int v = threadIdx.x;
    Acc<true> r;
    r.s = s_hi[v][0];
    r.c = s_lo[v][0];
#pragma unroll
    for (int w = 1; w < kRedThreads / 64; w++) r.merge(s_hi[v][w], s_lo[v][w]);
    partials[(size_t)v * kRedBlocks + blockIdx.x]        = r.s;
    partials[(size_t)(NV + v) * kRedBlocks + blockIdx.x] = r.c;
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
#pragma unroll
  for (int v = 0; v < NV; v++) {
    Acc<true> r;
    for (int b = threadIdx.x; b < (int)gridDim.x; b += kRedThreads) {
      const double phi = __hip_atomic_load(&partials[(size_t)v * kRedBlocks + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const double plo = __hip_atomic_load(&partials[(size_t)(NV + v) * kRedBlocks + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      r.merge(phi, plo);
    }
    wave_sum_dd(r);
    __syncthreads();
    if (lane == 0) {
      s_hi[v][wid] = r.s;
      s_lo[v][wid] = r.c;
    }
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    const int v = threadIdx.x;
    Acc<true> r;
    r.s = s_hi[v][0];
    r.c = s_lo[v][0];
#pragma unroll
    for (int w = 1; w < kRedThreads / 64; w++) r.merge(s_hi[v][w], s_lo[v][w]);
    if (out.pairs) {
      results[2 * v]     = r.s;
      results[2 * v + 1] = r.c;
    } else {
      const double f = r.s + r.c;  // the one rounding
      results[v]     = f;
      if (out.dres) out.dres[v] = f;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    *ticket = 0u;
    __threadfence_system();
    if (out.seq) __hip_atomic_store(out.flag, out.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// the end of every sum-reduction kernel: plain or compensated accumulators
template <int NV, bool COMP>
__device__ __forceinline__ void finish_sums(Acc<COMP> (&acc)[NV], RedOut out)
{
  if constexpr (COMP) block_finish_dd<NV>(acc, out);
  else {
    double a[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) a[v] = acc[v].s;
    block_finish<NV, RED_SUM>(a, out);
  }
}

}  // namespace hipx
