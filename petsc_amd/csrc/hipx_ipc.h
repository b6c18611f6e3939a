// hipx_ipc.h -- device-side pieces of the IPC transport shared by hipx_comm.hip (put / all-reduce kernels) and hipx_mat.hip (the off-diagonal kernel that
// waits for its ghost values itself).  Hand-off form (MI355X_MICROARCH "valid forms"): payload as WRITE-THROUGH stores (sc0 sc1: system scope -- the reader is
// another process, possibly another GPU over xGMI), the issuing wave drains them (s_waitcnt vmcnt(0)), a ticket collects the workgroups, ONE relaxed
// system-scope store raises the sequence flag; the reader polls the flag with relaxed system-scope loads and reads the payload with sc0 sc1 loads (or behind
// a kernel boundary).  No release / acquire fences: a system-scope fence writes back / invalidates the XCD's whole L2, which the product kernel running
// beside it is busy filling (round 6: the fences of the first put kernel slowed a concurrent 2 M-row product from 9 to 60 us).
#pragma once
#include "hipx_internal.h"

namespace hipx {

typedef double ipc_dbl2 __attribute__((ext_vector_type(2)));

// 16 bytes, written through to memory (system scope); p must be 16-byte aligned.  The compiler does not count inline-asm stores: callers drain with ipc_drain().
__device__ __forceinline__ void ipc_store16(double *p, double a, double b)
{
  ipc_dbl2 v;
  v.x = a;
  v.y = b;
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void ipc_store8(double *p, double a)
{
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ double ipc_load8(const double *p)
{
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
}
__device__ __forceinline__ void ipc_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// four 16-byte system-scope loads (never a stale cache line) in flight together, then the wait -- ONE asm statement: the compiler does not count loads
// issued by inline asm, so nothing may touch the destination registers between issue and wait.  Every pointer 16-byte aligned and valid.
__device__ __forceinline__ void ipc_load16x4(const double *p0, const double *p1, const double *p2, const double *p3, ipc_dbl2 &g0, ipc_dbl2 &g1, ipc_dbl2 &g2, ipc_dbl2 &g3)
{
  asm volatile("global_load_dwordx4 %0, %4, off sc0 sc1\n\tglobal_load_dwordx4 %1, %5, off sc0 sc1\n\tglobal_load_dwordx4 %2, %6, off sc0 sc1\n\tglobal_load_dwordx4 %3, %7, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
               : "=&v"(g0), "=&v"(g1), "=&v"(g2), "=&v"(g3)
               : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
               : "memory");
}

// spin until *flag >= want; gives up after `limit` ticks of the 100 MHz wall clock (or when another waiter has given up): *err <- 1, returns false
__device__ __forceinline__ bool ipc_wait_ge(const unsigned long long *flag, unsigned long long want, unsigned int *err, long long limit)
{
  long long t0 = 0;
  for (unsigned spins = 1;; spins++) {
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= want) return true;
    __builtin_amdgcn_s_sleep(4);
    if ((spins & 0x3ff) == 0) {
      const long long now = (long long)wall_clock64();
      if (!t0) t0 = now;
      if (now - t0 > limit || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) {
        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return false;
      }
    }
  }
}

// what a consumer kernel waits for before it touches the ghost values of an exchange (n = 0: nothing -- RCCL transport, or a wait kernel ran before it)
struct IpcWait {
  const unsigned long long *flag[4] = {nullptr, nullptr, nullptr, nullptr};
  int                       n      = 0;
  unsigned long long        want   = 0;
  unsigned int             *err    = nullptr;
  long long                 limit  = 0;
};

}  // namespace hipx
