// hipx_internal.h -- shared state of libhipx.so (one process = one GPU).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include "hipx.h"

namespace hipx {

constexpr int kRedBlocks     = 512;   // most workgroups a reduction launches (the stride of its partials).  The grid is a function of n
                                        // only (red_grid(): one 1024-thread workgroup per CU by default, HIPX_RED_BLOCKS = 512: two) -> fixed
                                        // summation order, and few arrivals on the ticket counter (one contended word takes ~12 ns per atomic)
constexpr int kRedThreads    = 1024;
constexpr int kMaxRedVals    = 32;    // sums produced by one reduction launch (MDot batches)
constexpr int kEwMaxBlocks   = 4096;  // grid cap of the grid-stride elementwise kernels
constexpr int kEwThreads     = 256;

struct Runtime {
  bool        initialized = false;
  int         device      = -1;
  hipStream_t compute     = nullptr;
  hipStream_t comm        = nullptr;
  // reduction scratch: per slot kMaxRedVals x kRedBlocks partials, a ticket counter and a pinned result line
  double       *d_partials = nullptr;  // [HIPX_MAX_RED_SLOTS][kMaxRedVals][kRedBlocks]
  unsigned int *d_tickets  = nullptr;  // [HIPX_MAX_RED_SLOTS]
  double       *h_results  = nullptr;  // pinned, mapped: [HIPX_MAX_RED_SLOTS][kMaxRedVals]
  double       *d_results  = nullptr;  // device alias of h_results
  unsigned long long *h_flags = nullptr;  // pinned, mapped: completion sequence number per slot (host polls it)
  unsigned long long *d_flags = nullptr;
  unsigned long long  seq[HIPX_MAX_RED_SLOTS] = {0};
  double       *d_scalars  = nullptr;  // staging for kernel arguments that exceed the arg buffer (MAXPY alphas, pointer tables)
  void        **d_ptrs     = nullptr;
  int           next_slot  = 0;
  int           red_exact  = 0;        // hipxSetReductionMode: 1 = compensated (Dot2 / Sum2) sums in every reduction kernel
  char          errmsg[512] = "no error";
};

Runtime &rt();
int      fail(int code, const char *what, const char *file, int line);

inline double *slot_partials(int slot) { return rt().d_partials + (size_t)slot * kMaxRedVals * kRedBlocks; }
inline double *slot_results_dev(int slot) { return rt().d_results + (size_t)slot * kMaxRedVals; }
inline double *slot_results_host(int slot) { return rt().h_results + (size_t)slot * kMaxRedVals; }

// where a reduction launch puts its partials / results and how it signals completion
struct RedOut {
  double             *partials;
  unsigned int       *ticket;
  double             *results;
  unsigned long long *flag;
  unsigned long long  seq;  // 0: do not signal (a later stage on the stream -- all-reduce + hipx::red_signal -- will)
  double             *dres; // optional device-memory copy of the results, for kernels queued behind this one (launch-ahead CG)
  int                 pairs; // compensated kernels only: leave every sum as an unrounded (hi, lo) pair in results[2v], results[2v+1]
                             // (a fold over the ranks follows on the stream, hipx_comm.hip); 0: results[v] = hi + lo
};
inline RedOut red_out(int slot, bool signal = true, double *dres = nullptr)
{
  Runtime &r = rt();
  return RedOut{slot_partials(slot), r.d_tickets + slot, slot_results_dev(slot), r.d_flags + slot, signal ? ++r.seq[slot] : 0ull, dres, 0};
}
int launch_dot(const double *x, const double *y, hipx_int n, int slot, double *dres);  // x.y -> host slot (+ device copy), enqueue only
int launch_sum(const double *x, hipx_int n, int slot, double *dres);  // fold of per-wave partials: results -> host slot (+ device copy)
// multi-GPU reductions without a host round trip between the local kernel and the all-reduce (hipx_comm.hip)
// The local kernel leaves its sums in device memory (dev_results), RCCL reduces them there, and red_signal() copies the
// reduced words into the slot's host-mapped result area before raising the sequence flag.
int launch_mdot_nosignal(const double *x, int nv, const double *const *y, hipx_int n, int slot, double *dev_results);
int launch_cg_fused_nosignal(double *x, double *r, double *z, const double *p, const double *w, const double *d, double a, hipx_int n, int slot, double *dev_results);
int launch_cg_fused_dev_nosignal(double *x, double *r, double *z, const double *p, const double *w, const double *d, double dconst, const double *dev_beta, const double *dev_dpi, hipx_int n,
                                 int slot, double *dev_results);
// hipx_pipe.hip: the PIPECG update kernel with an explicit reduction destination (hipx_comm.hip redirects the sums to the all-reduce staging line)
int launch_pipecg_update(const hipxPipeCGVecs *v, const double *d, double dconst, int normkind, int first, const double *dev_sums, const double *dev_sums_old, const double *dev_alpha_old,
                         double *dev_alpha_out, hipx_int n, const RedOut &o);
int launch_gropp_dir(double *p, double *s, double *x, const double *z, const double *Z, const double *dev_gamma_new, const double *dev_gamma_old, const double *dev_alpha_old, hipx_int n,
                     const RedOut &o);
int launch_gropp_update(double *r, double *z, const double *s, const double *d, double dconst, int normkind, const double *dev_gamma, const double *dev_t, double *dev_alpha_out, hipx_int n,
                        const RedOut &o);
int red_signal(int slot, const double *dev_results, int nvals, double *dres = nullptr);  // enqueue: publish to the host (values, then sequence flag) and optionally to device memory, stream-ordered
int red_wait(int slot, int nvals, double *out);

// optional HIP-event bracketing of named sections of the hot path (bench.py: per-rank diagnosis of a scaling curve).
// Sections: see HIPX_PROF_* in hipx.h.  Off by default; when on, every bracket costs two hipEventRecord calls.
int prof_section(int id, bool start, hipStream_t s);

}  // namespace hipx

#define HIPX_CHECK_INIT() \
  do { \
    if (!hipx::rt().initialized) return hipx::fail(HIPX_ERR_ORDER, "hipxInit() has not been called", __FILE__, __LINE__); \
  } while (0)

#define HIPX_HIP(call) \
  do { \
    hipError_t e_ = (call); \
    if (e_ != hipSuccess) return hipx::fail(HIPX_ERR_HIP_BASE + (int)e_, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

#define HIPX_LAUNCH_CHECK() HIPX_HIP(hipGetLastError())

#define HIPX_ARG(cond, msg) \
  do { \
    if (!(cond)) return hipx::fail(HIPX_ERR_ARG, msg, __FILE__, __LINE__); \
  } while (0)
