// hipx_sor.hip -- MatSOR_SeqAIJ replacement (aij.c:1797-2007) for gfx950: level-scheduled Gauss-Seidel sweeps.
//
// SOR is sequential by definition: row i of a forward sweep needs the NEW values of its lower neighbours j < i and
// the OLD values of its upper neighbours j > i.  Parity with the reference needs exactly that dependency order, so
// the rows are grouped into levels computed from the symmetrised pattern: level(i) > level(j) for every j < i that
// appears in row i or in whose row i appears.  All rows of one level are independent; forward sweeps run the levels
// in increasing order and backward sweeps in decreasing order, which reproduces the sequential new/old value usage
// of aij.c:1931-2002 for both directions.  Within a row the terms are subtracted left to right
// (PetscSparseDenseMinusDot, aij.h:519-560) without FMA, so the sweep is bit-identical to the CPU sweep.
//
// Layout: a level-ordered copy of the matrix (rows permuted so that a level is one contiguous range; each row keeps
// its [lower | diag | upper] entries and original column ids).  One launch per level, thread per row; neighbouring
// lanes walk neighbouring rows, so every fetched line of val/col is consumed by the wave across its k-loop.
// The launch sequence of a sweep is captured once into a hipGraph and replayed (hundreds of ~us-sized launches:
// 3n-2 levels for the 7-point stencil in natural ordering).
#include "hipx_internal.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace hipx;

// hipxMat_s is defined in hipx_mat.hip; the SOR state lives behind this accessor to keep one definition
struct hipxSorState {
  bool      ready = false;
  hipx_int  m = 0, nlevels = 0;
  bool      is64 = false;
  std::vector<hipx_int> lev_ptr;       // host copy, nlevels + 1
  hipx_int *d_perm = nullptr;          // permuted row -> original row
  int64_t  *d_pi = nullptr;            // permuted row offsets (m + 1)
  hipx_int *d_pd = nullptr;            // offset of the diagonal entry inside the permuted row
  hipx_int *d_pj = nullptr;
  double   *d_pa = nullptr;
  double   *d_idiag = nullptr, *d_mdiag = nullptr, *d_t = nullptr;
  // dependency-driven ("sync-free") sweeps: wave-aligned slot map, work vector, ticket and error words
  hipx_int  nslots = 0;
  hipx_int *d_slot = nullptr;   // slot -> permuted row, -1 = padding (every wave holds rows of ONE level)
  int4     *d_smeta = nullptr;  // per slot {original row | -1, diagonal offset, row length, 0}: one load instead of slot -> perm -> pi/pd
  int64_t  *d_sks = nullptr;    // per slot start of the row in pj/pa
  bool      smeta_valid = false;
  hipx_int  nslots4 = 0;        // the same with every level padded to 4 rows: the cooperative kernel (16 lanes per row, 4 rows per wave)
  hipx_int *d_slot4 = nullptr;
  int4     *d_smeta4 = nullptr;
  int64_t  *d_sks4 = nullptr;
  double   *d_w1 = nullptr, *d_w2 = nullptr;
  bool      mdiag_valid = false;  // d_mdiag holds the diagonal of the current values (Eisenstat in strand mode)
  unsigned int *d_ctl = nullptr;  // [0] block ticket, [1] error flag
  int       mode = 1;           // 1 = dependency-driven single launch per sweep, 0 = one launch per level
  double    omega = 0.0, shift = 0.0;
  bool      idiag_valid = false, values_valid = false;
  unsigned int zero_pivots = 0;
  void     *strand = nullptr;   // StrandState: strand-scheduled sweeps for template (stencil) matrices
  bool      strand_tried = false;
  const int *var_tstart = nullptr, *var_toff = nullptr;  // device tables of the pattern templates (owned by the matrix)
  int       last_mode = -1;
  unsigned long long strand_vstate = 0;  // value state of the matrix the strand tables were built from
  void     *inode = nullptr;    // InodeState: node-level sweeps of a matrix with inodes (MatSOR_SeqAIJ_Inode)
  void     *box = nullptr;      // hipx_sorbox.hip: plane-march schedule of constant-coefficient box stencils (zero-guess sweeps)
  bool      box_tried = false;
};

// hipx_sorbox.hip
extern "C" int  hipxSorBoxBuild_(long long m, int ntmpl, const int *tstart, const int *toff, const double *tval, const int *tdiag, const int64_t *tcount, const unsigned char *d_tid, void **out);
extern "C" int  hipxSorBoxRun_(void *box, int kind, const double *rhs, double *tout, double *xout, double omega, double shift, int xfull);
extern "C" int  hipxSorBoxError_(void *box, unsigned int *err, int sync);
extern "C" int  hipxSorBoxFill_(void *box, double *xout, int rev);
extern "C" void hipxSorBoxFree_(void *box);

extern "C" {
// accessors implemented in hipx_mat.hip
int   hipxMatInternal_(hipxMat A, hipx_int *m, hipx_int *n, int64_t *nnz, int *is64, void **d_i, hipx_int **d_j, double **d_a, int64_t **d_diagpos, int *diag_dense,
                       int *compressed, void ***sor_slot, unsigned long long *value_state);
int   hipxMatInodes_(hipxMat A, int *state, hipx_int *node_count, const hipx_int **sizes);
int   hipxMatInodesFound_(hipxMat A, hipx_int node_count, const hipx_int *sizes);
}

namespace {

constexpr int SOR_THREADS = 256;

__global__ void permute_rows_kernel(hipx_int m, const hipx_int *perm, const int64_t *pi, const void *ai_, int is64, const hipx_int *aj, const double *aa,
                                    const int64_t *diagpos, hipx_int *pj, double *pa, hipx_int *pd)
{
  for (hipx_int p = (hipx_int)blockIdx.x * blockDim.x + threadIdx.x; p < m; p += (hipx_int)gridDim.x * blockDim.x) {
    const hipx_int i  = perm[p];
    const int64_t  s  = is64 ? ((const int64_t *)ai_)[i] : (int64_t)((const hipx_int *)ai_)[i];
    const int64_t  e  = is64 ? ((const int64_t *)ai_)[i + 1] : (int64_t)((const hipx_int *)ai_)[i + 1];
    const int64_t  o  = pi[p];
    for (int64_t k = s; k < e; k++) {
      pj[o + (k - s)] = aj[k];
      pa[o + (k - s)] = aa[k];
    }
    pd[p] = (hipx_int)(diagpos[i] - s);
  }
}

__global__ void slot_meta_kernel(hipx_int nslots, const hipx_int *slot, const hipx_int *perm, const int64_t *pi, const hipx_int *pd, int4 *smeta, int64_t *sks)
{
  for (hipx_int s = (hipx_int)blockIdx.x * blockDim.x + threadIdx.x; s < nslots; s += (hipx_int)gridDim.x * blockDim.x) {
    const hipx_int p = slot[s];
    if (p < 0) {
      smeta[s] = make_int4(-1, 0, 0, 0);
      sks[s]   = 0;
    } else {
      smeta[s] = make_int4(perm[p], pd[p], (int)(pi[p + 1] - pi[p]), 0);
      sks[s]   = pi[p];
    }
  }
}

// MatInvertDiagonalForSOR_SeqAIJ (aij.c:1797-1840)
__global__ void invert_diag_kernel(hipx_int m, const int64_t *diagpos, const double *aa, double omega, double shift, int plain, double *idiag, double *mdiag,
                                   unsigned int *zero_pivots)
{
  for (hipx_int i = (hipx_int)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (hipx_int)gridDim.x * blockDim.x) {
    const double d = aa[diagpos[i]];
    mdiag[i]       = d;
    if (plain) {  // omega == 1 && shift <= 0
      if (d == 0.0) atomicAdd(zero_pivots, 1u);
      idiag[i] = 1.0 / d;
    } else idiag[i] = omega / (shift + d);
  }
}

// KIND 0: zero-guess forward   (aij.c:1931-1942)   t = b - L x ; x = t * idiag
// KIND 1: backward, xb == t    (aij.c:1944-1958)   x = (1-w) x + (t - U x) idiag
// KIND 2: backward, xb == b, zero guess (no forward sweep before)   x = (b - U x) idiag
// KIND 3: forward, general     (aij.c:1962-1978)   t = b - L x ; x = (1-w) x + (t - U x) idiag
// KIND 4: backward, whole row  (aij.c:1984-1990)   x = (1-w) x + (b - A x + mdiag x) idiag
template <int KIND>
__global__ __launch_bounds__(SOR_THREADS) void sor_level_kernel(hipx_int p0, hipx_int p1, const hipx_int *__restrict__ perm, const int64_t *__restrict__ pi,
                                                                 const hipx_int *__restrict__ pd, const hipx_int *__restrict__ pj, const double *__restrict__ pa,
                                                                 const double *__restrict__ idiag, const double *__restrict__ mdiag, const double *b, double *t, double *x,
                                                                 double omega)
{
  const hipx_int p = p0 + (hipx_int)blockIdx.x * SOR_THREADS + threadIdx.x;
  if (p >= p1) return;
  const hipx_int i = perm[p];
  const int64_t  s = pi[p], e = pi[p + 1], d = s + pd[p];
  double         sum;
  if (KIND == 0 || KIND == 3) {
    sum = b[i];
    for (int64_t k = s; k < d; k++) sum -= pa[k] * x[pj[k]];
    t[i] = sum;
    if (KIND == 0) x[i] = sum * idiag[i];
    else {
      for (int64_t k = d + 1; k < e; k++) sum -= pa[k] * x[pj[k]];
      x[i] = (1. - omega) * x[i] + sum * idiag[i];
    }
  } else if (KIND == 1 || KIND == 2) {
    sum = (KIND == 1) ? t[i] : b[i];
    for (int64_t k = d + 1; k < e; k++) sum -= pa[k] * x[pj[k]];
    if (KIND == 2) x[i] = sum * idiag[i];
    else x[i] = (1 - omega) * x[i] + sum * idiag[i];
  } else {
    sum = b[i];
    for (int64_t k = s; k < e; k++) sum -= pa[k] * x[pj[k]];
    x[i] = (1. - omega) * x[i] + (sum + mdiag[i] * x[i]) * idiag[i];
  }
}

// SOR_EISENSTAT, middle step (aij.c:1909-1911): t = b - scale * a_ii * x, evaluated as (scale * a_ii) * x like the host loop
__global__ void sor_eisenstat_mid_kernel(hipx_int m, const double *__restrict__ mdiag, const double *__restrict__ b, const double *__restrict__ x, double scale, double *__restrict__ t)
{
  for (hipx_int i = (hipx_int)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (hipx_int)gridDim.x * blockDim.x) t[i] = b[i] - scale * mdiag[i] * x[i];
}

// SOR_APPLY_UPPER (aij.c:1867-1884): x_i = b_i * (shift + d_i) / omega + sum_{j>i} a_ij b_j  -- no dependencies
__global__ void sor_apply_upper_kernel(hipx_int m, const hipx_int *perm, const int64_t *pi, const hipx_int *pd, const hipx_int *pj, const double *pa, const double *mdiag,
                                       const double *b, double *x, double omega, double shift)
{
  for (hipx_int p = (hipx_int)blockIdx.x * blockDim.x + threadIdx.x; p < m; p += (hipx_int)gridDim.x * blockDim.x) {
    const hipx_int i = perm[p];
    const int64_t  e = pi[p + 1], d = pi[p] + pd[p];
    double         sum = b[i] * (shift + mdiag[i]) / omega;
    for (int64_t k = d + 1; k < e; k++) sum += pa[k] * b[pj[k]];
    x[i] = sum;
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Dependency-driven sweep: ONE launch per sweep instead of one per level.
// Rows sit in level order in wave-aligned slots (a wave never mixes levels, so lanes of a wave never wait on each
// other).  A row reads the values it depends on from the NEW vector, which is pre-filled with a sentinel bit pattern
// (a signalling NaN no arithmetic produces); it simply polls those 8-byte words with agent-scope relaxed loads until
// they stop being the sentinel -- the datum is its own ready flag (one naturally aligned 8-byte granule written by
// one agent-scope store: no fence, no separate flag; MI355X_MICROARCH "handoff-1to1").  Values on the other side of
// the diagonal come from the OLD vector with plain loads.
// Progress: workgroups take a ticket at start and process slots in ticket order, so every row a workgroup can wait
// for belongs to a workgroup that has already started (and workgroups are never pre-empted).  Spins are bounded:
// a lane that gives up raises the error word instead of hanging the device.
constexpr unsigned long long SOR_SENTINEL = 0x7FF4DEADBEEF0001ULL;
constexpr long long          SOR_SPIN_TICKS = 400000000LL;  // 4 s of the 100 MHz wall clock (wall_clock64) before a launch is declared stuck:
                                                             // elapsed time, not poll counts, so GPU sharing / pre-emption cannot trip it

__device__ __forceinline__ double sor_poll(const double *p, unsigned int *err)
{
  const unsigned long long *q = reinterpret_cast<const unsigned long long *>(p);
  unsigned long long        v = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int                       spins = 0;
  long long                 t0 = 0;
  while (v == SOR_SENTINEL) {
    __builtin_amdgcn_s_sleep(8);
    v = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ++spins;
    if ((spins & 0xff) == 0) {  // global abort: once any lane has given up, nobody waits any more (bounds the whole launch)
      const long long now = (long long)wall_clock64();
      if (!t0) t0 = now;
      if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || now - t0 > SOR_SPIN_TICKS) {
        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
  return __longlong_as_double((long long)v);
}

// the cooperative kernels' poll: the same loop with the back-off between two looks selectable (psleep: 0 none, 1, 2, 4, else 8 units of 64 clocks)
__device__ __forceinline__ double sor_poll_sel(const double *p, unsigned int *err, const int psleep)
{
  const unsigned long long *q = reinterpret_cast<const unsigned long long *>(p);
  unsigned long long        v = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int                       spins = 0;
  long long                 t0 = 0;
  while (v == SOR_SENTINEL) {
    if (psleep == 1) __builtin_amdgcn_s_sleep(1);
    else if (psleep == 2) __builtin_amdgcn_s_sleep(2);
    else if (psleep == 4) __builtin_amdgcn_s_sleep(4);
    else if (psleep != 0) __builtin_amdgcn_s_sleep(8);
    v = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ++spins;
    if ((spins & 0xff) == 0) {
      const long long now = (long long)wall_clock64();
      if (!t0) t0 = now;
      if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || now - t0 > SOR_SPIN_TICKS) {
        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
  return __longlong_as_double((long long)v);
}

__device__ __forceinline__ void sor_publish(double *p, double v)
{
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void sor_fill_kernel(double *x, hipx_int n)
{
  unsigned long long *q = reinterpret_cast<unsigned long long *>(x);
  for (hipx_int i = (hipx_int)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (hipx_int)gridDim.x * blockDim.x) q[i] = SOR_SENTINEL;
}

// subtract a[k] * value(j) for k in [k0, k1) left to right; DEPLOW: columns < i come from xnew (polled), others from xold
// (DEPLOW = false: columns > i polled).  Four gathers are issued before the first dependent subtract.
template <bool DEPLOW>
__device__ __forceinline__ double sor_minusdot(double sum, hipx_int i, int64_t k0, int64_t k1, const hipx_int *__restrict__ pj, const double *__restrict__ pa,
                                               const double *xold, const double *xnew, unsigned int *err)
{
  constexpr int CH = 16;  // entries whose column/value loads are all in flight before the first poll (16: the 13 - 14 dependency entries of a 27-point row in ONE chunk -- with 8 the second chunk's loads waited behind the first chunk's polls)
  for (int64_t k = k0; k < k1; k += CH) {
    hipx_int j[CH];
    double   a[CH], v[CH];
    const int nk = (int)((k1 - k) < CH ? (k1 - k) : CH);
#pragma unroll
    for (int c = 0; c < CH; c++) {
      const int64_t kk = (c < nk) ? k + c : k;  // clamped: branch-free issue
      j[c]             = pj[kk];
      a[c]             = pa[kk];
    }
#pragma unroll
    for (int c = 0; c < CH; c++) {  // first look at every operand (loads overlap) ...
      const bool dep = DEPLOW ? (j[c] < i) : (j[c] > i);
      if (dep) v[c] = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(xnew + j[c]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      else v[c] = xold[j[c]];
    }
#pragma unroll
    for (int c = 0; c < CH; c++) {  // ... then wait only for the ones that are not published yet, and subtract left to right
      if (c < nk) {
        const bool dep = DEPLOW ? (j[c] < i) : (j[c] > i);
        if (dep && (unsigned long long)__double_as_longlong(v[c]) == SOR_SENTINEL) v[c] = sor_poll(xnew + j[c], err);
        sum -= a[c] * v[c];
      }
    }
  }
  return sum;
}

template <int KIND>
__global__ __launch_bounds__(SOR_THREADS) void sor_dep_kernel(hipx_int nslots, const int4 *__restrict__ smeta, const int64_t *__restrict__ sks, const hipx_int *__restrict__ pj, const double *__restrict__ pa,
                                                               const double *__restrict__ idiag, const double *__restrict__ mdiag, const double *b, double *t, const double *xold,
                                                               double *xnew, double omega, unsigned int *ctl)
{
  // persistent waves: each wave keeps taking the next 64-slot group (one level per group) in ticket order, so the number
  // of pollers is bounded by the launch size, not by the matrix size
  constexpr bool FWD = (KIND == 0 || KIND == 3);
  unsigned int  *err = ctl + 1;
  const int      lane = threadIdx.x & 63;
  const hipx_int ngroups = nslots >> 6;
  for (;;) {
    unsigned int v = 0;
    if (lane == 0) v = atomicAdd(&ctl[0], 1u);
    v = __shfl(v, 0, 64);
    if ((hipx_int)v >= ngroups) return;
    const hipx_int g = (hipx_int)v * 64 + lane;
    const hipx_int s = FWD ? g : nslots - 1 - g;
    const int4     mt = smeta[s];  // {row, offset of the diagonal, row length, -} ; row < 0 = padding
    const int64_t  ks = sks[s];
    if (mt.x >= 0) {  // padding lanes skip the body but stay with their wave (the ticket logic below needs all 64 lanes converged)
    const hipx_int i  = mt.x;
    const int64_t  kd = ks + mt.y, ke = ks + mt.z;
    double         sum, out;
    // (round 6) the last step's operands are requested with the row's first loads, not after its chain (see sor_dep_coop_kernel)
    const double idg = idiag[i], xo = (KIND == 1 || KIND == 3 || KIND == 4) ? xold[i] : 0.0, md = (KIND == 4) ? mdiag[i] : 0.0;
    if (KIND == 0) {
      sum  = sor_minusdot<true>(b[i], i, ks, kd, pj, pa, xold, xnew, err);
      t[i] = sum;
      out  = sum * idg;
    } else if (KIND == 3) {
      sum  = sor_minusdot<true>(b[i], i, ks, kd, pj, pa, xold, xnew, err);
      t[i] = sum;
      sum  = sor_minusdot<true>(sum, i, kd + 1, ke, pj, pa, xold, xnew, err);  // upper part: old values
      out  = (1. - omega) * xo + sum * idg;
    } else if (KIND == 1) {
      sum = sor_minusdot<false>(t[i], i, kd + 1, ke, pj, pa, xold, xnew, err);
      out = (1 - omega) * xo + sum * idg;
    } else if (KIND == 2) {
      sum = sor_minusdot<false>(b[i], i, kd + 1, ke, pj, pa, xold, xnew, err);
      out = sum * idg;
    } else {
      sum = sor_minusdot<false>(b[i], i, ks, ke, pj, pa, xold, xnew, err);  // whole row: lower + diagonal old, upper new
      out = (1. - omega) * xo + (sum + md * xo) * idg;
    }
    sor_publish(xnew + i, out);
    }
  }
}

// The cooperative form of the dependency-driven sweep (round 4, default): 16 lanes per row, 4 rows per wave.  One lane per row walks its
// entries in chunks, each chunk's loads behind the polls of the chunk before (6.8 us per dependency level on the 27-point operator);
// here a row's entries are dealt to its 16 lanes -- every column, value and operand load of up to 32 entries in flight at once -- each
// lane parks its products a_k x_{j_k} in LDS, and every lane of the group then runs the row's subtraction chain over them in CSR order
// (LDS broadcasts): the bits of PetscSparseDenseMinusDot.  Lane 0 of the group scales and publishes.  (The same shape as the inode sweep
// further down, sor_inode_coop_kernel, where it was measured first: 34.8 -> 7.2 ms.)
constexpr int DEP_G = 16, DEP_R = 2, DEP_CH = DEP_G * DEP_R;  // lanes per row, entries per lane and chunk, entries per chunk

// entries [k0, k1): operand of column j is the NEW value (polled) when DEPLOW ? j < i : j > i, else the old one -- as sor_minusdot
template <bool DEPLOW>
__device__ __forceinline__ double dep_coop_minus(double sum, double *Q, const int l, hipx_int i, int64_t k0, int64_t k1, const hipx_int *__restrict__ pj, const double *__restrict__ pa,
                                                 const double *xold, const double *xnew, unsigned int *err)
{
  for (int64_t kc = k0; kc < k1; kc += DEP_CH) {
    const int cnt = (int)((k1 - kc) < DEP_CH ? (k1 - kc) : DEP_CH);
    hipx_int  j[DEP_R];
    double    a[DEP_R], v[DEP_R];
    bool      ok[DEP_R], dep[DEP_R];
#pragma unroll
    for (int rr = 0; rr < DEP_R; rr++) {
      const int q      = l + rr * DEP_G;
      ok[rr]           = q < cnt;
      const int64_t kk = ok[rr] ? kc + q : k0;
      j[rr]            = pj[kk];
      a[rr]            = pa[kk];
    }
#pragma unroll
    for (int rr = 0; rr < DEP_R; rr++) {
      dep[rr] = DEPLOW ? (j[rr] < i) : (j[rr] > i);
      if (dep[rr]) v[rr] = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(xnew + j[rr]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      else v[rr] = xold[j[rr]];
    }
#pragma unroll
    for (int rr = 0; rr < DEP_R; rr++)
      if (ok[rr] && dep[rr] && (unsigned long long)__double_as_longlong(v[rr]) == SOR_SENTINEL) v[rr] = sor_poll(xnew + j[rr], err);
#pragma unroll
    for (int rr = 0; rr < DEP_R; rr++)
      if (ok[rr]) Q[l + rr * DEP_G] = a[rr] * v[rr];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {  // the row's chain in CSR order; round 6: eight products are read from LDS before the first of their subtractions (see inode_coop_minus)
      int q = 0;
      for (; q + 8 <= cnt; q += 8) {
        double tq[8];
#pragma unroll
        for (int u = 0; u < 8; u++) tq[u] = Q[q + u];
#pragma unroll
        for (int u = 0; u < 8; u++) asm volatile("" : "+v"(tq[u]));
#pragma unroll
        for (int u = 0; u < 8; u++) sum -= tq[u];
      }
      for (; q < cnt; q++) sum -= Q[q];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  return sum;
}

template <int KIND>
__global__ __launch_bounds__(SOR_THREADS) void sor_dep_coop_kernel(hipx_int nslots, const int4 *__restrict__ smeta, const int64_t *__restrict__ sks, const hipx_int *__restrict__ pj,
                                                                    const double *__restrict__ pa, const double *__restrict__ idiag, const double *__restrict__ mdiag, const double *b, double *t,
                                                                    const double *xold, double *xnew, double omega, unsigned int *ctl)
{
  constexpr bool FWD = (KIND == 0 || KIND == 3);
  __shared__ double s_q[SOR_THREADS / 64][64 / DEP_G][DEP_CH];
  unsigned int  *err = ctl + 1;
  const int      lane = threadIdx.x & 63, wv = threadIdx.x >> 6, grp = lane / DEP_G, l = lane % DEP_G;
  double        *Q = s_q[wv][grp];
  const hipx_int ngroups = nslots / (64 / DEP_G);
  for (;;) {
    unsigned int v = 0;
    if (lane == 0) v = atomicAdd(&ctl[0], 1u);
    v = __shfl(v, 0, 64);
    if ((hipx_int)v >= ngroups) return;
    const hipx_int g  = (hipx_int)v * (64 / DEP_G) + grp;
    const hipx_int s  = FWD ? g : nslots - 1 - g;
    const int4     mt = smeta[s];  // {row, offset of the diagonal, row length, -}; row < 0: padding
    if (mt.x >= 0) {
      const int64_t  ks = sks[s];
      const hipx_int i  = mt.x;
      const int64_t  kd = ks + mt.y, ke = ks + mt.z;
      double         sum, out;
      // (round 6) what the row's last step needs is requested NOW, with the row's other loads -- read after the chain, each of these was a memory round trip
      // between the arrival of the row's last operand and its publication, on every dependency level
      const double idg = idiag[i], xo = (KIND == 1 || KIND == 3 || KIND == 4) ? xold[i] : 0.0, md = (KIND == 4) ? mdiag[i] : 0.0;
      if (KIND == 0) {
        sum = dep_coop_minus<true>(b[i], Q, l, i, ks, kd, pj, pa, xold, xnew, err);
        if (l == 0) t[i] = sum;
        out = sum * idg;
      } else if (KIND == 3) {
        sum = dep_coop_minus<true>(b[i], Q, l, i, ks, kd, pj, pa, xold, xnew, err);
        if (l == 0) t[i] = sum;
        sum = dep_coop_minus<true>(sum, Q, l, i, kd + 1, ke, pj, pa, xold, xnew, err);
        out = (1. - omega) * xo + sum * idg;
      } else if (KIND == 1) {
        sum = dep_coop_minus<false>(t[i], Q, l, i, kd + 1, ke, pj, pa, xold, xnew, err);
        out = (1 - omega) * xo + sum * idg;
      } else if (KIND == 2) {
        sum = dep_coop_minus<false>(b[i], Q, l, i, kd + 1, ke, pj, pa, xold, xnew, err);
        out = sum * idg;
      } else {
        sum = dep_coop_minus<false>(b[i], Q, l, i, ks, ke, pj, pa, xold, xnew, err);
        out = (1. - omega) * xo + (sum + md * xo) * idg;
      }
      if (l == 0) sor_publish(xnew + i, out);
    }
  }
}

template <int KIND>
int run_dep(hipxSorState *S, const double *b, const double *xold, double *xnew, double omega)
{
  hipStream_t    st = rt().compute;
  const hipx_int g  = std::min<hipx_int>((S->m + 255) / 256, 4096);
  sor_fill_kernel<<<(unsigned)g, 256, 0, st>>>(xnew, S->m);
  HIPX_HIP(hipMemsetAsync(S->d_ctl, 0, sizeof(unsigned int), st));  // ticket only; the error word is sticky until read
  static int waves_per_cu = 0;
  if (!waves_per_cu) {
    const char *e = getenv("HIPX_SOR_WAVES_PER_CU");
    waves_per_cu  = e ? atoi(e) : 2;  // measured: fewer pollers = faster hops (2 beat 1, 4, 8 on 7-pt 192^3 and 27-pt 128^3)
    if (waves_per_cu < 1) waves_per_cu = 1;
    if (waves_per_cu > 32) waves_per_cu = 32;
  }
  // which form: the cooperative one shortens the hop from level to level (2.7 us against 6.8 - 13) but carries 4 rows per wave where the
  // lane-per-row form carries 64 -- it wins where the levels are NARROW (unstructured orderings: the elasticity stand-in as a point
  // matrix, ~380 rows per level: 85.6 -> 32.0 ms per symmetric sweep) and loses where they are wide (27-point 256^3 in natural
  // ordering, 9362 rows per level: 24.3 -> 96.8 ms): decided by the average level width
  int coop = (S->nlevels > 0 && (int64_t)S->m / S->nlevels <= 2048) ? 1 : 0, coop_blocks = 256;  // (one workgroup per CU: more pollers crowd out the publishers, see run_inode)
  {
    const char *e = getenv("HIPX_SOR_DEP_COOP");  // 0: one lane per row (rounds 1-3), 1: cooperative, whatever the level widths
    if (e) coop = atoi(e);
    e = getenv("HIPX_SOR_DEP_COOP_BLOCKS");
    if (e) coop_blocks = atoi(e);
    if (coop_blocks < 1) coop_blocks = 1;
    if (coop_blocks > 4096) coop_blocks = 4096;
  }
  if (coop && S->d_smeta4) {
    unsigned       cgrid = (unsigned)coop_blocks;
    const unsigned cneed = (unsigned)((S->nslots4 / (64 / DEP_G) * 64 + SOR_THREADS - 1) / SOR_THREADS);
    if (cgrid > cneed) cgrid = cneed ? cneed : 1;
    sor_dep_coop_kernel<KIND><<<cgrid, SOR_THREADS, 0, st>>>(S->nslots4, S->d_smeta4, S->d_sks4, S->d_pj, S->d_pa, S->d_idiag, S->d_mdiag, b, S->d_t, xold, xnew, omega, S->d_ctl);
    HIPX_LAUNCH_CHECK();
    return HIPX_SUCCESS;
  }
  unsigned grid = (unsigned)(256 * waves_per_cu * 64 / SOR_THREADS);
  const unsigned need = (unsigned)((S->nslots + SOR_THREADS - 1) / SOR_THREADS);
  if (grid > need) grid = need ? need : 1;
  sor_dep_kernel<KIND><<<grid, SOR_THREADS, 0, st>>>(S->nslots, S->d_smeta, S->d_sks, S->d_pj, S->d_pa, S->d_idiag, S->d_mdiag, b, S->d_t, xold, xnew, omega,
                                                     S->d_ctl);
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}

template <int KIND>
int run_levels(hipxSorState *S, bool forward, const double *b, double *x, double omega)
{
  hipStream_t st = rt().compute;
  for (hipx_int l = 0; l < S->nlevels; l++) {
    const hipx_int L  = forward ? l : S->nlevels - 1 - l;
    const hipx_int p0 = S->lev_ptr[L], p1 = S->lev_ptr[L + 1];
    if (p1 == p0) continue;
    const unsigned g = (unsigned)((p1 - p0 + SOR_THREADS - 1) / SOR_THREADS);
    sor_level_kernel<KIND><<<g, SOR_THREADS, 0, st>>>(p0, p1, S->d_perm, S->d_pi, S->d_pd, S->d_pj, S->d_pa, S->d_idiag, S->d_mdiag, b, S->d_t, x, omega);
  }
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}


// ---------------------------------------------------------------------------------------------------------------
// Strand-scheduled sweeps for stencil-structured matrices (row templates, hipx_mat.hip).
//
// The level schedule above pays one cross-CU memory hop (2.5-6 us) per dependency level: 1785 levels on a 27-point
// 512x512x64 slab = 24 ms per symmetric sweep.  Here the critical chain stays inside a wave:
//   * a STRAND is a run of L consecutive rows (an x-line of the grid; L is read off the interior template).  Inside a strand
//     every row depends on its predecessor, so a strand is sequential by nature: ONE LANE walks it front to back.
//   * a PANEL is 64 consecutive strands = one wave.  Lane l trails lane l-1 by the two rows the stencil demands, so a wave is
//     a skewed wavefront that advances every lane by one row per iteration; the values a lane needs from its neighbours'
//     strands go through a tagged LDS window ({value, position} slots: a slot is valid for exactly one row), never through
//     memory.
//   * strands of OTHER panels ("far": the plane below, the line next to the panel) are staged into the same window by a
//     LOADER wave of the workgroup, which polls the NEW vector (sentinel = not published yet, as above) a few rows ahead of
//     its consumers; it also streams the per-row operands (b or t, old x, template id) into an LDS ring.  The COMPUTE wave
//     therefore never waits for a global load: its iteration is LDS reads, a handful of fp64 operations and two stores.
//   * panels are taken in ticket order = natural order, so every value a panel can wait for belongs to a panel that is
//     already running (no deadlock, any grid size); waits are bounded by wall-clock time and end in an error code.
// The matrix itself is not read at all: rows come from the template table in LDS (1 byte per row from HBM).
// Arithmetic per row is MatSOR_SeqAIJ's (aij.c:1930-2002): same operands, same left-to-right order, no FMA -> bit-identical.
// SIMT model used to design and check the schedule: scripts/sor_strand_model.py.
constexpr int ST_WP   = 16;  // window positions per LDS row
constexpr int ST_RQ   = 16;  // per-row operand ring (positions per lane)
// LDS bank swizzle.  Lane s of the compute wave reads slot (p_s & 15) of window row (s + const): with the slots of a row in
// natural order the 64 addresses of one ds_read_b128 fall into the same four banks when the lanes are at the same position
// (64-way conflict: ~512 clocks per read, 13 reads per iteration) and into 16 bank groups when they are one position apart.
// The LDS pipe is shared by the two panels of a CU, so a neighbour SPINNING on such reads starved the panel that had work
// (measured: 613 clocks per burst; panels in "slow mode" at 3 us per row next to fast ones at 1.4).  Position q of window row
// w lives in slot (q + 7 w) & 15 instead: lanes k positions apart are then (7 - k) slots apart -- conflict-free for k = 0, 2, 4,
// 6, two-way for k = 1, 5, four-way for k = 3 (lanes in flow are 2 apart, blocked ones 1).  Same rotation for the operand ring.
constexpr int ST_ROT = 7;
constexpr int ST_SB   = 8;   // positions per staging batch
constexpr int ST_LA   = 10;  // staging look-ahead beyond the leading consumer
constexpr int ST_NB   = 3;   // bands of strands a panel may touch
constexpr int ST_MAXW = 4;   // strands per band
constexpr int ST_ME   = 16;  // most dependency entries a row may have on this schedule (27-point: 13)
constexpr int ST_MC   = 4;   // split kernel: entries of the C wave (the last ones of the list; the F waves take the first ME - ST_MC)
constexpr int ST_CQ   = 4;   // depth of the F -> C hand-over ring (16-byte records)
constexpr int ST_NULLPK  = 0x7fff7fff;  // pk of a padding entry
constexpr int ST_NULLTAG = 0x7ffffff0;  // tag of the null slot (no row position reaches it)

struct StBand {
  int dsmin, width, rowbase, pad;
};
struct StParams {
  hipx_int m, L, nstr, npanels;
  int      nbands, nrows, ntmpl, ndep, nold, maxchunks;
  StBand   band[ST_NB];
  int      off_win, off_rowq, off_prog, off_ctl, off_tinfo, off_tdiag, off_dep, off_old, off_null, me, lds_bytes;
  int      split, off_cq, off_progF, off_depF, off_depC;  // split kernel (ME 16): hand-over ring F -> C, F's progress, the two entry tables
  int      trace_it0;    // ... first iteration of the per-iteration log of lane trace_lane
  int      trace_rows;   // ... per-row stamps on (they cost one scattered store per row)
  int      trace_lane;   // ... and the loader lane whose passes are logged
  int      poll_adapt;   // ask for (last pass's rows + 2) rows per far strand instead of always 8 (default: the ME 4 kernels; HIPX_SOR_ADAPT=0|1)
  int      poll_sys;     // experiment (HIPX_SOR_POLL=sys): far polls at system scope
  int      lock, off_lock, lock_wrow;  // lockstep C wave (st_lock_c): on; its per-template table {c0,c1}{c2,c3}{mask}; window row of the strand before the panel's first
  unsigned rolemap;      // split kernel: role (0 C, 1 F even, 2 F odd, 3 loader) of the wave on SIMD s of the CU's first / second resident workgroup: nibble s / 4 + s; 0: by wave index
  int      trace_panel;  // HIPX_SOR_DEBUG + HIPX_SOR_TRACE_PANEL: the panel whose rows / loader passes are time-stamped (-1: none)
  // variable-coefficient kernels (pattern templates: the tables hold the STRUCTURE of every row only; the coefficients and the
  // inverse diagonal of every row come from a per-direction stream the loader stages into an LDS ring)
  int      var, cw, rq, off_cring, wgcu;  // on; doubles per coefficient record (ME coefficients, 1 / d, padding to even); ring depth
                                          // (operand AND coefficient ring); LDS offset of the coefficient ring; workgroups per CU that fit
};
struct __attribute__((aligned(16))) StEntry {  // dep: pk = (window row offset << 16) | (dp & 0xffff), lo = logical row offset; old: pk = ACTUAL column - row
  int    pk, lo;
  double val;
};
struct __attribute__((aligned(16))) StTinfo {
  int dstart, dcnt, ostart, ocnt;
};
struct __attribute__((aligned(16))) StDiag {
  double idiag, mdiag;
};
struct __attribute__((aligned(16))) StSlot {  // one window slot: valid for the row at position `tag` only
  double v;
  int    tag, pad;
};
struct __attribute__((aligned(16))) StRow {  // per-row operands staged by the loader
  double a, b;
  int    tid, tag, pad0, pad1;
};
typedef int st_int4 __attribute__((ext_vector_type(4)));
// LDS is addressed through explicit address-space-3 pointers: generic pointers made the compiler emit FLAT accesses, which
// count on vmcnt as well -- the compute wave must never wait on the memory counters.  `volatile` 16-byte vector accesses
// compile to single ds_read_b128 / ds_write_b128 (checked in the ISA): a slot is read and written whole.
typedef __attribute__((address_space(3))) char    st_lds_char;
typedef __attribute__((address_space(3))) int     st_lds_int;
typedef __attribute__((address_space(3))) st_int4 st_lds_int4;

// reads of slots another wave writes: plain 16-byte loads (one ds_read_b128 each, free to issue back to back) -- the compute
// loop declares memory clobbered once per iteration (asm volatile "" ::: "memory"), so nothing read in an earlier iteration is
// reused; `volatile` loads would be kept in strict order with a wait between them
__device__ __forceinline__ st_int4 st_ld4(st_lds_char *base, int byteoff) { return *reinterpret_cast<st_lds_int4 *>(base + byteoff); }
__device__ __forceinline__ void    st_st4v(st_lds_char *base, int byteoff, st_int4 v) { *reinterpret_cast<volatile st_lds_int4 *>(base + byteoff) = v; }
__device__ __forceinline__ double  st_dbl(int lo, int hi) { return __longlong_as_double(((long long)(unsigned)hi << 32) | (unsigned)lo); }
__device__ __forceinline__ st_int4 st_pack_slot(double v, int tag)
{
  const long long b = __double_as_longlong(v);
  st_int4         r;
  r.x = (int)(unsigned)b;
  r.y = (int)(unsigned)((unsigned long long)b >> 32);
  r.z = tag;
  r.w = 0;
  return r;
}

// ---- hand-issued memory operations of the compute wave.  hipcc keeps one in-order counter per memory class and is
// conservative at control-flow joins: a rarely taken global load (or a register it re-uses while a read is in flight) puts
// `s_waitcnt vmcnt(..)` / `lgkmcnt(0)` into the common path, and vmcnt also counts the wave's own stores -- i.e. HBM latency in
// a loop that must run at LDS speed.  The loads below are therefore issued from inline asm with their wait INSIDE the
// statement (cdna_hip_programming.md 5.7: the compiler does not count memory operations inside an asm statement, so every one
// of them is complete when the statement ends).
// 16 window / table reads as ONE burst: 16 x ds_read_b128 back to back, one s_waitcnt lgkmcnt(0).
__device__ __forceinline__ void st_lds_burst16(st_int4 (&o)[16], const unsigned (&a)[16])
{
  asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\tds_read_b128 %2, %10\n\tds_read_b128 %3, %11\n\t"
               "ds_read_b128 %4, %12\n\tds_read_b128 %5, %13\n\tds_read_b128 %6, %14\n\tds_read_b128 %7, %15"
               : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
               : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7])
               : "memory");
  // the second half ties the first eight results (read-write operands): nothing may use them before this statement's wait
  asm volatile("ds_read_b128 %8, %16\n\tds_read_b128 %9, %17\n\tds_read_b128 %10, %18\n\tds_read_b128 %11, %19\n\t"
               "ds_read_b128 %12, %20\n\tds_read_b128 %13, %21\n\tds_read_b128 %14, %22\n\tds_read_b128 %15, %23\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]), "+v"(o[6]), "+v"(o[7]), "=&v"(o[8]), "=&v"(o[9]), "=&v"(o[10]), "=&v"(o[11]),
                 "=&v"(o[12]), "=&v"(o[13]), "=&v"(o[14]), "=&v"(o[15])
               : "v"(a[8]), "v"(a[9]), "v"(a[10]), "v"(a[11]), "v"(a[12]), "v"(a[13]), "v"(a[14]), "v"(a[15])
               : "memory");
}
__device__ __forceinline__ void st_lds_burst4(st_int4 (&o)[4], const unsigned (&a)[4])
{
  asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
               : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3])
               : "memory");
}
__device__ __forceinline__ void st_lds_burst12(st_int4 (&o)[12], const unsigned (&a)[12])
{
  asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %7\n\tds_read_b128 %2, %8\n\tds_read_b128 %3, %9\n\tds_read_b128 %4, %10\n\tds_read_b128 %5, %11"
               : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5])
               : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5])
               : "memory");
  asm volatile("ds_read_b128 %6, %12\n\tds_read_b128 %7, %13\n\tds_read_b128 %8, %14\n\tds_read_b128 %9, %15\n\tds_read_b128 %10, %16\n\tds_read_b128 %11, %17\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]), "=&v"(o[6]), "=&v"(o[7]), "=&v"(o[8]), "=&v"(o[9]), "=&v"(o[10]), "=&v"(o[11])
               : "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]), "v"(a[10]), "v"(a[11])
               : "memory");
}
__device__ __forceinline__ void st_lds_burst13(st_int4 (&o)[13], const unsigned (&a)[13])
{
  asm volatile("ds_read_b128 %0, %7\n\tds_read_b128 %1, %8\n\tds_read_b128 %2, %9\n\tds_read_b128 %3, %10\n\tds_read_b128 %4, %11\n\tds_read_b128 %5, %12\n\tds_read_b128 %6, %13"
               : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6])
               : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6])
               : "memory");
  asm volatile("ds_read_b128 %7, %13\n\tds_read_b128 %8, %14\n\tds_read_b128 %9, %15\n\tds_read_b128 %10, %16\n\tds_read_b128 %11, %17\n\tds_read_b128 %12, %18\n\ts_waitcnt lgkmcnt(0)"
               : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]), "+v"(o[6]), "=&v"(o[7]), "=&v"(o[8]), "=&v"(o[9]), "=&v"(o[10]), "=&v"(o[11]), "=&v"(o[12])
               : "v"(a[7]), "v"(a[8]), "v"(a[9]), "v"(a[10]), "v"(a[11]), "v"(a[12])
               : "memory");
}
__device__ __forceinline__ void st_lds_burst9(st_int4 (&o)[9], const unsigned (&a)[9])
{
  asm volatile("ds_read_b128 %0, %9\n\tds_read_b128 %1, %10\n\tds_read_b128 %2, %11\n\tds_read_b128 %3, %12\n\tds_read_b128 %4, %13\n\tds_read_b128 %5, %14\n\tds_read_b128 %6, %15\n\tds_read_b128 %7, %16\n\tds_read_b128 %8, %17\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]), "=&v"(o[8])
               : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8])
               : "memory");
}
template <int ME>
__device__ __forceinline__ void st_lds_burst(st_int4 (&o)[ME], const unsigned (&a)[ME])
{
  if constexpr (ME == 4) st_lds_burst4(o, a);
  else if constexpr (ME == 9) st_lds_burst9(o, a);
  else if constexpr (ME == 12) st_lds_burst12(o, a);
  else if constexpr (ME == 13) st_lds_burst13(o, a);
  else st_lds_burst16(o, a);
}
// The per-iteration burst of the compute wave: the ME window slots of the current row AND the two halves of an operand-ring
// record (32 bytes at `ra`) in one go, one wait.  The record's SECOND half (template id, tag) is read BEFORE the first one
// (operands): the loader writes operands first and the tag last, and LDS executes a wave's accesses in order, so a tag that
// reads as valid guarantees the operands read after it are the ones it belongs to.
__device__ __forceinline__ void st_lds_burst_row16(st_int4 (&o)[16], st_int4 &w0, st_int4 &w1, const unsigned (&a)[16], unsigned ra, unsigned rt)
{
  asm volatile("ds_read_b128 %0, %9\n\tds_read_b128 %1, %10\n\tds_read_b128 %2, %11\n\tds_read_b128 %3, %12\n\t"
               "ds_read_b128 %4, %13\n\tds_read_b128 %5, %14\n\tds_read_b128 %6, %15\n\tds_read_b128 %7, %16\n\tds_read_b128 %8, %17"
               : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]), "=&v"(w1)
               : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(rt)
               : "memory");
  asm volatile("ds_read_b128 %9, %18\n\tds_read_b128 %10, %19\n\tds_read_b128 %11, %20\n\tds_read_b128 %12, %21\n\t"
               "ds_read_b128 %13, %22\n\tds_read_b128 %14, %23\n\tds_read_b128 %15, %24\n\tds_read_b128 %16, %25\n\tds_read_b128 %17, %26\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]), "+v"(o[6]), "+v"(o[7]), "+v"(w1), "=&v"(o[8]), "=&v"(o[9]), "=&v"(o[10]),
                 "=&v"(o[11]), "=&v"(o[12]), "=&v"(o[13]), "=&v"(o[14]), "=&v"(o[15]), "=&v"(w0)
               : "v"(a[8]), "v"(a[9]), "v"(a[10]), "v"(a[11]), "v"(a[12]), "v"(a[13]), "v"(a[14]), "v"(a[15]), "v"(ra)
               : "memory");
}
__device__ __forceinline__ void st_lds_burst_row4(st_int4 (&o)[4], st_int4 &w0, st_int4 &w1, const unsigned (&a)[4], unsigned ra, unsigned rt)
{
  asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %7\n\tds_read_b128 %2, %8\n\tds_read_b128 %3, %9\n\tds_read_b128 %5, %11\n\t"
               "ds_read_b128 %4, %10\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(w0), "=&v"(w1)
               : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(ra), "v"(rt)
               : "memory");
}
__device__ __forceinline__ void st_lds_burst_row12(st_int4 (&o)[12], st_int4 &w0, st_int4 &w1, const unsigned (&a)[12], unsigned ra, unsigned rt)
{
  asm volatile("ds_read_b128 %0, %7\n\tds_read_b128 %1, %8\n\tds_read_b128 %2, %9\n\tds_read_b128 %3, %10\n\tds_read_b128 %4, %11\n\tds_read_b128 %5, %12\n\t"
               "ds_read_b128 %6, %13"
               : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(w1)
               : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(rt)
               : "memory");
  asm volatile("ds_read_b128 %7, %14\n\tds_read_b128 %8, %15\n\tds_read_b128 %9, %16\n\tds_read_b128 %10, %17\n\tds_read_b128 %11, %18\n\tds_read_b128 %12, %19\n\t"
               "ds_read_b128 %13, %20\n\ts_waitcnt lgkmcnt(0)"
               : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]), "+v"(w1), "=&v"(o[6]), "=&v"(o[7]), "=&v"(o[8]), "=&v"(o[9]), "=&v"(o[10]), "=&v"(o[11]),
                 "=&v"(w0)
               : "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]), "v"(a[10]), "v"(a[11]), "v"(ra)
               : "memory");
}
__device__ __forceinline__ void st_lds_burst_row13(st_int4 (&o)[13], st_int4 &w0, st_int4 &w1, const unsigned (&a)[13], unsigned ra, unsigned rt)
{
  asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\tds_read_b128 %2, %10\n\tds_read_b128 %3, %11\n\tds_read_b128 %4, %12\n\tds_read_b128 %5, %13\n\tds_read_b128 %6, %14\n\tds_read_b128 %7, %15"
               : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(w1)
               : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(rt)
               : "memory");
  asm volatile("ds_read_b128 %8, %15\n\tds_read_b128 %9, %16\n\tds_read_b128 %10, %17\n\tds_read_b128 %11, %18\n\tds_read_b128 %12, %19\n\tds_read_b128 %13, %20\n\tds_read_b128 %14, %21\n\ts_waitcnt lgkmcnt(0)"
               : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]), "+v"(o[6]), "+v"(w1), "=&v"(o[7]), "=&v"(o[8]), "=&v"(o[9]), "=&v"(o[10]), "=&v"(o[11]), "=&v"(o[12]),
                 "=&v"(w0)
               : "v"(a[7]), "v"(a[8]), "v"(a[9]), "v"(a[10]), "v"(a[11]), "v"(a[12]), "v"(ra)
               : "memory");
}
__device__ __forceinline__ void st_lds_burst_row9(st_int4 (&o)[9], st_int4 &w0, st_int4 &w1, const unsigned (&a)[9], unsigned ra, unsigned rt)
{
  asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %7\n\tds_read_b128 %2, %8\n\tds_read_b128 %3, %9\n\tds_read_b128 %4, %10\n\tds_read_b128 %5, %11"
               : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(w1)
               : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(rt)
               : "memory");
  asm volatile("ds_read_b128 %6, %11\n\tds_read_b128 %7, %12\n\tds_read_b128 %8, %13\n\tds_read_b128 %9, %14\n\tds_read_b128 %10, %15\n\ts_waitcnt lgkmcnt(0)"
               : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(w1), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]), "=&v"(o[8]), "=&v"(w0)
               : "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(ra)
               : "memory");
}
template <int ME>
__device__ __forceinline__ void st_lds_burst_row(st_int4 (&o)[ME], st_int4 &w0, st_int4 &w1, const unsigned (&a)[ME], unsigned ra, unsigned rt)
{
  // w1 <- 16 bytes at rt (the half that carries the tag: read FIRST), w0 <- 16 bytes at ra
  if constexpr (ME == 4) st_lds_burst_row4(o, w0, w1, a, ra, rt);
  else if constexpr (ME == 9) st_lds_burst_row9(o, w0, w1, a, ra, rt);
  else if constexpr (ME == 12) st_lds_burst_row12(o, w0, w1, a, ra, rt);
  else if constexpr (ME == 13) st_lds_burst_row13(o, w0, w1, a, ra, rt);
  else st_lds_burst_row16(o, w0, w1, a, ra, rt);
}
// agent-scope 8-byte load, complete on return (rare paths only: it drains this wave's stores as well)
__device__ __forceinline__ unsigned long long st_gload64_wait(const void *p)
{
  unsigned long long v;
  asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ unsigned st_gload32_wait(const void *p)
{
  unsigned v;
  asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}

template <bool FWD>
__device__ __forceinline__ hipx_int st_actual(long long q, hipx_int m)
{
  return FWD ? (hipx_int)q : (hipx_int)((long long)m - 1 - q);
}

__device__ __forceinline__ int st_strand_len(long long S, const StParams &P)
{
  if (S < 0 || S >= P.nstr) return 0;
  const long long rem = (long long)P.m - S * P.L;
  return (int)(rem < P.L ? rem : P.L);
}

// One compute role of a panel.  ROLE 0: the whole row (two-wave kernel: ME = 4, 13 or 16 entries).  SPLIT kernel (27-point class, ME 13 / 16):
// ROLE 1 = an F wave: the first ME - 4 entries of the row's list -- the far ones, staged by the loader long before they are needed --
// subtracted from the operand; the partial sum goes to the C wave through an ST_CQ-deep LDS ring.  ROLE 2 = the free-running C wave
// (st_lock_c is the lockstep one, used whenever the matrix allows it): the last
// (at most) 4 entries -- the previous line of the same plane and the row's own predecessor: the ones the NEXT lane is waiting
// for -- then the scale, the publish, the stores.  The lane-to-lane critical path of a line is then ~1/3 of the instructions.
// The left-to-right order of the subtractions is unchanged (F's end is C's start), so the sums are bit-identical.
// CWT > 0 (ROLE 0 only): the variable-coefficient form -- the row's coefficients and 1 / d are read from the coefficient ring
// (CWT doubles per record, staged by the loader next to the operand record) when the row starts, instead of from the template.
template <int KIND, int ME, int ROLE, bool PAIR = false, int CWT = 0>
__device__ __forceinline__ void st_compute_role(const StParams &P, st_lds_char *lds, const unsigned lds_base, volatile st_lds_int *s_prog_own, volatile st_lds_int *s_prog_c,
                                                volatile st_lds_int *s_ctl, unsigned int *err, const int lane, const unsigned panel, const long long S, const int len, double *t,
                                                const double *xold, double *xnew, const double omega, unsigned long long *stats, const int par,
                                                unsigned long long *fst = nullptr)
{
  constexpr bool FWD = (KIND == 0 || KIND == 3);
  static_assert(CWT == 0 || (ROLE == 0 && KIND <= 2 && CWT >= ME + 1 && CWT % 2 == 0), "variable coefficients: whole-row compute wave, sweeps without old-value lists");
  const hipx_int m = P.m, L = P.L;
  const int      RQ = CWT ? P.rq : ST_RQ;  // depth of the operand ring (a power of two)
  const int      stride = ROLE == 1 ? 2 : 1;  // the two F waves take the even and the odd rows of every strand (row sums do not depend on each other)
  int       p = ROLE == 1 ? par : 0, ostart = 0, ocnt = 0, dtab = 0, cur_tid = -1, setp = 0;  // setp: the position apos[] / sa[] are set for; dtab: byte offset of the template's entry list
  bool      have = false;
  double    s0 = 0.0, rb = 0.0, idiag = 0.0, mdiag = 0.0;
  st_int4   tq0 = {0, 0, 0, 0}, tq1 = {0, 0, 0, 0}, tq2 = {0, 0, 0, 0};  // PAIR, kinds 0 / 3: the three pairs of t before the current one
  double    hold_out = 0.0, hold_sum = 0.0;  // PAIR (kernels for aligned strands): the even row of a pair, stored together with the odd one
  unsigned  sa[ME];    // LDS byte address of the slot of entry j for the row at position p
  unsigned  wrow[ME];  // ... of slot 0 of its window row (the null slot for a padding entry)
  int       inc[ME];   // what the tag moves by per row of this wave: 1 (2 for an F wave), or 0 for a padding entry (ST_NULLTAG & 15 == 0: its slot is the null slot itself)
  int       apos[ME];  // the tag the slot must carry: position + rotation of the window row (the slot is apos & 15)
  double    cf[ME];    // coefficient
#pragma unroll
  for (int j = 0; j < ME; j++) {
    sa[j] = wrow[j] = lds_base + (unsigned)P.off_null;
    inc[j]  = 0;
    apos[j] = ST_NULLTAG;
    cf[j]   = 0.0;
  }
  // the per-row record {operand a | partial sum, old value}{template id, tag}: from the loader's operand ring, or (ROLE 2) from the
  // two-deep hand-over ring the F wave fills
  const unsigned rec_base = lds_base + (unsigned)(P.off_rowq + 32 * RQ * lane);
  const int      crec_off = CWT ? P.off_cring + 8 * CWT * RQ * lane : 0;  // this lane's coefficient records (byte offset in the LDS region)
  const unsigned cq_base  = lds_base + (unsigned)(P.off_cq + 16 * ST_CQ * lane);  // SPLIT: {partial sum, template id, tag}, ST_CQ deep
  const int      rq_rot   = ST_ROT * lane;
  unsigned       pubrow[ST_NB];  // this lane's own window row in band b (byte address of slot 0), ~0u: the band has none for it
  int            pubrot[ST_NB];  // ... and its slot rotation
#pragma unroll
  for (int b = 0; b < ST_NB; b++) {
    pubrow[b] = ~0u;
    pubrot[b] = 0;
    if (b < P.nbands) {
      const int u = lane - P.band[b].dsmin;
      if (u >= 0 && u < 64 + P.band[b].width - 1) {
        pubrow[b] = lds_base + (unsigned)(P.off_win + 16 * ST_WP * (P.band[b].rowbase + u));
        pubrot[b] = ST_ROT * (P.band[b].rowbase + u);
      }
    }
  }
  bool pubany[ST_NB];  // wave-uniform: does any lane publish into band b at all?  (the far bands of a stencil: no -- skip the block)
#pragma unroll
  for (int b = 0; b < ST_NB; b++) pubany[b] = __any(pubrow[b] != ~0u);
  unsigned  st_iters = 0, st_rowwait = 0, st_depwait = 0, st_fallback = 0;  // HIPX_SOR_DEBUG statistics (stats != nullptr)
  unsigned  f_far = 0, f_full = 0, f_fire = 0, f_row = 0;
  unsigned  st_nfast = 0, st_nslow = 0, st_nsetup = 0;                     // iterations in which ANY lane took the path
  long long st_cburst = 0, st_cfast = 0;                                   // shader clocks spent in the burst / in the fast finish
  const long long st_t0 = stats ? (long long)wall_clock64() : 0;
  const long long st_c0 = stats ? (long long)clock64() : 0;
  if (stats && lane == 0) stats[16 + 4 * (size_t)panel] = (unsigned long long)st_t0;
  long long t0 = 0;
  // the template of the row at position p has changed (first row, boundary rows): entry table -> registers
  auto load_template = [&](int tnew) __attribute__((always_inline)) {
    st_int4 ti = {0, 0, 0, 0}, dg = {0, 0, 0, 0};
    if (ROLE != 1) {
      ti = st_ld4(lds, P.off_tinfo + 16 * tnew);
      dg = st_ld4(lds, P.off_tdiag + 16 * tnew);
    }
    dtab   = ROLE == 0 ? P.off_dep + 16 * ti.x : (ROLE == 1 ? P.off_depF : P.off_depC) + 16 * ME * tnew;
    ostart = ti.z;
    ocnt   = ti.w;
    st_int4  e[ME];
    unsigned ea[ME];
#pragma unroll
    for (int j = 0; j < ME; j++) ea[j] = lds_base + (unsigned)(dtab + 16 * j);
    st_lds_burst<ME>(e, ea);
#pragma unroll
    for (int j = 0; j < ME; j++) {
      const bool null = e[j].x == ST_NULLPK;
      const int wr = lane + (e[j].x >> 16);  // window row
      wrow[j] = lds_base + (unsigned)(null ? P.off_null : P.off_win + 16 * ST_WP * wr);
      inc[j]  = null ? 0 : stride;
      apos[j] = null ? ST_NULLTAG : p + (int)(short)(e[j].x & 0xffff) + ST_ROT * wr;
      sa[j]   = wrow[j] + (unsigned)((apos[j] & (ST_WP - 1)) << 4);
      cf[j]   = st_dbl(e[j].z, e[j].w);
    }
    idiag   = st_dbl(dg.x, dg.y);
    mdiag   = st_dbl(dg.z, dg.w);
    cur_tid = tnew;
    setp    = p;
  };
  // operands of the row at position p have arrived (w0 = {a, old value}, w1 = {template id, tag})
  auto start_row = [&](const st_int4 &w0, const st_int4 &w1) __attribute__((always_inline)) {
    s0 = st_dbl(w0.x, w0.y);
    rb = st_dbl(w0.z, w0.w);
    if (w1.x != cur_tid) load_template(w1.x);
    else if (setp != p) {  // same template, the wave's next row of this strand (p = setp + stride): every tag and slot moves on
#pragma unroll
      for (int j = 0; j < ME; j++) {
        apos[j] += inc[j];  // (a padding entry stays where it is)
        sa[j] = wrow[j] + (unsigned)((apos[j] & (ST_WP - 1)) << 4);
      }
      setp = p;
    }
    if constexpr (CWT > 0) {
      // the record was written before the operand record's tag (the loader's LDS writes execute in order) and its slot is not
      // reused before this lane's progress passes p: valid now.  Padding entries carry 0.0, like the templates' null entries.
      st_int4   cr[CWT / 2];
      const int cb = crec_off + 8 * CWT * ((p + rq_rot) & (RQ - 1));
#pragma unroll
      for (int h = 0; h < CWT / 2; h++) cr[h] = st_ld4(lds, cb + 16 * h);
#pragma unroll
      for (int j = 0; j < ME; j++) cf[j] = (j & 1) ? st_dbl(cr[j >> 1].z, cr[j >> 1].w) : st_dbl(cr[j >> 1].x, cr[j >> 1].y);
      idiag = (ME & 1) ? st_dbl(cr[ME >> 1].z, cr[ME >> 1].w) : st_dbl(cr[ME >> 1].x, cr[ME >> 1].y);
    }
    have = true;
    if (KIND == 4) {  // aij.c:1984-1990: the lower part and the diagonal use OLD values, in row order, first
      const hipx_int r = st_actual<FWD>(S * L + p, m);
      for (int q2 = 0; q2 < ocnt; q2++) {
        const st_int4 oe = st_ld4(lds, P.off_old + 16 * (ostart + q2));
        s0 -= st_dbl(oe.z, oe.w) * xold[(long long)r + oe.x];
      }
    }
  };
  // Two panels share a CU and their compute waves may share a SIMD: a wave that spins on values that are not there yet takes
  // issue cycles from one that has work (measured: panels next to a spinning neighbour ran at 2.9 us per row instead of 1.4).
  // So the wave runs at raised priority and, after an iteration in which none of its lanes finished a row, sleeps briefly.
  __builtin_amdgcn_s_setprio(3);
  int idle = 0;
  for (unsigned it = 1;; it++) {
    const bool active = p < len;
    if (!__any(active)) break;
    st_iters++;
    const int  p_before    = p;
    const bool have_before = have;
    int        dbg_diff = 0x7ffffff, dbg_mask = 0;
    const bool dbg_on = stats && P.trace_panel == (int)panel && lane == P.trace_lane;
    long long  dbg_c0 = dbg_on ? (long long)clock64() : 0, dbg_c1 = 0, dbg_c2 = 0, dbg_c3 = 0;
    asm volatile("" ::: "memory");  // other waves have written LDS since the last iteration: re-read, do not reuse
    if (active) {
      // ONE burst per iteration: the ME slots of the current row and the operand record of the NEXT row (of the current one
      // while it is still missing); one wait
      st_int4        sl[ME], w0, w1;
      const int      qr = have ? p + stride : p;
      const unsigned ra = rec_base + (unsigned)(32 * ((qr + rq_rot) & (RQ - 1)));
      const unsigned rt = ROLE == 2 ? cq_base + (unsigned)(16 * (qr & (ST_CQ - 1))) : ra + 16;
      const long long c_b0 = stats ? (long long)clock64() : 0;
      st_lds_burst_row<ME>(sl, w0, w1, sa, ra, rt);
      if (ROLE == 2) {  // w1 is F's record {sum, id, tag}, w0 the loader's {a, old value} of the same row (valid while C has not passed it:
                        // the loader recycles ring slots by C's progress): bring both into the usual shape
        w0.x = w1.x;
        w0.y = w1.y;
        w1.x = w1.z;
        w1.y = w1.w;
      }
      if (__any(!have)) dbg_mask |= 1 << 29;
      if (stats) st_cburst += (long long)clock64() - c_b0;
      if (stats && __any(!have)) st_nsetup++;
      if (!have) {
        if (w1.y == p) start_row(w0, w1);  // (the slots read above belonged to no row: compute in the next iteration)
        else {
          st_rowwait++;
          if (ROLE == 1 && fst && lane == 32) f_row++;
        }
      } else {
        const long long q = S * L + p;
        const hipx_int  r = st_actual<FWD>(q, m);
        int             diff = 0;  // OR of (tag - expected): 0 = all there; negative = at least one not produced yet (wait, nothing
                                   // else to find out); positive = a slot has moved on (rare: the value comes from memory)
#pragma unroll
        for (int j = 0; j < ME; j++) diff |= sl[j].z - apos[j];
        if (ROLE == 1 && fst && lane == 32) {  // F-wave statistics (HIPX_SOR_DEBUG): lane 32's view
          if (diff < 0) f_far++;
          else if (p - (int)s_prog_c[lane] > ST_CQ - 1) f_full++;
          else if (diff == 0) f_fire++;
        }
        if (ROLE == 1 && p - (int)s_prog_c[lane] > ST_CQ - 1) diff |= (int)0x80000000;  // the hand-over slot still holds row p - ST_CQ: wait for C
        if (dbg_on) dbg_c1 = (long long)clock64();  // (moves the burst/compare boundary to here: burst + whatever the !have lanes did + the tag compare)
        if (__any(diff > 0)) dbg_mask |= 1 << 30;
        if (stats && P.trace_panel == (int)panel && lane == P.trace_lane) {
          dbg_diff = diff;
#pragma unroll
          for (int j = 0; j < ME; j++) dbg_mask |= (sl[j].z < apos[j] ? 1 : 0) << j;
        }
        // the row's values are all there: subtraction chain, scale, publish, move on to the next row
        auto finish = [&](const double (&val)[ME]) __attribute__((always_inline)) {
          double sum = s0;
#pragma unroll
          for (int j = 0; j < ME; j++) sum -= cf[j] * val[j];  // left to right (PetscSparseDenseMinusDot); null entries subtract +0.0
          if (ROLE == 1) {  // hand the partial sum and the template id to the C wave
            const long long bs = __double_as_longlong(sum);
            *reinterpret_cast<volatile st_lds_int4 *>((st_lds_char *)(size_t)(cq_base + (unsigned)(16 * (p & (ST_CQ - 1))))) =
              st_int4{(int)(unsigned)bs, (int)(unsigned)((unsigned long long)bs >> 32), cur_tid, p};  // one 16-byte store: sum and tag arrive together
            p += stride;
            have = false;
            asm volatile("" ::: "memory");
            if (p < len && w1.y == p) start_row(w0, w1);
            return;
          }
          double       out;
          const double tsum = sum;  // what goes to t (kinds 0 and 3)
          if (KIND == 0) {
            if (!PAIR) t[r] = sum;
            out = sum * idiag;
          } else if (KIND == 1) {
            out = (1 - omega) * rb + sum * idiag;
          } else if (KIND == 2) {
            out = sum * idiag;
          } else if (KIND == 3) {
            if (!PAIR) t[r] = sum;
            for (int e2 = 0; e2 < ocnt; e2++) {  // upper part: old values (aij.c:1973-1976)
              const st_int4 oe = st_ld4(lds, P.off_old + 16 * (ostart + e2));
              sum -= st_dbl(oe.z, oe.w) * xold[(long long)r + oe.x];
            }
            out = (1. - omega) * rb + sum * idiag;
          } else {
            out = (1. - omega) * rb + (sum + mdiag * rb) * idiag;
          }
#pragma unroll
          for (int b = 0; b < ST_NB; b++)
            if (pubany[b] && pubrow[b] != ~0u)  // the tag is the position plus the row's rotation; so is the slot
              *reinterpret_cast<volatile st_lds_int4 *>((st_lds_char *)(size_t)(pubrow[b] + (unsigned)(((p + pubrot[b]) & (ST_WP - 1)) << 4))) = st_pack_slot(out, p + pubrot[b]);
          if (PAIR) {
            // rows go to memory in pairs (even position, next one): one 16-byte store instead of two 8-byte ones.  A store whose lanes
            // all hit different cache lines costs the CU's request path ~190 ns (scripts/diag/ta_probe.hip) whatever its width, and
            // that path -- shared with the loaders' polls -- is what the sweep waits for.  L and m are multiples of 8 here: strands
            // have even lengths and a pair is 16-byte aligned (backward sweeps: the pair's rows are in descending memory order).
            if (p & 1) {
              const double    lo_v = FWD ? hold_out : out, hi_v = FWD ? out : hold_out;
              const long long b0 = __double_as_longlong(lo_v), b1 = __double_as_longlong(hi_v);
              const st_int4   vo = {(int)(unsigned)b0, (int)(unsigned)((unsigned long long)b0 >> 32), (int)(unsigned)b1, (int)(unsigned)((unsigned long long)b1 >> 32)};
              double         *dst = xnew + (FWD ? r - 1 : r);
              asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(vo) : "memory");  // each 8-byte half is its own ready flag
              if (KIND == 0 || KIND == 3) {  // t: a line of 8 rows at a time, four consecutive plain stores (see st_lock_c)
                const long long c0 = __double_as_longlong(hold_sum), c1 = __double_as_longlong(tsum);
                const st_int4   vs = {(int)(unsigned)c0, (int)(unsigned)((unsigned long long)c0 >> 32), (int)(unsigned)c1, (int)(unsigned)((unsigned long long)c1 >> 32)};
                if ((p & 7) == 7) {
                  st_int4 *tp = reinterpret_cast<st_int4 *>(t + (r - 7));
                  tp[0] = tq0;
                  tp[1] = tq1;
                  tp[2] = tq2;
                  tp[3] = vs;
                }
                tq0 = tq1;
                tq1 = tq2;
                tq2 = vs;
              }
            } else {
              hold_out = out;
              hold_sum = tsum;
            }
          } else sor_publish(xnew + r, out);
          if (stats && p == 0 && (lane == 0 || lane == 63)) stats[16 + 4 * (size_t)panel + (lane ? 2 : 1)] = (unsigned long long)wall_clock64();
          if (stats && P.trace_panel == (int)panel && P.trace_rows)  // HIPX_SOR_TRACE_PANEL: completion time of every row of this panel
            stats[16 + 4 * (size_t)P.npanels + (size_t)lane * (size_t)L + (size_t)p] = (unsigned long long)wall_clock64();
          p++;
          have = false;
          asm volatile("" ::: "memory");
          if (p < len && w1.y == p) start_row(w0, w1);  // the next row's operands came with this iteration's burst
        };
        if (stats && __any(diff == 0)) st_nfast++;
        if (stats && __any(diff > 0)) st_nslow++;
        if (!diff) {  // the common case: straight from the registers the burst filled
          double val[ME];
#pragma unroll
          for (int j = 0; j < ME; j++) val[j] = st_dbl(sl[j].x, sl[j].y);
          const long long c_f0 = stats ? (long long)clock64() : 0;
          if (dbg_on) dbg_c2 = c_f0;
          finish(val);
          if (stats) st_cfast += (long long)clock64() - c_f0;
          if (dbg_on) dbg_c3 = (long long)clock64();
        } else if (diff < 0) {
          st_depwait++;
          if (stats && P.trace_panel == (int)panel && P.trace_rows) {  // which entry is late?  (first one in arithmetic order), all lanes together + lane 0 alone
            int jf = 0;  // (selects, not indexed reads: a register array indexed at run time goes to scratch)
#pragma unroll
            for (int j = ME - 1; j >= 0; j--)
              if (sl[j].z < apos[j]) jf = j;
            unsigned long long *h = stats + 16 + 4 * (size_t)P.npanels + 64 * (size_t)L + 2 * 4096 * 8;
            atomicAdd(&h[jf], 1ull);
            if (lane == 0) atomicAdd(&h[16 + jf], 1ull);
            if (lane == 32) atomicAdd(&h[32 + jf], 1ull);
          }
        } else {  // no tag behind, at least one ahead: the slot has moved on (this lane fell far behind its producer)
          bool   ok = true;
          double val[ME];
#pragma unroll
          for (int j = 0; j < ME; j++) {
            val[j] = st_dbl(sl[j].x, sl[j].y);
            if (sl[j].z < apos[j]) ok = false;
            else if (sl[j].z > apos[j]) {
              const int                elo = st_ld4(lds, dtab + 16 * j).y;  // logical row offset of the entry
              const unsigned long long v   = st_gload64_wait(xnew + st_actual<FWD>(q + elo, m));
              st_fallback++;
              if (v == SOR_SENTINEL) ok = false;
              else val[j] = __longlong_as_double((long long)v);
            }
          }
          if (ok) finish(val);
          else st_depwait++;
        }
      }
    }
    s_prog_own[lane] = p;
    if (ROLE == 1 && s_ctl[1]) break;  // C has given up (bounded wait)
    if (stats && P.trace_panel == (int)panel && lane == P.trace_lane && (int)it >= P.trace_it0 && (int)it < P.trace_it0 + 4096) {  // iteration log of one lane
      unsigned long long *ev = stats + 16 + 4 * (size_t)P.npanels + 64 * (size_t)L + 4096 * 8 + (size_t)((int)it - P.trace_it0) * 8;
      ev[0] = (unsigned long long)wall_clock64();
      ev[1] = (unsigned long long)it;
      ev[2] = (unsigned long long)p_before;
      ev[3] = (unsigned long long)p;
      ev[4] = (unsigned long long)((have_before ? 1 : 0) | (have ? 2 : 0));
      ev[5] = (unsigned long long)(unsigned)dbg_diff | ((unsigned long long)(unsigned)(dbg_c1 - dbg_c0) << 32);       // burst
      ev[6] = (unsigned long long)(unsigned)(dbg_c2 - dbg_c1) | ((unsigned long long)(unsigned)(dbg_c3 - dbg_c2) << 32);  // compare | finish
      ev[7] = (unsigned long long)(unsigned)dbg_mask | ((unsigned long long)(unsigned)((long long)clock64() - dbg_c0) << 32);  // whole iteration so far
    }
    if (!__any(p != p_before || have != have_before)) {
      idle = idle < 4 ? idle + 1 : 4;
      if (idle >= 2) __builtin_amdgcn_s_sleep(2);   // ~128 clocks; the loader pass that can change anything takes thousands
    } else idle = 0;
    if (ROLE != 1 && (it & 0x3ff) == 0) {  // bounded wait: elapsed wall-clock time, and a global abort word so one stuck panel ends the launch
      const long long now = (long long)wall_clock64();
      if (!t0) t0 = now;
      const unsigned abort_word = st_gload32_wait(err);
      if (abort_word || now - t0 > SOR_SPIN_TICKS) {
        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
  __builtin_amdgcn_s_setprio(0);
  if (ROLE == 1) {
    if (fst && lane == 32) {
      atomicAdd(&fst[0], (unsigned long long)st_iters);
      atomicAdd(&fst[1], (unsigned long long)f_row);
      atomicAdd(&fst[2], (unsigned long long)f_far);
      atomicAdd(&fst[3], (unsigned long long)f_full);
      atomicAdd(&fst[4], (unsigned long long)f_fire);
    }
    return;
  }
  if (lane == 0) s_ctl[1] = 1;
  if (stats) {
    for (int o = 32; o > 0; o >>= 1) {  // the finish-path clocks of the lane that finished most often
      const long long other = __shfl_xor(st_cfast, o);
      st_cfast = other > st_cfast ? other : st_cfast;
    }
    atomicAdd(&stats[1], (unsigned long long)st_rowwait);
    atomicAdd(&stats[2], (unsigned long long)st_depwait);
    atomicAdd(&stats[5], (unsigned long long)st_fallback);
    atomicAdd(&stats[7], (unsigned long long)len);
    if (lane == 0) {
      atomicAdd(&stats[0], (unsigned long long)st_iters);
      atomicAdd(&stats[9], (unsigned long long)st_nfast);
      atomicAdd(&stats[10], (unsigned long long)st_nslow);
      atomicAdd(&stats[11], (unsigned long long)st_nsetup);
      atomicAdd(&stats[12], (unsigned long long)st_cburst);
      atomicAdd(&stats[13], (unsigned long long)st_cfast);
      atomicAdd(&stats[14], (unsigned long long)((long long)clock64() - st_c0));
      atomicAdd(&stats[6], (unsigned long long)((long long)wall_clock64() - st_t0));
      atomicAdd(&stats[8], 1ull);
      stats[16 + 4 * (size_t)panel + 3] = (unsigned long long)wall_clock64();
    }
  }
}

// lane i <- lane i-1 (lane 0 keeps `first`): one DPP wavefront shift per 32-bit half, no LDS
__device__ __forceinline__ double st_from_prev_lane(double first, double v)
{
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(first), __double2loint(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(first), __double2hiint(v), 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// The C wave of the split kernel in LOCKSTEP (forward zero-guess sweep).  For the box stencils the near part of a row's list is
// a subset of {previous line: x-1, x, x+1; own line: x-1}, in that order: lane i needs rows p-1, p, p+1 of lane i-1 and its own
// row p-1.  So the wave runs with a fixed skew of two rows per lane -- in iteration k lane i is at row k - 2 i -- every lane keeps
// its last three results in registers, and the neighbour's three arrive with a DPP shift: no window slots, no tags, no publish
// into LDS for the near entries, nothing lane-divergent in the iteration.  What still comes through LDS: the partial sum of the
// far entries from the F waves (tagged hand-over ring, as before) and, for lane 0 only, the line before the panel's first one
// (another panel's: staged into the window by the loader, tagged).  If any lane's inputs are missing the whole wave retries --
// in the free-running form a late lane stalls every lane after it anyway.  The arithmetic is the same left-to-right chain
// (absent entries subtract +0.0 * +0.0 like the padding entries of the other kernels): bit-identical.
// Eligibility is decided by the host (strand_build): every template's list = far entries (strands of other panels: delta
// <= -64), then near entries from the canonical four in canonical order.
template <bool PAIR>
__device__ __forceinline__ void st_lock_c(const StParams &P, const unsigned lds_base, volatile st_lds_int *s_prog, volatile st_lds_int *s_ctl, unsigned int *err, const int lane,
                                          const long long S, const int len, double *t, double *xnew, unsigned long long *stats, const unsigned panel)
{
  unsigned st_iters = 0, st_stall = 0;  // HIPX_SOR_DEBUG (stats != nullptr: the DBG instantiation only)
  if (stats && lane == 0) stats[16 + 4 * (size_t)panel] = (unsigned long long)wall_clock64();
  const unsigned cq_base = lds_base + (unsigned)(P.off_cq + 16 * ST_CQ * lane);
  const unsigned null_a  = lds_base + (unsigned)P.off_null;
  const unsigned up_row  = lds_base + (unsigned)(P.off_win + 16 * ST_WP * P.lock_wrow);
  const int      up_rot  = ST_ROT * P.lock_wrow;
  const unsigned lock_b  = lds_base + (unsigned)P.off_lock;
  const unsigned diag_b  = lds_base + (unsigned)P.off_tdiag;
  const long long r0     = S * (long long)P.L;
  int       p = -2 * lane, cur_tid = -1, mask = 0, idle = 0;
  double    c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0, idiag = 0.0, H1 = 0.0, H2 = 0.0, H3 = 0.0, psum = 0.0;
  st_int4   tq0 = {0, 0, 0, 0}, tq1 = {0, 0, 0, 0}, tq2 = {0, 0, 0, 0};  // the three pairs of t before the current one (one line = four pairs)
  long long t0 = 0;
  __builtin_amdgcn_s_setprio(3);
  for (unsigned it = 1;; it++) {
    if (!__any(p < len)) break;
    const bool active = p >= 0 && p < len;
    asm volatile("" ::: "memory");  // other waves have written LDS since the last iteration
    // one burst: the F waves' record of row p {sum, template id, tag}; lane 0: positions p-1, p, p+1 of the line before the panel
    const int e0 = p - 1 + up_rot;  // the tag position p-1 carries in that window row (slot = tag & 15)
    unsigned  a[4];
    st_int4   o[4];
    a[0] = cq_base + (unsigned)(16 * (p & (ST_CQ - 1)));
    a[1] = lane == 0 ? up_row + (unsigned)((e0 & (ST_WP - 1)) << 4) : null_a;
    a[2] = lane == 0 ? up_row + (unsigned)(((e0 + 1) & (ST_WP - 1)) << 4) : null_a;
    a[3] = lane == 0 ? up_row + (unsigned)(((e0 + 2) & (ST_WP - 1)) << 4) : null_a;
    st_lds_burst4(o, a);
    bool ok = !active || o[0].w == p;
    if (active && o[0].w == p && o[0].z != cur_tid) {  // the row's template differs from the last row's (strand ends): coefficients -> registers
      unsigned b[4];
      st_int4  q[4];
      b[0] = lock_b + (unsigned)(48 * o[0].z);
      b[1] = b[0] + 16;
      b[2] = b[0] + 32;
      b[3] = diag_b + (unsigned)(16 * o[0].z);
      st_lds_burst4(q, b);
      c0      = st_dbl(q[0].x, q[0].y);
      c1      = st_dbl(q[0].z, q[0].w);
      c2      = st_dbl(q[1].x, q[1].y);
      c3      = st_dbl(q[1].z, q[1].w);
      mask    = q[2].x;
      idiag   = st_dbl(q[3].x, q[3].y);
      cur_tid = o[0].z;
    }
    if (lane == 0 && active && ok) {  // the entries of the line before: there only if the loader has staged them
      if (((mask & 1) && o[1].z != e0) || ((mask & 2) && o[2].z != e0 + 1) || ((mask & 4) && o[3].z != e0 + 2)) ok = false;
    }
    if ((it & 0x3ff) == 0) {  // bounded wait: elapsed wall-clock time, and the global abort word
      const long long now = (long long)wall_clock64();
      if (!t0) t0 = now;
      const unsigned abort_word = st_gload32_wait(err);
      if (abort_word || now - t0 > SOR_SPIN_TICKS) {
        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
    st_iters++;
    if (__any(!ok)) {  // a record or a staged value is not there yet: the whole wave tries again
      st_stall++;
      idle = idle < 4 ? idle + 1 : 4;
      if (idle >= 2) __builtin_amdgcn_s_sleep(1);
      continue;
    }
    idle = 0;
    const double u1 = st_from_prev_lane(st_dbl(o[3].x, o[3].y), H1);  // previous line, position p+1 (lane i-1 is at row p+2: its last result)
    const double u2 = st_from_prev_lane(st_dbl(o[2].x, o[2].y), H2);  // ... position p
    const double u3 = st_from_prev_lane(st_dbl(o[1].x, o[1].y), H3);  // ... position p-1
    double       sum = st_dbl(o[0].x, o[0].y);
    sum -= c0 * ((mask & 1) ? u3 : 0.0);
    sum -= c1 * ((mask & 2) ? u2 : 0.0);
    sum -= c2 * ((mask & 4) ? u1 : 0.0);
    sum -= c3 * ((mask & 8) ? H1 : 0.0);
    const double out = sum * idiag;
    if (stats && p == 0 && active && (lane == 0 || lane == 63)) stats[16 + 4 * (size_t)panel + (lane ? 2 : 1)] = (unsigned long long)wall_clock64();
    if (PAIR) {
      // rows leave in pairs (even row, next one): one 16-byte store each for t and x instead of two 8-byte ones -- every lane's store is
      // its own cache line, and the CU's address pipeline is what the sweep waits for.  The skew is even, so "p is odd" is the same
      // in every lane; strands have even lengths here (L, m multiples of 8): both rows of a pair are active or neither is.
      if ((p & 1) && active) {
        const long long bs0 = __double_as_longlong(psum), bs1 = __double_as_longlong(sum), bo0 = __double_as_longlong(H1), bo1 = __double_as_longlong(out);
        const st_int4   vs = {(int)(unsigned)bs0, (int)(unsigned)((unsigned long long)bs0 >> 32), (int)(unsigned)bs1, (int)(unsigned)((unsigned long long)bs1 >> 32)};
        const st_int4   vo = {(int)(unsigned)bo0, (int)(unsigned)((unsigned long long)bo0 >> 32), (int)(unsigned)bo1, (int)(unsigned)((unsigned long long)bo1 >> 32)};
        if ((p & 7) == 7) {
          // t: nobody waits for it, so a whole line of 8 rows (this pair and the three before it, kept in registers) goes out as four
          // consecutive plain stores, which the cache combines: 68 instead of 180 ns of the CU's request path each
          // (profiles/r02_request_path_probe.txt, modes 10 / 12).  Strand lengths are multiples of 8 here.
          st_int4 *tp = reinterpret_cast<st_int4 *>(t + (r0 + p - 7));
          tp[0] = tq0;
          tp[1] = tq1;
          tp[2] = tq2;
          tp[3] = vs;
        }
        tq0 = tq1;
        tq1 = tq2;
        tq2 = vs;
        // (s_nop: a store of more than 8 bytes must not be followed at once by a write to its data registers, and the compiler's hazard
        // recognizer does not look into asm statements)
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(xnew + (r0 + p - 1)), "v"(vo) : "memory");  // each 8-byte half is its own ready flag
      }
      psum = sum;
    } else if (active) {
      t[r0 + p] = sum;
      sor_publish(xnew + (r0 + p), out);
    }
    H3 = H2;
    H2 = H1;
    H1 = active ? out : 0.0;
    p++;
    s_prog[lane] = p > 0 ? p : 0;
  }
  __builtin_amdgcn_s_setprio(0);
  if (lane == 0) s_ctl[1] = 1;
  if (stats && lane == 0) {
    stats[16 + 4 * (size_t)panel + 3] = (unsigned long long)wall_clock64();
    atomicAdd(&stats[0], (unsigned long long)st_iters);
    atomicAdd(&stats[2], (unsigned long long)st_stall * 64ull);  // (reported as lane-iterations waiting for dependencies)
    atomicAdd(&stats[8], 1ull);
    atomicAdd(&stats[6], stats[16 + 4 * (size_t)panel + 3] - stats[16 + 4 * (size_t)panel]);
  }
  if (stats) atomicAdd(&stats[7], (unsigned long long)len);
}

// The loader wave of a panel: operands of the own strands into the operand ring, far strands into the window.
// CWT > 0: variable coefficients -- the loader also stages every row's coefficient record (CWT doubles from the stream cs, which is
// indexed by LOGICAL position: ascending addresses in both sweep directions) into the coefficient ring, before the operand record's tag.
template <int KIND, bool ALIGNED, bool SPLIT, int CWT = 0>
__device__ __forceinline__ void st_loader_role(const StParams &P, st_lds_char *lds, volatile st_lds_int *s_lead, volatile st_lds_int *s_trail, volatile st_lds_int *s_ctl, const int lane,
                                               const unsigned panel, const long long S0, const int cnt, const long long S, const int len, const unsigned char *__restrict__ tid,
                                               const double *asrc,
                                               const double *xold, const double *xnew, unsigned long long *stats, const double *__restrict__ cs = nullptr)
{
  constexpr bool FWD     = (KIND == 0 || KIND == 3);
  constexpr bool NEEDOLD = (KIND == 1 || KIND == 3 || KIND == 4);  // the row's own old value
  constexpr int  VSB     = CWT == 0 ? ST_SB : (CWT <= 6 ? 8 : 4);  // rows staged per pass (the coefficient records of a pass sit in registers: VSB * CWT doubles)
  // the 27-point class with streamed coefficients: its ring is short (LDS), so rows are staged one by one as slots come free (not in
  // whole groups of 4, which lets the ring run empty before it is refilled) and the operands take the per-row loads of the unaligned form
  constexpr bool FINE    = CWT > 6;
  typedef double st_dbl2 __attribute__((ext_vector_type(2)));
  const hipx_int m = P.m, L = P.L;
  const int      RQ = CWT ? P.rq : ST_RQ;
  int rqf = 0;  // next position of the own strand whose operands are to be staged
  unsigned st_pass = 0, st_idle = 0;
  int sf[2 * ST_NB];
  int want[2 * ST_NB];  // rows to ask for per pass: what the last pass got + 2 (a follower gets 1-3 rows per pass from its producer;
                        // asking for 8 every time issues 5-7 loads per duty that come back as sentinels)
#pragma unroll
  for (int d = 0; d < 2 * ST_NB; d++) {
    sf[d]   = 0;
    want[d] = ST_SB;
  }
  // which far duties does this lane have at all (fixed for the panel)?  The second half (d >= ST_NB) is empty for most patterns:
  // its loads, registers and LDS writes are skipped as a whole
  bool dvalid[2 * ST_NB];
  bool any_hi = false;
#pragma unroll
  for (int d = 0; d < 2 * ST_NB; d++) {
    const int b = d >> 1, which = d & 1;
    dvalid[d]   = false;
    if (b < P.nbands) {
      const int       w = P.band[b].width, u = lane + 64 * which;
      const long long strand = S0 + u + P.band[b].dsmin;
      dvalid[d] = u < 64 + w - 1 && (strand < S0 || strand >= S0 + cnt) && strand >= 0 && strand < P.nstr;
    }
    if (d >= ST_NB && dvalid[d]) any_hi = true;
  }
  const bool wave_hi = __any(any_hi);
  for (;;) {
    if (s_ctl[1]) break;
    bool      issued = false;
    // SPLIT: s_lead = the two F waves' progress arrays (even rows, odd rows): every row below the smaller one has been consumed
    const int myp    = SPLIT ? min((int)s_lead[lane], (int)s_lead[64 + lane]) : (int)s_lead[lane];
    // (a) operands of the own strand: up to ST_SB positions from rqf on, in groups of 4, as far as the ring has room.  (Whole
    // groups of 8 only -- the first version -- capped the strand at 8 rows per two passes: ~0.9 us per row.)
    int           nrow = 0;
    double        va[ST_SB], vb[ST_SB];
    unsigned char vt[ST_SB];
    unsigned      tb[2] = {0, 0};  // ALIGNED: the template ids of each group of 4 as loaded; unpacked when they land
    const int ring_tail = SPLIT ? (int)s_trail[lane] : myp;  // SPLIT: the C wave still reads the row's old value from the ring
    if (rqf < len) {
      int room = FINE ? (ring_tail + RQ - rqf) : ((ring_tail + RQ - rqf) & ~3);
      if (room > VSB) room = VSB;
      nrow = (len - rqf) < room ? (len - rqf) : room;
    }
    if (nrow > 0) {
      issued = true;
      const long long q0 = S * L + rqf;
      if (ALIGNED && !FINE) {  // L, m multiples of 8 and 16-byte aligned vectors: every group of 4 rows is one aligned 32-byte run
        typedef double dbl2 __attribute__((ext_vector_type(2)));
        if (nrow > 4) {  // both groups: their eight template ids are eight consecutive bytes (4-byte aligned): one request instead of two
          typedef unsigned uint2a __attribute__((ext_vector_type(2), aligned(4)));
          const uint2a t8 = *reinterpret_cast<const uint2a *>(tid + (FWD ? (long long)q0 : (long long)m - 8 - q0));
          tb[FWD ? 0 : 1] = t8.x;
          tb[FWD ? 1 : 0] = t8.y;
        }
#pragma unroll
        for (int g = 0; g < 2; g++) {
          if (4 * g < nrow) {
            const long long qg = q0 + 4 * g;
            const hipx_int  rg = FWD ? (hipx_int)qg : (hipx_int)((long long)m - 4 - qg);  // lowest actual row of the group
            const dbl2     *pa = reinterpret_cast<const dbl2 *>(asrc + rg);
            const dbl2     *pb = reinterpret_cast<const dbl2 *>(xold + (NEEDOLD ? rg : 0));
            if (nrow <= 4) tb[g] = *reinterpret_cast<const unsigned *>(tid + rg);
#pragma unroll
            for (int h = 0; h < 2; h++) {
              const dbl2 a2 = pa[h];
              dbl2       b2 = {0.0, 0.0};
              if (NEEDOLD) b2 = pb[h];
              const int j0 = 4 * g + (FWD ? 2 * h : 3 - 2 * h), j1 = 4 * g + (FWD ? 2 * h + 1 : 2 - 2 * h);
              va[j0] = a2.x;
              va[j1] = a2.y;
              vb[j0] = b2.x;
              vb[j1] = b2.y;
            }
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < ST_SB; j++) {
          if (FINE && j >= nrow) continue;  // (row-granular staging: no padding loads)
          const hipx_int r = st_actual<FWD>(q0 + (j < nrow ? j : nrow - 1), m);
          va[j]            = asrc[r];
          vb[j]            = NEEDOLD ? xold[r] : 0.0;
          vt[j]            = tid[r];
        }
      }
    }
    st_dbl2 cv[CWT ? VSB : 1][CWT ? CWT / 2 : 1];
    if constexpr (CWT > 0) {
      if (nrow > 0) {
        const st_dbl2 *src = reinterpret_cast<const st_dbl2 *>(cs + (S * L + rqf) * CWT);  // (hipMalloc'ed, CWT even: 16-byte aligned)
#pragma unroll
        for (int j = 0; j < VSB; j++)
          if (j < nrow) {
#pragma unroll
            for (int h = 0; h < CWT / 2; h++) cv[j][h] = src[j * (CWT / 2) + h];
          }
      }
    }
    // (b) far strands: every duty = one window row of another panel's strand, staged ahead of its consumers.  Two rounds of
    // ST_NB duties (the second one is usually empty), so that only ST_NB * ST_SB values are in registers at a time.
    int fn[2 * ST_NB];
#pragma unroll
    for (int d = 0; d < 2 * ST_NB; d++) fn[d] = 0;
#pragma unroll
    for (int half = 0; half < 2; half++) {
      if (half == 1 && !wave_hi) break;
      unsigned long long fv[ST_NB][ST_SB];
      st_int4            fq[ST_NB][ST_SB / 2];  // poll16: the same rows as loaded, two per register quad
      int                frow[ST_NB], fodd[ST_NB];
#pragma unroll
      for (int i = 0; i < ST_NB; i++) {
        const int d = half * ST_NB + i;
        frow[i]     = 0;
        fodd[i]     = 0;
#pragma unroll
        for (int g = 0; g < ST_SB / 2; g++) fq[i][g] = st_int4{0, 0, 0, 0};
        if (dvalid[d]) {
          const int       b = d >> 1, which = d & 1;
          const int       w = P.band[b].width, u = lane + 64 * which;
          const long long strand = S0 + u + P.band[b].dsmin;
          int lead = -1000000, trail = 1000000;  // most / least advanced consumer of this row that is still running
          for (int c = u - (w - 1); c <= u; c++) {
            if (c >= 0 && c < cnt) {
              const int pl = SPLIT ? max((int)s_lead[c], (int)s_lead[64 + c]) : (int)s_lead[c], pt = s_trail[c];  // (SPLIT: the F waves lead, the C wave trails)
              if (pt < st_strand_len(S0 + c, P)) {
                if (pl > lead) lead = pl;
                if (pt < trail) trail = pt;
              }
            }
          }
          const int slen = st_strand_len(strand, P);
          int       tgt  = lead + ST_LA + 1;
          // ... but never over a slot the slowest consumer still needs: position q goes where q - 16 was, and a consumer at
          // `trail` reads positions >= trail - 1.  (Without this bound a panel whose producers are far ahead always stages to
          // the limit, the third consumer of a row -- four positions behind the first -- finds its slot gone, takes the
          // memory path, which drains the wave's stores (~2 us), falls further behind ...: measured, a whole plane at 3 us
          // per row instead of 1.3, and every later plane behind it.)
          if (tgt > trail + ST_WP - 2) tgt = trail + ST_WP - 2;
          if (tgt > slen) tgt = slen;
          int n = tgt - sf[d];
          if (n > want[d]) n = want[d];
          if (n > 0) {
            issued  = true;
            fn[d]   = n;
            frow[i] = P.band[b].rowbase + u;
            const long long q0 = strand * L + sf[d];
            if (ALIGNED) {
              // two rows per request: the pair (even position, next one) is one aligned 16-byte run (L and m are multiples of 8).  The
              // loaders' polls are most of what the CU's address pipeline sees -- every lane of every request is its own cache line --
              // and that pipeline, shared with the compute wave's stores, is what the sweep waits for.
              const int       odd = sf[d] & 1;  // an odd start: the pair's first half was accepted by an earlier pass, asked for again, skipped below
              const long long qa  = q0 - odd;
              fodd[i] = odd;
#pragma unroll
              for (int g = 0; g < ST_SB / 2; g++) {
                if (2 * g < n + odd) {
                  const double *src = FWD ? xnew + (qa + 2 * g) : xnew + ((long long)m - 2 - qa - 2 * g);
                  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(fq[i][g]) : "v"(src) : "memory");
                } else {
                  fq[i][g].x = fq[i][g].z = (int)(unsigned)SOR_SENTINEL;
                  fq[i][g].y = fq[i][g].w = (int)(unsigned)(SOR_SENTINEL >> 32);
                }
              }
            } else {
#pragma unroll
              for (int j = 0; j < ST_SB; j++) {
                if (j < n) {  // (wave-divergent count: lanes that ask for fewer rows skip the loads)
                  const unsigned long long *src = reinterpret_cast<const unsigned long long *>(xnew + st_actual<FWD>(q0 + j, m));
                  fv[i][j] = P.poll_sys ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else fv[i][j] = SOR_SENTINEL;
              }
            }
          }
        }
      }
      // everything of this round is in flight; now land it in LDS (first round: the operands too)
      if (half == 0 && nrow > 0) {
        if (ALIGNED && !FINE) {
#pragma unroll
          for (int g = 0; g < 2; g++)
#pragma unroll
            for (int k = 0; k < 4; k++) vt[4 * g + (FWD ? k : 3 - k)] = (unsigned char)(tb[g] >> (8 * k));
        }
#pragma unroll
        for (int j = 0; j < ST_SB; j++) {
          if (j < nrow) {
            const int       ro = P.off_rowq + 32 * (lane * RQ + ((rqf + j + ST_ROT * lane) & (RQ - 1)));
            if constexpr (CWT > 0) {
              if (j < VSB) {
                const int co = P.off_cring + 8 * CWT * (lane * RQ + ((rqf + j + ST_ROT * lane) & (RQ - 1)));
#pragma unroll
                for (int h = 0; h < CWT / 2; h++) {
                  const long long c0 = __double_as_longlong(cv[j < VSB ? j : 0][h].x), c1 = __double_as_longlong(cv[j < VSB ? j : 0][h].y);
                  st_st4v(lds, co + 16 * h, st_int4{(int)(unsigned)c0, (int)(unsigned)((unsigned long long)c0 >> 32), (int)(unsigned)c1, (int)(unsigned)((unsigned long long)c1 >> 32)});
                }
              }
            }
            const long long ba = __double_as_longlong(va[j]), bb = __double_as_longlong(vb[j]);
            st_int4         w0, w1;
            w0.x = (int)(unsigned)ba;
            w0.y = (int)(unsigned)((unsigned long long)ba >> 32);
            w0.z = (int)(unsigned)bb;
            w0.w = (int)(unsigned)((unsigned long long)bb >> 32);
            w1.x = (int)vt[j];
            w1.y = rqf + j;
            w1.z = 0;
            w1.w = 0;
            st_st4v(lds, ro, w0);  // operands first, tag last: the compute wave tests the tag (LDS executes a wave's accesses in order)
            st_st4v(lds, ro + 16, w1);
          }
        }
        rqf += nrow;
      }
      if (ALIGNED) {
        // the hand-issued loads are not counted by the compiler: one explicit wait, tied to every register they fill
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(fq[0][0]), "+v"(fq[0][1]), "+v"(fq[0][2]), "+v"(fq[0][3]), "+v"(fq[1][0]), "+v"(fq[1][1]), "+v"(fq[1][2]), "+v"(fq[1][3]), "+v"(fq[2][0]),
                       "+v"(fq[2][1]), "+v"(fq[2][2]), "+v"(fq[2][3])
                     :
                     : "memory");
        static_assert(ST_NB == 3 && ST_SB == 8, "the wait above names the registers");
      }
      auto fval = [&](int i, int j) __attribute__((always_inline)) -> unsigned long long {  // row j of duty i as loaded (static indices only)
        if (ALIGNED) {  // backward: the pair's halves are in descending logical order
          const st_int4 &q      = fq[i][j >> 1];
          const bool     second = FWD ? (j & 1) != 0 : (j & 1) == 0;
          return second ? (((unsigned long long)(unsigned)q.w << 32) | (unsigned)q.z) : (((unsigned long long)(unsigned)q.y << 32) | (unsigned)q.x);
        }
        return fv[i][j];
      };
#pragma unroll
      for (int i = 0; i < ST_NB; i++) {
        const int d = half * ST_NB + i;
        if (fn[d] > 0) {
          int       cnt = 0;
          bool      acc = true;
          const int a0  = sf[d] - fodd[i];  // position of fv[i][0] (poll16: the even position at or below sf[d])
#pragma unroll
          for (int j = 0; j < ST_SB; j++) {
            if (j >= fodd[i]) {
              const unsigned long long v = fval(i, j);
              if (j < fn[d] + fodd[i] && acc && v != SOR_SENTINEL) {
                st_st4v(lds, P.off_win + 16 * (frow[i] * ST_WP + ((a0 + j + ST_ROT * frow[i]) & (ST_WP - 1))), st_pack_slot(__longlong_as_double((long long)v), a0 + j + ST_ROT * frow[i]));
                cnt++;
              } else acc = false;
            }
          }
          sf[d] += cnt;
          if (P.poll_adapt) want[d] = cnt + 2 < ST_SB ? cnt + 2 : ST_SB;
        }
      }
    }
    if (stats && P.trace_panel == (int)panel && st_pass < 4096 && lane == P.trace_lane) {  // trace: the loader's view
      unsigned long long *tr = stats + 16 + 4 * (size_t)P.npanels + 64 * (size_t)L + (size_t)st_pass * 8;
      tr[0] = (unsigned long long)wall_clock64();
      tr[1] = (unsigned long long)myp;
      tr[2] = (unsigned long long)rqf;
#pragma unroll
      for (int d = 0; d < 3; d++) tr[3 + d] = (unsigned long long)sf[d];
      tr[6] = (unsigned long long)(fn[0] | (fn[1] << 8) | (fn[2] << 16));  // rows asked for in this pass
      tr[7] = (unsigned long long)(nrow | (s_trail[0] << 8) | ((unsigned long long)s_trail[1] << 24) | ((unsigned long long)s_trail[2] << 40));
    }
    st_pass++;
    if (!__any(issued)) {
      st_idle++;
      __builtin_amdgcn_s_sleep(4);
    }
  }
  if (stats && lane == 0) {
    atomicAdd(&stats[3], (unsigned long long)st_pass);
    atomicAdd(&stats[4], (unsigned long long)st_idle);
  }
}

// KIND as in sor_level_kernel: 0 fwd zero guess, 1 bwd with t, 2 bwd zero guess, 3 fwd general, 4 bwd whole row
// ME: dependency entries per row (the templates' lists are padded to ME with null entries: coefficient 0, pointing at a slot that
// always reads {0.0, ST_NULLTAG}); 4 for the 5-/7-point operators, 16 for the 27-point one.  SPLIT (ME 16 only): three waves
// per panel -- C, F, loader (see st_compute_role) -- instead of compute + loader.
// DBG: the HIPX_SOR_DEBUG instrumentation is compiled into its own instantiation (the production kernel carries none of the
// ~12 `if (stats)` tests per iteration)
// CWT > 0: variable coefficients (pattern templates + the coefficient stream cs; two-wave kernel, kinds 0-2).
template <int KIND, bool ALIGNED, int ME, bool SPLIT, bool DBG, int CWT = 0>
__global__ __launch_bounds__(SPLIT ? 256 : 128, SPLIT ? 2 : 1) void sor_strand_kernel(const StParams P, const unsigned char *__restrict__ tid, const StTinfo *__restrict__ g_tinfo,
                                                                       const StDiag *__restrict__ g_tdiag, const StEntry *__restrict__ g_dep, const StEntry *__restrict__ g_old,
                                                                       const StEntry *__restrict__ g_depF, const StEntry *__restrict__ g_depC, const int *__restrict__ pstart,
                                                                       const double *asrc, double *t,
                                                                       const double *xold, double *xnew, double omega, unsigned int *ctl, unsigned long long *stats_arg,
                                                                       const double *__restrict__ cs)
{
  static_assert(CWT == 0 || !SPLIT, "variable coefficients: two-wave kernel only");
  const int RQ = CWT ? P.rq : ST_RQ;
  constexpr int NT = SPLIT ? 256 : 128;
  unsigned long long *const stats = DBG ? stats_arg : nullptr;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  st_lds_char          *lds    = (st_lds_char *)smem;
  const unsigned        lds_base = (unsigned)(size_t)lds;  // byte address of the dynamic LDS region (for the hand-issued reads)
  volatile st_lds_int  *s_prog  = (volatile st_lds_int *)(lds + P.off_prog);   // progress of the compute (C) wave's lanes
  volatile st_lds_int  *s_progF = (volatile st_lds_int *)(lds + P.off_progF);  // SPLIT: progress of the two F waves' lanes (64 + 64)
  volatile st_lds_int  *s_ctl   = (volatile st_lds_int *)(lds + P.off_ctl);
  unsigned int *err  = ctl + 1;
  const int     lane = threadIdx.x & 63;
  // template tables -> LDS (16-byte records)
  for (int i = threadIdx.x; i < P.ntmpl; i += NT) {
    st_st4v(lds, P.off_tinfo + 16 * i, reinterpret_cast<const st_int4 *>(g_tinfo)[i]);
    st_st4v(lds, P.off_tdiag + 16 * i, reinterpret_cast<const st_int4 *>(g_tdiag)[i]);
  }
  if (SPLIT) {
    for (int i = threadIdx.x; i < P.ntmpl * (ME - ST_MC); i += NT) st_st4v(lds, P.off_depF + 16 * i, reinterpret_cast<const st_int4 *>(g_depF)[i]);
    for (int i = threadIdx.x; i < P.ntmpl * ST_MC; i += NT) st_st4v(lds, P.off_depC + 16 * i, reinterpret_cast<const st_int4 *>(g_depC)[i]);
    if (KIND == 0 && P.lock)  // the lockstep C wave's table: three records per template, stored behind the C entries
      for (int i = threadIdx.x; i < 3 * P.ntmpl; i += NT) st_st4v(lds, P.off_lock + 16 * i, reinterpret_cast<const st_int4 *>(g_depC)[P.ntmpl * ST_MC + i]);
  } else {
    for (int i = threadIdx.x; i < P.ndep; i += NT) st_st4v(lds, P.off_dep + 16 * i, reinterpret_cast<const st_int4 *>(g_dep)[i]);
  }
  for (int i = threadIdx.x; i < P.nold; i += NT) st_st4v(lds, P.off_old + 16 * i, reinterpret_cast<const st_int4 *>(g_old)[i]);
  // Roles by SIMD, not by wave index.  The four waves of a workgroup land on the CU's four SIMDs in the same order for both
  // resident workgroups, so with roles by wave index the two C waves -- the critical path of both panels -- share one SIMD's
  // issue slots while the SIMD with the two loaders idles.  The first workgroup to arrive on a CU (a counter per CU in ctl[4...])
  // keeps the natural order, the second one takes the rotated one.
  int wave = threadIdx.x >> 6;
  if (SPLIT && P.rolemap) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);  // HW_REG_HW_ID: SIMD_ID 5:4, CU_ID 11:8, SH_ID 12, SE_ID 15:13
    if (lane == 0) s_ctl[4 + wave] = (int)((hw >> 4) & 3);
    if (threadIdx.x == 0) {
      const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15;  // HW_REG_XCC_ID
      s_ctl[2] = (int)(atomicAdd(&ctl[4 + ((xcc << 8) | ((hw >> 8) & 0xff))], 1u) & 1u);
    }
    __syncthreads();
    const int sm = (1 << s_ctl[4]) | (1 << s_ctl[5]) | (1 << s_ctl[6]) | (1 << s_ctl[7]);
    if (sm == 15) wave = __builtin_amdgcn_readfirstlane((int)((P.rolemap >> (4 * (4 * s_ctl[2] + s_ctl[4 + wave]))) & 3u));  // (one wave per SIMD; otherwise keep the wave index)
  }
  for (;;) {
    __syncthreads();  // all waves are done with the previous panel (and the tables are in place)
    if (threadIdx.x == 0) s_ctl[0] = (int)atomicAdd(&ctl[0], 1u);
    {
      const st_int4 empty = {0, 0, -1, 0};
      for (int i = threadIdx.x; i < P.nrows * ST_WP; i += NT) st_st4v(lds, P.off_win + 16 * i, empty);
      const st_int4 empty_row = {0, -1, 0, 0};  // second half of a StRow: {tid, tag, -, -}
      for (int i = threadIdx.x; i < 64 * RQ; i += NT) st_st4v(lds, P.off_rowq + 32 * i + 16, empty_row);
      if (SPLIT)
        for (int i = threadIdx.x; i < 64 * ST_CQ; i += NT) st_st4v(lds, P.off_cq + 16 * i, st_int4{0, 0, 0, -1});
    }
    if (threadIdx.x < 64) {
      s_prog[threadIdx.x] = 0;
      if (SPLIT) {
        s_progF[threadIdx.x]      = 0;
        s_progF[64 + threadIdx.x] = 1;
      }
    }
    if (threadIdx.x == 0) {
      s_ctl[1] = 0;
      st_st4v(lds, P.off_null, st_int4{0, 0, ST_NULLTAG, 0});  // the null slot: value +0.0
    }
    __syncthreads();
    const unsigned panel = (unsigned)s_ctl[0];
    if (panel >= (unsigned)P.npanels) return;
    // a panel is a run of <= 64 consecutive strands (pstart: built by the host, see strand_build: runs are cut so that a
    // panel's far strands end where a producer panel ends)
    const long long S0  = pstart[panel];
    const int       cnt = pstart[panel + 1] - pstart[panel];
    const long long S   = S0 + lane;
    const int       len = lane < cnt ? st_strand_len(S, P) : 0;
    if (SPLIT) {
      if (wave == 0) {
        if (KIND == 0 && P.lock) st_lock_c<ALIGNED>(P, lds_base, s_prog, s_ctl, err, lane, S, len, t, xnew, stats, panel);
        else st_compute_role<KIND, ST_MC, 2, ALIGNED>(P, lds, lds_base, s_prog, s_prog, s_ctl, err, lane, panel, S, len, t, xold, xnew, omega, stats, 0);
      } else if (wave <= 2)
        st_compute_role<KIND, (SPLIT ? ME - ST_MC : ME), 1>(P, lds, lds_base, s_progF + 64 * (wave - 1), s_prog, s_ctl, err, lane, panel, S, len, t, xold, xnew, omega, nullptr, wave - 1,
                                        stats ? stats + 16 + 4 * (size_t)P.npanels + 64 * (size_t)P.L + 2 * 4096 * 8 + 48 + 8 * (wave - 1) : nullptr);
      else st_loader_role<KIND, ALIGNED, true>(P, lds, s_progF, s_prog, s_ctl, lane, panel, S0, cnt, S, len, tid, asrc, xold, xnew, stats);
    } else {
      if (wave == 0) st_compute_role<KIND, ME, 0, ALIGNED, CWT>(P, lds, lds_base, s_prog, s_prog, s_ctl, err, lane, panel, S, len, t, xold, xnew, omega, stats, 0);
      else st_loader_role<KIND, ALIGNED, false, CWT>(P, lds, s_prog, s_prog, s_ctl, lane, panel, S0, cnt, S, len, tid, asrc, xold, xnew, stats, cs);
    }
    // All roles share this function: without this, loads the LOADER branch may leave pending at the back edge of the panel loop
    // count as pending in the COMPUTE branches too (the compiler merges the paths), which plants vmcnt waits -- i.e. waits for
    // the compute wave's own stores -- in front of every register those loads use.  Drain everything here, explicitly.
    __builtin_amdgcn_s_waitcnt(0);
  }
}


__global__ void st_verify_kernel(const StParams P, int forward, const unsigned char *__restrict__ tid, const StTinfo *__restrict__ tinfo, const StEntry *__restrict__ dep, unsigned int *bad)
{
  for (hipx_int r = (hipx_int)blockIdx.x * blockDim.x + threadIdx.x; r < P.m; r += (hipx_int)gridDim.x * blockDim.x) {
    const long long q = forward ? (long long)r : (long long)P.m - 1 - r;
    const long long S = q / P.L, p = q - S * P.L;
    const StTinfo   ti = tinfo[tid[r]];
    bool            ok = true;
    for (int k = 0; k < ti.dcnt; k++) {
      const StEntry   e  = dep[ti.dstart + k];
      const int       dp = (int)(short)(e.pk & 0xffff);
      const long long qs = q + e.lo;  // logical source row: must sit in strand S + ds at position p + dp
      const long long ds = (e.lo - dp) / P.L;
      ok = ok && qs >= 0 && qs < P.m && qs / P.L == S + ds && qs - (S + ds) * P.L == p + dp && (e.lo - dp) % P.L == 0;
    }
    if (!ok) atomicAdd(bad, 1u);
  }
}

// Variable coefficients: one record of cw doubles per LOGICAL position q (row q forward, row m - 1 - q backward): the row's
// dependency-side coefficients in CSR order (= the order of the pattern template's entry list, padded with 0.0 to me), then the
// inverse diagonal with the same IEEE operations as invert_diag_kernel (aij.c:1807-1830), then 0.0 up to cw.
template <typename IT>
__global__ void st_cstream_kernel(hipx_int m, int forward, int me, int cw, const IT *__restrict__ ai, const double *__restrict__ aa, const int64_t *__restrict__ diagpos,
                                  const unsigned char *__restrict__ tid, const int *__restrict__ tstart, const int *__restrict__ toff, double omega, double shift, int plain,
                                  double *__restrict__ cs, unsigned int *zero_count)
{
  for (hipx_int q = (hipx_int)blockIdx.x * blockDim.x + threadIdx.x; q < m; q += (hipx_int)gridDim.x * blockDim.x) {
    const hipx_int r  = forward ? q : m - 1 - q;
    const int      t  = tid[r];
    const int64_t  k0 = (int64_t)ai[r];
    const int      e0 = tstart[t], e1 = tstart[t + 1];
    double        *out = cs + (size_t)q * (size_t)cw;
    int            j = 0;
    for (int e = e0; e < e1; e++) {
      const int off = toff[e];
      if ((forward ? off < 0 : off > 0) && j < me) out[j++] = aa[k0 + (e - e0)];
    }
    for (; j < me; j++) out[j] = 0.0;
    const double d = aa[diagpos[r]];
    if (d == 0.0 && plain && shift == 0.0) atomicAdd(zero_count, 1u);
    out[me] = plain ? 1.0 / d : omega / (shift + d);
    for (j = me + 1; j < cw; j++) out[j] = 0.0;
  }
}

struct StrandDir {
  bool      ok = false;
  double   *d_cs = nullptr;  // variable coefficients: the coefficient stream of this direction (m * P.cw doubles)
  StParams  P, Ps;  // Ps: the layout of the split kernel (P.split)
  StTinfo  *d_tinfo = nullptr;
  StDiag   *d_tdiag = nullptr;
  StEntry  *d_dep = nullptr, *d_old = nullptr, *d_depF = nullptr, *d_depC = nullptr;
  int      *d_pstart = nullptr;  // first strand of every panel (npanels + 1)
  std::vector<int>    h_pstart;
  std::vector<StDiag> h_tdiag;
};
struct StrandState {
  bool                 ok = false;
  const unsigned char *d_tid = nullptr;
  int                  ntmpl = 0;
  std::vector<double>  diagval;  // diagonal value of every template
  StrandDir            dir[2];   // [0] forward (dependencies = lower part), [1] backward
  unsigned int        *d_ctl = nullptr;
  unsigned long long  *d_stats = nullptr;
  long long            stats_panels = 0;
  unsigned long long   value_state = 0;
  double               omega = 0.0, shift = 0.0;
  bool                 diag_uploaded = false;
  bool                 var = false;       // built from PATTERN templates: coefficients per row from the streams (kinds 0-2 only)
  bool                 cs_valid = false;  // ... which hold the current values, omega and shift
  unsigned int         zero_pivots = 0;
};

void strand_free(StrandState *T)
{
  if (!T) return;
  for (auto &D : T->dir) {
    (void)hipFree(D.d_tinfo);
    (void)hipFree(D.d_tdiag);
    (void)hipFree(D.d_dep);
    (void)hipFree(D.d_depF);
    (void)hipFree(D.d_depC);
    (void)hipFree(D.d_pstart);
    (void)hipFree(D.d_old);
    (void)hipFree(D.d_cs);
  }
  (void)hipFree(T->d_ctl);
  (void)hipFree(T->d_stats);
  delete T;
}

// strand length: the centre of the cluster of offsets closest above {-1, 0, +1} in the most common template (n for an
// n x n x nz grid in natural ordering)
hipx_int strand_length(const int *toff, int len)
{
  std::vector<int> offs(toff, toff + len);
  std::sort(offs.begin(), offs.end());
  hipx_int best = 0;
  for (size_t i = 0; i < offs.size();) {
    size_t j = i;
    while (j + 1 < offs.size() && offs[j + 1] == offs[j] + 1) j++;
    const long long c = ((long long)offs[i] + offs[j]) / 2;
    if (c > 1 && (!best || c < best)) best = (hipx_int)c;
    i = j + 1;
  }
  return best;
}

// var: the tables come from PATTERN templates (tval == nullptr): structure only, no split / lockstep kernels, no old-value lists
int strand_build(StrandState *T, hipx_int m, int ntmpl, const int *tstart, const int *toff, const double *tval, const int *tdiag, const int64_t *tcount, const unsigned char *d_tid,
                 bool var = false)
{
  T->ok    = false;
  T->var   = var;
  T->d_tid = d_tid;
  T->ntmpl = ntmpl;
  int best = 0;
  for (int t = 1; t < ntmpl; t++)
    if (tcount[t] > tcount[best]) best = t;
  const hipx_int L = strand_length(toff + tstart[best], tstart[best + 1] - tstart[best]);
  if (L < 4 || L > m) return HIPX_SUCCESS;
  T->diagval.assign((size_t)ntmpl, 0.0);
  for (int t = 0; t < ntmpl; t++) {
    if (tdiag[t] < 0) return HIPX_SUCCESS;
    T->diagval[(size_t)t] = var ? 1.0 : tval[tstart[t] + tdiag[t]];
  }
  hipStream_t st = rt().compute;
  HIPX_HIP(hipMalloc((void **)&T->d_ctl, sizeof(unsigned int) * (4 + 4096)));  // [0] ticket, [1] error, [2] verify, [4...] arrivals per CU (role parity)
  HIPX_HIP(hipMemsetAsync(T->d_ctl, 0, sizeof(unsigned int) * (4 + 4096), st));
  for (int dirn = 0; dirn < 2; dirn++) {
    StrandDir &D   = T->dir[dirn];
    const bool fwd = dirn == 0;
    // logical strand deltas of the dependency side
    std::vector<int> dsall = {0};
    bool             fits  = true;
    auto decomp = [&](int lo, int &ds, int &dp) {
      ds = (int)std::floor((double)lo / (double)L + 0.5);
      dp = lo - ds * (int)L;
    };
    for (int t = 0; t < ntmpl && fits; t++)
      for (int k = tstart[t]; k < tstart[t + 1]; k++) {
        const int off = toff[k];
        if (fwd ? off < 0 : off > 0) {
          int ds, dp;
          decomp(fwd ? off : -off, ds, dp);
          if (dp < -3 || dp > 3) fits = false;
          dsall.push_back(ds);
        }
      }
    if (!fits) continue;
    std::sort(dsall.begin(), dsall.end());
    dsall.erase(std::unique(dsall.begin(), dsall.end()), dsall.end());
    StParams &P = D.P;
    memset(&P, 0, sizeof(P));
    P.m = m;
    P.L = L;
    P.nstr    = (hipx_int)(((long long)m + L - 1) / L);
    P.npanels = (P.nstr + 63) / 64;
    P.ntmpl   = ntmpl;
    int nb = 0;
    for (size_t i = 0; i < dsall.size() && fits;) {
      size_t j = i;
      while (j + 1 < dsall.size() && dsall[j + 1] == dsall[j] + 1) j++;
      if (nb >= ST_NB || (int)(j - i + 1) > ST_MAXW) fits = false;
      else {
        P.band[nb].dsmin   = dsall[i];
        P.band[nb].width   = (int)(j - i + 1);
        P.band[nb].rowbase = P.nrows;
        P.nrows += 64 + P.band[nb].width - 1;
        nb++;
      }
      i = j + 1;
    }
    if (!fits) continue;
    P.nbands = nb;
    // Panel boundaries.  With 64 strands per panel and ny a multiple of 64 the boundaries of every plane line up, and the last
    // lane of panel (z, b) needs line 64 (b + 1) of plane z - 1 = the FIRST lane of panel (z - 1, b + 1), which itself waits for
    // the last lane of (z - 1, b): two loader hand-offs per plane (measured 12.4 us per plane hop = 2 rows + 2 x 4.9 us).  So the
    // boundaries of plane z are shifted by z (mod 64) lines: its panels end where panels of plane z - 1 end, seen through the
    // largest delta of the farthest band, and a plane hop costs ONE hand-off.  Planes stay separate panels (the first line of a
    // plane does not depend on the last line of the plane below it; a panel straddling the two would chain them: tried, 25x
    // slower).  Any partition into runs of <= 64 consecutive strands is valid (dependencies only point to lower strands), so this
    // is a scheduling choice only (HIPX_SOR_STAGGER=0: off).
    {
      static const bool stagger_on = !(getenv("HIPX_SOR_STAGGER") && atoi(getenv("HIPX_SOR_STAGGER")) == 0);
      long long np = 0, sh = 0;  // strands per plane, boundary shift per plane
      if (nb >= 2 && P.band[0].width >= 2 && (P.band[0].width & 1)) {
        np = -((long long)P.band[0].dsmin + (P.band[0].width - 1) / 2);      // centre delta of the farthest band: one plane down
        sh = np + ((long long)P.band[0].dsmin + P.band[0].width - 1);         // np - |largest delta| (1 for the 27-point stencil)
        if (!stagger_on || np < 128 || sh <= 0 || sh >= 64) np = sh = 0;
      }
      D.h_pstart.clear();
      if (!np) {
        for (long long s0 = 0; s0 < P.nstr; s0 += 64) D.h_pstart.push_back((int)s0);
      } else {
        for (long long z = 0; z * np < P.nstr; z++) {
          const long long base = z * np, end = std::min<long long>(P.nstr, base + np), off = (z * sh) % 64;
          long long       s0 = base, c = off ? 64 - off : 64;
          while (s0 < end) {
            D.h_pstart.push_back((int)s0);
            s0 += std::min<long long>(c, end - s0);
            c = 64;
          }
        }
      }
      D.h_pstart.push_back((int)P.nstr);
      P.npanels = (hipx_int)D.h_pstart.size() - 1;
    }
    auto band_of = [&](int ds) {
      for (int b = 0; b < nb; b++)
        if (ds >= P.band[b].dsmin && ds < P.band[b].dsmin + P.band[b].width) return b;
      return -1;
    };
    std::vector<StTinfo> tinfo((size_t)ntmpl);
    std::vector<StEntry> dep, old;
    int                  maxdep = 0;
    for (int t = 0; t < ntmpl; t++) {  // first pass: the longest dependency list decides the kernel's entry count ME
      int c = 0;
      for (int k = tstart[t]; k < tstart[t + 1]; k++) c += (fwd ? toff[k] < 0 : toff[k] > 0) ? 1 : 0;
      maxdep = std::max(maxdep, c);
    }
    if (maxdep > (var ? 13 : ST_ME)) continue;  // rows with more dependency entries than the compute wave handles: level-ordered schedule
    const int ME = maxdep <= 4 ? 4 : (maxdep <= 13 ? 13 : ST_ME);  // 13: the 27-point class (3 fewer padding entries per row and iteration)
    for (int t = 0; t < ntmpl; t++) {
      StTinfo &ti = tinfo[(size_t)t];
      ti.dstart   = (int)dep.size();
      ti.ostart   = (int)old.size();
      int ndep_t  = 0;
      for (int k = tstart[t]; k < tstart[t + 1]; k++) {
        const int off = toff[k];
        if (fwd ? off < 0 : off > 0) {
          int ds, dp;
          decomp(fwd ? off : -off, ds, dp);
          const int b = band_of(ds);
          StEntry   e;
          e.pk  = ((P.band[b].rowbase + ds - P.band[b].dsmin) << 16) | (dp & 0xffff);
          e.lo  = fwd ? off : -off;
          e.val = var ? 0.0 : tval[k];
          dep.push_back(e);
          ndep_t++;
        } else if (!var && (fwd ? off > 0 : off <= 0)) {  // forward: upper part (KIND 3); backward: lower part then the diagonal (KIND 4)
          StEntry e;
          e.pk  = off;
          e.lo  = 0;
          e.val = tval[k];
          old.push_back(e);
        }
      }
      ti.dcnt = ndep_t;
      for (int k = ndep_t; k < ME; k++) dep.push_back(StEntry{ST_NULLPK, 0, 0.0});  // padding: coefficient 0, reads the null slot
      ti.ocnt = (int)old.size() - ti.ostart;
    }
    P.ndep      = (int)dep.size();
    P.nold      = (int)old.size();
    P.maxchunks = maxdep;
    P.me        = ME;
    // split kernel (ME 16): the list of a template cut into its first (up to) 12 and last (up to) 4 entries, fixed strides
    static const bool split_on = !(getenv("HIPX_SOR_SPLIT") && atoi(getenv("HIPX_SOR_SPLIT")) == 0);
    std::vector<StEntry> depF, depC;
    P.split = (ME >= 13 && split_on && !var) ? 1 : 0;
    // lockstep C wave (st_lock_c; forward sweep): every list must be far entries (strands of other panels: delta <= -64) followed
    // by near entries out of {(-1,-1), (-1,0), (-1,+1), (0,-1)} in this order.  HIPX_SOR_LOCKSTEP=0|1.
    static const bool lock_on = !(getenv("HIPX_SOR_LOCKSTEP") && atoi(getenv("HIPX_SOR_LOCKSTEP")) == 0);
    bool              lock = P.split && fwd && lock_on && L >= 4;
    std::vector<int>  nfar((size_t)ntmpl, 0);
    std::vector<StEntry> lockrec;  // per template: {c0, c1} {c2, c3} {mask}
    for (int t = 0; t < ntmpl && lock; t++) {
      const StTinfo &ti = tinfo[(size_t)t];
      double         c[4] = {0.0, 0.0, 0.0, 0.0};
      int            msk = 0, last = -1;
      for (int k = 0; k < ti.dcnt; k++) {
        const StEntry &e = dep[(size_t)ti.dstart + k];
        int            ds, dp;
        decomp(e.lo, ds, dp);
        if (ds <= -64) {
          if (last >= 0) lock = false;  // a far entry behind a near one: the F waves' part is not a prefix
          nfar[(size_t)t]++;
        } else {
          const int ci = (ds == -1 && dp >= -1 && dp <= 1) ? dp + 1 : ((ds == 0 && dp == -1) ? 3 : -1);
          if (ci < 0 || ci <= last) lock = false;
          else {
            last  = ci;
            c[ci] = e.val;
            msk |= 1 << ci;
          }
        }
      }
      if (nfar[(size_t)t] > ME - ST_MC) lock = false;
      StEntry r0, r1, r2;
      memcpy(&r0, &c[0], 16);
      memcpy(&r1, &c[2], 16);
      r2 = StEntry{msk, 0, 0.0};
      lockrec.push_back(r0);
      lockrec.push_back(r1);
      lockrec.push_back(r2);
    }
    if (getenv("HIPX_SOR_TRACE")) fprintf(stderr, "[hipx sor] strand %s: ME %d split %d lockstep %d\n", fwd ? "forward" : "backward", ME, P.split, lock ? 1 : 0);
    if (P.split) {
      for (int t = 0; t < ntmpl; t++) {
        const StTinfo &ti = tinfo[(size_t)t];
        const int      nF = lock ? nfar[(size_t)t] : ti.dcnt - std::min(ti.dcnt, ST_MC), nC = ti.dcnt - nF;
        for (int k = 0; k < ME - ST_MC; k++) depF.push_back(k < nF ? dep[(size_t)ti.dstart + k] : StEntry{ST_NULLPK, 0, 0.0});
        for (int k = 0; k < ST_MC; k++) depC.push_back(k < nC ? dep[(size_t)ti.dstart + nF + k] : StEntry{ST_NULLPK, 0, 0.0});
      }
    }
    if (var) {
      // ring depth: 8 rows ahead fit two workgroups per CU for the 5-/7-point class (48-byte records); the 27-point class (112-byte
      // records) needs 57 KiB for that depth -> one workgroup per CU, or depth 4 with two (HIPX_SOR_VAR_RING=4|8|16)
      static const int ring_env = getenv("HIPX_SOR_VAR_RING") ? atoi(getenv("HIPX_SOR_VAR_RING")) : 0;
      P.var = 1;
      P.cw  = (ME + 2) & ~1;  // ME coefficients + 1 / d, padded to whole 16-byte words: 6 (ME 4), 14 (ME 13)
      P.rq  = (ring_env == 4 || ring_env == 8 || ring_env == 16) ? ring_env : 8;
    }
    const int RQ = var ? P.rq : ST_RQ;
    int o = 0;
    P.off_win   = o; o += P.nrows * ST_WP * (int)sizeof(StSlot);
    P.off_rowq  = o; o += 64 * RQ * (int)sizeof(StRow);
    if (var) { P.off_cring = o; o += 64 * RQ * P.cw * 8; }
    P.off_tinfo = o; o += ntmpl * (int)sizeof(StTinfo);
    P.off_tdiag = o; o += ntmpl * (int)sizeof(StDiag);
    P.off_dep   = o; o += std::max(P.ndep, 1) * (int)sizeof(StEntry);
    P.off_old   = o; o += std::max(P.nold, 1) * (int)sizeof(StEntry);
    P.off_prog  = o; o += 64 * 4;
    P.off_ctl   = o; o += 32;
    P.off_null  = o; o += 16;
    P.off_cq = P.off_progF = P.off_depF = P.off_depC = 0;
    P.lds_bytes = o;
    P.wgcu      = P.lds_bytes > 78 * 1024 ? 1 : 2;
    if (P.lds_bytes > (var ? 156 : 78) * 1024) continue;  // two workgroups per CU must fit the 160 KiB (variable coefficients: one may do)
    D.Ps = P;
    if (P.split) {  // the split kernel's own layout: its two tables instead of the whole-row one, no old-value lists (kinds 0-2 only)
      StParams &Q = D.Ps;
      o           = P.off_dep;
      Q.off_depF  = o; o += ntmpl * (ME - ST_MC) * (int)sizeof(StEntry);
      Q.off_depC  = o; o += ntmpl * ST_MC * (int)sizeof(StEntry);
      Q.off_old   = o;
      Q.nold      = 0;
      Q.off_prog  = o; o += 64 * 4;
      Q.off_progF = o; o += 2 * 64 * 4;
      Q.off_ctl   = o; o += 32;
      Q.off_null  = o; o += 16;
      Q.off_cq    = o; o += 64 * ST_CQ * 16;
      if (lock) {
        Q.lock     = 1;
        Q.off_lock = o; o += 48 * ntmpl;
        for (int b = 0; b < nb; b++)
          if (P.band[b].dsmin <= -1 && -1 < P.band[b].dsmin + P.band[b].width) Q.lock_wrow = P.band[b].rowbase + (-1 - P.band[b].dsmin);  // window row of (lane 0, strand delta -1)
        depC.insert(depC.end(), lockrec.begin(), lockrec.end());
      }
      Q.lds_bytes = o;
      if (Q.lds_bytes > 78 * 1024) P.split = Q.split = 0;
    }
    if (dep.empty()) dep.push_back(StEntry{0, 0, 0.0});
    if (old.empty()) old.push_back(StEntry{0, 0, 0.0});
    HIPX_HIP(hipMalloc((void **)&D.d_tinfo, sizeof(StTinfo) * (size_t)ntmpl));
    HIPX_HIP(hipMalloc((void **)&D.d_tdiag, sizeof(StDiag) * (size_t)ntmpl));
    HIPX_HIP(hipMalloc((void **)&D.d_dep, sizeof(StEntry) * dep.size()));
    HIPX_HIP(hipMalloc((void **)&D.d_old, sizeof(StEntry) * old.size()));
    HIPX_HIP(hipMemcpyAsync(D.d_tinfo, tinfo.data(), sizeof(StTinfo) * (size_t)ntmpl, hipMemcpyHostToDevice, st));
    HIPX_HIP(hipMemcpyAsync(D.d_dep, dep.data(), sizeof(StEntry) * dep.size(), hipMemcpyHostToDevice, st));
    HIPX_HIP(hipMemcpyAsync(D.d_old, old.data(), sizeof(StEntry) * old.size(), hipMemcpyHostToDevice, st));
    HIPX_HIP(hipMalloc((void **)&D.d_pstart, sizeof(int) * D.h_pstart.size()));
    HIPX_HIP(hipMemcpyAsync(D.d_pstart, D.h_pstart.data(), sizeof(int) * D.h_pstart.size(), hipMemcpyHostToDevice, st));
    if (P.split) {
      HIPX_HIP(hipMalloc((void **)&D.d_depF, sizeof(StEntry) * depF.size()));
      HIPX_HIP(hipMalloc((void **)&D.d_depC, sizeof(StEntry) * depC.size()));
      HIPX_HIP(hipMemcpyAsync(D.d_depF, depF.data(), sizeof(StEntry) * depF.size(), hipMemcpyHostToDevice, st));
      HIPX_HIP(hipMemcpyAsync(D.d_depC, depC.data(), sizeof(StEntry) * depC.size(), hipMemcpyHostToDevice, st));
    }
    // every row's dependency entries must decompose into (strand delta, position delta) as the tables say
    HIPX_HIP(hipMemsetAsync(T->d_ctl + 2, 0, sizeof(unsigned int), st));
    st_verify_kernel<<<(unsigned)std::min<hipx_int>((m + 255) / 256, 4096), 256, 0, st>>>(P, fwd ? 1 : 0, d_tid, D.d_tinfo, D.d_dep, T->d_ctl + 2);
    unsigned int bad = 0;
    HIPX_HIP(hipMemcpyAsync(&bad, T->d_ctl + 2, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
    HIPX_HIP(hipStreamSynchronize(st));  // also: tinfo / dep / old (host vectors) are consumed
    HIPX_LAUNCH_CHECK();
    D.ok = bad == 0;
  }
  T->ok = T->dir[0].ok && T->dir[1].ok;
  return HIPX_SUCCESS;
}

// inverse diagonals per template: the same IEEE operations as invert_diag_kernel on the same doubles
int strand_set_diag(StrandState *T, double omega, double shift)
{
  if (T->diag_uploaded && T->omega == omega && T->shift == shift) return HIPX_SUCCESS;
  const bool plain = (omega == 1.0 && shift <= 0.0);
  for (auto &D : T->dir) {
    D.h_tdiag.resize((size_t)T->ntmpl);
    for (int t = 0; t < T->ntmpl; t++) {
      const double d = T->diagval[(size_t)t];
      D.h_tdiag[(size_t)t].mdiag = d;
      D.h_tdiag[(size_t)t].idiag = plain ? 1.0 / d : omega / (shift + d);
    }
    HIPX_HIP(hipMemcpy(D.d_tdiag, D.h_tdiag.data(), sizeof(StDiag) * (size_t)T->ntmpl, hipMemcpyHostToDevice));
  }
  T->omega = omega;
  T->shift = shift;
  T->diag_uploaded = true;
  return HIPX_SUCCESS;
}

// variable coefficients: (re)build the two coefficient streams from the matrix's current values, omega and shift
int strand_set_cs(StrandState *T, hipx_int m, int is64, const void *d_i, const double *d_a, const int64_t *d_diagpos, const int *d_tstart, const int *d_toff, double omega, double shift)
{
  if (T->cs_valid && T->omega == omega && T->shift == shift) return HIPX_SUCCESS;
  hipStream_t   st    = rt().compute;
  const int     plain = (omega == 1.0 && shift <= 0.0) ? 1 : 0;
  unsigned int *cnt   = T->d_ctl + 3;
  HIPX_HIP(hipMemsetAsync(cnt, 0, sizeof(unsigned int), st));
  const unsigned g = (unsigned)std::min<hipx_int>((m + 255) / 256, 8192);
  for (int dirn = 0; dirn < 2; dirn++) {
    StrandDir &D = T->dir[dirn];
    if (!D.d_cs) HIPX_HIP(hipMalloc((void **)&D.d_cs, sizeof(double) * (size_t)m * (size_t)D.P.cw));
    if (is64)
      st_cstream_kernel<int64_t><<<g, 256, 0, st>>>(m, dirn == 0 ? 1 : 0, D.P.me, D.P.cw, (const int64_t *)d_i, d_a, d_diagpos, T->d_tid, d_tstart, d_toff, omega, shift, plain, D.d_cs, cnt);
    else
      st_cstream_kernel<hipx_int><<<g, 256, 0, st>>>(m, dirn == 0 ? 1 : 0, D.P.me, D.P.cw, (const hipx_int *)d_i, d_a, d_diagpos, T->d_tid, d_tstart, d_toff, omega, shift, plain, D.d_cs, cnt);
    HIPX_LAUNCH_CHECK();
  }
  HIPX_HIP(hipMemcpyAsync(&T->zero_pivots, cnt, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
  HIPX_HIP(hipStreamSynchronize(st));
  T->zero_pivots /= 2;  // (both directions count the same rows)
  T->omega    = omega;
  T->shift    = shift;
  T->cs_valid = true;
  return HIPX_SUCCESS;
}

template <int KIND>
int run_strand(StrandState *T, const double *asrc, double *t, const double *xold, double *xnew, double omega)
{
  constexpr bool FWD = (KIND == 0 || KIND == 3);
  StrandDir     &D   = T->dir[FWD ? 0 : 1];
  StParams       P   = D.P;
  P.trace_panel      = -1;
  P.trace_lane       = 32;
  P.trace_it0        = getenv("HIPX_SOR_TRACE_IT0") ? atoi(getenv("HIPX_SOR_TRACE_IT0")) : 600;
  P.trace_rows       = getenv("HIPX_SOR_TRACE_ROWS") ? atoi(getenv("HIPX_SOR_TRACE_ROWS")) : 1;
  {
    static const bool ps = getenv("HIPX_SOR_POLL") && !strcmp(getenv("HIPX_SOR_POLL"), "sys");
    P.poll_sys = ps ? 1 : 0;
    // measured on one box: 7-pt 256^3 GMRES+SOR 153.7 -> 167.7 it/s with the adaptive poll width, 27-pt 256^3 96.0 -> 92.5
    static const int pa = getenv("HIPX_SOR_ADAPT") ? atoi(getenv("HIPX_SOR_ADAPT")) : -1;
    P.poll_adapt = pa >= 0 ? (pa != 0) : (P.me == 4);
  }
  hipStream_t    st  = rt().compute;
  const hipx_int g   = std::min<hipx_int>((P.m + 255) / 256, 4096);
  sor_fill_kernel<<<(unsigned)g, 256, 0, st>>>(xnew, P.m);
  HIPX_HIP(hipMemsetAsync(T->d_ctl, 0, sizeof(unsigned int), st));  // ticket; the error word is sticky until read
  static int per_cu = 0;
  if (!per_cu) {
    const char *e = getenv("HIPX_SOR_WG_PER_CU");
    per_cu        = e ? atoi(e) : 2;
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 2) per_cu = 2;
  }
  unsigned grid = (unsigned)std::min<long long>((long long)P.npanels, 256LL * std::min(per_cu, P.wgcu ? P.wgcu : 2));
  static const bool want_aligned = !(getenv("HIPX_SOR_ALIGNED") && atoi(getenv("HIPX_SOR_ALIGNED")) == 0);  // 0: the kernels for any L / alignment (8-byte polls, scalar operand loads)
  const bool aligned = want_aligned && (P.L % 8 == 0) && (P.m % 8 == 0) && ((reinterpret_cast<uintptr_t>(asrc) | reinterpret_cast<uintptr_t>(xold) | reinterpret_cast<uintptr_t>(xnew) | reinterpret_cast<uintptr_t>(t)) % 16 == 0);  // (a null t / xold counts as aligned)
  static const bool dbg = getenv("HIPX_SOR_DEBUG") != nullptr;
  static bool attr_set[5][24] = {{false}};
  // The split kernel pays when the far entries come FIRST in the row's list (forward sweeps: lower planes, then the previous line,
  // then the row's predecessor): the F waves run ahead with them.  In a backward sweep the list starts with the near entries, the
  // far subtractions depend on them, and nothing can run ahead (measured on the config-3 slab: forward 2.1 us per line and 0.94 us
  // per row against 3.25 / 1.4 with two waves; backward 2x SLOWER than with two waves).  HIPX_SOR_SPLIT = 0 off | 1 forward
  // zero-guess sweep only (default) | 2 every sweep of kinds 0-2.
  static const int split_mode = getenv("HIPX_SOR_SPLIT") ? atoi(getenv("HIPX_SOR_SPLIT")) : 1;
  const bool     split    = P.split && (split_mode >= 2 ? KIND <= 2 : (split_mode == 1 && KIND == 0));
  const unsigned nthreads = (P.me != 4 && split) ? 256 : 128;
  if (split) {
    const int tp = P.trace_panel, tl = P.trace_lane, ti = P.trace_it0, tr = P.trace_rows, ps = P.poll_sys, pad = P.poll_adapt;
    P = D.Ps;
    P.trace_panel = tp; P.trace_lane = tl; P.trace_it0 = ti; P.trace_rows = tr; P.poll_sys = ps; P.poll_adapt = pad;
    // roles by SIMD: C on SIMD 0, the F waves on 1 and 2, the loader on 3, in BOTH resident workgroups (measured on the slab: 5.33 ms
    // against 5.49 by wave index -- which the hardware already rotates between the two workgroups -- and 5.41 for the two mappings
    // that keep the C waves on different SIMDs).  HIPX_SOR_ROLEMAP=<hex nibbles, SIMD 0 first / second workgroup in the high half> | 0
    static const unsigned rolemap = getenv("HIPX_SOR_ROLEMAP") ? (unsigned)strtoul(getenv("HIPX_SOR_ROLEMAP"), nullptr, 16) : 0x32103210u;
    P.rolemap = rolemap;
  }
  auto launch = [&](auto kern, int ai) -> int {
    if (!attr_set[KIND][ai]) {
      HIPX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (P.var ? 158 : 80) * 1024));
      attr_set[KIND][ai] = true;
    }
    kern<<<grid, nthreads, (size_t)P.lds_bytes, st>>>(P, T->d_tid, D.d_tinfo, D.d_tdiag, D.d_dep, D.d_old, D.d_depF, D.d_depC, D.d_pstart, asrc, t, xold, xnew, omega, T->d_ctl,
                                                       (dbg && !P.var) ? T->d_stats : nullptr, D.d_cs);
    return HIPX_SUCCESS;
  };
  if (dbg) {
    const long long stats_words = 16 + 4 * (long long)P.npanels + 64 * (long long)P.L + 2 * 4096 * 8 + 64;
    if (T->d_stats && T->stats_panels < stats_words) {
      (void)hipFree(T->d_stats);
      T->d_stats = nullptr;
    }
    if (!T->d_stats) {
      HIPX_HIP(hipMalloc((void **)&T->d_stats, sizeof(unsigned long long) * (size_t)stats_words));
      T->stats_panels = stats_words;
    }
    HIPX_HIP(hipMemsetAsync(T->d_stats, 0, sizeof(unsigned long long) * (size_t)stats_words, st));
    if (const char *tp = getenv("HIPX_SOR_TRACE_PANEL")) P.trace_panel = atoi(tp);
    P.trace_lane = getenv("HIPX_SOR_TRACE_LANE") ? atoi(getenv("HIPX_SOR_TRACE_LANE")) : 32;
    HIPX_HIP(hipStreamSynchronize(st));
    fprintf(stderr, "[hipx sor] strand KIND %d aligned %d m %d L %d nstr %d npanels %d nbands %d nrows %d ntmpl %d ndep %d nold %d maxchunks %d lds %d grid %u  tid %p asrc %p t %p xold %p xnew %p ctl %p\n", KIND,
            (int)aligned, P.m, P.L, P.nstr, P.npanels, P.nbands, P.nrows, P.ntmpl, P.ndep, P.nold, P.maxchunks, P.lds_bytes, grid, (const void *)T->d_tid, (const void *)asrc, (void *)t, (const void *)xold,
            (void *)xnew, (void *)T->d_ctl);
  }
  int ierr;
#define HIPX_ST_LAUNCH(AL, MEV, SP, IDX) (dbg ? launch(sor_strand_kernel<KIND, AL, MEV, SP, true>, 2 * (IDX) + 1) : launch(sor_strand_kernel<KIND, AL, MEV, SP, false>, 2 * (IDX)))
  if (P.var) {  // variable coefficients: two-wave kernels of kinds 0-2, entry counts 4 and 13 (the production instantiation only)
    if constexpr (KIND <= 2) {
      if (P.me == 4) ierr = aligned ? launch(sor_strand_kernel<KIND, true, 4, false, false, 6>, 20) : launch(sor_strand_kernel<KIND, false, 4, false, false, 6>, 21);
      else if (P.me == 13) ierr = aligned ? launch(sor_strand_kernel<KIND, true, 13, false, false, 14>, 22) : launch(sor_strand_kernel<KIND, false, 13, false, false, 14>, 23);
      else ierr = HIPX_ERR_SUP;
    } else ierr = HIPX_ERR_SUP;
  } else if (P.me == 4) ierr = aligned ? HIPX_ST_LAUNCH(true, 4, false, 1) : HIPX_ST_LAUNCH(false, 4, false, 0);
  else if (split) {
    if constexpr (KIND <= 2) {
      if (P.me == 13) ierr = aligned ? HIPX_ST_LAUNCH(true, 13, true, 9) : HIPX_ST_LAUNCH(false, 13, true, 8);
      else ierr = aligned ? HIPX_ST_LAUNCH(true, ST_ME, true, 5) : HIPX_ST_LAUNCH(false, ST_ME, true, 4);
    } else ierr = HIPX_ERR_SUP;
  } else if (P.me == 13) ierr = aligned ? HIPX_ST_LAUNCH(true, 13, false, 7) : HIPX_ST_LAUNCH(false, 13, false, 6);
  else ierr = aligned ? HIPX_ST_LAUNCH(true, ST_ME, false, 3) : HIPX_ST_LAUNCH(false, ST_ME, false, 2);
#undef HIPX_ST_LAUNCH
  if (ierr) return ierr;
  HIPX_LAUNCH_CHECK();
  if (dbg) {
    HIPX_HIP(hipStreamSynchronize(st));
    unsigned long long hs[16];
    HIPX_HIP(hipMemcpy(hs, T->d_stats, sizeof(hs), hipMemcpyDeviceToHost));
    const double np = (double)std::max<unsigned long long>(hs[8], 1);
    fprintf(stderr, "[hipx sor] strand KIND %d done: panels %llu, compute iterations/panel %.0f, wall/panel %.1f us (%.3f us/iteration), rows %llu, lane-iterations waiting: operands %llu deps %llu, "
                    "fallback loads %llu, loader passes/panel %.0f (idle %.0f)\n",
            KIND, hs[8], hs[0] / np, hs[6] / np / 100.0, hs[0] ? hs[6] / 100.0 / (double)hs[0] : 0.0, hs[7], hs[1], hs[2], hs[5], hs[3] / np, hs[4] / np);
    fprintf(stderr, "[hipx sor]   per panel (lane 0's view): iterations with a finishing lane %.0f, with a fallback lane %.0f, with a lane setting up %.0f; shader clocks: total %.0f (%.0f per "
                    "iteration), in the burst %.0f (%.0f per iteration), in the finish path %.0f (%.0f per finishing iteration)\n",
            hs[9] / np, hs[10] / np, hs[11] / np, hs[14] / np, hs[0] ? (double)hs[14] / hs[0] : 0.0, hs[12] / np, hs[0] ? (double)hs[12] / hs[0] : 0.0, hs[13] / np, hs[9] ? (double)hs[13] / hs[9] : 0.0);
    if (split) {
      unsigned long long fs[16];
      HIPX_HIP(hipMemcpy(fs, T->d_stats + 16 + 4 * (size_t)P.npanels + 64 * (size_t)P.L + 2 * 4096 * 8 + 48, sizeof(fs), hipMemcpyDeviceToHost));
      for (int w = 0; w < 2; w++)
        fprintf(stderr, "[hipx sor]   F wave %d, lane 32, per panel: iterations %.0f, waiting for operands %.0f, for far values %.0f, for room in the hand-over ring %.0f, rows done %.0f\n", w,
                fs[8 * w] / np, fs[8 * w + 1] / np, fs[8 * w + 2] / np, fs[8 * w + 3] / np, fs[8 * w + 4] / np);
    }
    if (const char *dump = getenv("HIPX_SOR_DEBUG_DUMP")) {  // per panel: start, first row of lane 0, first row of lane 63, end (wall-clock ticks, 10 ns)
      std::vector<unsigned long long> pt(4 * (size_t)P.npanels);
      HIPX_HIP(hipMemcpy(pt.data(), T->d_stats + 16, sizeof(unsigned long long) * pt.size(), hipMemcpyDeviceToHost));
      char name[512];
      snprintf(name, sizeof(name), "%s_kind%d.txt", dump, KIND);
      if (FILE *f = fopen(name, "w")) {
        unsigned long long base = ~0ull;
        for (size_t i = 0; i < (size_t)P.npanels; i++) base = std::min(base, pt[4 * i]);
        for (size_t i = 0; i < (size_t)P.npanels; i++)
          fprintf(f, "%zu %.2f %.2f %.2f %.2f\n", i, (pt[4 * i] - base) / 100.0, (pt[4 * i + 1] - base) / 100.0, (pt[4 * i + 2] - base) / 100.0, (pt[4 * i + 3] - base) / 100.0);
        fclose(f);
      }
      if (getenv("HIPX_SOR_TRACE_PANEL")) {
        std::vector<unsigned long long> tr(64 * (size_t)P.L + 2 * 4096 * 8 + 64);
        HIPX_HIP(hipMemcpy(tr.data(), T->d_stats + 16 + 4 * (size_t)P.npanels, sizeof(unsigned long long) * tr.size(), hipMemcpyDeviceToHost));
        snprintf(name, sizeof(name), "%s_trace%d.bin", dump, KIND);
        if (FILE *f = fopen(name, "wb")) {
          const long long hdr[4] = {(long long)P.L, 64, 4096, 8};
          fwrite(hdr, sizeof(hdr), 1, f);
          fwrite(tr.data(), sizeof(unsigned long long), tr.size(), f);
          fclose(f);
        }
      }
    }
  }
  return HIPX_SUCCESS;
}

// ---------------------------------------------------------------------------------------------------------------
// Matrices with INODES: MatSOR_SeqAIJ_Inode (inode.c:2494-3810), which MatSOR_SeqAIJ runs instead of its own loops when the matrix
// has inodes and omega == 1, fshift == 0 (aij.c:1852).  A node = up to 5 consecutive rows with ONE column list (the unknowns of a
// mesh node); the sweep is block Gauss-Seidel over the nodes:
//     s_r = rhs_r - sum over the node's off-block entries, taken in PAIRS: s_r -= (a_r[k] x[j_k] + a_r[k+1] x[j_k+1])   (inode.c:2589-2608)
//     x_r = sum_c s_c * D^-1[r, c]   (c ascending; D = the node's dense diagonal block, inverted once per set of values by LINPACK's
//           dgefa + dgedi: dgefa3.c:14 and its siblings for 2, 4, 5 rows)                                                (inode.c:2612-2614)
// -- not the arithmetic of the point sweep, so these matrices get their own schedule: the level-ordered dependency-driven sweep of
// above on the NODE graph (one lane per node, a wave = 64 nodes of one level, values polled out of the sentinel-filled new vector).
// The node-level copy of the matrix: the shared column list once per node, the values entry-major ([entry][row of the node], padded to
// the largest node of the matrix): a lane streams its node's entries with unit stride.
struct InodeState {
  bool      ready = false, values_valid = false;
  hipx_int  nnodes = 0, nslots = 0, nlevels = 0;
  int       nsm = 2;              // rows of the largest node
  int4     *d_nmeta = nullptr;    // per node (level order): {first row, rows, entries before the diagonal block, entries per row}
  int64_t  *d_nks = nullptr;      // per node: start of its entries in d_nj / d_nv
  int4     *d_smeta = nullptr;    // per slot: the node's meta (first row = -1: padding)
  int64_t  *d_sks = nullptr;
  hipx_int *d_sp = nullptr;       // per slot: node index (level order)
  hipx_int *d_nj = nullptr;
  double   *d_nv = nullptr;
  double   *d_ibd = nullptr, *d_bd = nullptr;  // per node nsm * nsm: inverse / copy of the diagonal block, column-major with the NODE's size as stride
  // the same slot tables with every level padded to 4 nodes: the cooperative kernel (16 lanes per node, 4 nodes per wave)
  hipx_int  nslots4 = 0;
  int4     *d_smeta4 = nullptr;
  int64_t  *d_sks4 = nullptr;
  hipx_int *d_sp4 = nullptr;
  unsigned int zero_pivots = 0;
  int64_t   nentries = 0;
};

void inode_free(InodeState *T)
{
  if (!T) return;
  (void)hipFree(T->d_nmeta);
  (void)hipFree(T->d_nks);
  (void)hipFree(T->d_smeta);
  (void)hipFree(T->d_sks);
  (void)hipFree(T->d_sp);
  (void)hipFree(T->d_nj);
  (void)hipFree(T->d_nv);
  (void)hipFree(T->d_ibd);
  (void)hipFree(T->d_bd);
  (void)hipFree(T->d_smeta4);
  (void)hipFree(T->d_sks4);
  (void)hipFree(T->d_sp4);
  delete T;
}

// MatSeqAIJCheckInode's comparison (inode.c:3948-3953), row against the row before it: same[i] = 1 when row i + 1 has the column list of row i
__global__ void inode_same_kernel(hipx_int m, const void *ai_, int is64, const hipx_int *__restrict__ aj, unsigned char *__restrict__ same, unsigned int *nsame)
{
  unsigned int mine = 0;
  for (hipx_int i = (hipx_int)blockIdx.x * blockDim.x + threadIdx.x; i + 1 < m; i += (hipx_int)gridDim.x * blockDim.x) {
    const int64_t s0 = is64 ? ((const int64_t *)ai_)[i] : (int64_t)((const hipx_int *)ai_)[i];
    const int64_t s1 = is64 ? ((const int64_t *)ai_)[i + 1] : (int64_t)((const hipx_int *)ai_)[i + 1];
    const int64_t s2 = is64 ? ((const int64_t *)ai_)[i + 2] : (int64_t)((const hipx_int *)ai_)[i + 2];
    unsigned char eq = (s1 - s0) == (s2 - s1);
    for (int64_t k = 0; eq && k < s1 - s0; k++) eq = aj[s0 + k] == aj[s1 + k];
    same[i] = eq;
    mine += eq;
  }
  // how many rows repeat their predecessor's column list, for the host's early way out (inode_find)
  for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(nsame, mine);
}

// the nodes of the matrix as the reference finds them at assembly (inode.c:3940-3965; limit 5 = the default of -mat_inode_limit):
// sizes <- node_count + 1 row offsets; returns node_count, 0 when the reference would not use the inode routines
int inode_find(hipx_int m, int is64, const void *d_i, const hipx_int *d_j, std::vector<hipx_int> &sizes, hipx_int *node_count)
{
  *node_count = 0;
  sizes.clear();
  if (m < 2) return HIPX_SUCCESS;
  hipStream_t    st = rt().compute;
  unsigned char *d_same;
  HIPX_HIP(hipMalloc((void **)&d_same, (size_t)m));
  unsigned int *d_cnt = rt().d_tickets + (HIPX_MAX_RED_SLOTS - 1), nsame = 0;
  HIPX_HIP(hipMemsetAsync(d_cnt, 0, sizeof(unsigned int), st));
  inode_same_kernel<<<(unsigned)std::min<hipx_int>((m + 255) / 256, 8192), 256, 0, st>>>(m, d_i, is64, d_j, d_same, d_cnt);
  HIPX_LAUNCH_CHECK();
  HIPX_HIP(hipMemcpyAsync(&nsame, d_cnt, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
  HIPX_HIP(hipStreamSynchronize(st));
  HIPX_HIP(hipMemsetAsync(d_cnt, 0, sizeof(unsigned int), st));
  // Every node of s rows merges s - 1 rows that repeat their predecessor: node_count >= m - nsame.  When that alone is beyond the reference's 0.8 m
  // (inode.c:3962: "do not use inodes") the scan below cannot end otherwise -- scalar stencils leave here without moving m bytes to the host and walking
  // them (27-pt 512^3: 0.56 s of the first product, round 5)
  if ((double)((int64_t)m - (int64_t)nsame) > .8 * (double)m) {
    HIPX_HIP(hipFree(d_same));
    return HIPX_SUCCESS;
  }
  std::vector<unsigned char> same((size_t)m, 0);
  HIPX_HIP(hipMemcpyAsync(same.data(), d_same, (size_t)m - 1, hipMemcpyDeviceToHost, st));
  HIPX_HIP(hipStreamSynchronize(st));
  HIPX_HIP(hipFree(d_same));
  constexpr hipx_int limit = 5;
  sizes.push_back(0);
  hipx_int i = 0, nc = 0;
  while (i < m) {
    hipx_int j = i + 1, blk = 1;
    for (; j < m && blk < limit; ++j, ++blk)
      if (!same[(size_t)j - 1]) break;
    sizes.push_back(sizes.back() + blk);
    nc++;
    i = j;
  }
  if ((double)nc > .8 * (double)m) {  // inode.c:3962
    sizes.clear();
    nc = 0;
  }
  *node_count = nc;
  return HIPX_SUCCESS;
}

// node levels of the symmetrised node graph (as build_schedule does for rows), wave-aligned slots, the per-node tables
int inode_build(InodeState *T, hipx_int m, int64_t nnz, int is64, const void *d_i, const hipx_int *d_j, const int64_t *d_diagpos, hipx_int nnodes, const hipx_int *sizes)
{
  std::vector<int64_t>  hi((size_t)m + 1), hd((size_t)m);
  std::vector<hipx_int> hj((size_t)nnz);
  if (is64) HIPX_HIP(hipMemcpy(hi.data(), d_i, sizeof(int64_t) * ((size_t)m + 1), hipMemcpyDeviceToHost));
  else {
    std::vector<hipx_int> tmp((size_t)m + 1);
    HIPX_HIP(hipMemcpy(tmp.data(), d_i, sizeof(hipx_int) * ((size_t)m + 1), hipMemcpyDeviceToHost));
    for (hipx_int r = 0; r <= m; r++) hi[r] = tmp[r];
  }
  if (nnz) HIPX_HIP(hipMemcpy(hj.data(), d_j, sizeof(hipx_int) * (size_t)nnz, hipMemcpyDeviceToHost));
  HIPX_HIP(hipMemcpy(hd.data(), d_diagpos, sizeof(int64_t) * (size_t)m, hipMemcpyDeviceToHost));
  std::vector<hipx_int> r2n((size_t)m);
  int nsm = 1;
  for (hipx_int u = 0; u < nnodes; u++) {
    for (hipx_int r = sizes[u]; r < sizes[u + 1]; r++) r2n[r] = u;
    nsm = std::max(nsm, (int)(sizes[u + 1] - sizes[u]));
  }
  // what the reference's loops assume of a node (inode.c:2529-2531, 2719-2721): every row has the node's column list (given: that is how
  // the nodes were found, or what the caller vouches for) and the node's own columns sit together in it, starting at the first row's diagonal
  for (hipx_int u = 0; u < nnodes; u++) {
    const hipx_int r0 = sizes[u], ns = sizes[u + 1] - sizes[u];
    const int64_t  len = hi[r0 + 1] - hi[r0], szl = hd[r0] - hi[r0];
    bool           ok  = hd[r0] >= 0 && szl + ns <= len;
    for (hipx_int r = 0; ok && r < ns; r++) ok = (hi[r0 + r + 1] - hi[r0 + r] == len) && hj[hi[r0] + szl + r] == r0 + r;
    for (hipx_int r = 1; ok && r < ns; r++) ok = len == 0 || memcmp(&hj[hi[r0]], &hj[hi[r0 + r]], (size_t)len * sizeof(hipx_int)) == 0;  // (a caller's partition is not taken on trust)
    if (!ok) return fail(73 /* PETSC_ERR_ARG_WRONGSTATE */, "inodes: a node's rows must share one column list that holds the node's own columns (the diagonal block)", __FILE__, __LINE__);
  }
  std::vector<hipx_int> lev((size_t)nnodes, 0);
  hipx_int              nlev = 0;
  for (hipx_int u = 0; u < nnodes; u++) {
    hipx_int      l  = lev[u];
    const int64_t k0 = hi[sizes[u]], k1 = hi[sizes[u] + 1];
    for (int64_t k = k0; k < k1; k++) {
      const hipx_int j = hj[k];
      if (j >= 0 && j < m && r2n[j] < u) l = std::max(l, lev[r2n[j]] + 1);
    }
    lev[u] = l;
    for (int64_t k = k0; k < k1; k++) {
      const hipx_int j = hj[k];
      if (j >= 0 && j < m && r2n[j] > u) lev[r2n[j]] = std::max(lev[r2n[j]], l + 1);
    }
    nlev = std::max(nlev, l + 1);
  }
  std::vector<hipx_int> lp((size_t)nlev + 1, 0);
  for (hipx_int u = 0; u < nnodes; u++) lp[lev[u] + 1]++;
  for (hipx_int l = 0; l < nlev; l++) lp[l + 1] += lp[l];
  std::vector<hipx_int> perm((size_t)nnodes), fill(lp.begin(), lp.end() - 1);
  for (hipx_int u = 0; u < nnodes; u++) perm[fill[lev[u]]++] = u;
  std::vector<int4>    nmeta((size_t)nnodes);
  std::vector<int64_t> nks((size_t)nnodes + 1, 0);
  for (hipx_int p = 0; p < nnodes; p++) {
    const hipx_int u = perm[p], r0 = sizes[u];
    nmeta[p]   = make_int4(r0, sizes[u + 1] - r0, (int)(hd[r0] - hi[r0]), (int)(hi[r0 + 1] - hi[r0]));
    nks[p + 1] = nks[p] + (hi[r0 + 1] - hi[r0]);
  }
  std::vector<int4>     smeta, smeta4;
  std::vector<int64_t>  sks, sks4;
  std::vector<hipx_int> sp, sp4;
  const auto slots = [&](size_t pad, std::vector<int4> &sm, std::vector<int64_t> &sk, std::vector<hipx_int> &spp) {
    for (hipx_int l = 0; l < nlev; l++) {
      for (hipx_int p = lp[l]; p < lp[l + 1]; p++) {
        sm.push_back(nmeta[p]);
        sk.push_back(nks[p]);
        spp.push_back(p);
      }
      while (sm.size() % pad) {
        sm.push_back(make_int4(-1, 0, 0, 0));
        sk.push_back(0);
        spp.push_back(0);
      }
    }
  };
  slots(64, smeta, sks, sp);
  slots(4, smeta4, sks4, sp4);
  T->nnodes   = nnodes;
  T->nlevels  = nlev;
  T->nslots   = (hipx_int)smeta.size();
  T->nsm      = std::max(nsm, 2);
  T->nentries = nks[nnodes];
  const size_t ne = (size_t)std::max<int64_t>(T->nentries, 1);
  HIPX_HIP(hipMalloc((void **)&T->d_nmeta, sizeof(int4) * (size_t)nnodes));
  HIPX_HIP(hipMalloc((void **)&T->d_nks, sizeof(int64_t) * ((size_t)nnodes + 1)));
  HIPX_HIP(hipMalloc((void **)&T->d_smeta, sizeof(int4) * smeta.size()));
  HIPX_HIP(hipMalloc((void **)&T->d_sks, sizeof(int64_t) * sks.size()));
  HIPX_HIP(hipMalloc((void **)&T->d_sp, sizeof(hipx_int) * sp.size()));
  HIPX_HIP(hipMalloc((void **)&T->d_nj, sizeof(hipx_int) * ne));
  HIPX_HIP(hipMalloc((void **)&T->d_nv, sizeof(double) * ne * (size_t)T->nsm));
  HIPX_HIP(hipMalloc((void **)&T->d_ibd, sizeof(double) * (size_t)nnodes * (size_t)(T->nsm * T->nsm)));
  HIPX_HIP(hipMalloc((void **)&T->d_bd, sizeof(double) * (size_t)nnodes * (size_t)(T->nsm * T->nsm)));
  HIPX_HIP(hipMemcpy(T->d_nmeta, nmeta.data(), sizeof(int4) * (size_t)nnodes, hipMemcpyHostToDevice));
  HIPX_HIP(hipMemcpy(T->d_nks, nks.data(), sizeof(int64_t) * ((size_t)nnodes + 1), hipMemcpyHostToDevice));
  HIPX_HIP(hipMemcpy(T->d_smeta, smeta.data(), sizeof(int4) * smeta.size(), hipMemcpyHostToDevice));
  HIPX_HIP(hipMemcpy(T->d_sks, sks.data(), sizeof(int64_t) * sks.size(), hipMemcpyHostToDevice));
  HIPX_HIP(hipMemcpy(T->d_sp, sp.data(), sizeof(hipx_int) * sp.size(), hipMemcpyHostToDevice));
  T->nslots4 = (hipx_int)smeta4.size();
  HIPX_HIP(hipMalloc((void **)&T->d_smeta4, sizeof(int4) * std::max<size_t>(smeta4.size(), 1)));
  HIPX_HIP(hipMalloc((void **)&T->d_sks4, sizeof(int64_t) * std::max<size_t>(sks4.size(), 1)));
  HIPX_HIP(hipMalloc((void **)&T->d_sp4, sizeof(hipx_int) * std::max<size_t>(sp4.size(), 1)));
  HIPX_HIP(hipMemcpy(T->d_smeta4, smeta4.data(), sizeof(int4) * smeta4.size(), hipMemcpyHostToDevice));
  HIPX_HIP(hipMemcpy(T->d_sks4, sks4.data(), sizeof(int64_t) * sks4.size(), hipMemcpyHostToDevice));
  HIPX_HIP(hipMemcpy(T->d_sp4, sp4.data(), sizeof(hipx_int) * sp4.size(), hipMemcpyHostToDevice));
  T->ready = true;
  return HIPX_SUCCESS;
}

// the node-level copy of the values (+ the shared column lists) and the diagonal blocks with their inverses: once per set of values
__global__ void inode_pack_kernel(hipx_int nnodes, int nsm, const int4 *__restrict__ nmeta, const int64_t *__restrict__ nks, const void *ai_, int is64, const hipx_int *__restrict__ aj,
                                  const double *__restrict__ aa, hipx_int *__restrict__ nj, double *__restrict__ nv)
{
  // one wave per node: lanes walk the entries
  const int      lane = threadIdx.x & 63;
  const hipx_int w0 = (hipx_int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), nw = (hipx_int)((gridDim.x * blockDim.x) >> 6);
  for (hipx_int p = w0; p < nnodes; p += nw) {
    const int4    mt = nmeta[p];
    const int64_t ks = nks[p];
    for (int r = 0; r < nsm; r++) {
      const int64_t s = (r < mt.y) ? (is64 ? ((const int64_t *)ai_)[mt.x + r] : (int64_t)((const hipx_int *)ai_)[mt.x + r]) : 0;
      for (int k = lane; k < mt.w; k += 64) {
        nv[(ks + k) * nsm + r] = (r < mt.y) ? aa[s + k] : 0.0;
        if (r == 0) nj[ks + k] = aj[s + k];
      }
    }
  }
}

// MatInvertDiagonalForSOR_SeqAIJ_Inode (inode.c:2449-2489): the node's block, column-major (element (j, k) = a[diag[row + j] - j + k]), and
// its inverse by PetscKernel_A_gets_inverse_A_<n> (dgefa2.c:14, dgefa3.c:14, dgefa4.c, dgefa5.c:14 with shift = 0): LINPACK dgefa -- the
// first largest entry of the column is the pivot, multipliers -1 / pivot, column updates y += t x -- then dgedi: inverse(U) column by
// column, inverse(U) inverse(L) from the last column back, the interchanges undone on the columns.  One thread per node.
__global__ void inode_invert_kernel(hipx_int nnodes, int nsm, const int4 *__restrict__ nmeta, const int64_t *__restrict__ diagpos, const double *__restrict__ aa,
                                    double *__restrict__ ibd, double *__restrict__ bd, unsigned int *zero_pivots)
{
  for (hipx_int p = (hipx_int)blockIdx.x * blockDim.x + threadIdx.x; p < nnodes; p += (hipx_int)gridDim.x * blockDim.x) {
    const int4 mt = nmeta[p];
    const int  n  = mt.y;
    double     a[25], work[5];
    int        ipvt[5];
    bool       zero = false;
    for (int j = 0; j < n; j++)
      for (int k = 0; k < n; k++) a[k * n + j] = aa[diagpos[mt.x + j] - j + k];
    double *bo = bd + (size_t)p * (size_t)(nsm * nsm), *io = ibd + (size_t)p * (size_t)(nsm * nsm);
    for (int e = 0; e < n * n; e++) bo[e] = a[e];
#define A_(i, j) a[(i) + (j) * n]
    if (n == 1) {
      if (fabs(a[0]) < 100. * 2.220446049250313e-16) zero = true;  // inode.c:2459
      a[0] = 1.0 / a[0];
    } else {
      for (int k = 0; k < n - 1; k++) {
        int    l   = k;
        double max = fabs(A_(k, k));
        for (int i = k + 1; i < n; i++)
          if (fabs(A_(i, k)) > max) {
            max = fabs(A_(i, k));
            l   = i;
          }
        ipvt[k] = l;
        if (A_(l, k) == 0.0) zero = true;
        if (l != k) {
          const double t = A_(l, k);
          A_(l, k)       = A_(k, k);
          A_(k, k)       = t;
        }
        const double tm = -1. / A_(k, k);
        for (int i = k + 1; i < n; i++) A_(i, k) *= tm;
        for (int j = k + 1; j < n; j++) {
          const double t = A_(l, j);
          if (l != k) {
            A_(l, j) = A_(k, j);
            A_(k, j) = t;
          }
          for (int i = k + 1; i < n; i++) A_(i, j) += t * A_(i, k);
        }
      }
      ipvt[n - 1] = n - 1;
      if (A_(n - 1, n - 1) == 0.0) zero = true;
      for (int k = 0; k < n; k++) {
        A_(k, k)        = 1.0 / A_(k, k);
        const double tk = -A_(k, k);
        for (int i = 0; i < k; i++) A_(i, k) *= tk;
        for (int j = k + 1; j < n; j++) {
          const double t = A_(k, j);
          A_(k, j)       = 0.0;
          for (int i = 0; i <= k; i++) A_(i, j) += t * A_(i, k);
        }
      }
      for (int k = n - 2; k >= 0; k--) {
        for (int i = k + 1; i < n; i++) {
          work[i]  = A_(i, k);
          A_(i, k) = 0.0;
        }
        for (int j = k + 1; j < n; j++) {
          const double t = work[j];
          for (int i = 0; i < n; i++) A_(i, k) += t * A_(i, j);
        }
        if (ipvt[k] != k)
          for (int i = 0; i < n; i++) {
            const double t = A_(i, k);
            A_(i, k)       = A_(i, ipvt[k]);
            A_(i, ipvt[k]) = t;
          }
      }
    }
#undef A_
    for (int e = 0; e < n * n; e++) io[e] = a[e];
    if (zero) atomicAdd(zero_pivots, 1u);
  }
}

// the entries [k0, k1) of a node, in pairs from k0 (inode.c:2589-2608): sum_r -= v_r[k] x[j_k] + v_r[k+1] x[j_k+1]; a last odd entry alone.
// SRC 0: every operand is a NEW value (polled out of xnew), 1: every operand an old one (xold), 2: columns >= thr new, the others old
template <int NSM, int SRC>
__device__ __forceinline__ void inode_minus(double (&sum)[NSM], int64_t k0, int64_t k1, hipx_int thr, const hipx_int *__restrict__ nj, const double *__restrict__ nv, const double *xold,
                                            const double *xnew, unsigned int *err)
{
  constexpr int CH = (NSM <= 3) ? 8 : 4;  // entries whose column / value loads are in flight before the first poll (even: pairs never straddle chunks)
  for (int64_t k = k0; k < k1; k += CH) {
    hipx_int  j[CH];
    double    a[CH][NSM], v[CH];
    const int nk = (int)((k1 - k) < CH ? (k1 - k) : CH);
#pragma unroll
    for (int c = 0; c < CH; c++) {
      const int64_t kk = (c < nk) ? k + c : k;
      j[c]             = nj[kk];
#pragma unroll
      for (int r = 0; r < NSM; r++) a[c][r] = nv[kk * NSM + r];
    }
#pragma unroll
    for (int c = 0; c < CH; c++) {
      const bool dep = SRC == 0 || (SRC == 2 && j[c] >= thr);
      if (dep) v[c] = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(xnew + j[c]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      else v[c] = xold[j[c]];
    }
#pragma unroll
    for (int c = 0; c < CH; c++) {
      const bool dep = SRC == 0 || (SRC == 2 && j[c] >= thr);
      if (c < nk && dep && (unsigned long long)__double_as_longlong(v[c]) == SOR_SENTINEL) v[c] = sor_poll(xnew + j[c], err);
    }
#pragma unroll
    for (int c = 0; c < CH; c += 2) {
      if (c + 1 < nk) {
#pragma unroll
        for (int r = 0; r < NSM; r++) sum[r] -= a[c][r] * v[c] + a[c + 1][r] * v[c + 1];
      } else if (c < nk) {
#pragma unroll
        for (int r = 0; r < NSM; r++) sum[r] -= a[c][r] * v[c];
      }
    }
  }
}

// KIND 0: zero-guess forward   (inode.c:2527-2712)  s = b - L x_new;  t = s;  x = D^-1 s
// KIND 1: backward after a forward sweep, rhs = t   (inode.c:2714-2888, 3238-3253)  x = D^-1 (t - U x_new)
// KIND 2: backward, zero guess, rhs = b             (the same loops with xb = b; Eisenstat's first step inode.c:3379-3553)
// KIND 3: forward, general     (inode.c:2892-3207)  s = b - L x_new;  t = s;  x = D^-1 (s - U x_old)
// KIND 4: backward alone, general (inode.c:3219-3237, 3311-3339): the WHOLE rows, block included, in one run of pairs:  x = x_old + D^-1 (b - A x)
// KIND 5: Eisenstat's last step (inode.c:3634-3804): forward on t:  t_new = D^-1 (t - L t_new);  x += t_new
template <int KIND, int NSM>
__global__ __launch_bounds__(SOR_THREADS) void sor_inode_kernel(hipx_int nslots, const int4 *__restrict__ smeta, const int64_t *__restrict__ sks, const hipx_int *__restrict__ sp,
                                                                 const hipx_int *__restrict__ nj, const double *__restrict__ nv, const double *__restrict__ ibd, const double *rhs, double *t,
                                                                 const double *xold, double *xnew, double *xacc, unsigned int *ctl)
{
  constexpr bool FWD = (KIND == 0 || KIND == 3 || KIND == 5);
  unsigned int  *err = ctl + 1;
  const int      lane = threadIdx.x & 63;
  const hipx_int ngroups = nslots >> 6;
  for (;;) {
    unsigned int v = 0;
    if (lane == 0) v = atomicAdd(&ctl[0], 1u);
    v = __shfl(v, 0, 64);
    if ((hipx_int)v >= ngroups) return;
    const hipx_int g  = (hipx_int)v * 64 + lane;
    const hipx_int s  = FWD ? g : nslots - 1 - g;
    const int4     mt = smeta[s];  // {first row, rows, entries before the block, entries per row}; first row < 0: padding
    if (mt.x >= 0) {
      const int64_t  ks = sks[s];
      const hipx_int r0 = mt.x;
      const int      ns = mt.y;
      const double  *D  = ibd + (size_t)sp[s] * (size_t)(NSM * NSM);
      double         sum[NSM], out[NSM];
#pragma unroll
      for (int r = 0; r < NSM; r++) sum[r] = (r < ns) ? rhs[r0 + r] : 0.0;
      if (KIND == 0 || KIND == 3 || KIND == 5) {
        inode_minus<NSM, 0>(sum, ks, ks + mt.z, 0, nj, nv, xold, xnew, err);
        if (KIND != 5) {
#pragma unroll
          for (int r = 0; r < NSM; r++)
            if (r < ns) t[r0 + r] = sum[r];
        }
        if (KIND == 3) inode_minus<NSM, 1>(sum, ks + mt.z + ns, ks + mt.w, 0, nj, nv, xold, xnew, err);
      } else if (KIND == 1 || KIND == 2) {
        inode_minus<NSM, 0>(sum, ks + mt.z + ns, ks + mt.w, 0, nj, nv, xold, xnew, err);
      } else {
        inode_minus<NSM, 2>(sum, ks, ks + mt.w, r0 + ns, nj, nv, xold, xnew, err);
      }
      // x_r = sum_c s_c D^-1[r, c], c ascending (inode.c:2612-2614; the backward loops write the same expression from the last row up)
#pragma unroll
      for (int r = 0; r < NSM; r++) {
        double acc = sum[0] * D[r < ns ? r : 0];
#pragma unroll
        for (int c = 1; c < NSM; c++)
          if (c < ns) acc = acc + sum[c] * D[(r < ns) ? c * ns + r : 0];
        out[r] = acc;
      }
#pragma unroll
      for (int r = 0; r < NSM; r++)
        if (r < ns) {
          if (KIND == 4) out[r] = xold[r0 + r] + out[r];
          sor_publish(xnew + r0 + r, out[r]);
          if (KIND == 5) xacc[r0 + r] += out[r];
        }
    }
  }
}

// The cooperative form (default): 16 lanes per node, 4 nodes per wave.  With one lane per node a lane walks its ~40 entries in chunks and
// every chunk's loads wait behind the polls of the chunk before: ~13 us per dependency level on the elasticity stand-in (1344 levels).
// Here the PAIRS of a segment are dealt round-robin to the node's 16 lanes -- all column, value and operand loads of (up to) 64 entries in
// flight at once, coalesced -- each lane forms its pairs' terms q = a[k] x[j_k] + a[k+1] x[j_k+1] for the node's rows and parks them in
// LDS; then every lane of the group runs the SAME sequential chain sum_r -= q_r over the pairs in their order (LDS broadcasts), so the
// rounding is the reference's; lane r < (rows of the node) then owns row r: block row times the sums, publish.  Waves take tickets in
// level order; with 8 waves per SIMD-slot-free CU the loads of ~20 levels ahead are already waiting in their polls.
constexpr int INO_G = 16, INO_R = 2, INO_CHP = INO_G * INO_R;  // lanes per node, pairs per lane and chunk, pairs per chunk

template <int NSM, int SRC>
__device__ __forceinline__ void inode_coop_minus(double (&sum)[NSM], double (*Q)[NSM], const int l, int64_t k0, int64_t k1, hipx_int thr, const hipx_int *__restrict__ nj,
                                                 const double *__restrict__ nv, const double *xold, const double *xnew, unsigned int *err, const int psleep)
{
  for (int64_t kc = k0; kc < k1; kc += 2 * INO_CHP) {
    const int cnt = (int)((k1 - kc) < 2 * INO_CHP ? (k1 - kc) : 2 * INO_CHP);  // entries of this chunk
    const int np  = (cnt + 1) >> 1;
    hipx_int  j0[INO_R], j1[INO_R];
    double    a0[INO_R][NSM], a1[INO_R][NSM], x0[INO_R], x1[INO_R];
    bool      v0[INO_R], v1[INO_R];
#pragma unroll
    for (int rr = 0; rr < INO_R; rr++) {
      const int q = l + rr * INO_G;
      v0[rr]      = 2 * q < cnt;
      v1[rr]      = 2 * q + 1 < cnt;
      const int64_t e0 = v0[rr] ? kc + 2 * q : k0, e1 = v1[rr] ? kc + 2 * q + 1 : e0;
      j0[rr] = nj[e0];
      j1[rr] = nj[e1];
#pragma unroll
      for (int r = 0; r < NSM; r++) {
        a0[rr][r] = nv[e0 * NSM + r];
        a1[rr][r] = nv[e1 * NSM + r];
      }
    }
#pragma unroll
    for (int rr = 0; rr < INO_R; rr++) {
      const bool d0 = SRC == 0 || (SRC == 2 && j0[rr] >= thr), d1 = SRC == 0 || (SRC == 2 && j1[rr] >= thr);
      if (d0) x0[rr] = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(xnew + j0[rr]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      else x0[rr] = xold[j0[rr]];
      if (d1) x1[rr] = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(xnew + j1[rr]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      else x1[rr] = xold[j1[rr]];
    }
#pragma unroll
    for (int rr = 0; rr < INO_R; rr++) {
      const bool d0 = SRC == 0 || (SRC == 2 && j0[rr] >= thr), d1 = SRC == 0 || (SRC == 2 && j1[rr] >= thr);
      if (v0[rr] && d0 && (unsigned long long)__double_as_longlong(x0[rr]) == SOR_SENTINEL) x0[rr] = sor_poll_sel(xnew + j0[rr], err, psleep);
      if (v1[rr] && d1 && (unsigned long long)__double_as_longlong(x1[rr]) == SOR_SENTINEL) x1[rr] = sor_poll_sel(xnew + j1[rr], err, psleep);
    }
#pragma unroll
    for (int rr = 0; rr < INO_R; rr++) {
      const int q = l + rr * INO_G;
      if (v0[rr]) {
#pragma unroll
        for (int r = 0; r < NSM; r++) {
          const double p0 = a0[rr][r] * x0[rr];
          Q[q][r]         = v1[rr] ? p0 + a1[rr][r] * x1[rr] : p0;  // the last odd entry of a segment stands alone (inode.c:2603-2608)
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {  // the sequential chain sum_r -= q_r over the pairs in their order.  Round 6: the terms of FOUR pairs are read from LDS before the first of their
       // subtractions (the loop as written above read, waited ~64 clocks and subtracted once per pair: ~0.6 us of every dependency level's ~2.6 us
       // were LDS latencies of this chain); the subtractions themselves keep their order.
      int q = 0;
      for (; q + 4 <= np; q += 4) {
        double tq[4][NSM];
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
          for (int r = 0; r < NSM; r++) tq[u][r] = Q[q + u][r];
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
          for (int r = 0; r < NSM; r++) asm volatile("" : "+v"(tq[u][r]));  // (all twelve reads issued; none re-materialised behind a subtraction)
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
          for (int r = 0; r < NSM; r++) sum[r] -= tq[u][r];
      }
      for (; q < np; q++) {
#pragma unroll
        for (int r = 0; r < NSM; r++) sum[r] -= Q[q][r];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

template <int KIND, int NSM>
__global__ __launch_bounds__(SOR_THREADS) void sor_inode_coop_kernel(hipx_int nslots, const int4 *__restrict__ smeta, const int64_t *__restrict__ sks, const hipx_int *__restrict__ sp,
                                                                      const hipx_int *__restrict__ nj, const double *__restrict__ nv, const double *__restrict__ ibd, const double *rhs,
                                                                      double *t, const double *xold, double *xnew, double *xacc, unsigned int *ctl, const int psleep)
{
  constexpr bool FWD = (KIND == 0 || KIND == 3 || KIND == 5);
  __shared__ double s_q[SOR_THREADS / 64][64 / INO_G][INO_CHP][NSM];
  unsigned int  *err = ctl + 1;
  const int      lane = threadIdx.x & 63, wv = threadIdx.x >> 6, grp = lane / INO_G, l = lane % INO_G;
  double(*Q)[NSM]     = s_q[wv][grp];
  const hipx_int ngroups = nslots / (64 / INO_G);
  for (;;) {
    unsigned int v = 0;
    if (lane == 0) v = atomicAdd(&ctl[0], 1u);
    v = __shfl(v, 0, 64);
    if ((hipx_int)v >= ngroups) return;
    const hipx_int g  = (hipx_int)v * (64 / INO_G) + grp;
    const hipx_int s  = FWD ? g : nslots - 1 - g;
    const int4     mt = smeta[s];
    if (mt.x >= 0) {
      const int64_t  ks = sks[s];
      const hipx_int r0 = mt.x;
      const int      ns = mt.y;
      const double  *D  = ibd + (size_t)sp[s] * (size_t)(NSM * NSM);
      double         sum[NSM];
#pragma unroll
      for (int r = 0; r < NSM; r++) sum[r] = (r < ns) ? rhs[r0 + r] : 0.0;
      double dcol[NSM];  // row l of the block inverse: D^-1[l, c] = D[c * ns + l]
#pragma unroll
      for (int c = 0; c < NSM; c++) dcol[c] = (l < ns && c < ns) ? D[c * ns + l] : 0.0;
      const double xo = (KIND == 4 && l < ns) ? xold[r0 + l] : 0.0;
      if (KIND == 0 || KIND == 3 || KIND == 5) {
        inode_coop_minus<NSM, 0>(sum, Q, l, ks, ks + mt.z, 0, nj, nv, xold, xnew, err, psleep);
        if (KIND != 5 && l < ns) {
          double sl = sum[0];
#pragma unroll
          for (int c = 1; c < NSM; c++) sl = (l == c) ? sum[c] : sl;
          t[r0 + l] = sl;
        }
        if (KIND == 3) inode_coop_minus<NSM, 1>(sum, Q, l, ks + mt.z + ns, ks + mt.w, 0, nj, nv, xold, xnew, err, psleep);
      } else if (KIND == 1 || KIND == 2) {
        inode_coop_minus<NSM, 0>(sum, Q, l, ks + mt.z + ns, ks + mt.w, 0, nj, nv, xold, xnew, err, psleep);
      } else {
        inode_coop_minus<NSM, 2>(sum, Q, l, ks, ks + mt.w, r0 + ns, nj, nv, xold, xnew, err, psleep);
      }
      if (l < ns) {
        double acc = sum[0] * dcol[0];
#pragma unroll
        for (int c = 1; c < NSM; c++)
          if (c < ns) acc = acc + sum[c] * dcol[c];
        if (KIND == 4) acc = xo + acc;
        sor_publish(xnew + r0 + l, acc);
        if (KIND == 5) xacc[r0 + l] += acc;
      }
    }
  }
}

// Eisenstat's middle step on the nodes (inode.c:3559-3628): t = b - D x, the block product summed over the node's columns in ascending order
__global__ void inode_eisenstat_mid_kernel(hipx_int nnodes, int nsm, const int4 *__restrict__ nmeta, const double *__restrict__ bd, const double *__restrict__ b, const double *__restrict__ x,
                                           double *__restrict__ t)
{
  for (hipx_int p = (hipx_int)blockIdx.x * blockDim.x + threadIdx.x; p < nnodes; p += (hipx_int)gridDim.x * blockDim.x) {
    const int4    mt = nmeta[p];
    const int     ns = mt.y;
    const double *D  = bd + (size_t)p * (size_t)(nsm * nsm);
    for (int r = 0; r < ns; r++) {
      double acc = (ns == 1) ? D[0] * x[mt.x] : x[mt.x] * D[r];
      for (int c = 1; c < ns; c++) acc = acc + x[mt.x + c] * D[c * ns + r];
      t[mt.x + r] = b[mt.x + r] - acc;
    }
  }
}

template <int KIND>
int run_inode(hipxSorState *S, const double *rhs, const double *xold, double *xnew, double *xacc)
{
  InodeState    *T  = (InodeState *)S->inode;
  hipStream_t    st = rt().compute;
  const hipx_int g  = std::min<hipx_int>((S->m + 255) / 256, 4096);
  sor_fill_kernel<<<(unsigned)g, 256, 0, st>>>(xnew, S->m);
  HIPX_HIP(hipMemsetAsync(S->d_ctl, 0, sizeof(unsigned int), st));
  static int waves_per_cu = 0;
  if (!waves_per_cu) {
    const char *e = getenv("HIPX_SOR_INODE_WAVES_PER_CU");
    waves_per_cu  = e ? atoi(e) : 2;
    if (waves_per_cu < 1) waves_per_cu = 1;
    if (waves_per_cu > 32) waves_per_cu = 32;
  }
  unsigned       grid = (unsigned)(256 * waves_per_cu * 64 / SOR_THREADS);
  const unsigned need = (unsigned)((T->nslots + SOR_THREADS - 1) / SOR_THREADS);
  if (grid > need) grid = need ? need : 1;
  // workgroups of 4 waves; measured on the elasticity stand-in (symmetric sweep): 256 (one per CU) 7.2 ms, 512: 8.1, 1024: 13.4, 2048: 18.8 --
  // pollers crowd out the publishers.  Wide levels (> 2048 nodes on average) keep the lane-per-node form: see run_dep
  int coop = (T->nlevels > 0 && (int64_t)T->nnodes / T->nlevels <= 2048) ? 1 : 0, coop_blocks = 256;
  {
    const char *e = getenv("HIPX_SOR_INODE_COOP");  // 0: one lane per node (the first form of this schedule: 34.8 ms)
    if (e) coop = atoi(e);
    e = getenv("HIPX_SOR_INODE_COOP_BLOCKS");
    if (e) coop_blocks = atoi(e);
    if (coop_blocks < 1) coop_blocks = 1;
    if (coop_blocks > 4096) coop_blocks = 4096;
  }
  if (coop) {
    const int      psleep = getenv("HIPX_SOR_COOP_SLEEP") ? atoi(getenv("HIPX_SOR_COOP_SLEEP")) : 1;  // back-off between two looks of a poll: s_sleep 8 / 4 / 2 / 1 / none = 7.23 / 7.13 / 7.08 / 7.06 / 7.06 ms
    unsigned       cgrid = (unsigned)coop_blocks;
    const unsigned cneed = (unsigned)((T->nslots4 / (64 / INO_G) * 64 + SOR_THREADS - 1) / SOR_THREADS);
    if (cgrid > cneed) cgrid = cneed ? cneed : 1;
#define HIPX_INODE_COOP(NSM) \
  sor_inode_coop_kernel<KIND, NSM><<<cgrid, SOR_THREADS, 0, st>>>(T->nslots4, T->d_smeta4, T->d_sks4, T->d_sp4, T->d_nj, T->d_nv, T->d_ibd, rhs, S->d_t, xold, xnew, xacc, S->d_ctl, psleep)
    switch (T->nsm) {
    case 2: HIPX_INODE_COOP(2); break;
    case 3: HIPX_INODE_COOP(3); break;
    case 4: HIPX_INODE_COOP(4); break;
    default: HIPX_INODE_COOP(5); break;
    }
#undef HIPX_INODE_COOP
    HIPX_LAUNCH_CHECK();
    return HIPX_SUCCESS;
  }
#define HIPX_INODE_LAUNCH(NSM) \
  sor_inode_kernel<KIND, NSM><<<grid, SOR_THREADS, 0, st>>>(T->nslots, T->d_smeta, T->d_sks, T->d_sp, T->d_nj, T->d_nv, T->d_ibd, rhs, S->d_t, xold, xnew, xacc, S->d_ctl)
  switch (T->nsm) {
  case 2: HIPX_INODE_LAUNCH(2); break;
  case 3: HIPX_INODE_LAUNCH(3); break;
  case 4: HIPX_INODE_LAUNCH(4); break;
  default: HIPX_INODE_LAUNCH(5); break;
  }
#undef HIPX_INODE_LAUNCH
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}

int build_schedule(hipxSorState *S, hipx_int m, int64_t nnz, int is64, const void *d_i, const hipx_int *d_j)
{
  // host copy of the pattern (set-up only; the sweeps never touch the host)
  std::vector<int64_t>  hi((size_t)m + 1);
  std::vector<hipx_int> hj((size_t)nnz);
  if (is64) HIPX_HIP(hipMemcpy(hi.data(), d_i, sizeof(int64_t) * ((size_t)m + 1), hipMemcpyDeviceToHost));
  else {
    std::vector<hipx_int> tmp((size_t)m + 1);
    HIPX_HIP(hipMemcpy(tmp.data(), d_i, sizeof(hipx_int) * ((size_t)m + 1), hipMemcpyDeviceToHost));
    for (hipx_int r = 0; r <= m; r++) hi[r] = tmp[r];
  }
  if (nnz) HIPX_HIP(hipMemcpy(hj.data(), d_j, sizeof(hipx_int) * (size_t)nnz, hipMemcpyDeviceToHost));
  std::vector<hipx_int> lev((size_t)m, 0);
  hipx_int              nlev = 0;
  for (hipx_int i = 0; i < m; i++) {
    hipx_int l = lev[i];  // already raised by earlier rows that have i in their upper part
    for (int64_t k = hi[i]; k < hi[i + 1]; k++) {
      const hipx_int j = hj[k];
      if (j < i && j >= 0) l = std::max(l, lev[j] + 1);
    }
    lev[i] = l;
    for (int64_t k = hi[i]; k < hi[i + 1]; k++) {
      const hipx_int j = hj[k];
      if (j > i && j < m) lev[j] = std::max(lev[j], l + 1);
    }
    nlev = std::max(nlev, l + 1);
  }
  S->nlevels = nlev;
  S->lev_ptr.assign((size_t)nlev + 1, 0);
  for (hipx_int i = 0; i < m; i++) S->lev_ptr[lev[i] + 1]++;
  for (hipx_int l = 0; l < nlev; l++) S->lev_ptr[l + 1] += S->lev_ptr[l];
  std::vector<hipx_int> perm((size_t)m), fill(S->lev_ptr.begin(), S->lev_ptr.end() - 1);
  for (hipx_int i = 0; i < m; i++) perm[fill[lev[i]]++] = i;  // rows of a level stay in increasing order
  std::vector<int64_t> pi((size_t)m + 1, 0);
  for (hipx_int p = 0; p < m; p++) pi[p + 1] = pi[p] + (hi[perm[p] + 1] - hi[perm[p]]);
  HIPX_HIP(hipMalloc((void **)&S->d_perm, sizeof(hipx_int) * (size_t)m));
  HIPX_HIP(hipMalloc((void **)&S->d_pi, sizeof(int64_t) * ((size_t)m + 1)));
  HIPX_HIP(hipMalloc((void **)&S->d_pd, sizeof(hipx_int) * (size_t)m));
  HIPX_HIP(hipMalloc((void **)&S->d_pj, sizeof(hipx_int) * (size_t)(nnz ? nnz : 1)));
  HIPX_HIP(hipMalloc((void **)&S->d_pa, sizeof(double) * (size_t)(nnz ? nnz : 1)));
  HIPX_HIP(hipMemcpy(S->d_perm, perm.data(), sizeof(hipx_int) * (size_t)m, hipMemcpyHostToDevice));
  HIPX_HIP(hipMemcpy(S->d_pi, pi.data(), sizeof(int64_t) * ((size_t)m + 1), hipMemcpyHostToDevice));
  {  // wave-aligned slot map: every level starts on a multiple of 64 slots
    std::vector<hipx_int> slot;
    slot.reserve((size_t)m + (size_t)nlev * 64);
    for (hipx_int l = 0; l < nlev; l++) {
      for (hipx_int p = S->lev_ptr[l]; p < S->lev_ptr[l + 1]; p++) slot.push_back(p);
      while (slot.size() % 64) slot.push_back(-1);
    }
    S->nslots = (hipx_int)slot.size();
    HIPX_HIP(hipMalloc((void **)&S->d_slot, sizeof(hipx_int) * std::max<size_t>(slot.size(), 1)));
    if (!slot.empty()) HIPX_HIP(hipMemcpy(S->d_slot, slot.data(), sizeof(hipx_int) * slot.size(), hipMemcpyHostToDevice));
    slot.clear();  // ... and on a multiple of 4 for the cooperative kernel
    for (hipx_int l = 0; l < nlev; l++) {
      for (hipx_int p = S->lev_ptr[l]; p < S->lev_ptr[l + 1]; p++) slot.push_back(p);
      while (slot.size() % 4) slot.push_back(-1);
    }
    S->nslots4 = (hipx_int)slot.size();
    HIPX_HIP(hipMalloc((void **)&S->d_slot4, sizeof(hipx_int) * std::max<size_t>(slot.size(), 1)));
    if (!slot.empty()) HIPX_HIP(hipMemcpy(S->d_slot4, slot.data(), sizeof(hipx_int) * slot.size(), hipMemcpyHostToDevice));
  }
  S->m     = m;
  S->is64  = is64 != 0;
  S->ready = true;
  return HIPX_SUCCESS;
}

}  // namespace

extern "C" void hipxSorInvalidate_(void *p)
{
  hipxSorState *S = (hipxSorState *)p;
  if (S) S->values_valid = S->idiag_valid = S->mdiag_valid = false;
  if (S && S->inode) ((InodeState *)S->inode)->values_valid = false;
}

extern "C" void hipxSorStateFree_(void *p)
{
  hipxSorState *S = (hipxSorState *)p;
  if (!S) return;
  (void)hipFree(S->d_perm);
  (void)hipFree(S->d_pi);
  (void)hipFree(S->d_pd);
  (void)hipFree(S->d_pj);
  (void)hipFree(S->d_pa);
  (void)hipFree(S->d_idiag);
  (void)hipFree(S->d_mdiag);
  (void)hipFree(S->d_t);
  (void)hipFree(S->d_slot);
  (void)hipFree(S->d_smeta);
  (void)hipFree(S->d_sks);
  (void)hipFree(S->d_slot4);
  (void)hipFree(S->d_smeta4);
  (void)hipFree(S->d_sks4);
  (void)hipFree(S->d_w1);
  (void)hipFree(S->d_w2);
  (void)hipFree(S->d_ctl);
  strand_free((StrandState *)S->strand);
  inode_free((InodeState *)S->inode);
  hipxSorBoxFree_(S->box);
  delete S;
}

extern "C" {
int hipxMatTemplates_(hipxMat A, int *ok, int *ntmpl, const int **tstart, const int **toff, const double **tval, const int **tdiag, const int64_t **tcount,
                      const unsigned char **d_tid);
int hipxMatPatternTemplates_(hipxMat A, int *ok, int *ntmpl, const int **tstart, const int **toff, const int **tdiag, const int64_t **tcount, const unsigned char **d_tid,
                             const int **d_tstart, const int **d_toff);
}

// the inode partition of a matrix nobody has described yet: looked for as MatAssemblyEnd_SeqAIJ does (hipx_mat.hip asks before its first product)
extern "C" int hipxMatEnsureInodes_(hipxMat A)
{
  hipx_int           m, n;
  int64_t            nnz;
  int                is64, diag_dense, compressed, istate;
  void              *d_i;
  hipx_int          *d_j, nnodes;
  const hipx_int    *isz;
  double            *d_a;
  int64_t           *d_diagpos;
  void             **slot;
  unsigned long long vstate;
  int ierr = hipxMatInternal_(A, &m, &n, &nnz, &is64, &d_i, &d_j, &d_a, &d_diagpos, &diag_dense, &compressed, &slot, &vstate);
  if (ierr) return ierr;
  if ((ierr = hipxMatInodes_(A, &istate, &nnodes, &isz))) return ierr;
  if (istate >= 0) return HIPX_SUCCESS;
  std::vector<hipx_int> sizes;
  nnodes = 0;
  // square matrices only: an off-diagonal block (rectangular in all but degenerate splits) has inodes switched off in the reference (mpiaij.c:824)
  if (!compressed && m == n && (ierr = inode_find(m, is64, d_i, d_j, sizes, &nnodes))) return ierr;
  return hipxMatInodesFound_(A, nnodes, sizes.data());
}

// schedule the last hipxMatSOR call used: 0 one launch per level, 1 level-ordered dependency-driven, 2 strands; -1 = none yet
extern "C" int hipxMatGetSORMode(hipxMat A, int *mode)
{
  HIPX_ARG(A && mode, "null argument");
  hipx_int           m, n;
  int64_t            nnz;
  int                is64, diag_dense, compressed;
  void              *d_i;
  hipx_int          *d_j;
  double            *d_a;
  int64_t           *d_diagpos;
  void             **slot;
  unsigned long long vstate;
  int ierr = hipxMatInternal_(A, &m, &n, &nnz, &is64, &d_i, &d_j, &d_a, &d_diagpos, &diag_dense, &compressed, &slot, &vstate);
  if (ierr) return ierr;
  *mode = *slot ? ((hipxSorState *)*slot)->last_mode : -1;
  return HIPX_SUCCESS;
}

namespace {
// one sweep in the active mode (2 = strands, 1 = level-ordered dependency-driven)
template <int KIND>
int run_sweep(hipxSorState *S, const double *b, const double *xold, double *xnew, double omega)
{
  if (S->mode == 2) return run_strand<KIND>((StrandState *)S->strand, KIND == 1 ? S->d_t : b, S->d_t, xold, xnew, omega);
  return run_dep<KIND>(S, b, xold, xnew, omega);
}
}  // namespace

static int mat_sor_impl(hipxMat A, const double *b, double omega, int flag, double shift, hipx_int its, hipx_int lits, double *x);

extern "C" int hipxMatSOR(hipxMat A, const double *b, double omega, int flag, double shift, hipx_int its, hipx_int lits, double *x)
{
  HIPX_CHECK_INIT();
  int ierr = prof_section(HIPX_PROF_SOR, true, rt().compute);  // bench.py: HIP events around the whole call (all its sweeps)
  if (ierr) return ierr;
  if ((ierr = mat_sor_impl(A, b, omega, flag, shift, its, lits, x))) return ierr;
  return prof_section(HIPX_PROF_SOR, false, rt().compute);
}

static int mat_sor_impl(hipxMat A, const double *b, double omega, int flag, double shift, hipx_int its, hipx_int lits, double *x)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(A && b && x, "null argument");
  HIPX_ARG(its > 0 && lits > 0, "Relaxation requires global its and local its > 0 (matrix.c:4377)");
  hipx_int           m, n;
  int64_t            nnz;
  int                is64, diag_dense, compressed;
  void              *d_i;
  hipx_int          *d_j;
  double            *d_a;
  int64_t           *d_diagpos;
  void             **slot;
  unsigned long long vstate;
  int ierr = hipxMatInternal_(A, &m, &n, &nnz, &is64, &d_i, &d_j, &d_a, &d_diagpos, &diag_dense, &compressed, &slot, &vstate);
  if (ierr) return ierr;
  HIPX_ARG(!compressed && m == n, "MatSOR needs a square, uncompressed matrix");
  if (!diag_dense) return fail(73 /* PETSC_ERR_ARG_WRONGSTATE */, "Matrix must have all diagonal locations to invert them (aij.c:1809)", __FILE__, __LINE__);
  if (flag & 128) return fail(HIPX_ERR_SUP, "SOR_APPLY_LOWER is not implemented (aij.c:1886: the reference raises the same error)", __FILE__, __LINE__);
  if (!m) return HIPX_SUCCESS;
  hipxSorState *S = (hipxSorState *)*slot;
  if (!S) {
    S     = new hipxSorState;
    *slot = S;
  }
  hipStream_t st = rt().compute;
  // mode: HIPX_SOR_MODE = strand | dep | levels; default: strands when the matrix has row templates, else the level-ordered
  // dependency-driven sweep (one launch per level when padding every level to a wave would blow up the slot map)
  int want = -1;
  {
    const char *e = getenv("HIPX_SOR_MODE");
    if (e && !strcmp(e, "levels")) want = 0;
    else if (e && !strcmp(e, "dep")) want = 1;
    else if (e && !strcmp(e, "strand")) want = 2;
    else if (e && !strcmp(e, "box")) want = 4;
  }
  // a matrix with inodes, relaxed with omega == 1 and no shift: MatSOR_SeqAIJ_Inode (aij.c:1852) -- the node-level sweeps
  bool use_inode = false;
  if (omega == 1.0 && shift == 0.0 && !getenv("HIPX_MAT_NO_INODE")) {
    int             istate;
    hipx_int        nnodes;
    const hipx_int *isz;
    if ((ierr = hipxMatInodes_(A, &istate, &nnodes, &isz))) return ierr;
    if (istate < 0) {  // not told: look, as MatAssemblyEnd_SeqAIJ does (inode.c:3920)
      if ((ierr = hipxMatEnsureInodes_(A))) return ierr;
      if ((ierr = hipxMatInodes_(A, &istate, &nnodes, &isz))) return ierr;
    }
    use_inode = nnodes > 0;
  }
  if (S->strand_vstate != vstate) {  // new values: the templates (which carry the values) are rebuilt
    if (S->strand) {
      HIPX_HIP(hipStreamSynchronize(st));
      strand_free((StrandState *)S->strand);
      S->strand = nullptr;
    }
    if (S->box) {
      HIPX_HIP(hipStreamSynchronize(st));
      hipxSorBoxFree_(S->box);
      S->box = nullptr;
    }
    S->box_tried     = false;
    S->strand_tried  = false;
    S->strand_vstate = vstate;
  }
  // Plane march (hipx_sorbox.hip, round 5): PCSOR's default application -- a zero-guess forward / backward / symmetric sweep, one iteration -- on a
  // constant-coefficient box stencil in natural ordering.  HIPX_SOR_MODE=box forces it (error when it does not apply), HIPX_SOR_BOX=0 turns it off
  {
    static const bool box_off = getenv("HIPX_SOR_BOX") && atoi(getenv("HIPX_SOR_BOX")) == 0;
    const bool box_sweep = (flag & 16) && !(flag & 32) && flag != 64 && (int64_t)its * (int64_t)lits == 1 && ((flag & 1) || (flag & 2) || (flag & 4) || (flag & 8));
    const bool aligned   = !((reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(x)) & 15);
    if ((want == 4 || (want == -1 && !box_off)) && box_sweep && aligned && !use_inode) {
      if (!S->box_tried) {
        S->box_tried = true;
        int                  tok = 0, ntmpl = 0;
        const int           *tstart, *toff, *tdiag;
        const double        *tval;
        const int64_t       *tcount;
        const unsigned char *d_tid;
        if ((ierr = hipxMatTemplates_(A, &tok, &ntmpl, &tstart, &toff, &tval, &tdiag, &tcount, &d_tid))) return ierr;
        if (tok && (ierr = hipxSorBoxBuild_((long long)m, ntmpl, tstart, toff, tval, tdiag, tcount, d_tid, &S->box))) return ierr;
      }
      if (S->box) {
        const bool plain = (omega == 1.0 && shift <= 0.0);
        double     dgv   = 0.0;
        {
          int                  tok = 0, ntmpl = 0;
          const int           *tstart, *toff, *tdiag;
          const double        *tval;
          const int64_t       *tcount;
          const unsigned char *d_tid;
          if ((ierr = hipxMatTemplates_(A, &tok, &ntmpl, &tstart, &toff, &tval, &tdiag, &tcount, &d_tid))) return ierr;
          dgv = tok ? tval[tstart[0] + tdiag[0]] : 1.0;  // (one diagonal value on these matrices: hipxSorBoxBuild_ checked)
        }
        if (plain && shift == 0.0 && dgv == 0.0) return fail(HIPX_ERR_ZEROPIVOT, "Zero diagonal on a row (aij.c:1820)", __FILE__, __LINE__);
        if (!S->d_t) {
          HIPX_HIP(hipMalloc((void **)&S->d_t, sizeof(double) * (size_t)m));
          HIPX_HIP(hipMalloc((void **)&S->d_w1, sizeof(double) * (size_t)m));
          HIPX_HIP(hipMalloc((void **)&S->d_idiag, sizeof(double) * (size_t)m));
          HIPX_HIP(hipMalloc((void **)&S->d_mdiag, sizeof(double) * (size_t)m));
          HIPX_HIP(hipMalloc((void **)&S->d_ctl, sizeof(unsigned int) * 2));
          HIPX_HIP(hipMemsetAsync(S->d_ctl, 0, sizeof(unsigned int) * 2, st));
          S->m = m;
        }
        const hipx_int gf  = std::min<hipx_int>((m + 255) / 256, 4096);
        const bool     fwd = (flag & 1) || (flag & 4), bwd = (flag & 2) || (flag & 8);
        if (fwd && bwd) {  // the forward result itself is not needed: the backward sweep re-forms x = t idiag from t (aij.c:1955)
          if ((ierr = hipxSorBoxFill_(S->box, S->d_w1, 0)) || (ierr = hipxSorBoxFill_(S->box, x, 1))) return ierr;
          if ((ierr = hipxSorBoxRun_(S->box, 0, b, S->d_t, S->d_w1, omega, shift, 0))) return ierr;
          if ((ierr = hipxSorBoxRun_(S->box, 1, S->d_t, nullptr, x, omega, shift, 1))) return ierr;
        } else {
          if ((ierr = hipxSorBoxFill_(S->box, x, fwd ? 0 : 1))) return ierr;
          if ((ierr = hipxSorBoxRun_(S->box, fwd ? 0 : 2, b, S->d_t, x, omega, shift, 1))) return ierr;
        }
        S->mode = S->last_mode = 4;
        unsigned int herr = 0;
        if ((ierr = hipxSorBoxError_(S->box, &herr, 0))) return ierr;  // (the word of the application before this one; this one's: next time)
        if (herr) return fail(HIPX_ERR_GPU, "MatSOR (plane march): a dependency was never published (wait limit reached)", __FILE__, __LINE__);
        return HIPX_SUCCESS;
      }
    }
    if (want == 4) return fail(HIPX_ERR_SUP, "HIPX_SOR_MODE=box: not a zero-guess sweep of a constant-coefficient box stencil in natural ordering (or unaligned vectors)", __FILE__, __LINE__);
  }
  if ((want == -1 || want == 2) && flag != 64 && !S->strand_tried && !use_inode) {
    S->strand_tried = true;
    int                  tok = 0, ntmpl = 0;
    const int           *tstart, *toff, *tdiag;
    const double        *tval;
    const int64_t       *tcount;
    const unsigned char *d_tid;
    if ((ierr = hipxMatTemplates_(A, &tok, &ntmpl, &tstart, &toff, &tval, &tdiag, &tcount, &d_tid))) return ierr;
    if (tok) {
      StrandState *T = new StrandState;
      S->strand      = T;
      if ((ierr = strand_build(T, m, ntmpl, tstart, toff, tval, tdiag, tcount, d_tid))) return ierr;
      if (!T->ok) {
        strand_free(T);
        S->strand = nullptr;
      }
    }
    // arbitrary values on a stencil pattern (variable-coefficient operators): the same schedule from the PATTERN templates, the
    // coefficients streamed per row (HIPX_SOR_VAR=0: off)
    static const bool var_on = !(getenv("HIPX_SOR_VAR") && atoi(getenv("HIPX_SOR_VAR")) == 0);
    if (!S->strand && var_on) {
      const int *d_pts = nullptr, *d_pto = nullptr;
      tok = 0;
      if ((ierr = hipxMatPatternTemplates_(A, &tok, &ntmpl, &tstart, &toff, &tdiag, &tcount, &d_tid, &d_pts, &d_pto))) return ierr;
      if (tok) {
        StrandState *T = new StrandState;
        S->strand      = T;
        if ((ierr = strand_build(T, m, ntmpl, tstart, toff, nullptr, tdiag, tcount, d_tid, true))) return ierr;
        if (!T->ok) {
          strand_free(T);
          S->strand = nullptr;
        } else {
          S->var_tstart = d_pts;
          S->var_toff   = d_pto;
        }
      }
    }
  }
  // the variable-coefficient kernels cover the sweeps without old-value lists (kinds 0-2): zero initial guess with one iteration
  // (what PCSOR applies by default) and Eisenstat; anything else on such a matrix takes the level-ordered schedule
  const bool var_fits   = !S->strand || !((StrandState *)S->strand)->var || (flag & 32) || ((flag & 16) && (int64_t)its * (int64_t)lits == 1);
  const bool use_strand = S->strand && var_fits && (want == -1 || want == 2) && flag != 64 && !use_inode;
  if (want == 2 && !use_strand && flag != 64) return fail(HIPX_ERR_SUP, "HIPX_SOR_MODE=strand: the matrix has no row templates / strand structure", __FILE__, __LINE__);
  if (!S->d_t) {  // work vectors shared by every mode
    HIPX_HIP(hipMalloc((void **)&S->d_t, sizeof(double) * (size_t)m));
    HIPX_HIP(hipMalloc((void **)&S->d_w1, sizeof(double) * (size_t)m));
    HIPX_HIP(hipMalloc((void **)&S->d_idiag, sizeof(double) * (size_t)m));
    HIPX_HIP(hipMalloc((void **)&S->d_mdiag, sizeof(double) * (size_t)m));
    HIPX_HIP(hipMalloc((void **)&S->d_ctl, sizeof(unsigned int) * 2));
    HIPX_HIP(hipMemsetAsync(S->d_ctl, 0, sizeof(unsigned int) * 2, st));
    S->m = m;
  }
  const hipx_int g = std::min<hipx_int>((m + 255) / 256, 4096);
  if (use_inode) {
    if (flag & (64 | 128)) return fail(HIPX_ERR_SUP, "SOR_APPLY_UPPER / SOR_APPLY_LOWER on a matrix with inodes: MatSOR_SeqAIJ_Inode has no such branch (inode.c:2494); use -mat_no_inode", __FILE__, __LINE__);
    InodeState *T = (InodeState *)S->inode;
    if (!T) {
      int             istate;
      hipx_int        nnodes;
      const hipx_int *isz;
      if ((ierr = hipxMatInodes_(A, &istate, &nnodes, &isz))) return ierr;
      T        = new InodeState;
      S->inode = T;
      HIPX_HIP(hipStreamSynchronize(st));
      if ((ierr = inode_build(T, m, nnz, is64, d_i, d_j, d_diagpos, nnodes, isz))) return ierr;
    }
    if (!T->ready) return fail(73, "inodes: the node schedule could not be built for this matrix", __FILE__, __LINE__);
    if (!T->values_valid) {
      unsigned int *cnt = rt().d_tickets + (HIPX_MAX_RED_SLOTS - 1);
      HIPX_HIP(hipMemsetAsync(cnt, 0, sizeof(unsigned int), st));
      inode_pack_kernel<<<(unsigned)std::min<hipx_int>((T->nnodes + 3) / 4, 8192), 256, 0, st>>>(T->nnodes, T->nsm, T->d_nmeta, T->d_nks, d_i, is64, d_j, d_a, T->d_nj, T->d_nv);
      HIPX_LAUNCH_CHECK();
      inode_invert_kernel<<<(unsigned)std::min<hipx_int>((T->nnodes + 255) / 256, 4096), 256, 0, st>>>(T->nnodes, T->nsm, T->d_nmeta, d_diagpos, d_a, T->d_ibd, T->d_bd, cnt);
      HIPX_LAUNCH_CHECK();
      HIPX_HIP(hipMemcpyAsync(&T->zero_pivots, cnt, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
      HIPX_HIP(hipStreamSynchronize(st));
      HIPX_HIP(hipMemsetAsync(cnt, 0, sizeof(unsigned int), st));
      // a singular diagonal block: the reference's block inverses (PetscKernel_A_gets_inverse_A_2..5, inode.c:2460-2490) report a zero pivot --
      // MAT_FACTOR_NUMERIC_ZEROPIVOT, or an error when erroriffailure is set.  Same code as the point sweep's zero diagonal: the caller decides
      // (plugin/mathipx.c sets A->factorerrortype / honours A->erroriffailure); x is not touched
      if (T->zero_pivots) return fail(HIPX_ERR_ZEROPIVOT, "Zero pivot in the diagonal block of a node (inode.c:2460-2490)", __FILE__, __LINE__);
      T->values_valid = true;
    }
    S->mode = S->last_mode = 3;
    double *W = S->d_w1;
    const size_t bytes = sizeof(double) * (size_t)m;
    // (`lits` is not used: MatSOR_SeqAIJ_Inode never multiplies its by it)
    if (flag & 32) {  // SOR_EISENSTAT (inode.c:3375-3806): x = (U + D)^-1 b;  t = b - D x;  t = (L + D)^-1 t;  x += t
      if (!S->d_w2) HIPX_HIP(hipMalloc((void **)&S->d_w2, sizeof(double) * (size_t)m));
      if ((ierr = run_inode<2>(S, b, nullptr, x, nullptr))) return ierr;
      inode_eisenstat_mid_kernel<<<(unsigned)std::min<hipx_int>((T->nnodes + 255) / 256, 4096), 256, 0, st>>>(T->nnodes, T->nsm, T->d_nmeta, T->d_bd, b, x, W);
      HIPX_LAUNCH_CHECK();
      if ((ierr = run_inode<5>(S, W, nullptr, S->d_w2, x))) return ierr;
    } else {
      const bool fwd = (flag & 1) || (flag & 4), bwd = (flag & 2) || (flag & 8);
      if (flag & 16) {  // SOR_ZERO_INITIAL_GUESS (inode.c:2526-2890)
        if (fwd && bwd) {
          if ((ierr = run_inode<0>(S, b, nullptr, W, nullptr))) return ierr;
          if ((ierr = run_inode<1>(S, S->d_t, W, x, nullptr))) return ierr;
        } else if (fwd) {
          if ((ierr = run_inode<0>(S, b, nullptr, x, nullptr))) return ierr;
        } else if (bwd) {
          if ((ierr = run_inode<2>(S, b, nullptr, x, nullptr))) return ierr;
        }
        its--;
      }
      while (its-- > 0) {  // inode.c:2891-3374
        if (fwd && bwd) {
          if ((ierr = run_inode<3>(S, b, x, W, nullptr))) return ierr;
          if ((ierr = run_inode<1>(S, S->d_t, W, x, nullptr))) return ierr;
        } else if (fwd) {
          if ((ierr = run_inode<3>(S, b, x, W, nullptr))) return ierr;
          HIPX_HIP(hipMemcpyAsync(x, W, bytes, hipMemcpyDeviceToDevice, st));
        } else if (bwd) {
          if ((ierr = run_inode<4>(S, b, x, W, nullptr))) return ierr;
          HIPX_HIP(hipMemcpyAsync(x, W, bytes, hipMemcpyDeviceToDevice, st));
        }
      }
    }
    unsigned int herr = 0;
    HIPX_HIP(hipMemcpyAsync(&herr, S->d_ctl + 1, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
    HIPX_HIP(hipStreamSynchronize(st));
    if (herr) {
      HIPX_HIP(hipMemsetAsync(S->d_ctl, 0, 2 * sizeof(unsigned int), st));
      return fail(HIPX_ERR_GPU, "MatSOR: a dependency was never published (wait limit reached)", __FILE__, __LINE__);
    }
    return HIPX_SUCCESS;
  }
  if (use_strand) {
    StrandState *T = (StrandState *)S->strand;
    const bool plain = (omega == 1.0 && shift <= 0.0);
    if (T->var) {
      if ((ierr = strand_set_cs(T, m, is64, d_i, d_a, d_diagpos, S->var_tstart, S->var_toff, omega, shift))) return ierr;
      if (T->zero_pivots && plain && shift == 0.0) return fail(HIPX_ERR_ZEROPIVOT, "Zero diagonal on a row (aij.c:1820)", __FILE__, __LINE__);
    } else {
      if (plain && shift == 0.0)
        for (double d : T->diagval)
          if (d == 0.0) return fail(HIPX_ERR_ZEROPIVOT, "Zero diagonal on a row (aij.c:1820)", __FILE__, __LINE__);
      if ((ierr = strand_set_diag(T, omega, shift))) return ierr;
    }
    S->mode = 2;
  } else {
    if (!S->ready) {
      HIPX_HIP(hipStreamSynchronize(st));
      if ((ierr = build_schedule(S, m, nnz, is64, d_i, d_j))) return ierr;
    }
    if (!S->values_valid) {
      permute_rows_kernel<<<(unsigned)g, 256, 0, st>>>(m, S->d_perm, S->d_pi, d_i, is64, d_j, d_a, d_diagpos, S->d_pj, S->d_pa, S->d_pd);
      HIPX_LAUNCH_CHECK();
      S->values_valid = true;
      S->idiag_valid  = false;
      if (!S->d_smeta) {
        HIPX_HIP(hipMalloc((void **)&S->d_smeta, sizeof(int4) * (size_t)std::max<hipx_int>(S->nslots, 1)));
        HIPX_HIP(hipMalloc((void **)&S->d_sks, sizeof(int64_t) * (size_t)std::max<hipx_int>(S->nslots, 1)));
      }
      if (S->nslots) {
        slot_meta_kernel<<<(unsigned)std::min<hipx_int>((S->nslots + 255) / 256, 4096), 256, 0, st>>>(S->nslots, S->d_slot, S->d_perm, S->d_pi, S->d_pd, S->d_smeta, S->d_sks);
        HIPX_LAUNCH_CHECK();
      }
      if (!S->d_smeta4) {
        HIPX_HIP(hipMalloc((void **)&S->d_smeta4, sizeof(int4) * (size_t)std::max<hipx_int>(S->nslots4, 1)));
        HIPX_HIP(hipMalloc((void **)&S->d_sks4, sizeof(int64_t) * (size_t)std::max<hipx_int>(S->nslots4, 1)));
      }
      if (S->nslots4) {
        slot_meta_kernel<<<(unsigned)std::min<hipx_int>((S->nslots4 + 255) / 256, 4096), 256, 0, st>>>(S->nslots4, S->d_slot4, S->d_perm, S->d_pi, S->d_pd, S->d_smeta4, S->d_sks4);
        HIPX_LAUNCH_CHECK();
      }
    }
    if (!S->idiag_valid || S->omega != omega || S->shift != shift) {  // aij.c:1807
      unsigned int *cnt = rt().d_tickets + (HIPX_MAX_RED_SLOTS - 1);
      HIPX_HIP(hipMemsetAsync(cnt, 0, sizeof(unsigned int), st));
      const int plain = (omega == 1.0 && shift <= 0.0);
      invert_diag_kernel<<<(unsigned)g, 256, 0, st>>>(m, d_diagpos, d_a, omega, shift, plain, S->d_idiag, S->d_mdiag, cnt);
      HIPX_LAUNCH_CHECK();
      HIPX_HIP(hipMemcpyAsync(&S->zero_pivots, cnt, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
      HIPX_HIP(hipStreamSynchronize(st));
      HIPX_HIP(hipMemsetAsync(cnt, 0, sizeof(unsigned int), st));
      S->omega       = omega;
      S->shift       = shift;
      S->idiag_valid = true;
      if (S->zero_pivots && plain && shift == 0.0) return fail(HIPX_ERR_ZEROPIVOT, "Zero diagonal on a row (aij.c:1820)", __FILE__, __LINE__);
    }
    // a level-per-wave slot map pads every level to 64 slots: with long dependency chains (nlevels ~ m: banded / 1-D
    // orderings) that is 64 slots per row -- run one launch per level there instead
    S->mode = (want == 0 || (int64_t)S->nslots > 4 * (int64_t)m + 65536) ? 0 : 1;
  }
  S->last_mode = S->mode;
  its = its * lits;  // aij.c:1855
  if (flag == 64) {  // SOR_APPLY_UPPER
    sor_apply_upper_kernel<<<(unsigned)g, 256, 0, st>>>(m, S->d_perm, S->d_pi, S->d_pd, S->d_pj, S->d_pa, S->d_mdiag, b, x, omega, shift);
    HIPX_LAUNCH_CHECK();
    return HIPX_SUCCESS;
  }
  if (flag & 32) {
    // SOR_EISENSTAT (aij.c:1887-1929): x = (E + U)^-1 b  [a backward zero-guess sweep];  t = b - (2/omega - 1) D x;
    // t = (E + L)^-1 t  [a forward zero-guess sweep with t as right-hand side];  x = x + t
    if (!S->d_w2) HIPX_HIP(hipMalloc((void **)&S->d_w2, sizeof(double) * (size_t)m));
    if (S->mode == 2 && !S->mdiag_valid) {  // the diagonal itself (strand mode keeps 1/d per template only)
      unsigned int *cnt = rt().d_tickets + (HIPX_MAX_RED_SLOTS - 1);
      invert_diag_kernel<<<(unsigned)g, 256, 0, st>>>(m, d_diagpos, d_a, omega, shift, 0, S->d_idiag, S->d_mdiag, cnt);
      HIPX_LAUNCH_CHECK();
      S->mdiag_valid = true;
      S->idiag_valid = false;
    }
    const double scale = (2.0 / omega) - 1.0;
    if (S->mode >= 1) {
      if ((ierr = run_sweep<2>(S, b, nullptr, x, omega))) return ierr;
      sor_eisenstat_mid_kernel<<<(unsigned)g, 256, 0, st>>>(m, S->d_mdiag, b, x, scale, S->d_w1);
      HIPX_LAUNCH_CHECK();
      if ((ierr = run_sweep<0>(S, S->d_w1, nullptr, S->d_w2, omega))) return ierr;  // (writes the unscaled sums to d_t: unused here)
      if ((ierr = hipxVecAXPY(x, 1.0, S->d_w2, m))) return ierr;
      unsigned int *ctl  = S->mode == 2 ? ((StrandState *)S->strand)->d_ctl : S->d_ctl;
      unsigned int  herr = 0;
      HIPX_HIP(hipMemcpyAsync(&herr, ctl + 1, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
      HIPX_HIP(hipStreamSynchronize(st));
      if (herr) {
        HIPX_HIP(hipMemsetAsync(ctl, 0, 2 * sizeof(unsigned int), st));
        return fail(HIPX_ERR_GPU, "MatSOR: a dependency was never published (wait limit reached)", __FILE__, __LINE__);
      }
    } else {
      if ((ierr = run_levels<2>(S, false, b, x, omega))) return ierr;
      sor_eisenstat_mid_kernel<<<(unsigned)g, 256, 0, st>>>(m, S->d_mdiag, b, x, scale, S->d_w1);
      HIPX_LAUNCH_CHECK();
      if ((ierr = run_levels<0>(S, true, S->d_w1, S->d_w2, omega))) return ierr;
      if ((ierr = hipxVecAXPY(x, 1.0, S->d_w2, m))) return ierr;
    }
    return HIPX_SUCCESS;
  }
  const bool fwd = (flag & 1) || (flag & 4), bwd = (flag & 2) || (flag & 8);
  if (S->mode >= 1) {
    // dependency-driven sweeps: each sweep reads OLD, writes NEW (sentinel-filled); the result ends in the user's x
    const size_t bytes = sizeof(double) * (size_t)m;
    double      *W     = S->d_w1;
    if (flag & 16) {  // SOR_ZERO_INITIAL_GUESS, aij.c:1930-1960
      if (fwd && bwd) {
        if ((ierr = run_sweep<0>(S, b, nullptr, W, omega))) return ierr;
        if ((ierr = run_sweep<1>(S, b, W, x, omega))) return ierr;
      } else if (fwd) {
        if ((ierr = run_sweep<0>(S, b, nullptr, x, omega))) return ierr;
      } else if (bwd) {
        if ((ierr = run_sweep<2>(S, b, nullptr, x, omega))) return ierr;
      }
      its--;
    }
    while (its--) {  // aij.c:1961-2002
      if (fwd && bwd) {
        if ((ierr = run_sweep<3>(S, b, x, W, omega))) return ierr;
        if ((ierr = run_sweep<1>(S, b, W, x, omega))) return ierr;
      } else if (fwd) {
        if ((ierr = run_sweep<3>(S, b, x, W, omega))) return ierr;
        HIPX_HIP(hipMemcpyAsync(x, W, bytes, hipMemcpyDeviceToDevice, st));
      } else if (bwd) {
        if ((ierr = run_sweep<4>(S, b, x, W, omega))) return ierr;
        HIPX_HIP(hipMemcpyAsync(x, W, bytes, hipMemcpyDeviceToDevice, st));
      }
    }
    unsigned int *ctl  = S->mode == 2 ? ((StrandState *)S->strand)->d_ctl : S->d_ctl;
    unsigned int  herr = 0;
    HIPX_HIP(hipMemcpyAsync(&herr, ctl + 1, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
    HIPX_HIP(hipStreamSynchronize(st));
    if (herr) {
      HIPX_HIP(hipMemsetAsync(ctl, 0, 2 * sizeof(unsigned int), st));
      return fail(HIPX_ERR_GPU, "MatSOR: a dependency was never published (wait limit reached)", __FILE__, __LINE__);
    }
    return HIPX_SUCCESS;
  }
  if (flag & 16) {  // SOR_ZERO_INITIAL_GUESS, aij.c:1930-1960
    if (fwd && (ierr = run_levels<0>(S, true, b, x, omega))) return ierr;
    if (bwd) {
      if (fwd) ierr = run_levels<1>(S, false, b, x, omega);
      else ierr = run_levels<2>(S, false, b, x, omega);
      if (ierr) return ierr;
    }
    its--;
  }
  while (its--) {  // aij.c:1961-2002
    if (fwd && (ierr = run_levels<3>(S, true, b, x, omega))) return ierr;
    if (bwd) {
      if (fwd) ierr = run_levels<1>(S, false, b, x, omega);
      else ierr = run_levels<4>(S, false, b, x, omega);
      if (ierr) return ierr;
    }
  }
  return HIPX_SUCCESS;
}
