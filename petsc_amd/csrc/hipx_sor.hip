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
  double   *d_w1 = nullptr;
  unsigned int *d_ctl = nullptr;  // [0] block ticket, [1] error flag
  int       mode = 1;           // 1 = dependency-driven single launch per sweep, 0 = one launch per level
  double    omega = 0.0, shift = 0.0;
  bool      idiag_valid = false, values_valid = false;
  unsigned int zero_pivots = 0;
};

extern "C" {
// accessors implemented in hipx_mat.hip
int   hipxMatInternal_(hipxMat A, hipx_int *m, hipx_int *n, int64_t *nnz, int *is64, void **d_i, hipx_int **d_j, double **d_a, int64_t **d_diagpos, int *diag_dense,
                       int *compressed, void ***sor_slot, unsigned long long *value_state);
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
constexpr int                SOR_SPIN_MAX = 1 << 20;  // ~0.5 s of polling per lane before the launch is declared stuck

__device__ __forceinline__ double sor_poll(const double *p, unsigned int *err)
{
  const unsigned long long *q = reinterpret_cast<const unsigned long long *>(p);
  unsigned long long        v = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int                       spins = 0;
  while (v == SOR_SENTINEL) {
    __builtin_amdgcn_s_sleep(8);
    v = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ++spins;
    if ((spins & 0xff) == 0) {  // global abort: once any lane has given up, nobody waits any more (bounds the whole launch)
      if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || spins > SOR_SPIN_MAX) {
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
  constexpr int CH = 8;  // entries whose column/value loads are all in flight before the first poll
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
    if (KIND == 0) {
      sum  = sor_minusdot<true>(b[i], i, ks, kd, pj, pa, xold, xnew, err);
      t[i] = sum;
      out  = sum * idiag[i];
    } else if (KIND == 3) {
      sum  = sor_minusdot<true>(b[i], i, ks, kd, pj, pa, xold, xnew, err);
      t[i] = sum;
      sum  = sor_minusdot<true>(sum, i, kd + 1, ke, pj, pa, xold, xnew, err);  // upper part: old values
      out  = (1. - omega) * xold[i] + sum * idiag[i];
    } else if (KIND == 1) {
      sum = sor_minusdot<false>(t[i], i, kd + 1, ke, pj, pa, xold, xnew, err);
      out = (1 - omega) * xold[i] + sum * idiag[i];
    } else if (KIND == 2) {
      sum = sor_minusdot<false>(b[i], i, kd + 1, ke, pj, pa, xold, xnew, err);
      out = sum * idiag[i];
    } else {
      sum = sor_minusdot<false>(b[i], i, ks, ke, pj, pa, xold, xnew, err);  // whole row: lower + diagonal old, upper new
      out = (1. - omega) * xold[i] + (sum + mdiag[i] * xold[i]) * idiag[i];
    }
    sor_publish(xnew + i, out);
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
  HIPX_HIP(hipMalloc((void **)&S->d_idiag, sizeof(double) * (size_t)m));
  HIPX_HIP(hipMalloc((void **)&S->d_mdiag, sizeof(double) * (size_t)m));
  HIPX_HIP(hipMalloc((void **)&S->d_t, sizeof(double) * (size_t)m));
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
    HIPX_HIP(hipMalloc((void **)&S->d_w1, sizeof(double) * (size_t)m));
    HIPX_HIP(hipMalloc((void **)&S->d_ctl, sizeof(unsigned int) * 2));
    HIPX_HIP(hipMemset(S->d_ctl, 0, sizeof(unsigned int) * 2));
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
  if (S) S->values_valid = S->idiag_valid = false;
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
  (void)hipFree(S->d_w1);
  (void)hipFree(S->d_ctl);
  delete S;
}

extern "C" int hipxMatSOR(hipxMat A, const double *b, double omega, int flag, double shift, hipx_int its, hipx_int lits, double *x)
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
  if (flag & 128) return fail(HIPX_ERR_SUP, "SOR_APPLY_LOWER is not implemented (aij.c:1886)", __FILE__, __LINE__);
  if (flag & 32) return fail(HIPX_ERR_SUP, "SOR_EISENSTAT is not provided by MATSEQAIJHIPX", __FILE__, __LINE__);
  if (!m) return HIPX_SUCCESS;
  hipxSorState *S = (hipxSorState *)*slot;
  if (!S) {
    S     = new hipxSorState;
    *slot = S;
  }
  hipStream_t st = rt().compute;
  if (!S->ready) {
    HIPX_HIP(hipStreamSynchronize(st));
    if ((ierr = build_schedule(S, m, nnz, is64, d_i, d_j))) return ierr;
  }
  static unsigned long long last_state_dummy = 0;
  (void)last_state_dummy;
  const hipx_int g = std::min<hipx_int>((m + 255) / 256, 4096);
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
  its = its * lits;  // aij.c:1855
  if (flag == 64) {  // SOR_APPLY_UPPER
    sor_apply_upper_kernel<<<(unsigned)g, 256, 0, st>>>(m, S->d_perm, S->d_pi, S->d_pd, S->d_pj, S->d_pa, S->d_mdiag, b, x, omega, shift);
    HIPX_LAUNCH_CHECK();
    return HIPX_SUCCESS;
  }
  const bool fwd = (flag & 1) || (flag & 4), bwd = (flag & 2) || (flag & 8);
  {
    const char *e = getenv("HIPX_SOR_MODE");  // "levels": one launch per level (debugging / comparison)
    S->mode = (e && !strcmp(e, "levels")) ? 0 : 1;
  }
  if (S->mode == 1) {
    // dependency-driven sweeps: each sweep reads OLD, writes NEW (sentinel-filled); the result ends in the user's x
    const size_t bytes = sizeof(double) * (size_t)m;
    double      *W     = S->d_w1;
    if (flag & 16) {  // SOR_ZERO_INITIAL_GUESS, aij.c:1930-1960
      if (fwd && bwd) {
        if ((ierr = run_dep<0>(S, b, nullptr, W, omega))) return ierr;
        if ((ierr = run_dep<1>(S, b, W, x, omega))) return ierr;
      } else if (fwd) {
        if ((ierr = run_dep<0>(S, b, nullptr, x, omega))) return ierr;
      } else if (bwd) {
        if ((ierr = run_dep<2>(S, b, nullptr, x, omega))) return ierr;
      }
      its--;
    }
    while (its--) {  // aij.c:1961-2002
      if (fwd && bwd) {
        if ((ierr = run_dep<3>(S, b, x, W, omega))) return ierr;
        if ((ierr = run_dep<1>(S, b, W, x, omega))) return ierr;
      } else if (fwd) {
        if ((ierr = run_dep<3>(S, b, x, W, omega))) return ierr;
        HIPX_HIP(hipMemcpyAsync(x, W, bytes, hipMemcpyDeviceToDevice, st));
      } else if (bwd) {
        if ((ierr = run_dep<4>(S, b, x, W, omega))) return ierr;
        HIPX_HIP(hipMemcpyAsync(x, W, bytes, hipMemcpyDeviceToDevice, st));
      }
    }
    unsigned int herr = 0;
    HIPX_HIP(hipMemcpyAsync(&herr, S->d_ctl + 1, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
    HIPX_HIP(hipStreamSynchronize(st));
    if (herr) {
      HIPX_HIP(hipMemsetAsync(S->d_ctl, 0, 2 * sizeof(unsigned int), st));
      return fail(HIPX_ERR_GPU, "MatSOR: a dependency was never published (spin limit reached)", __FILE__, __LINE__);
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
