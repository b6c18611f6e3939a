// hipx_mat.hip -- CSR (Mat_SeqAIJ) kernels for gfx950 behind include/hipx.h.
//
// SpMV ("row-block stream" kernel, the MatMult_SeqAIJ replacement, aij.c:1444-1502)
//   HBM-bound: 12 B per nonzero (8 B value + 4 B column) + 4 B per row offset + x + y.
//   * the host cuts the rows into row blocks: consecutive rows whose nonzeros fit an LDS tile of CAP
//     products (and at most 256 rows, one row per thread in the reduce phase);
//   * phase 1 (stream): the 256 threads of a workgroup read the block's val[] and col[] ranges as one
//     contiguous, 32/16-byte-per-lane coalesced stream (4 nonzeros per lane per step, all loads issued
//     before first use), gather x[col] through L1/L2 and park the rounded products in LDS;
//   * phase 2 (row sums): thread t adds row t's products from LDS left to right, starting from 0 (or
//     from y_t for MatMultAdd): the same association as PetscSparseDensePlusDot (aij.h:608-614), one
//     rounded multiply + one rounded add per entry, no FMA -> y is bit-identical to MatMult_SeqAIJ;
//   * XCD-aware block order: workgroup b runs on XCD b % 8 (observed dispatch rule, speed only); the
//     row blocks are remapped so that each XCD walks one contiguous slab of rows, which keeps the
//     x-window a slab touches (rows +-n^2 for the 7-point stencil) resident in that XCD's 4 MiB L2
//     instead of being fetched by all eight;
//   * val/col are read once -> optional non-temporal loads keep them from evicting x out of L2.
//   Rows longer than the LDS tile take a block-wide strided path (tree sum, not left-to-right).
#include "hipx_internal.h"
#include <type_traits>
#include "hipx_reduce.h"
#include "hipx_ipc.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace hipx;

// Pair form of the template SpMV (spmv_pair_kernel): the base template's entries grouped by the EVEN column offsets e_j whose
// 16-byte pairs (x[r + e], x[r + e + 1], r = the thread's even row) are loaded; slot 1 of a pair is the entry at offset e itself, slot
// 0 the entry at e - 1 (row r takes the previous lane's second element, row r + 1 the pair's first), slot 2 the entry at e + 1 (row r
// takes the second element, row r + 1 the next lane's first).  kb = the entry's index in the base template (its bit in the row
// masks) or -1; pairs ascend in e, so slots 0, 1, 2 of pair 0, 1, ... is the entries' own (ascending-column) order.
// March form of the template SpMV (spmv_march_kernel): the base template's offsets o_k = a_k S + b_k with a_k in {-1, 0, +1} (the 'plane' below,
// the row's own plane, the plane above: S = the stride between them, e.g. n^2 for an n^3 grid in natural ordering) and |b_k| <= H.  A workgroup
// owns the rows i0 ... i0 + L - 1 of every plane of its segment and marches through the planes keeping three of them (with H halo elements
// either side) in LDS: every element of x is fetched ONCE per workgroup (plus halo) instead of once per entry.
struct hipxMarchPlan {
  int      ne, nlo, nmid;  // base template: entries [0, nlo) on the plane below, [nlo, nlo + nmid) on the own plane, the rest above (ascending columns)
  int      S, H, L;        // plane stride (even), halo (even), rows per plane and workgroup
  unsigned full;           // mask of all ne entries
  int      b[32];
  double   a[32];
};
struct hipxPairPlan {
  int    npairs, jdiag;
  int    jodd, eodd;  // jodd >= 0: some pair has entries at e - 1 / e + 1 (the kernel then loads the waves' edge elements); eodd: unused
  int    e[16];
  int    kb[16][3];
  double a[16][3];
};

// Chebyshev epilogue of the pair-form SpMV (hipxMatMultChebyshev): instead of y = A x the kernel stores
//   pnext = alpha pprev + beta x + gamma (dinv .* (b - A x))      (x = the current iterate: the diagonal pair of the walk)
// element by element the operations of hipxVecChebyshevStep (cheby.c:475-511).  br: VecAXPBYPCZ_Seq's association order (bvec1.c:120-147).
struct hipxPairEpi {
  const double *b, *dinv, *pprev;
  double        alpha, beta, gamma;
  int           br;
};

struct hipxMat_s {
  hipx_int  m = 0, n = 0;
  int64_t   nnz       = 0;
  bool      is64      = false;  // 64-bit row offsets
  void     *d_i       = nullptr;
  hipx_int *d_j       = nullptr;
  double   *d_a       = nullptr;
  int64_t  *d_diagpos = nullptr;  // position of a_ii in a[], -1 if absent
  // row blocks, one set per kernel geometry (built lazily from the host copy of the row offsets)
  static constexpr int kMaxCfg = 8;
  hipx_int *d_rb[kMaxCfg]    = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipx_int  nblocks[kMaxCfg] = {0, 0, 0, 0, 0, 0, 0, 0};
  bool      rb_ready[kMaxCfg] = {false, false, false, false, false, false, false, false};
  hipx_int *d_sched[kMaxCfg] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // launch slot -> row block (null = identity)
  int64_t   far_offset = 0;   // typical max |col - row| of a row (0 = unknown / irregular): drives the block schedule
  int       sched_mode = 0;   // 1 = band-aware schedule (cuts x re-fetch from the Infinity Cache; no time gain measured), 0 = natural order
  // packed-column formats (built lazily on the host, once per nonzero pattern)
  int            tile_mode  = 0;   // 2 = packed 16-bit columns, products staged in LDS (variant 22); 3 = + row-parallel gather (variant 23)
  bool           pk_ready   = false;
  int            pk_cfg     = 0;         // row-block geometry (kCfg index) the packed format was built on
  void          *d_pkdesc   = nullptr;   // PkDesc per row block: (r0, r1, k0, k1) in one load
  unsigned short *d_pk      = nullptr;   // (window id << 12) | offset inside the window
  hipx_int      *d_pkbase   = nullptr;   // PK_WMAX window starts per row block (-1 in slot 0 = block keeps 32-bit columns)
  int64_t        pk_fallback_blocks = 0;
  // value dictionary (<= 256 distinct bit patterns in a[]): 1-byte code per nonzero, rebuilt when the values change
  bool           auto_sel = true;     // variant 0: the library picks the kernel form
  int            vd_mode  = 0;        // 1 = use the dictionary kernels when the dictionary exists
  bool           vd_ready = false;    // dictionary attempted for the current values
  bool           vd_ok    = false;    // ... and it fits
  int            vd_count = 0;
  unsigned char *d_vc     = nullptr;
  double        *d_vdict  = nullptr;  // 256 entries
  // row templates (stencil-structured matrices): every row is one of <= 256 (offset, value) sequences; the matrix then is
  // ONE byte per row.  Built lazily on the device, verified entry by entry (hash collisions cannot slip through).
  int            tmpl_mode  = 0;      // 1 = use the template kernel when the templates exist
  bool           tmpl_ready = false;  // attempted for the current pattern + values
  bool           tmpl_ok    = false;
  int            ntmpl = 0, tmpl_nent = 0, tmpl_maxlen = 0;
  unsigned char *d_tid    = nullptr;  // template id per row
  int           *d_tstart = nullptr;  // ntmpl + 1 offsets into toff / tval
  // sub-template form: every template is the most common one (tmpl_base) with some entries left out (boundary rows of a stencil:
  // same offsets, same values, fewer neighbours) -> bit k of d_tmask[t] says whether entry k of the base template is in template t
  unsigned int  *d_tmask  = nullptr;
  int            tmpl_base = -1;
  bool           pair_ok = false;  // ... and the base template fits the pair form (<= 16 even-offset pairs)
  hipxPairPlan   pair_plan;
  bool           march_ok = false;  // ... and its offsets split into three 'planes' S rows apart (march form, spmv_march_kernel)
  hipxMarchPlan  march_plan;
  bool           march_force = false;  // hipxMatSetSpMVVariant(30)
  int            march_nt = 256;       // 512: the plan needs the 512-thread form of spmv_march2_kernel (4096-row tiles, halos beyond the 80 KiB of the two-workgroup forms: lines of up to 1024 points)
  int            march2_state = 0;     // spmv_march2_kernel's extra conditions (whole planes and tiles, plane-periodic template ids, run structure): 0 not checked yet, 1 hold, -1 do not
  hipx_int       dot_npart_used = 0;  // dot partials the last fused template launch wrote when it was not the count dot_partials_count() gave (march form)
  int           *d_toff   = nullptr;  // column - row
  double        *d_tval   = nullptr;
  double             *d_march_dump = nullptr;  // march form, long templates: where the dot partials go when no dot was asked for
  unsigned long long *d_tq = nullptr;   // chunk queue of the template kernel: one ticket counter per XCD (64 bytes apart), never reset
  unsigned long long  tq_launches = 0;  // launches so far on this geometry (the kernel subtracts launches * tickets-per-launch)
  int                 tq_geom = -1;     // geometry (grid, rows per chunk) the counters have been used with
  long long           tmpl_maxoff = -1; // farthest forward offset of the most common template (-1: not computed yet)
  std::vector<int>     h_tstart, h_toff, h_tdiag;  // host copies (SOR set-up reads them); h_tdiag = index of the diagonal entry or -1
  std::vector<int64_t> h_tcount;                   // rows per template
  std::vector<double>  h_tval;
  int       probe      = 0;   // phase-attribution probe kernels (scripts/spmv_variants.py); results are NOT A x
  std::vector<int64_t> h_i;  // host copy of the row offsets (set-up only)
  void     *sor_state = nullptr;  // hipxSorState, owned by hipx_sor.hip
  int                   inode_state = -1;  // -1: not determined (hipx_sor.hip looks at the first hipxMatSOR call), 0: none, 1: inode_sizes holds the nodes
  std::vector<hipx_int> inode_sizes;       // node_count + 1 row offsets (Mat_SeqAIJ_Inode::size_csr)
  // pattern templates (variant 29): the rows' (column - row) lists only -- <= 256 distinct ones; the values stay in a[] and are streamed
  bool           ptm_ready = false, ptm_ok = false, ptm_build = false;  // ptm_build: build_templates() is running in pattern-only mode
  int            ptm_mode = 0, ptm_ntmpl = 0, ptm_nent = 0, ptm_maxlen = 0;
  unsigned char *d_ptid    = nullptr;
  int           *d_ptstart = nullptr, *d_ptoff = nullptr;
  std::vector<int>     h_ptstart, h_ptoff, h_ptdiag;  // host copies (SOR set-up for variable-coefficient stencils reads them)
  std::vector<int64_t> h_ptcount;
  void     *sell_state = nullptr; // SellState, owned by hipx_sell.hip (variant 28: sliced-ELLPACK copy)
  int       sell_mode  = 0;
  unsigned long long value_state = 1;
  // compressed rows (off-diagonal block of MPIAIJ): logical rows = nrows_c, y index = ridx[row]
  bool      compressed = false;
  hipx_int  nrows_c    = 0;
  hipx_int *d_ridx     = nullptr;
  int       variant    = 0;
  int64_t   device_bytes = 0;
  bool      diag_dense = true;
  // fused SpMV+dot partials
  double   *d_dotpart = nullptr;
  hipx_int  dotpart_cap = 0;
};

extern "C" void hipxSorStateFree_(void *p);
extern "C" void     hipxSellFree_(void *p);
extern "C" void     hipxSellValuesChanged_(void *p);
extern "C" void     hipxSellInvalidate_(void *p);
extern "C" int      hipxSellEnsure_(hipxMat A, void **slot, int *ok, int *packed, double *pad_ratio, int64_t *bytes);
extern "C" hipx_int hipxSellDotPartials_(void *p);
extern "C" int      hipxSellLaunch_(void *p, int mode, int dot, const double *x, const double *yin, double *yout, double *dotpart, int pair);
extern "C" int      hipxMatEnsureInodes_(hipxMat A);  // hipx_sor.hip: looks for inodes if nobody has said (MatSeqAIJCheckInode)
extern "C" void hipxSorInvalidate_(void *p);
static void     mpicg_forget(hipxMat M);  // (plans of hipxMatMPICGPlan_ that name this matrix)

namespace {

// optional event bracketing of every SpMV launch (bench.py roofline.achieved)
struct SpmvProf {
  bool                    on = false;
  std::vector<hipEvent_t> ev;  // pairs
  size_t                  used = 0;
};
SpmvProf &prof()
{
  static SpmvProf p;
  return p;
}
inline int prof_mark(bool start)
{
  SpmvProf &p = prof();
  if (!p.on) return HIPX_SUCCESS;
  if (start && p.used + 2 > p.ev.size()) {
    for (int k = 0; k < 2; k++) {
      hipEvent_t e;
      HIPX_HIP(hipEventCreate(&e));
      p.ev.push_back(e);
    }
  }
  HIPX_HIP(hipEventRecord(p.ev[p.used++], rt().compute));
  return HIPX_SUCCESS;
}

// kernel geometries: THREADS per workgroup, CAP = products staged in LDS per row block, RPT = rows per thread in the
// row-sum phase (a block holds at most THREADS * RPT rows)
struct SpmvCfg {
  int threads, cap, rpt;
};
constexpr SpmvCfg kCfg[] = {
  {256, 2048, 1},  // 0: 16 KiB LDS, 8 workgroups / CU
  {256, 4096, 2},  // 1: 32 KiB LDS
  {512, 4096, 1},  // 2
  {128, 1024, 1},  // 3
  {256, 3072, 2},  // 4: 24 KiB
  {512, 8192, 2},  // 5: 64 KiB
  {256, 8192, 4},  // 6: row blocks of the value-dictionary kernel with 4 rows per thread (3 B of LDS per nonzero there)
};
constexpr int kNumCfg = sizeof(kCfg) / sizeof(kCfg[0]);

typedef double dbl2 __attribute__((ext_vector_type(2)));
typedef int    int4v __attribute__((ext_vector_type(4)));
typedef unsigned short ushort4v __attribute__((ext_vector_type(4)));
typedef unsigned char  uchar4v __attribute__((ext_vector_type(4)));

template <bool NT, typename T>
__device__ __forceinline__ T stream_load(const T *p)
{
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}

// MODE: 0 y = A x ; 1 z = y + A x
// DBG (probe builds only, never selected by the product path): 1 = skip the x gather, 2 = skip gather and row sums,
// 3 = gather but skip the row sums.  Used by scripts/spmv_variants.py to attribute time to the kernel's phases.
template <typename IT, int SPMV_THREADS, int SPMV_CAP, int RPT, bool NT, int MODE, bool CPROW, bool DOT, int DBG = 0>
__global__ __launch_bounds__(SPMV_THREADS) void spmv_stream_kernel(const hipx_int *__restrict__ rb, const hipx_int *__restrict__ sched, hipx_int nblocks, hipx_int blocks_per_xcd, const IT *__restrict__ ai,
                                                                    const hipx_int *__restrict__ aj, const double *__restrict__ aa, const double *__restrict__ x,
                                                                    const double *yin, double *yout, const hipx_int *__restrict__ ridx, double *dotpart)
{
  __shared__ double prod[SPMV_CAP];
  // XCD-aware remap: hardware block id -> (xcd, slot) -> contiguous slab per XCD
  const hipx_int bid = (hipx_int)blockIdx.x;
  const hipx_int slot = (bid & 7) * blocks_per_xcd + (bid >> 3);
  const hipx_int b    = (sched && slot < nblocks) ? sched[slot] : slot;
  double         mydot = 0.0;
  if (b < nblocks) {
    const hipx_int r0 = rb[b], r1 = rb[b + 1];
    const IT       k0 = ai[r0], k1 = ai[r1];
    const IT       ka = k0 & ~(IT)3;  // 32-byte aligned start of the val stream (arrays are padded by 4)
    const int      t  = threadIdx.x;
    // this thread's row extents for phase 2 (rows r0 + t + j*THREADS): issue the loads now, consume after the barrier
    IT rs[RPT], re[RPT];
#pragma unroll
    for (int j = 0; j < RPT; j++) {
      const hipx_int row = r0 + t + j * SPMV_THREADS;
      rs[j] = re[j] = 0;
      if (row < r1) {
        rs[j] = ai[row];
        re[j] = ai[row + 1];
      }
    }
    if ((k1 - ka) <= (IT)SPMV_CAP) {
      const IT nq = (k1 - ka + 3) >> 2;  // quads to stream (>= 1 unless every row of the block is empty)
      if (nq > 0) {
        const dbl2   *a2  = reinterpret_cast<const dbl2 *>(aa + ka);
        const int4v  *j4  = reinterpret_cast<const int4v *>(aj + ka);
        constexpr int NIT = SPMV_CAP / 4 / SPMV_THREADS;
        dbl2          va[NIT], vb[NIT];
        int4v         vc[NIT];
        double        xv[NIT][4];
        // branch-free issue: out-of-range lanes re-read the last quad (clamped), so every load of the
        // block is in flight before the first use
#pragma unroll
        for (int it = 0; it < NIT; it++) {
          const IT q  = (IT)t + (IT)it * SPMV_THREADS;
          const IT qc = q < nq ? q : nq - 1;
          va[it]      = stream_load<NT>(a2 + 2 * qc);
          vb[it]      = stream_load<NT>(a2 + 2 * qc + 1);
          vc[it]      = stream_load<NT>(j4 + qc);
        }
#pragma unroll
        for (int it = 0; it < NIT; it++) {
          if (DBG == 1 || DBG == 2) {
            xv[it][0] = xv[it][1] = xv[it][2] = xv[it][3] = (double)(vc[it].x + vc[it].y + vc[it].z + vc[it].w);
          } else {
            xv[it][0] = x[vc[it].x];
            xv[it][1] = x[vc[it].y];
            xv[it][2] = x[vc[it].z];
            xv[it][3] = x[vc[it].w];
          }
        }
#pragma unroll
        for (int it = 0; it < NIT; it++) {
          const IT q = (IT)t + (IT)it * SPMV_THREADS;
          if (q < nq) {
            dbl2 p0, p1;
            p0.x = va[it].x * xv[it][0];
            p0.y = va[it].y * xv[it][1];
            p1.x = vb[it].x * xv[it][2];
            p1.y = vb[it].y * xv[it][3];
            reinterpret_cast<dbl2 *>(prod)[2 * q]     = p0;
            reinterpret_cast<dbl2 *>(prod)[2 * q + 1] = p1;
          }
        }
      }
      __syncthreads();
      if (DBG >= 2) {
        if (r0 + t < r1) yout[r0 + t] = prod[t] + (double)(rs[0] + re[0]);
        return;
      }
#pragma unroll
      for (int j = 0; j < RPT; j++) {
        const hipx_int row = r0 + t + j * SPMV_THREADS;
        if (row < r1) {
          const hipx_int orow = CPROW ? ridx[row] : row;
          double         sum  = (MODE == 1) ? yin[orow] : 0.0;
          const double  *pr   = prod + (int)(rs[j] - ka);
          const int      len  = (int)(re[j] - rs[j]);
          int            k    = 0;
          for (; k + 4 <= len; k += 4) {  // four independent LDS reads, then the dependent left-to-right adds
            const double p0 = pr[k], p1 = pr[k + 1], p2 = pr[k + 2], p3 = pr[k + 3];
            sum += p0;
            sum += p1;
            sum += p2;
            sum += p3;
          }
          if (k + 2 <= len) {
            const double p0 = pr[k], p1 = pr[k + 1];
            sum += p0;
            sum += p1;
            k += 2;
          }
          if (k < len) sum += pr[k];
          yout[orow] = sum;
          if (DOT) mydot += x[orow] * sum;
        }
      }
    } else {
      // long row(s): the host guarantees r1 == r0 + 1 here.  Block-wide strided products, tree sum.
      double acc = 0.0;
      for (IT k = k0 + t; k < k1; k += SPMV_THREADS) acc += aa[k] * x[aj[k]];
      acc = hipx::wave_sum(acc);
      if ((t & 63) == 0) prod[t >> 6] = acc;
      __syncthreads();
      if (t == 0) {
        const hipx_int orow = CPROW ? ridx[r0] : r0;
        double         sum  = (MODE == 1) ? yin[orow] : 0.0;
        double         tot  = prod[0];
        for (int w = 1; w < SPMV_THREADS / 64; w++) tot += prod[w];
        sum += tot;
        yout[orow] = sum;
        if (DOT) mydot = x[orow] * sum;
      }
    }
  }
  if (DOT) {  // one partial per WAVE, no barrier: the workgroup retires as soon as its rows are written
    const double w = hipx::wave_sum(mydot);
    if ((threadIdx.x & 63) == 0) dotpart[(size_t)bid * (blockDim.x >> 6) + (threadIdx.x >> 6)] = w;
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Packed-column SpMV: the stream kernel with the 4-byte column replaced by a 2-byte (window id : 4, offset : 12) code.
// Each row block records up to 16 windows of x (start columns); column = start[id] + offset.  The window starts of a
// block live in lanes 0..15 of every wave and are fetched with a wave shuffle, so the gather still goes through the
// hardware (L1/L2) path, with no extra barrier and no extra LDS.  Index traffic: 2 instead of 4 bytes per nonzero.
// Blocks whose columns need more than 16 windows of 4096 keep their 32-bit columns (per-block fallback).
constexpr int PK_WMAX = 16;
constexpr int PK_WLEN = 4096;

// VD = true: values come from the <= 256-entry dictionary of exact bit patterns through a 1-byte code (see spmv_vd_kernel);
// the dictionary is read through L1 (2 KiB, always resident), so no extra barrier is needed before the products.
template <typename IT, int MODE, bool DOT, bool VD>
__global__ __launch_bounds__(256) void spmv_pk16_kernel(const hipx_int *__restrict__ rb, hipx_int nblocks, hipx_int blocks_per_xcd, const IT *__restrict__ ai,
                                                        const hipx_int *__restrict__ aj, const unsigned short *__restrict__ pk, const hipx_int *__restrict__ pkbase,
                                                        const double *__restrict__ aa, const unsigned char *__restrict__ vc, const double *__restrict__ vdict,
                                                        const double *__restrict__ x, const double *yin, double *yout, double *dotpart, hipx_int ncols)
{
  constexpr int THREADS = 256, CAP = 2048;
  __shared__ double prod[CAP];
  const hipx_int bid = (hipx_int)blockIdx.x;
  const hipx_int b   = (bid & 7) * blocks_per_xcd + (bid >> 3);
  double         mydot = 0.0;
  if (b < nblocks) {
    const hipx_int r0 = rb[b], r1 = rb[b + 1];
    const IT       k0 = ai[r0], k1 = ai[r1];
    const IT       ka = k0 & ~(IT)3;
    const int      t  = threadIdx.x;
    const hipx_int row = r0 + t;
    IT             rs = 0, re = 0;
    if (row < r1) {
      rs = ai[row];
      re = ai[row + 1];
    }
    const double xrow = (DOT && row < r1) ? x[row] : 0.0;  // early: see spmv_pk16r_kernel
    const int base_reg = pkbase[(size_t)b * PK_WMAX + (t & (PK_WMAX - 1))];  // lanes 0..15 of each wave hold the window starts
    const int packed   = __shfl(base_reg, 0, 64) >= 0;                          // wave-uniform: slot 0 is -1 for fallback blocks
    if ((k1 - ka) <= (IT)CAP) {
      const IT      nq  = (k1 - ka + 3) >> 2;
      constexpr int NIT = CAP / 4 / THREADS;
      if (nq > 0) {
        const dbl2 *a2 = reinterpret_cast<const dbl2 *>(aa + ka);
        dbl2        va[NIT], vb[NIT];
        int         c[NIT][4];
#pragma unroll
        for (int it = 0; it < NIT; it++) {
          const IT q  = (IT)t + (IT)it * THREADS;
          const IT qc = q < nq ? q : nq - 1;
          if (VD && packed) {
            const uchar4v u = reinterpret_cast<const uchar4v *>(vc + ka)[qc];
            va[it].x = vdict[u.x];
            va[it].y = vdict[u.y];
            vb[it].x = vdict[u.z];
            vb[it].y = vdict[u.w];
          } else {
            va[it] = a2[2 * qc];
            vb[it] = a2[2 * qc + 1];
          }
          if (packed) {
            const ushort4v v = reinterpret_cast<const ushort4v *>(pk + ka)[qc];
            c[it][0] = __shfl(base_reg, v.x >> 12, 64) + (v.x & 0xfff);
            c[it][1] = __shfl(base_reg, v.y >> 12, 64) + (v.y & 0xfff);
            c[it][2] = __shfl(base_reg, v.z >> 12, 64) + (v.z & 0xfff);
            c[it][3] = __shfl(base_reg, v.w >> 12, 64) + (v.w & 0xfff);
            // the first / last quad of a block also carries entries of the neighbouring blocks, whose codes refer to THEIR
            // windows: decoded against ours they may point anywhere -> clamp (their products are never summed)
#pragma unroll
            for (int e = 0; e < 4; e++) c[it][e] = ((unsigned)c[it][e] < (unsigned)ncols) ? c[it][e] : 0;
          } else {
            const int4v v = reinterpret_cast<const int4v *>(aj + ka)[qc];
            c[it][0] = v.x;
            c[it][1] = v.y;
            c[it][2] = v.z;
            c[it][3] = v.w;
          }
        }
        double xv[NIT][4];
#pragma unroll
        for (int it = 0; it < NIT; it++) {
#pragma unroll
          for (int e = 0; e < 4; e++) xv[it][e] = x[c[it][e]];
        }
#pragma unroll
        for (int it = 0; it < NIT; it++) {
          const IT q = (IT)t + (IT)it * THREADS;
          if (q < nq) {
            dbl2 p0, p1;
            p0.x = va[it].x * xv[it][0];
            p0.y = va[it].y * xv[it][1];
            p1.x = vb[it].x * xv[it][2];
            p1.y = vb[it].y * xv[it][3];
            reinterpret_cast<dbl2 *>(prod)[2 * q]     = p0;
            reinterpret_cast<dbl2 *>(prod)[2 * q + 1] = p1;
          }
        }
      }
      __syncthreads();
      if (row < r1) {
        double        sum = (MODE == 1) ? yin[row] : 0.0;
        const double *pr  = prod + (int)(rs - ka);
        const int     len = (int)(re - rs);
        int           k   = 0;
        for (; k + 4 <= len; k += 4) {
          const double p0 = pr[k], p1 = pr[k + 1], p2 = pr[k + 2], p3 = pr[k + 3];
          sum += p0;
          sum += p1;
          sum += p2;
          sum += p3;
        }
        if (k + 2 <= len) {
          const double p0 = pr[k], p1 = pr[k + 1];
          sum += p0;
          sum += p1;
          k += 2;
        }
        if (k < len) sum += pr[k];
        yout[row] = sum;
        if (DOT) mydot = xrow * sum;
      }
    } else {  // one long row
      double acc = 0.0;
      for (IT k = k0 + t; k < k1; k += THREADS) acc += aa[k] * x[aj[k]];
      acc = hipx::wave_sum(acc);
      if ((t & 63) == 0) prod[t >> 6] = acc;
      __syncthreads();
      if (t == 0) {
        double sum = (MODE == 1) ? yin[r0] : 0.0;
        double tot = prod[0];
        for (int w = 1; w < THREADS / 64; w++) tot += prod[w];
        sum += tot;
        yout[r0] = sum;
        if (DOT) mydot = x[r0] * sum;
      }
    }
  }
  if (DOT) {  // one partial per WAVE, no barrier: the workgroup retires as soon as its rows are written
    const double w = hipx::wave_sum(mydot);
    if ((threadIdx.x & 63) == 0) dotpart[(size_t)bid * (blockDim.x >> 6) + (threadIdx.x >> 6)] = w;
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Packed-column SpMV, row-parallel gather ("pk16r").  Phase 1 only moves the block's values and 16-bit column codes
// into LDS (coalesced 32 B + 8 B per lane); phase 2 lets thread t walk row t left to right: value and code from LDS,
// x through the hardware gather.  Because neighbouring lanes now hold neighbouring ROWS, the k-th gather of a wave
// touches x[row + offset_k] for 64 consecutive rows -- 4-5 cache lines instead of the ~20 that the nonzero-major
// order of spmv_pk16_kernel spreads one gather instruction over (7 stencil offsets interleaved across the lanes).
// Same products, same left-to-right sums: y is bit-identical.  Blocks without a packed code keep the pk16 path.
template <typename IT, int MODE, bool DOT>
__global__ __launch_bounds__(256) void spmv_pk16r_kernel(const hipx_int *__restrict__ rb, hipx_int nblocks, hipx_int blocks_per_xcd, const IT *__restrict__ ai,
                                                         const hipx_int *__restrict__ aj, const unsigned short *__restrict__ pk, const hipx_int *__restrict__ pkbase,
                                                         const double *__restrict__ aa, const double *__restrict__ x, const double *yin, double *yout, double *dotpart, hipx_int ncols,
                                                         const int nt = 0)
{
  // nt (round 6, HIPX_SPMV_NT_STREAM): the value / code streams are read once per product -- loaded non-temporally they do not push the planes of x
  // the gathers re-use (a plane serves three generations of workgroups) out of the XCD's L2
  constexpr int THREADS = 256, CAP = 2048;
  __shared__ double         vals[CAP];
  __shared__ unsigned short codes[CAP];
  const hipx_int bid = (hipx_int)blockIdx.x;
  const hipx_int b   = (bid & 7) * blocks_per_xcd + (bid >> 3);
  double         mydot = 0.0;
  if (b < nblocks) {
    const hipx_int r0 = rb[b], r1 = rb[b + 1];
    const IT       k0 = ai[r0], k1 = ai[r1];
    const IT       ka = k0 & ~(IT)3;
    const int      t  = threadIdx.x;
    const hipx_int row = r0 + t;
    IT             rs = 0, re = 0;
    if (row < r1) {
      rs = ai[row];
      re = ai[row + 1];
    }
    // x[row] for the fused dot is fetched NOW: issued at the tail it would add one memory round trip to the life of every
    // wave (262k waves / 8k resident = 32 generations x ~1 us = the 32 us the fused kernel used to lose)
    const double xrow = (DOT && row < r1) ? x[row] : 0.0;
    const int base_reg = pkbase[(size_t)b * PK_WMAX + (t & (PK_WMAX - 1))];
    const int packed   = __shfl(base_reg, 0, 64) >= 0;
    if ((k1 - ka) <= (IT)CAP) {
      const IT      nq  = (k1 - ka + 3) >> 2;
      constexpr int NIT = CAP / 4 / THREADS;
      if (packed) {
        if (nq > 0) {
          const dbl2     *a2 = reinterpret_cast<const dbl2 *>(aa + ka);
          const ushort4v *c4 = reinterpret_cast<const ushort4v *>(pk + ka);
          dbl2            va[NIT], vb[NIT];
          ushort4v        vq[NIT];
#pragma unroll
          for (int it = 0; it < NIT; it++) {
            const IT q  = (IT)t + (IT)it * THREADS;
            const IT qc = q < nq ? q : nq - 1;
            if (nt) {
              va[it] = __builtin_nontemporal_load(a2 + 2 * qc);
              vb[it] = __builtin_nontemporal_load(a2 + 2 * qc + 1);
              vq[it] = __builtin_nontemporal_load(c4 + qc);
            } else {
              va[it] = a2[2 * qc];
              vb[it] = a2[2 * qc + 1];
              vq[it] = c4[qc];
            }
          }
#pragma unroll
          for (int it = 0; it < NIT; it++) {
            const IT q = (IT)t + (IT)it * THREADS;
            if (q < nq) {
              reinterpret_cast<dbl2 *>(vals)[2 * q]     = va[it];
              reinterpret_cast<dbl2 *>(vals)[2 * q + 1] = vb[it];
              reinterpret_cast<ushort4v *>(codes)[q]    = vq[it];
            }
          }
        }
        __syncthreads();
        // all 64 lanes of a wave run the shuffle loop the same number of times (wave-max row length): __shfl needs
        // the source lanes (0..15, which hold the window starts) active
        const int len = (row < r1) ? (int)(re - rs) : 0;
        int       maxlen = len;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off, 64));
        const int s0  = (int)(rs - ka);
        double    sum = (MODE == 1 && row < r1) ? yin[row] : 0.0;
        for (int k = 0; k < maxlen; k += 4) {
          double xv[4], av[4];
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const bool     on   = (k + e) < len;
            const int      idx  = on ? s0 + k + e : 0;
            const unsigned code = codes[idx];
            const int      col  = __shfl(base_reg, code >> 12, 64) + (int)(code & 0xfff);
            av[e]               = vals[idx];
            xv[e]               = on ? x[col] : 0.0;
          }
#pragma unroll
          for (int e = 0; e < 4; e++)
            if ((k + e) < len) sum += av[e] * xv[e];
        }
        if (row < r1) {
          yout[row] = sum;
          if (DOT) mydot = xrow * sum;
        }
      } else {
        // 32-bit columns: products staged in LDS (the spmv_stream_kernel path)
        double *prod = vals;
        if (nq > 0) {
          const dbl2 *a2 = reinterpret_cast<const dbl2 *>(aa + ka);
#pragma unroll
          for (int it = 0; it < NIT; it++) {
            const IT q = (IT)t + (IT)it * THREADS;
            if (q < nq) {
              const dbl2  v0 = a2[2 * q], v1 = a2[2 * q + 1];
              const int4v c  = reinterpret_cast<const int4v *>(aj + ka)[q];
              dbl2        p0, p1;
              p0.x = v0.x * x[c.x];
              p0.y = v0.y * x[c.y];
              p1.x = v1.x * x[c.z];
              p1.y = v1.y * x[c.w];
              reinterpret_cast<dbl2 *>(prod)[2 * q]     = p0;
              reinterpret_cast<dbl2 *>(prod)[2 * q + 1] = p1;
            }
          }
        }
        __syncthreads();
        if (row < r1) {
          double        sum = (MODE == 1) ? yin[row] : 0.0;
          const double *pr  = prod + (int)(rs - ka);
          const int     len = (int)(re - rs);
          for (int k = 0; k < len; k++) sum += pr[k];
          yout[row] = sum;
          if (DOT) mydot = xrow * sum;
        }
      }
    } else {  // one long row
      double acc = 0.0;
      for (IT k = k0 + t; k < k1; k += THREADS) acc += aa[k] * x[aj[k]];
      acc = hipx::wave_sum(acc);
      if ((t & 63) == 0) vals[t >> 6] = acc;
      __syncthreads();
      if (t == 0) {
        double sum = (MODE == 1) ? yin[r0] : 0.0;
        double tot = vals[0];
        for (int w = 1; w < THREADS / 64; w++) tot += vals[w];
        sum += tot;
        yout[r0] = sum;
        if (DOT) mydot = x[r0] * sum;
      }
    }
  }
  if (DOT) {  // one partial per WAVE, no barrier: the workgroup retires as soon as its rows are written
    const double w = hipx::wave_sum(mydot);
    if ((threadIdx.x & 63) == 0) dotpart[(size_t)bid * (blockDim.x >> 6) + (threadIdx.x >> 6)] = w;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Value-dictionary SpMV with RPT rows per thread ("vd").  With 3 bytes of matrix per nonzero the kernel is no longer bound
// by HBM but by the length of each wave's dependent chain (block bounds -> row offsets / codes -> barrier -> gathers), so:
//  * the block bounds (r0, r1, k0, k1) come from one 24-byte descriptor instead of rb[] followed by ai[rb[]];
//  * a thread owns RPT rows (row blocks of 256*RPT rows, 2048*RPT nonzeros), whose gathers are issued together: the fixed
//    latencies are paid once per RPT rows.
//  * phase 1 merges the 16-bit column code and the 8-bit value code of a nonzero into one 32-bit LDS word, so phase 2 issues
//    one bank-conflict-free LDS read per nonzero (a row's words are consecutive; rows of odd length spread over all banks).
// Products and left-to-right row sums are those of MatMult_SeqAIJ (aij.c:1486-1494): y is bit-identical.
struct PkDesc {
  hipx_int  r0, r1;
  long long k0, k1;
};

template <typename IT, int MODE, bool DOT, int RPT, int W, int DBG = 0>
__global__ __launch_bounds__(256) void spmv_vd_kernel(const PkDesc *__restrict__ desc, hipx_int nblocks, hipx_int blocks_per_xcd, const IT *__restrict__ ai, const hipx_int *__restrict__ aj,
                                                      const unsigned short *__restrict__ pk, const hipx_int *__restrict__ pkbase, const double *__restrict__ aa,
                                                      const unsigned char *__restrict__ vc, const double *__restrict__ vdict, int ndict, const double *__restrict__ x, const double *yin,
                                                      double *yout, double *dotpart, hipx_int ncols)
{
  constexpr int THREADS = 256, CAP = 2048 * RPT;
  __shared__ double   dict[256];
  __shared__ unsigned words[CAP + 16];  // (value code << 16) | column code: one conflict-free 32-bit LDS read per nonzero in phase 2
  const hipx_int bid = (hipx_int)blockIdx.x;
  const hipx_int b   = (bid & 7) * blocks_per_xcd + (bid >> 3);
  double         mydot = 0.0;
  if (b < nblocks) {
    const PkDesc   d  = desc[b];
    const hipx_int r0 = d.r0, r1 = d.r1;
    const IT       k0 = (IT)d.k0, k1 = (IT)d.k1;
    const IT       ka = k0 & ~(IT)3;
    const int      t  = threadIdx.x;
    // The gather address unit is the busiest block of this kernel (profiles/r01c_spmv_counters.json), so the bookkeeping
    // loads are kept to the lanes that need them: one row offset per lane (the end offset comes from the next lane),
    // window starts on lanes 0..15 of each wave, the dictionary on its first ndict lanes.
    IT             rs[RPT], re[RPT];
    double         xrow[RPT];
#pragma unroll
    for (int rr = 0; rr < RPT; rr++) {
      const hipx_int row = r0 + t + rr * THREADS;
      const IT       me  = (row <= r1) ? ai[row] : (IT)0;           // ai[r1] exists: r1 <= number of rows
      IT             nx;                                            // = ai[row + 1] for lanes 0..62
      if constexpr (sizeof(IT) == 4) nx = (IT)__shfl_down((int)me, 1, 64);
      else nx = (IT)__shfl_down((long long)me, 1, 64);
      if ((t & 63) == 63) nx = (row < r1) ? ai[row + 1] : (IT)0;
      rs[rr]   = (row < r1) ? me : (IT)0;
      re[rr]   = (row < r1) ? nx : (IT)0;
      xrow[rr] = (DOT && row < r1) ? x[row] : 0.0;
    }
    const int base_reg = ((t & 63) < PK_WMAX) ? pkbase[(size_t)b * PK_WMAX + (t & (PK_WMAX - 1))] : 0;
    const int packed   = __shfl(base_reg, 0, 64) >= 0;
    if ((k1 - ka) <= (IT)CAP) {
      if (packed) {
        const IT  ka8 = k0 & ~(IT)7;
        const IT  nq8 = (k1 - ka8 + 7) >> 3;
        if (t < ndict) dict[t] = vdict[t];
        for (IT q = t; q < nq8; q += THREADS) {
          int4v              c8;
          unsigned long long v8;
          if constexpr (DBG == 3) {
            c8 = int4v{0, 0, 0, 0};
            v8 = 0;
          } else {
            c8 = reinterpret_cast<const int4v *>(pk + ka8)[q];
            v8 = reinterpret_cast<const unsigned long long *>(vc + ka8)[q];
          }
          int4v w0, w1;  // 8 nonzeros: column codes c8 = 8 x u16, value codes v8 = 8 x u8
          const unsigned vlo = (unsigned)v8, vhi = (unsigned)(v8 >> 32);
          w0.x = (int)(((unsigned)c8.x & 0xffffu) | ((vlo & 0xffu) << 16));
          w0.y = (int)(((unsigned)c8.x >> 16) | ((vlo & 0xff00u) << 8));
          w0.z = (int)(((unsigned)c8.y & 0xffffu) | (vlo & 0xff0000u));
          w0.w = (int)(((unsigned)c8.y >> 16) | ((vlo >> 24) << 16));
          w1.x = (int)(((unsigned)c8.z & 0xffffu) | ((vhi & 0xffu) << 16));
          w1.y = (int)(((unsigned)c8.z >> 16) | ((vhi & 0xff00u) << 8));
          w1.z = (int)(((unsigned)c8.w & 0xffffu) | (vhi & 0xff0000u));
          w1.w = (int)(((unsigned)c8.w >> 16) | ((vhi >> 24) << 16));
          reinterpret_cast<int4v *>(words)[2 * q]     = w0;
          reinterpret_cast<int4v *>(words)[2 * q + 1] = w1;
        }
        __syncthreads();
        int    len[RPT], s0[RPT], maxlen = 0;
        double sum[RPT];
#pragma unroll
        for (int rr = 0; rr < RPT; rr++) {
          const hipx_int row = r0 + t + rr * THREADS;
          len[rr] = (row < r1) ? (int)(re[rr] - rs[rr]) : 0;
          s0[rr]  = (int)(rs[rr] - ka8);
          sum[rr] = (MODE == 1 && row < r1) ? yin[row] : 0.0;
          maxlen  = max(maxlen, len[rr]);
        }
        // every lane of a wave runs the same trip count: __shfl needs lanes 0..15 (the window starts) active
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off, 64));
        if (DBG == 2) maxlen = 0;
        for (int k = 0; k < maxlen; k += W) {
          double xv[RPT][W], av[RPT][W];
#pragma unroll
          for (int rr = 0; rr < RPT; rr++) {
#pragma unroll
            for (int e = 0; e < W; e++) {
              const bool     on   = (k + e) < len[rr];
              const int      idx  = on ? s0[rr] + k + e : 0;
              const unsigned word = words[idx];
              const int      col  = __shfl(base_reg, (word >> 12) & 15, 64) + (int)(word & 0xfff);
              av[rr][e]           = dict[word >> 16];
              if constexpr (DBG == 1) xv[rr][e] = (double)col;
              else xv[rr][e] = on ? x[col] : 0.0;
            }
          }
#pragma unroll
          for (int rr = 0; rr < RPT; rr++) {
#pragma unroll
            for (int e = 0; e < W; e++)
              if ((k + e) < len[rr]) sum[rr] += av[rr][e] * xv[rr][e];
          }
        }
#pragma unroll
        for (int rr = 0; rr < RPT; rr++) {
          const hipx_int row = r0 + t + rr * THREADS;
          if (row < r1) {
            yout[row] = sum[rr];
            if (DOT) mydot += xrow[rr] * sum[rr];
          }
        }
      } else {
        // block kept its 32-bit columns (scattered columns AND few distinct values: rare): plain row walk
#pragma unroll
        for (int rr = 0; rr < RPT; rr++) {
          const hipx_int row = r0 + t + rr * THREADS;
          if (row < r1) {
            double sum = (MODE == 1) ? yin[row] : 0.0;
            for (IT k = rs[rr]; k < re[rr]; k++) sum += aa[k] * x[aj[k]];
            yout[row] = sum;
            if (DOT) mydot += xrow[rr] * sum;
          }
        }
      }
    } else {  // one long row
      double acc = 0.0;
      for (IT k = k0 + t; k < k1; k += THREADS) acc += aa[k] * x[aj[k]];
      acc = hipx::wave_sum(acc);
      if ((t & 63) == 0) dict[t >> 6] = acc;
      __syncthreads();
      if (t == 0) {
        double sum = (MODE == 1) ? yin[r0] : 0.0;
        double tot = dict[0];
        for (int w = 1; w < THREADS / 64; w++) tot += dict[w];
        sum += tot;
        yout[r0] = sum;
        if (DOT) mydot = x[r0] * sum;
      }
    }
  }
  if (DOT) {
    const double w = hipx::wave_sum(mydot);
    if ((threadIdx.x & 63) == 0) dotpart[(size_t)bid * (blockDim.x >> 6) + (threadIdx.x >> 6)] = w;
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Pattern-template SpMV ("tp", variant 29): spmv_pk16r_kernel without column codes.  The column of a row's k-th entry is
// row + toff[tstart[id] + k] with id = ptid[row] (one byte per row, <= 256 distinct (column - row) lists: stencil PATTERNS with any
// values); the values are streamed from a[] exactly as in the CSR kernels.  Matrix bytes per launch: 8 nnz + (1 + 4) N against
// 12 nnz + 4 N of CSR and 10 nnz + 4 N of the packed-column kernels.  Phase 1 copies the row block's values to LDS (coalesced),
// phase 2 lets thread t walk row t left to right (value from LDS, offset from the table -- neighbouring lanes share the pattern,
// so the table read is one cache line per wave --, x through the gather: x[row + offset] of 64 consecutive rows = 4-5 lines).
// Same products, same order: y is bit-identical.
template <typename IT, int MODE, bool DOT>
__global__ __launch_bounds__(256) void spmv_tp_kernel(const hipx_int *__restrict__ rb, hipx_int nblocks, hipx_int blocks_per_xcd, const IT *__restrict__ ai, const unsigned char *__restrict__ ptid,
                                                      const int *__restrict__ tstart, const int *__restrict__ toff, const double *__restrict__ aa, const double *__restrict__ x, const double *yin,
                                                      double *yout, double *dotpart, const int nt = 0)
{
  constexpr int THREADS = 256, CAP = 2048;
  __shared__ double vals[CAP];
  const hipx_int bid = (hipx_int)blockIdx.x;
  const hipx_int b   = (bid & 7) * blocks_per_xcd + (bid >> 3);
  double         mydot = 0.0;
  if (b < nblocks) {
    const hipx_int r0 = rb[b], r1 = rb[b + 1];
    const IT       k0 = ai[r0], k1 = ai[r1];
    const IT       ka = k0 & ~(IT)3;
    const int      t  = threadIdx.x;
    const hipx_int row = r0 + t;
    IT             rs = 0, re = 0;
    int            ts = 0;
    if (row < r1) {
      rs = ai[row];
      re = ai[row + 1];
      ts = tstart[ptid[row]];
    }
    const double xrow = (DOT && row < r1) ? x[row] : 0.0;
    const IT     nq  = (k1 - ka + 3) >> 2;  // (row blocks hold <= CAP entries: patterns longer than 1024 entries never get here)
    constexpr int NIT = CAP / 4 / THREADS;
    if (nq > 0) {
      const dbl2 *a2 = reinterpret_cast<const dbl2 *>(aa + ka);
      dbl2        va[NIT], vb[NIT];
#pragma unroll
      for (int it = 0; it < NIT; it++) {
        const IT q  = (IT)t + (IT)it * THREADS;
        const IT qc = q < nq ? q : nq - 1;
        if (nt) {  // (see spmv_pk16r_kernel)
          va[it] = __builtin_nontemporal_load(a2 + 2 * qc);
          vb[it] = __builtin_nontemporal_load(a2 + 2 * qc + 1);
        } else {
          va[it] = a2[2 * qc];
          vb[it] = a2[2 * qc + 1];
        }
      }
#pragma unroll
      for (int it = 0; it < NIT; it++) {
        const IT q = (IT)t + (IT)it * THREADS;
        if (q < nq) {
          reinterpret_cast<dbl2 *>(vals)[2 * q]     = va[it];
          reinterpret_cast<dbl2 *>(vals)[2 * q + 1] = vb[it];
        }
      }
    }
    __syncthreads();
    if (row < r1) {
      const int  len = (int)(re - rs);
      const int  s0  = (int)(rs - ka);
      const int *to  = toff + ts;
      double     sum = (MODE == 1) ? yin[row] : 0.0;
      for (int k = 0; k < len; k += 4) {
        double xv[4], av[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const bool on = (k + e) < len;
          const int  kk = on ? k + e : 0;
          av[e]         = vals[s0 + kk];
          xv[e]         = x[(long long)row + to[kk]];
        }
#pragma unroll
        for (int e = 0; e < 4; e++)
          if ((k + e) < len) sum += av[e] * xv[e];
      }
      yout[row] = sum;
      if (DOT) mydot = xrow * sum;
    }
  }
  if (DOT) {
    const double w = hipx::wave_sum(mydot);
    if ((threadIdx.x & 63) == 0) dotpart[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = w;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Set-up of the packed formats ON THE DEVICE (once per nonzero pattern / value state; ~1 ms at 117 M nonzeros, where the
// first host-side version needed 1.4 GB of D2H copies and 0.5 s of host work).
//
// pk_build_kernel: one workgroup per row block.  Windows are aligned to 4096 columns (window key = col >> 12): the block's
// distinct keys are collected in a 64-slot LDS set (atomicCAS), sorted by one thread, and every column is rewritten as
// (rank of its key : 4 | col & 4095 : 12).  More than 16 keys -> the block keeps its 32-bit columns (base[] = -1).
__global__ __launch_bounds__(256) void pk_build_kernel(const PkDesc *__restrict__ desc, hipx_int nblocks, long long cap, const hipx_int *__restrict__ aj, unsigned short *__restrict__ pk,
                                                       hipx_int *__restrict__ pkbase, unsigned int *fallback)
{
  __shared__ int keys[64];
  __shared__ int sorted[PK_WMAX];
  __shared__ int nkeys, bad;
  const hipx_int b = (hipx_int)blockIdx.x;
  if (b >= nblocks) return;
  const PkDesc    d    = desc[b];
  const long long k0   = d.k0, k1 = d.k1;
  hipx_int       *base = pkbase + (size_t)b * PK_WMAX;
  const int       t    = threadIdx.x;
  if (k1 - (k0 & ~3LL) > cap || k1 == k0) {  // one long row (block-wide path) or an empty block
    if (t < PK_WMAX) base[t] = -1;
    return;
  }
  if (t < 64) keys[t] = -1;
  if (t == 0) {
    nkeys = 0;
    bad   = 0;
  }
  __syncthreads();
  for (long long k = k0 + t; k < k1; k += 256) {
    const int key = aj[k] >> 12;
    unsigned  s   = ((unsigned)key * 0x9E3779B1u) >> 26;
    for (int probe = 0; probe < 64; probe++) {
      const int old = atomicCAS(&keys[s], -1, key);
      if (old == -1) {
        if (atomicAdd(&nkeys, 1) >= PK_WMAX) bad = 1;
        break;
      }
      if (old == key) break;
      s = (s + 1) & 63;
    }
    if (bad) break;
  }
  __syncthreads();
  if (bad || nkeys > PK_WMAX) {
    if (t < PK_WMAX) base[t] = -1;
    if (t == 0) atomicAdd(fallback, 1u);
    return;
  }
  if (t == 0) {
    int n = 0;
    for (int s = 0; s < 64; s++)
      if (keys[s] != -1) {
        int key = keys[s], i = n++;
        while (i > 0 && sorted[i - 1] > key) {
          sorted[i] = sorted[i - 1];
          i--;
        }
        sorted[i] = key;
      }
    for (int w = n; w < PK_WMAX; w++) sorted[w] = sorted[n - 1];
  }
  __syncthreads();
  if (t < PK_WMAX) base[t] = sorted[t] << 12;
  for (long long k = k0 + t; k < k1; k += 256) {
    const int c = aj[k], key = c >> 12;
    int       w = 0;
#pragma unroll
    for (int i = PK_WMAX - 1; i >= 0; i--)
      if (sorted[i] == key) w = i;  // first match (the padding repeats the last key)
    pk[k] = (unsigned short)((w << 12) | (c & (PK_WLEN - 1)));
  }
}

// Value dictionary, pass 1: every workgroup collects the distinct bit patterns of its contiguous chunk of a[] in a 512-slot
// LDS set (64-bit atomicCAS) and appends them (<= 256) to a global list; the host merges the lists (a few thousand
// entries), sorts, and assigns the codes.  More than 256 distinct patterns anywhere -> overflow flag, everybody stops.
constexpr unsigned long long VD_EMPTY = 0x7FF8DEADBEEF0001ull;  // a NaN payload no assembled matrix holds; if one does: no dictionary
__device__ __forceinline__ unsigned vd_hash(unsigned long long k, int bits) { return (unsigned)((k * 0x9E3779B97F4A7C15ull) >> (64 - bits)); }

__global__ __launch_bounds__(256) void vd_collect_kernel(const unsigned long long *__restrict__ a, long long nnz, unsigned long long *__restrict__ list, unsigned int *counters,
                                                         unsigned int list_cap)
{
  __shared__ unsigned long long tab[512];
  __shared__ unsigned int       cnt, pos, basei, ovf;
  const int t = threadIdx.x;
  tab[t]       = VD_EMPTY;
  tab[t + 256] = VD_EMPTY;
  if (t == 0) cnt = pos = basei = ovf = 0;
  __syncthreads();
  const long long chunk = (nnz + gridDim.x - 1) / gridDim.x;
  const long long c0 = (long long)blockIdx.x * chunk, c1 = (c0 + chunk < nnz) ? c0 + chunk : nnz;
  unsigned long long last = VD_EMPTY;
  for (long long k = c0 + t; k < c1; k += 256) {
    if (((k - c0) & 0xffff) < 256 && __hip_atomic_load(&counters[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;  // someone overflowed
    const unsigned long long v = a[k];
    if (v == last) continue;
    last = v;
    if (v == VD_EMPTY) {
      ovf = 1;
      break;
    }
    unsigned s = vd_hash(v, 9);
    for (;;) {
      const unsigned long long old = atomicCAS(&tab[s], VD_EMPTY, v);
      if (old == VD_EMPTY) {
        if (atomicAdd(&cnt, 1u) >= 256u) ovf = 1;
        break;
      }
      if (old == v) break;
      s = (s + 1) & 511;
    }
    if (ovf) break;
  }
  __syncthreads();
  if (ovf) {
    if (t == 0) atomicExch(&counters[1], 1u);
    return;
  }
  if (t == 0) basei = atomicAdd(&counters[0], cnt);
  __syncthreads();
  for (int s = t; s < 512; s += 256)
    if (tab[s] != VD_EMPTY) {
      const unsigned r = basei + atomicAdd(&pos, 1u);
      if (r < list_cap) list[r] = tab[s];
    }
}

// pass 2: a[k] -> 1-byte code through the (host-built) 1024-slot table, 8 nonzeros per thread and store
__global__ __launch_bounds__(256) void vd_encode_kernel(const unsigned long long *__restrict__ a, long long nnz, const unsigned long long *__restrict__ gkeys, const short *__restrict__ gcodes,
                                                        unsigned char *__restrict__ vc)
{
  __shared__ unsigned long long keys[1024];
  __shared__ short              codes[1024];
  for (int s = threadIdx.x; s < 1024; s += 256) {
    keys[s]  = gkeys[s];
    codes[s] = gcodes[s];
  }
  __syncthreads();
  const long long nq = (nnz + 7) >> 3;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < nq; q += (long long)gridDim.x * 256) {
    unsigned long long out = 0;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const long long k = q * 8 + e;
      if (k < nnz) {
        const unsigned long long v = a[k];
        unsigned                 s = vd_hash(v, 10);
        while (codes[s] >= 0 && keys[s] != v) s = (s + 1) & 1023;
        out |= (unsigned long long)(unsigned char)codes[s] << (8 * e);
      }
    }
    reinterpret_cast<unsigned long long *>(vc)[q] = out;
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Row templates.  A stencil matrix in natural ordering repeats a handful of rows: the sequence of (column - row, value)
// pairs of a row is one of 27 "templates" for the 7- and 27-point operators (interior + boundary combinations).  When a
// matrix has <= 256 distinct templates (<= TMPL_MAX_ENT entries in total) it is stored as ONE BYTE PER ROW plus the
// template table; MatMult then moves x, y and 1 B/row instead of 12 B per nonzero.  The products and the left-to-right
// row sums are those of MatMult_SeqAIJ (aij.c:1486-1494) on the same doubles: y is bit-identical.
// Set-up on the device: 64-bit hash per row -> distinct hashes (vd_collect_kernel) -> ids + representative rows ->
// tmpl_verify_kernel compares EVERY row with its template entry by entry (a hash collision disables the format, it can
// never produce a wrong matrix).
constexpr int TMPL_MAX     = 256;
constexpr int TMPL_MAX_ENT = 3072;

__device__ __forceinline__ unsigned long long tm_mix(unsigned long long h, unsigned long long v)
{
  h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
  h *= 0xff51afd7ed558ccdULL;
  h ^= h >> 32;
  return h;
}

// Coalesced row walk for the set-up passes that look at EVERY entry of the matrix once (round 5).  With one thread per row and each thread reading
// its own row, neighbouring lanes are a whole row apart: 27-pt 512^3 (43 GB of columns and values) took 111 ms per pass = 0.39 TB/s.  Here a wave
// owns 64 CONSECUTIVE rows = one contiguous span of the column / value arrays: the span passes through LDS in windows of RW_CH entries, loaded by
// consecutive lanes (4- and 8-byte accesses, 256 / 512 B per wave instruction); lane l then walks the entries of row r0 + l that lie in the window,
// in order, out of LDS and hands them to f(k, column, value bits).  Per-row results are exactly the ones of the thread-per-row loops.
constexpr int RW_CH = 1024;
__device__ __forceinline__ void rw_wave_sync()
{
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (a wave's LDS operations complete in order; the barrier keeps the compiler from moving them)
  __builtin_amdgcn_wave_barrier();
}
template <typename IT, int CH = RW_CH, class F>
__device__ __forceinline__ void rowwalk64(hipx_int m, const IT *__restrict__ ai, const hipx_int *__restrict__ aj, const unsigned long long *__restrict__ aa, hipx_int r0, int *lcol,
                                          unsigned long long *lval, F &&f)
{
  const int       lane = threadIdx.x & 63;
  const long long r    = (long long)r0 + lane;
  const bool      has  = r < (long long)m;
  const IT        s = has ? ai[r] : (IT)0, e = has ? ai[r + 1] : (IT)0;
  const hipx_int  rl   = (m - r0 > 64) ? r0 + 64 : m;
  const IT        wbeg = ai[r0], wend = ai[rl];
  IT              k    = s;
  for (IT w0 = wbeg; w0 < wend; w0 += CH) {
    const IT wl = (wend - w0 > (IT)CH) ? w0 + (IT)CH : wend;
#pragma unroll 4
    for (int i = lane; i < CH; i += 64) {
      const IT g = w0 + i;
      if (g < wl) {
        lcol[i] = aj[g];
        if (lval) lval[i] = aa[g];
      }
    }
    rw_wave_sync();
    while (k < e && k < wl) {
      const int q = (int)(k - w0);
      f(k, lcol[q], lval ? lval[q] : 0ull);
      k++;
    }
    rw_wave_sync();
  }
}

template <typename IT>
__global__ __launch_bounds__(256) void tmpl_hash_kernel(hipx_int m, const IT *__restrict__ ai, const hipx_int *__restrict__ aj, const unsigned long long *__restrict__ aa,
                                                        unsigned long long *__restrict__ hash, int with_values)
{
  __shared__ int                lcol[4][RW_CH];
  __shared__ unsigned long long lval[4][RW_CH];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (long long rb = (long long)blockIdx.x * 256 + 64 * wv; rb < (long long)m; rb += (long long)gridDim.x * 256) {
    const hipx_int     r0  = (hipx_int)rb;
    const long long    r   = rb + lane;
    const bool         has = r < (long long)m;
    unsigned long long h   = tm_mix(0x243F6A8885A308D3ull, has ? (unsigned long long)(ai[r + 1] - ai[r]) : 0ull);
    rowwalk64<IT>(m, ai, aj, aa, r0, lcol[wv], with_values ? lval[wv] : nullptr, [&](IT, int c, unsigned long long v) {
      h = tm_mix(h, (unsigned long long)(unsigned)(c - (int)r));
      if (with_values) h = tm_mix(h, v);
    });
    if (h == VD_EMPTY) h ^= 1;
    if (has) hash[r] = h;
  }
}

// hash -> template id through the host-built open-addressing table; first row and row count of every template
__global__ __launch_bounds__(256) void tmpl_assign_kernel(hipx_int m, const unsigned long long *__restrict__ hash, const unsigned long long *__restrict__ gkeys,
                                                          const short *__restrict__ gcodes, unsigned char *__restrict__ tid, int *rep, unsigned long long *count)
{
  __shared__ unsigned long long keys[1024];
  __shared__ short              codes[1024];
  __shared__ unsigned int       hist[TMPL_MAX];
  __shared__ int                first[TMPL_MAX];
  for (int s = threadIdx.x; s < 1024; s += 256) {
    keys[s]  = gkeys[s];
    codes[s] = gcodes[s];
  }
  hist[threadIdx.x]  = 0;
  first[threadIdx.x] = 0x7fffffff;
  __syncthreads();
  for (hipx_int r = (hipx_int)blockIdx.x * 256 + threadIdx.x; r < m; r += (hipx_int)gridDim.x * 256) {
    const unsigned long long v = hash[r];
    unsigned                 s = vd_hash(v, 10);
    while (codes[s] >= 0 && keys[s] != v) s = (s + 1) & 1023;
    const int id = codes[s] >= 0 ? codes[s] : 0;
    tid[r]       = (unsigned char)id;
    atomicAdd(&hist[id], 1u);
    atomicMin(&first[id], (int)r);
  }
  __syncthreads();
  if (hist[threadIdx.x]) {
    atomicAdd(&count[threadIdx.x], (unsigned long long)hist[threadIdx.x]);
    atomicMin(&rep[threadIdx.x], first[threadIdx.x]);
  }
}

template <typename IT, int CH>
__global__ __launch_bounds__(256) void tmpl_verify_kernel(hipx_int m, hipx_int ncols, const IT *__restrict__ ai, const hipx_int *__restrict__ aj, const unsigned long long *__restrict__ aa,
                                                          const unsigned char *__restrict__ tid, const int *__restrict__ tstart, const int *__restrict__ toff,
                                                          const unsigned long long *__restrict__ tval, unsigned int *bad, int with_values, int nent)
{
  // dynamic LDS: the four waves' windows (4 CH values, 4 CH columns) and the template table (nent values, nent offsets) -- one LDS read per comparison
  // instead of a dependent global one.  CH = 1024 when the table leaves room for it inside 64 KiB (few templates: every stencil), else 512
  extern __shared__ __attribute__((aligned(16))) char vsm[];
  unsigned long long *lval   = reinterpret_cast<unsigned long long *>(vsm);
  unsigned long long *s_tval = lval + 4 * CH;
  int                *lcol   = reinterpret_cast<int *>(s_tval + nent);
  int                *s_toff = lcol + 4 * CH;
  for (int k = threadIdx.x; k < nent; k += 256) {
    s_toff[k] = toff[k];
    s_tval[k] = with_values ? tval[k] : 0ull;
  }
  __syncthreads();
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (long long rb = (long long)blockIdx.x * 256 + 64 * wv; rb < (long long)m; rb += (long long)gridDim.x * 256) {
    const hipx_int  r0  = (hipx_int)rb;
    const long long r   = rb + lane;
    const bool      has = r < (long long)m;
    const IT        s   = has ? ai[r] : (IT)0, e = has ? ai[r + 1] : (IT)0;
    const int       t = has ? tid[r] : 0, ts = tstart[t], te = tstart[t + 1];
    bool            ok = !has || (long long)(e - s) == (long long)(te - ts);
    rowwalk64<IT, CH>(m, ai, aj, aa, r0, lcol + wv * CH, with_values ? lval + wv * CH : nullptr, [&](IT k, int cj, unsigned long long v) {
      if (ok) {  // (a row of another length than its template's is bad already: its entries are not compared)
        const int       kk = (int)(k - s), off = s_toff[ts + kk];
        const long long c  = r + off;
        ok                 = (cj - (int)r) == off && (!with_values || v == s_tval[ts + kk]) && c >= 0 && c < ncols;
      }
    });
    if (!ok) atomicAdd(bad, 1u);
  }
}

// The chunk queue's ticket, drawn WITHOUT an immediate wait.  Written as atomicAdd() the draw compiles to the returning atomic followed at
// once by s_waitcnt vmcnt(0) (the value is wanted in a scalar register): the drawing wave stood still for the L2 round trip of a
// contended atomic -- 1.6-2.3 us -- at the START of every pass, before it issued its loads, and the other three waves met it at the
// pass's barrier (HIPX_TMPL_TRACE: that was the longest phase of a pass after the scalar-cache misses were gone).  Here the atomic is
// issued by hand and its result is collected at the END of the pass (ticket_collect), a whole pass of work later.
__device__ __forceinline__ void ticket_draw(unsigned long long *ctr, unsigned long long &raw)
{
  const unsigned long long one = 1ull;
  asm volatile("global_atomic_add_x2 %0, %1, %2, off sc0" : "=&v"(raw) : "v"(ctr), "v"(one) : "memory");
}
__device__ __forceinline__ long long ticket_collect(unsigned long long &raw)
{
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(raw) : : "memory");
  return (long long)raw;
}

// Template SpMV.  Persistent workgroups (the template table is loaded into LDS once per workgroup), each XCD walks one
// contiguous slab of row chunks and the workgroups of an XCD take neighbouring chunks, so the x planes a chunk touches
// (rows +-n, +-n^2) are shared through that XCD's L2.  Thread t of a chunk owns rows base + t + rr*256: the k-th gather of
// a wave reads x[row + off_k] for 64 consecutive rows = one 512-byte contiguous run when the lanes share a template
// (interior), and the LDS reads of the template entries broadcast.
// PROBE != 0: developer timing probes (HIPX_TMPL_PROBE, WRONG RESULTS by construction; scripts/spmv_variants.py): 1 = y is not stored;
// 2 = every gather of a row reads x[row + k] (k = entry index: the row's own cache lines -- no far lines at all); 3 = far offsets
// folded into +-4096 doubles of the row (same number of distinct lines per gather, all of them recently touched: no far-plane HBM stream);
// 4 = correct results, but the chunks are assigned statically (round robin over the XCD's workgroups): no ticket atomics, no barriers;
// 5 = 4 without the first-touch prefetch
template <int MODE, bool DOT, int RPT, int W, bool UNI, int PROBE = 0>
__global__ __launch_bounds__(256) void spmv_tmpl_kernel(hipx_int m, hipx_int nchunks, hipx_int chunks_per_xcd, const unsigned char *__restrict__ tid, const int *__restrict__ tstart,
                                                        const int *__restrict__ toff, const double *__restrict__ tval, int ntmpl, int nent, const double *__restrict__ x,
                                                        const double *yin, double *yout, double *dotpart, unsigned long long *tq, unsigned long long launch, long long pf_off,
                                                        const unsigned int *__restrict__ tmask, int tsub)
{
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ long long s_tk;
  double *s_val   = reinterpret_cast<double *>(smem);                      // nent (padded to even)
  int    *s_off   = reinterpret_cast<int *>(smem + 8 * (size_t)((nent + 1) & ~1));
  int    *s_start = s_off + ((nent + 3) & ~3);                             // ntmpl + 1
  unsigned int *s_mask = reinterpret_cast<unsigned int *>(s_start + ntmpl + 1);  // ntmpl (sub-template form)
  const int t = threadIdx.x;
  for (int k = t; k < nent; k += 256) {
    s_val[k] = tval[k];
    s_off[k] = toff[k];
  }
  for (int k = t; k <= ntmpl; k += 256) s_start[k] = tstart[k];
  if (tsub >= 0)
    for (int k = t; k < ntmpl; k += 256) s_mask[k] = tmask[k];
  __syncthreads();
  const hipx_int bid = (hipx_int)blockIdx.x, xcd = bid & 7, bpx = (hipx_int)gridDim.x >> 3;
  const hipx_int c0 = xcd * chunks_per_xcd, c1 = (c0 + chunks_per_xcd < nchunks) ? c0 + chunks_per_xcd : nchunks;
  double         mydot = 0.0;
  // Chunk queue: the workgroups of an XCD (hardware block b runs on XCD b % 8: observed rule, used for locality only) pull the
  // chunks of that XCD's slab IN ORDER from one ticket counter, so at any moment they work on a window of consecutive chunks:
  // the planes of x a window touches (rows +-n, +-n^2) stay in that XCD's 4 MiB L2 and x is fetched from HBM about once
  // (a static chunk -> workgroup map lets the workgroups drift apart: measured 2.6x).  The counters never need a reset: a launch
  // consumes a known number of tickets per XCD (see tbase).
  const long long     nloc = (long long)(c1 > c0 ? c1 - c0 : 0);
  unsigned long long *ctr  = tq + (size_t)xcd * 8;
  // two tickets are always in flight per workgroup (the current chunk and the next one, whose template ids are prefetched while
  // the current chunk computes): every workgroup takes exactly two tickets beyond the end of its XCD's slab
  const long long     tbase = (long long)(launch * (unsigned long long)(nloc + 2 * bpx));
  __shared__ long long s_tk2[2];
  constexpr bool STATICQ = PROBE >= 4;
  if (!STATICQ) {
    if (t == 0) {
      s_tk2[0] = (long long)atomicAdd(ctr, 1ull) - tbase;
      s_tk2[1] = (long long)atomicAdd(ctr, 1ull) - tbase;
    }
    __syncthreads();
  }
  long long tk = STATICQ ? (long long)(bid >> 3) : s_tk2[0], tk1 = STATICQ ? (long long)(bid >> 3) + bpx : s_tk2[1];
  // First-touch prefetch.  Of the gathers of a row only ONE stream misses the L2: the farthest forward offset (+n^2 for a 3-D
  // stencil: the plane nobody has touched yet); everything else was fetched one or two planes ago.  With ~1/7 of the loads
  // going to HBM the chip holds too few bytes in flight (measured 2-3 TB/s).  So the first 16*RPT lanes of a workgroup touch
  // that stream for the chunk one round (= workgroups per XCD) ahead of the one they just finished: one load instruction per
  // workgroup and chunk, the same lines the gathers fetch later, only earlier.  The value is consumed at the END of the next
  // pass (memory operations retire in order: by then every younger load has been waited for anyway).
  double    pf = 0.0;
  unsigned  sink = 0;
  int       idn[RPT];  // template ids of the NEXT chunk (one dependent memory round trip less per chunk)
#pragma unroll
  for (int rr = 0; rr < RPT; rr++) {
    const long long row = (tk < nloc) ? ((long long)(c0 + tk) * (256 * RPT) + t + rr * 256) : (long long)m;
    idn[rr]             = (row < m) ? tid[row] : 0;
  }
  while (tk < nloc) {
    unsigned long long nxt_raw = 0;
    if (!STATICQ && t == 0) ticket_draw(ctr, nxt_raw);  // the ticket after next travels while this chunk is processed (collected at the end of the pass)
    const hipx_int c    = c0 + (hipx_int)tk;
    const hipx_int base = c * (256 * RPT);
    int            id[RPT];
    double         sum[RPT], xrow[RPT];
    bool           uni = UNI && (base + 256 * RPT <= m) && ((unsigned long long)m < (1ull << 28));  // whole chunk inside the matrix; 32-bit byte offsets
    const bool     whole = uni;
#pragma unroll
    for (int rr = 0; rr < RPT; rr++) {
      const hipx_int row = base + t + rr * 256;
      id[rr]   = idn[rr];
      sum[rr]  = 0.0;
      xrow[rr] = 0.0;
      if (row < m) {
        if (MODE == 1) sum[rr] = yin[row];
        if (DOT && !UNI) xrow[rr] = x[row];
      }
    }
#pragma unroll
    for (int rr = 0; rr < RPT; rr++) {  // issue the next chunk's id loads now; they are consumed at the top of the next pass
      const long long row = (tk1 < nloc) ? ((long long)(c0 + tk1) * (256 * RPT) + t + rr * 256) : (long long)m;
      idn[rr]             = (row < m) ? tid[row] : 0;
    }
    int id0 = 0;
    if (UNI) {
      id0 = __builtin_amdgcn_readfirstlane(id[0]);
#pragma unroll
      for (int rr = 0; rr < RPT; rr++) uni = uni && (id[rr] == id0);
      uni = __all(uni);
    }
    if (UNI && uni) {
      // every lane of the wave walks the SAME template (interior rows): offsets and values are wave-uniform scalars read from
      // the global table through the scalar cache, the gather address is (x + off) [scalar] + row * 8 [per lane, computed once]:
      // per nonzero the vector unit issues one load, one multiply and one add, nothing else
      const int ts = tstart[id0], te = tstart[id0 + 1];
      unsigned  rb[RPT];
#pragma unroll
      for (int rr = 0; rr < RPT; rr++) rb[rr] = (unsigned)(base + t + rr * 256) * 8u;
      bool got = false;  // fused dot: x[row] is the diagonal entry's gather (offset 0) -- one request per row block less than loading it again
#pragma unroll 4
      for (int k = ts; k < te; k++) {
        const double a  = tval[k];
        int          o  = toff[k];
        if (PROBE == 2) o = k - ts;
        if (PROBE == 3) o = (o > 4096 || o < -4096) ? (o % 4096) : o;
        const char  *xb = reinterpret_cast<const char *>(x + o);
        double       xv[RPT];
#pragma unroll
        for (int rr = 0; rr < RPT; rr++) xv[rr] = *reinterpret_cast<const double *>(xb + rb[rr]);
#pragma unroll
        for (int rr = 0; rr < RPT; rr++) sum[rr] += a * xv[rr];
        if (DOT && o == 0) {  // (wave-uniform: the template's offsets are scalars)
          got = true;
#pragma unroll
          for (int rr = 0; rr < RPT; rr++) xrow[rr] = xv[rr];
        }
      }
      if (DOT && !got) {
#pragma unroll
        for (int rr = 0; rr < RPT; rr++) xrow[rr] = x[base + t + rr * 256];
      }
    } else if (UNI && PROBE == 0 && whole && tsub >= 0) {
      // Sub-template walk.  The lanes of this wave do not share a template (a line's first / last row, ...), but every template is
      // the base template with entries left out: all lanes walk the BASE template's entries -- scalar offsets and values, one
      // gather address computation per row as above -- and a lane takes part in entry k only if bit k of its template's mask is
      // set (exec-masked load, multiply, add: same operands, same order as the lane's own list).  Without this the wave -- and,
      // through the chunk barrier, its whole workgroup -- fell to the per-lane walk below (LDS look-ups per entry, ~4x the
      // instructions): half of the waves of a 256-wide grid line contain such a row.
      const int ts = tstart[tsub], te = tstart[tsub + 1];
      unsigned  rb[RPT], mk[RPT];
#pragma unroll
      for (int rr = 0; rr < RPT; rr++) {
        rb[rr] = (unsigned)(base + t + rr * 256) * 8u;
        mk[rr] = s_mask[id[rr]];
      }
#pragma unroll 4
      for (int k = ts; k < te; k++) {
        const double   a   = tval[k];
        const int      o   = toff[k];
        const unsigned bit = 1u << (k - ts);
        const char    *xb  = reinterpret_cast<const char *>(x + o);
        double         xv[RPT];
#pragma unroll
        for (int rr = 0; rr < RPT; rr++) xv[rr] = (mk[rr] & bit) ? *reinterpret_cast<const double *>(xb + rb[rr]) : 0.0;
#pragma unroll
        for (int rr = 0; rr < RPT; rr++)
          if (mk[rr] & bit) sum[rr] += a * xv[rr];
        if (DOT && o == 0) {  // (every template holds the diagonal: checked when the masks were built)
#pragma unroll
          for (int rr = 0; rr < RPT; rr++) xrow[rr] = xv[rr];
        }
      }
    } else {
      if (DOT && UNI) {
#pragma unroll
        for (int rr = 0; rr < RPT; rr++) {
          const hipx_int row = base + t + rr * 256;
          if (row < m) xrow[rr] = x[row];
        }
      }
      int s0[RPT], len[RPT], maxlen = 0;
#pragma unroll
      for (int rr = 0; rr < RPT; rr++) {
        const hipx_int row = base + t + rr * 256;
        s0[rr]  = 0;
        len[rr] = 0;
        if (row < m) {
          s0[rr]  = s_start[id[rr]];
          len[rr] = s_start[id[rr] + 1] - s0[rr];
        }
        maxlen = max(maxlen, len[rr]);
      }
      for (int k = 0; k < maxlen; k += W) {
        double xv[RPT][W], av[RPT][W];
#pragma unroll
        for (int rr = 0; rr < RPT; rr++) {
          const hipx_int row = base + t + rr * 256;
#pragma unroll
          for (int e = 0; e < W; e++) {
            const bool on  = (k + e) < len[rr];
            const int  idx = on ? s0[rr] + k + e : 0;
            av[rr][e]      = s_val[idx];
            xv[rr][e]      = on ? x[row + s_off[idx]] : 0.0;
          }
        }
#pragma unroll
        for (int rr = 0; rr < RPT; rr++) {
#pragma unroll
          for (int e = 0; e < W; e++)
            if ((k + e) < len[rr]) sum[rr] += av[rr][e] * xv[rr][e];
        }
      }
    }
    double cdot = 0.0;
#pragma unroll
    for (int rr = 0; rr < RPT; rr++) {
      const hipx_int row = base + t + rr * 256;
      if (row < m) {
        if (PROBE != 1) yout[row] = sum[rr];
        if (DOT || PROBE == 1) cdot += xrow[rr] * sum[rr];
      }
    }
    if (PROBE == 1 && !DOT && cdot == 1.2345e300) yout[0] = cdot;  // keeps the sums alive
    if (DOT) {  // one partial per wave and CHUNK (not per workgroup: which workgroup gets which chunk changes from run to run,
                // the fused dot must not): the fold reads them in chunk order
      const double w = hipx::wave_sum(cdot);
      if ((threadIdx.x & 63) == 0) dotpart[(size_t)c * 4 + (threadIdx.x >> 6)] = w;
    }
    sink += (__double_as_longlong(pf) == 0x7ff8123456789abcLL) ? 1u : 0u;  // consume the previous pass's prefetch (no wait: see above)
    if (PROBE < 5 && pf_off && t < 16 * RPT) {
      const long long prow = (long long)base + pf_off + (long long)t * 16;  // pf_off = farthest forward offset + the look-ahead distance
      if (prow < (long long)m) pf = x[prow];
    }
    if (STATICQ) {
      tk  = tk1;
      tk1 = tk1 + bpx;
    } else {
      __syncthreads();  // everybody has read the tickets
      if (t == 0) s_tk = ticket_collect(nxt_raw) - tbase;
      __syncthreads();
      tk  = tk1;
      tk1 = s_tk;
    }
  }
  (void)mydot;
  if (sink == 0xffffffffu) yout[0] = pf;  // never true: keeps the prefetch loads alive
}

// lane i <- lane i - 1 (lane 0 gets `first`): DPP wavefront shift, no LDS
__device__ __forceinline__ double pair_prev_lane(double first, double v)
{
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(first), __double2loint(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(first), __double2hiint(v), 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// lane i <- lane i + 1 (lane 63 gets `last`): through the LDS crossbar (ds_bpermute: no memory access, no bank conflicts)
__device__ __forceinline__ double pair_next_lane(double last, double v, int lane)
{
  const int src = ((lane + 1) & 63) << 2;
  const int lo  = __builtin_amdgcn_ds_bpermute(src, __double2loint(v));
  const int hi  = __builtin_amdgcn_ds_bpermute(src, __double2hiint(v));
  return lane == 63 ? last : __hiloint2double(hi, lo);
}

// Template SpMV, pair form (the sub-template form's matrices: every row = the base template with entries left out; base templates with
// <= 8 even-offset pairs: the 5-/7-point class).  A thread owns the two CONSECUTIVE rows r, r + 1 (r even) and loads x in aligned
// 16-byte pairs (x[r + e], x[r + e + 1]) for the EVEN offsets e only; the entries at e - 1 and e + 1 take their operands from the
// neighbouring lanes' pairs (one DPP shift / one ds_bpermute; the wave's first / last lane from a scalar load of the element just
// outside the wave's 128-element run).  7-point: 5 pair loads per 128 rows instead of 14 gathers; y leaves as one 16-byte store, the
// two template ids arrive as one 2-byte load.  Arithmetic per row: the row's own entries in ascending column order, product and sum
// rounded separately -- the bits of MatMult_SeqAIJ (aij.c:1486-1494).  Whole chunks of 512 rows only (the rest:
// spmv_tmpl_tail_kernel); chunk queue, XCD slabs and first-touch prefetch as in spmv_tmpl_kernel.  Measured (7-pt 256^3, same-box
// A/B against spmv_tmpl_kernel with the sub-template walk): -11 %, 0 %, -15 % on three boxes (DESIGN 3.1 lists what else was tried on
// this kernel in round 3 and did not move it: the 27-point class with 16 pairs (register pressure: slower), several ticket counters per
// XCD, a non-persistent one-chunk-per-workgroup launch, a fifth wave that does the prefetching).
// One chunk of 512 rows (rows c * 512 ... + 511) in the pair form: thread t owns rows r = c * 512 + 2 t and r + 1; id2 = the two rows'
// template ids.  Shared by the two persistent kernels below.
template <int MODE, bool DOT, int NP, int EPI = 0>
__device__ __forceinline__ void pair_chunk(const hipx_int m, const hipxPairPlan &plan, const double *__restrict__ x, const double *yin, double *yout, double *dotpart,
                                           const unsigned int *s_mask, const int *s_pe, const int *s_ph, const hipx_int c, const int t, const int lane, const int wv, const unsigned id2,
                                           const hipxPairEpi &epi = hipxPairEpi{})
{
  typedef double dbl2 __attribute__((ext_vector_type(2)));
  const long long base = (long long)c * 512;
  const long long r    = base + 2 * t;     // this thread's even row
  const long long W    = base + 128 * wv;   // first row of this wave's run
  dbl2            eb = dbl2{0.0, 0.0}, ep = dbl2{0.0, 0.0}, ed = dbl2{1.0, 1.0};
  if (EPI == 1) {  // the epilogue's own streams, issued with the pairs
    eb = *reinterpret_cast<const dbl2 *>(epi.b + r);
    ep = *reinterpret_cast<const dbl2 *>(epi.pprev + r);
    if (epi.dinv) ed = *reinterpret_cast<const dbl2 *>(epi.dinv + r);
  }
  // (1) every pair of the chunk in flight at once (zero outside the vector: such a pair is used by no row that exists)
  dbl2 P[NP];
#pragma unroll
  for (int j = 0; j < NP; j++) {
    P[j] = dbl2{0.0, 0.0};
    if (j < plan.npairs) {
      const long long qp = r + plan.e[j];
      if (qp >= 0 && qp + 1 < (long long)m) P[j] = *reinterpret_cast<const dbl2 *>(x + qp);
      else if (qp >= 0 && qp < (long long)m) P[j].x = x[qp];  // (an odd row count: the vector's last element is the first half of a pair)
    }
  }
  // (2) the elements just outside the wave's run, for the entries at e - 1 (the first lane needs x[W - 1 + e]) and e + 1 (the last lane
  // needs x[W + 128 + e]).  Wave-uniform addresses, which the compiler would fetch through the SCALAR cache -- where data that is new in
  // every pass always misses, and a miss cost 1.5-5 us under this kernel's load (HIPX_TMPL_TRACE: the longest phase of a pass, and what
  // made some workgroups three times slower than others).  So: ONE vector load for all pairs, lane 2 j on pair j's left element, lane
  // 2 j + 1 on its right one (offsets per lane from a small LDS table); read back with v_readlane after the pairs have arrived.
  double edge = 0.0;
  if (plan.jodd >= 0 && lane < 2 * plan.npairs) {
    const int       pj = lane >> 1, right = lane & 1;
    const long long qe = W + s_pe[pj] + (right ? 128 : -1);
    if (((s_ph[pj] >> right) & 1) && qe >= 0 && qe < (long long)m) edge = x[qe];
  }
  dbl2 s2 = dbl2{0.0, 0.0};
  if (MODE == 1) s2 = *reinterpret_cast<const dbl2 *>(yin + r);
  double         sum0 = s2.x, sum1 = s2.y, xr0 = 0.0, xr1 = 0.0;
  const unsigned mk0 = s_mask[id2 & 0xffu], mk1 = s_mask[id2 >> 8];
  // (3) the walk: pairs in ascending offset, slots 0, 1, 2 = the base template's entries in their own order
#pragma unroll
  for (int j = 0; j < NP; j++) {
    if (j < plan.npairs) {
      if (plan.kb[j][0] >= 0) {  // entry at e - 1: row r <- x[r + e - 1] = the previous lane's second element, row r + 1 <- x[r + e]
        const double   A = pair_prev_lane(__hiloint2double(__builtin_amdgcn_readlane(__double2hiint(edge), 2 * j), __builtin_amdgcn_readlane(__double2loint(edge), 2 * j)), P[j].y), B = P[j].x;
        const unsigned bit = 1u << plan.kb[j][0];
        const double   a = plan.a[j][0];
        if (mk0 & bit) sum0 += a * A;
        if (mk1 & bit) sum1 += a * B;
      }
      if (plan.kb[j][1] >= 0) {  // entry at e
        const unsigned bit = 1u << plan.kb[j][1];
        const double   a = plan.a[j][1];
        if (mk0 & bit) sum0 += a * P[j].x;
        if (mk1 & bit) sum1 += a * P[j].y;
      }
      if (plan.kb[j][2] >= 0) {  // entry at e + 1: row r <- x[r + e + 1], row r + 1 <- x[r + e + 2] = the next lane's first element
        const double   A = P[j].y, B = pair_next_lane(__hiloint2double(__builtin_amdgcn_readlane(__double2hiint(edge), 2 * j + 1), __builtin_amdgcn_readlane(__double2loint(edge), 2 * j + 1)), P[j].x, lane);
        const unsigned bit = 1u << plan.kb[j][2];
        const double   a = plan.a[j][2];
        if (mk0 & bit) sum0 += a * A;
        if (mk1 & bit) sum1 += a * B;
      }
      if ((DOT || EPI) && j == plan.jdiag) {
        xr0 = P[j].x;
        xr1 = P[j].y;
      }
    }
  }
  if (EPI == 1) {
    const double r0 = eb.x - sum0, r1 = eb.y - sum1;
    const double z0 = epi.dinv ? r0 * ed.x : r0, z1 = epi.dinv ? r1 * ed.y : r1;
    const double a = epi.alpha, b = epi.beta, g = epi.gamma;
    if (epi.br == 0) {
      sum0 = ep.x + b * xr0 + g * z0;
      sum1 = ep.y + b * xr1 + g * z1;
    } else if (epi.br == 1) {
      sum0 = a * ep.x + b * xr0 + z0;
      sum1 = a * ep.y + b * xr1 + z1;
    } else if (epi.br == 2) {
      sum0 = a * ep.x + b * xr0;
      sum1 = a * ep.y + b * xr1;
    } else {
      sum0 = a * ep.x + b * xr0 + g * z0;
      sum1 = a * ep.y + b * xr1 + g * z1;
    }
  }
  *reinterpret_cast<dbl2 *>(yout + r) = dbl2{sum0, sum1};
  if (DOT) {  // one partial per wave and CHUNK, folded in chunk order by the caller (as spmv_tmpl_kernel)
    const double w = hipx::wave_sum(xr0 * sum0 + xr1 * sum1);
    if (lane == 0) dotpart[(size_t)c * 4 + wv] = w;
  }
}

template <int MODE, bool DOT, int NP, bool TRACE = false, int EPI = 0>
__global__ __launch_bounds__(256) void spmv_pair_kernel(hipx_int m, hipx_int nchunks, hipx_int chunks_per_xcd, const unsigned char *__restrict__ tid, const unsigned int *__restrict__ tmask,
                                                        int ntmpl, const hipxPairPlan plan, const double *__restrict__ x, const double *yin, double *yout, double *dotpart,
                                                        unsigned long long *tq, unsigned long long launch, long long pf_off, int tg, unsigned long long *trace = nullptr, const hipxPairEpi epi = hipxPairEpi{})
{
  __shared__ unsigned int s_mask[256];
  __shared__ int          s_pe[16], s_ph[16];  // per pair: its offset e; bit 0 / 1: it has an entry at e - 1 / e + 1 (the edge load's per-lane table)
  __shared__ long long    s_tk;
  __shared__ long long    s_tk2[2];
  const int t = threadIdx.x, lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  for (int k = t; k < ntmpl; k += 256) s_mask[k] = tmask[k];
  if (t < 16) {
    s_pe[t] = plan.e[t];
    s_ph[t] = (t < plan.npairs) ? ((plan.kb[t][0] >= 0 ? 1 : 0) | (plan.kb[t][2] >= 0 ? 2 : 0)) : 0;
  }
  const hipx_int bid = (hipx_int)blockIdx.x, xcd = bid & 7, bpx = (hipx_int)gridDim.x >> 3;
  const hipx_int c0 = xcd * chunks_per_xcd, c1 = (c0 + chunks_per_xcd < nchunks) ? c0 + chunks_per_xcd : nchunks;
  const long long     nall  = (long long)(c1 > c0 ? c1 - c0 : 0);
  // one ticket = tg consecutive chunks: the returning atomic on the XCD's counter takes 1.6-2.3 us under this kernel's load, and because a
  // wave's memory operations retire in order it holds up whatever that wave loads next -- drawn once per tg passes it is amortised,
  // and the workgroup's waves only meet at a barrier when the ticket changes (between the chunks of a ticket they run free)
  const long long     nloc  = (nall + tg - 1) / tg;  // tickets of this XCD's slab
  unsigned long long *ctr   = tq + (size_t)xcd * 8;
  const long long     tbase = (long long)(launch * (unsigned long long)(nloc + 2 * bpx));
  if (t == 0) {
    s_tk2[0] = (long long)atomicAdd(ctr, 1ull) - tbase;
    s_tk2[1] = (long long)atomicAdd(ctr, 1ull) - tbase;
  }
  __syncthreads();
  long long tk = s_tk2[0], tk1 = s_tk2[1];
  double    pf = 0.0;
  unsigned  sink = 0;
  const unsigned short *tid2 = reinterpret_cast<const unsigned short *>(tid);  // rows r, r + 1: one 2-byte load (r even)
  unsigned idn = (tk < nloc) ? (unsigned)tid2[((long long)(c0 + tk * tg) * 512 + 2 * t) >> 1] : 0u;
  int                sub = 0;      // chunk of the current ticket
  unsigned long long nxt_raw = 0;  // (thread 0) the ticket after next, in flight
  unsigned ntr = 0;  // TRACE (HIPX_TMPL_TRACE: developer timing of one workgroup's passes; 100 MHz wall clock)
  unsigned long long npass = 0, t_first = TRACE ? wall_clock64() : 0;
  while (tk < nloc) {
    unsigned long long ts[6];
    if (TRACE) ts[0] = wall_clock64();
    if (t == 0 && sub == 0) ticket_draw(ctr, nxt_raw);  // the ticket after next travels while this ticket's chunks are processed
    const long long ci   = tk * tg + sub;  // (< nall: the loop's last statement sees to it)
    const hipx_int  c    = c0 + (hipx_int)ci;
    const long long base = (long long)c * 512;
    const unsigned  id2  = idn;
    {  // the next chunk's ids (this ticket's next chunk, else the first chunk of the next ticket): consumed at the top of the next pass
      const bool      same = sub + 1 < tg && ci + 1 < nall;
      const long long cn   = same ? ci + 1 : tk1 * tg;
      idn = (same || tk1 < nloc) ? (unsigned)tid2[((long long)(c0 + cn) * 512 + 2 * t) >> 1] : 0u;
    }
    pair_chunk<MODE, DOT, NP, EPI>(m, plan, x, yin, yout, dotpart, s_mask, s_pe, s_ph, c, t, lane, wv, id2, epi);
    if (TRACE) ts[1] = ts[2] = wall_clock64();
    sink += (__double_as_longlong(pf) == 0x7ff8123456789abcLL) ? 1u : 0u;  // consume the previous pass's prefetch
    if (pf_off && t < 32) {
      const long long prow = base + pf_off + (long long)t * 16;
      if (prow < (long long)m) pf = x[prow];
    }
    if (TRACE) ts[3] = wall_clock64();  // store, dot partial and prefetch issued
    npass++;
    sub++;
    const bool advance = sub == tg || ci + 1 >= nall;  // (wave-uniform, workgroup-uniform)
    if (advance) {
      __syncthreads();  // everybody has read the tickets
      if (TRACE) ts[4] = wall_clock64();
      if (t == 0) s_tk = ticket_collect(nxt_raw) - tbase;
      __syncthreads();
    }
    if (TRACE) {
      if (!advance) ts[4] = wall_clock64();
      ts[5] = wall_clock64();
      if (trace && (bid == 8 || bid == 1032) && t == 0 && ntr < 64) {
        unsigned long long *o = trace + ((bid == 8 ? 0 : 64) + ntr) * 8;
        for (int q = 0; q < 6; q++) o[q] = ts[q];
        o[6] = (unsigned long long)c;
        ntr++;
      }
    }
    if (advance) {
      tk  = tk1;
      tk1 = s_tk;
      sub = 0;
    }
  }
  if (TRACE && trace && t == 0) {  // per workgroup: passes done, first and last stamp
    unsigned long long *o = trace + 128 * 8 + (size_t)bid * 4;
    o[0] = npass;
    o[1] = t_first;
    o[2] = wall_clock64();
  }
  if (sink == 0xffffffffu) yout[0] = pf;  // never true: keeps the prefetch loads alive
}

// Template SpMV, march form (hipxMarchPlan; sub-template matrices whose base template spans three planes S rows apart: the 3-D stencils in natural
// ordering, S = n^2; 2-D ones with S = n).  The pair form fetches a row's operands entry by entry: 5 pair loads per two rows of the 7-point
// operator, every one an L2 hit but every one a trip through the CU's L1-L2 request path -- 48 bytes per row against the 17 the format moves
// from HBM -- and that path is what bounds it (round 3: every latency measure left it where it was, doubling the operand loads doubled its time).
// Here a workgroup owns rows i0 ... i0 + L - 1 of EVERY plane of its segment and marches through the planes with three of them resident in
// LDS (L + 2 H elements each: H = the largest in-plane offset): per plane step it loads the next plane's L + 2 H elements ONCE (16-byte
// loads, a whole step ahead of their use, through registers), and every entry's operand is an LDS read.  (L + 2 H) / L * 8 + 8 + 1 bytes per
// row through the L2 = 19-20.  Static work split (tiles x segments of planes, all resident at once: no tickets); segments start with two extra
// plane loads.  Arithmetic per row: the base template's entries in ascending column order, entries a row does not have skipped (template
// mask), product and sum rounded separately -- the bits of MatMult_SeqAIJ (aij.c:1486-1494).  The dot partial is per workgroup and wave
// (static split: deterministic).
// NEMAX: the entries the loops are unrolled for; EXACT: the base template has exactly NEMAX entries (5 / 7 / 9 / 27: no test per entry -- with one,
// every entry is its own basic block and waits for its own LDS reads)
template <int MODE, bool DOT, int NEMAX, bool EXACT = false, bool TRACE = false>
__global__ __launch_bounds__(256, 2) void spmv_march_kernel(hipx_int m, const hipxMarchPlan plan, const unsigned char *__restrict__ tid, const unsigned int *__restrict__ tmask, int ntmpl,
                                                         const double *__restrict__ x, const double *yin, double *yout, double *dotpart, int tiles, int nseg, int pps, int nplanes, int xcdmap,
                                                         unsigned long long *trace = nullptr)
{
  typedef double dbl2 __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *sb = reinterpret_cast<double *>(smem);  // three plane buffers of W doubles
  __shared__ unsigned int s_mask[256];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int S = plan.S, H = plan.H, L = plan.L, W = L + 2 * H, W2 = W >> 1;
  for (int k = t; k < ntmpl; k += 256) s_mask[k] = tmask[k];
  const int units = tiles * nseg;
  int       u = (int)blockIdx.x;
  if (xcdmap) u = ((int)blockIdx.x & 7) * (units >> 3) + ((int)blockIdx.x >> 3);  // an XCD's workgroups: neighbouring tiles (they share halos through its L2)
  const int tj = u % tiles, seg = u / tiles;
  const int i0 = tj * L, Lw = (S - i0 < L) ? S - i0 : L;
  const int k0 = seg * pps, k1 = (k0 + pps < nplanes) ? k0 + pps : nplanes;
  double    acc = 0.0;
  // the base template's values and in-plane offsets in VECTOR registers (every lane the same value): as scalars they do not fit (27 entries =
  // 81 SGPRs), the compiler re-fetches them with scalar loads, and a scalar load's wait (lgkmcnt(0): SMEM returns out of order) also
  // drains every LDS read in flight -- the entry loop below then runs one LDS latency per entry.  This kernel has registers to spare
  // (LDS limits it to two workgroups per CU).
  double av[NEMAX];
  int    bv2[(NEMAX + 1) / 2];  // two 16-bit offsets per register (|b| <= H <= 1040)
#pragma unroll
  for (int e = 0; e < NEMAX; e++) {
    int lo = 0, hi = 0;
    if (EXACT || e < plan.ne) {
      lo = __double2loint(plan.a[e]);
      hi = __double2hiint(plan.a[e]);
    }
    int vlo, vhi;
    asm volatile("v_mov_b32 %0, %1" : "=v"(vlo) : "s"(lo));
    asm volatile("v_mov_b32 %0, %1" : "=v"(vhi) : "s"(hi));
    av[e] = __hiloint2double(vhi, vlo);
  }
#pragma unroll
  for (int e = 0; e < NEMAX; e += 2) {
    const int b0 = (EXACT || e < plan.ne) ? plan.b[e] : 0, b1 = (e + 1 < NEMAX && (EXACT || e + 1 < plan.ne)) ? plan.b[e + 1 < 32 ? e + 1 : 31] : 0;
    const int pk = (b0 & 0xffff) | (b1 << 16);
    int       vb;
    asm volatile("v_mov_b32 %0, %1" : "=v"(vb) : "s"(pk));
    bv2[e >> 1] = vb;
  }
  // Every memory instruction of the plane loop is issued UNCONDITIONALLY (clamped addresses, the value selected afterwards) and in a fixed
  // order per step -- y of the plane before, the next plane's template ids, then the plane loads: a wave's memory operations retire in
  // order, so a wait for one load is a wait for everything issued before it, and the compiler can only count what is certain to have
  // been issued after it.  With this order the wait on plane k + 1 (issued two steps before) leaves the loads of plane k + 2 in flight.
  constexpr int NLD = 7;  // 16-byte loads per thread and plane: W <= 3584
  constexpr int NQ  = 8;  // rows per thread and plane step (L <= 2048)
  dbl2          R0[NLD], R1[NLD];             // planes in flight: two steps ahead of their use
  unsigned char id0[NQ], id1[NQ];             // template ids of this thread's rows, one step ahead
  const auto load_plane = [&](int p, dbl2 (&R)[NLD]) {
    const long long g0 = (long long)p * S + i0 - H;  // even: S, L, H are (m too: the launcher sees to it)
#pragma unroll
    for (int q = 0; q < NLD; q++) {
      const int       idx = q * 256 + t;
      const long long g   = g0 + 2 * (long long)(idx < W2 ? idx : W2 - 1);
      const bool      in  = g >= 0 && g < (long long)m;
      const dbl2      v   = *reinterpret_cast<const dbl2 *>(x + (in ? g : 0));
      R[q].x = in ? v.x : 0.0;
      R[q].y = in ? v.y : 0.0;
    }
  };
  const auto store_plane = [&](int slot, const dbl2 (&R)[NLD]) {
    dbl2 *d = reinterpret_cast<dbl2 *>(sb + (size_t)slot * W);
#pragma unroll
    for (int q = 0; q < NLD; q++) {
      const int idx = q * 256 + t;
      if (idx < W2) d[idx] = R[q];
    }
  };
  // thread t's row of group q (rows q * 256 ... + 255 of the tile): rotated by 32 + 64 (q / 2).  The rows that lack entries come in adjacent
  // pairs (last point of a grid line, first point of the next); unrotated they would sit in the same one or two waves for every group (256-
  // point lines: lanes 0 and 255, waves 0 and 3 in all eight groups), and a wave with such a row takes the select path below, twice the
  // instructions -- the whole workgroup then runs at that wave's pace.  Rotated, each wave meets them in one pair of groups out of four.
  int pq[NQ / 2];
#pragma unroll
  for (int j = 0; j < NQ / 2; j++) pq[j] = (t + 32 + 64 * j) & 255;
  const auto load_ids = [&](int p, unsigned char (&ids)[NQ]) {
    const long long rb = (long long)p * S + i0;
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const int i = q * 256 + pq[q >> 1];
      ids[q] = tid[(i < Lw && rb + i < (long long)m) ? rb + i : 0];
    }
  };
  double    sprev[NQ];
  unsigned  mprev[NQ];
  long long rbprev = 0;
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    sprev[q] = 0.0;
    mprev[q] = 0u;
  }
  int s_lo = 0, s_mid = 1, s_hi = 2;
  load_ids(k0, id0);
  load_plane(k0 - 1, R0);
  load_plane(k0, R1);
  store_plane(s_lo, R0);
  store_plane(s_mid, R1);
  load_plane(k0 + 1, R0);
  load_plane(k0 + 2, R1);
  __syncthreads();
  // one plane step: RA holds plane k + 1 (loaded two steps ago), idc the ids of plane k; plane k + 3 goes into RA, the ids of plane k + 1 into idn
  const auto step = [&](int k, dbl2 (&RA)[NLD], const unsigned char (&idc)[NQ], unsigned char (&idn)[NQ]) {
    unsigned long long ts[6];
    if (TRACE) ts[0] = wall_clock64();
    store_plane(s_hi, RA);
    if (TRACE) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      ts[1] = wall_clock64();
    }
    __syncthreads();
    if (TRACE) ts[2] = wall_clock64();
    if (k > k0) {  // y of the plane before: the stores have this whole step to complete
#pragma unroll
      for (int q = 0; q < NQ; q++)
        if (mprev[q]) yout[rbprev + q * 256 + pq[q >> 1]] = sprev[q];
    }
    load_ids(k + 1, idn);
    load_plane(k + 3, RA);
    if (TRACE) ts[3] = wall_clock64();
    const long long rowbase = (long long)k * S + i0;
    const double   *plo = sb + (size_t)s_lo * W + H, *pmid = sb + (size_t)s_mid * W + H, *phi = sb + (size_t)s_hi * W + H;
    double          sum[NQ];
    unsigned        mk[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const int       i = q * 256 + pq[q >> 1];
      const long long row = rowbase + i;
      const bool      valid = i < Lw && row < (long long)m;
      mk[q]  = valid ? s_mask[idc[q]] : 0u;
      sum[q] = (MODE == 1 && valid) ? yin[row] : 0.0;
    }
    // two groups (128 rows of this wave) at a time, entry by entry: 2 independent sums, every LDS read issued unconditionally.  All 128 rows
    // interior rows: product and sum per entry.  Otherwise the select form: a row that lacks the entry keeps its sum (the product is formed
    // and dropped -- the bits of skipping it); the asm statement pins the read outside any branch (left alone the compiler turns the select
    // back into a branch around read + multiply + add per row and entry: one LDS latency each, measured 3.8 us per step for 7 entries).
#pragma unroll
    for (int j = 0; j < NQ / 2; j++) {
      if (2 * j * 256 < L) {  // (L = 1024: the buffers end after four groups)
        const int  q0 = 2 * j, q1 = 2 * j + 1;
        const bool uni = __builtin_amdgcn_ballot_w64(((mk[q0] ^ plan.full) | (mk[q1] ^ plan.full)) != 0u) == 0ull;
        // entries in batches of EC: the batch's 2 EC LDS reads are issued together (the asm statements pin them there: left alone the compiler
        // reuses one address and one destination register for every read, i.e. one LDS latency per entry), then the sums.  EC = 8, or 4 for
        // the long templates (their 27 values and offsets already fill the register file: 8 spilled to scratch and cost more than it gained)
        constexpr int EC = (NEMAX > 8) ? 4 : 8;
#pragma unroll
        for (int c0 = 0; c0 < NEMAX; c0 += EC) {
          double xa[EC], xb[EC];
#pragma unroll
          for (int w = 0; w < EC; w++) {
            const int e = c0 + w;
            xa[w] = xb[w] = 0.0;
            if (e < NEMAX && (EXACT || e < plan.ne)) {
              const double *pb = ((e < plan.nlo) ? plo : ((e < plan.nlo + plan.nmid) ? pmid : phi)) + ((e & 1) ? (bv2[e >> 1] >> 16) : ((bv2[e >> 1] << 16) >> 16)) + pq[j];
              xa[w] = pb[q0 * 256];
              xb[w] = pb[q1 * 256];
            }
          }
          if constexpr (EC == 8) {
            asm volatile("" : "+v"(xa[0]), "+v"(xa[1]), "+v"(xa[2]), "+v"(xa[3]), "+v"(xa[4]), "+v"(xa[5]), "+v"(xa[6]), "+v"(xa[7]) : : "memory");
            asm volatile("" : "+v"(xb[0]), "+v"(xb[1]), "+v"(xb[2]), "+v"(xb[3]), "+v"(xb[4]), "+v"(xb[5]), "+v"(xb[6]), "+v"(xb[7]));
          } else {
            asm volatile("" : "+v"(xa[0]), "+v"(xa[1]), "+v"(xa[2]), "+v"(xa[3]), "+v"(xb[0]), "+v"(xb[1]), "+v"(xb[2]), "+v"(xb[3]) : : "memory");
          }
          if (uni) {
#pragma unroll
            for (int w = 0; w < EC; w++) {
              const int e = c0 + w;
              if (e < NEMAX && (EXACT || e < plan.ne)) {
                sum[q0] += av[e] * xa[w];
                sum[q1] += av[e] * xb[w];
              }
            }
          } else {
#pragma unroll
            for (int w = 0; w < EC; w++) {
              const int e = c0 + w;
              if (e < NEMAX && (EXACT || e < plan.ne)) {
                const double t0 = sum[q0] + av[e] * xa[w], t1 = sum[q1] + av[e] * xb[w];
                sum[q0] = ((mk[q0] >> e) & 1u) ? t0 : sum[q0];
                sum[q1] = ((mk[q1] >> e) & 1u) ? t1 : sum[q1];
              }
            }
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      if (DOT && mk[q]) acc += pmid[q * 256 + pq[q >> 1]] * sum[q];  // (every existing row has its diagonal entry: mk != 0)
      sprev[q] = sum[q];
      mprev[q] = mk[q];
    }
    rbprev = rowbase;
    if (TRACE) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      ts[4] = wall_clock64();
    }
    __syncthreads();
    if (TRACE) {
      ts[5] = wall_clock64();
      if (trace && t == 0 && (blockIdx.x == 8 || blockIdx.x == 301) && k - k0 < 40) {
        unsigned long long *o = trace + ((blockIdx.x == 8 ? 0 : 40) + (k - k0)) * 8;
        for (int q = 0; q < 6; q++) o[q] = ts[q];
        o[6] = (unsigned long long)k;
      }
    }
    const int o = s_lo;
    s_lo  = s_mid;
    s_mid = s_hi;
    s_hi  = o;
  };
  for (int k = k0; k < k1; k += 2) {
    step(k, R0, id0, id1);
    if (k + 1 < k1) step(k + 1, R1, id1, id0);
  }
#pragma unroll
  for (int q = 0; q < NQ; q++)
    if (mprev[q]) yout[rbprev + q * 256 + pq[q >> 1]] = sprev[q];
  if (DOT) {
    const double w = hipx::wave_sum(acc);
    if (lane == 0) dotpart[(size_t)blockIdx.x * 4 + wv] = w;
  }
}

// ---- march form, second generation (round 4): spmv_march2_kernel.  Same idea and same arithmetic as spmv_march_kernel -- a workgroup owns rows
// i0 ... i0 + L - 1 of every plane of its segment, three planes of x (+ halos) resident in LDS, every operand an LDS read, the base template's
// entries in ascending column order with product and sum rounded separately (the bits of MatMult_SeqAIJ, aij.c:1486-1494) -- but written for
// the instruction budget: round 3's kernel issues ~410 vector instructions per wave and plane step for the 112 multiplies and adds of the
// 7-point operator (64-bit bounds-checked addresses for every load and store, template ids and mask look-ups per row and step, per-entry LDS
// address arithmetic, SGPR spills: it is VALU-issue-bound at 0.5 of the HBM peak, profiles/r03_tmpl_sq_counters.txt).  What this kernel
// assumes (checked when the plan is built, hipxMatMarch2Check; anything else keeps spmv_march_kernel):
//   * whole planes and whole tiles: m = nplanes * S, S a multiple of L: every row of every step is a row of the matrix -- no per-row validity;
//   * the template ids are PLANE-PERIODIC: row k S + i has the template of row S + i for every interior plane k (1 <= k <= nplanes - 2) -- a
//     device pass compares them at set-up.  The masks of a thread's rows are then loop invariants: loaded once; only the first and the
//     last plane of the grid look theirs up (a uniform branch);
//   * the base template's entries form RUNS of consecutive in-plane offsets (5: 1-3-1, 7: 1-1-3-1-1, 9: 3-3-3, 27: nine runs of 3): one LDS
//     address per run, pair of row groups and step; the operands of a run are ds_read_b64 at immediate offsets from it (256 B/clk, twice
//     the rate of the ds_read2st64_b64 pairs the first kernel used, MI355X_MICROARCH LDS table);
//   * a plane's 16-byte loads are split into the thread's OWN rows (L / 512 per thread, uniform base + a per-thread constant) and the halo
//     (H / 256 per thread, rounded up): no clamping per load; only the first tile of plane 0 and the last tile of the last plane redirect
//     their out-of-range halo loads (the rows that would use those operands lack the entries: the products are formed and dropped).
// Rows a wave cannot treat uniformly (some lane's row lacks an entry) add under the row's mask; the thread-to-row rotation of the first
// kernel (32 + 64 j) keeps those in one pair of row groups per wave for 256-point lines.
template <int NE>
struct MarchRuns;
template <>
struct MarchRuns<5> {
  static constexpr int NR = 3, NLO = 1, NMID = 3;
  static __host__ __device__ constexpr int st(int r) { return r == 0 ? 0 : (r == 1 ? 1 : 4); }
  static __host__ __device__ constexpr int ln(int r) { return r == 1 ? 3 : 1; }
};
template <>
struct MarchRuns<7> {
  static constexpr int NR = 5, NLO = 1, NMID = 5;
  static __host__ __device__ constexpr int st(int r) { return r == 0 ? 0 : (r == 1 ? 1 : (r == 2 ? 2 : (r == 3 ? 5 : 6))); }
  static __host__ __device__ constexpr int ln(int r) { return r == 2 ? 3 : 1; }
};
template <>
struct MarchRuns<9> {
  static constexpr int NR = 3, NLO = 3, NMID = 3;
  static __host__ __device__ constexpr int st(int r) { return 3 * r; }
  static __host__ __device__ constexpr int ln(int) { return 3; }
};
template <>
struct MarchRuns<27> {
  static constexpr int NR = 9, NLO = 9, NMID = 9;
  static __host__ __device__ constexpr int st(int r) { return 3 * r; }
  static __host__ __device__ constexpr int ln(int) { return 3; }
};

// CG = true: the CG direction update as the kernel's PROLOGUE (round 4; cg.c:248-249 and the x update of cg.c:305 left over from the iteration
// before, i.e. hipxCGAypxAxpy...): x[] is the OLD direction p; while a plane (with its halos) passes from registers into LDS the kernel forms
//     p_new = (z * dconst) + b p        z: the preconditioned residual, or the residual itself with the constant Jacobi diagonal / PCNONE (dconst)
// element by element -- the operations, operands and order of cg_aypx_axpy_kernel: the same bits, in the halo as in the rows of their owner --
// multiplies by the matrix out of LDS as before (y = A p_new, dot = p_new . y), and for the rows it owns also stores p_new (to a SECOND
// direction vector: other workgroups still read p in their halos) and x += a p.  One launch and one pass over p less per iteration:
// A(i) + B(i) read p, z, x, p(1.1x), ids and write p, x, y; this kernel reads p(1.1x), z(1.1x), x and writes p_new, x, y.
struct hipxMarchCG {
  const double *z;                                   // z (dconst = 1) or r (dconst = the constant inverse diagonal)
  double       *pnew, *xsol;
  double        dconst, b, a;                        // b, a: used when dev_beta_new == NULL
  const double *dev_beta_new, *dev_beta_old, *dev_dpi;  // device-resident sums of the kernels queued before: b = beta_new / beta_old, a = beta_old / dpi
};

// NT = 512 (round 4, lines of up to 1024 points -- BASELINE config 5's 1024 x 1024 planes): eight waves share three plane buffers of 4096 + 2 H
// doubles (144 KiB: one workgroup per CU, the same two waves per SIMD); thread t works in the half t / 256 of the tile.
template <int NE, int NQ, int NHALO, bool DOT, bool CG = false, int NT = 256>
__global__ __launch_bounds__(NT, (NT == 256 || NQ <= 4) ? 2 : 1) void spmv_march2_kernel(const hipxMarchPlan plan, const unsigned char *__restrict__ tid, const unsigned int *__restrict__ tmask, const int ntmpl,
                                                              const double *__restrict__ x, double *__restrict__ yout, double *__restrict__ dotpart, const int tiles, const int pps,
                                                              const int nplanes, const int xcdmap, const hipxMarchCG cg = hipxMarchCG{}, const RedOut red = RedOut{}, const int npart_extra = 0)
{
  using RS = MarchRuns<NE>;
  typedef double dbl2 __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ unsigned int s_mask[256];
  constexpr int L = NQ * NT, NOWN = NQ / 2, NJ = NQ / 2, DIAG = NE / 2;
  constexpr int RB = (NE > 9) ? 3 : RS::NR;  // runs per operand batch (27 entries: 9 entries x 2 rows at a time, 36 registers)
  // SOLO: the operand reads as single ds_read_b64 (volatile: the compiler's load/store optimizer pairs neighbouring reads into ds_read2_b64 /
  // ds_read2st64_b64, which move 128 B/clk where two single reads move 256, MI355X_MICROARCH LDS table).  The plain product of the long
  // templates is bound by that rate (27 entries: 432 operand reads per thread and step against 432 multiplies and adds): 27-pt 256^3
  // 0.097 -> 0.090 ms, 512^3 0.697 -> 0.619 ms (run r04s, same box).  The short templates are not and keep the pairs (fewer instructions);
  // so does the CG-prologue form of the long ones (measured 4-8 % SLOWER with single reads at 512^3: it waits for memory, not for LDS)
  typedef __attribute__((address_space(3))) char lds_char;
  typedef volatile __attribute__((address_space(3))) double lds_vdbl;
#ifdef HIPX_MARCH_PAIRED_READS  // developer A/B build
  constexpr bool SOLO = false;
#else
  constexpr bool SOLO = NE > 9 && !CG;
#endif
  constexpr int AHEAD = (NE > 9) ? 1 : 2;    // planes in flight in registers: two steps ahead of their use, or one for the long templates (their
                                             // 27 values fill the register file, and a step is four times as long: one is ahead enough)
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int S = plan.S, H = plan.H, W = L + 2 * H, H2 = H >> 1;
  for (int k = t; k < ntmpl; k += NT) s_mask[k] = tmask[k];
  const int units = (int)gridDim.x;
  int       u     = (int)blockIdx.x;
  if (xcdmap & 1) u = ((int)blockIdx.x & 7) * (units >> 3) + ((int)blockIdx.x >> 3);  // an XCD's workgroups: neighbouring tiles (they share halos through its L2)
  const int tj = u % tiles, seg = u / tiles;
  const int i0 = tj * L;
  const int k0 = seg * pps, k1 = (k0 + pps < nplanes) ? k0 + pps : nplanes;
  // the base template's values and the runs' LDS offsets in VECTOR registers (every lane the same value): left in the kernel-argument
  // segment the compiler re-fetches them with scalar loads inside the plane loop, and a scalar load's wait drains every LDS read in flight
  double av[NE];
#pragma unroll
  for (int e = 0; e < NE; e++) {
    int vlo, vhi;
    asm volatile("v_mov_b32 %0, %1" : "=v"(vlo) : "s"(__double2loint(plan.a[e])));
    asm volatile("v_mov_b32 %0, %1" : "=v"(vhi) : "s"(__double2hiint(plan.a[e])));
    av[e] = __hiloint2double(vhi, vlo);
  }
  int rjb[NJ];  // byte offset of this thread's row inside a group of 256 rows, per pair of groups (rotated: see spmv_march_kernel)
#pragma unroll
  for (int j = 0; j < NJ; j++) rjb[j] = ((((t & 255) + 32 + 64 * j) & 255) + (t >> 8) * NQ * 256) * 8;  // (+ the thread's half of the tile: NQ groups of 256 rows)
  int cv[RS::NR];  // (H + b) * 8 of each run's first entry
#pragma unroll
  for (int r = 0; r < RS::NR; r++) {
    int v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"((H + plan.b[RS::st(r)]) * 8));
    cv[r] = v;
  }
  // halo loads of this thread: element offset from the first own row of the plane, and where the 16 bytes go inside a plane buffer
  int  he[NHALO], hl[NHALO];
  bool hact[NHALO], hlow[NHALO];
#pragma unroll
  for (int hh = 0; hh < NHALO; hh++) {
    const int hidx0 = hh * NT + t;
    hact[hh]        = hidx0 < H;
    const int hidx  = hact[hh] ? hidx0 : H - 1;
    hlow[hh]        = hidx < H2;
    he[hh]          = hlow[hh] ? 2 * hidx - H : L + 2 * (hidx - H2);
    hl[hh]          = (he[hh] + H) * 8;
  }
  // the registers of one plane in flight: its own rows and halo (p when CG), and with the CG prologue the same of z plus the own rows of x
  struct PlaneRegs {
    dbl2 o[NOWN], h[NHALO];
    dbl2 zo[CG ? NOWN : 1], zh[CG ? NHALO : 1], xo[CG ? NOWN : 1];
  };
  double cgb = 0.0, cga = 0.0, cgd = 1.0;
  if (CG) {
    cgb = cg.dev_beta_new ? (*cg.dev_beta_new / *cg.dev_beta_old) : cg.b;  // cg.c:248
    cga = cg.dev_beta_new ? (*cg.dev_beta_old / *cg.dev_dpi) : cg.a;       // cg.c:288 of the iteration before
    cgd = cg.dconst;
  }
  const auto load_plane = [&](int p, PlaneRegs &R) {
    // planes outside the grid: the rows that would use them lack those entries.  Planes beyond the segment's last halo plane (k1): nobody
    // uses them, but the loads stay (a wave's memory operations retire in order and the compiler can only count what is certain to have
    // been issued: a conditional load here turns the wait for the plane before into a wait for everything) -- they re-read plane k1 (L2 hits)
    const int       pe  = p > k1 ? k1 : p;
    const int       pc  = pe < 0 ? 0 : (pe >= nplanes ? nplanes - 1 : pe);
    const long long g0  = (long long)pc * S + i0;
    const double   *src = x + g0;
#pragma unroll
    for (int qq = 0; qq < NOWN; qq++) R.o[qq] = *reinterpret_cast<const dbl2 *>(src + 2 * (qq * NT + t));
    const bool lofix = pc == 0 && i0 < H, hifix = pc == nplanes - 1 && i0 + L + H > S;  // halo beyond the vector's ends (uniform)
    int        off[NHALO];
#pragma unroll
    for (int hh = 0; hh < NHALO; hh++) {
      off[hh] = he[hh];
      if ((lofix && hlow[hh]) || (hifix && !hlow[hh])) off[hh] = 0;
      R.h[hh] = *reinterpret_cast<const dbl2 *>(src + off[hh]);
    }
    if (CG) {
      const double *zs = cg.z + g0, *xs = cg.xsol + g0;
#pragma unroll
      for (int qq = 0; qq < NOWN; qq++) R.zo[qq] = *reinterpret_cast<const dbl2 *>(zs + 2 * (qq * NT + t));
#pragma unroll
      for (int hh = 0; hh < NHALO; hh++) R.zh[hh] = *reinterpret_cast<const dbl2 *>(zs + off[hh]);
#pragma unroll
      for (int qq = 0; qq < NOWN; qq++) {
        if (xcdmap & 4) {  // developer variant: x is read once and overwritten -- non-temporal
          const double *q = xs + 2 * (qq * NT + t);
          R.xo[qq].x      = __builtin_nontemporal_load(q);
          R.xo[qq].y      = __builtin_nontemporal_load(q + 1);
        } else R.xo[qq] = *reinterpret_cast<const dbl2 *>(xs + 2 * (qq * NT + t));
      }
    }
  };
  // plane p from registers into the buffer at sbase; CG: as p_new, and for the planes this workgroup owns (k0 <= p < k1) p_new and x += a p go to memory
  const auto store_plane = [&](int sbase, int p, const PlaneRegs &R) {
    char *d = smem + sbase;
    if (!CG) {
#pragma unroll
      for (int qq = 0; qq < NOWN; qq++) *reinterpret_cast<dbl2 *>(d + H * 8 + (qq * NT + t) * 16) = R.o[qq];
#pragma unroll
      for (int hh = 0; hh < NHALO; hh++)
        if (hact[hh]) *reinterpret_cast<dbl2 *>(d + hl[hh]) = R.h[hh];
    } else {
      const bool own = p >= k0 && p < k1;
      dbl2       pn[NOWN];
#pragma unroll
      for (int qq = 0; qq < NOWN; qq++) {
        dbl2 zv = R.zo[qq];
        zv.x    = zv.x * cgd;
        zv.y    = zv.y * cgd;
        pn[qq].x = zv.x + cgb * R.o[qq].x;
        pn[qq].y = zv.y + cgb * R.o[qq].y;
        *reinterpret_cast<dbl2 *>(d + H * 8 + (qq * NT + t) * 16) = pn[qq];
      }
#pragma unroll
      for (int hh = 0; hh < NHALO; hh++) {
        dbl2 zv = R.zh[hh], ph;
        zv.x    = zv.x * cgd;
        zv.y    = zv.y * cgd;
        ph.x    = zv.x + cgb * R.h[hh].x;
        ph.y    = zv.y + cgb * R.h[hh].y;
        if (hact[hh]) *reinterpret_cast<dbl2 *>(d + hl[hh]) = ph;
      }
      if (own) {
        const long long g0 = (long long)p * S + i0;
        double         *pd = cg.pnew + g0, *xd = cg.xsol + g0;
#pragma unroll
        for (int qq = 0; qq < NOWN; qq++) {
          dbl2 xv = R.xo[qq];
          xv.x    = xv.x + cga * R.o[qq].x;
          xv.y    = xv.y + cga * R.o[qq].y;
          if (xcdmap & 8) {  // developer variant: non-temporal stores of x (nobody reads it before the next iteration's own rows)
            __builtin_nontemporal_store(xv.x, xd + 2 * (qq * NT + t));
            __builtin_nontemporal_store(xv.y, xd + 2 * (qq * NT + t) + 1);
          } else *reinterpret_cast<dbl2 *>(xd + 2 * (qq * NT + t)) = xv;
          *reinterpret_cast<dbl2 *>(pd + 2 * (qq * NT + t)) = pn[qq];
        }
      }
    }
  };
  PlaneRegs R0, R1;  // planes in flight: AHEAD steps ahead of their use
  int       s_lo = 0, s_mid = W * 8, s_hi = 2 * W * 8;
  load_plane(k0 - 1, R0);
  load_plane(k0, R1);
  __syncthreads();  // s_mask
  // masks of this thread's rows, and which pairs of row groups a wave can treat uniformly (every lane's rows have the full template).  The
  // template ids are plane-periodic over the interior planes: the masks are looked up when the march starts, after the grid's first plane and
  // before its last one -- not per step
  unsigned mk[NQ];
  bool     uni[NJ];
  const auto load_masks = [&](int kp) {
    const long long rb = (long long)kp * S + i0;
#pragma unroll
    for (int q = 0; q < NQ; q++) mk[q] = s_mask[tid[rb + q * 256 + (rjb[q >> 1] >> 3)]];
#pragma unroll
    for (int j = 0; j < NJ; j++) uni[j] = __builtin_amdgcn_ballot_w64(((mk[2 * j] ^ plan.full) | (mk[2 * j + 1] ^ plan.full)) != 0u) == 0ull;
  };
  load_masks(k0);
  store_plane(s_lo, k0 - 1, R0);
  store_plane(s_mid, k0, R1);
  load_plane(k0 + 1, R0);
  if (AHEAD == 2) load_plane(k0 + 2, R1);
  double acc = 0.0;
  // one plane step: R holds plane k + 1 (loaded AHEAD steps ago); plane k + 1 + AHEAD goes into it
  const auto step = [&](int k, PlaneRegs &R) {
    store_plane(s_hi, k + 1, R);
    __syncthreads();
    load_plane(k + 1 + AHEAD, R);
    // (the runs' address constants are loop invariants; so are their sums with the row offsets, which the compiler would otherwise keep in
    // NR x NJ more registers across the whole march)
#pragma unroll
    for (int r = 0; r < RS::NR; r++) asm volatile("" : "+v"(cv[r]));
    if ((k == 1 && k0 == 0) || (k == nplanes - 1 && k > k0)) load_masks(k);  // leaving the grid's first plane / entering its last one (uniform, twice per launch)
    const long long rowbase = (long long)k * S + i0;
    double *yb = yout + rowbase;
    // MPIAIJ diagonal block (round 6, hipxMatMultMPICGDirectionDotBegin): the rows of the slab's first / last plane get their off-diagonal terms
    // added by offdiag_dot_kernel, which also forms their share p_i w_i of the dot with the COMPLETE w_i -- here they are left out of the sum
    const bool dskip = DOT && (((xcdmap & 16) && k == 0) || ((xcdmap & 32) && k == nplanes - 1));  // (uniform)
    // one pair of row groups: the operands of a batch of runs are read together, then the sums in ascending column order.  UNI: every row of
    // this wave's 128 has the full template -- plain multiply-add pairs; otherwise a row that lacks the entry keeps its sum (the product is
    // formed and dropped: the bits of skipping it).  The two forms are separate instantiations behind a wave-uniform branch: written as one
    // body with the test inside, the compiler folds them into the select form for every row.
    const auto pairbody = [&](const int j, auto uni_tag) {
      constexpr bool UNI = decltype(uni_tag)::value;
      const int      q0 = 2 * j, q1 = 2 * j + 1;
      double         sum0 = 0.0, sum1 = 0.0, xd0 = 0.0, xd1 = 0.0;
      if (!UNI) asm volatile("; rows with fewer entries" ::: "memory");
#pragma unroll
      for (int r0 = 0; r0 < RS::NR; r0 += RB) {
        double xa[3 * RB], xb[3 * RB];
        if (RB < RS::NR) asm volatile("" ::: "memory");  // the next batch's reads stay behind this batch's arithmetic (registers)
#pragma unroll
        for (int rr = 0; rr < RB; rr++) {
          const int r = r0 + rr;
          if (r < RS::NR) {
            const int     sb = (RS::st(r) < RS::NLO) ? s_lo : ((RS::st(r) < RS::NLO + RS::NMID) ? s_mid : s_hi);
            if (SOLO) {
              const lds_vdbl *pb = reinterpret_cast<const lds_vdbl *>((lds_char *)smem + (cv[r] + rjb[j] + sb));
#pragma unroll
              for (int d = 0; d < 3; d++)
                if (d < RS::ln(r)) xa[3 * rr + d] = pb[q0 * 256 + d];
#pragma unroll
              for (int d = 0; d < 3; d++)
                if (d < RS::ln(r)) xb[3 * rr + d] = pb[q1 * 256 + d];
            } else {
              const double *pb = reinterpret_cast<const double *>(smem + (cv[r] + rjb[j] + sb));
#pragma unroll
              for (int d = 0; d < 3; d++)
                if (d < RS::ln(r)) {
                  xa[3 * rr + d] = pb[q0 * 256 + d];
                  xb[3 * rr + d] = pb[q1 * 256 + d];
                }
            }
          }
        }
        // pin the batch's reads ahead of its arithmetic (left alone the compiler serialises read -> multiply -> add per entry: one LDS latency each)
#pragma unroll
        for (int rr = 0; rr < RB; rr++)
#pragma unroll
          for (int d = 0; d < 3; d++)
            if (r0 + rr < RS::NR && d < RS::ln(r0 + rr)) asm volatile("" : "+v"(xa[3 * rr + d]), "+v"(xb[3 * rr + d]));
#pragma unroll
        for (int rr = 0; rr < RB; rr++) {
          const int r = r0 + rr;
          if (r < RS::NR) {
#pragma unroll
            for (int d = 0; d < 3; d++)
              if (d < RS::ln(r)) {
                const int    e  = RS::st(r) + d;
                const double p0 = av[e] * xa[3 * rr + d], p1 = av[e] * xb[3 * rr + d];
                if (e == DIAG) {
                  xd0 = xa[3 * rr + d];
                  xd1 = xb[3 * rr + d];
                }
                if (UNI) {
                  sum0 += p0;
                  sum1 += p1;
                  asm volatile("" : "+v"(sum0), "+v"(sum1));  // keep the two rows' dependent add chains interleaved (the scheduler strings each row's seven adds together)
                } else {
                  const double t0 = sum0 + p0, t1 = sum1 + p1;
                  sum0 = ((mk[q0] >> e) & 1u) ? t0 : sum0;
                  sum1 = ((mk[q1] >> e) & 1u) ? t1 : sum1;
                }
              }
          }
        }
      }
      if (xcdmap & 2) {  // developer variant (HIPX_MARCH_NT_STORE): non-temporal stores of y
        __builtin_nontemporal_store(sum0, &yb[q0 * 256 + (rjb[j] >> 3)]);
        __builtin_nontemporal_store(sum1, &yb[q1 * 256 + (rjb[j] >> 3)]);
      } else {
        yb[q0 * 256 + (rjb[j] >> 3)] = sum0;
        yb[q1 * 256 + (rjb[j] >> 3)] = sum1;
      }
      if (DOT && !dskip) {  // (every row has its diagonal entry: checked with the masks)
        acc += xd0 * sum0;
        acc += xd1 * sum1;
      }
    };
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      if (uni[j]) pairbody(j, std::true_type{});
      else pairbody(j, std::false_type{});
    }
    __syncthreads();
    const int o = s_lo;
    s_lo  = s_mid;
    s_mid = s_hi;
    s_hi  = o;
  };
  if (AHEAD == 2) {
    for (int k = k0; k < k1; k += 2) {
      step(k, R0);
      if (k + 1 < k1) step(k + 1, R1);
    }
  } else {
    for (int k = k0; k < k1; k++) step(k, R0);
  }
  if (DOT) {
    const double w = hipx::wave_sum(acc);
    if (!red.ticket) {
      if (lane == 0) dotpart[(size_t)blockIdx.x * (NT / 64) + wv] = w;
    } else {
      // The fold of the partials by the LAST workgroup to finish (round 5: sum_kernel's launch, 5-8 us per CG iteration, saved).  The sum is formed in sum_kernel<false>'s own order for a one-workgroup launch (the caller passes `red` only
      // when red_grid(npart) == 1): virtual thread v of 1024 adds partials v, v + 1024, ...; wave_sum per virtual wave; the 16 wave sums left to
      // right -- so the value is bit for bit the one the separate kernel produced (real thread t plays the virtual threads t, t + NT, ...).
      // NO agent-scope release fence here: it writes back the XCD's whole L2, which this kernel has just filled with y, p and x (measured: +24 us
      // per launch).  The partial goes out as an agent-scope atomic store (sc1: written through to memory), the wave waits for it (vmcnt), the
      // barrier collects the four waves, then the ticket; the last workgroup reads the partials with agent-scope atomic loads (sc1: never a stale line).
      // ISA ASSUMPTION: this is the gfx950 "sc1 stores AND sc1 loads on both sides" hand-off of MI355X_MICROARCH (valid forms), which rests on
      // relaxed agent-scope atomics lowering to global_store/load ... sc1 (write-through / L1-bypassing) and on the ticket RMW being issued after the
      // drained store -- not on the HIP memory model.  Pinned by tests/test_gpu_mat.py::test_march2_in_kernel_fold_equals_the_separate_fold_under_load
      // (bitwise against the separate fold kernel, back-to-back launches); HIPX_MARCH_NOFOLD=1 restores the separate kernel.
      __shared__ unsigned s_lastwg;
      __shared__ double   s_fw[kRedThreads / 64];
      if (lane == 0) __hip_atomic_store(&dotpart[(size_t)blockIdx.x * (NT / 64) + wv], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t == 0) {
        const unsigned tk = __hip_atomic_fetch_add(red.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_lastwg          = (tk == gridDim.x - 1);
      }
      __syncthreads();
      if (!s_lastwg) return;
      const int npart = (int)gridDim.x * (NT / 64) + npart_extra;  // (+ the partials of spmv_march2_rem_kernel, written by the launch before this one)
      constexpr int VQ = kRedThreads / NT, VJ = 8;  // (npart <= 8 kRedThreads: at most eight partials per virtual thread -- all loads in flight at once)
      double        pv[VQ][VJ];
#pragma unroll
      for (int q = 0; q < VQ; q++)
#pragma unroll
        for (int j = 0; j < VJ; j++) {
          const int i = t + NT * q + kRedThreads * j;
          pv[q][j]    = (i < npart) ? __hip_atomic_load(&dotpart[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        }
#pragma unroll
      for (int q = 0; q < VQ; q++) {
        double a = 0.0;
#pragma unroll
        for (int j = 0; j < VJ; j++)
          if (t + NT * q + kRedThreads * j < npart) a += pv[q][j];
        a = hipx::wave_sum(a);
        if (lane == 0) s_fw[wv + (NT / 64) * q] = a;
      }
      __syncthreads();
      if (t == 0) {
        double r = s_fw[0];
#pragma unroll
        for (int k = 1; k < kRedThreads / 64; k++) r += s_fw[k];
        r = 0.0 + r;  // (sum_kernel's last-workgroup fold of its single partial: 0 + r, then additions of +0.0 only)
        red.results[0] = r;
        if (red.dres) red.dres[0] = r;
        *red.ticket = 0u;
        __threadfence_system();
        if (red.seq) __hip_atomic_store(red.flag, red.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// The rows a plane has beyond its whole tiles (round 5): spmv_march2_kernel wants S to be a multiple of L; on other grids (200^3: S = 40000 = 19 x 2048
// + 1088) it now takes the whole tiles and this kernel the rest -- rows k S + row0 + i, i < rem, of every plane k, one thread per row, the base
// template's entries under the row's mask in ascending column order with the operands gathered from memory (coalesced along the rows): the sums of
// MatMult_SeqAIJ, and with CG the same p_new = z d + b p per operand (and per own row, stored) and x += a p as the march kernel's prologue forms.
// One dot partial per workgroup behind the march kernel's (dotpart + pbase); launched BEFORE the march kernel, which may fold all of them.
template <bool DOT, bool CG>
__global__ __launch_bounds__(256) void spmv_march2_rem_kernel(const hipxMarchPlan plan, const unsigned char *__restrict__ tid, const unsigned int *__restrict__ tmask, const double *__restrict__ x,
                                                              double *__restrict__ yout, double *__restrict__ dotpart, const int row0, const int rem, const hipxMarchCG cg, const int skipmask = 0)
{
  __shared__ double s_w[4];
  const int         i = (int)blockIdx.x * 256 + (int)threadIdx.x, k = (int)blockIdx.y;
  const bool        dskip = ((skipmask & 1) && k == 0) || ((skipmask & 2) && k == (int)gridDim.y - 1);  // (the boundary planes of an MPIAIJ diagonal block: see spmv_march2_kernel)
  double            acc = 0.0;
  if (i < rem) {
    const long long row = (long long)k * plan.S + row0 + i;
    const unsigned  mk  = tmask[tid[row]];
    double          cgb = 0.0, cga = 0.0, cgd = 1.0;
    if (CG) {
      cgb = cg.dev_beta_new ? (*cg.dev_beta_new / *cg.dev_beta_old) : cg.b;
      cga = cg.dev_beta_new ? (*cg.dev_beta_old / *cg.dev_dpi) : cg.a;
      cgd = cg.dconst;
    }
    double sum = 0.0, xd = 0.0;
    for (int e = 0; e < plan.ne; e++) {
      if ((mk >> e) & 1u) {
        const long long c = row + (e < plan.nlo ? -(long long)plan.S : (e < plan.nlo + plan.nmid ? 0ll : (long long)plan.S)) + plan.b[e];
        double          xv = x[c];
        if (CG) {
          const double zv = cg.z[c] * cgd;
          xv              = zv + cgb * xv;
        }
        sum += plan.a[e] * xv;
        if (e == plan.ne / 2) xd = xv;
      }
    }
    yout[row] = sum;
    if (CG) {
      const double po = x[row], zv = cg.z[row] * cgd;
      cg.pnew[row]    = zv + cgb * po;
      cg.xsol[row]    = cg.xsol[row] + cga * po;
    }
    if (DOT && !dskip) acc = xd * sum;
  }
  if (DOT) {
    const double w = hipx::wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
      double r = s_w[0];
      r += s_w[1];
      r += s_w[2];
      r += s_w[3];
      // (agent-scope store: the march kernel's last workgroup reads the partials with agent-scope loads; the kernel boundary orders the two launches)
      __hip_atomic_store(&dotpart[(size_t)blockIdx.y * gridDim.x + blockIdx.x], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// set-up check of spmv_march2_kernel's plane periodicity: flag <- 1 if some row k S + i (2 <= k <= nplanes - 2) has another template than row S + i
__global__ __launch_bounds__(256) void march2_periodic_kernel(const unsigned char *__restrict__ tid, long long S, int nplanes, unsigned int *flag)
{
  const long long n = (long long)(nplanes - 3) * S;
  bool            bad = false;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) bad |= tid[2 * S + i] != tid[S + (i % S)];
  if (bad) atomicOr(flag, 1u);
}

// rows [row0, m) of a template matrix, one row per thread and pass (the partial last chunk of spmv_pair_kernel): the row's own
// template from the global tables; dot partials in the layout of the chunked kernels (4 per chunk of 512 rows: rows t, t + 256)
template <int MODE, bool DOT, int EPI = 0>
__global__ __launch_bounds__(256) void spmv_tmpl_tail_kernel(hipx_int m, hipx_int row0, hipx_int chunk, const unsigned char *__restrict__ tid, const int *__restrict__ tstart,
                                                             const int *__restrict__ toff, const double *__restrict__ tval, const double *__restrict__ x, const double *yin, double *yout,
                                                             double *dotpart, const hipxPairEpi epi = hipxPairEpi{})
{
  double cdot = 0.0;
  for (int rr = 0; rr < 2; rr++) {
    const long long row = (long long)row0 + threadIdx.x + rr * 256;
    if (row < (long long)m) {
      const int id = tid[row];
      double    sum = (MODE == 1) ? yin[row] : 0.0;
      for (int k = tstart[id]; k < tstart[id + 1]; k++) sum += tval[k] * x[row + toff[k]];
      if (EPI == 1) {
        const double r0 = epi.b[row] - sum, z0 = epi.dinv ? r0 * epi.dinv[row] : r0, xp = epi.pprev[row], xc = x[row];
        if (epi.br == 0) sum = xp + epi.beta * xc + epi.gamma * z0;
        else if (epi.br == 1) sum = epi.alpha * xp + epi.beta * xc + z0;
        else if (epi.br == 2) sum = epi.alpha * xp + epi.beta * xc;
        else sum = epi.alpha * xp + epi.beta * xc + epi.gamma * z0;
      }
      yout[row] = sum;
      if (DOT) cdot += x[row] * sum;
    }
  }
  if (DOT) {
    const double w = hipx::wave_sum(cdot);
    if ((threadIdx.x & 63) == 0) dotpart[(size_t)chunk * 4 + (threadIdx.x >> 6)] = w;
  }
}

template <typename IT>
__global__ void diagpos_kernel(hipx_int m, const IT *ai, const hipx_int *aj, int64_t *diagpos, unsigned int *missing)
{
  __shared__ int lcol[4][RW_CH];  // (launched with 256 threads: a wave per 64 consecutive rows, rowwalk64)
  const int      wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (long long rb = (long long)blockIdx.x * 256 + 64 * wv; rb < (long long)m; rb += (long long)gridDim.x * 256) {
    const long long r   = rb + lane;
    int64_t         pos = -1;
    rowwalk64<IT>(m, ai, aj, nullptr, (hipx_int)rb, lcol[wv], nullptr, [&](IT k, int c, unsigned long long) {
      if (pos < 0 && c == (int)r) pos = (int64_t)k;  // the FIRST entry on the diagonal, as the row loop found it
    });
    if (r < (long long)m) {
      diagpos[r] = pos;
      if (pos < 0) atomicAdd(missing, 1u);
    }
  }
}

__global__ void getdiag_kernel(hipx_int m, const int64_t *diagpos, const double *aa, double *d)
{
  for (hipx_int r = (hipx_int)blockIdx.x * blockDim.x + threadIdx.x; r < m; r += (hipx_int)gridDim.x * blockDim.x) {
    int64_t p = diagpos[r];
    d[r]      = p >= 0 ? aa[p] : 0.0;  // aij.c:1347-1380: missing diagonal -> 0
  }
}

// PCSetUp_Jacobi (jacobi.c:205-266): d = diag; VecReciprocal (0 stays 0); zeros -> 1.  One pass.
__global__ void jacobi_setup_kernel(hipx_int m, const int64_t *diagpos, const double *aa, double *dinv)
{
  for (hipx_int r = (hipx_int)blockIdx.x * blockDim.x + threadIdx.x; r < m; r += (hipx_int)gridDim.x * blockDim.x) {
    int64_t p = diagpos[r];
    double  v = p >= 0 ? aa[p] : 0.0;
    if (v != 0.0) v = 1.0 / v;
    if (v == 0.0) v = 1.0;
    dinv[r] = v;
  }
}

void build_row_blocks(hipx_int nrows, const int64_t *ai, int cfg, std::vector<hipx_int> &rb)
{
  const int64_t  cap  = kCfg[cfg].cap;
  const hipx_int rows = kCfg[cfg].threads * kCfg[cfg].rpt;
  rb.clear();
  rb.push_back(0);
  hipx_int r = 0;
  while (r < nrows) {
    const int64_t ka = ai[r] & ~(int64_t)3;
    hipx_int      r1 = r;
    while (r1 < nrows && (r1 - r) < rows && (ai[r1 + 1] - ka) <= cap) r1++;
    if (r1 == r) r1 = r + 1;  // one long row: block-wide path
    rb.push_back(r1);
    r = r1;
  }
}

// variant 0 (auto): packed 16-bit columns once the host set-up pays off (>= 2^20 nonzeros); short rows (<= 16 entries on
// average: every thread of a block owns a row) use the row-parallel gather, longer rows the product-staging kernel.
// Measured on MI355X: 7-pt 256^3 0.302 (row-parallel) / 0.319 (staged) / 0.346 ms (32-bit columns); 27-pt 128^3
// 0.155 / 0.128 / 0.139 ms.
int auto_tile_mode(hipxMat A)
{
  if (A->compressed || A->nnz < ((int64_t)1 << 20) || A->nrows_c <= 0) return 0;
  return (A->nnz <= (int64_t)16 * A->nrows_c) ? 3 : 2;
}

// Value dictionary: succeeds when a[] holds at most 256 distinct bit patterns (compared as 64-bit integers, so -0.0, NaN
// payloads etc. stay exact).  VdTable: the host side of the code table (prefix check, code assignment); the passes over
// a[] run on the device (vd_collect_kernel / vd_encode_kernel).
struct VdTable {
  static constexpr int kSlots = 1024;
  uint64_t key[kSlots];
  int      code[kSlots];
  int      count = 0;
  VdTable() { std::fill(code, code + kSlots, -1); }
  static inline unsigned slot_of(uint64_t k) { return (unsigned)((k * 0x9E3779B97F4A7C15ull) >> 54); }
  inline int find(uint64_t k) const
  {
    unsigned s = slot_of(k);
    while (code[s] >= 0) {
      if (key[s] == k) return code[s];
      s = (s + 1) & (kSlots - 1);
    }
    return -1;
  }
  inline bool insert(uint64_t k)  // false = dictionary full
  {
    unsigned s = slot_of(k);
    while (code[s] >= 0) {
      if (key[s] == k) return true;
      s = (s + 1) & (kSlots - 1);
    }
    if (count >= 256) return false;
    key[s]  = k;
    code[s] = count++;
    return true;
  }
};

template <typename IT>
int create_common(hipx_int m, hipx_int n, hipx_int nrows, const IT *ai, const hipx_int *ridx, const hipx_int *aj, const double *aa, bool is64, hipxMat *out)
{
  HIPX_ARG(m >= 0 && n >= 0 && nrows >= 0 && out, "bad sizes");
  hipxMat A  = new hipxMat_s;
  A->m       = m;
  A->n       = n;
  A->is64    = is64;
  A->nnz     = nrows ? (int64_t)ai[nrows] : 0;
  A->compressed = ridx != nullptr;
  A->nrows_c = nrows;
  const size_t pad = 8;  // the stream kernel reads whole quads: pad so a quad never leaves the allocation
  HIPX_HIP(hipMalloc(&A->d_i, sizeof(IT) * ((size_t)nrows + 1)));
  HIPX_HIP(hipMalloc((void **)&A->d_j, sizeof(hipx_int) * ((size_t)A->nnz + pad)));
  HIPX_HIP(hipMalloc((void **)&A->d_a, sizeof(double) * ((size_t)A->nnz + pad)));
  HIPX_HIP(hipMemsetAsync(A->d_j + A->nnz, 0, sizeof(hipx_int) * pad, rt().compute));
  HIPX_HIP(hipMemsetAsync(A->d_a + A->nnz, 0, sizeof(double) * pad, rt().compute));
  static const IT zero = 0;
  HIPX_HIP(hipMemcpyAsync(A->d_i, nrows ? ai : &zero, sizeof(IT) * ((size_t)nrows + 1), hipMemcpyHostToDevice, rt().compute));
  if (A->nnz) {
    HIPX_HIP(hipMemcpyAsync(A->d_j, aj, sizeof(hipx_int) * (size_t)A->nnz, hipMemcpyHostToDevice, rt().compute));
    if (aa) HIPX_HIP(hipMemcpyAsync(A->d_a, aa, sizeof(double) * (size_t)A->nnz, hipMemcpyHostToDevice, rt().compute));
    else HIPX_HIP(hipMemsetAsync(A->d_a, 0, sizeof(double) * (size_t)A->nnz, rt().compute));  // pattern only (values follow from hipxMatSetValuesCOO)
  }
  A->device_bytes = (int64_t)(sizeof(IT) * ((size_t)nrows + 1) + (sizeof(hipx_int) + sizeof(double)) * ((size_t)A->nnz + pad));
  if (ridx) {
    HIPX_HIP(hipMalloc((void **)&A->d_ridx, sizeof(hipx_int) * ((size_t)nrows + 1)));
    if (nrows) HIPX_HIP(hipMemcpyAsync(A->d_ridx, ridx, sizeof(hipx_int) * (size_t)nrows, hipMemcpyHostToDevice, rt().compute));
    A->device_bytes += (int64_t)sizeof(hipx_int) * nrows;
  }
  if (!ridx && nrows > 4096 && A->nnz) {
    // typical far offset: median over sampled rows of max |col - row|
    std::vector<int64_t> offs;
    const hipx_int       step = std::max<hipx_int>(1, nrows / 2048);
    for (hipx_int r = step / 2; r < nrows; r += step) {
      int64_t mx = 0;
      for (IT k = ai[r]; k < ai[r + 1]; k++) mx = std::max<int64_t>(mx, std::llabs((long long)aj[k] - (long long)r));
      offs.push_back(mx);
    }
    std::nth_element(offs.begin(), offs.begin() + offs.size() / 2, offs.end());
    const int64_t med = offs[offs.size() / 2];
    // accept only a regular band: at least half of the samples within 2% of the median
    size_t close = 0;
    for (int64_t o : offs) close += (std::llabs((long long)(o - med)) * 50 <= med) ? 1 : 0;
    if (med > 0 && close * 2 >= offs.size()) A->far_offset = med;
  }
  A->h_i.resize((size_t)nrows + 1);
  for (hipx_int r = 0; r <= nrows; r++) A->h_i[r] = nrows ? (int64_t)ai[r] : 0;
  HIPX_HIP(hipStreamSynchronize(rt().compute));  // the caller's arrays may go away after return
  if (!ridx && m) {
    HIPX_HIP(hipMalloc((void **)&A->d_diagpos, sizeof(int64_t) * (size_t)m));
    unsigned int *cnt = rt().d_tickets + (HIPX_MAX_RED_SLOTS - 1);
    HIPX_HIP(hipMemsetAsync(cnt, 0, sizeof(unsigned int), rt().compute));
    hipx_int g = std::min<hipx_int>((m + 255) / 256, 4096);
    diagpos_kernel<IT><<<(unsigned)g, 256, 0, rt().compute>>>(m, (const IT *)A->d_i, A->d_j, A->d_diagpos, cnt);
    HIPX_LAUNCH_CHECK();
    unsigned int missing = 0;
    HIPX_HIP(hipMemcpyAsync(&missing, cnt, sizeof(unsigned int), hipMemcpyDeviceToHost, rt().compute));
    HIPX_HIP(hipStreamSynchronize(rt().compute));
    HIPX_HIP(hipMemsetAsync(cnt, 0, sizeof(unsigned int), rt().compute));
    A->diag_dense = (missing == 0) && (m <= n);
    A->device_bytes += (int64_t)sizeof(int64_t) * m;
  }
  A->tile_mode = auto_tile_mode(A);  // variant 0
  A->vd_mode   = A->tile_mode ? 1 : 0;
  A->tmpl_mode = A->tile_mode ? 1 : 0;
  A->ptm_mode  = A->tile_mode ? 1 : 0;
  A->sell_mode = A->tile_mode ? 2 : 0;
  *out = A;
  return HIPX_SUCCESS;
}

// variant -> (geometry, non-temporal loads).  0 = auto.
inline void decode_variant(int variant, int &cfg, bool &nt)
{
  if (variant <= 0) variant = 1;
  cfg = (variant - 1) >> 1;
  nt  = ((variant - 1) & 1) != 0;
  if (cfg >= kNumCfg) cfg = 0;
}

int ensure_row_blocks(hipxMat A, int cfg)
{
  if (A->rb_ready[cfg]) return HIPX_SUCCESS;
  std::vector<hipx_int> rb;
  const hipx_int        nrows = A->nrows_c;
  if (nrows) build_row_blocks(nrows, A->h_i.data(), cfg, rb);
  else rb.assign(1, 0);
  A->nblocks[cfg] = (hipx_int)rb.size() - 1;
  HIPX_HIP(hipMalloc((void **)&A->d_rb[cfg], sizeof(hipx_int) * rb.size()));
  HIPX_HIP(hipMemcpyAsync(A->d_rb[cfg], rb.data(), sizeof(hipx_int) * rb.size(), hipMemcpyHostToDevice, rt().compute));
  HIPX_HIP(hipStreamSynchronize(rt().compute));
  A->device_bytes += (int64_t)(sizeof(hipx_int) * rb.size());
  // Band-aware schedule.  In natural order an XCD re-reads x[i] three times for a 3-D stencil (as the +D, 0 and -D
  // neighbour, D = far_offset = n^2) with 2*D rows of matrix stream in between -- more than its 4 MiB L2 holds, so x is
  // fetched from HBM ~3x (measured: FETCH_SIZE 1.88 GB vs 1.61 GB algorithmic on 7-pt 256^3).  Walking the rows as
  // (tile of T rows inside a period of D) x (period index) makes the three uses fall within 2*T rows of stream;
  // T is sized so that this fits comfortably in L2.  Rows keep their block, only the ORDER of blocks changes,
  // so y is bit-identical.  Skipped when the matrix has no far band or the natural reuse distance already fits.
  const hipx_int nb = A->nblocks[cfg];
  if (nb > 16 && A->far_offset > 0 && nrows > 0) {
    const double  bytes_per_row = 12.0 * (double)A->nnz / (double)nrows + 12.0;
    const double  l2_budget     = 1.5e6;  // of the 4 MiB per XCD; the rest holds the streams in flight
    const int64_t D             = A->far_offset;
    if (2.0 * (double)D * bytes_per_row > l2_budget && D < (int64_t)nrows) {
      int64_t T = (int64_t)(l2_budget / (2.0 * bytes_per_row));
      if (T < 256) T = 256;
      const int64_t ntiles = (D + T - 1) / T;
      T                    = (D + ntiles - 1) / ntiles;
      const hipx_int per_xcd = (nb + 7) / 8;
      std::vector<hipx_int> sched((size_t)nb);
      std::vector<std::pair<int64_t, hipx_int>> keyed;
      for (int x = 0; x < 8; x++) {
        const hipx_int b0 = std::min<hipx_int>(nb, x * per_xcd), b1 = std::min<hipx_int>(nb, (x + 1) * per_xcd);
        keyed.clear();
        for (hipx_int b = b0; b < b1; b++) {
          const int64_t r0 = rb[b], period = r0 / D, tile = (r0 % D) / T;
          keyed.emplace_back((tile << 32) | period, b);  // tile-major, period-minor, block order inside
        }
        std::stable_sort(keyed.begin(), keyed.end(), [](const auto &a, const auto &c) { return a.first < c.first; });
        for (hipx_int k = 0; k < (hipx_int)keyed.size(); k++) sched[b0 + k] = keyed[k].second;
      }
      HIPX_HIP(hipMalloc((void **)&A->d_sched[cfg], sizeof(hipx_int) * (size_t)nb));
      HIPX_HIP(hipMemcpy(A->d_sched[cfg], sched.data(), sizeof(hipx_int) * (size_t)nb, hipMemcpyHostToDevice));
      A->device_bytes += (int64_t)(sizeof(hipx_int) * (size_t)nb);
    }
  }
  A->rb_ready[cfg] = true;
  return HIPX_SUCCESS;
}

template <typename IT, int CFG, bool NT, int MODE, bool CPROW, bool DOT>
int launch_cfg(hipxMat A, const double *x, const double *yin, double *yout, double *dotpart)
{
  const hipx_int nb = A->nblocks[CFG];
  if (nb == 0) return HIPX_SUCCESS;
  const hipx_int per_xcd = (nb + 7) / 8;
  const unsigned grid    = (unsigned)(per_xcd * 8);
  spmv_stream_kernel<IT, kCfg[CFG].threads, kCfg[CFG].cap, kCfg[CFG].rpt, NT, MODE, CPROW, DOT>
    <<<grid, kCfg[CFG].threads, 0, rt().compute>>>(A->d_rb[CFG], A->sched_mode ? A->d_sched[CFG] : nullptr, nb, per_xcd, (const IT *)A->d_i, A->d_j, A->d_a, x, yin, yout, A->d_ridx, dotpart);
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}

template <typename IT, int MODE, bool CPROW, bool DOT>
int launch_spmv_c(hipxMat A, const double *x, const double *yin, double *yout, double *dotpart)
{
  int  cfg;
  bool nt;
  decode_variant(A->variant, cfg, nt);
  int ierr = ensure_row_blocks(A, cfg);
  if (ierr) return ierr;
#define HIPX_CASE(C) \
  case C: \
    return nt ? launch_cfg<IT, C, true, MODE, CPROW, DOT>(A, x, yin, yout, dotpart) : launch_cfg<IT, C, false, MODE, CPROW, DOT>(A, x, yin, yout, dotpart);
  switch (cfg) {
    HIPX_CASE(0)
    HIPX_CASE(1)
    HIPX_CASE(2)
    HIPX_CASE(3)
    HIPX_CASE(4)
    HIPX_CASE(5)
  }
#undef HIPX_CASE
  return fail(HIPX_ERR_ARG, "unknown SpMV geometry", __FILE__, __LINE__);
}


struct SetupTimer {  // HIPX_SETUP_TIMING=1: wall time of the packed-format set-up steps on stderr
  const char *what;
  std::chrono::steady_clock::time_point t0;
  static bool on() { static const bool v = getenv("HIPX_SETUP_TIMING") != nullptr; return v; }
  explicit SetupTimer(const char *w) : what(w), t0(std::chrono::steady_clock::now()) {}
  ~SetupTimer()
  {
    if (!on()) return;
    (void)hipStreamSynchronize(rt().compute);
    fprintf(stderr, "[hipx set-up] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
};

int ensure_vdict(hipxMat A)
{
  if (A->vd_ready) return HIPX_SUCCESS;
  SetupTimer tm("value dictionary (device)");
  A->vd_ready = true;
  A->vd_ok    = false;
  const size_t nnz = (size_t)A->nnz;
  if (!nnz) return HIPX_SUCCESS;
  hipStream_t st = rt().compute;
  {  // cheap reject for general matrices: 64 Ki-entry prefix on the host
    const size_t          np = std::min<size_t>(nnz, 65536);
    std::vector<uint64_t> pre(np);
    HIPX_HIP(hipMemcpyAsync(pre.data(), A->d_a, sizeof(uint64_t) * np, hipMemcpyDeviceToHost, st));
    HIPX_HIP(hipStreamSynchronize(st));
    VdTable t;
    for (size_t k = 0; k < np; k++)
      if (!t.insert(pre[k])) return HIPX_SUCCESS;
  }
  const unsigned      grid = 2048, cap = grid * 256;
  unsigned long long *d_list = nullptr;
  unsigned int       *d_cnt  = nullptr;
  HIPX_HIP(hipMalloc((void **)&d_list, sizeof(unsigned long long) * cap));
  HIPX_HIP(hipMalloc((void **)&d_cnt, sizeof(unsigned int) * 2));
  HIPX_HIP(hipMemsetAsync(d_cnt, 0, sizeof(unsigned int) * 2, st));
  vd_collect_kernel<<<grid, 256, 0, st>>>((const unsigned long long *)A->d_a, (long long)nnz, d_list, d_cnt, cap);
  unsigned int hc[2] = {0, 0};
  HIPX_HIP(hipMemcpyAsync(hc, d_cnt, sizeof(hc), hipMemcpyDeviceToHost, st));
  HIPX_HIP(hipStreamSynchronize(st));
  std::vector<uint64_t> dict;
  if (!hc[1] && hc[0] <= cap) {
    dict.resize(hc[0]);
    if (hc[0]) HIPX_HIP(hipMemcpy(dict.data(), d_list, sizeof(uint64_t) * hc[0], hipMemcpyDeviceToHost));
    std::sort(dict.begin(), dict.end());
    dict.erase(std::unique(dict.begin(), dict.end()), dict.end());
  }
  (void)hipFree(d_list);
  (void)hipFree(d_cnt);
  if (hc[1] || hc[0] > cap || dict.empty() || dict.size() > 256) return HIPX_SUCCESS;
  VdTable global;
  for (uint64_t v : dict) global.insert(v);  // code = rank in the sorted dictionary (deterministic)
  std::vector<short> codes(VdTable::kSlots);
  for (int sl = 0; sl < VdTable::kSlots; sl++) codes[(size_t)sl] = (short)global.code[sl];
  const size_t vcbytes = ((nnz + 7) / 8) * 8 + 16;
  if (!A->d_vc) {
    HIPX_HIP(hipMalloc((void **)&A->d_vc, vcbytes));
    HIPX_HIP(hipMalloc((void **)&A->d_vdict, sizeof(uint64_t) * 256));
    A->device_bytes += (int64_t)(vcbytes + sizeof(uint64_t) * 256);
  }
  unsigned long long *d_keys  = nullptr;
  short              *d_codes = nullptr;
  HIPX_HIP(hipMalloc((void **)&d_keys, sizeof(uint64_t) * VdTable::kSlots));
  HIPX_HIP(hipMalloc((void **)&d_codes, sizeof(short) * VdTable::kSlots));
  HIPX_HIP(hipMemcpyAsync(d_keys, global.key, sizeof(uint64_t) * VdTable::kSlots, hipMemcpyHostToDevice, st));
  HIPX_HIP(hipMemcpyAsync(d_codes, codes.data(), sizeof(short) * VdTable::kSlots, hipMemcpyHostToDevice, st));
  HIPX_HIP(hipMemsetAsync(A->d_vc, 0, vcbytes, st));
  vd_encode_kernel<<<4096, 256, 0, st>>>((const unsigned long long *)A->d_a, (long long)nnz, d_keys, d_codes, A->d_vc);
  dict.resize(256, 0);
  HIPX_HIP(hipMemcpyAsync(A->d_vdict, dict.data(), sizeof(uint64_t) * 256, hipMemcpyHostToDevice, st));
  HIPX_HIP(hipStreamSynchronize(st));
  HIPX_LAUNCH_CHECK();
  (void)hipFree(d_keys);
  (void)hipFree(d_codes);
  A->vd_count = (int)global.count;
  A->vd_ok    = true;
  return HIPX_SUCCESS;
}

int ensure_pk16(hipxMat A, int cfg = 0)
{
  if (A->pk_ready && A->pk_cfg == cfg) return HIPX_SUCCESS;
  int ierr;
  {
    SetupTimer tm("row blocks (host)");
    if ((ierr = ensure_row_blocks(A, cfg))) return ierr;
  }
  SetupTimer tm("packed columns (device)");
  hipStream_t st = rt().compute;
  if (A->pk_ready) {  // built on another block geometry (the value dictionary appeared / went away): rebuild
    HIPX_HIP(hipStreamSynchronize(st));
    (void)hipFree(A->d_pk);
    (void)hipFree(A->d_pkbase);
    (void)hipFree(A->d_pkdesc);
    A->d_pk = nullptr;
    A->d_pkbase = nullptr;
    A->d_pkdesc = nullptr;
    A->pk_ready = false;
  }
  const hipx_int nb = A->nblocks[cfg];
  std::vector<hipx_int> rb((size_t)nb + 1);
  HIPX_HIP(hipMemcpy(rb.data(), A->d_rb[cfg], sizeof(hipx_int) * ((size_t)nb + 1), hipMemcpyDeviceToHost));
  const int64_t      *hi = A->h_i.data();
  std::vector<PkDesc> desc((size_t)std::max<hipx_int>(nb, 1));
  for (hipx_int b = 0; b < nb; b++) desc[(size_t)b] = PkDesc{rb[b], rb[b + 1], (long long)hi[rb[b]], (long long)hi[rb[b + 1]]};
  const size_t npk = (size_t)A->nnz + 16, nbase = (size_t)std::max<hipx_int>(nb, 1) * PK_WMAX;
  HIPX_HIP(hipMalloc(&A->d_pkdesc, sizeof(PkDesc) * desc.size()));
  HIPX_HIP(hipMalloc((void **)&A->d_pk, sizeof(unsigned short) * npk));
  HIPX_HIP(hipMalloc((void **)&A->d_pkbase, sizeof(hipx_int) * nbase));
  HIPX_HIP(hipMemcpyAsync(A->d_pkdesc, desc.data(), sizeof(PkDesc) * desc.size(), hipMemcpyHostToDevice, st));
  HIPX_HIP(hipMemsetAsync(A->d_pk, 0, sizeof(unsigned short) * npk, st));
  HIPX_HIP(hipMemsetAsync(A->d_pkbase, 0xff, sizeof(hipx_int) * nbase, st));
  unsigned int *cnt = rt().d_tickets + (HIPX_MAX_RED_SLOTS - 1);  // scratch word, zero between uses
  HIPX_HIP(hipMemsetAsync(cnt, 0, sizeof(unsigned int), st));
  if (nb) pk_build_kernel<<<(unsigned)nb, 256, 0, st>>>((const PkDesc *)A->d_pkdesc, nb, (long long)kCfg[cfg].cap, A->d_j, A->d_pk, A->d_pkbase, cnt);
  unsigned int fb = 0;
  HIPX_HIP(hipMemcpyAsync(&fb, cnt, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
  HIPX_HIP(hipStreamSynchronize(st));  // desc[] (host) is read by the copy above
  HIPX_LAUNCH_CHECK();
  HIPX_HIP(hipMemsetAsync(cnt, 0, sizeof(unsigned int), st));
  A->pk_fallback_blocks = fb;
  A->device_bytes += (int64_t)(sizeof(unsigned short) * npk + sizeof(hipx_int) * nbase + sizeof(PkDesc) * desc.size());
  A->pk_cfg   = cfg;
  A->pk_ready = true;
  return HIPX_SUCCESS;
}


// Row templates (see spmv_tmpl_kernel): attempted once per pattern + value state.
void free_templates(hipxMat A)
{
  (void)hipFree(A->d_tid);
  (void)hipFree(A->d_tstart);
  (void)hipFree(A->d_toff);
  (void)hipFree(A->d_tval);
  (void)hipFree(A->d_tq);
  (void)hipFree(A->d_march_dump);
  A->d_march_dump = nullptr;
  (void)hipFree(A->d_tmask);
  A->d_tmask   = nullptr;
  A->tmpl_base = -1;
  A->pair_ok   = false;
  A->march_ok  = false;
  A->march_nt  = 256;
  A->march2_state = 0;
  A->d_tq = nullptr;
  A->tq_launches = 0;
  A->tq_geom = -1;
  A->tmpl_maxoff = -1;
  A->d_tid = nullptr;
  A->d_tstart = nullptr;
  A->d_toff = nullptr;
  A->d_tval = nullptr;
  A->tmpl_ok = false;
  A->ntmpl = A->tmpl_nent = 0;
}

// HIPX_SETUP_TRACE=1 (developer switch): wall-clock stamps of the format decisions a product goes through (device synchronised at every stamp; the
// first product of a matrix builds its formats -- where that time goes)
static void setup_trace(const char *what)
{
  static const bool on = getenv("HIPX_SETUP_TRACE") != nullptr;
  if (!on) return;
  static double t_prev = 0.0;
  (void)hipDeviceSynchronize();
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  const double t = ts.tv_sec + 1e-9 * ts.tv_nsec;
  fprintf(stderr, "[hipx setup trace] %-34s %+9.3f ms\n", what, t_prev == 0.0 ? 0.0 : 1e3 * (t - t_prev));
  t_prev = t;
}

template <typename IT>
int build_templates(hipxMat A)
{
  SetupTimer tm("row templates (device)");
  hipStream_t    st = rt().compute;
  const hipx_int m  = A->nrows_c;
  unsigned long long *d_hash = nullptr, *d_list = nullptr, *d_keys = nullptr, *d_count = nullptr;
  unsigned int       *d_cnt = nullptr;
  short              *d_codes = nullptr;
  int                *d_rep = nullptr;
  struct Guard {
    void **p[7];
    ~Guard()
    {
      for (auto q : p)
        if (q && *q) (void)hipFree(*q);
    }
  } guard{{(void **)&d_hash, (void **)&d_list, (void **)&d_keys, (void **)&d_count, (void **)&d_cnt, (void **)&d_codes, (void **)&d_rep}};
  const unsigned grid = 2048, cap = grid * 256;
  HIPX_HIP(hipMalloc((void **)&d_hash, sizeof(unsigned long long) * (size_t)m));
  HIPX_HIP(hipMalloc((void **)&d_list, sizeof(unsigned long long) * cap));
  HIPX_HIP(hipMalloc((void **)&d_cnt, sizeof(unsigned int) * 2));
  HIPX_HIP(hipMemsetAsync(d_cnt, 0, sizeof(unsigned int) * 2, st));
  const unsigned g = (unsigned)std::min<hipx_int>((m + 255) / 256, 8192);
  tmpl_hash_kernel<IT><<<g, 256, 0, st>>>(m, (const IT *)A->d_i, A->d_j, (const unsigned long long *)A->d_a, d_hash, A->ptm_build ? 0 : 1);
  vd_collect_kernel<<<grid, 256, 0, st>>>(d_hash, (long long)m, d_list, d_cnt, cap);
  unsigned int hc[2] = {0, 0};
  HIPX_HIP(hipMemcpyAsync(hc, d_cnt, sizeof(hc), hipMemcpyDeviceToHost, st));
  HIPX_HIP(hipStreamSynchronize(st));
  HIPX_LAUNCH_CHECK();
  setup_trace("  templates: row hashes + distinct");
  if (hc[1] || hc[0] > cap || !hc[0]) return HIPX_SUCCESS;  // more than 256 distinct rows somewhere: no templates
  std::vector<uint64_t> keys(hc[0]);
  HIPX_HIP(hipMemcpy(keys.data(), d_list, sizeof(uint64_t) * hc[0], hipMemcpyDeviceToHost));
  std::sort(keys.begin(), keys.end());
  keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
  if (keys.size() > (size_t)TMPL_MAX) return HIPX_SUCCESS;
  const int nt = (int)keys.size();
  VdTable   table;
  for (uint64_t v : keys) table.insert(v);
  std::vector<short> codes(VdTable::kSlots);
  for (int sl = 0; sl < VdTable::kSlots; sl++) codes[(size_t)sl] = (short)table.code[sl];
  std::vector<int> rep(TMPL_MAX, 0x7fffffff);
  HIPX_HIP(hipMalloc((void **)&d_keys, sizeof(uint64_t) * VdTable::kSlots));
  HIPX_HIP(hipMalloc((void **)&d_codes, sizeof(short) * VdTable::kSlots));
  HIPX_HIP(hipMalloc((void **)&d_rep, sizeof(int) * TMPL_MAX));
  HIPX_HIP(hipMalloc((void **)&d_count, sizeof(unsigned long long) * TMPL_MAX));
  HIPX_HIP(hipMalloc((void **)&A->d_tid, (size_t)m + 64));
  HIPX_HIP(hipMemcpyAsync(d_keys, table.key, sizeof(uint64_t) * VdTable::kSlots, hipMemcpyHostToDevice, st));
  HIPX_HIP(hipMemcpyAsync(d_codes, codes.data(), sizeof(short) * VdTable::kSlots, hipMemcpyHostToDevice, st));
  HIPX_HIP(hipMemcpyAsync(d_rep, rep.data(), sizeof(int) * TMPL_MAX, hipMemcpyHostToDevice, st));
  HIPX_HIP(hipMemsetAsync(d_count, 0, sizeof(unsigned long long) * TMPL_MAX, st));
  HIPX_HIP(hipMemsetAsync(A->d_tid, 0, (size_t)m + 64, st));
  tmpl_assign_kernel<<<std::min<unsigned>(g, 2048u), 256, 0, st>>>(m, d_hash, d_keys, d_codes, A->d_tid, d_rep, d_count);
  std::vector<unsigned long long> count(TMPL_MAX);
  HIPX_HIP(hipMemcpyAsync(rep.data(), d_rep, sizeof(int) * TMPL_MAX, hipMemcpyDeviceToHost, st));
  HIPX_HIP(hipMemcpyAsync(count.data(), d_count, sizeof(unsigned long long) * TMPL_MAX, hipMemcpyDeviceToHost, st));
  HIPX_HIP(hipStreamSynchronize(st));
  HIPX_LAUNCH_CHECK();
  // representative rows -> template table (host copies of those few rows)
  const int64_t *hi = A->h_i.data();
  A->h_tstart.assign((size_t)nt + 1, 0);
  A->h_tcount.assign((size_t)nt, 0);
  for (int t = 0; t < nt; t++) {
    if (rep[t] < 0 || rep[t] >= m) return HIPX_SUCCESS;
    A->h_tstart[t + 1] = A->h_tstart[t] + (int)(hi[rep[t] + 1] - hi[rep[t]]);
    A->h_tcount[t]     = (int64_t)count[t];
    if (A->h_tstart[t + 1] > TMPL_MAX_ENT) return HIPX_SUCCESS;
  }
  const int nent = A->h_tstart[nt];
  A->h_toff.assign((size_t)std::max(nent, 1), 0);
  A->h_tval.assign((size_t)std::max(nent, 1), 0.0);
  A->h_tdiag.assign((size_t)nt, -1);
  std::vector<hipx_int> cols;
  for (int t = 0; t < nt; t++) {
    const int len = A->h_tstart[t + 1] - A->h_tstart[t];
    if (!len) continue;
    cols.resize((size_t)len);
    HIPX_HIP(hipMemcpy(cols.data(), A->d_j + hi[rep[t]], sizeof(hipx_int) * (size_t)len, hipMemcpyDeviceToHost));
    HIPX_HIP(hipMemcpy(A->h_tval.data() + A->h_tstart[t], A->d_a + hi[rep[t]], sizeof(double) * (size_t)len, hipMemcpyDeviceToHost));
    for (int k = 0; k < len; k++) {
      A->h_toff[(size_t)A->h_tstart[t] + k] = (int)(cols[(size_t)k] - rep[t]);
      if (cols[(size_t)k] == rep[t] && A->h_tdiag[t] < 0) A->h_tdiag[t] = k;
    }
  }
  HIPX_HIP(hipMalloc((void **)&A->d_tstart, sizeof(int) * ((size_t)nt + 1)));
  HIPX_HIP(hipMalloc((void **)&A->d_toff, sizeof(int) * (A->h_toff.size() + 8)));
  HIPX_HIP(hipMalloc((void **)&A->d_tval, sizeof(double) * (A->h_tval.size() + 8)));
  HIPX_HIP(hipMemsetAsync(A->d_toff, 0, sizeof(int) * (A->h_toff.size() + 8), st));
  HIPX_HIP(hipMemsetAsync(A->d_tval, 0, sizeof(double) * (A->h_tval.size() + 8), st));
  HIPX_HIP(hipMemcpyAsync(A->d_tstart, A->h_tstart.data(), sizeof(int) * ((size_t)nt + 1), hipMemcpyHostToDevice, st));
  HIPX_HIP(hipMemcpyAsync(A->d_toff, A->h_toff.data(), sizeof(int) * A->h_toff.size(), hipMemcpyHostToDevice, st));
  HIPX_HIP(hipMemcpyAsync(A->d_tval, A->h_tval.data(), sizeof(double) * A->h_tval.size(), hipMemcpyHostToDevice, st));
  HIPX_HIP(hipMemsetAsync(d_cnt, 0, sizeof(unsigned int) * 2, st));
  {
    const int    vn  = (int)A->h_toff.size();
    const size_t tab = 12 * (size_t)vn;
    if (4 * 1024 * 12 + tab <= 64 * 1024)
      tmpl_verify_kernel<IT, 1024><<<g, 256, 4 * 1024 * 12 + tab, st>>>(m, A->n, (const IT *)A->d_i, A->d_j, (const unsigned long long *)A->d_a, A->d_tid, A->d_tstart, A->d_toff,
                                                                          (const unsigned long long *)A->d_tval, d_cnt, A->ptm_build ? 0 : 1, vn);
    else
      tmpl_verify_kernel<IT, 512><<<g, 256, 4 * 512 * 12 + tab, st>>>(m, A->n, (const IT *)A->d_i, A->d_j, (const unsigned long long *)A->d_a, A->d_tid, A->d_tstart, A->d_toff,
                                                                        (const unsigned long long *)A->d_tval, d_cnt, A->ptm_build ? 0 : 1, vn);
  }
  HIPX_HIP(hipMemcpyAsync(hc, d_cnt, sizeof(hc), hipMemcpyDeviceToHost, st));
  HIPX_HIP(hipStreamSynchronize(st));
  HIPX_LAUNCH_CHECK();
  setup_trace("  templates: ids + table + verify");
  if (hc[0]) return HIPX_SUCCESS;  // a hash collision (or a column outside the matrix): keep the general formats
  A->ntmpl     = nt;
  A->tmpl_nent = nent;
  A->tmpl_ok   = true;
  A->tmpl_maxlen = 0;
  for (int tt = 0; tt < nt; tt++) A->tmpl_maxlen = std::max(A->tmpl_maxlen, A->h_tstart[(size_t)tt + 1] - A->h_tstart[(size_t)tt]);
  A->device_bytes += (int64_t)m + 64 + (int64_t)(sizeof(int) * ((size_t)nt + 1) + 12 * A->h_toff.size());
  if (!A->ptm_build) {  // sub-template form (see d_tmask): is every template an ordered subsequence of the most common one, diagonal included?
    int best = 0;
    for (int tt = 1; tt < nt; tt++)
      if (A->h_tcount[(size_t)tt] > A->h_tcount[(size_t)best]) best = tt;
    const int b0 = A->h_tstart[(size_t)best], len0 = A->h_tstart[(size_t)best + 1] - b0;
    bool      sub = len0 >= 1 && len0 <= 32;
    std::vector<unsigned int> mask((size_t)nt, 0u);
    for (int tt = 0; tt < nt && sub; tt++) {
      int  j = 0;
      bool diag = false;
      for (int k = A->h_tstart[(size_t)tt]; k < A->h_tstart[(size_t)tt + 1] && sub; k++) {
        while (j < len0 && !(A->h_toff[(size_t)b0 + j] == A->h_toff[(size_t)k] && !memcmp(&A->h_tval[(size_t)b0 + j], &A->h_tval[(size_t)k], sizeof(double)))) j++;
        if (j >= len0) sub = false;
        else {
          mask[(size_t)tt] |= 1u << j;
          diag = diag || A->h_toff[(size_t)k] == 0;
          j++;
        }
      }
      if (!diag) sub = false;
    }
    if (sub) {
      HIPX_HIP(hipMalloc((void **)&A->d_tmask, sizeof(unsigned int) * (size_t)nt));
      HIPX_HIP(hipMemcpy(A->d_tmask, mask.data(), sizeof(unsigned int) * (size_t)nt, hipMemcpyHostToDevice));
      A->tmpl_base = best;
      // pair plan (hipxPairPlan): even offsets -> pairs; an odd offset o hangs on pair(o + 1) as slot 0 if that pair exists, else on
      // pair(o - 1) as slot 2 (created if need be)
      hipxPairPlan &pp = A->pair_plan;
      memset(&pp, 0, sizeof(pp));
      std::vector<int> ev;
      for (int k = 0; k < len0; k++) {
        const int o = A->h_toff[(size_t)b0 + k];
        if ((o & 1) == 0) ev.push_back(o);
      }
      for (int k = 0; k < len0; k++) {
        const int o = A->h_toff[(size_t)b0 + k];
        if ((o & 1) && std::find(ev.begin(), ev.end(), o + 1) == ev.end() && std::find(ev.begin(), ev.end(), o - 1) == ev.end()) ev.push_back(o - 1);
      }
      std::sort(ev.begin(), ev.end());
      bool pok = ev.size() <= 16 && !ev.empty();
      if (pok) {
        pp.npairs = (int)ev.size();
        pp.jdiag  = -1;
        for (int j = 0; j < 16; j++) pp.kb[j][0] = pp.kb[j][1] = pp.kb[j][2] = -1;
        for (int j = 0; j < pp.npairs; j++) {
          pp.e[j] = ev[(size_t)j];
          if (ev[(size_t)j] == 0) pp.jdiag = j;
        }
        int lastj = -1, lasts = -1;
        for (int k = 0; k < len0 && pok; k++) {
          const int    o = A->h_toff[(size_t)b0 + k];
          int          j = -1, sl = -1;
          const auto   at = [&](int off) { const auto it = std::find(ev.begin(), ev.end(), off); return it == ev.end() ? -1 : (int)(it - ev.begin()); };
          if ((o & 1) == 0) { j = at(o); sl = 1; }
          else if (at(o + 1) >= 0) { j = at(o + 1); sl = 0; }
          else { j = at(o - 1); sl = 2; }
          if (j < 0 || pp.kb[j][sl] >= 0 || j < lastj || (j == lastj && sl <= lasts)) pok = false;  // (order of the walk must be the entries' order)
          else {
            pp.kb[j][sl] = k;
            pp.a[j][sl]  = A->h_tval[(size_t)b0 + k];
            lastj = j;
            lasts = sl;
          }
        }
        if (pp.jdiag < 0 || pp.kb[pp.jdiag][1] < 0) pok = false;
        pp.jodd = -1;
        pp.eodd = 0;
        for (int j = 0; j < pp.npairs && pok; j++)
          if (pp.kb[j][0] >= 0 || pp.kb[j][2] >= 0) {
            pp.jodd = j;
            pp.eodd = pp.e[j];
          }
      }
      A->pair_ok = pok && A->m == A->n;  // (both forms bound their loads of x by the ROW count)
      // march plan (hipxMarchPlan): split the ascending offsets at the largest gap on either side of the diagonal
      {
        hipxMarchPlan &mp = A->march_plan;
        memset(&mp, 0, sizeof(mp));
        A->march_ok = false;
        int kd = -1;
        for (int k = 0; k < len0; k++)
          if (A->h_toff[(size_t)b0 + k] == 0) kd = k;
        int cutlo = -1, cuthi = -1;  // entries [0, cutlo] below, [cuthi, len0) above
        long long glo = 0, ghi = 0;
        for (int k = 0; k + 1 <= kd; k++) {
          const long long g = (long long)A->h_toff[(size_t)b0 + k + 1] - A->h_toff[(size_t)b0 + k];
          if (g > glo) { glo = g; cutlo = k; }
        }
        for (int k = kd; k + 1 < len0; k++) {
          const long long g = (long long)A->h_toff[(size_t)b0 + k + 1] - A->h_toff[(size_t)b0 + k];
          if (g > ghi) { ghi = g; cuthi = k + 1; }
        }
        if (kd >= 0 && cutlo >= 0 && cuthi >= 0 && len0 <= 32 && A->m == A->n) {
          const long long lo0 = A->h_toff[(size_t)b0], lo1 = A->h_toff[(size_t)b0 + cutlo], hi0 = A->h_toff[(size_t)b0 + cuthi], hi1 = A->h_toff[(size_t)b0 + len0 - 1];
          const long long S = (hi0 + hi1) / 2;
          long long       H = 0;
          bool            ok = (hi0 + hi1) % 2 == 0 && lo0 + lo1 == -(hi0 + hi1) && S % 2 == 0 && S >= 1024;
          for (int k = 0; k < len0 && ok; k++) {
            const long long o = A->h_toff[(size_t)b0 + k], a = k <= cutlo ? -1 : (k >= cuthi ? 1 : 0), b = o - a * S;
            H       = std::max(H, b < 0 ? -b : b);
            mp.b[k] = (int)b;
            mp.a[k] = A->h_tval[(size_t)b0 + k];
          }
          H = (H + 1) & ~1ll;
          const int Lsel = (getenv("HIPX_TMPL_MARCH_L") ? atoi(getenv("HIPX_TMPL_MARCH_L")) == 1024 : S / 2048 < 16) ? 1024 : 2048;  // (HIPX_TMPL_MARCH_L: developer switch)
          // three plane buffers of L + 2 H doubles, two workgroups per CU: <= 80 KiB (7-pt / 27-pt up to 680-point lines; longer lines keep the pair form)
          bool big = false;  // halos beyond the two-workgroup budget: one 512-thread workgroup per CU over 4096-row tiles (spmv_march2_kernel<..., 512> only)
          if (ok && 2 * H < S && 3 * (size_t)(Lsel + 2 * H) * sizeof(double) > 80 * 1024 && H <= 1024 && S % 4096 == 0 && 3 * (size_t)(4096 + 2 * H) * sizeof(double) <= 150 * 1024) big = true;
          if (ok && 2 * H < S && (big || 3 * (size_t)(Lsel + 2 * H) * sizeof(double) <= 80 * 1024)) {
            mp.ne    = len0;
            mp.nlo   = cutlo + 1;
            mp.nmid  = cuthi - cutlo - 1;
            mp.S     = (int)S;
            mp.H     = (int)H;
            mp.L     = big ? 4096 : Lsel;
            mp.full  = len0 == 32 ? 0xffffffffu : ((1u << len0) - 1u);
            A->march_ok = true;
            A->march_nt = big ? 512 : 256;
            if (!big && mp.L == 2048 && getenv("HIPX_MARCH_NT512")) A->march_nt = 512;  // developer switch: 2048-row tiles by 512 threads (4 rows each), two workgroups per CU = 4 waves per SIMD
          }
        }
      }
    }
  }
  return HIPX_SUCCESS;
}

int ensure_templates(hipxMat A)
{
  if (A->tmpl_ready) return HIPX_SUCCESS;
  A->tmpl_ready = true;
  free_templates(A);
  if (A->compressed || A->nrows_c <= 0 || A->nnz <= 0) return HIPX_SUCCESS;
  int ierr = A->is64 ? build_templates<int64_t>(A) : build_templates<hipx_int>(A);
  if (ierr || !A->tmpl_ok) free_templates(A);
  return ierr;
}


// Pattern templates (variant 29): the same device passes as the row templates with the values left out of the hash and of the
// verification, so a matrix with ARBITRARY values on a stencil pattern qualifies (variable-coefficient operators).  build_templates()
// writes into the row-template fields: they are stashed, the build runs in pattern-only mode, its result moves to the ptm fields.
void free_pattern_templates(hipxMat A)
{
  (void)hipFree(A->d_ptid);
  (void)hipFree(A->d_ptstart);
  (void)hipFree(A->d_ptoff);
  A->d_ptid    = nullptr;
  A->d_ptstart = nullptr;
  A->d_ptoff   = nullptr;
  A->ptm_ok    = false;
  A->ptm_ntmpl = A->ptm_nent = A->ptm_maxlen = 0;
}

int ensure_pattern_templates(hipxMat A)
{
  if (A->ptm_ready) return HIPX_SUCCESS;
  A->ptm_ready = true;
  free_pattern_templates(A);
  if (A->compressed || A->nrows_c <= 0 || A->nnz <= 0) return HIPX_SUCCESS;
  // stash the row templates
  unsigned char *s_tid = A->d_tid;
  int           *s_tstart = A->d_tstart, *s_toff = A->d_toff;
  double        *s_tval = A->d_tval;
  const bool     s_ok = A->tmpl_ok;
  const int      s_nt = A->ntmpl, s_ne = A->tmpl_nent, s_ml = A->tmpl_maxlen;
  std::vector<int>     h1 = std::move(A->h_tstart), h2 = std::move(A->h_toff), h3 = std::move(A->h_tdiag);
  std::vector<int64_t> h4 = std::move(A->h_tcount);
  std::vector<double>  h5 = std::move(A->h_tval);
  A->d_tid = nullptr;
  A->d_tstart = A->d_toff = nullptr;
  A->d_tval = nullptr;
  A->tmpl_ok = false;
  A->ntmpl = A->tmpl_nent = A->tmpl_maxlen = 0;
  A->ptm_build = true;
  const int64_t bytes0 = A->device_bytes;
  int ierr = A->is64 ? build_templates<int64_t>(A) : build_templates<hipx_int>(A);
  A->ptm_build = false;
  if (!ierr && A->tmpl_ok && A->tmpl_maxlen <= 1024) {
    A->d_ptid     = A->d_tid;
    A->d_ptstart  = A->d_tstart;
    A->d_ptoff    = A->d_toff;
    A->ptm_ok     = true;
    A->ptm_ntmpl  = A->ntmpl;
    A->ptm_nent   = A->tmpl_nent;
    A->ptm_maxlen = A->tmpl_maxlen;
    A->h_ptstart  = std::move(A->h_tstart);
    A->h_ptoff    = std::move(A->h_toff);
    A->h_ptdiag   = std::move(A->h_tdiag);
    A->h_ptcount  = std::move(A->h_tcount);
    (void)hipFree(A->d_tval);
  } else {
    (void)hipFree(A->d_tid);
    (void)hipFree(A->d_tstart);
    (void)hipFree(A->d_toff);
    (void)hipFree(A->d_tval);
    A->device_bytes = bytes0;
  }
  // restore the row templates
  A->d_tid = s_tid;
  A->d_tstart = s_tstart;
  A->d_toff = s_toff;
  A->d_tval = s_tval;
  A->tmpl_ok = s_ok;
  A->ntmpl = s_nt;
  A->tmpl_nent = s_ne;
  A->tmpl_maxlen = s_ml;
  A->h_tstart = std::move(h1);
  A->h_toff = std::move(h2);
  A->h_tdiag = std::move(h3);
  A->h_tcount = std::move(h4);
  A->h_tval = std::move(h5);
  return ierr;
}

// geometry of the template kernel: HIPX_TMPL_CFG = 0 (2 rows per thread, per-lane walk only) | 1 (2 rows, uniform fast path:
// default -- the smaller chunk keeps the x window of an XCD inside its L2) | 2 (4 rows, uniform fast path).  Tried and dropped
// (measured on MI355X, 7-pt 256^3 inside CG): the template cached in scalar registers with all gathers of a chunk issued as one
// group -- 116 VGPRs, 4 waves per SIMD: 0.163 ms against 0.146 ms; the kernel wants occupancy, not fewer round trips.  Also tried:
// adjacent row pairs per thread with one 16-byte gather per entry (half the memory instructions): 0.169 ms against 0.144 ms; the x
// ranges a chunk touches copied into an LDS tile with 16-byte coalesced loads and the row sums formed from LDS (one global round
// trip per chunk, 40 VGPRs, 16 KiB of LDS per workgroup): 0.163 ms against 0.146 ms (0.137 against 0.114 stand-alone); one chunk per
// workgroup without the ticket queue (XCD-aware static map, 32768 one-shot workgroups): 0.157 ms against 0.146 ms.
// Round 3 (profiles/r03_tmpl_probes.txt, 7-pt 256^3 stand-alone 0.119-0.121 ms): timing probes (HIPX_TMPL_PROBE) show that neither the
// y stores (0.116), nor the gathers' far lines (every gather folded onto the row's own lines: 0.118; far offsets folded into +-4096:
// 0.120) carry the time; static round-robin chunks without tickets and barriers: 0.102 stand-alone but 0.158 inside CG (the workgroups
// drift, x is re-fetched); the same with a progress throttle (done counter polled per chunk): 1.2 ms (the throttle's bounded spins
// expire: not every workgroup is co-resident); a fifth wave per workgroup that first-touches ids and far x lines two rounds ahead:
// 0.144 (a quarter of the compute waves gone, nothing gained); the current template kept in scalar registers across chunks: +-0.
// Developer switches of the template / march kernels, read ONCE per process (the fused CG path looked eight of them up per iteration --
// each getenv scans environ, microseconds of host time per launch, and is not safe against a concurrent setenv).  Tests that want another setting
// start another process.  Only HIPX_MAT_NO_INODE is still read per call (tests flip it inside one process; one lookup per product).
struct DevSwitches {
  bool nomarch, nosub, nopair, march1, trace, nt_store;
  int  probe, march_units, nt_x, pairmax, nt_stream;
};
static const DevSwitches &dev_sw()
{
  static const DevSwitches v = [] {
    DevSwitches d;
    d.nomarch     = getenv("HIPX_TMPL_NOMARCH") != nullptr;
    d.nosub       = getenv("HIPX_TMPL_NOSUB") != nullptr;
    d.nopair      = getenv("HIPX_TMPL_NOPAIR") != nullptr;
    d.march1      = getenv("HIPX_MARCH1") != nullptr;
    d.trace       = getenv("HIPX_TMPL_TRACE") != nullptr;
    d.nt_store    = getenv("HIPX_MARCH_NT_STORE") != nullptr;
    d.probe       = getenv("HIPX_TMPL_PROBE") ? atoi(getenv("HIPX_TMPL_PROBE")) : 0;
    d.march_units = getenv("HIPX_TMPL_MARCH_UNITS") ? atoi(getenv("HIPX_TMPL_MARCH_UNITS")) : 0;
    d.nt_x        = getenv("HIPX_MARCH_NT_X") ? atoi(getenv("HIPX_MARCH_NT_X")) : 0;
    d.pairmax     = getenv("HIPX_TMPL_PAIRMAX") ? atoi(getenv("HIPX_TMPL_PAIRMAX")) : 16;
    d.nt_stream   = getenv("HIPX_SPMV_NT_STREAM") ? atoi(getenv("HIPX_SPMV_NT_STREAM")) : 0;
    return d;
  }();
  return v;
}

int tmpl_cfg()
{
  static const int v = getenv("HIPX_TMPL_CFG") ? atoi(getenv("HIPX_TMPL_CFG")) : 1;
  return v;
}
int tmpl_blocks()
{
  static const int v = getenv("HIPX_TMPL_BLOCKS") ? atoi(getenv("HIPX_TMPL_BLOCKS")) : 2048;
  return v;
}

// the conditions of spmv_march2_kernel beyond those of the march plan (see the kernel); checked once per pattern + values
template <int NE>
static bool march2_runs_ok(const hipxMarchPlan &mp)
{
  using RS = MarchRuns<NE>;
  if (mp.ne != NE || mp.nlo != RS::NLO || mp.nmid != RS::NMID || mp.b[NE / 2] != 0 || NE / 2 < RS::NLO || NE / 2 >= RS::NLO + RS::NMID) return false;
  for (int r = 0; r < RS::NR; r++)
    for (int d = 1; d < RS::ln(r); d++)
      if (mp.b[RS::st(r) + d] != mp.b[RS::st(r)] + d) return false;
  return true;
}
static int march2_check(hipxMat A)
{
  if (A->march2_state) return HIPX_SUCCESS;
  A->march2_state = -1;
  const hipxMarchPlan &mp = A->march_plan;
  const hipx_int       m  = A->nrows_c;
  if (!A->march_ok || !A->d_tid || !A->d_tmask || A->ntmpl > 256 || A->compressed || m != A->m) return HIPX_SUCCESS;
  // (S % L != 0, round 5: the whole tiles of every plane go to the march kernel, the rest to spmv_march2_rem_kernel -- 256-thread forms with at least one whole tile)
  static const bool norem = getenv("HIPX_MARCH2_NOREM") != nullptr;
  // ... whose rest is at least a halo long: the last whole tile's upper halo then lies inside the plane, as every other tile's does)
  if (mp.S % mp.L && (norem || A->march_nt != 256 || mp.S < mp.L || mp.S % mp.L < mp.H)) return HIPX_SUCCESS;
  if (m % mp.S || (mp.L != 2048 && mp.L != 1024 && !(mp.L == 4096 && A->march_nt == 512)) || (mp.H & 1) || mp.H < 2 || mp.H > (A->march_nt == 512 ? 1024 : 768) || mp.H > mp.L) return HIPX_SUCCESS;
  const int nplanes = (int)(m / mp.S);
  if (nplanes < 3) return HIPX_SUCCESS;
  const bool runs = mp.ne == 7 ? march2_runs_ok<7>(mp) : (mp.ne == 27 ? march2_runs_ok<27>(mp) : (mp.ne == 5 ? march2_runs_ok<5>(mp) : (mp.ne == 9 ? march2_runs_ok<9>(mp) : false)));
  if (!runs) return HIPX_SUCCESS;
  const int nq = mp.L / A->march_nt, nh = (mp.H + A->march_nt - 1) / A->march_nt;  // the instantiated shapes (launch_march2)
  const bool shape = A->march_nt == 512 ? (mp.ne == 7 && ((nq == 8 && nh <= 2) || (nq == 4 && nh == 1))) : ((mp.ne == 7 && nq == 8 && nh <= 2) || (mp.ne == 27 && nq == 8 && nh <= 3) || (nq == 4 && nh == 1));
  if (!shape) return HIPX_SUCCESS;
  unsigned int *d_flag = nullptr, h_flag = 0;
  HIPX_HIP(hipMalloc((void **)&d_flag, sizeof(unsigned int)));
  HIPX_HIP(hipMemsetAsync(d_flag, 0, sizeof(unsigned int), rt().compute));
  if (nplanes > 3) {
    march2_periodic_kernel<<<2048, 256, 0, rt().compute>>>(A->d_tid, (long long)mp.S, nplanes, d_flag);
    HIPX_LAUNCH_CHECK();
  }
  HIPX_HIP(hipMemcpyAsync(&h_flag, d_flag, sizeof(unsigned int), hipMemcpyDeviceToHost, rt().compute));
  HIPX_HIP(hipStreamSynchronize(rt().compute));
  (void)hipFree(d_flag);
  if (!h_flag) A->march2_state = 1;
  return HIPX_SUCCESS;
}

// work split of the march forms: tiles x segments of planes ~ HIPX_TMPL_MARCH_UNITS workgroups (2 per CU resident), segments of >= 8 planes
static void march_geometry(hipxMat A, int &tiles, int &nseg, int &pps, int &nplanes, int &units)
{
  static const int     units_env0 = getenv("HIPX_TMPL_MARCH_UNITS") ? atoi(getenv("HIPX_TMPL_MARCH_UNITS")) : 0;
  const int            units_env = units_env0 ? units_env0 : ((A->march_nt == 512 && A->march_plan.L == 4096) ? 256 : 512);  // (long lines: one 512-thread workgroup per CU)
  const hipxMarchPlan &mp = A->march_plan;
  const hipx_int       m  = A->nrows_c;
  tiles   = (mp.S + mp.L - 1) / mp.L;
  nplanes = (int)((m + mp.S - 1) / mp.S);
  nseg    = std::max(1, std::min(nplanes / 8, (units_env + tiles / 2) / tiles));
  pps     = (nplanes + nseg - 1) / nseg;
  nseg    = (nplanes + pps - 1) / pps;
  units   = tiles * nseg;
}

// hipxMatMultDotBegin / hipxMatMultCGDirectionDotBegin: the reduction slot (and device copy) the product's dot is wanted in; a march2 launch whose
// partials one sum_kernel workgroup would fold folds them itself and says so (HIPX_MARCH_NOFOLD=1: always the separate kernel)
static int     g_m2_fold_slot = -1;
static double *g_m2_fold_dres = nullptr;
static bool    g_m2_folded    = false;
static int     g_m2_extra     = 0;  // dot partials spmv_march2_rem_kernel has written behind the march kernel's (this launch)
static int     g_m2_skip      = 0;  // bit 0 / 1: leave the rows of the first / last plane out of the dot (MPIAIJ diagonal block, hipxMatMultCGDirectionPartial_)
// rows of a plane beyond its whole tiles (0: S is a multiple of L) and the workgroups (= dot partials) of the kernel that takes them
static inline int march2_rem(hipxMat A) { return (A->march_ok && A->march_plan.L > 0) ? A->march_plan.S % A->march_plan.L : 0; }
static inline hipx_int march2_rem_parts(hipxMat A)
{
  const int rem = march2_rem(A);
  return rem ? (hipx_int)(A->nrows_c / A->march_plan.S) * ((rem + 255) / 256) : 0;
}
// the launch of the remainder kernel; tiles / units / xm become those of the whole tiles
template <bool DOT, bool CG>
static int march2_rem_launch(hipxMat A, const double *x, double *yout, double *dotpart, int &units, int &tiles, int nplanes, int &xm, const hipxMarchCG &cg)
{
  g_m2_extra    = 0;
  const int rem = march2_rem(A);
  if (!rem) return HIPX_SUCCESS;
  const hipxMarchPlan &mp   = A->march_plan;
  const int            nseg = units / tiles;
  tiles                     = mp.S / mp.L;
  units                     = tiles * nseg;
  xm                        = (xm & ~1) | ((units % 8 == 0) ? 1 : 0);
  const dim3 grid((unsigned)((rem + 255) / 256), (unsigned)nplanes);
  double    *dp = (DOT && dotpart) ? dotpart + (size_t)units * (A->march_nt / 64) : nullptr;
  if (DOT && dp) spmv_march2_rem_kernel<true, CG><<<grid, 256, 0, rt().compute>>>(mp, A->d_tid, A->d_tmask, x, yout, dp, tiles * mp.L, rem, cg, g_m2_skip);
  else spmv_march2_rem_kernel<false, CG><<<grid, 256, 0, rt().compute>>>(mp, A->d_tid, A->d_tmask, x, yout, nullptr, tiles * mp.L, rem, cg);
  HIPX_LAUNCH_CHECK();
  if (DOT && dp) {
    g_m2_extra        = (int)(grid.x * grid.y);
    A->dot_npart_used = (hipx_int)units * (A->march_nt / 64) + g_m2_extra;
  }
  return HIPX_SUCCESS;
}
template <int NE, int NQ, int NHALO, bool DOT, bool CG = false, int NT = 256>
static int launch_march2_inst(hipxMat A, const double *x, double *yout, double *dotpart, int units, int tiles, int pps, int nplanes, int xm, size_t lds, const hipxMarchCG &cg = hipxMarchCG{})
{
  static bool attr = false;
  if (!attr) {
    HIPX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&spmv_march2_kernel<NE, NQ, NHALO, DOT, CG, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, NT == 512 ? 152 * 1024 : 96 * 1024));
    attr = true;
  }
  RedOut red{};
  if (DOT && g_m2_fold_slot >= 0 && (hipx_int)units * (NT / 64) + g_m2_extra <= (hipx_int)kRedThreads * 8) {  // (= red_grid(npart) == 1: the fold rides in the kernel's last workgroup)
    red          = red_out(g_m2_fold_slot, true, g_m2_fold_dres);
    g_m2_folded  = true;
  }
  spmv_march2_kernel<NE, NQ, NHALO, DOT, CG, NT><<<(unsigned)units, NT, lds, rt().compute>>>(A->march_plan, A->d_tid, A->d_tmask, A->ntmpl, x, yout, dotpart, tiles, pps, nplanes, xm | (dev_sw().nt_store ? 2 : 0) | (4 * dev_sw().nt_x) | (DOT ? 16 * g_m2_skip : 0), cg, red, g_m2_extra);
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}
// the shapes the CG prologue is instantiated for (short templates: the long ones have no registers left for two more streams in flight)
static bool march2_cg_shape(hipxMat A)
{
  const hipxMarchPlan &mp = A->march_plan;
  const int            nq = mp.L / A->march_nt, nh = (mp.H + A->march_nt - 1) / A->march_nt;
  if (A->march_nt == 512) return mp.ne == 7 && ((nq == 8 && nh <= 2) || (nq == 4 && nh == 1));
  // the 27-entry kernels too (242-252 registers, no spills): their product is VALU-bound, so the prologue's streams ride along for free --
  // 27-pt 512^3 CG+Jacobi 450 -> 556 it/s, 27-pt 256^3 3.55-3.62k -> 4.27k (same box; HIPX_MARCH_NOCG27 keeps the direction kernel)
  static const bool nocg27 = getenv("HIPX_MARCH_NOCG27") != nullptr;
  if (!nocg27 && mp.ne == 27 && nq == 8 && nh <= 3) return true;
  return (mp.ne == 7 && nq == 8 && nh <= 2) || ((mp.ne == 7 || mp.ne == 5 || mp.ne == 9) && nq == 4 && nh == 1);
}
template <bool DOT>
static int launch_march2_cg(hipxMat A, const double *x, double *yout, double *dotpart, int units, int tiles, int pps, int nplanes, int xm, size_t lds, const hipxMarchCG &cg)
{
  if (int ierr = march2_rem_launch<DOT, true>(A, x, yout, dotpart, units, tiles, nplanes, xm, cg)) return ierr;
  const hipxMarchPlan &mp = A->march_plan;
  const int            nq = mp.L / A->march_nt, nh = (mp.H + A->march_nt - 1) / A->march_nt;
  if (A->march_nt == 512) {
    if (mp.ne == 7 && nq == 8 && nh == 1) return launch_march2_inst<7, 8, 1, DOT, true, 512>(A, x, yout, dotpart, units, tiles, pps, nplanes, xm, lds, cg);
    if (mp.ne == 7 && nq == 8 && nh == 2) return launch_march2_inst<7, 8, 2, DOT, true, 512>(A, x, yout, dotpart, units, tiles, pps, nplanes, xm, lds, cg);
    if (mp.ne == 7 && nq == 4 && nh == 1) return launch_march2_inst<7, 4, 1, DOT, true, 512>(A, x, yout, dotpart, units, tiles, pps, nplanes, xm, lds, cg);
    return fail(HIPX_ERR_ARG, "march2 (CG prologue, 512 threads): shape not instantiated", __FILE__, __LINE__);
  }
#define HIPX_M2(NE, NQ, NH) return launch_march2_inst<NE, NQ, NH, DOT, true>(A, x, yout, dotpart, units, tiles, pps, nplanes, xm, lds, cg)
  if (mp.ne == 7 && nq == 8 && nh == 1) HIPX_M2(7, 8, 1);
  if (mp.ne == 7 && nq == 8 && nh == 2) HIPX_M2(7, 8, 2);
  if (mp.ne == 7 && nq == 4 && nh == 1) HIPX_M2(7, 4, 1);
  if (mp.ne == 5 && nq == 4 && nh == 1) HIPX_M2(5, 4, 1);
  if (mp.ne == 9 && nq == 4 && nh == 1) HIPX_M2(9, 4, 1);
  if (mp.ne == 27 && nq == 8 && nh == 1) HIPX_M2(27, 8, 1);  // (lines of up to 254 points on planes that are not whole tiles: 200^3)
  if (mp.ne == 27 && nq == 8 && nh == 2) HIPX_M2(27, 8, 2);
  if (mp.ne == 27 && nq == 8 && nh == 3) HIPX_M2(27, 8, 3);
#undef HIPX_M2
  return fail(HIPX_ERR_ARG, "march2 (CG prologue): shape not instantiated", __FILE__, __LINE__);
}
template <bool DOT>
static int launch_march2(hipxMat A, const double *x, double *yout, double *dotpart, int units, int tiles, int pps, int nplanes, int xm, size_t lds)
{
  if (int ierr = march2_rem_launch<DOT, false>(A, x, yout, dotpart, units, tiles, nplanes, xm, hipxMarchCG{})) return ierr;
  const hipxMarchPlan &mp = A->march_plan;
  const int            nq = mp.L / A->march_nt, nh = (mp.H + A->march_nt - 1) / A->march_nt;
  if (A->march_nt == 512) {
    if (mp.ne == 7 && nq == 8 && nh == 1) return launch_march2_inst<7, 8, 1, DOT, false, 512>(A, x, yout, dotpart, units, tiles, pps, nplanes, xm, lds);
    if (mp.ne == 7 && nq == 8 && nh == 2) return launch_march2_inst<7, 8, 2, DOT, false, 512>(A, x, yout, dotpart, units, tiles, pps, nplanes, xm, lds);
    if (mp.ne == 7 && nq == 4 && nh == 1) return launch_march2_inst<7, 4, 1, DOT, false, 512>(A, x, yout, dotpart, units, tiles, pps, nplanes, xm, lds);
    return fail(HIPX_ERR_ARG, "march2 (512 threads): shape not instantiated", __FILE__, __LINE__);
  }
#define HIPX_M2(NE, NQ, NH) return launch_march2_inst<NE, NQ, NH, DOT>(A, x, yout, dotpart, units, tiles, pps, nplanes, xm, lds)
  if (mp.ne == 7 && nq == 8 && nh == 1) HIPX_M2(7, 8, 1);
  if (mp.ne == 7 && nq == 8 && nh == 2) HIPX_M2(7, 8, 2);
  if (mp.ne == 27 && nq == 8 && nh == 1) HIPX_M2(27, 8, 1);
  if (mp.ne == 27 && nq == 8 && nh == 2) HIPX_M2(27, 8, 2);
  if (mp.ne == 27 && nq == 8 && nh == 3) HIPX_M2(27, 8, 3);
  if (mp.ne == 7 && nq == 4 && nh == 1) HIPX_M2(7, 4, 1);
  if (mp.ne == 27 && nq == 4 && nh == 1) HIPX_M2(27, 4, 1);
  if (mp.ne == 5 && nq == 4 && nh == 1) HIPX_M2(5, 4, 1);
  if (mp.ne == 9 && nq == 4 && nh == 1) HIPX_M2(9, 4, 1);
#undef HIPX_M2
  return fail(HIPX_ERR_ARG, "march2: shape not instantiated", __FILE__, __LINE__);
}

// hipxMatMultChebyshev: the epilogue the next pair-form launch (MODE 0, no dot) applies; g_epi_done tells the caller it did
static hipxPairEpi g_epi;
static bool        g_epi_on = false, g_epi_done = false;

template <int MODE, bool DOT>
int launch_tmpl(hipxMat A, const double *x, const double *yin, double *yout, double *dotpart, hipx_int *npart)
{
  const int      cfg = tmpl_cfg();
  const int      rpt = (cfg == 2) ? 4 : 2;
  const hipx_int m = A->nrows_c;
  hipx_int       nchunks = (m + 256 * rpt - 1) / (256 * rpt);
  if (npart) {
    *npart = nchunks * 4;  // one dot partial per wave and chunk
    return HIPX_SUCCESS;
  }
  static const bool nosub = getenv("HIPX_TMPL_NOSUB") != nullptr;
  const int      tbase = (A->d_tmask && !nosub) ? A->tmpl_base : -1;
  // pair form (spmv_pair_kernel): sub-template matrices, 16-byte aligned vectors, the default geometry; it takes the whole chunks
  static const bool nopair = getenv("HIPX_TMPL_NOPAIR") != nullptr;
  static const int  pair_maxp = getenv("HIPX_TMPL_PAIRMAX") ? atoi(getenv("HIPX_TMPL_PAIRMAX")) : 16;  // most pairs a base template may have for the pair form
  static const int  probe0 = getenv("HIPX_TMPL_PROBE") ? atoi(getenv("HIPX_TMPL_PROBE")) : 0;
  const bool     vec_aligned = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(yout) | (MODE == 1 ? reinterpret_cast<uintptr_t>(yin) : (uintptr_t)0) |
                                 ((MODE == 0 && !DOT && g_epi_on) ? (reinterpret_cast<uintptr_t>(g_epi.b) | reinterpret_cast<uintptr_t>(g_epi.pprev) | reinterpret_cast<uintptr_t>(g_epi.dinv)) : (uintptr_t)0)) & 15) == 0;
  const bool     use_pair = A->pair_ok && A->pair_plan.npairs <= pair_maxp && tbase >= 0 && !nopair && !probe0 && cfg == 1 && vec_aligned && m >= 512 && A->ntmpl <= 256;
  if (DOT) A->dot_npart_used = 0;
  // march form (spmv_march_kernel): three-plane base templates, 16-byte aligned x, enough tiles x segments to fill the chip
  {
    static const bool nomarch = getenv("HIPX_TMPL_NOMARCH") != nullptr;
    const hipxMarchPlan &mp = A->march_plan;
    bool                 big_ok = true;  // plans that only the 512-thread form of spmv_march2_kernel can run (long lines): MatMult of a matrix that passed its checks
    if (A->march_ok && A->march_nt == 512) {
      big_ok = false;
      if (MODE == 0 && !dev_sw().march1 && !dev_sw().trace) {
        int ierr2 = march2_check(A);
        if (ierr2) return ierr2;
        big_ok = A->march2_state == 1 && (reinterpret_cast<uintptr_t>(yout) & 7) == 0;
      }
    }
    if (A->march_ok && big_ok && tbase >= 0 && !nomarch && !probe0 && cfg == 1 && !g_epi_on && A->ntmpl <= 256 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && m % 2 == 0) {
      int tiles, nseg, pps, nplanes, units;  // tiles x segments of >= 8 planes (a segment's two extra plane loads stay <= 25 %)
      march_geometry(A, tiles, nseg, pps, nplanes, units);
      if ((units >= 192 || A->march_force) && (hipx_int)units <= nchunks && mp.ne <= 32) {
        const size_t lds = 3 * (size_t)(mp.L + 2 * mp.H) * sizeof(double);
        const int    xm  = (units % 8 == 0) ? 1 : 0;
        if (DOT) A->dot_npart_used = (hipx_int)units * (A->march_nt / 64);
        // the long templates run the kernel WITH the dot also when nobody wants it (the partials go to a dump): without it the compiler schedules the
        // 27-entry loop into 80 registers of scratch per lane (312 bytes; with the dot: none)
        if (!DOT && mp.ne > 9 && !A->d_march_dump) HIPX_HIP(hipMalloc((void **)&A->d_march_dump, sizeof(double) * 4 * 8192));
        if (!DOT && mp.ne > 9 && units > 8192) return fail(HIPX_ERR_ARG, "march form: more than 8192 workgroups", __FILE__, __LINE__);
#define HIPX_MARCH_LAUNCH(NE, EX)                                                                                                                                                      \
  do {                                                                                                                                                                             \
    static bool attr = false;                                                                                                                                                      \
    if (!attr) {                                                                                                                                                                   \
      HIPX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&spmv_march_kernel<MODE, (DOT || (NE > 9)), NE, EX>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));   \
      attr = true;                                                                                                                                                                 \
    }                                                                                                                                                                              \
    spmv_march_kernel<MODE, (DOT || (NE > 9)), NE, EX><<<(unsigned)units, 256, lds, rt().compute>>>(m, mp, A->d_tid, A->d_tmask, A->ntmpl, x, yin, yout, DOT ? dotpart : A->d_march_dump, tiles, nseg, pps, nplanes, xm); \
  } while (0)
        static const bool mtrace = getenv("HIPX_TMPL_TRACE") != nullptr;
        static const bool march1 = getenv("HIPX_MARCH1") != nullptr;  // developer switch: keep the first march kernel (same-box A/B timing)
        if (MODE == 0 && !mtrace && !march1 && (reinterpret_cast<uintptr_t>(yout) & 7) == 0) {
          if constexpr (MODE == 0) {
            int ierr2 = march2_check(A);
            if (ierr2) return ierr2;
            if (A->march2_state == 1) return launch_march2<DOT>(A, x, yout, dotpart, units, tiles, pps, nplanes, xm, lds);
          }
        }
        if (mtrace && MODE == 0 && DOT) {  // developer timing: the steps of workgroups 8 and 301 (10 ns ticks) on stderr, twice
          if constexpr (MODE == 0 && DOT) {
            static unsigned long long *d_tr = nullptr;
            static int                 nl = 0, dumps = 0;
            if (!d_tr) HIPX_HIP(hipMalloc((void **)&d_tr, 80 * 8 * sizeof(unsigned long long)));
            HIPX_HIP(hipMemsetAsync(d_tr, 0, 80 * 8 * sizeof(unsigned long long), rt().compute));
            static bool attr2 = false;
            if (!attr2) {
              HIPX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&spmv_march_kernel<0, true, 7, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
              HIPX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&spmv_march_kernel<0, true, 27, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
              attr2 = true;
            }
            if (mp.ne != 7 && mp.ne != 27) return fail(HIPX_ERR_ARG, "HIPX_TMPL_TRACE: the march-form trace is built for 7 and 27 entries", __FILE__, __LINE__);
            if (mp.ne == 7) spmv_march_kernel<0, true, 7, true, true><<<(unsigned)units, 256, lds, rt().compute>>>(m, mp, A->d_tid, A->d_tmask, A->ntmpl, x, yin, yout, dotpart, tiles, nseg, pps, nplanes, xm, d_tr);
            else spmv_march_kernel<0, true, 27, true, true><<<(unsigned)units, 256, lds, rt().compute>>>(m, mp, A->d_tid, A->d_tmask, A->ntmpl, x, yin, yout, dotpart, tiles, nseg, pps, nplanes, xm, d_tr);
            HIPX_LAUNCH_CHECK();
            if (++nl >= 6 && dumps < 2) {
              unsigned long long h[80 * 8];
              HIPX_HIP(hipStreamSynchronize(rt().compute));
              HIPX_HIP(hipMemcpy(h, d_tr, sizeof(h), hipMemcpyDeviceToHost));
              for (int w = 0; w < 2; w++)
                for (int e = 0; e < 40 && h[(w * 40 + e) * 8]; e++) {
                  const unsigned long long *o = h + (w * 40 + e) * 8;
                  fprintf(stderr, "[hipx march trace] wg %3d plane %4llu start %8.2f us | wait+lds-store %5.2f barrier1 %5.2f issue %5.2f compute %5.2f barrier2 %5.2f | step %5.2f us\n", w ? 301 : 8, o[6],
                          (o[0] - h[0]) / 100.0, (o[1] - o[0]) / 100.0, (o[2] - o[1]) / 100.0, (o[3] - o[2]) / 100.0, (o[4] - o[3]) / 100.0, (o[5] - o[4]) / 100.0, (o[5] - o[0]) / 100.0);
                }
              dumps++;
            }
            return HIPX_SUCCESS;
          }
        }
        if (mp.ne == 7) HIPX_MARCH_LAUNCH(7, true);
        else if (mp.ne == 27) HIPX_MARCH_LAUNCH(27, true);
        else if (mp.ne == 5) HIPX_MARCH_LAUNCH(5, true);
        else if (mp.ne == 9) HIPX_MARCH_LAUNCH(9, true);
        else if (mp.ne <= 8) HIPX_MARCH_LAUNCH(8, false);
        else HIPX_MARCH_LAUNCH(32, false);
#undef HIPX_MARCH_LAUNCH
        HIPX_LAUNCH_CHECK();
        return HIPX_SUCCESS;
      }
    }
  }
  if (use_pair) nchunks = m / 512;
  hipx_int       grid = std::min<hipx_int>((hipx_int)((tmpl_blocks() + 7) / 8 * 8), ((nchunks + 7) / 8) * 8);
  if (grid < 8) grid = 8;
  const hipx_int cpx  = (nchunks + 7) / 8;
  const size_t   smem = 8 * (size_t)((A->tmpl_nent + 1) & ~1) + 4 * (size_t)((A->tmpl_nent + 3) & ~3) + 4 * ((size_t)A->ntmpl + 1) + 4 * (size_t)A->ntmpl + 16;
  const int      geom = (int)grid * 16 + (use_pair ? 9 : rpt);
  if (!A->d_tq || A->tq_geom != geom) {  // ticket counters of the chunk queue (zeroed once per geometry)
    if (!A->d_tq) HIPX_HIP(hipMalloc((void **)&A->d_tq, 8 * 64));
    HIPX_HIP(hipMemsetAsync(A->d_tq, 0, 8 * 64, rt().compute));
    A->tq_launches = 0;
    A->tq_geom     = geom;
  }
  const unsigned long long launch = A->tq_launches++;
  // first-touch prefetch offset: the largest forward offset of the most common template, when it lies beyond a chunk (else off)
  long long pf_off = 0;
  {
    static const bool off = getenv("HIPX_TMPL_NOPF") != nullptr;
    if (A->tmpl_maxoff < 0) {
      int best = 0;
      for (int tt = 1; tt < A->ntmpl; tt++)
        if (A->h_tcount[(size_t)tt] > A->h_tcount[(size_t)best]) best = tt;
      A->tmpl_maxoff = 0;
      for (int k = A->h_tstart[(size_t)best]; k < A->h_tstart[(size_t)best + 1]; k++) A->tmpl_maxoff = std::max<long long>(A->tmpl_maxoff, A->h_toff[(size_t)k]);
    }
    pf_off = A->tmpl_maxoff;
    static const char *ds = getenv("HIPX_TMPL_PFDIST");  // look-ahead in chunks; default: one round of the XCD's workgroups
    const long long    dist = ds ? atoll(ds) : (long long)(grid >> 3);
    if (off || pf_off < 4 * 256 * rpt) pf_off = 0;
    else pf_off += dist * 256 * rpt;
  }
  if (use_pair) {
    // chunks per ticket (HIPX_TMPL_TG = 1 | 2 | 4 | 8): measured 7-pt 256^3 0.1212 / 0.1241 / 0.1320 ms for 1 / 2 / 4, 7-pt 512^3 0.864 / 0.790 / 0.797:
    // one chunk per ticket while the slabs are short, two on large matrices
    static const int tg_env = getenv("HIPX_TMPL_TG") ? atoi(getenv("HIPX_TMPL_TG")) : 0;
    const int        tg = (tg_env == 1 || tg_env == 2 || tg_env == 4 || tg_env == 8) ? tg_env : (nchunks > 49152 ? 2 : 1);
    if (MODE == 0 && !DOT && g_epi_on) {  // SpMV + Chebyshev step in one kernel
      if constexpr (MODE == 0 && !DOT) {
        if (A->pair_plan.npairs <= 8)
          spmv_pair_kernel<0, false, 8, false, 1><<<(unsigned)grid, 256, 0, rt().compute>>>(m, nchunks, cpx, A->d_tid, A->d_tmask, A->ntmpl, A->pair_plan, x, yin, yout, dotpart, A->d_tq, launch, pf_off, tg, nullptr, g_epi);
        else if (A->pair_plan.npairs <= 12)
          spmv_pair_kernel<0, false, 12, false, 1><<<(unsigned)grid, 256, 0, rt().compute>>>(m, nchunks, cpx, A->d_tid, A->d_tmask, A->ntmpl, A->pair_plan, x, yin, yout, dotpart, A->d_tq, launch, pf_off, tg, nullptr, g_epi);
        else
          spmv_pair_kernel<0, false, 16, false, 1><<<(unsigned)grid, 256, 0, rt().compute>>>(m, nchunks, cpx, A->d_tid, A->d_tmask, A->ntmpl, A->pair_plan, x, yin, yout, dotpart, A->d_tq, launch, pf_off, tg, nullptr, g_epi);
        HIPX_LAUNCH_CHECK();
        if (nchunks * 512 < m) {
          spmv_tmpl_tail_kernel<0, false, 1><<<1, 256, 0, rt().compute>>>(m, nchunks * 512, nchunks, A->d_tid, A->d_tstart, A->d_toff, A->d_tval, x, yin, yout, dotpart, g_epi);
          HIPX_LAUNCH_CHECK();
        }
        g_epi_done = true;
        return HIPX_SUCCESS;
      }
    }
    static const bool tracing = getenv("HIPX_TMPL_TRACE") != nullptr;
    if (tracing) {  // developer timing: passes of workgroups 8 and 1032 (start, issued, loads back, stored, barrier 1, barrier 2; 10 ns ticks) on stderr
      static unsigned long long *d_tr = nullptr;
      const size_t trw = 128 * 8 + 4 * 4096;
      if (!d_tr) HIPX_HIP(hipMalloc((void **)&d_tr, trw * sizeof(unsigned long long)));
      HIPX_HIP(hipMemsetAsync(d_tr, 0, trw * sizeof(unsigned long long), rt().compute));
      spmv_pair_kernel<MODE, DOT, 8, true><<<(unsigned)grid, 256, 0, rt().compute>>>(m, nchunks, cpx, A->d_tid, A->d_tmask, A->ntmpl, A->pair_plan, x, yin, yout, dotpart, A->d_tq, launch, pf_off, tg, d_tr);
      HIPX_LAUNCH_CHECK();
      static int dumps = 0;
      if (dumps < 2 && launch >= 5) {
        unsigned long long h[128 * 8];
        HIPX_HIP(hipStreamSynchronize(rt().compute));
        HIPX_HIP(hipMemcpy(h, d_tr, sizeof(h), hipMemcpyDeviceToHost));
        for (int w = 0; w < 2; w++)
          for (int e = 0; e < 64 && h[(w * 64 + e) * 8]; e++) {
            const unsigned long long *o = h + (w * 64 + e) * 8;
            fprintf(stderr, "[hipx tmpl trace] wg %d pass %2d chunk %6llu start %8.2f us | issue %5.2f loads %5.2f store %5.2f barrier1 %5.2f barrier2 %5.2f | pass %5.2f us\n", w ? 1032 : 8, e, o[6],
                    (o[0] - h[0]) / 100.0, (o[1] - o[0]) / 100.0, (o[2] - o[1]) / 100.0, (o[3] - o[2]) / 100.0, (o[4] - o[3]) / 100.0, (o[5] - o[4]) / 100.0, (o[5] - o[0]) / 100.0);
          }
        {  // passes per workgroup and the span it was alive: by XCD, and the spread
          std::vector<unsigned long long> w(4 * 4096);
          HIPX_HIP(hipMemcpy(w.data(), d_tr + 128 * 8, w.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
          unsigned long long t0 = ~0ull;
          for (hipx_int b = 0; b < grid && b < 4096; b++)
            if (w[4 * b + 1]) t0 = std::min(t0, w[4 * b + 1]);
          for (int xc = 0; xc < 8; xc++) {
            unsigned long long np = 0, mn = ~0ull, mx = 0, late = 0, nw = 0;
            for (hipx_int b = xc; b < grid && b < 4096; b += 8) {
              np += w[4 * b];
              if (w[4 * b]) {
                mn = std::min(mn, w[4 * b]);
                mx = std::max(mx, w[4 * b]);
                nw++;
              }
              if (w[4 * b + 1] - t0 > 1000) late++;  // started more than 10 us after the first workgroup
            }
            fprintf(stderr, "[hipx tmpl trace] XCD %d: %llu passes by %llu workgroups (min %llu, max %llu per workgroup), %llu workgroups started > 10 us late\n", xc, np, nw, mn, mx, late);
          }
          for (hipx_int b = 0; b < 64 && b < grid; b++)
            fprintf(stderr, "[hipx tmpl trace] wg %4d: %3llu passes, start %7.2f us, end %7.2f us\n", (int)(b * 8), w[4 * (b * 8)], (w[4 * (b * 8) + 1] - t0) / 100.0, (w[4 * (b * 8) + 2] - t0) / 100.0);
        }
        dumps++;
      }
    } else
    if (A->pair_plan.npairs <= 8)
      spmv_pair_kernel<MODE, DOT, 8><<<(unsigned)grid, 256, 0, rt().compute>>>(m, nchunks, cpx, A->d_tid, A->d_tmask, A->ntmpl, A->pair_plan, x, yin, yout, dotpart, A->d_tq, launch, pf_off, tg);
    else if (A->pair_plan.npairs <= 12)
      spmv_pair_kernel<MODE, DOT, 12><<<(unsigned)grid, 256, 0, rt().compute>>>(m, nchunks, cpx, A->d_tid, A->d_tmask, A->ntmpl, A->pair_plan, x, yin, yout, dotpart, A->d_tq, launch, pf_off, tg);
    else
      spmv_pair_kernel<MODE, DOT, 16><<<(unsigned)grid, 256, 0, rt().compute>>>(m, nchunks, cpx, A->d_tid, A->d_tmask, A->ntmpl, A->pair_plan, x, yin, yout, dotpart, A->d_tq, launch, pf_off, tg);
    HIPX_LAUNCH_CHECK();
    if (nchunks * 512 < m) {
      spmv_tmpl_tail_kernel<MODE, DOT><<<1, 256, 0, rt().compute>>>(m, nchunks * 512, nchunks, A->d_tid, A->d_tstart, A->d_toff, A->d_tval, x, yin, yout, dotpart);
      HIPX_LAUNCH_CHECK();
    }
    return HIPX_SUCCESS;
  }
#define HIPX_TMPL_LAUNCH(R, WW, U) \
  spmv_tmpl_kernel<MODE, DOT, R, WW, U><<<(unsigned)grid, 256, smem, rt().compute>>>(m, nchunks, cpx, A->d_tid, A->d_tstart, A->d_toff, A->d_tval, A->ntmpl, A->tmpl_nent, x, yin, yout, dotpart, \
                                                                                      A->d_tq, launch, pf_off, A->d_tmask, tbase)
  switch (cfg) {
  case 0: HIPX_TMPL_LAUNCH(2, 4, false); break;
  case 2: HIPX_TMPL_LAUNCH(4, 4, true); break;
  default: {
    static const int probe = getenv("HIPX_TMPL_PROBE") ? atoi(getenv("HIPX_TMPL_PROBE")) : 0;
    if (probe >= 1 && probe <= 5) {
#define HIPX_TMPL_LAUNCH_P(PR) \
  spmv_tmpl_kernel<MODE, DOT, 2, 2, true, PR><<<(unsigned)grid, 256, smem, rt().compute>>>(m, nchunks, cpx, A->d_tid, A->d_tstart, A->d_toff, A->d_tval, A->ntmpl, A->tmpl_nent, x, yin, yout, \
                                                                                           dotpart, A->d_tq, launch, pf_off, A->d_tmask, tbase)
      if (probe == 1) HIPX_TMPL_LAUNCH_P(1);
      else if (probe == 2) HIPX_TMPL_LAUNCH_P(2);
      else if (probe == 3) HIPX_TMPL_LAUNCH_P(3);
      else if (probe == 4) HIPX_TMPL_LAUNCH_P(4);
      else HIPX_TMPL_LAUNCH_P(5);
#undef HIPX_TMPL_LAUNCH_P
    } else HIPX_TMPL_LAUNCH(2, 2, true);
    break;
  }
  }
#undef HIPX_TMPL_LAUNCH
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}

// does the next product run the template kernel?  (builds the templates if they are pending)
int use_templates(hipxMat A, bool &use)
{
  use = false;
  static const bool off = getenv("HIPX_NO_TMPL") != nullptr;
  if (!A->tmpl_mode || A->compressed || A->probe || (off && A->auto_sel)) return HIPX_SUCCESS;
  int ierr;
  if (A->auto_sel) {  // auto: only matrices whose values already fit the 256-entry dictionary are worth the hashing passes
    if ((ierr = ensure_vdict(A))) return ierr;
    setup_trace("  value dictionary");
    if (!A->vd_ok) return HIPX_SUCCESS;
  }
  if ((ierr = ensure_templates(A))) return ierr;
  use = A->tmpl_ok;
  return HIPX_SUCCESS;
}

// which packed form the next launch takes: vd (dictionary kernel), rowpar (row-parallel gather), rpt (rows per thread of
// spmv_vd_kernel, 0 = other kernels), cfg (row-block geometry the packed format is built on)
int select_pk(hipxMat A, bool &vd, bool &rowpar, int &rpt, int &cfg)
{
  // tuning hook: rows per thread of the dictionary kernel.  Measured on MI355X (7-pt 256^3 / 27-pt 160^3, ms):
  // 1 row: 0.188 / 0.191, 2 rows: 0.167 / 0.176 (default), 4 rows: 0.172 / 0.204
  static const int vd_rpt = getenv("HIPX_VD_RPT") ? atoi(getenv("HIPX_VD_RPT")) : 2;
  vd = false;
  if (A->vd_mode) {
    int ierr = ensure_vdict(A);
    if (ierr) return ierr;
    vd = A->vd_ok;
  }
  rowpar = A->tile_mode >= 3 || (vd && A->auto_sel);  // with the dictionary the row-parallel form wins for long rows too
  rpt    = (vd && rowpar) ? ((vd_rpt == 1 || vd_rpt == 4) ? vd_rpt : 2) : 0;
  cfg    = rpt == 2 ? 1 : rpt == 4 ? 6 : 0;
  return ensure_pk16(A, cfg);
}

template <typename IT, int MODE, bool DOT>
int launch_pk16(hipxMat A, const double *x, const double *yin, double *yout, double *dotpart)
{
  bool vd, rowpar;
  int  rpt, cfg;
  int  ierr = select_pk(A, vd, rowpar, rpt, cfg);
  if (ierr) return ierr;
  const hipx_int nb = A->nblocks[cfg];
  if (nb == 0) return HIPX_SUCCESS;
  const hipx_int per_xcd = (nb + 7) / 8;
  const unsigned grid = (unsigned)(per_xcd * 8);
#define HIPX_PK_ARGS A->d_rb[0], nb, per_xcd, (const IT *)A->d_i, A->d_j, A->d_pk, A->d_pkbase, A->d_a
#define HIPX_VD_LAUNCH(R, WW, D) \
  spmv_vd_kernel<IT, MODE, DOT, R, WW, D><<<grid, 256, 0, rt().compute>>>((const PkDesc *)A->d_pkdesc, nb, per_xcd, (const IT *)A->d_i, A->d_j, A->d_pk, A->d_pkbase, A->d_a, A->d_vc, \
                                                                            A->d_vdict, A->vd_count, x, yin, yout, dotpart, A->n)
  if (rpt > 0) {
    const int probe = (MODE == 0 && !DOT) ? A->probe : 0;
    if (probe) {
      if constexpr (MODE == 0 && !DOT) {
        if (probe == 1) HIPX_VD_LAUNCH(2, 4, 1);
        else if (probe == 2) HIPX_VD_LAUNCH(2, 4, 2);
        else HIPX_VD_LAUNCH(2, 4, 3);
      }
    } else if (rpt == 1) HIPX_VD_LAUNCH(1, 8, 0);
    else if (rpt == 4) HIPX_VD_LAUNCH(4, 2, 0);
    else HIPX_VD_LAUNCH(2, 4, 0);
  } else if (rowpar) {
    spmv_pk16r_kernel<IT, MODE, DOT><<<grid, 256, 0, rt().compute>>>(HIPX_PK_ARGS, x, yin, yout, dotpart, A->n, dev_sw().nt_stream);
  } else {
    if (vd) spmv_pk16_kernel<IT, MODE, DOT, true><<<grid, 256, 0, rt().compute>>>(HIPX_PK_ARGS, A->d_vc, A->d_vdict, x, yin, yout, dotpart, A->n);
    else spmv_pk16_kernel<IT, MODE, DOT, false><<<grid, 256, 0, rt().compute>>>(HIPX_PK_ARGS, A->d_vc, A->d_vdict, x, yin, yout, dotpart, A->n);
  }
#undef HIPX_PK_ARGS
#undef HIPX_VD_LAUNCH
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}

// variant 29: pattern templates + streamed values (spmv_tp_kernel) when the matrix has <= 256 row patterns
int use_pattern_templates(hipxMat A, bool &use)
{
  use = false;
  if (!A->ptm_mode || A->compressed || A->probe) return HIPX_SUCCESS;
  int ierr;
  if (A->auto_sel) {  // auto: short rows only (long rows stream better through the staged kernel), and only when the values do not fit the
                      // 8-bit dictionary (3 bytes per nonzero beat 8)
    if (A->nnz < (1 << 20) || A->nnz > 16 * (int64_t)A->nrows_c) return HIPX_SUCCESS;
    if ((ierr = ensure_vdict(A))) return ierr;
    if (A->vd_ok) return HIPX_SUCCESS;
  }
  if ((ierr = ensure_pattern_templates(A))) return ierr;
  if (!A->ptm_ok) return HIPX_SUCCESS;
  if ((ierr = ensure_row_blocks(A, 0))) return ierr;
  use = true;
  return HIPX_SUCCESS;
}

template <typename IT, int MODE, bool DOT>
int launch_tp(hipxMat A, const double *x, const double *yin, double *yout, double *dotpart)
{
  const hipx_int nb = A->nblocks[0];
  if (nb == 0) return HIPX_SUCCESS;
  const hipx_int per_xcd = (nb + 7) / 8;
  const unsigned grid = (unsigned)(per_xcd * 8);
  spmv_tp_kernel<IT, MODE, DOT><<<grid, 256, 0, rt().compute>>>(A->d_rb[0], nb, per_xcd, (const IT *)A->d_i, A->d_ptid, A->d_ptstart, A->d_ptoff, A->d_a, x, yin, yout, dotpart, dev_sw().nt_stream);
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}

// variant 28: the SELL-64 copy (hipx_sell.hip) when it applies; *use = false -> the CSR kernels
int use_sell(hipxMat A, bool &use, bool auto_stage = false)
{
  use = false;
  if (!A->sell_mode || A->compressed || A->probe) return HIPX_SUCCESS;
  if ((A->sell_mode == 2) != auto_stage) return HIPX_SUCCESS;  // explicit (variant 28): ahead of everything; auto: behind the template forms
  if (auto_stage) {  // auto: long rows with arbitrary values (FEM): measured 4-8 % ahead of the staged packed kernel, a tie on 27-point rows
    if (A->nnz < (1 << 20) || A->nnz <= 32 * (int64_t)A->nrows_c) return HIPX_SUCCESS;
    int ierr = ensure_vdict(A);
    if (ierr) return ierr;
    if (A->vd_ok) return HIPX_SUCCESS;
  }
  int     ok = 0, packed = 0;
  double  pad = 0.0;
  int64_t bytes = 0;
  int     ierr = hipxSellEnsure_(A, &A->sell_state, &ok, &packed, &pad, &bytes);
  if (ierr) return ierr;
  use = ok != 0;
  return HIPX_SUCCESS;
}

// A matrix with inodes: the reference multiplies it with MatMult_SeqAIJ_Inode / MatMultAdd_SeqAIJ_Inode (aij.c:1459, 1617; inode.c:356-760),
// whose row sums take the terms in pairs -- rounding-level differences from MatMult_SeqAIJ's left-to-right sums, so these matrices keep
// to the two kernel forms that know that order: the SELL-64 copy (what long FEM rows get anyway) or, where it does not apply, a plain
// row-per-lane CSR kernel.  The kernel variants of hipxMatSetSpMVVariant are forms of the LEFT-TO-RIGHT sum: they apply once the matrix
// is declared free of inodes (hipxMatSetInodes(A, 0, NULL) = -mat_no_inode).
int use_inode_pair(hipxMat A, bool &use, bool &sell)
{
  use = sell = false;
  if (A->compressed || A->probe) return HIPX_SUCCESS;
  if (getenv("HIPX_MAT_NO_INODE")) return HIPX_SUCCESS;  // developer switch, read per call (tests flip it): every matrix as under -mat_no_inode
  int ierr;
  if (A->inode_state < 0 && (ierr = hipxMatEnsureInodes_(A))) return ierr;
  if (A->inode_state != 1) return HIPX_SUCCESS;
  use = true;
  {
    int     ok = 0, packed = 0;
    double  pad = 0.0;
    int64_t bytes = 0;
    if ((ierr = hipxSellEnsure_(A, &A->sell_state, &ok, &packed, &pad, &bytes))) return ierr;
    sell = ok != 0;
  }
  return HIPX_SUCCESS;
}

template <typename IT, int MODE, bool DOT>
__global__ __launch_bounds__(256) void spmv_inode_kernel(hipx_int m, const IT *__restrict__ ai, const hipx_int *__restrict__ aj, const double *__restrict__ aa, const double *__restrict__ x,
                                                        const double *yin, double *yout, double *dotpart)
{
  const hipx_int row = (hipx_int)blockIdx.x * 256 + threadIdx.x;
  double         sum = 0.0, p = 0.0;
  if (row < m) {
    const int64_t s = (int64_t)ai[row], e = (int64_t)ai[row + 1];
    if (MODE == 1) sum = yin[row];
    int64_t k = s;
    for (; k + 1 < e; k += 2) sum += aa[k] * x[aj[k]] + aa[k + 1] * x[aj[k + 1]];
    if (k < e) sum += aa[k] * x[aj[k]];
    yout[row] = sum;
    if (DOT) p = x[row] * sum;
  }
  if (DOT) {
    const double t = hipx::wave_sum(p);
    if ((threadIdx.x & 63) == 0) dotpart[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = t;
  }
}

template <typename IT, int MODE, bool DOT>
int launch_spmv_t(hipxMat A, const double *x, const double *yin, double *yout, double *dotpart)
{
  setup_trace("product: enter");
  {
    bool ip = false, ips = false;
    int  ierr = use_inode_pair(A, ip, ips);
    setup_trace("inode decision");
    if (ierr) return ierr;
    if (ip && ips) return hipxSellLaunch_(A->sell_state, MODE, DOT ? 1 : 0, x, yin, yout, dotpart, 1);
    if (ip) {
      if (!A->m) return HIPX_SUCCESS;
      spmv_inode_kernel<IT, MODE, DOT><<<(unsigned)((A->m + 255) / 256), 256, 0, rt().compute>>>(A->m, (const IT *)A->d_i, A->d_j, A->d_a, x, yin, yout, dotpart);
      HIPX_LAUNCH_CHECK();
      return HIPX_SUCCESS;
    }
  }
  {
    bool sl = false;
    int  ierr = use_sell(A, sl);
    setup_trace("SELL decision");
    if (ierr) return ierr;
    if (sl) return hipxSellLaunch_(A->sell_state, MODE, DOT ? 1 : 0, x, yin, yout, dotpart, 0);
  }
  {
    bool tm = false;
    int  ierr = use_templates(A, tm);
    setup_trace("row templates (decision + build)");
    if (ierr) return ierr;
    if (tm) {
      ierr = launch_tmpl<MODE, DOT>(A, x, yin, yout, dotpart, nullptr);
      setup_trace("template product launched");
      return ierr;
    }
  }
  {
    bool tp = false;
    int  ierr = use_pattern_templates(A, tp);
    if (ierr) return ierr;
    if (tp) return launch_tp<IT, MODE, DOT>(A, x, yin, yout, dotpart);
  }
  {
    bool sl = false;
    int  ierr = use_sell(A, sl, true);
    if (ierr) return ierr;
    if (sl) return hipxSellLaunch_(A->sell_state, MODE, DOT ? 1 : 0, x, yin, yout, dotpart, 0);
  }
  if (A->tile_mode >= 2 && !A->compressed && (!A->probe || A->vd_mode)) return launch_pk16<IT, MODE, DOT>(A, x, yin, yout, dotpart);
  if (A->probe && MODE == 0 && !DOT && !A->compressed && !A->is64) {
    int ierr = ensure_row_blocks(A, 0);
    if (ierr) return ierr;
    const hipx_int nb = A->nblocks[0], per_xcd = (nb + 7) / 8;
    const unsigned grid = (unsigned)(per_xcd * 8);
    const hipx_int *sch = A->sched_mode ? A->d_sched[0] : nullptr;
#define HIPX_PROBE(D) \
  spmv_stream_kernel<hipx_int, 256, 2048, 1, false, 0, false, false, D><<<grid, 256, 0, rt().compute>>>(A->d_rb[0], sch, nb, per_xcd, (const hipx_int *)A->d_i, A->d_j, A->d_a, x, yin, yout, A->d_ridx, dotpart)
    if (A->probe == 1) HIPX_PROBE(1);
    else if (A->probe == 2) HIPX_PROBE(2);
    else HIPX_PROBE(3);
#undef HIPX_PROBE
    HIPX_LAUNCH_CHECK();
    return HIPX_SUCCESS;
  }
  return A->compressed ? launch_spmv_c<IT, MODE, true, DOT>(A, x, yin, yout, dotpart) : launch_spmv_c<IT, MODE, false, DOT>(A, x, yin, yout, dotpart);
}

// number of per-wave dot partials the next fused launch writes (every one of them is written by every launch of that
// kernel form, so the fold reads exactly these and no clearing pass is needed)
int dot_partials_count(hipxMat A, hipx_int *npart)
{
  int cfg, waves;
  {
    bool ip = false, ips = false;
    int  ierr = use_inode_pair(A, ip, ips);
    if (ierr) return ierr;
    if (ip) {
      *npart = ips ? hipxSellDotPartials_(A->sell_state) : (hipx_int)((A->m + 255) / 256) * 4;
      return HIPX_SUCCESS;
    }
  }
  {
    bool sl = false;
    int  ierr = use_sell(A, sl);
    if (ierr) return ierr;
    if (sl) {
      *npart = hipxSellDotPartials_(A->sell_state);
      return HIPX_SUCCESS;
    }
  }
  {
    bool tm = false;
    int  ierr = use_templates(A, tm);
    if (ierr) return ierr;
    if (tm) return launch_tmpl<0, true>(A, nullptr, nullptr, nullptr, nullptr, npart);
    bool tp = false;
    if ((ierr = use_pattern_templates(A, tp))) return ierr;
    if (tp) {
      *npart = (hipx_int)(((A->nblocks[0] + 7) / 8) * 8) * 4;
      return HIPX_SUCCESS;
    }
    bool sa = false;
    if ((ierr = use_sell(A, sa, true))) return ierr;
    if (sa) {
      *npart = hipxSellDotPartials_(A->sell_state);
      return HIPX_SUCCESS;
    }
  }
  if (A->tile_mode >= 2 && !A->compressed && (!A->probe || A->vd_mode)) {
    bool vd, rowpar;
    int  rpt;
    int  ierr = select_pk(A, vd, rowpar, rpt, cfg);
    if (ierr) return ierr;
    waves = 4;
  } else {
    bool nt;
    decode_variant(A->variant, cfg, nt);
    int ierr = ensure_row_blocks(A, cfg);
    if (ierr) return ierr;
    waves = kCfg[cfg].threads / 64;
  }
  *npart = (hipx_int)(((A->nblocks[cfg] + 7) / 8) * 8) * waves;
  return HIPX_SUCCESS;
}

template <int MODE, bool DOT>
int launch_spmv(hipxMat A, const double *x, const double *yin, double *yout, double *dotpart);

template <int MODE, bool DOT>
int launch_spmv(hipxMat A, const double *x, const double *yin, double *yout, double *dotpart)
{
  const bool timed = (MODE == 0) && !A->compressed;  // the MatMult kernel proper
  int        ierr;
  if (timed && (ierr = prof_mark(true))) return ierr;
  ierr = A->is64 ? launch_spmv_t<int64_t, MODE, DOT>(A, x, yin, yout, dotpart) : launch_spmv_t<hipx_int, MODE, DOT>(A, x, yin, yout, dotpart);
  if (ierr) return ierr;
  if (timed && (ierr = prof_mark(false))) return ierr;
  return HIPX_SUCCESS;
}

}  // namespace

// ---- value-only operations on the device copy (MatScale / MatZeroEntries / MatDiagonalScale without a host round trip)
namespace {
template <typename IT>
__global__ void mat_rowscale_kernel(hipx_int nrows, const IT *__restrict__ ai, const hipx_int *__restrict__ ridx, const double *__restrict__ l, double *a)
{
  for (hipx_int r = (hipx_int)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += (hipx_int)gridDim.x * blockDim.x) {
    const double x = l[ridx ? ridx[r] : r];  // aij.c:2350-2354: (*v++) *= l[i]
    for (IT k = ai[r]; k < ai[r + 1]; k++) a[k] *= x;
  }
}
__global__ void mat_colscale_kernel(int64_t nnz, const hipx_int *__restrict__ aj, const double *__restrict__ rvec, double *a)
{
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * blockDim.x) a[k] *= rvec[aj[k]];  // aij.c:2365
}
__global__ void mat_axpy_kernel(int64_t nnz, double alpha, const double *__restrict__ x, double *y)
{
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * blockDim.x) y[k] = y[k] + alpha * x[k];  // daxpy on the value arrays (aij.c:2940)
}
__global__ void mat_scale_kernel(int64_t nnz, double alpha, double *a)
{
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * blockDim.x) a[k] *= alpha;  // dscal (aij.c:2613)
}
void values_changed(hipxMat A)
{
  A->vd_ready   = false;
  A->tmpl_ready = false;
  A->value_state++;
  hipxSorInvalidate_(A->sor_state);
  hipxSellValuesChanged_(A->sell_state);
}
}  // namespace


extern "C" {

int hipxProfileSpMV(int enable)
{
  HIPX_CHECK_INIT();
  prof().on   = enable != 0;
  prof().used = 0;
  return HIPX_SUCCESS;
}

int hipxProfileSpMVGet(int *count, double *total_ms)
{
  HIPX_CHECK_INIT();
  SpmvProf &p = prof();
  HIPX_HIP(hipStreamSynchronize(rt().compute));
  double tot = 0.0;
  for (size_t k = 0; k + 1 < p.used; k += 2) {
    float ms = 0.f;
    HIPX_HIP(hipEventElapsedTime(&ms, p.ev[k], p.ev[k + 1]));
    tot += ms;
  }
  if (count) *count = (int)(p.used / 2);
  if (total_ms) *total_ms = tot;
  p.used = 0;
  return HIPX_SUCCESS;
}

int hipxMatCreateCSR(hipx_int m, hipx_int n, const hipx_int *i, const hipx_int *j, const double *a, hipxMat *A)
{
  HIPX_CHECK_INIT();
  return create_common<hipx_int>(m, n, m, i, nullptr, j, a, false, A);
}

int hipxMatCreateCSR64(hipx_int m, hipx_int n, const int64_t *i, const hipx_int *j, const double *a, hipxMat *A)
{
  HIPX_CHECK_INIT();
  return create_common<int64_t>(m, n, m, i, nullptr, j, a, true, A);
}

int hipxMatCreateCSRCompressedRow(hipx_int m, hipx_int n, hipx_int nrows, const hipx_int *ci, const hipx_int *ridx, const hipx_int *j, const double *a, hipxMat *A)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(ridx || nrows == 0, "compressed-row matrix needs the row index list (Mat_CompressedRow.rindex)");
  static const hipx_int dummy = 0;  // nrows == 0: marks the matrix as compressed, never dereferenced
  return create_common<hipx_int>(m, n, nrows, ci, ridx ? ridx : &dummy, j, a, false, A);
}

int hipxMatUpdateValues(hipxMat A, const double *a)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(A, "null matrix");
  if (A->nnz) HIPX_HIP(hipMemcpyAsync(A->d_a, a, sizeof(double) * (size_t)A->nnz, hipMemcpyHostToDevice, rt().compute));
  HIPX_HIP(hipStreamSynchronize(rt().compute));
  A->vd_ready = false;  // the value dictionary is rebuilt at the next product
  A->tmpl_ready = false;  // ... and so are the row templates
  A->value_state++;  // SOR's level-ordered copy and inverse diagonal must be rebuilt (aij.c:1807 idiagState)
  hipxSorInvalidate_(A->sor_state);
  hipxSellValuesChanged_(A->sell_state);
  return HIPX_SUCCESS;
}

int hipxMatScale(hipxMat A, double alpha)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(A, "null matrix");
  if (A->nnz) {
    const unsigned g = (unsigned)std::min<int64_t>((A->nnz + 255) / 256, 8192);
    mat_scale_kernel<<<g, 256, 0, rt().compute>>>(A->nnz, alpha, A->d_a);
    HIPX_LAUNCH_CHECK();
  }
  values_changed(A);
  return HIPX_SUCCESS;
}

int hipxMatAXPY(hipxMat Y, double alpha, hipxMat X)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(Y && X, "null matrix");
  HIPX_ARG(Y->m == X->m && Y->n == X->n && Y->nnz == X->nnz && Y->compressed == X->compressed, "MatAXPY (SAME_NONZERO_PATTERN): the two matrices must share one nonzero pattern");
  if (Y->nnz && alpha != 0.0) {
    const unsigned g = (unsigned)std::min<int64_t>((Y->nnz + 255) / 256, 8192);
    mat_axpy_kernel<<<g, 256, 0, rt().compute>>>(Y->nnz, alpha, X->d_a, Y->d_a);
    HIPX_LAUNCH_CHECK();
  }
  values_changed(Y);
  return HIPX_SUCCESS;
}

int hipxMatZeroEntries(hipxMat A)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(A, "null matrix");
  if (A->nnz) HIPX_HIP(hipMemsetAsync(A->d_a, 0, sizeof(double) * (size_t)A->nnz, rt().compute));
  values_changed(A);
  return HIPX_SUCCESS;
}

int hipxMatDiagonalScale(hipxMat A, const double *l, const double *r)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(A, "null matrix");
  if (A->nnz && l) {  // left first, then right: (a * l_i) * r_j, the reference's two passes (aij.c:2343-2369)
    const unsigned g = (unsigned)std::min<hipx_int>((A->nrows_c + 255) / 256, 8192);
    if (A->is64) mat_rowscale_kernel<int64_t><<<g ? g : 1, 256, 0, rt().compute>>>(A->nrows_c, (const int64_t *)A->d_i, A->compressed ? A->d_ridx : nullptr, l, A->d_a);
    else mat_rowscale_kernel<hipx_int><<<g ? g : 1, 256, 0, rt().compute>>>(A->nrows_c, (const hipx_int *)A->d_i, A->compressed ? A->d_ridx : nullptr, l, A->d_a);
    HIPX_LAUNCH_CHECK();
  }
  if (A->nnz && r) {
    const unsigned g = (unsigned)std::min<int64_t>((A->nnz + 255) / 256, 8192);
    mat_colscale_kernel<<<g, 256, 0, rt().compute>>>(A->nnz, A->d_j, r, A->d_a);
    HIPX_LAUNCH_CHECK();
  }
  values_changed(A);
  return HIPX_SUCCESS;
}

int hipxMatDestroy(hipxMat *pA)
{
  HIPX_CHECK_INIT();
  if (!pA || !*pA) return HIPX_SUCCESS;
  hipxMat A = *pA;
  mpicg_forget(A);
  HIPX_HIP(hipStreamSynchronize(rt().compute));
  (void)hipFree(A->d_i);
  (void)hipFree(A->d_j);
  (void)hipFree(A->d_a);
  (void)hipFree(A->d_diagpos);
  for (int c = 0; c < hipxMat_s::kMaxCfg; c++) {
    (void)hipFree(A->d_rb[c]);
    (void)hipFree(A->d_sched[c]);
  }
  (void)hipFree(A->d_ridx);
  (void)hipFree(A->d_dotpart);
  (void)hipFree(A->d_pk);
  (void)hipFree(A->d_pkbase);
  (void)hipFree(A->d_pkdesc);
  (void)hipFree(A->d_vc);
  (void)hipFree(A->d_vdict);
  free_templates(A);
  free_pattern_templates(A);
  hipxSorStateFree_(A->sor_state);
  hipxSellFree_(A->sell_state);
  delete A;
  *pA = nullptr;
  return HIPX_SUCCESS;
}

// internal accessor for hipx_sor.hip (not part of the public ABI)
int hipxMatInternal_(hipxMat A, hipx_int *m, hipx_int *n, int64_t *nnz, int *is64, void **d_i, hipx_int **d_j, double **d_a, int64_t **d_diagpos, int *diag_dense,
                     int *compressed, void ***sor_slot, unsigned long long *value_state)
{
  HIPX_ARG(A, "null matrix");
  *m = A->m;
  *n = A->n;
  *nnz = A->nnz;
  *is64 = A->is64;
  *d_i = A->d_i;
  *d_j = A->d_j;
  *d_a = A->d_a;
  *d_diagpos = A->d_diagpos;
  *diag_dense = A->diag_dense;
  *compressed = A->compressed;
  *sor_slot = &A->sor_state;
  *value_state = A->value_state;
  return HIPX_SUCCESS;
}

// inodes (include/hipx.h).  The set-up lives with the relaxation (hipx_sor.hip); the matrix only remembers the partition
int hipxMatSetInodes(hipxMat A, hipx_int node_count, const hipx_int *size_csr)
{
  HIPX_ARG(A, "null matrix");
  HIPX_ARG(node_count >= 0 && (node_count == 0 || size_csr), "node_count > 0 needs the row offsets of the nodes");
  if (node_count > 0) {
    HIPX_ARG(size_csr[0] == 0 && size_csr[node_count] == A->m, "the nodes must cover rows 0 ... m - 1");
    for (hipx_int i = 0; i < node_count; i++) HIPX_ARG(size_csr[i + 1] > size_csr[i] && size_csr[i + 1] - size_csr[i] <= 5, "a node has 1 to 5 rows (inode.c:2484)");
    A->inode_sizes.assign(size_csr, size_csr + node_count + 1);
    A->inode_state = 1;
  } else {
    A->inode_sizes.clear();
    A->inode_state = 0;
  }
  hipxSorStateFree_(A->sor_state);  // schedules are per partition
  A->sor_state = nullptr;
  return HIPX_SUCCESS;
}

int hipxMatGetInodes(hipxMat A, hipx_int *node_count)
{
  HIPX_ARG(A && node_count, "null argument");
  *node_count = A->inode_state < 0 ? -1 : (A->inode_state ? (hipx_int)A->inode_sizes.size() - 1 : 0);
  return HIPX_SUCCESS;
}

// internal, hipx_sor.hip: the partition (state as in the struct; sizes = node_count + 1 host offsets) / the result of its own look
int hipxMatInodes_(hipxMat A, int *state, hipx_int *node_count, const hipx_int **sizes)
{
  *state      = A->inode_state;
  *node_count = A->inode_state == 1 ? (hipx_int)A->inode_sizes.size() - 1 : 0;
  *sizes      = A->inode_state == 1 ? A->inode_sizes.data() : nullptr;
  return HIPX_SUCCESS;
}
int hipxMatInodesFound_(hipxMat A, hipx_int node_count, const hipx_int *sizes)
{
  if (node_count > 0) {
    A->inode_sizes.assign(sizes, sizes + node_count + 1);
    A->inode_state = 1;
  } else A->inode_state = 0;
  return HIPX_SUCCESS;
}

// internal accessor for hipx_sor.hip: the row templates (host tables + device ids), built on demand; *ok = 0 when the matrix
// has none
int hipxMatTemplates_(hipxMat A, int *ok, int *ntmpl, const int **tstart, const int **toff, const double **tval, const int **tdiag, const int64_t **tcount,
                      const unsigned char **d_tid)
{
  HIPX_ARG(A, "null matrix");
  *ok = 0;
  if (A->compressed || A->nrows_c <= 0) return HIPX_SUCCESS;
  int ierr;
  if ((ierr = ensure_vdict(A))) return ierr;
  if (!A->vd_ok) return HIPX_SUCCESS;
  if ((ierr = ensure_templates(A))) return ierr;
  if (!A->tmpl_ok) return HIPX_SUCCESS;
  *ok     = 1;
  *ntmpl  = A->ntmpl;
  *tstart = A->h_tstart.data();
  *toff   = A->h_toff.data();
  *tval   = A->h_tval.data();
  *tdiag  = A->h_tdiag.data();
  *tcount = A->h_tcount.data();
  *d_tid  = A->d_tid;
  return HIPX_SUCCESS;
}

// internal accessor for hipx_sor.hip: the PATTERN templates (the rows' (column - row) lists without the values: matrices with
// arbitrary values on a stencil pattern), built on demand; host tables + the device copies the coefficient-stream kernel reads
int hipxMatPatternTemplates_(hipxMat A, int *ok, int *ntmpl, const int **tstart, const int **toff, const int **tdiag, const int64_t **tcount, const unsigned char **d_tid,
                             const int **d_tstart, const int **d_toff)
{
  HIPX_ARG(A, "null matrix");
  *ok = 0;
  if (A->compressed || A->nrows_c <= 0) return HIPX_SUCCESS;
  int ierr;
  if ((ierr = ensure_pattern_templates(A))) return ierr;
  if (!A->ptm_ok || A->h_ptstart.empty()) return HIPX_SUCCESS;
  *ok       = 1;
  *ntmpl    = A->ptm_ntmpl;
  *tstart   = A->h_ptstart.data();
  *toff     = A->h_ptoff.data();
  *tdiag    = A->h_ptdiag.data();
  *tcount   = A->h_ptcount.data();
  *d_tid    = A->d_ptid;
  *d_tstart = A->d_ptstart;
  *d_toff   = A->d_ptoff;
  return HIPX_SUCCESS;
}

int hipxMatGetInfo(hipxMat A, hipx_int *m, hipx_int *n, int64_t *nnz, int64_t *device_bytes)
{
  HIPX_ARG(A, "null matrix");
  if (m) *m = A->m;
  if (n) *n = A->n;
  if (nnz) *nnz = A->nnz;
  if (device_bytes) *device_bytes = A->device_bytes;
  return HIPX_SUCCESS;
}

int hipxMatSetSpMVVariant(hipxMat A, int variant)
{
  HIPX_ARG(A && variant >= 0, "variant: 0 auto, else 1 + 2*geometry + (1 if non-temporal loads); add 100 for the band-aware block schedule");
  A->probe      = variant / 1000;  // 1000/2000/3000 + v: probe kernels
  variant %= 1000;
  // 22: packed columns, 23: packed columns + row-parallel gather, 24 / 25: the same two with the value dictionary
  // 26: row templates (falls back to 25 when the matrix has more than 256 distinct rows)
  // 30: 26 with the march form of the template kernel whenever the base template has its three-plane shape (auto and 26 want >= 192 workgroups)
  A->march_force = (variant == 30);
  if (variant == 30) variant = 26;
  A->tile_mode = (variant == 22 || variant == 24) ? 2 : (variant == 23 || variant == 25 || variant == 26) ? 3 : 0;
  A->vd_mode   = (variant == 24 || variant == 25 || variant == 26) ? 1 : 0;
  A->tmpl_mode = (variant == 26) ? 1 : 0;
  A->sell_mode = (variant == 28) ? 1 : 0;  // 28: SELL-64 copy (hipx_sell.hip); falls back to 23 when the format does not apply
  A->ptm_mode = (variant == 29) ? 1 : 0;   // 29: pattern templates + streamed values (spmv_tp_kernel); falls back to 23
  if (variant == 28 || variant == 29) {
    A->tile_mode = 3;
    variant      = 23;
  }
  A->auto_sel = (variant == 0);
  if (variant == 0) {
    A->tile_mode = auto_tile_mode(A);
    A->vd_mode   = A->tile_mode ? 1 : 0;
    A->tmpl_mode = A->tile_mode ? 1 : 0;
    A->ptm_mode  = A->tile_mode ? 1 : 0;  // (used when neither the row templates nor the value dictionary apply: arbitrary values on a stencil pattern)
    A->sell_mode = A->tile_mode ? 2 : 0;  // (auto stage: long rows with arbitrary values)
  }
  if (variant >= 22 && variant <= 26) variant = 1;
  A->sched_mode = variant >= 100 ? 1 : 0;
  variant %= 100;
  HIPX_ARG(variant <= 2 * kNumCfg, "unknown SpMV variant");
  if (A->d_dotpart && variant != A->variant) {
    HIPX_HIP(hipStreamSynchronize(rt().compute));
    (void)hipFree(A->d_dotpart);
    A->d_dotpart   = nullptr;
    A->dotpart_cap = 0;
  }
  A->variant = variant;
  return HIPX_SUCCESS;
}

// (what launch_tmpl decides for 16-byte aligned vectors)
static bool march_applies(hipxMat A)
{
  if (!A->march_ok || A->tmpl_base < 0 || !A->d_tmask || A->nrows_c % 2 || dev_sw().nomarch || dev_sw().nosub || dev_sw().probe || tmpl_cfg() != 1 || A->ntmpl > 256) return false;
  const hipx_int m = A->nrows_c;
  if (A->march_nt == 512) {  // long lines: the 512-thread form of spmv_march2_kernel or nothing
    if (dev_sw().march1 || dev_sw().trace || march2_check(A) || A->march2_state != 1) return false;
  }
  int tiles, nseg, pps, nplanes, units;
  march_geometry(A, tiles, nseg, pps, nplanes, units);
  return (units >= 192 || A->march_force) && (hipx_int)units <= (m + 511) / 512;
}

int hipxMatGetSpMVKernel(hipxMat A, char *buf, size_t len)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(A && buf && len > 0, "null argument");
  const char *name = "spmv_stream_kernel (CSR MatMult, 32-bit columns)";
  {
    bool ip = false, ips = false;
    int  ierr = use_inode_pair(A, ip, ips);
    if (ierr) return ierr;
    if (ip) {
      snprintf(buf, len, "%s", ips ? "spmv_sell_kernel (MatMult on the SELL-64 copy: one lane per row, 16-bit window-coded columns; matrix with inodes: pairwise row sums of MatMult_SeqAIJ_Inode)"
                                   : "spmv_inode_kernel (CSR MatMult, one lane per row; matrix with inodes: pairwise row sums of MatMult_SeqAIJ_Inode)");
      return HIPX_SUCCESS;
    }
  }
  bool        tm   = false;
  {
    int ierr = use_templates(A, tm);
    if (ierr) return ierr;
  }
  bool sl = false;
  {
    int ierr = use_sell(A, sl);
    if (ierr) return ierr;
  }
  bool tp = false;
  if (!sl && !tm) {
    int ierr = use_pattern_templates(A, tp);
    if (ierr) return ierr;
  }
  if (!sl && !tm && !tp) {
    int ierr = use_sell(A, sl, true);
    if (ierr) return ierr;
  }
  if (sl) name = "spmv_sell_kernel (MatMult on the SELL-64 copy: one lane per row, 16-bit window-coded columns)";
  else if (tm && march_applies(A)) {
    int ierr = march2_check(A);
    if (ierr) return ierr;
    if (A->march2_state == 1 && !dev_sw().march1 && !dev_sw().trace)
      name = "spmv_march2_kernel (CSR MatMult, row templates: 1 byte per row, plane-periodic; three planes of x resident in LDS, every operand an LDS read at a run's immediate offset)";
    else name = "spmv_march_kernel (CSR MatMult, row templates: 1 byte per row; three planes of x resident in LDS, every operand an LDS read)";
  }
  else if (tm && A->pair_ok && A->pair_plan.npairs <= dev_sw().pairmax && A->d_tmask && !dev_sw().nosub && !dev_sw().nopair && !dev_sw().probe && tmpl_cfg() == 1 && A->nrows_c >= 512)
    name = "spmv_pair_kernel (CSR MatMult, row templates: 1 byte per row; two consecutive rows per thread, 16-byte loads of x at the even offsets, +-1 entries from the neighbouring lanes)";
  else if (tm && A->d_tmask && !dev_sw().nosub) name = "spmv_tmpl_kernel (CSR MatMult, row templates: 1 byte per row; every template a subset of the interior one: uniform masked walk)";
  else if (tm) name = "spmv_tmpl_kernel (CSR MatMult, row templates: 1 byte per row)";
  else if (tp) name = "spmv_tp_kernel (CSR MatMult, pattern templates: 1-byte pattern id per row, values streamed from a[])";
  else if (false) name = "spmv_tmpl_kernel (CSR MatMult, row templates: 1 byte per row)";
  else if (A->tile_mode >= 2 && !A->compressed && !A->probe) {
    bool vd, rowpar;
    int  rpt, cfg;
    int  ierr = select_pk(A, vd, rowpar, rpt, cfg);
    if (ierr) return ierr;
    if (rpt > 0) name = "spmv_vd_kernel (CSR MatMult, packed 16-bit columns, 8-bit value dictionary, row-parallel gather)";
    else if (rowpar) name = "spmv_pk16r_kernel (CSR MatMult, packed 16-bit columns, row-parallel gather)";
    else name = vd ? "spmv_pk16_kernel<VD> (CSR MatMult, packed 16-bit columns, 8-bit value dictionary)" : "spmv_pk16_kernel (CSR MatMult, packed 16-bit columns)";
  }
  snprintf(buf, len, "%s", name);
  return HIPX_SUCCESS;
}

// Round 6 (verdict r5 item 7): everything the FIRST product would build lazily -- the inode search, row / pattern templates and their verification, the march
// plan and its plane-periodicity check, the SELL-64 copy or the packed-column format -- built NOW, so that a caller's set-up phase (MatAssemblyEnd, PCSetUp)
// pays for it and not its first timed MatMult.  The selection chain is the one hipxMatGetSpMVKernel reports from.
int hipxMatSetUp(hipxMat A)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(A, "null argument");
  if (A->compressed || A->m <= 0 || A->nnz <= 0) return HIPX_SUCCESS;  // (off-diagonal blocks take the row-block stream kernel: its row blocks are cut at creation)
  char buf[8];
  return hipxMatGetSpMVKernel(A, buf, sizeof(buf));
}

// replaces one iteration of KSPSolve_Chebyshev_FirstKind (cheby.c:475-511 with PCJACOBI / PCNONE, no norm): pnext = alpha pprev + beta pcur +
// gamma (dinv .* (b - A pcur)) -- ONE kernel when the matrix takes the pair form (the Chebyshev step as the SpMV's epilogue: the current
// iterate is the walk's diagonal pair, so the kernel adds three streams to the SpMV's and stores pnext instead of A pcur), else the
// SpMV followed by hipxVecChebyshevStep in place.  Element by element the operations of the reference's loops: bit-identical.
int hipxMatMultChebyshev(hipxMat A, const double *pcur, double *pnext, double alpha, double beta, double gamma, const double *pprev, const double *dinv, const double *b)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(A && !A->compressed && A->m == A->n && (A->m == 0 || (pcur && pnext && pprev && b)), "null argument / not a square uncompressed matrix");
  HIPX_ARG(pcur != pnext && pprev != pnext, "the three iterates must be different vectors");
  if (!A->m) return HIPX_SUCCESS;
  g_epi      = hipxPairEpi{b, dinv, pprev, alpha, beta, gamma, (alpha == 1.0) ? 0 : ((gamma == 1.0) ? 1 : ((gamma == 0.0) ? 2 : 3))};
  g_epi_on   = true;
  g_epi_done = false;
  int ierr   = launch_spmv<0, false>(A, pcur, nullptr, pnext, nullptr);
  g_epi_on   = false;
  if (!ierr && !g_epi_done) ierr = hipxVecChebyshevStep(pnext, alpha, beta, gamma, pprev, pcur, dinv, b, pnext, nullptr, A->m);  // (A pcur sits in pnext: in place)
  return ierr;
}

int hipxMatMult(hipxMat A, const double *x, double *y)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(A && (x || !A->n) && (y || !A->m), "null argument");
  HIPX_ARG(x != y, "MatMult: x and y must differ (matrix.c:2706)");
  if (A->compressed) {  // aij.c:1468: y is zeroed, only the listed rows are written
    if (A->m) HIPX_HIP(hipMemsetAsync(y, 0, sizeof(double) * (size_t)A->m, rt().compute));
  }
  return launch_spmv<0, false>(A, x, nullptr, y, nullptr);
}

int hipxMatMultAdd(hipxMat A, const double *x, const double *y, double *z)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(A && (x || !A->n), "null argument");
  HIPX_ARG(x != z, "MatMultAdd: x and z must differ (matrix.c:2860)");
  if (A->compressed && y != z && A->m) HIPX_HIP(hipMemcpyAsync(z, y, sizeof(double) * (size_t)A->m, hipMemcpyDeviceToDevice, rt().compute));  // aij.c:1629
  return launch_spmv<1, false>(A, x, A->compressed ? z : y, z, nullptr);
}

static void m2_fold_arm(int slot, double *dev_dot)
{
  static const bool nofold = getenv("HIPX_MARCH_NOFOLD") != nullptr;
  g_m2_fold_slot = nofold ? -1 : slot;
  g_m2_fold_dres = dev_dot;
  g_m2_folded    = false;
}

static int matmultdot_launch(hipxMat A, const double *x, double *y, hipx_int *npart_out)
{
  hipx_int npart = 0;
  int      ierr0 = dot_partials_count(A, &npart);
  if (ierr0) return ierr0;
  *npart_out = npart;
  if (!npart) return HIPX_SUCCESS;
  const hipx_int room = npart + march2_rem_parts(A);  // (+ the partials of the rows beyond the whole tiles, should the march form take this product)
  if (room > A->dotpart_cap) {
    HIPX_HIP(hipStreamSynchronize(rt().compute));
    (void)hipFree(A->d_dotpart);
    HIPX_HIP(hipMalloc((void **)&A->d_dotpart, sizeof(double) * (size_t)room));
    A->dotpart_cap = room;
  }
  A->dot_npart_used = 0;
  int ierr = launch_spmv<0, true>(A, x, nullptr, y, A->d_dotpart);
  *npart_out = A->dot_npart_used ? A->dot_npart_used : npart;  // (the march form of the template kernel writes one partial per workgroup and wave)
  return ierr;
}

int hipxMatMultDot(hipxMat A, const double *x, double *y, double *dot)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(A && !A->compressed && A->m == A->n, "MatMultDot needs a square, uncompressed matrix");
  *dot = 0.0;
  if (rt().red_exact) {  // compensated mode: the product, then Dot2 over the complete vectors (the epilogue's per-wave partials are plain sums)
    int ierr = hipxMatMult(A, x, y);
    return ierr ? ierr : hipxVecDot(x, y, A->m, dot);
  }
  hipx_int npart = 0;
  int      ierr  = matmultdot_launch(A, x, y, &npart);
  if (ierr || !npart) return ierr;
  return hipxVecSum(A->d_dotpart, npart, dot);
}

int hipxMatMultDotBegin(hipxMat A, const double *x, double *y, int slot, double *dev_dot)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(A && !A->compressed && A->m == A->n && A->m > 0, "MatMultDotBegin needs a square, non-empty, uncompressed matrix");
  HIPX_ARG(slot >= 0 && slot < HIPX_MAX_RED_SLOTS - 2, "reduction slot out of range");
  if (rt().red_exact) {
    int ierr = hipxMatMult(A, x, y);
    return ierr ? ierr : launch_dot(x, y, A->m, slot, dev_dot);
  }
  hipx_int npart = 0;
  m2_fold_arm(slot, dev_dot);
  int ierr = matmultdot_launch(A, x, y, &npart);
  g_m2_fold_slot = -1;
  if (ierr || g_m2_folded) return ierr;
  return launch_sum(A->d_dotpart, npart, slot, dev_dot);
}

// the common part of hipxMatMultCGDirectionDotBegin (one rank: slot >= 0, the dot folded into the slot) and hipxMatMultCGDirectionPartial_ (the diagonal
// block of an MPIAIJ operator: slot < 0, the partials of the rows WITHOUT off-diagonal entries are left in A->d_dotpart for offdiag_dot_kernel)
static int cgdir_product(hipxMat A, const double *p_old, double *p_new, const double *z, double dconst, double *x, double b, double a, const double *dev_beta_new, const double *dev_beta_old,
                         const double *dev_dpi, double *w, int slot, double *dev_dot, int skipmask, bool want_partials, int *fused, hipx_int *npart_out)
{
  *fused = 0;
  static const bool off = getenv("HIPX_NO_CGFUSE") != nullptr;
  if (off || A->compressed || A->m != A->n || A->m <= 0) return HIPX_SUCCESS;
  if ((reinterpret_cast<uintptr_t>(p_old) | reinterpret_cast<uintptr_t>(p_new) | reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15) return HIPX_SUCCESS;
  bool tm = false, ip = false, ips = false;
  int  ierr = use_inode_pair(A, ip, ips);  // a matrix with inodes is multiplied in MatMult_SeqAIJ_Inode's order: never by the template kernels
  if (ierr) return ierr;
  if (ip) return HIPX_SUCCESS;
  if ((ierr = use_templates(A, tm))) return ierr;
  if (!tm || !march_applies(A) || dev_sw().march1 || dev_sw().trace) return HIPX_SUCCESS;
  if ((ierr = march2_check(A))) return ierr;
  if (A->march2_state != 1 || !march2_cg_shape(A)) return HIPX_SUCCESS;
  int tiles, nseg, pps, nplanes, units;
  march_geometry(A, tiles, nseg, pps, nplanes, units);
  hipx_int npart = (hipx_int)units * (A->march_nt / 64) + march2_rem_parts(A);  // (an upper bound when a plane has rows beyond its whole tiles; the launch says what it wrote)
  if (npart > A->dotpart_cap) {
    HIPX_HIP(hipStreamSynchronize(rt().compute));
    (void)hipFree(A->d_dotpart);
    HIPX_HIP(hipMalloc((void **)&A->d_dotpart, sizeof(double) * (size_t)npart));
    A->dotpart_cap = npart;
  }
  const hipxMarchPlan &mp  = A->march_plan;
  const size_t         lds = 3 * (size_t)(mp.L + 2 * mp.H) * sizeof(double);
  const int            xm  = (units % 8 == 0) ? 1 : 0;
  const hipxMarchCG    cg{z, p_new, x, dconst, b, a, dev_beta_new, dev_beta_old, dev_dpi};
  if ((ierr = prof_mark(true))) return ierr;
  if (!want_partials && (rt().red_exact || slot < 0)) {  // compensated mode: the epilogue's per-wave partials are plain sums -- Dot2 over the complete vectors instead
    if ((ierr = launch_march2_cg<false>(A, p_old, w, nullptr, units, tiles, pps, nplanes, xm, lds, cg))) return ierr;
    if ((ierr = prof_mark(false))) return ierr;
    if (slot >= 0 && (ierr = launch_dot(p_new, w, A->m, slot, dev_dot))) return ierr;
    npart = 0;
  } else {
    if (slot >= 0) m2_fold_arm(slot, dev_dot);
    else {
      g_m2_fold_slot = -1;
      g_m2_folded    = false;
    }
    g_m2_skip         = skipmask;
    A->dot_npart_used = 0;
    ierr              = launch_march2_cg<true>(A, p_old, w, A->d_dotpart, units, tiles, pps, nplanes, xm, lds, cg);
    g_m2_fold_slot    = -1;
    g_m2_skip         = 0;
    if (ierr) return ierr;
    if (A->dot_npart_used) npart = A->dot_npart_used;
    if ((ierr = prof_mark(false))) return ierr;
    if (slot >= 0 && !g_m2_folded && (ierr = launch_sum(A->d_dotpart, npart, slot, dev_dot))) return ierr;
  }
  if (npart_out) *npart_out = npart;
  *fused = 1;
  return HIPX_SUCCESS;
}

int hipxMatMultCGDirectionDotBegin(hipxMat A, const double *p_old, double *p_new, const double *z, double dconst, double *x, double b, double a, const double *dev_beta_new,
                                   const double *dev_beta_old, const double *dev_dpi, double *w, int slot, double *dev_dot, int *fused)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(A && fused && p_old && p_new && z && x && w, "null argument");
  HIPX_ARG(slot >= 0 && slot < HIPX_MAX_RED_SLOTS - 2, "reduction slot out of range");
  HIPX_ARG(p_old != p_new && p_new != w && p_old != w && x != w && x != p_new && z != w && z != p_new, "the vectors must be distinct (z may not alias w: W = Z in cg.c:145 is for the caller to unalias)");
  return cgdir_product(A, p_old, p_new, z, dconst, x, b, a, dev_beta_new, dev_beta_old, dev_dpi, w, slot, dev_dot, 0, false, fused, nullptr);
}

// ---- the same on the diagonal block of an MPIAIJ operator (round 6; internal: hipx_comm.hip's hipxMatMultMPICGDirectionDotBegin drives it) --------------
// hipxMatMPICGPlan_: can the pair (Ad, Bo) run the two-kernel form?  Ad must take the CG-prologue march kernel; Bo (compressed rows, 32-bit offsets) must have
// its rows in EXACTLY the first and / or the last plane of Ad's grid (every row of such a plane): a z-slab of a natural-ordered stencil grid.  *skipmask: bit 0
// = the first plane's rows carry off-diagonal entries, bit 1 = the last plane's.  Decided once per pair (a host copy of Bo's row list, nrows_c integers).
// B->ops->multadd of MatMult_MPIAIJ (mpiaij.c:1059) on the listed rows, w_i = ((w_i + b_i0 g_0) + b_i1 g_1) ... (MatMultAdd_SeqAIJ's compressed-row loop,
// aij.c:1629-1641: products and sums rounded separately, left to right), fused with the rows' share of the dot p . w (cg.c:258) and with the FOLD of all the dot
// partials -- the product kernel's (rows without off-diagonal entries) and this kernel's: a partition of the rows, every product p_i w_i formed once with
// the complete w_i.  A workgroup takes 256 consecutive listed rows; its nonzeros are one contiguous range, read coalesced, the products parked in LDS (the
// row-block idea of spmv_stream_kernel); a range beyond the tile falls back to the per-row walk.  The last workgroup (ticket; release / acquire at agent scope:
// the R1 hand-off of hipx_reduce.h) adds partA[0 .. npartA) and the workgroups' partials in index order per thread, wave tree, 4 wave sums left to right.
}  // extern "C"
namespace {
constexpr int OD_CAP = 2560, OD_WIN = 2048, OD_GRP = 32;
// the ghost columns the entries of every 256-row workgroup touch: (cmin rounded down to a pair, cmax) -- once per off-diagonal block (hipxMatMultAddDotFold_)
__global__ __launch_bounds__(256) void offdiag_window_kernel(hipx_int nrows, const hipx_int *__restrict__ bi, const hipx_int *__restrict__ bj, int2 *__restrict__ win)
{
  __shared__ hipx_int s_c[2][4];
  const int      t  = threadIdx.x;
  const hipx_int r0 = (hipx_int)blockIdx.x * 256, r1 = (r0 + 256 < nrows) ? r0 + 256 : nrows;
  hipx_int       cmin = 0x7fffffff, cmax = -1;
  for (hipx_int k = bi[r0] + t; k < bi[r1]; k += 256) {
    const hipx_int c = bj[k];
    cmin = c < cmin ? c : cmin;
    cmax = c > cmax ? c : cmax;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const hipx_int a = __shfl_down(cmin, off, 64), b2 = __shfl_down(cmax, off, 64);
    cmin = a < cmin ? a : cmin;
    cmax = b2 > cmax ? b2 : cmax;
  }
  if ((t & 63) == 0) {
    s_c[0][t >> 6] = cmin;
    s_c[1][t >> 6] = cmax;
  }
  __syncthreads();
  if (t == 0) {
    for (int q = 1; q < 4; q++) {
      cmin = s_c[0][q] < cmin ? s_c[0][q] : cmin;
      cmax = s_c[1][q] > cmax ? s_c[1][q] : cmax;
    }
    win[blockIdx.x] = make_int2(cmin & ~1, cmax);
  }
}

__global__ __launch_bounds__(256) void offdiag_dot_kernel(hipx_int nrows, const hipx_int *__restrict__ bi, const hipx_int *__restrict__ bj, const double *__restrict__ ba,
                                                          const hipx_int *__restrict__ ridx, const double *ghost, double *__restrict__ w, const double *__restrict__ p,
                                                          double *part, const double *partA, int npartA, unsigned int *ticket, unsigned int *gticket, double *dst, const IpcWait wt,
                                                          const hipx_int ncols, const int2 *__restrict__ win)
{
  __shared__ double   s_prod[OD_CAP];
  __shared__ double   s_g[OD_WIN];
  __shared__ double   s_w[4];
  __shared__ unsigned s_last;
  const int      t  = threadIdx.x;
  const hipx_int r0 = (hipx_int)blockIdx.x * 256, r1 = (r0 + 256 < nrows) ? r0 + 256 : nrows;
  const hipx_int k0 = bi[r0], k1 = bi[r1];
  const bool     staged = (k1 - k0) <= OD_CAP;
  // IPC transport: the neighbours' put kernels raise this rank's sequence flags when their write-through stores have drained; one lane per flag polls, and the
  // ghost values are then read with system-scope (sc0 sc1) loads -- the "sc1 stores and sc1 loads on both sides" hand-off of hipx_ipc.h, no wait kernel in front
  const bool     sysload = wt.n > 0;
  // the ghost columns this workgroup's rows touch (a stencil's boundary rows: a few consecutive runs): when the window fits it is loaded ONCE, coalesced,
  // 16 bytes per lane, into LDS and every operand is an LDS read (per-entry system-scope loads go to memory one by one)
  const int2     wn     = win[blockIdx.x];
  const hipx_int cmin   = wn.x, cmax = wn.y;
  // (round 6) what the row's last step needs -- its place in w, the diagonal block's sum, p, its entry range -- is requested NOW: read where it is used, each of
  // these (two of them dependent on ridx) was a memory round trip behind the ghost wait, the window and the products, in every workgroup's life
  const bool     rowok = r0 + t < r1;
  const hipx_int myrow = rowok ? ridx[r0 + t] : 0;
  const hipx_int mk0 = rowok ? bi[r0 + t] : k0, mk1 = rowok ? bi[r0 + t + 1] : k0;
  const double   w_in = rowok ? w[myrow] : 0.0, p_in = rowok ? p[myrow] : 0.0;
  // ... and so are the workgroup's entries (values and ghost columns: <= OD_CAP / 256 per thread, all in flight at once, before the ghost wait): as a loop with a
  // run-time trip count each of its ~9 rounds was a memory round trip of its own
  constexpr int NI = OD_CAP / 256;
  double        av[NI];
  hipx_int      cv[NI];
  if (staged) {
#pragma unroll
    for (int u = 0; u < NI; u++) {
      const hipx_int k  = k0 + t + 256 * u;
      const hipx_int kk = (k < k1) ? k : k0;  // (beyond the range: a valid entry re-read, never used)
      av[u]             = (k1 > k0) ? ba[kk] : 0.0;
      cv[u]             = (k1 > k0) ? bj[kk] : 0;
    }
  }
  const bool     window = staged && cmax >= cmin && (cmax - cmin + 2) <= OD_WIN && ((reinterpret_cast<uintptr_t>(ghost) >> 3) & 1) == 0;
  if (sysload) {
    if (t < wt.n) ipc_wait_ge(wt.flag[t], wt.want, wt.err, wt.limit);
    __syncthreads();
  }
  if (window) {
    const hipx_int np = (cmax - cmin + 2) >> 1;  // pairs (<= OD_WIN / 2: four per thread); one that would reach beyond the ghost vector's last element is read as a single
    static_assert(OD_WIN / 512 == 4, "four pairs per thread");
    const double *src[4];
    bool          single[4];
    ipc_dbl2      gv[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      hipx_int q = t + 256 * u;
      q          = q < np ? q : np - 1;                            // (threads beyond the window re-read its last pair: every load valid, no branch around the loads)
      single[u]  = cmin + 2 * q + 1 >= ncols;                      // the pair's second element lies beyond the vector: load the pair before it, fix up below
      src[u]     = ghost + cmin + 2 * ((single[u] && q > 0) ? q - 1 : q);
    }
    if (sysload) ipc_load16x4(src[0], src[1], src[2], src[3], gv[0], gv[1], gv[2], gv[3]);  // all in flight, one wait
    else {
#pragma unroll
      for (int u = 0; u < 4; u++) gv[u] = *reinterpret_cast<const ipc_dbl2 *>(src[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const hipx_int q = t + 256 * u;
      if (q < np) {
        if (single[u]) {
          const double *s1 = ghost + cmin + 2 * q;
          s_g[2 * q]     = sysload ? ipc_load8(s1) : s1[0];
          s_g[2 * q + 1] = 0.0;
        } else {
          s_g[2 * q]     = gv[u].x;
          s_g[2 * q + 1] = gv[u].y;
        }
      }
    }
    __syncthreads();
  }
  if (staged) {
#pragma unroll
    for (int u = 0; u < NI; u++) {
      const hipx_int k = k0 + t + 256 * u;
      if (k < k1) s_prod[k - k0] = av[u] * (window ? s_g[cv[u] - cmin] : (sysload ? ipc_load8(ghost + cv[u]) : ghost[cv[u]]));
    }
    __syncthreads();
  }
  double acc = 0.0;
  if (rowok) {
    double sum = w_in;
    if (staged)
      for (hipx_int k = mk0; k < mk1; k++) sum += s_prod[k - k0];
    else
      for (hipx_int k = mk0; k < mk1; k++) sum += ba[k] * (sysload ? ipc_load8(ghost + bj[k]) : ghost[bj[k]]);
    w[myrow] = sum;
    acc      = p_in * sum;
  }
  acc = hipx::wave_sum(acc);
  if ((t & 63) == 0) s_w[t >> 6] = acc;
  __syncthreads();
  if (t == 0) {
    double r = s_w[0];
    r += s_w[1];
    r += s_w[2];
    r += s_w[3];
    // (no agent-scope release fence: it writes back the XCD's L2, full of the product kernel's w / p / x.  The partial goes out as an agent-scope atomic store
    // (sc1: written through), is drained, then the tickets; the last workgroup reads with agent-scope atomic loads: the hand-off spmv_march2_kernel's
    // in-kernel fold uses, pinned by tests/test_gpu_mat.py::test_march2_in_kernel_fold_equals_the_separate_fold_under_load.)  Two ticket levels: one
    // contended word takes ~12 ns per arrival -- 2048 workgroups on one word were 25 us of this kernel; groups of OD_GRP arrive on their own words, the last
    // of a group on the top one.
    __hip_atomic_store(&part[blockIdx.x], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned grp = blockIdx.x / OD_GRP, ngrp = (gridDim.x + OD_GRP - 1) / OD_GRP;
    const unsigned gsz = (grp == ngrp - 1) ? gridDim.x - grp * OD_GRP : OD_GRP;
    bool           last = false;
    if (__hip_atomic_fetch_add(&gticket[grp], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsz - 1) {
      __hip_atomic_store(&gticket[grp], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngrp - 1;
    }
    s_last = last;
  }
  __syncthreads();
  if (!s_last) return;
  double f = 0.0;
  // (the same sums in the same order; eight loads in flight per round instead of one: the last workgroup's fold is the tail of every launch)
  const auto fold = [&](const double *src, const int n) {
    for (int i0 = t; i0 < n; i0 += 8 * 256) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int i = i0 + 256 * u;
        v[u]        = __hip_atomic_load(&src[i < n ? i : i0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (i0 + 256 * u < n) f += v[u];
    }
  };
  fold(partA, npartA);  // (written by the launch before this one)
  fold(part, (int)gridDim.x);
  f = hipx::wave_sum(f);
  __syncthreads();
  if ((t & 63) == 0) s_w[t >> 6] = f;
  __syncthreads();
  if (t == 0) {
    double r = s_w[0];
    r += s_w[1];
    r += s_w[2];
    r += s_w[3];
    dst[0]  = r;
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-armed for the next launch (stream-ordered)
  }
}
}  // namespace
extern "C" {
}  // extern "C"
struct MpiCgPlan {
  hipxMat       B = nullptr;
  int           ok = 0, skip = 0;
  int2         *d_win     = nullptr;  // per 256-row workgroup of offdiag_dot_kernel: the ghost columns its entries touch
  unsigned int *d_gticket = nullptr;  // its group tickets (zero between launches)
};
static std::vector<std::pair<hipxMat, MpiCgPlan>> g_mpicg;  // keyed by Ad (dropped in hipxMatDestroy)
static void mpicg_forget(hipxMat M)
{
  for (size_t k = g_mpicg.size(); k-- > 0;)
    if (g_mpicg[k].first == M || g_mpicg[k].second.B == M) {
      (void)hipFree(g_mpicg[k].second.d_win);
      (void)hipFree(g_mpicg[k].second.d_gticket);
      g_mpicg.erase(g_mpicg.begin() + (long)k);
    }
}
extern "C" {

extern "C" int hipxMatMPICGPlan_(hipxMat A, hipxMat B, int *ok, int *skipmask)
{
  *ok = 0;
  *skipmask = 0;
  for (auto &e : g_mpicg)
    if (e.first == A && e.second.B == B) {
      *ok       = e.second.ok;
      *skipmask = e.second.skip;
      return HIPX_SUCCESS;
    }
  MpiCgPlan pl;
  pl.B = B;
  do {
    if (!A || !B || A->compressed || !B->compressed || B->is64 || A->m != A->n || A->m <= 0 || B->m != A->m) break;
    bool tm = false, ip = false, ips = false;
    int  ierr = use_inode_pair(A, ip, ips);
    if (ierr) return ierr;
    if (ip) break;
    if ((ierr = use_templates(A, tm))) return ierr;
    if (!tm || !march_applies(A) || dev_sw().march1 || dev_sw().trace) break;
    if ((ierr = march2_check(A))) return ierr;
    if (A->march2_state != 1 || !march2_cg_shape(A)) break;
    const hipx_int S = A->march_plan.S, m = A->m, nr = B->nrows_c;
    if (S <= 0 || m % S || m / S < 3 || nr <= 0) break;
    std::vector<hipx_int> ridx((size_t)nr);
    HIPX_HIP(hipMemcpy(ridx.data(), B->d_ridx, sizeof(hipx_int) * (size_t)nr, hipMemcpyDeviceToHost));
    hipx_int nlo = 0, nhi = 0;
    bool     sorted = true;
    for (hipx_int k = 0; k < nr; k++) {
      if (k && ridx[(size_t)k] <= ridx[(size_t)k - 1]) sorted = false;
      if (ridx[(size_t)k] < S) nlo++;
      else if (ridx[(size_t)k] >= m - S) nhi++;
    }
    if (!sorted || nlo + nhi != nr || (nlo && nlo != S) || (nhi && nhi != S)) break;
    const hipx_int g = (nr + 255) / 256;
    HIPX_HIP(hipMalloc((void **)&pl.d_win, sizeof(int2) * (size_t)g));
    HIPX_HIP(hipMalloc((void **)&pl.d_gticket, sizeof(unsigned int) * (size_t)((g + OD_GRP - 1) / OD_GRP)));
    HIPX_HIP(hipMemsetAsync(pl.d_gticket, 0, sizeof(unsigned int) * (size_t)((g + OD_GRP - 1) / OD_GRP), rt().compute));
    offdiag_window_kernel<<<(unsigned)g, 256, 0, rt().compute>>>(nr, (const hipx_int *)B->d_i, B->d_j, pl.d_win);
    HIPX_LAUNCH_CHECK();
    pl.ok   = 1;
    pl.skip = (nlo ? 1 : 0) | (nhi ? 2 : 0);
  } while (0);
  g_mpicg.push_back({A, pl});
  *ok       = pl.ok;
  *skipmask = pl.skip;
  return HIPX_SUCCESS;
}

// w = Ad p_new with the CG direction update as the prologue (cgdir_product); with want_dot the dot partials of the rows outside the planes of `skipmask`
// are left in *dotpart (npart of them); otherwise (exact reductions: the caller runs Dot2 over the complete vectors) no partials
extern "C" int hipxMatMultCGDirectionPartial_(hipxMat A, int skipmask, const double *p_old, double *p_new, const double *z, double dconst, double *x, double b, double a,
                                              const double *dev_beta_new, const double *dev_beta_old, const double *dev_dpi, double *w, int want_dot, int *fused, const double **dotpart,
                                              hipx_int *npart)
{
  *npart   = 0;
  *dotpart = nullptr;
  int ierr = cgdir_product(A, p_old, p_new, z, dconst, x, b, a, dev_beta_new, dev_beta_old, dev_dpi, w, -1, nullptr, skipmask, want_dot != 0, fused, npart);
  if (!ierr && *fused && want_dot) *dotpart = A->d_dotpart;
  return ierr;
}


extern "C" int hipxMatMultAddDotFold_(hipxMat A, hipxMat B, const double *ghost, double *w, const double *p, const double *partA, hipx_int npartA, int slot, double *dst, const hipx::IpcWait *wt)
{
  HIPX_ARG(B && B->compressed && !B->is64 && B->nrows_c > 0, "off-diagonal block: compressed rows with 32-bit offsets expected");
  const MpiCgPlan *pl = nullptr;
  for (auto &e : g_mpicg)
    if (e.first == A && e.second.B == B && e.second.ok) pl = &e.second;
  HIPX_ARG(pl && pl->d_win, "off-diagonal block: hipxMatMPICGPlan_ has not accepted this pair of blocks");
  const hipx_int g = (B->nrows_c + 255) / 256;
  HIPX_ARG(g <= (hipx_int)kMaxRedVals * kRedBlocks, "off-diagonal block: too many rows for the slot's partials");
  offdiag_dot_kernel<<<(unsigned)g, 256, 0, rt().compute>>>(B->nrows_c, (const hipx_int *)B->d_i, B->d_j, B->d_a, B->d_ridx, ghost, w, p, slot_partials(slot), partA, (int)npartA,
                                                            rt().d_tickets + slot, pl->d_gticket, dst, wt ? *wt : IpcWait{}, B->n, pl->d_win);
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}

int hipxMatGetDiagonal(hipxMat A, double *d)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(A && !A->compressed, "MatGetDiagonal: needs an uncompressed matrix");
  if (!A->m) return HIPX_SUCCESS;
  hipx_int g = std::min<hipx_int>((A->m + 255) / 256, 4096);
  getdiag_kernel<<<(unsigned)g, 256, 0, rt().compute>>>(A->m, A->d_diagpos, A->d_a, d);
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}

int hipxPCJacobiSetUp(hipxMat A, double *dinv)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(A && !A->compressed, "PCJacobiSetUp: needs an uncompressed matrix");
  if (!A->m) return HIPX_SUCCESS;
  hipx_int g = std::min<hipx_int>((A->m + 255) / 256, 4096);
  jacobi_setup_kernel<<<(unsigned)g, 256, 0, rt().compute>>>(A->m, A->d_diagpos, A->d_a, dinv);
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// COO assembly on the device (MatSetValuesCOO_SeqAIJ, aij.c:4710-4733).  The integer part -- sorting the (i, j) pairs, merging
// repeats: jmap / perm of MatSetPreallocationCOO_SeqAIJ (aij.c:4524-4707) -- is the reference's own host routine, run once per
// pattern; its two maps live on the device afterwards and every MatSetValuesCOO is one kernel:
//   a[k] = (INSERT ? 0 : a[k]) + sum_{q = jmap[k]}^{jmap[k+1]-1} v[perm[q]]      (left to right: the host loop's order)
struct hipxCOO_s {
  int64_t  nz = 0, ntot = 0;
  int64_t *d_jmap = nullptr, *d_perm = nullptr;
  int64_t *d_imap = nullptr;  // indexed form (hipxCOOCreateIndexed): entry k of the maps belongs to a[imap[k]]
  double  *d_v = nullptr;  // staging for host-resident value arrays
  int64_t  v_cap = 0;
};

namespace {
__global__ __launch_bounds__(256) void coo_setvalues_kernel(int64_t nz, const int64_t *__restrict__ jmap, const int64_t *__restrict__ perm, const double *__restrict__ v, int insert,
                                                            double *__restrict__ a)
{
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < nz; k += (int64_t)gridDim.x * 256) {
    double sum = 0.0;
    for (int64_t q = jmap[k]; q < jmap[k + 1]; q++) sum += v[perm[q]];
    a[k] = (insert ? 0.0 : a[k]) + sum;
  }
}
// remote part of MatSetValuesCOO_MPIAIJ (mpiaij.c:6817-6822): a[imap[k]] += v[perm[q]] one by one, left to right, ON TOP of the value the
// local part left there (no partial sum: the reference adds every received entry to the matrix value itself)
__global__ __launch_bounds__(256) void coo_addindexed_kernel(int64_t nz, const int64_t *__restrict__ imap, const int64_t *__restrict__ jmap, const int64_t *__restrict__ perm,
                                                             const double *__restrict__ v, double *__restrict__ a)
{
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < nz; k += (int64_t)gridDim.x * 256) {
    const int64_t dst = imap[k];
    double        s   = a[dst];
    for (int64_t q = jmap[k]; q < jmap[k + 1]; q++) s += v[perm[q]];
    a[dst] = s;
  }
}
}  // namespace

extern "C" {

int hipxCOOCreateIndexed(int64_t nz, const int64_t *imap, const int64_t *jmap, int64_t ntot, const int64_t *perm, hipxCOO *out)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(nz >= 0 && (imap || !nz), "bad COO maps");
  int ierr = hipxCOOCreate(nz, jmap, ntot, perm, out);
  if (ierr) return ierr;
  HIPX_HIP(hipMalloc((void **)&(*out)->d_imap, sizeof(int64_t) * (size_t)std::max<int64_t>(nz, 1)));
  if (nz) HIPX_HIP(hipMemcpy((*out)->d_imap, imap, sizeof(int64_t) * (size_t)nz, hipMemcpyHostToDevice));
  return HIPX_SUCCESS;
}

int hipxMatAddValuesCOOIndexed(hipxMat A, hipxCOO c, const double *v_dev)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(A && c && c->d_imap && (v_dev || !c->ntot), "null argument / not an indexed COO map");
  if (c->nz) {
    const unsigned g = (unsigned)std::min<int64_t>((c->nz + 255) / 256, 16384);
    coo_addindexed_kernel<<<g, 256, 0, rt().compute>>>(c->nz, c->d_imap, c->d_jmap, c->d_perm, v_dev, A->d_a);
    HIPX_LAUNCH_CHECK();
  }
  A->vd_ready   = false;
  A->tmpl_ready = false;
  A->value_state++;
  hipxSorInvalidate_(A->sor_state);
  hipxSellValuesChanged_(A->sell_state);
  return HIPX_SUCCESS;
}

int hipxCOOCreate(int64_t nz, const int64_t *jmap, int64_t ntot, const int64_t *perm, hipxCOO *out)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(nz >= 0 && ntot >= 0 && out && (jmap || !nz) && (perm || !ntot), "bad COO maps");
  hipxCOO c = new hipxCOO_s;
  c->nz     = nz;
  c->ntot   = ntot;
  HIPX_HIP(hipMalloc((void **)&c->d_jmap, sizeof(int64_t) * ((size_t)nz + 1)));
  HIPX_HIP(hipMalloc((void **)&c->d_perm, sizeof(int64_t) * (size_t)std::max<int64_t>(ntot, 1)));
  static const int64_t zero = 0;
  HIPX_HIP(hipMemcpy(c->d_jmap, nz ? jmap : &zero, sizeof(int64_t) * ((size_t)nz + 1), hipMemcpyHostToDevice));
  if (ntot) HIPX_HIP(hipMemcpy(c->d_perm, perm, sizeof(int64_t) * (size_t)ntot, hipMemcpyHostToDevice));
  *out = c;
  return HIPX_SUCCESS;
}

int hipxCOODestroy(hipxCOO *pc)
{
  if (!pc || !*pc) return HIPX_SUCCESS;
  hipxCOO c = *pc;
  HIPX_HIP(hipStreamSynchronize(rt().compute));
  (void)hipFree(c->d_jmap);
  (void)hipFree(c->d_perm);
  (void)hipFree(c->d_imap);
  (void)hipFree(c->d_v);
  delete c;
  *pc = nullptr;
  return HIPX_SUCCESS;
}

int hipxMatSetValuesCOO(hipxMat A, hipxCOO c, const double *v, int64_t n, int v_on_device, int insert)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(A && c && (v || !n) && n >= 0, "null argument");
  HIPX_ARG(c->nz == A->nnz, "the COO maps were built for another nonzero pattern");
  hipStream_t   st = rt().compute;
  const double *dv = v;
  if (!v_on_device && n) {
    if (n > c->v_cap) {
      HIPX_HIP(hipStreamSynchronize(st));
      (void)hipFree(c->d_v);
      HIPX_HIP(hipMalloc((void **)&c->d_v, sizeof(double) * (size_t)n));
      c->v_cap = n;
    }
    HIPX_HIP(hipMemcpyAsync(c->d_v, v, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st));
    dv = c->d_v;
  }
  if (c->nz) {
    const unsigned g = (unsigned)std::min<int64_t>((c->nz + 255) / 256, 16384);
    coo_setvalues_kernel<<<g, 256, 0, st>>>(c->nz, c->d_jmap, c->d_perm, dv, insert, A->d_a);
    HIPX_LAUNCH_CHECK();
  }
  if (!v_on_device) HIPX_HIP(hipStreamSynchronize(st));  // the caller's host array may change after return
  A->vd_ready   = false;
  A->tmpl_ready = false;
  A->value_state++;
  hipxSorInvalidate_(A->sor_state);
  hipxSellValuesChanged_(A->sell_state);
  return HIPX_SUCCESS;
}

int hipxMatGetValues(hipxMat A, double *a_host)
{
  HIPX_CHECK_INIT();
  HIPX_ARG(A && (a_host || !A->nnz), "null argument");
  if (A->nnz) HIPX_HIP(hipMemcpyAsync(a_host, A->d_a, sizeof(double) * (size_t)A->nnz, hipMemcpyDeviceToHost, rt().compute));
  HIPX_HIP(hipStreamSynchronize(rt().compute));
  return HIPX_SUCCESS;
}

int hipxPointerIsDevice(const void *p, int *is_device)
{
  HIPX_ARG(is_device, "null argument");
  *is_device = 0;
  if (!p) return HIPX_SUCCESS;
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) == hipSuccess) *is_device = (at.type == hipMemoryTypeDevice) ? 1 : 0;
  else (void)hipGetLastError();  // unregistered host memory: not an error
  return HIPX_SUCCESS;
}

}  // extern "C"
