// hipx_sell.hip -- sliced-ELLPACK storage of a CSR matrix (SELL-64) for MatMult / MatMultAdd: SURVEY.md 8(f4), the format the
// reference keeps in src/mat/impls/sell/seq/sell.c (MatMult_SeqSELL sell.c:319-460: slices of rows stored column-major so that
// consecutive lanes read consecutive memory).  Here a slice is 64 rows = one wavefront, lane l owns row 64 s + l:
//
//   values   val[soff[s] + k * 64 + l]                       k-th entry of the lane's row (8 B per lane: one 512-B run per wave and k)
//   columns  col16[coff[s] + (k / 4) * 256 + l * 4 + k % 4]  16-bit code (window id : 4 | offset : 12), 4 entries per 8-byte load;
//            base[s * 16 + id] = first column of window id (<= 16 windows of 4096 columns per slice); matrices with a slice
//            that needs more windows keep 32-bit columns col32[soff[s] + k * 64 + l]
//   len[row] entries of the row (a slice is as wide as its longest row; the padding is never multiplied)
//   Triple-run coding (round 4, "TRI": matrices whose EVERY row consists of runs of three consecutive columns -- the rows of a FEM matrix
//   with three unknowns per node, dense 3 x 3 blocks: MatMult_SeqBAIJ_3's operands, baij2.c:334): ONE 16-bit code per run of three entries
//   (the run's first column; the other two are + 1, + 2): 8 + 2/3 bytes per nonzero instead of 10 -- col16[coff[s] + (k / 12) * 256 + l * 4 + (k / 3) % 4].
//
// Why it exists next to the CSR kernels (hipx_mat.hip): those walk a row with ONE thread after staging it through LDS; with ~80
// entries per row (FEM matrices: Flan_1565) only 25 of a workgroup's 256 threads work in that phase (0.59 of the HBM peak, DESIGN
// section 5).  Here every lane streams its own row: no LDS, no idle lanes, loads of 512 contiguous bytes per wave and entry, and at
// entry k the 64 lanes gather x for 64 NEIGHBOURING rows (stencil / FEM numbering: a handful of cache lines).
// The row sum is formed left to right from 0 (or y_i) with separately rounded products, one lane per row: bit-identical to
// MatMult_SeqAIJ (aij.c:1486-1494) -- the storage order of a row's entries is the CSR order.
#include "hipx_internal.h"
#include "hipx_reduce.h"
#include <algorithm>
#include <cstdlib>
#include <vector>

using namespace hipx;

extern "C" int hipxMatInternal_(hipxMat A, hipx_int *m, hipx_int *n, int64_t *nnz, int *is64, void **d_i, hipx_int **d_j, double **d_a, int64_t **d_diagpos, int *diag_dense,
                                int *compressed, void ***sor_slot, unsigned long long *value_state);

namespace {

constexpr int SELL_C    = 64;
constexpr int SELL_WMAX = 16;   // windows per slice
constexpr int SELL_WLEN = 4096; // columns per window
constexpr int64_t kSlack = 16 * 64;  // slots past the last slice the unrolled stream may read (never used)

struct SellState {
  bool               built = false, ok = false, packed = false, tri = false;
  unsigned long long vstate = 0;
  hipx_int           m = 0, n = 0, nslices = 0;
  int64_t            total = 0, ctotal = 0;  // value slots / code slots
  int64_t           *d_soff = nullptr, *d_coff = nullptr;
  unsigned short    *d_len = nullptr;
  double            *d_val = nullptr;
  unsigned short    *d_col16 = nullptr;
  hipx_int          *d_base = nullptr, *d_col32 = nullptr;
  double             pad_ratio = 0.0;
  int64_t            bytes = 0;
};

void sell_free(SellState *S)
{
  (void)hipFree(S->d_soff);
  (void)hipFree(S->d_coff);
  (void)hipFree(S->d_len);
  (void)hipFree(S->d_val);
  (void)hipFree(S->d_col16);
  (void)hipFree(S->d_base);
  (void)hipFree(S->d_col32);
  *S = SellState();
}

template <typename IT>
__global__ __launch_bounds__(256) void sell_width_kernel(hipx_int m, const IT *__restrict__ ai, int *__restrict__ width, unsigned int *toolong)
{
  const hipx_int row = (hipx_int)blockIdx.x * 256 + threadIdx.x;
  long long      len = 0;
  if (row < m) len = (long long)(ai[row + 1] - ai[row]);
  if (len > 65535) {
    atomicAdd(toolong, 1u);
    len = 65535;
  }
  int w = (int)len;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) w = max(w, __shfl_xor(w, o));
  const hipx_int s = row >> 6;
  if ((threadIdx.x & 63) == 0 && (hipx_int)blockIdx.x * 256 + (threadIdx.x & ~63) < m) width[s] = w;
}

// does every row consist of runs of three consecutive columns?  (flag <- 1 otherwise)
template <typename IT>
__global__ __launch_bounds__(256) void sell_tri_check_kernel(hipx_int m, const IT *__restrict__ ai, const hipx_int *__restrict__ aj, unsigned int *flag)
{
  const hipx_int row = (hipx_int)blockIdx.x * 256 + threadIdx.x;
  if (row >= m) return;
  const IT k0 = ai[row], k1 = ai[row + 1];
  bool     bad = ((long long)(k1 - k0) % 3) != 0;
  for (IT k = k0; k + 2 < k1 && !bad; k += 3) bad = aj[k + 1] != aj[k] + 1 || aj[k + 2] != aj[k] + 2;
  if (bad) atomicOr(flag, 1u);
}

// one wave per slice: window keys of the slice (LDS set), then the lane copies its row into the slice.  TRI: one code per run of three entries.
template <typename IT, bool TRI = false>
__global__ __launch_bounds__(256) void sell_fill_kernel(hipx_int m, hipx_int nslices, const IT *__restrict__ ai, const hipx_int *__restrict__ aj, const double *__restrict__ aa,
                                                        const int64_t *__restrict__ soff, const int64_t *__restrict__ coff, unsigned short *__restrict__ lenout, double *__restrict__ val,
                                                        unsigned short *__restrict__ col16, hipx_int *__restrict__ base, hipx_int *__restrict__ col32, unsigned int *unpackable)
{
  __shared__ int keys[4][64];
  __shared__ int sorted[4][SELL_WMAX];
  __shared__ int nkeys[4];
  const int      wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const hipx_int s = (hipx_int)blockIdx.x * 4 + wave;
  const bool     live = s < nslices;
  const hipx_int row = live ? s * SELL_C + lane : m;
  IT             k0 = 0;
  int            len = 0;
  if (row < m) {
    k0  = ai[row];
    len = (int)min((long long)(ai[row + 1] - k0), 65535LL);
  }
  const int64_t so = live ? soff[s] : 0;
  const int     w  = live ? (int)((soff[s + 1] - so) >> 6) : 0;
  keys[wave][lane] = -1;
  if (lane == 0) nkeys[wave] = 0;
  __syncthreads();
  if (col16) {
    for (int k = 0; k < len; k++) {
      const int key = aj[k0 + k] >> 12;
      unsigned  h   = ((unsigned)key * 0x9E3779B1u) >> 26;
      for (int probe = 0; probe < 64; probe++) {
        const int old = atomicCAS(&keys[wave][h], -1, key);
        if (old == -1) {
          atomicAdd(&nkeys[wave], 1);
          break;
        }
        if (old == key) break;
        h = (h + 1) & 63;
      }
      if (nkeys[wave] > SELL_WMAX) break;
    }
  }
  __syncthreads();
  const bool pack = col16 && nkeys[wave] <= SELL_WMAX;
  if (col16 && live && !pack && lane == 0) atomicAdd(unpackable, 1u);
  if (pack && lane == 0) {
    int n = 0;
    for (int q = 0; q < 64; q++)
      if (keys[wave][q] != -1) {
        int key = keys[wave][q], i = n++;
        while (i > 0 && sorted[wave][i - 1] > key) {
          sorted[wave][i] = sorted[wave][i - 1];
          i--;
        }
        sorted[wave][i] = key;
      }
    for (int q = n; q < SELL_WMAX; q++) sorted[wave][q] = n ? sorted[wave][n - 1] : 0;
  }
  __syncthreads();
  if (!live) return;
  if (pack && lane < SELL_WMAX) base[(size_t)s * SELL_WMAX + lane] = sorted[wave][lane] << 12;
  if (row < m) lenout[row] = (unsigned short)len;
  const int64_t co = col16 ? coff[s] : 0;
  for (int k = 0; k < w; k++) {
    const bool     on = k < len;
    const hipx_int c  = on ? aj[k0 + k] : (len ? aj[k0 + len - 1] : 0);  // padding: a column the row touches anyway (never multiplied)
    val[so + (int64_t)k * SELL_C + lane] = on ? aa[k0 + k] : 0.0;
    if (pack) {
      const int key = c >> 12;
      int       id  = 0;
#pragma unroll
      for (int i = SELL_WMAX - 1; i >= 0; i--)
        if (sorted[wave][i] == key) id = i;
      if (!TRI) col16[co + (int64_t)(k >> 2) * 256 + lane * 4 + (k & 3)] = (unsigned short)((id << 12) | (c & (SELL_WLEN - 1)));
      else if (k % 3 == 0) {  // the run's first column (padding runs repeat the row's last run start: never multiplied)
        const int      t  = k / 3;
        const hipx_int c0 = on ? c : (len ? aj[k0 + len - 3] : 0);
        int            i0 = 0;
#pragma unroll
        for (int i = SELL_WMAX - 1; i >= 0; i--)
          if (sorted[wave][i] == (c0 >> 12)) i0 = i;
        col16[co + (int64_t)(t >> 2) * 256 + lane * 4 + (t & 3)] = (unsigned short)((i0 << 12) | (c0 & (SELL_WLEN - 1)));
      }
    } else if (col32) col32[so + (int64_t)k * SELL_C + lane] = c;
  }
  if (pack) {
    const int nc = TRI ? w / 3 : w;
    for (int k = nc; k < ((nc + 3) & ~3); k++) col16[co + (int64_t)(k >> 2) * 256 + lane * 4 + (k & 3)] = 0;
  }
}

// values only (same pattern, new numbers)
template <typename IT>
__global__ __launch_bounds__(256) void sell_values_kernel(hipx_int m, hipx_int nslices, const IT *__restrict__ ai, const double *__restrict__ aa, const int64_t *__restrict__ soff,
                                                          double *__restrict__ val)
{
  const int      wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const hipx_int s = (hipx_int)blockIdx.x * 4 + wave;
  if (s >= nslices) return;
  const hipx_int row = s * SELL_C + lane;
  IT             k0 = 0;
  int            len = 0;
  if (row < m) {
    k0  = ai[row];
    len = (int)min((long long)(ai[row + 1] - k0), 65535LL);
  }
  const int64_t so = soff[s];
  const int     w  = (int)((soff[s + 1] - so) >> 6);
  for (int k = 0; k < w; k++) val[so + (int64_t)k * SELL_C + lane] = k < len ? aa[k0 + k] : 0.0;
}

// MODE 0: y = A x; MODE 1: z = y + A x (the sum starts from y_i, aij.c:1648).  DOT: one partial of x . y per slice (fixed order).
// U entries per pass: U value loads + U / 4 code loads in flight, then U gathers in flight.
// PAIR: the row sums of MatMult_SeqAIJ_Inode / MatMultAdd_SeqAIJ_Inode (inode.c:356-560, 563-760), which the reference runs on a matrix with
// inodes (aij.c:1459, 1617): the terms enter in pairs, sum += a[k] x[j_k] + a[k+1] x[j_k+1], a last odd term alone.
template <int MODE, bool DOT, bool PACK, int U, bool TRI = false, bool PAIR = false>
__global__ __launch_bounds__(256) void spmv_sell_kernel(hipx_int m, hipx_int nslices, hipx_int slices_per_xcd, const int64_t *__restrict__ soff, const int64_t *__restrict__ coff,
                                                        const unsigned short *__restrict__ lens, const double *__restrict__ val, const unsigned short *__restrict__ col16,
                                                        const hipx_int *__restrict__ base, const hipx_int *__restrict__ col32, const double *__restrict__ x, const double *yin, double *yout,
                                                        double *dotpart)
{
  // hardware block b runs on XCD b % 8 (observed; locality only): every XCD walks one contiguous slab of slices
  const hipx_int b = (hipx_int)blockIdx.x, xcd = b & 7, q = b >> 3;
  const int      wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const hipx_int s = xcd * slices_per_xcd + q * 4 + wave;
  if (q * 4 + wave >= slices_per_xcd || s >= nslices) return;
  const hipx_int row = s * SELL_C + lane;
  const int      len = row < m ? (int)lens[row] : 0;
  const int64_t  so = soff[s];
  const int      w  = (int)((soff[s + 1] - so) >> 6);
  const double  *vp = val + so + lane;
  double         sum = (MODE == 1 && row < m) ? yin[row] : 0.0;
  int            mybase = 0;
  const unsigned short *cp = nullptr;
  const hipx_int       *cq = nullptr;
  if (PACK) {
    mybase = base[(size_t)s * SELL_WMAX + (lane & (SELL_WMAX - 1))];
    cp     = col16 + coff[s] + lane * 4;
  } else cq = col32 + so + lane;
  for (int k = 0; k < w; k += U) {
    double a[U];
    int    c[U];
#pragma unroll
    for (int e = 0; e < U; e++) a[e] = vp[(int64_t)(k + e) * SELL_C];  // (the arrays carry U * 64 slots of slack: no bounds test on the stream)
    if (PACK && TRI) {  // U = 12: four runs of three entries per 8-byte code load
      static_assert(!TRI || U == 12, "triple-run coding walks 12 entries per pass");
      const uint2    cw = *reinterpret_cast<const uint2 *>(cp + (int64_t)(k / 12) * 256);
      const unsigned cc[4] = {cw.x & 0xffffu, cw.x >> 16, cw.y & 0xffffu, cw.y >> 16};
#pragma unroll
      for (int f = 0; f < 4; f++) {
        const int c0 = __shfl(mybase, (int)(cc[f] >> 12), SELL_WMAX) + (int)(cc[f] & (SELL_WLEN - 1));
        if (3 * f + 2 < U) {
          c[3 * f]     = c0;
          c[3 * f + 1] = c0 + 1;
          c[3 * f + 2] = c0 + 2;
        }
      }
    } else if (PACK) {
#pragma unroll
      for (int e = 0; e < U; e += 4) {
        const uint2 cw = *reinterpret_cast<const uint2 *>(cp + (int64_t)((k + e) >> 2) * 256);
        const unsigned cc[4] = {cw.x & 0xffffu, cw.x >> 16, cw.y & 0xffffu, cw.y >> 16};
#pragma unroll
        for (int f = 0; f < 4; f++)
          if (e + f < U) c[e + f] = __shfl(mybase, (int)(cc[f] >> 12), SELL_WMAX) + (int)(cc[f] & (SELL_WLEN - 1));
      }
    } else {
#pragma unroll
      for (int e = 0; e < U; e++) c[e] = cq[(int64_t)(k + e) * SELL_C];
    }
    double xv[U];
#pragma unroll
    for (int e = 0; e < U; e++) xv[e] = (k + e < len) ? x[c[e]] : 0.0;
    if (PAIR) {
      static_assert(U % 2 == 0, "pairs never straddle passes");
#pragma unroll
      for (int e = 0; e < U; e += 2) {
        if (k + e + 1 < len) sum += a[e] * xv[e] + a[e + 1] * xv[e + 1];
        else if (k + e < len) sum += a[e] * xv[e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < U; e++)
        if (k + e < len) sum += a[e] * xv[e];  // product rounded, then the sum: left to right in the row's CSR order
    }
  }
  if (row < m) yout[row] = sum;
  if (DOT) {
    const double p = (row < m) ? x[row] * sum : 0.0;
    const double t = hipx::wave_sum(p);
    if (lane == 0) dotpart[s] = t;
  }
}

}  // namespace

extern "C" {

void hipxSellFree_(void *p)
{
  if (!p) return;
  SellState *S = (SellState *)p;
  sell_free(S);
  delete S;
}

// values changed (same pattern): the slices are refilled at the next product
void hipxSellValuesChanged_(void *p)
{
  if (p) ((SellState *)p)->vstate = 0;
}

// pattern changed
void hipxSellInvalidate_(void *p)
{
  if (p) {
    SellState *S = (SellState *)p;
    sell_free(S);
  }
}

// builds / refreshes the SELL copy.  *ok = 0: the format does not apply (rows longer than 65535, or more than `max_pad` padding)
int hipxSellEnsure_(hipxMat A, void **slot, int *ok, int *packed, double *pad_ratio, int64_t *bytes)
{
  hipx_int           m, n;
  int64_t            nnz;
  int                is64, diag_dense, compressed;
  void              *d_i;
  hipx_int          *d_j;
  double            *d_a;
  int64_t           *d_diagpos;
  void             **sor_slot;
  unsigned long long vstate;
  int ierr = hipxMatInternal_(A, &m, &n, &nnz, &is64, &d_i, &d_j, &d_a, &d_diagpos, &diag_dense, &compressed, &sor_slot, &vstate);
  if (ierr) return ierr;
  *ok = 0;
  if (compressed || m <= 0 || nnz <= 0) return HIPX_SUCCESS;
  SellState *S = (SellState *)*slot;
  if (!S) {
    S     = new SellState;
    *slot = S;
  }
  hipStream_t st = rt().compute;
  if (S->built && S->ok && S->vstate != vstate) {  // new values on the same pattern
    const unsigned g = (unsigned)((S->nslices + 3) / 4);
    if (is64) sell_values_kernel<int64_t><<<g, 256, 0, st>>>(m, S->nslices, (const int64_t *)d_i, d_a, S->d_soff, S->d_val);
    else sell_values_kernel<hipx_int><<<g, 256, 0, st>>>(m, S->nslices, (const hipx_int *)d_i, d_a, S->d_soff, S->d_val);
    HIPX_LAUNCH_CHECK();
    S->vstate = vstate;
  }
  if (!S->built) {
    S->built   = true;
    S->m       = m;
    S->n       = n;
    S->nslices = (m + SELL_C - 1) / SELL_C;
    static const double max_pad = getenv("HIPX_SELL_MAXPAD") ? atof(getenv("HIPX_SELL_MAXPAD")) : 1.25;
    int          *d_w = nullptr;
    unsigned int *d_cnt = nullptr;
    HIPX_HIP(hipMalloc((void **)&d_w, sizeof(int) * (size_t)S->nslices));
    HIPX_HIP(hipMalloc((void **)&d_cnt, sizeof(unsigned int) * 2));
    HIPX_HIP(hipMemsetAsync(d_cnt, 0, sizeof(unsigned int) * 2, st));
    const unsigned g = (unsigned)((m + 255) / 256);
    if (is64) sell_width_kernel<int64_t><<<g, 256, 0, st>>>(m, (const int64_t *)d_i, d_w, d_cnt);
    else sell_width_kernel<hipx_int><<<g, 256, 0, st>>>(m, (const hipx_int *)d_i, d_w, d_cnt);
    std::vector<int> w((size_t)S->nslices);
    unsigned int     cnt[2] = {0, 0};
    HIPX_HIP(hipMemcpyAsync(w.data(), d_w, sizeof(int) * (size_t)S->nslices, hipMemcpyDeviceToHost, st));
    HIPX_HIP(hipMemcpyAsync(cnt, d_cnt, sizeof(cnt), hipMemcpyDeviceToHost, st));
    HIPX_HIP(hipStreamSynchronize(st));
    HIPX_LAUNCH_CHECK();
    (void)hipFree(d_w);
    // triple-run coding: every row a sequence of runs of three consecutive columns (3 unknowns per node, dense 3 x 3 blocks)?
    static const bool notri = getenv("HIPX_SELL_NOTRI") != nullptr;
    S->tri = false;
    if (!notri && !cnt[0] && nnz % 3 == 0) {
      HIPX_HIP(hipMemsetAsync(d_cnt, 0, sizeof(unsigned int) * 2, st));
      if (is64) sell_tri_check_kernel<int64_t><<<g, 256, 0, st>>>(m, (const int64_t *)d_i, d_j, d_cnt);
      else sell_tri_check_kernel<hipx_int><<<g, 256, 0, st>>>(m, (const hipx_int *)d_i, d_j, d_cnt);
      unsigned int bad = 1;
      HIPX_HIP(hipMemcpyAsync(&bad, d_cnt, sizeof(bad), hipMemcpyDeviceToHost, st));
      HIPX_HIP(hipStreamSynchronize(st));
      HIPX_LAUNCH_CHECK();
      S->tri = bad == 0;
      HIPX_HIP(hipMemsetAsync(d_cnt, 0, sizeof(unsigned int) * 2, st));
    }
    std::vector<int64_t> soff((size_t)S->nslices + 1, 0), coff((size_t)S->nslices + 1, 0);
    for (hipx_int s = 0; s < S->nslices; s++) {
      const int ncodes = S->tri ? w[(size_t)s] / 3 : w[(size_t)s];
      soff[(size_t)s + 1] = soff[(size_t)s] + (int64_t)w[(size_t)s] * SELL_C;
      coff[(size_t)s + 1] = coff[(size_t)s] + (int64_t)((ncodes + 3) & ~3) * SELL_C;
    }
    S->total     = soff[(size_t)S->nslices];
    S->ctotal    = coff[(size_t)S->nslices];
    S->pad_ratio = (double)S->total / (double)nnz;
    if (cnt[0] || S->pad_ratio > max_pad) {  // a row beyond 65535 entries, or too ragged: the CSR kernels stay
      (void)hipFree(d_cnt);
      *pad_ratio = S->pad_ratio;
      return HIPX_SUCCESS;
    }
    HIPX_HIP(hipMalloc((void **)&S->d_soff, sizeof(int64_t) * ((size_t)S->nslices + 1)));
    HIPX_HIP(hipMalloc((void **)&S->d_coff, sizeof(int64_t) * ((size_t)S->nslices + 1)));
    HIPX_HIP(hipMalloc((void **)&S->d_len, sizeof(unsigned short) * (size_t)m));
    HIPX_HIP(hipMalloc((void **)&S->d_val, sizeof(double) * (size_t)(S->total + kSlack)));
    HIPX_HIP(hipMalloc((void **)&S->d_col16, sizeof(unsigned short) * (size_t)(S->ctotal + 4 * kSlack)));
    HIPX_HIP(hipMemsetAsync(S->d_col16, 0, sizeof(unsigned short) * (size_t)(S->ctotal + 4 * kSlack), st));
    HIPX_HIP(hipMalloc((void **)&S->d_base, sizeof(hipx_int) * (size_t)S->nslices * SELL_WMAX));
    HIPX_HIP(hipMemcpyAsync(S->d_soff, soff.data(), sizeof(int64_t) * soff.size(), hipMemcpyHostToDevice, st));
    HIPX_HIP(hipMemcpyAsync(S->d_coff, coff.data(), sizeof(int64_t) * coff.size(), hipMemcpyHostToDevice, st));
    const unsigned gs = (unsigned)((S->nslices + 3) / 4);
    if (S->tri) {
      if (is64) sell_fill_kernel<int64_t, true><<<gs, 256, 0, st>>>(m, S->nslices, (const int64_t *)d_i, d_j, d_a, S->d_soff, S->d_coff, S->d_len, S->d_val, S->d_col16, S->d_base, nullptr, d_cnt + 1);
      else sell_fill_kernel<hipx_int, true><<<gs, 256, 0, st>>>(m, S->nslices, (const hipx_int *)d_i, d_j, d_a, S->d_soff, S->d_coff, S->d_len, S->d_val, S->d_col16, S->d_base, nullptr, d_cnt + 1);
    } else if (is64) sell_fill_kernel<int64_t><<<gs, 256, 0, st>>>(m, S->nslices, (const int64_t *)d_i, d_j, d_a, S->d_soff, S->d_coff, S->d_len, S->d_val, S->d_col16, S->d_base, nullptr, d_cnt + 1);
    else sell_fill_kernel<hipx_int><<<gs, 256, 0, st>>>(m, S->nslices, (const hipx_int *)d_i, d_j, d_a, S->d_soff, S->d_coff, S->d_len, S->d_val, S->d_col16, S->d_base, nullptr, d_cnt + 1);
    HIPX_HIP(hipMemcpyAsync(cnt, d_cnt, sizeof(cnt), hipMemcpyDeviceToHost, st));
    HIPX_HIP(hipStreamSynchronize(st));  // soff / coff (host vectors) are read by the copies above
    HIPX_LAUNCH_CHECK();
    S->packed = cnt[1] == 0;
    if (!S->packed) S->tri = false;
    if (!S->packed) {  // some slice spans more than 16 column windows: 32-bit columns for the whole matrix
      (void)hipFree(S->d_col16);
      (void)hipFree(S->d_base);
      S->d_col16 = nullptr;
      S->d_base  = nullptr;
      HIPX_HIP(hipMalloc((void **)&S->d_col32, sizeof(hipx_int) * (size_t)(S->total + kSlack)));
      HIPX_HIP(hipMemsetAsync(S->d_col32, 0, sizeof(hipx_int) * (size_t)(S->total + kSlack), st));
      if (is64) sell_fill_kernel<int64_t><<<gs, 256, 0, st>>>(m, S->nslices, (const int64_t *)d_i, d_j, d_a, S->d_soff, S->d_coff, S->d_len, S->d_val, nullptr, nullptr, S->d_col32, d_cnt + 1);
      else sell_fill_kernel<hipx_int><<<gs, 256, 0, st>>>(m, S->nslices, (const hipx_int *)d_i, d_j, d_a, S->d_soff, S->d_coff, S->d_len, S->d_val, nullptr, nullptr, S->d_col32, d_cnt + 1);
      HIPX_HIP(hipStreamSynchronize(st));
      HIPX_LAUNCH_CHECK();
    }
    (void)hipFree(d_cnt);
    S->bytes  = 8 * S->total + (S->packed ? 2 * S->ctotal + 4 * (int64_t)S->nslices * SELL_WMAX : 4 * S->total) + 2 * (int64_t)m + 16 * ((int64_t)S->nslices + 1);
    S->ok     = true;
    S->vstate = vstate;
  }
  *ok        = S->ok ? 1 : 0;
  *packed    = S->packed ? 1 : 0;
  *pad_ratio = S->pad_ratio;
  *bytes     = S->bytes;
  return HIPX_SUCCESS;
}

hipx_int hipxSellDotPartials_(void *p) { return p ? ((SellState *)p)->nslices : 0; }

int hipxSellLaunch_(void *p, int mode, int dot, const double *x, const double *yin, double *yout, double *dotpart, int pair)
{
  SellState *S = (SellState *)p;
  if (!S || !S->ok) return fail(HIPX_ERR_ORDER, "SELL copy not built", __FILE__, __LINE__);
  const hipx_int spx  = (((S->nslices + 7) / 8) + 3) / 4 * 4;  // slices per XCD, whole workgroups
  const unsigned grid = (unsigned)((spx / 4) * 8);
  static const int u = getenv("HIPX_SELL_U") ? atoi(getenv("HIPX_SELL_U")) : 8;
  if (pair) {  // a matrix with inodes: the pairwise row sums of MatMult_SeqAIJ_Inode
#define HIPX_SELL_PAIR(MODE, DOT) \
  do { \
    if (S->tri && S->packed) \
      spmv_sell_kernel<MODE, DOT, true, 12, true, true><<<grid, 256, 0, rt().compute>>>(S->m, S->nslices, spx, S->d_soff, S->d_coff, S->d_len, S->d_val, S->d_col16, S->d_base, S->d_col32, x, yin, yout, dotpart); \
    else if (S->packed) \
      spmv_sell_kernel<MODE, DOT, true, 8, false, true><<<grid, 256, 0, rt().compute>>>(S->m, S->nslices, spx, S->d_soff, S->d_coff, S->d_len, S->d_val, S->d_col16, S->d_base, S->d_col32, x, yin, yout, dotpart); \
    else \
      spmv_sell_kernel<MODE, DOT, false, 8, false, true><<<grid, 256, 0, rt().compute>>>(S->m, S->nslices, spx, S->d_soff, S->d_coff, S->d_len, S->d_val, S->d_col16, S->d_base, S->d_col32, x, yin, yout, dotpart); \
  } while (0)
    if (mode == 0 && !dot) HIPX_SELL_PAIR(0, false);
    else if (mode == 0) HIPX_SELL_PAIR(0, true);
    else if (!dot) HIPX_SELL_PAIR(1, false);
    else HIPX_SELL_PAIR(1, true);
#undef HIPX_SELL_PAIR
    HIPX_LAUNCH_CHECK();
    return HIPX_SUCCESS;
  }
  if (S->tri && S->packed) {  // one code per run of three entries, 12 entries per pass
#define HIPX_SELL_TRI(MODE, DOT) \
  spmv_sell_kernel<MODE, DOT, true, 12, true><<<grid, 256, 0, rt().compute>>>(S->m, S->nslices, spx, S->d_soff, S->d_coff, S->d_len, S->d_val, S->d_col16, S->d_base, S->d_col32, x, yin, yout, dotpart)
    if (mode == 0 && !dot) HIPX_SELL_TRI(0, false);
    else if (mode == 0) HIPX_SELL_TRI(0, true);
    else if (!dot) HIPX_SELL_TRI(1, false);
    else HIPX_SELL_TRI(1, true);
#undef HIPX_SELL_TRI
    HIPX_LAUNCH_CHECK();
    return HIPX_SUCCESS;
  }
#define HIPX_SELL_GO(MODE, DOT, PACK, UU) \
  spmv_sell_kernel<MODE, DOT, PACK, UU><<<grid, 256, 0, rt().compute>>>(S->m, S->nslices, spx, S->d_soff, S->d_coff, S->d_len, S->d_val, S->d_col16, S->d_base, S->d_col32, x, yin, yout, dotpart)
#define HIPX_SELL_U(MODE, DOT, PACK) \
  do { \
    if (u == 4) HIPX_SELL_GO(MODE, DOT, PACK, 4); \
    else HIPX_SELL_GO(MODE, DOT, PACK, 8); \
  } while (0)
#define HIPX_SELL_P(MODE, DOT) \
  do { \
    if (S->packed) HIPX_SELL_U(MODE, DOT, true); \
    else HIPX_SELL_U(MODE, DOT, false); \
  } while (0)
  if (mode == 0 && !dot) HIPX_SELL_P(0, false);
  else if (mode == 0) HIPX_SELL_P(0, true);
  else if (!dot) HIPX_SELL_P(1, false);
  else HIPX_SELL_P(1, true);
#undef HIPX_SELL_P
#undef HIPX_SELL_U
#undef HIPX_SELL_GO
  HIPX_LAUNCH_CHECK();
  return HIPX_SUCCESS;
}

}  // extern "C"
