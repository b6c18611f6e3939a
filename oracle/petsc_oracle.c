/*
 * petsc_oracle.c -- CPU restatement of the reference KSP hot path.  TEST INFRASTRUCTURE ONLY
 * (see petsc_oracle.h).  Build: gcc -O2 -ffp-contract=off -fPIC -shared (oracle/Makefile).
 *
 * Each function follows the cited reference routine statement by statement where the arithmetic
 * order matters (SpMV row sums, SOR sweeps, AYPX/WAXPY/AXPBY special cases, Givens updates).
 */
#include "petsc_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ============================ synthetic operators ============================================ */

/* src/ksp/ksp/tutorials/ex2.c:70-94.  Ii = i*n + j, j fastest; -1 at Ii-n (i>0), Ii+n (i<m-1),
   Ii-1 (j>0), Ii+1 (j<n-1); 4 on the diagonal.  MatAssemblyEnd_SeqAIJ keeps columns sorted, so
   the assembled row is (Ii-n, Ii-1, Ii, Ii+1, Ii+n). */
int64_t orc_laplace2d_5pt(OInt m, OInt n, OInt rstart, OInt rend, OInt *ai, OInt *aj, OScalar *aa)
{
  int64_t nz = 0;
  for (OInt Ii = rstart; Ii < rend; Ii++) {
    OInt i = Ii / n, j = Ii - i * n;
    if (ai) ai[Ii - rstart] = (OInt)nz;
#define PUT(J, V) \
  do { \
    if (ai) { aj[nz] = (J); aa[nz] = (V); } \
    nz++; \
  } while (0)
    if (i > 0) PUT(Ii - n, -1.0);
    if (j > 0) PUT(Ii - 1, -1.0);
    PUT(Ii, 4.0);
    if (j < n - 1) PUT(Ii + 1, -1.0);
    if (i < m - 1) PUT(Ii + n, -1.0);
  }
  if (ai) ai[rend - rstart] = (OInt)nz;
  return nz;
}

/* 3-D analogue of ex2: Ii = x + n*y + n*n*z; 6 on the diagonal, -1 at +-1, +-n, +-n^2. */
int64_t orc_poisson3d_7pt(OInt n, OInt rstart, OInt rend, OInt *ai, OInt *aj, OScalar *aa)
{
  int64_t nz = 0;
  const OInt n2 = n * n;
  for (OInt Ii = rstart; Ii < rend; Ii++) {
    OInt x = Ii % n, y = (Ii / n) % n, z = Ii / n2;
    if (ai) ai[Ii - rstart] = (OInt)nz;
    if (z > 0) PUT(Ii - n2, -1.0);
    if (y > 0) PUT(Ii - n, -1.0);
    if (x > 0) PUT(Ii - 1, -1.0);
    PUT(Ii, 6.0);
    if (x < n - 1) PUT(Ii + 1, -1.0);
    if (y < n - 1) PUT(Ii + n, -1.0);
    if (z < n - 1) PUT(Ii + n2, -1.0);
  }
  if (ai) ai[rend - rstart] = (OInt)nz;
  return nz;
}

/* the same operator on an nx x ny x nz box (x fastest): BASELINE config 5's per-GPU share is 1024 x 1024 x 128 */
int64_t orc_poisson3d_7pt_box(OInt nx, OInt ny, OInt nzz, OInt rstart, OInt rend, OInt *ai, OInt *aj, OScalar *aa)
{
  int64_t    nz = 0;
  const OInt n2 = nx * ny;
  for (OInt Ii = rstart; Ii < rend; Ii++) {
    OInt x = Ii % nx, y = (Ii / nx) % ny, z = Ii / n2;
    if (ai) ai[Ii - rstart] = (OInt)nz;
    if (z > 0) PUT(Ii - n2, -1.0);
    if (y > 0) PUT(Ii - nx, -1.0);
    if (x > 0) PUT(Ii - 1, -1.0);
    PUT(Ii, 6.0);
    if (x < nx - 1) PUT(Ii + 1, -1.0);
    if (y < ny - 1) PUT(Ii + nx, -1.0);
    if (z < nzz - 1) PUT(Ii + n2, -1.0);
  }
  if (ai) ai[rend - rstart] = (OInt)nz;
  return nz;
}

/* src/ksp/ksp/tutorials/bench_kspsolve.c:115-303 (FillCOO): h = 1/(n-1); corner -h/13, edge -3h/26,
   face -3h/13, centre 44h/13, written exactly as the reference writes them (-1.0/13*h, ...).  The COO
   triples are sorted by column at assembly, so the row is emitted in increasing column order
   (dz outer, dy, dx inner). */
int64_t orc_poisson3d_27pt(OInt n, OInt rstart, OInt rend, OInt *ai, OInt *aj, OScalar *aa)
{
  int64_t       nz = 0;
  const OInt    n2 = n * n, n1 = n - 1;
  const OScalar h     = 1.0 / (n - 1);
  const OScalar vcorn = -1.0 / 13 * h, vedge = -3.0 / 26 * h, vface = -3.0 / 13 * h, vcent = 44.0 / 13 * h;
  for (OInt Ii = rstart; Ii < rend; Ii++) {
    OInt x = Ii % n, y = (Ii / n) % n, z = Ii / n2;
    if (ai) ai[Ii - rstart] = (OInt)nz;
    for (int dz = -1; dz <= 1; dz++) {
      if ((dz < 0 && z == 0) || (dz > 0 && z == n1)) continue;
      for (int dy = -1; dy <= 1; dy++) {
        if ((dy < 0 && y == 0) || (dy > 0 && y == n1)) continue;
        for (int dx = -1; dx <= 1; dx++) {
          if ((dx < 0 && x == 0) || (dx > 0 && x == n1)) continue;
          int     order = (dx != 0) + (dy != 0) + (dz != 0);
          OScalar v     = order == 3 ? vcorn : order == 2 ? vedge : order == 1 ? vface : vcent;
          PUT(Ii + dx + dy * n + dz * n2, v);
        }
      }
    }
  }
  if (ai) ai[rend - rstart] = (OInt)nz;
  return nz;
}
#undef PUT

/* ============================ Mat_SeqAIJ ====================================================== */

/* src/mat/impls/aij/seq/aij.c:1486-1494 with PetscSparseDensePlusDot (aij.h:608-614, the plain-loop
   branch: no unroll macro, no AVX-512 at -O2 without -march): sum starts at 0 and accumulates
   left to right, one rounded multiply and one rounded add per entry. */
void orc_MatMult_SeqAIJ(OInt m, const OInt *ai, const OInt *aj, const OScalar *aa, const OScalar *x, OScalar *y)
{
  for (OInt i = 0; i < m; i++) {
    OInt           n   = ai[i + 1] - ai[i];
    const OInt    *idx = aj + ai[i];
    const OScalar *v   = aa + ai[i];
    OScalar        sum = 0.0;
    for (OInt k = 0; k < n; k++) sum += v[k] * x[idx[k]];
    y[i] = sum;
  }
}

/* aij.c:1467-1481: compressed-row variant (y zeroed, only rows listed in ridx are computed). */
void orc_MatMult_SeqAIJ_cprow(OInt m, OInt nrows, const OInt *ci, const OInt *ridx, const OInt *aj, const OScalar *aa, const OScalar *x, OScalar *y)
{
  memset(y, 0, (size_t)m * sizeof(OScalar));
  for (OInt i = 0; i < nrows; i++) {
    OInt    n   = ci[i + 1] - ci[i];
    OScalar sum = 0.0;
    for (OInt k = 0; k < n; k++) sum += aa[ci[i] + k] * x[aj[ci[i] + k]];
    y[ridx[i]] = sum;
  }
}

/* aij.c:1606-1658: z_i = y_i + sum_k a_k x_jk, the sum seeded with y_i. */
void orc_MatMultAdd_SeqAIJ(OInt m, const OInt *ai, const OInt *aj, const OScalar *aa, const OScalar *x, const OScalar *y, OScalar *z)
{
  for (OInt i = 0; i < m; i++) {
    OInt    n   = ai[i + 1] - ai[i];
    OScalar sum = y[i];
    for (OInt k = 0; k < n; k++) sum += aa[ai[i] + k] * x[aj[ai[i] + k]];
    z[i] = sum;
  }
}

/* include/petsc/private/matimpl.h:1835-1890 (MatGetDiagonalMarkers): diag[i] = index of the entry
   with column i in row i; if missing, the reference points at the first entry beyond the diagonal and
   reports diagDense = false.  Returns 1 when every diagonal entry is present. */
int orc_MatGetDiagonalMarkers_SeqAIJ(OInt m, const OInt *ai, const OInt *aj, OInt *diag)
{
  int dense = 1;
  for (OInt i = 0; i < m; i++) {
    OInt k, found = 0;
    for (k = ai[i]; k < ai[i + 1]; k++) {
      if (aj[k] >= i) {
        found = (aj[k] == i);
        break;
      }
    }
    diag[i] = k;
    if (!found) dense = 0;
  }
  return dense;
}

/* aij.c:1347-1380: v_i = a[diag[i]] when the diagonal entry exists, else 0. */
void orc_MatGetDiagonal_SeqAIJ(OInt m, const OInt *ai, const OInt *aj, const OScalar *aa, OScalar *v)
{
  for (OInt i = 0; i < m; i++) {
    v[i] = 0.0;
    for (OInt k = ai[i]; k < ai[i + 1]; k++) {
      if (aj[k] == i) {
        v[i] = aa[k];
        break;
      }
    }
  }
}

/* aij.c:1797-1840 (MatInvertDiagonalForSOR_SeqAIJ) + aij.c:1842-2007 (MatSOR_SeqAIJ).
   PetscSparseDenseMinusDot(sum,r,xv,xi,nnz): sum -= xv[k]*r[xi[k]], left to right (aij.h:519-560).
   Returns 0, or 1 if a zero diagonal was met (the reference flags MAT_FACTOR_NUMERIC_ZEROPIVOT). */
int orc_MatSOR_SeqAIJ(OInt m, const OInt *ai, const OInt *aj, const OScalar *aa, const OScalar *b, OScalar omega, int flag, OScalar fshift, OInt its,
                      OInt lits, OScalar *x)
{
  OInt    *diag  = (OInt *)malloc((size_t)(m + 1) * sizeof(OInt));
  OScalar *idiag = (OScalar *)malloc((size_t)(m + 1) * sizeof(OScalar));
  OScalar *mdiag = (OScalar *)malloc((size_t)(m + 1) * sizeof(OScalar));
  OScalar *t     = (OScalar *)malloc((size_t)(m + 1) * sizeof(OScalar));
  int      zeropivot = 0;
  OInt     i, n;
  const OInt    *idx;
  const OScalar *v, *xb;
  OScalar        sum, d, scale;

  its = its * lits; /* aij.c:1855 */
  orc_MatGetDiagonalMarkers_SeqAIJ(m, ai, aj, diag);
  if (omega == 1.0 && fshift <= 0.0) { /* aij.c:1815-1827 */
    for (i = 0; i < m; i++) {
      mdiag[i] = aa[diag[i]];
      if (!fabs(mdiag[i])) zeropivot = 1;
      idiag[i] = 1.0 / aa[diag[i]];
    }
  } else { /* aij.c:1829-1833 */
    for (i = 0; i < m; i++) {
      mdiag[i] = aa[diag[i]];
      idiag[i] = omega / (fshift + aa[diag[i]]);
    }
  }

#define MINUSDOT(sum, r, xv, xi, nnz) \
  do { \
    for (OInt __k = 0; __k < (nnz); __k++) (sum) -= (xv)[__k] * (r)[(xi)[__k]]; \
  } while (0)
#define PLUSDOT(sum, r, xv, xi, nnz) \
  do { \
    for (OInt __k = 0; __k < (nnz); __k++) (sum) += (xv)[__k] * (r)[(xi)[__k]]; \
  } while (0)

  if (flag == ORC_SOR_APPLY_UPPER) { /* aij.c:1867-1884 */
    for (i = 0; i < m; i++) {
      d   = fshift + mdiag[i];
      n   = ai[i + 1] - diag[i] - 1;
      idx = aj + diag[i] + 1;
      v   = aa + diag[i] + 1;
      sum = b[i] * d / omega;
      PLUSDOT(sum, b, v, idx, n);
      x[i] = sum;
    }
    goto done;
  }
  if (flag & ORC_SOR_EISENSTAT) { /* aij.c:1887-1929 */
    scale = (2.0 / omega) - 1.0;
    for (i = m - 1; i >= 0; i--) {
      n   = ai[i + 1] - diag[i] - 1;
      idx = aj + diag[i] + 1;
      v   = aa + diag[i] + 1;
      sum = b[i];
      MINUSDOT(sum, x, v, idx, n);
      x[i] = sum * idiag[i];
    }
    for (i = 0; i < m; i++) t[i] = b[i] - scale * (aa[diag[i]]) * x[i];
    for (i = 0; i < m; i++) {
      n   = diag[i] - ai[i];
      idx = aj + ai[i];
      v   = aa + ai[i];
      sum = t[i];
      MINUSDOT(sum, t, v, idx, n);
      t[i] = sum * idiag[i];
      x[i] += t[i];
    }
    goto done;
  }
  if (flag & ORC_SOR_ZERO_INITIAL_GUESS) { /* aij.c:1930-1960 */
    if (flag & ORC_SOR_FORWARD_SWEEP || flag & ORC_SOR_LOCAL_FORWARD_SWEEP) {
      for (i = 0; i < m; i++) {
        n   = diag[i] - ai[i];
        idx = aj + ai[i];
        v   = aa + ai[i];
        sum = b[i];
        MINUSDOT(sum, x, v, idx, n);
        t[i] = sum;
        x[i] = sum * idiag[i];
      }
      xb = t;
    } else xb = b;
    if (flag & ORC_SOR_BACKWARD_SWEEP || flag & ORC_SOR_LOCAL_BACKWARD_SWEEP) {
      for (i = m - 1; i >= 0; i--) {
        n   = ai[i + 1] - diag[i] - 1;
        idx = aj + diag[i] + 1;
        v   = aa + diag[i] + 1;
        sum = xb[i];
        MINUSDOT(sum, x, v, idx, n);
        if (xb == b) x[i] = sum * idiag[i];
        else x[i] = (1 - omega) * x[i] + sum * idiag[i];
      }
    }
    its--;
  }
  while (its--) { /* aij.c:1961-2002 */
    if (flag & ORC_SOR_FORWARD_SWEEP || flag & ORC_SOR_LOCAL_FORWARD_SWEEP) {
      for (i = 0; i < m; i++) {
        n   = diag[i] - ai[i];
        idx = aj + ai[i];
        v   = aa + ai[i];
        sum = b[i];
        MINUSDOT(sum, x, v, idx, n);
        t[i] = sum;
        n    = ai[i + 1] - diag[i] - 1;
        idx  = aj + diag[i] + 1;
        v    = aa + diag[i] + 1;
        MINUSDOT(sum, x, v, idx, n);
        x[i] = (1. - omega) * x[i] + sum * idiag[i];
      }
      xb = t;
    } else xb = b;
    if (flag & ORC_SOR_BACKWARD_SWEEP || flag & ORC_SOR_LOCAL_BACKWARD_SWEEP) {
      for (i = m - 1; i >= 0; i--) {
        sum = xb[i];
        if (xb == b) {
          n   = ai[i + 1] - ai[i];
          idx = aj + ai[i];
          v   = aa + ai[i];
          MINUSDOT(sum, x, v, idx, n);
          x[i] = (1. - omega) * x[i] + (sum + mdiag[i] * x[i]) * idiag[i];
        } else {
          n   = ai[i + 1] - diag[i] - 1;
          idx = aj + diag[i] + 1;
          v   = aa + diag[i] + 1;
          MINUSDOT(sum, x, v, idx, n);
          x[i] = (1. - omega) * x[i] + sum * idiag[i];
        }
      }
    }
  }
done:
  free(diag);
  free(idiag);
  free(mdiag);
  free(t);
  return zeropivot;
}

/* ============================ inodes (Mat_SeqAIJ with identical consecutive rows) ============== */

/* MatMult_SeqAIJ_Inode (inode.c:356-560), what MatMult_SeqAIJ runs when the matrix has inodes (aij.c:1459): every row's terms are
   added in PAIRS, sum += a[k] x[j_k] + a[k+1] x[j_k+1], a last odd term alone -- the rows of a node side by side, which does not
   change any row's arithmetic: written row by row here.  Rounding-level differences from MatMult_SeqAIJ's left-to-right sums. */
void orc_MatMult_SeqAIJ_Inode(OInt m, const OInt *ai, const OInt *aj, const OScalar *aa, const OScalar *x, OScalar *y)
{
  for (OInt i = 0; i < m; i++) {
    const OInt     sz  = ai[i + 1] - ai[i];
    const OInt    *idx = aj + ai[i];
    const OScalar *v   = aa + ai[i];
    OScalar        sum = 0.;
    OInt           n;
    for (n = 0; n < sz - 1; n += 2) sum += v[n] * x[idx[n]] + v[n + 1] * x[idx[n + 1]];
    if (n == sz - 1) sum += v[n] * x[idx[n]];
    y[i] = sum;
  }
}

/* MatMultAdd_SeqAIJ_Inode (inode.c:563-760): z = y + A x with the same pairing, every row's sum starting from y_i */
void orc_MatMultAdd_SeqAIJ_Inode(OInt m, const OInt *ai, const OInt *aj, const OScalar *aa, const OScalar *x, const OScalar *y, OScalar *z)
{
  for (OInt i = 0; i < m; i++) {
    const OInt     sz  = ai[i + 1] - ai[i];
    const OInt    *idx = aj + ai[i];
    const OScalar *v   = aa + ai[i];
    OScalar        sum = y[i];
    OInt           n;
    for (n = 0; n < sz - 1; n += 2) sum += v[n] * x[idx[n]] + v[n + 1] * x[idx[n + 1]];
    if (n == sz - 1) sum += v[n] * x[idx[n]];
    z[i] = sum;
  }
}

/* MatSeqAIJCheckInode (inode.c:3920-3985): consecutive rows with the same column list form a node of at most `limit` rows
   (-mat_inode_limit, default 5).  ns[0..node_count] receives the row offsets of the nodes (size_csr).  Returns node_count, or 0
   when the reference does NOT use the inode routines: no rows, or more than 0.8 m nodes (inode.c:3962). */
OInt orc_MatSeqAIJCheckInode(OInt m, const OInt *ai, const OInt *aj, OInt limit, OInt *ns)
{
  OInt i = 0, node_count = 0;
  ns[0] = 0;
  while (i < m) {
    const OInt  nzx = ai[i + 1] - ai[i];
    const OInt *idx = aj + ai[i];
    OInt        j, blk = 1;
    for (j = i + 1; j < m && blk < limit; ++j, ++blk) {
      if (ai[j + 1] - ai[j] != nzx) break;
      if (nzx && memcmp(idx, aj + ai[j], (size_t)nzx * sizeof(OInt))) break;
    }
    ns[node_count + 1] = ns[node_count] + blk;
    node_count++;
    i = j;
  }
  if (!m || (double)node_count > .8 * (double)m) return 0;
  return node_count;
}

/* PetscKernel_A_gets_inverse_A_2 ... _5 (src/mat/impls/baij/seq/dgefa2.c:14, dgefa3.c:14, dgefa4.c, dgefa5.c:14): LINPACK dgefa
   (LU with partial pivoting, the first largest entry of the column wins) followed by dgedi (inverse(U), then inverse(U) * inverse(L)
   and the column interchanges), column-major, in place; every update is `y += t * x` with product and sum rounded separately.  Written
   once for any n <= 5 (shift = 0: what MatInvertDiagonalForSOR_SeqAIJ_Inode passes, inode.c:2425).  Returns 1 on a zero pivot (the
   reference flags MAT_FACTOR_NUMERIC_ZEROPIVOT and goes on). */
int orc_inode_invert_block(OScalar *a, int n)
{
  int     ipvt[5], zero = 0;
  OScalar work[5];
#define A_(i, j) a[(i) + (j) * n]
  if (n == 1) { /* inode.c:2457-2467 */
    if (fabs(a[0]) < 100. * 2.220446049250313e-16) zero = 1;
    a[0] = 1.0 / a[0];
    return zero;
  }
  for (int k = 0; k < n - 1; k++) {
    int     l   = k;
    OScalar max = fabs(A_(k, k));
    for (int i = k + 1; i < n; i++)
      if (fabs(A_(i, k)) > max) {
        max = fabs(A_(i, k));
        l   = i;
      }
    ipvt[k] = l;
    if (A_(l, k) == 0.0) zero = 1;
    if (l != k) {
      const OScalar t = A_(l, k);
      A_(l, k)        = A_(k, k);
      A_(k, k)        = t;
    }
    {
      const OScalar t = -1. / A_(k, k);
      for (int i = k + 1; i < n; i++) A_(i, k) *= t;
    }
    for (int j = k + 1; j < n; j++) {
      const OScalar t = A_(l, j);
      if (l != k) {
        A_(l, j) = A_(k, j);
        A_(k, j) = t;
      }
      for (int i = k + 1; i < n; i++) A_(i, j) += t * A_(i, k);
    }
  }
  ipvt[n - 1] = n - 1;
  if (A_(n - 1, n - 1) == 0.0) zero = 1;
  for (int k = 0; k < n; k++) { /* inverse(U) */
    A_(k, k) = 1.0 / A_(k, k);
    {
      const OScalar t = -A_(k, k);
      for (int i = 0; i < k; i++) A_(i, k) *= t;
    }
    for (int j = k + 1; j < n; j++) {
      const OScalar t = A_(k, j);
      A_(k, j)        = 0.0;
      for (int i = 0; i <= k; i++) A_(i, j) += t * A_(i, k);
    }
  }
  for (int k = n - 2; k >= 0; k--) { /* inverse(U) * inverse(L) */
    for (int i = k + 1; i < n; i++) {
      work[i]  = A_(i, k);
      A_(i, k) = 0.0;
    }
    for (int j = k + 1; j < n; j++) {
      const OScalar t = work[j];
      for (int i = 0; i < n; i++) A_(i, k) += t * A_(i, j);
    }
    if (ipvt[k] != k)
      for (int i = 0; i < n; i++) {
        const OScalar t  = A_(i, k);
        A_(i, k)         = A_(i, ipvt[k]);
        A_(i, ipvt[k])   = t;
      }
  }
#undef A_
  return zero;
}

/* the sums of one node over a segment of its rows' entries (inode.c:2539-2552 and every loop like it): the entries are taken in PAIRS,
   `sum_r -= v_r[k] * x[idx[k]] + v_r[k+1] * x[idx[k+1]]` (two products, their sum, one subtraction), a last odd entry on its own */
static void inode_minus(int ns, const OScalar *const *v, OInt off, const OInt *idx, OInt sz, const OScalar *src, OScalar *sum)
{
  OInt n;
  for (n = 0; n < sz - 1; n += 2) {
    const OScalar t0 = src[idx[n]], t1 = src[idx[n + 1]];
    for (int r = 0; r < ns; r++) sum[r] -= v[r][off + n] * t0 + v[r][off + n + 1] * t1;
  }
  if (n == sz - 1) {
    const OScalar t0 = src[idx[n]];
    for (int r = 0; r < ns; r++) sum[r] -= v[r][off + n] * t0;
  }
}

/* out_r = sum_c s_c * ibd[c * ns + r], c ascending, left to right (inode.c:2612-2614 forward, 2798-2800 backward: the same
   expression written from the last row up) */
static void inode_apply(int ns, const OScalar *ibd, const OScalar *s, OScalar *out)
{
  for (int r = 0; r < ns; r++) {
    OScalar acc = s[0] * ibd[r];
    for (int c = 1; c < ns; c++) acc = acc + s[c] * ibd[c * ns + r];
    out[r] = acc;
  }
}

/* MatInvertDiagonalForSOR_SeqAIJ_Inode (inode.c:2420-2492) + MatSOR_SeqAIJ_Inode (inode.c:2494-3810), what MatSOR_SeqAIJ runs
   when the matrix has inodes and omega == 1, fshift == 0 (aij.c:1852): block Gauss-Seidel with the nodes' diagonal blocks.  `lits`
   is not used (the inode routine never multiplies its by lits).  Returns 0, 1 on a zero pivot, 2 for what the reference's routine
   does not do (omega != 1, fshift != 0: aij.c:1852 sends those to the point routine -- call orc_MatSOR_SeqAIJ; SOR_APPLY_UPPER). */
int orc_MatSOR_SeqAIJ_Inode(OInt m, const OInt *ai, const OInt *aj, const OScalar *aa, OInt node_count, const OInt *ns, const OScalar *b, OScalar omega,
                            int flag, OScalar fshift, OInt its, OInt lits, OScalar *x)
{
  (void)lits;
  if (omega != 1.0 || fshift != 0.0 || (flag & ORC_SOR_APPLY_UPPER) || (flag & ORC_SOR_APPLY_LOWER)) return 2;
  OInt    *diag = (OInt *)malloc((size_t)(m + 1) * sizeof(OInt));
  OScalar *t    = (OScalar *)malloc((size_t)(m + 1) * sizeof(OScalar));
  size_t   cnt  = 0;
  int      zero = 0;
  for (OInt i = 0; i < node_count; i++) cnt += (size_t)(ns[i + 1] - ns[i]) * (size_t)(ns[i + 1] - ns[i]);
  OScalar *ibdiag = (OScalar *)malloc((cnt + 1) * sizeof(OScalar)), *bdiag = (OScalar *)malloc((cnt + 1) * sizeof(OScalar));
  size_t  *boff   = (size_t *)malloc((size_t)(node_count + 1) * sizeof(size_t));
  orc_MatGetDiagonalMarkers_SeqAIJ(m, ai, aj, diag);
  cnt = 0;
  for (OInt i = 0; i < node_count; i++) { /* inode.c:2449-2489 */
    const OInt row = ns[i], sz = ns[i + 1] - ns[i];
    boff[i] = cnt;
    for (OInt j = 0; j < sz; j++)
      for (OInt k = 0; k < sz; k++) bdiag[cnt + (size_t)(k * sz + j)] = aa[diag[row + j] - j + k];
    memcpy(ibdiag + cnt, bdiag + cnt, (size_t)(sz * sz) * sizeof(OScalar));
    zero |= orc_inode_invert_block(ibdiag + cnt, (int)sz);
    cnt += (size_t)(sz * sz);
  }
  const int fwd = (flag & ORC_SOR_FORWARD_SWEEP) || (flag & ORC_SOR_LOCAL_FORWARD_SWEEP);
  const int bwd = (flag & ORC_SOR_BACKWARD_SWEEP) || (flag & ORC_SOR_LOCAL_BACKWARD_SWEEP);
  const OScalar *v[5], *xb;
  OScalar        sum[5], out[5];
#define NODE(i) \
  const OInt row = ns[i]; \
  const int  sz  = (int)(ns[(i) + 1] - ns[i]); \
  const OInt szl = diag[row] - ai[row], len = ai[row + 1] - ai[row], szu = len - szl - sz; \
  const OInt *idx = aj + ai[row]; \
  for (int r = 0; r < sz; r++) v[r] = aa + ai[row + r]; \
  (void)szl; (void)szu; (void)idx
  if (flag & ORC_SOR_ZERO_INITIAL_GUESS) {
    if (fwd) { /* inode.c:2527-2712 */
      for (OInt i = 0; i < node_count; i++) {
        NODE(i);
        for (int r = 0; r < sz; r++) sum[r] = b[row + r];
        inode_minus(sz, v, 0, idx, szl, x, sum);
        for (int r = 0; r < sz; r++) t[row + r] = sum[r];
        inode_apply(sz, ibdiag + boff[i], sum, out);
        for (int r = 0; r < sz; r++) x[row + r] = out[r];
      }
      xb = t;
    } else xb = b;
    if (bwd) { /* inode.c:2714-2888 */
      for (OInt i = node_count - 1; i >= 0; i--) {
        NODE(i);
        for (int r = 0; r < sz; r++) sum[r] = xb[row + r];
        inode_minus(sz, v, szl + sz, idx + szl + sz, szu, x, sum);
        inode_apply(sz, ibdiag + boff[i], sum, out);
        for (int r = 0; r < sz; r++) x[row + r] = out[r];
      }
    }
    its--;
  }
  while (its-- > 0) { /* inode.c:2891-3374 */
    if (fwd) {
      for (OInt i = 0; i < node_count; i++) {
        NODE(i);
        for (int r = 0; r < sz; r++) sum[r] = b[row + r];
        inode_minus(sz, v, 0, idx, szl, x, sum);
        for (int r = 0; r < sz; r++) t[row + r] = sum[r];
        inode_minus(sz, v, szl + sz, idx + szl + sz, szu, x, sum);
        inode_apply(sz, ibdiag + boff[i], sum, out);
        for (int r = 0; r < sz; r++) x[row + r] = out[r];
      }
      xb = t;
    } else xb = b;
    if (bwd) {
      for (OInt i = node_count - 1; i >= 0; i--) {
        NODE(i);
        for (int r = 0; r < sz; r++) sum[r] = xb[row + r];
        if (xb == b) { /* the whole rows, the diagonal block with them: x += D^-1 (b - A x) (inode.c:3219-3237, 3311-3339) */
          inode_minus(sz, v, 0, idx, len, x, sum);
          inode_apply(sz, ibdiag + boff[i], sum, out);
          for (int r = 0; r < sz; r++) x[row + r] += out[r];
        } else {
          inode_minus(sz, v, szl + sz, idx + szl + sz, szu, x, sum);
          inode_apply(sz, ibdiag + boff[i], sum, out);
          for (int r = 0; r < sz; r++) x[row + r] = out[r];
        }
      }
    }
  }
  if (flag & ORC_SOR_EISENSTAT) { /* inode.c:3375-3806 */
    for (OInt i = node_count - 1; i >= 0; i--) { /* x = (U + D)^-1 b */
      NODE(i);
      for (int r = 0; r < sz; r++) sum[r] = b[row + r];
      inode_minus(sz, v, szl + sz, idx + szl + sz, szu, x, sum);
      inode_apply(sz, ibdiag + boff[i], sum, out);
      for (int r = 0; r < sz; r++) x[row + r] = out[r];
    }
    for (OInt i = 0; i < node_count; i++) { /* t = b - D x (inode.c:3559-3628) */
      const OInt row = ns[i];
      const int  sz  = (int)(ns[i + 1] - ns[i]);
      if (sz == 1) t[row] = b[row] - bdiag[boff[i]] * x[row];
      else {
        inode_apply(sz, bdiag + boff[i], x + row, out);
        for (int r = 0; r < sz; r++) t[row + r] = b[row + r] - out[r];
      }
    }
    for (OInt i = 0; i < node_count; i++) { /* t = (L + D)^-1 t, x += t (inode.c:3634-3804) */
      NODE(i);
      for (int r = 0; r < sz; r++) sum[r] = t[row + r];
      inode_minus(sz, v, 0, idx, szl, t, sum);
      inode_apply(sz, ibdiag + boff[i], sum, out);
      for (int r = 0; r < sz; r++) {
        t[row + r] = out[r];
        x[row + r] += out[r];
      }
    }
  }
#undef NODE
  free(diag);
  free(t);
  free(ibdiag);
  free(bdiag);
  free(boff);
  return zero;
}

/* MatMult / MatMultAdd on a MATSEQAIJ matrix as the reference dispatches them (aij.c:1459, 1617): the inode routines when the matrix
   has inodes (default limit 5; no_inode = -mat_no_inode).  y == NULL: MatMult. */
void orc_MatMult_SeqAIJ_dispatch(OInt m, const OInt *ai, const OInt *aj, const OScalar *aa, const OScalar *x, const OScalar *y, OScalar *z, int no_inode)
{
  int inode = 0;
  if (!no_inode && m > 0) {
    OInt *ns = (OInt *)malloc((size_t)(m + 1) * sizeof(OInt));
    inode    = orc_MatSeqAIJCheckInode(m, ai, aj, 5, ns) > 0;
    free(ns);
  }
  if (inode) {
    if (y) orc_MatMultAdd_SeqAIJ_Inode(m, ai, aj, aa, x, y, z);
    else orc_MatMult_SeqAIJ_Inode(m, ai, aj, aa, x, z);
  } else {
    if (y) orc_MatMultAdd_SeqAIJ(m, ai, aj, aa, x, y, z);
    else orc_MatMult_SeqAIJ(m, ai, aj, aa, x, z);
  }
}

/* MatSOR on a MATSEQAIJ matrix as the reference dispatches it (aij.c:1852): the inode routine when the matrix has inodes (checked
   at assembly with the default limit 5; `no_inode` = -mat_no_inode) and omega == 1, fshift == 0; the point routine otherwise */
int orc_MatSOR_SeqAIJ_dispatch(OInt m, const OInt *ai, const OInt *aj, const OScalar *aa, const OScalar *b, OScalar omega, int flag, OScalar fshift, OInt its,
                               OInt lits, OScalar *x, int no_inode)
{
  if (!no_inode && omega == 1.0 && fshift == 0.0 && m > 0) {
    OInt *ns = (OInt *)malloc((size_t)(m + 1) * sizeof(OInt));
    OInt  nc = orc_MatSeqAIJCheckInode(m, ai, aj, 5, ns);
    if (nc) {
      int r = orc_MatSOR_SeqAIJ_Inode(m, ai, aj, aa, nc, ns, b, omega, flag, fshift, its, lits, x);
      free(ns);
      return r;
    }
    free(ns);
  }
  return orc_MatSOR_SeqAIJ(m, ai, aj, aa, b, omega, flag, fshift, its, lits, x);
}

/* ============================ Mat_MPIAIJ set-up ================================================ */

static int cmp_oint(const void *a, const void *b)
{
  OInt x = *(const OInt *)a, y = *(const OInt *)b;
  return (x > y) - (x < y);
}

/* Row slab [.,.) with global columns -> diagonal block A (columns in [cstart,cend), stored local:
   col - cstart; mpiaij.c MatSetValues_MPIAIJ:560-640) and off-diagonal block B; then mmaij.c:27-65:
   collect the distinct global columns of B, SORT them (PetscSortInt, mmaij.c:51) into garray and
   rewrite B's column ids to positions in garray.  Returns ec = number of ghost columns. */
OInt orc_MatSetUpMultiply_MPIAIJ(OInt m, OInt cstart, OInt cend, const OInt *ai, const OInt *aj, const OScalar *aa, OInt *Ai, OInt *Aj, OScalar *Aa, OInt *Bi,
                                 OInt *Bj, OScalar *Ba, OInt *garray)
{
  OInt na = 0, nb = 0, ec = 0;
  for (OInt i = 0; i < m; i++) {
    Ai[i] = na;
    Bi[i] = nb;
    for (OInt k = ai[i]; k < ai[i + 1]; k++) {
      if (aj[k] >= cstart && aj[k] < cend) {
        Aj[na] = aj[k] - cstart;
        Aa[na] = aa[k];
        na++;
      } else {
        Bj[nb] = aj[k];
        Ba[nb] = aa[k];
        nb++;
      }
    }
  }
  Ai[m] = na;
  Bi[m] = nb;
  /* distinct global columns of B, sorted */
  if (nb) {
    OInt *tmp = (OInt *)malloc((size_t)nb * sizeof(OInt));
    memcpy(tmp, Bj, (size_t)nb * sizeof(OInt));
    qsort(tmp, (size_t)nb, sizeof(OInt), cmp_oint);
    for (OInt k = 0; k < nb; k++)
      if (!k || tmp[k] != tmp[k - 1]) garray[ec++] = tmp[k];
    free(tmp);
    for (OInt k = 0; k < nb; k++) { /* gid -> lid by binary search (the reference uses a hash map) */
      OInt lo = 0, hi = ec - 1, g = Bj[k];
      while (lo < hi) {
        OInt mid = (lo + hi) / 2;
        if (garray[mid] < g) lo = mid + 1;
        else hi = mid;
      }
      Bj[k] = lo;
    }
  }
  return ec;
}

/* src/mat/utils/compressedrow.c MatCheckCompressedRow: list of rows with at least one entry. */
OInt orc_MatCheckCompressedRow(OInt m, const OInt *bi, OInt *ci, OInt *ridx)
{
  OInt nrows = 0;
  ci[0]      = 0;
  for (OInt i = 0; i < m; i++) {
    if (bi[i + 1] > bi[i]) {
      ridx[nrows]   = i;
      ci[nrows + 1] = bi[i + 1];
      nrows++;
    }
  }
  return nrows;
}

/* src/sys/utils/psplit.c PetscSplitOwnership: n = N/size + ((N % size) > rank). */
void orc_PetscSplitOwnership(OInt N, int size, OInt *ranges)
{
  ranges[0] = 0;
  for (int r = 0; r < size; r++) ranges[r + 1] = ranges[r] + N / size + ((N % size) > r);
}

/* ============================ Vec_Seq ========================================================== */

/* Reduction mode of the oracle.  0 (default): the published definition of ddot, left to right in double -- what a reference
   BLAS does up to its own blocking.  1: "exactly rounded" reductions -- the dot product evaluated as if in twice the working
   precision (Dot2 of Ogita, Rump & Oishi, SIAM J. Sci. Comput. 26(6), 2005: error-free TwoProduct via fma + TwoSum), i.e. the
   correctly rounded value for every vector the tests use.  The reference's BLAS (MKL) and the GPU's fixed tree are both
   roundings of THAT number; mode 1 lets a test measure each of them against it instead of against each other. */
static int orc_exact_reductions = 0;
void orc_set_exact_reductions(int on) { orc_exact_reductions = on; }

static OScalar dot2(OInt n, const OScalar *x, const OScalar *y)
{
  OScalar p = 0.0, s = 0.0;
  for (OInt i = 0; i < n; i++) {
    const OScalar h = x[i] * y[i];
    const OScalar r = fma(x[i], y[i], -h); /* x*y = h + r exactly */
    const OScalar t = p + h;               /* TwoSum(p, h) = (t, q) */
    const OScalar z = t - p;
    const OScalar q = (p - (t - z)) + (h - z);
    p = t;
    s += q + r;
  }
  return p + s;
}

/* bvec1.c:10-49 -> BLAS ddot (third-party; see header).  Published definition, left-to-right. */
OScalar orc_VecDot_Seq(OInt n, const OScalar *x, const OScalar *y)
{
  if (orc_exact_reductions) return dot2(n, x, y);
  OScalar s = 0.0;
  for (OInt i = 0; i < n; i++) s += x[i] * y[i];
  return s;
}

void orc_VecMDot_Seq(OInt n, const OScalar *x, OInt nv, const OScalar *const *y, OScalar *z)
{
  for (OInt j = 0; j < nv; j++) z[j] = orc_VecDot_Seq(n, x, y[j]);
}

/* bvec2.c:185-235.  NORM_2 = sqrt(ddot(x,x)) (:204); NORM_1 = dasum; NORM_INFINITY with NaN
   propagation (:207-216); NORM_1_AND_2 fills z2[0], z2[1]. Returns the requested norm (z2 optional). */
OScalar orc_VecNorm_Seq(OInt n, const OScalar *x, int type, OScalar *z2)
{
  OScalar z[2] = {0.0, 0.0};
  if (n) {
    if (type == ORC_NORM_2 || type == ORC_NORM_FROBENIUS) {
      z[0] = sqrt(orc_VecDot_Seq(n, x, x));
    } else if (type == ORC_NORM_INFINITY) {
      for (OInt i = 0; i < n; i++) {
        OScalar tmp = fabs(x[i]);
        if ((tmp > z[0]) || (tmp != tmp)) {
          z[0] = tmp;
          if (tmp != tmp) break;
        }
      }
    } else if (type == ORC_NORM_1 || type == ORC_NORM_1_AND_2) {
      for (OInt i = 0; i < n; i++) z[0] += fabs(x[i]);
      if (type == ORC_NORM_1_AND_2) z[1] = sqrt(orc_VecDot_Seq(n, x, x));
    }
  }
  if (z2) {
    z2[0] = z[0];
    if (type == ORC_NORM_1_AND_2) z2[1] = z[1];
  }
  return z[0];
}

/* bvec1.c:70-89: alpha == 0 is a no-op (:75); otherwise BLAS daxpy: y_i += a*x_i. */
void orc_VecAXPY_Seq(OInt n, OScalar *y, OScalar a, const OScalar *x)
{
  if (a == 0.0) return;
  for (OInt i = 0; i < n; i++) y[i] += a * x[i];
}

/* dvec2.c:753-782 */
void orc_VecAYPX_Seq(OInt n, OScalar *y, OScalar b, const OScalar *x)
{
  if (b == 0.0) memcpy(y, x, (size_t)n * sizeof(OScalar));
  else if (b == 1.0) orc_VecAXPY_Seq(n, y, b, x);
  else if (b == -1.0)
    for (OInt i = 0; i < n; i++) y[i] = x[i] - y[i];
  else
    for (OInt i = 0; i < n; i++) y[i] = x[i] + b * y[i];
}

/* bvec1.c:91-118 */
void orc_VecAXPBY_Seq(OInt n, OScalar *y, OScalar a, OScalar b, const OScalar *x)
{
  if (a == 0.0) orc_VecScale_Seq(n, y, b);
  else if (b == 1.0) orc_VecAXPY_Seq(n, y, a, x);
  else if (a == 1.0) orc_VecAYPX_Seq(n, y, b, x);
  else if (b == 0.0)
    for (OInt i = 0; i < n; i++) y[i] = a * x[i];
  else
    for (OInt i = 0; i < n; i++) y[i] = a * x[i] + b * y[i];
}

/* dvec2.c:791-822 */
void orc_VecWAXPY_Seq(OInt n, OScalar *w, OScalar a, const OScalar *x, const OScalar *y)
{
  if (a == 1.0)
    for (OInt i = 0; i < n; i++) w[i] = y[i] + x[i];
  else if (a == -1.0)
    for (OInt i = 0; i < n; i++) w[i] = y[i] - x[i];
  else if (a == 0.0) memcpy(w, y, (size_t)n * sizeof(OScalar));
  else
    for (OInt i = 0; i < n; i++) w[i] = y[i] + a * x[i];
}

/* bvec1.c:120-147 */
void orc_VecAXPBYPCZ_Seq(OInt n, OScalar *z, OScalar a, OScalar b, OScalar c, const OScalar *x, const OScalar *y)
{
  if (a == 1.0)
    for (OInt i = 0; i < n; i++) z[i] = x[i] + b * y[i] + c * z[i];
  else if (c == 1.0)
    for (OInt i = 0; i < n; i++) z[i] = a * x[i] + b * y[i] + z[i];
  else if (c == 0.0)
    for (OInt i = 0; i < n; i++) z[i] = a * x[i] + b * y[i];
  else
    for (OInt i = 0; i < n; i++) z[i] = a * x[i] + b * y[i] + c * z[i];
}

/* dvec2.c:658-693 with the plain-loop PetscKernelAXPY{,2,3,4} (petscaxpy.h:197-235): the first nv&3
   vectors in one pass, then groups of four: y_i += a1*p1_i + a2*p2_i + a3*p3_i + a4*p4_i. */
void orc_VecMAXPY_Seq(OInt n, OScalar *y, OInt nv, const OScalar *alpha, const OScalar *const *x)
{
  OInt j_rem = nv & 0x3;
  switch (j_rem) {
  case 3:
    for (OInt i = 0; i < n; i++) y[i] += alpha[0] * x[0][i] + alpha[1] * x[1][i] + alpha[2] * x[2][i];
    break;
  case 2:
    for (OInt i = 0; i < n; i++) y[i] += alpha[0] * x[0][i] + alpha[1] * x[1][i];
    break;
  case 1:
    for (OInt i = 0; i < n; i++) y[i] += alpha[0] * x[0][i];
  default:
    break;
  }
  for (OInt j = j_rem; j < nv; j += 4) {
    const OScalar a0 = alpha[j], a1 = alpha[j + 1], a2 = alpha[j + 2], a3 = alpha[j + 3];
    const OScalar *p0 = x[j], *p1 = x[j + 1], *p2 = x[j + 2], *p3 = x[j + 3];
    for (OInt i = 0; i < n; i++) y[i] += a0 * p0[i] + a1 * p1[i] + a2 * p2[i] + a3 * p3[i];
  }
}

/* rvector.c:1394-1440 with no ops->maxpby on VECSEQ: beta == 0 -> VecSet(y,0), else VecScale(y,beta);
   then VecMAXPY. */
void orc_VecMAXPBY(OInt n, OScalar *y, OInt nv, const OScalar *alpha, OScalar beta, const OScalar *const *x)
{
  if (beta == 0.0) memset(y, 0, (size_t)n * sizeof(OScalar));
  else orc_VecScale_Seq(n, y, beta);
  orc_VecMAXPY_Seq(n, y, nv, alpha, x);
}

/* bvec2.c:72-97 */
void orc_VecPointwiseMult_Seq(OInt n, OScalar *w, const OScalar *x, const OScalar *y)
{
  for (OInt i = 0; i < n; i++) w[i] = x[i] * y[i];
}

/* bvec2.c:99-109 via ScalDiv (bvec2.c:99-102): y == 0 -> (x == 0 ? 1 : 0), else x / y. */
void orc_VecPointwiseDivide_Seq(OInt n, OScalar *w, const OScalar *x, const OScalar *y)
{
  for (OInt i = 0; i < n; i++) w[i] = (y[i] == 0.0) ? (x[i] == 0.0 ? 1.0 : 0.0) : x[i] / y[i];
}

/* vinv.c:1208-1229 (VecReciprocal_Default via VecApplyUnary_Private): x != 0 -> 1/x, 0 stays 0. */
void orc_VecReciprocal(OInt n, OScalar *x)
{
  for (OInt i = 0; i < n; i++)
    if (x[i] != 0.0) x[i] = 1.0 / x[i];
}

/* bvec2.c:167-183: alpha == 0 -> VecSet(0); alpha == 1 -> no-op; else BLAS dscal. */
void orc_VecScale_Seq(OInt n, OScalar *x, OScalar a)
{
  if (a == 0.0) memset(x, 0, (size_t)n * sizeof(OScalar));
  else if (a != 1.0)
    for (OInt i = 0; i < n; i++) x[i] *= a;
}

void orc_VecSet_Seq(OInt n, OScalar *x, OScalar a)
{
  for (OInt i = 0; i < n; i++) x[i] = a;
}

void orc_VecCopy_Seq(OInt n, const OScalar *x, OScalar *y)
{
  if (x != y) memcpy(y, x, (size_t)n * sizeof(OScalar));
}

/* ============================ PC =============================================================== */

/* jacobi.c:205-266 (PC_JACOBI_DIAGONAL, useabs = FALSE, fixdiag = TRUE, matrix not flagged SPD):
   diag <- MatGetDiagonal; VecReciprocal; zeros -> 1. */
void orc_PCSetUp_Jacobi(OInt m, const OInt *ai, const OInt *aj, const OScalar *aa, OScalar *diag)
{
  orc_MatGetDiagonal_SeqAIJ(m, ai, aj, aa, diag);
  orc_VecReciprocal(m, diag);
  for (OInt i = 0; i < m; i++)
    if (diag[i] == 0.0) diag[i] = 1.0;
}

/* ============================ KSP ============================================================== */

void orc_KSPSetDefaults(OrcKSP *ksp)
{
  /* itfunc.c KSPCreate defaults; cg.c:696; gmres.c:906-915; sor.c:442-446 */
  ksp->nranks           = 1;
  ksp->ranges           = NULL;
  ksp->pc_type          = ORC_PC_JACOBI;
  ksp->sor_flag         = ORC_SOR_LOCAL_SYMMETRIC_SWEEP;
  ksp->sor_omega        = 1.0;
  ksp->sor_shift        = 0.0;
  ksp->sor_its          = 1;
  ksp->sor_lits         = 1;
  ksp->normtype         = ORC_KSP_NORM_PRECONDITIONED;
  ksp->rtol             = 1e-5;
  ksp->abstol           = 1e-50;
  ksp->divtol           = 1e4;
  ksp->max_it           = 10000;
  ksp->min_it           = 0;
  ksp->gmres_restart    = 30;
  ksp->gmres_haptol     = 1e-30;
  ksp->gmres_cgs_refine = 0;
  ksp->guess_nonzero    = 0;
  ksp->its              = 0;
  ksp->reason           = 0;
  ksp->rnorm            = 0;
  ksp->history          = NULL;
  ksp->hist_len         = 0;
  ksp->hist_n           = 0;
  ksp->no_inode         = 0;
  ksp->mult_cb          = NULL;
  ksp->pc_cb            = NULL;
  ksp->user             = NULL;
}

typedef struct {
  OrcKSP  *ksp;
  OScalar *jdiag;   /* PCJACOBI inverse diagonal */
  OScalar  rnorm0, ttol;
  /* per-rank blocks of a simulated MPIAIJ partition (nranks > 1): diagonal block A (local columns), off-diagonal block B (columns compacted
     through garray), as MatSetUpMultiply_MPIAIJ leaves them (mmaij.c:27-65) */
  OInt    **Ai, **Aj, **Bi, **Bj, **ga;
  OScalar **Aa, **Ba;
  OInt     *ng;
  int       inode_mult; /* the operator has inodes (one rank): MatMult is MatMult_SeqAIJ_Inode (aij.c:1459).  On a simulated row partition of a
                           BLOCKED matrix the per-block inode dispatch (diagonal block with its own inodes, off-diagonal block without: mpiaij.c:824)
                           is not restated -- blocked matrices on several ranks are checked against the live MPI reference instead
                           (tests/test_gpu_plugin_mpi.py); the relaxation is dispatched per block either way (pc_apply) */
} Ctx;

/* MatMult.  One rank: MatMult_SeqAIJ (or its inode form).  nranks > 1 (round 5: restated faithfully; until then the whole sorted row was summed
   left to right, which differs from the reference by rounding): MatMult_MPIAIJ mpiaij.c:1047-1061 -- every rank multiplies its DIAGONAL block
   with its own part of x (MatMult_SeqAIJ), then ADDS the off-diagonal block's terms one by one onto that sum (MatMultAdd_SeqAIJ aij.c:1606-1658
   with lvec[k] = x[garray[k]], mmaij.c:108-117). */
static void ksp_mult(const Ctx *c, const OScalar *x, OScalar *y)
{
  const OrcKSP *ksp = c->ksp;
  if (ksp->mult_cb) {
    ksp->mult_cb(ksp->user, x, y);
    return;
  }
  if (ksp->nranks > 1 && c->Ai) {
    for (int r = 0; r < ksp->nranks; r++) {
      OInt rs = ksp->ranges[r], ml = ksp->ranges[r + 1] - rs;
      orc_MatMult_SeqAIJ(ml, c->Ai[r], c->Aj[r], c->Aa[r], x + rs, y + rs);
      if (c->ng[r] > 0) {
        OScalar *lvec = (OScalar *)malloc((size_t)c->ng[r] * sizeof(OScalar));
        for (OInt k = 0; k < c->ng[r]; k++) lvec[k] = x[c->ga[r][k]];
        orc_MatMultAdd_SeqAIJ(ml, c->Bi[r], c->Bj[r], c->Ba[r], lvec, y + rs, y + rs);
        free(lvec);
      }
    }
    return;
  }
  if (c->inode_mult) orc_MatMult_SeqAIJ_Inode(ksp->m, ksp->ai, ksp->aj, ksp->aa, x, y);
  else orc_MatMult_SeqAIJ(ksp->m, ksp->ai, ksp->aj, ksp->aa, x, y);
}

static void ctx_setup(Ctx *c, OrcKSP *ksp)
{
  c->ksp   = ksp;
  c->jdiag = NULL;
  c->Ai = c->Aj = c->Bi = c->Bj = c->ga = NULL;
  c->Aa = c->Ba = NULL;
  c->ng         = NULL;
  c->inode_mult = 0;
  if (ksp->mult_cb && ksp->pc_cb) return; /* streamed operator: nothing to set up here */
  if (!ksp->no_inode && ksp->nranks <= 1 && ksp->m > 0 && !ksp->mult_cb) {
    OInt *ns      = (OInt *)malloc((size_t)(ksp->m + 1) * sizeof(OInt));
    c->inode_mult = orc_MatSeqAIJCheckInode(ksp->m, ksp->ai, ksp->aj, 5, ns) > 0;
    free(ns);
  }
  if (ksp->pc_type == ORC_PC_JACOBI) {
    c->jdiag = (OScalar *)malloc((size_t)ksp->m * sizeof(OScalar));
    orc_PCSetUp_Jacobi(ksp->m, ksp->ai, ksp->aj, ksp->aa, c->jdiag); /* the diagonal is owned by the row's rank: same values at any nranks */
  }
  if (ksp->nranks > 1 && !ksp->mult_cb) {
    int nr = ksp->nranks;
    c->Ai  = (OInt **)calloc((size_t)nr, sizeof(OInt *));
    c->Aj  = (OInt **)calloc((size_t)nr, sizeof(OInt *));
    c->Bi  = (OInt **)calloc((size_t)nr, sizeof(OInt *));
    c->Bj  = (OInt **)calloc((size_t)nr, sizeof(OInt *));
    c->ga  = (OInt **)calloc((size_t)nr, sizeof(OInt *));
    c->Aa  = (OScalar **)calloc((size_t)nr, sizeof(OScalar *));
    c->Ba  = (OScalar **)calloc((size_t)nr, sizeof(OScalar *));
    c->ng  = (OInt *)calloc((size_t)nr, sizeof(OInt));
    for (int r = 0; r < nr; r++) {
      OInt rs = ksp->ranges[r], re = ksp->ranges[r + 1], ml = re - rs;
      OInt nz = ksp->ai[re] - ksp->ai[rs], nb = 0;
      OInt *li = (OInt *)malloc((size_t)(ml + 1) * sizeof(OInt));
      for (OInt i = 0; i <= ml; i++) li[i] = ksp->ai[rs + i] - ksp->ai[rs];
      for (OInt k = ksp->ai[rs]; k < ksp->ai[re]; k++) nb += (ksp->aj[k] < rs || ksp->aj[k] >= re);
      c->Ai[r] = (OInt *)malloc((size_t)(ml + 1) * sizeof(OInt));
      c->Aj[r] = (OInt *)malloc((size_t)(nz - nb + 1) * sizeof(OInt));
      c->Aa[r] = (OScalar *)malloc((size_t)(nz - nb + 1) * sizeof(OScalar));
      c->Bi[r] = (OInt *)malloc((size_t)(ml + 1) * sizeof(OInt));
      c->Bj[r] = (OInt *)malloc((size_t)(nb + 1) * sizeof(OInt));
      c->Ba[r] = (OScalar *)malloc((size_t)(nb + 1) * sizeof(OScalar));
      c->ga[r] = (OInt *)malloc((size_t)(nb + 1) * sizeof(OInt));
      c->ng[r] = orc_MatSetUpMultiply_MPIAIJ(ml, rs, re, li, ksp->aj + ksp->ai[rs], ksp->aa + ksp->ai[rs], c->Ai[r], c->Aj[r], c->Aa[r], c->Bi[r], c->Bj[r], c->Ba[r], c->ga[r]);
      free(li);
    }
  }
}

static void ctx_free(Ctx *c)
{
  free(c->jdiag);
  if (c->Ai) {
    for (int r = 0; r < c->ksp->nranks; r++) {
      free(c->Ai[r]);
      free(c->Aj[r]);
      free(c->Aa[r]);
      free(c->Bi[r]);
      free(c->Bj[r]);
      free(c->Ba[r]);
      free(c->ga[r]);
    }
    free(c->Ai);
    free(c->Aj);
    free(c->Aa);
    free(c->Bi);
    free(c->Bj);
    free(c->Ba);
    free(c->ga);
    free(c->ng);
  }
}

/* y = A x on a simulated partition over `nranks` ranks (MatMult_MPIAIJ, see ksp_mult): what the reference's drivers form b = A * 1 with under mpiexec. */
void orc_MatMult_MPIAIJ(OInt m, const OInt *ai, const OInt *aj, const OScalar *aa, int nranks, const OScalar *x, OScalar *y)
{
  OrcKSP k;
  Ctx    c;
  OInt  *ranges = (OInt *)malloc((size_t)(nranks + 1) * sizeof(OInt));
  orc_KSPSetDefaults(&k);
  orc_PetscSplitOwnership(m, nranks, ranges);
  k.m       = m;
  k.ai      = ai;
  k.aj      = aj;
  k.aa      = aa;
  k.nranks  = nranks;
  k.ranges  = ranges;
  k.pc_type = ORC_PC_NONE;
  k.no_inode = 1;
  ctx_setup(&c, &k);
  ksp_mult(&c, x, y);
  ctx_free(&c);
  free(ranges);
}

/* KSP_PCApply -> PCApply_Jacobi (jacobi.c:354-362) | PCApply_SOR (sor.c:27-36) -> MatSOR_SeqAIJ, or
   MatSOR_MPIAIJ (mpiaij.c:1394-1486): with SOR_ZERO_INITIAL_GUESS and its == 1 each rank does one local
   sweep on its diagonal block (mpiaij.c:1408-1412: `sor(A, bb, omega, flag, fshift, lits, 1, xx)`). */
static void pc_apply(Ctx *c, const OScalar *r, OScalar *z)
{
  OrcKSP *ksp = c->ksp;
  if (ksp->pc_cb) {
    ksp->pc_cb(ksp->user, r, z);
    return;
  }
  if (ksp->pc_type == ORC_PC_NONE) memcpy(z, r, (size_t)ksp->m * sizeof(OScalar)); /* pcnone: VecCopy */
  else if (ksp->pc_type == ORC_PC_JACOBI) orc_VecPointwiseMult_Seq(ksp->m, z, r, c->jdiag);
  else {
    int flag = ksp->sor_flag | ORC_SOR_ZERO_INITIAL_GUESS;
    if (ksp->nranks <= 1) orc_MatSOR_SeqAIJ_dispatch(ksp->m, ksp->ai, ksp->aj, ksp->aa, r, ksp->sor_omega, flag, ksp->sor_shift, ksp->sor_its, ksp->sor_lits, z, ksp->no_inode);
    else
      for (int rk = 0; rk < ksp->nranks; rk++) {
        OInt rs = ksp->ranges[rk], ml = ksp->ranges[rk + 1] - rs;
        orc_MatSOR_SeqAIJ_dispatch(ml, c->Ai[rk], c->Aj[rk], c->Aa[rk], r + rs, ksp->sor_omega, flag, ksp->sor_shift, ksp->sor_lits, 1, z + rs, ksp->no_inode);
      }
  }
}

static void log_history(OrcKSP *ksp, OScalar rnorm)
{
  if (ksp->history && ksp->hist_n < ksp->hist_len) ksp->history[ksp->hist_n] = rnorm;
  ksp->hist_n++;
}

/* iterativ.c:1490-1585 KSPConvergedDefault (zero or nonzero guess with the default context:
   initialrtol = mininitialrtol = convmaxits = FALSE). */
static int converged_default(Ctx *c, OInt n, OScalar rnorm, const OScalar *b)
{
  OrcKSP *ksp = c->ksp;
  if (ksp->normtype == ORC_KSP_NORM_NONE) return 0;
  if (!n) {
    if (ksp->guess_nonzero) {
      OScalar snorm = 0.0;
      if (ksp->normtype == ORC_KSP_NORM_UNPRECONDITIONED) snorm = orc_VecNorm_Seq(ksp->m, b, ORC_NORM_2, NULL);
      else {
        OScalar *z = (OScalar *)malloc((size_t)ksp->m * sizeof(OScalar));
        pc_apply(c, b, z);
        if (ksp->normtype == ORC_KSP_NORM_PRECONDITIONED) snorm = orc_VecNorm_Seq(ksp->m, z, ORC_NORM_2, NULL);
        else snorm = sqrt(fabs(orc_VecDot_Seq(ksp->m, b, z)));
        free(z);
      }
      if (!snorm) snorm = rnorm;
      c->rnorm0 = snorm;
    } else c->rnorm0 = rnorm;
    c->ttol = fmax(ksp->rtol * c->rnorm0, ksp->abstol);
  }
  /* `if (n <= ksp->chknorm) return` (iterativ.c:1546): chknorm = -1 by default (KSPCreate, itcreate.c:816) -- the test runs at n == 0 too (round 6: until then
     this restatement skipped it there; it differs only when the initial residual already meets the tolerance: b = 0, or a good nonzero guess) */
  if (isnan(rnorm) || isinf(rnorm)) return ORC_KSP_DIVERGED_NANORINF;
  if (n < ksp->min_it) return 0;
  if (rnorm <= c->ttol) return (rnorm < ksp->abstol) ? ORC_KSP_CONVERGED_ATOL : ORC_KSP_CONVERGED_RTOL;
  if (rnorm >= ksp->divtol * c->rnorm0) return ORC_KSP_DIVERGED_DTOL;
  return 0;
}

/* cg.c:119-352 (KSP_CG_SYMMETRIC, no trust region, no eigenvalue estimate).  W aliases Z (cg.c:145). */
int orc_KSPSolve_CG(OrcKSP *ksp, const OScalar *B, OScalar *X)
{
  Ctx      c;
  OInt     n = ksp->m, i;
  OScalar  dpi = 0.0, a = 1.0, beta = 0.0, betaold = 1.0, b = 0, dpiold, dp = 0.0;
  OScalar *R = (OScalar *)malloc((size_t)n * sizeof(OScalar));
  OScalar *Z = (OScalar *)malloc((size_t)n * sizeof(OScalar));
  OScalar *P = (OScalar *)malloc((size_t)n * sizeof(OScalar));
  OScalar *W = Z;

  ctx_setup(&c, ksp);
  ksp->its    = 0;
  ksp->reason = 0;
  ksp->hist_n = 0;
  if (!ksp->guess_nonzero) memset(X, 0, (size_t)n * sizeof(OScalar)); /* itfunc.c:908 VecSet(x,0) */
  if (ksp->guess_nonzero) {
    ksp_mult(&c, X, R); /* cg.c:154 */
    orc_VecAYPX_Seq(n, R, -1.0, B);                          /* cg.c:156 */
  } else orc_VecCopy_Seq(n, B, R);                           /* cg.c:162 */

  switch (ksp->normtype) { /* cg.c:168-190 */
  case ORC_KSP_NORM_PRECONDITIONED:
    pc_apply(&c, R, Z);
    dp = orc_VecNorm_Seq(n, Z, ORC_NORM_2, NULL);
    break;
  case ORC_KSP_NORM_UNPRECONDITIONED:
    dp = orc_VecNorm_Seq(n, R, ORC_NORM_2, NULL);
    break;
  case ORC_KSP_NORM_NATURAL:
    pc_apply(&c, R, Z);
    beta = orc_VecDot_Seq(n, Z, R);
    dp   = sqrt(fabs(beta));
    break;
  default:
    dp = 0.0;
  }
  if (isnan(dp) || isinf(dp)) {
    ksp->reason = ORC_KSP_DIVERGED_NANORINF;
    goto done;
  }
  log_history(ksp, dp);
  ksp->rnorm  = dp;
  ksp->reason = converged_default(&c, 0, dp, B); /* cg.c:205 */
  if (ksp->reason) goto done;

  if (ksp->normtype != ORC_KSP_NORM_PRECONDITIONED && ksp->normtype != ORC_KSP_NORM_NATURAL) pc_apply(&c, R, Z); /* cg.c:214 */
  if (ksp->normtype != ORC_KSP_NORM_NATURAL) beta = orc_VecDot_Seq(n, Z, R);                                      /* cg.c:216 */

  i = 0;
  do {
    ksp->its = i + 1;
    if (beta == 0.0) { /* cg.c:223 */
      ksp->reason = ORC_KSP_CONVERGED_ATOL;
      break;
    } else if ((i > 0) && (beta * betaold < 0.0)) { /* cg.c:228 */
      ksp->reason = ORC_KSP_DIVERGED_INDEFINITE_PC;
      break;
    }
    if (!i) {
      orc_VecCopy_Seq(n, Z, P); /* cg.c:236 */
      b = 0.0;
    } else {
      b = beta / betaold;
      orc_VecAYPX_Seq(n, P, b, Z); /* cg.c:249 */
    }
    dpiold = dpi;
    ksp_mult(&c, P, W); /* cg.c:257 */
    dpi     = orc_VecDot_Seq(n, P, W);                       /* cg.c:258 */
    betaold = beta;
    if (isnan(dpi) || isinf(dpi)) {
      ksp->reason = ORC_KSP_DIVERGED_NANORINF;
      break;
    }
    if ((dpi == 0.0) || ((i > 0) && (((dpi > 0) - (dpi < 0)) * ((dpiold > 0) - (dpiold < 0)) < 0.0))) { /* cg.c:262 */
      ksp->reason = ORC_KSP_DIVERGED_INDEFINITE_MAT;
      break;
    }
    a = beta / dpi;                    /* cg.c:288 */
    orc_VecAXPY_Seq(n, X, a, P);       /* cg.c:305 */
    orc_VecAXPY_Seq(n, R, -a, W);      /* cg.c:306 */
    if (ksp->normtype == ORC_KSP_NORM_PRECONDITIONED) {
      pc_apply(&c, R, Z);                                  /* cg.c:308 */
      dp = orc_VecNorm_Seq(n, Z, ORC_NORM_2, NULL);        /* cg.c:309 */
    } else if (ksp->normtype == ORC_KSP_NORM_UNPRECONDITIONED) {
      dp = orc_VecNorm_Seq(n, R, ORC_NORM_2, NULL);
    } else if (ksp->normtype == ORC_KSP_NORM_NATURAL) {
      pc_apply(&c, R, Z);
      beta = orc_VecDot_Seq(n, Z, R);
      dp   = sqrt(fabs(beta));
    } else dp = 0.0;
    if (isnan(dp) || isinf(dp)) {
      ksp->reason = ORC_KSP_DIVERGED_NANORINF;
      break;
    }
    ksp->rnorm = dp;
    log_history(ksp, dp);
    ksp->reason = converged_default(&c, i + 1, dp, B); /* cg.c:328 */
    if (ksp->reason) break;
    if (ksp->normtype != ORC_KSP_NORM_PRECONDITIONED && ksp->normtype != ORC_KSP_NORM_NATURAL) pc_apply(&c, R, Z); /* cg.c:342 */
    if (ksp->normtype != ORC_KSP_NORM_NATURAL) beta = orc_VecDot_Seq(n, Z, R);                                      /* cg.c:344 */
    i++;
  } while (i < ksp->max_it);
  if (i >= ksp->max_it) ksp->reason = ORC_KSP_DIVERGED_ITS;
done:
  ctx_free(&c);
  free(R);
  free(Z);
  free(P);
  return ksp->reason;
}

/* KSPSolve_PIPECG, src/ksp/ksp/impls/cg/pipecg/pipecg.c:20-160, statement by statement.  VecNormBegin/VecDotBegin ... End on one rank are the local
   reductions (comb.c:338-379: dot_local at Begin, the value handed back at End); on a simulated partition the sums are over the whole vectors as everywhere
   in this oracle.  Note the loop bound `i <= max_it` (pipecg.c:160) and that nothing but the convergence test ends the loop early. */
int orc_KSPSolve_PIPECG(OrcKSP *ksp, const OScalar *B, OScalar *X)
{
  Ctx      c;
  OInt     n = ksp->m, i;
  OScalar  alpha = 0.0, beta = 0.0, gamma = 0.0, gammaold = 0.0, delta = 0.0, dp = 0.0;
  OScalar *V = (OScalar *)malloc((size_t)n * 9 * sizeof(OScalar));
  OScalar *R = V, *Z = V + n, *P = V + 2 * n, *N = V + 3 * n, *W = V + 4 * n, *Q = V + 5 * n, *U = V + 6 * n, *M = V + 7 * n, *S = V + 8 * n;

  ctx_setup(&c, ksp);
  ksp->its    = 0;
  ksp->reason = 0;
  ksp->hist_n = 0;
  if (!ksp->guess_nonzero) memset(X, 0, (size_t)n * sizeof(OScalar)); /* itfunc.c:908 */
  if (ksp->guess_nonzero) {
    ksp_mult(&c, X, R);              /* pipecg.c:49 */
    orc_VecAYPX_Seq(n, R, -1.0, B);  /* pipecg.c:50 */
  } else orc_VecCopy_Seq(n, B, R);   /* pipecg.c:52 */
  pc_apply(&c, R, U);                /* pipecg.c:55 */
  switch (ksp->normtype) {           /* pipecg.c:57-86 */
  case ORC_KSP_NORM_PRECONDITIONED:
    dp = orc_VecNorm_Seq(n, U, ORC_NORM_2, NULL);
    ksp_mult(&c, U, W);
    break;
  case ORC_KSP_NORM_UNPRECONDITIONED:
    dp = orc_VecNorm_Seq(n, R, ORC_NORM_2, NULL);
    ksp_mult(&c, U, W);
    break;
  case ORC_KSP_NORM_NATURAL:
    gamma = orc_VecDot_Seq(n, R, U);
    ksp_mult(&c, U, W);
    if (isnan(gamma) || isinf(gamma)) { /* KSPCheckDot */
      ksp->reason = ORC_KSP_DIVERGED_NANORINF;
      goto done;
    }
    dp = sqrt(fabs(gamma));
    break;
  default:
    ksp_mult(&c, U, W);
    dp = 0.0;
  }
  log_history(ksp, dp);
  ksp->rnorm  = dp;
  ksp->reason = converged_default(&c, 0, dp, B); /* pipecg.c:90 */
  if (ksp->reason) goto done;

  i = 0;
  do {
    if (i > 0 && ksp->normtype == ORC_KSP_NORM_UNPRECONDITIONED) dp = orc_VecNorm_Seq(n, R, ORC_NORM_2, NULL);      /* pipecg.c:95-99 */
    else if (i > 0 && ksp->normtype == ORC_KSP_NORM_PRECONDITIONED) dp = orc_VecNorm_Seq(n, U, ORC_NORM_2, NULL);
    if (!(i == 0 && ksp->normtype == ORC_KSP_NORM_NATURAL)) gamma = orc_VecDot_Seq(n, R, U);                        /* pipecg.c:100 */
    delta = orc_VecDot_Seq(n, W, U);                                                                                 /* pipecg.c:101 */
    pc_apply(&c, W, M);  /* pipecg.c:104 */
    ksp_mult(&c, M, N);  /* pipecg.c:105 */
    if (i > 0) {
      if (ksp->normtype == ORC_KSP_NORM_NATURAL) dp = sqrt(fabs(gamma));
      else if (ksp->normtype == ORC_KSP_NORM_NONE) dp = 0.0;
      ksp->rnorm = dp;
      log_history(ksp, dp);
      ksp->reason = converged_default(&c, i, dp, B); /* pipecg.c:124 */
      if (ksp->reason) goto done;
    }
    if (i == 0) {
      alpha = gamma / delta;      /* pipecg.c:129 */
      orc_VecCopy_Seq(n, N, Z);
      orc_VecCopy_Seq(n, M, Q);
      orc_VecCopy_Seq(n, U, P);
      orc_VecCopy_Seq(n, W, S);
    } else {
      beta  = gamma / gammaold;                             /* pipecg.c:135 */
      alpha = gamma / (delta - beta / alpha * gamma);       /* pipecg.c:136 */
      orc_VecAYPX_Seq(n, Z, beta, N);
      orc_VecAYPX_Seq(n, Q, beta, M);
      orc_VecAYPX_Seq(n, P, beta, U);
      orc_VecAYPX_Seq(n, S, beta, W);
    }
    orc_VecAXPY_Seq(n, X, alpha, P);   /* pipecg.c:142-145 */
    orc_VecAXPY_Seq(n, U, -alpha, Q);
    orc_VecAXPY_Seq(n, W, -alpha, Z);
    orc_VecAXPY_Seq(n, R, -alpha, S);
    gammaold = gamma;
    i++;
    ksp->its = i;
  } while (i <= ksp->max_it);
  if (!ksp->reason) ksp->reason = ORC_KSP_DIVERGED_ITS;
done:
  ctx_free(&c);
  free(V);
  return ksp->reason;
}

/* KSPSolve_GROPPCG, src/ksp/ksp/impls/cg/groppcg/groppcg.c:23-140, statement by statement. */
int orc_KSPSolve_GROPPCG(OrcKSP *ksp, const OScalar *B, OScalar *X)
{
  Ctx      c;
  OInt     n = ksp->m, i;
  OScalar  alpha, beta = 0.0, gamma, gammaNew = 0.0, t, dp = 0.0;
  OScalar *V = (OScalar *)malloc((size_t)n * 6 * sizeof(OScalar));
  OScalar *r = V, *p = V + n, *s = V + 2 * n, *S = V + 3 * n, *z = V + 4 * n, *Z = V + 5 * n;

  ctx_setup(&c, ksp);
  ksp->its    = 0;
  ksp->reason = 0;
  ksp->hist_n = 0;
  if (!ksp->guess_nonzero) memset(X, 0, (size_t)n * sizeof(OScalar));
  if (ksp->guess_nonzero) {
    ksp_mult(&c, X, r);              /* groppcg.c:48 */
    orc_VecAYPX_Seq(n, r, -1.0, B);
  } else orc_VecCopy_Seq(n, B, r);   /* groppcg.c:51 */
  pc_apply(&c, r, z);                /* groppcg.c:54 */
  orc_VecCopy_Seq(n, z, p);          /* groppcg.c:55 */
  gamma = orc_VecDot_Seq(n, r, z);   /* groppcg.c:56-59 */
  ksp_mult(&c, p, s);                /* groppcg.c:58 */
  switch (ksp->normtype) {           /* groppcg.c:61-80 */
  case ORC_KSP_NORM_PRECONDITIONED: dp = orc_VecNorm_Seq(n, z, ORC_NORM_2, NULL); break;
  case ORC_KSP_NORM_UNPRECONDITIONED: dp = orc_VecNorm_Seq(n, r, ORC_NORM_2, NULL); break;
  case ORC_KSP_NORM_NATURAL:
    if (isnan(gamma) || isinf(gamma)) {
      ksp->reason = ORC_KSP_DIVERGED_NANORINF;
      goto done;
    }
    dp = sqrt(fabs(gamma));
    break;
  default: dp = 0.0;
  }
  log_history(ksp, dp);
  ksp->rnorm  = dp;
  ksp->reason = converged_default(&c, 0, dp, B); /* groppcg.c:84 */
  if (ksp->reason) goto done;

  i = 0;
  do {
    ksp->its = i + 1;
    i++;
    t = orc_VecDot_Seq(n, p, s);       /* groppcg.c:91 */
    pc_apply(&c, s, S);                /* groppcg.c:94 */
    alpha = gamma / t;                 /* groppcg.c:98 */
    orc_VecAXPY_Seq(n, X, alpha, p);   /* groppcg.c:99-101 */
    orc_VecAXPY_Seq(n, r, -alpha, s);
    orc_VecAXPY_Seq(n, z, -alpha, S);
    if (ksp->normtype == ORC_KSP_NORM_UNPRECONDITIONED) dp = orc_VecNorm_Seq(n, r, ORC_NORM_2, NULL);
    else if (ksp->normtype == ORC_KSP_NORM_PRECONDITIONED) dp = orc_VecNorm_Seq(n, z, ORC_NORM_2, NULL);
    gammaNew = orc_VecDot_Seq(n, r, z); /* groppcg.c:108 */
    ksp_mult(&c, z, Z);                 /* groppcg.c:111 */
    if (ksp->normtype == ORC_KSP_NORM_NATURAL) {
      if (isnan(gammaNew) || isinf(gammaNew)) {
        ksp->reason = ORC_KSP_DIVERGED_NANORINF;
        goto done;
      }
      dp = sqrt(fabs(gammaNew));
    } else if (ksp->normtype == ORC_KSP_NORM_NONE) dp = 0.0;
    ksp->rnorm = dp;
    log_history(ksp, dp);
    ksp->reason = converged_default(&c, i, dp, B); /* groppcg.c:129 */
    if (ksp->reason) goto done;
    beta  = gammaNew / gamma; /* groppcg.c:132 */
    gamma = gammaNew;
    orc_VecAYPX_Seq(n, p, beta, z); /* groppcg.c:134 */
    orc_VecAYPX_Seq(n, s, beta, Z); /* groppcg.c:135 */
  } while (i < ksp->max_it);
  if (i >= ksp->max_it) ksp->reason = ORC_KSP_DIVERGED_ITS;
done:
  ctx_free(&c);
  free(V);
  return ksp->reason;
}

/* gmres.c:88-238 (KSPGMRESCycle, KSPSolve_GMRES), :298-345 (BuildSoln), :349-395 (UpdateHessenberg),
   borthog2.c:35-113 (classical Gram-Schmidt), left preconditioning (KSP_PCApplyBAorAB: w = B A v,
   kspimpl.h), KSPInitialResidual (itres.c:35-75). */
int orc_KSPSolve_GMRES(OrcKSP *ksp, const OScalar *B, OScalar *X)
{
  Ctx       c;
  const OInt n = ksp->m, max_k = ksp->gmres_restart, N = max_k + 1;
  OScalar **VV   = (OScalar **)malloc((size_t)(max_k + 2) * sizeof(OScalar *));
  OScalar  *TEMP = (OScalar *)malloc((size_t)n * sizeof(OScalar));
  OScalar  *TMOP = (OScalar *)malloc((size_t)n * sizeof(OScalar));
  OScalar  *hh   = (OScalar *)calloc((size_t)(max_k + 2) * (max_k + 1), sizeof(OScalar));
  OScalar  *grs  = (OScalar *)calloc((size_t)(max_k + 2), sizeof(OScalar));
  OScalar  *cc   = (OScalar *)calloc((size_t)(max_k + 2), sizeof(OScalar));
  OScalar  *ss   = (OScalar *)calloc((size_t)(max_k + 2), sizeof(OScalar));
  OScalar  *nrs  = (OScalar *)calloc((size_t)(max_k + 2), sizeof(OScalar));
  OScalar  *lhh  = (OScalar *)calloc((size_t)(max_k + 2), sizeof(OScalar));
  OInt      itcount = 0;
  int       guess_nonzero = ksp->guess_nonzero;
  (void)N;
#define HH(a, b) (hh + (b) * (max_k + 2) + (a)) /* gmresimpl.h: column-major, leading dimension max_k+2 */
  for (OInt k = 0; k < max_k + 2; k++) VV[k] = (OScalar *)malloc((size_t)n * sizeof(OScalar));
  ctx_setup(&c, ksp);
  ksp->its    = 0;
  ksp->reason = 0;
  ksp->hist_n = 0;
  ksp->rnorm  = -1.0;
  if (!ksp->guess_nonzero) memset(X, 0, (size_t)n * sizeof(OScalar));

  while (!ksp->reason) {
    OScalar res, tt, hapbnd;
    OInt    it     = 0;
    int     hapend = 0;
    /* KSPInitialResidual, PC_LEFT */
    if (ksp->guess_nonzero) {
      ksp_mult(&c, X, TEMP);
      orc_VecCopy_Seq(n, B, TMOP);
      orc_VecAXPY_Seq(n, TMOP, -1.0, TEMP);
      pc_apply(&c, TMOP, VV[0]);
    } else {
      orc_VecCopy_Seq(n, B, TMOP);
      pc_apply(&c, B, VV[0]);
    }
    /* KSPGMRESCycle */
    res = orc_VecNorm_Seq(n, VV[0], ORC_NORM_2, NULL); /* VecNormalize gmres.c:98 */
    if (res != 0.0) orc_VecScale_Seq(n, VV[0], 1.0 / res);
    if (isnan(res) || isinf(res)) {
      ksp->reason = ORC_KSP_DIVERGED_NANORINF;
      break;
    }
    grs[0]     = res;
    ksp->rnorm = res;
    log_history(ksp, res);
    if (!res) {
      ksp->reason = ORC_KSP_CONVERGED_ATOL;
      break;
    }
    ksp->reason = converged_default(&c, ksp->its, res, B);
    while (!ksp->reason && it < max_k && ksp->its < ksp->max_it) {
      if (it) log_history(ksp, res);
      /* KSP_PCApplyBAorAB, left: VV[it+1] = B (A VV[it]) */
      ksp_mult(&c, VV[it], TMOP);
      pc_apply(&c, TMOP, VV[it + 1]);
      /* classical Gram-Schmidt, borthog2.c */
      {
        OScalar *h      = HH(0, it);
        int      refine = (ksp->gmres_cgs_refine == 2);
        for (OInt j = 0; j <= it; j++) h[j] = 0.0;
        orc_VecMDot_Seq(n, VV[it + 1], it + 1, (const OScalar *const *)VV, lhh);
        for (OInt j = 0; j <= it; j++) lhh[j] = -lhh[j];
        orc_VecMAXPY_Seq(n, VV[it + 1], it + 1, lhh, (const OScalar *const *)VV);
        for (OInt j = 0; j <= it; j++) h[j] -= lhh[j];
        if (ksp->gmres_cgs_refine == 1) {
          OScalar hnrm = 0.0, wnrm;
          for (OInt j = 0; j <= it; j++) hnrm += lhh[j] * lhh[j];
          hnrm = sqrt(hnrm);
          wnrm = orc_VecNorm_Seq(n, VV[it + 1], ORC_NORM_2, NULL);
          if (wnrm < hnrm) refine = 1;
        }
        if (refine) {
          orc_VecMDot_Seq(n, VV[it + 1], it + 1, (const OScalar *const *)VV, lhh);
          for (OInt j = 0; j <= it; j++) lhh[j] = -lhh[j];
          orc_VecMAXPY_Seq(n, VV[it + 1], it + 1, lhh, (const OScalar *const *)VV);
          for (OInt j = 0; j <= it; j++) h[j] -= lhh[j];
        }
      }
      tt = orc_VecNorm_Seq(n, VV[it + 1], ORC_NORM_2, NULL); /* VecNormalize gmres.c:143 */
      if (tt != 0.0) orc_VecScale_Seq(n, VV[it + 1], 1.0 / tt);
      if (isnan(tt) || isinf(tt)) {
        ksp->reason = ORC_KSP_DIVERGED_NANORINF;
        break;
      }
      *HH(it + 1, it) = tt;
      hapbnd          = fabs(tt / grs[it]);
      if (hapbnd > ksp->gmres_haptol) hapbnd = ksp->gmres_haptol;
      if (tt < hapbnd) hapend = 1;
      { /* KSPGMRESUpdateHessenberg gmres.c:349-395 */
        OScalar *h = HH(0, it), *cp = cc, *sp = ss, t;
        for (OInt j = 1; j <= it; j++) {
          t  = *h;
          *h = *cp * t + *sp * *(h + 1);
          h++;
          *h = *cp++ * *h - (*sp++ * t);
        }
        if (!hapend) {
          t = sqrt(*h * *h + *(h + 1) * *(h + 1));
          if (t == 0.0) { /* gmres.c:374-378 sets KSP_DIVERGED_NULL and returns; gmres.c:158-161 then counts the iteration and breaks */
            ksp->reason = ORC_KSP_DIVERGED_NULL;
            it++;
            ksp->its++;
            ksp->rnorm = res;
            break;
          }
          *cp         = *h / t;
          *sp         = *(h + 1) / t;
          grs[it + 1] = -(*sp * grs[it]);
          grs[it]     = *cp * grs[it];
          *h          = *cp * *h + *sp * *(h + 1);
          res         = fabs(grs[it + 1]);
        } else res = 0.0;
      }
      it++;
      ksp->its++;
      ksp->rnorm  = res;
      ksp->reason = converged_default(&c, ksp->its, res, B);
      if (hapend) {
        if (ksp->normtype == ORC_KSP_NORM_NONE) ksp->reason = ORC_KSP_CONVERGED_HAPPY_BREAKDOWN;
        else if (!ksp->reason) {
          ksp->reason = ORC_KSP_DIVERGED_BREAKDOWN;
          break;
        }
      }
    }
    /* KSPGMRESBuildSoln(GRS(0), x, x, ksp, it-1) gmres.c:298-345 */
    if (it - 1 >= 0) {
      OInt itl = it - 1;
      if (*HH(itl, itl) != 0.0) {
        nrs[itl] = grs[itl] / *HH(itl, itl);
        for (OInt ii = 1; ii <= itl; ii++) {
          OInt    k = itl - ii;
          OScalar t = grs[k];
          for (OInt j = k + 1; j <= itl; j++) t = t - *HH(k, j) * nrs[j];
          nrs[k] = t / *HH(k, k);
        }
        orc_VecMAXPBY(n, TEMP, itl + 1, nrs, 0.0, (const OScalar *const *)VV); /* gmres.c:337 */
        /* KSPUnwindPreconditioner: nothing to do for left preconditioning */
        orc_VecAXPY_Seq(n, X, 1.0, TEMP); /* gmres.c:342 */
      } else ksp->reason = ORC_KSP_DIVERGED_BREAKDOWN;
    }
    if (ksp->reason == 0 && ksp->its >= ksp->max_it) ksp->reason = ORC_KSP_DIVERGED_ITS;
    if (it && ksp->reason) log_history(ksp, res);
    itcount += it;
    if (itcount >= ksp->max_it) {
      if (!ksp->reason) ksp->reason = ORC_KSP_DIVERGED_ITS;
      break;
    }
    ksp->guess_nonzero = 1; /* gmres.c:233 */
  }
  ksp->guess_nonzero = guess_nonzero;
#undef HH
  ctx_free(&c);
  for (OInt k = 0; k < max_k + 2; k++) free(VV[k]);
  free(VV);
  free(TEMP);
  free(TMOP);
  free(hh);
  free(grs);
  free(cc);
  free(ss);
  free(nrs);
  free(lhh);
  return ksp->reason;
}
