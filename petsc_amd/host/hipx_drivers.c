/*
 * hipx_drivers.c -- problem set-up for the benchmark drivers: the assembly loops of the reference's own
 * tutorials, producing the CSR that MatAssemblyEnd_SeqAIJ would leave (columns sorted within a row).
 *   HipxAssemble_ex2        src/ksp/ksp/tutorials/ex2.c:70-94          2-D 5-point Laplacian, m x n
 *   HipxAssemble_poisson7   3-D analogue: diag 6, -1 at +-1, +-n, +-n^2
 *   HipxAssemble_bench27    src/ksp/ksp/tutorials/bench_kspsolve.c:115-303  27-point stencil, h = 1/(n-1)
 * Rows [rstart, rend) with GLOBAL column ids (what each MPI rank inserts).  Call with ai == NULL to count.
 */
#include "hipx_ksp.h"

#define EMIT(J, V) \
  do { \
    if (ai) { \
      aj[nz] = (hipx_int)(J); \
      aa[nz] = (V); \
    } \
    nz++; \
  } while (0)

int64_t HipxAssemble_ex2(hipx_int m, hipx_int n, hipx_int rstart, hipx_int rend, hipx_int *ai, hipx_int *aj, double *aa)
{
  int64_t nz = 0;
  for (hipx_int Ii = rstart; Ii < rend; Ii++) {
    const hipx_int i = Ii / n, j = Ii - i * n;
    if (ai) ai[Ii - rstart] = (hipx_int)nz;
    if (i > 0) EMIT(Ii - n, -1.0);
    if (j > 0) EMIT(Ii - 1, -1.0);
    EMIT(Ii, 4.0);
    if (j < n - 1) EMIT(Ii + 1, -1.0);
    if (i < m - 1) EMIT(Ii + n, -1.0);
  }
  if (ai) ai[rend - rstart] = (hipx_int)nz;
  return nz;
}

static int64_t assemble_poisson7(hipx_int nx, hipx_int ny, hipx_int nz, hipx_int rstart, hipx_int rend, hipx_int *ai, int64_t *ai64, hipx_int *aj, double *aa);

int64_t HipxAssemble_poisson7(hipx_int n, hipx_int rstart, hipx_int rend, hipx_int *ai, hipx_int *aj, double *aa)
{
  return assemble_poisson7(n, n, n, rstart, rend, ai, NULL, aj, aa);
}

int64_t HipxAssemble_poisson7_64(hipx_int n, hipx_int rstart, hipx_int rend, int64_t *ai, hipx_int *aj, double *aa)
{
  return assemble_poisson7(n, n, n, rstart, rend, NULL, ai, aj, aa);
}

#undef EMIT
#define EMIT(J, V) \
  do { \
    if (aj) { \
      aj[nz] = (hipx_int)(J); \
      aa[nz] = (V); \
    } \
    nz++; \
  } while (0)

/* nx x ny x nz box, x fastest (the cube of SURVEY 8(d) is nx = ny = nz; config 5's weak scaling stacks z-slabs: nz grows with the
   rank count) */
int64_t HipxAssemble_poisson7_box(hipx_int nx, hipx_int ny, hipx_int nz, hipx_int rstart, hipx_int rend, hipx_int *ai, int64_t *ai64, hipx_int *aj, double *aa)
{
  return assemble_poisson7(nx, ny, nz, rstart, rend, ai, ai64, aj, aa);
}

static int64_t assemble_poisson7(hipx_int n, hipx_int ny, hipx_int nzz, hipx_int rstart, hipx_int rend, hipx_int *ai, int64_t *ai64, hipx_int *aj, double *aa)
{
  int64_t        nz = 0;
  const hipx_int n2 = n * ny;
  hipx_int       x = rstart % n, y = (rstart / n) % ny, z = rstart / n2;
  for (hipx_int Ii = rstart; Ii < rend; Ii++) {
    if (ai) ai[Ii - rstart] = (hipx_int)nz;
    if (ai64) ai64[Ii - rstart] = nz;
    if (z > 0) EMIT(Ii - n2, -1.0);
    if (y > 0) EMIT(Ii - n, -1.0);
    if (x > 0) EMIT(Ii - 1, -1.0);
    EMIT(Ii, 6.0);
    if (x < n - 1) EMIT(Ii + 1, -1.0);
    if (y < ny - 1) EMIT(Ii + n, -1.0);
    if (z < nzz - 1) EMIT(Ii + n2, -1.0);
    if (++x == n) {
      x = 0;
      if (++y == ny) {
        y = 0;
        z++;
      }
    }
  }
  if (ai) ai[rend - rstart] = (hipx_int)nz;
  if (ai64) ai64[rend - rstart] = nz;
  return nz;
}

/* 27-point stencil, both row-offset widths (ai32 or ai64 non-NULL; both NULL = count only): 512^3 has 3.6e9 nonzeros */
static int64_t assemble_bench27(hipx_int n, hipx_int rstart, hipx_int rend, hipx_int *ai, int64_t *ai64, hipx_int *aj, double *aa);

int64_t HipxAssemble_bench27(hipx_int n, hipx_int rstart, hipx_int rend, hipx_int *ai, hipx_int *aj, double *aa)
{
  return assemble_bench27(n, rstart, rend, ai, NULL, aj, aa);
}

int64_t HipxAssemble_bench27_64(hipx_int n, hipx_int rstart, hipx_int rend, int64_t *ai, hipx_int *aj, double *aa)
{
  return assemble_bench27(n, rstart, rend, NULL, ai, aj, aa);
}

#undef EMIT
#define EMIT(J, V) \
  do { \
    if (aj) { \
      aj[nz] = (hipx_int)(J); \
      aa[nz] = (V); \
    } \
    nz++; \
  } while (0)

static int64_t assemble_bench27(hipx_int n, hipx_int rstart, hipx_int rend, hipx_int *ai, int64_t *ai64, hipx_int *aj, double *aa)
{
  int64_t        nz = 0;
  const hipx_int n2 = n * n, n1 = n - 1;
  const double   h = 1.0 / (n - 1);
  const double   vcorn = -1.0 / 13 * h, vedge = -3.0 / 26 * h, vface = -3.0 / 13 * h, vcent = 44.0 / 13 * h; /* bench_kspsolve.c:122-126 */
  const double   val[4] = {vcent, vface, vedge, vcorn};
  hipx_int       x = rstart % n, y = (rstart / n) % n, z = rstart / n2;
  for (hipx_int Ii = rstart; Ii < rend; Ii++) {
    if (ai) ai[Ii - rstart] = (hipx_int)nz;
    if (ai64) ai64[Ii - rstart] = nz;
    for (int dz = -1; dz <= 1; dz++) {
      if ((dz < 0 && z == 0) || (dz > 0 && z == n1)) continue;
      for (int dy = -1; dy <= 1; dy++) {
        if ((dy < 0 && y == 0) || (dy > 0 && y == n1)) continue;
        for (int dx = -1; dx <= 1; dx++) {
          if ((dx < 0 && x == 0) || (dx > 0 && x == n1)) continue;
          EMIT(Ii + dx + dy * n + dz * n2, val[(dx != 0) + (dy != 0) + (dz != 0)]);
        }
      }
    }
    if (++x == n) {
      x = 0;
      if (++y == n) {
        y = 0;
        z++;
      }
    }
  }
  if (ai) ai[rend - rstart] = (hipx_int)nz;
  if (ai64) ai64[rend - rstart] = nz;
  return nz;
}
