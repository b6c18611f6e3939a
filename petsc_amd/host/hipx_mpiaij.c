/*
 * hipx_mpiaij.c -- host set-up of MATMPIAIJHIPX: the integer work of MatSetUpMultiply_MPIAIJ
 * (src/mat/impls/aij/mpi/mmaij.c:8-125) and of the diag/off-diag split done by MatSetValues_MPIAIJ
 * (src/mat/impls/aij/mpi/mpiaij.c:560-640), plus the receive side of the ghost-exchange plan.
 * Bit-exact requirement: garray (sorted distinct global ghost columns) and the compacted column ids of B
 * must equal the reference's.
 */
#include "hipx_ksp.h"
#include <stdlib.h>
#include <string.h>


static int cmp_int(const void *a, const void *b)
{
  hipx_int x = *(const hipx_int *)a, y = *(const hipx_int *)b;
  return (x > y) - (x < y);
}

void HipxMPIAIJSplitFree(HipxMPIAIJSplit *s)
{
  free(s->Ai);
  free(s->Aj);
  free(s->Aa);
  free(s->Bi);
  free(s->Bj);
  free(s->Ba);
  free(s->ridx);
  free(s->garray);
  memset(s, 0, sizeof(*s));
}

/* rows: local slab with GLOBAL column ids; [cstart,cend) = owned column range */
int HipxMatSetUpMultiply_MPIAIJ(hipx_int m, hipx_int cstart, hipx_int cend, const hipx_int *ai, const hipx_int *aj, const double *aa, HipxMPIAIJSplit *s)
{
  hipx_int na = 0, nb = 0;
  memset(s, 0, sizeof(*s));
  s->m = m;
  for (hipx_int k = 0; k < ai[m]; k++) {
    if (aj[k] >= cstart && aj[k] < cend) na++;
    else nb++;
  }
  s->Ai = (hipx_int *)malloc(sizeof(hipx_int) * ((size_t)m + 1));
  s->Aj = (hipx_int *)malloc(sizeof(hipx_int) * ((size_t)na + 1));
  s->Aa = (double *)malloc(sizeof(double) * ((size_t)na + 1));
  s->Bi = (hipx_int *)malloc(sizeof(hipx_int) * ((size_t)m + 1));
  s->Bj = (hipx_int *)malloc(sizeof(hipx_int) * ((size_t)nb + 1));
  s->Ba = (double *)malloc(sizeof(double) * ((size_t)nb + 1));
  s->ridx   = (hipx_int *)malloc(sizeof(hipx_int) * ((size_t)m + 1));
  s->garray = (hipx_int *)malloc(sizeof(hipx_int) * ((size_t)nb + 1));
  if (!s->Ai || !s->Aj || !s->Aa || !s->Bi || !s->Bj || !s->Ba || !s->ridx || !s->garray) return HIPX_ERR_MEM;
  na = nb = 0;
  hipx_int nrows_c = 0;
  s->Bi[0] = 0;
  for (hipx_int i = 0; i < m; i++) {
    hipx_int nb0 = nb;
    s->Ai[i] = na;
    for (hipx_int k = ai[i]; k < ai[i + 1]; k++) {
      if (aj[k] >= cstart && aj[k] < cend) {
        s->Aj[na] = aj[k] - cstart;
        s->Aa[na] = aa[k];
        na++;
      } else {
        s->Bj[nb] = aj[k];
        s->Ba[nb] = aa[k];
        nb++;
      }
    }
    if (nb > nb0) { /* MatCheckCompressedRow: keep only rows with entries */
      s->ridx[nrows_c]   = i;
      s->Bi[nrows_c + 1] = nb;
      nrows_c++;
    }
  }
  s->Ai[m]   = na;
  s->nrows_c = nrows_c;
  /* mmaij.c:27-65: distinct global columns -> PetscSortInt -> garray; rewrite B's columns to garray positions */
  hipx_int ec = 0;
  if (nb) {
    hipx_int *tmp = (hipx_int *)malloc(sizeof(hipx_int) * (size_t)nb);
    if (!tmp) return HIPX_ERR_MEM;
    memcpy(tmp, s->Bj, sizeof(hipx_int) * (size_t)nb);
    qsort(tmp, (size_t)nb, sizeof(hipx_int), cmp_int);
    for (hipx_int k = 0; k < nb; k++)
      if (!k || tmp[k] != tmp[k - 1]) s->garray[ec++] = tmp[k];
    free(tmp);
    for (hipx_int k = 0; k < nb; k++) {
      hipx_int lo = 0, hi = ec - 1, g = s->Bj[k];
      while (lo < hi) {
        hipx_int mid = lo + (hi - lo) / 2;
        if (s->garray[mid] < g) lo = mid + 1;
        else hi = mid;
      }
      s->Bj[k] = lo;
    }
  }
  s->nghost = ec;
  return 0;
}

/* Receive side of the ghost plan: garray is sorted, so the entries owned by one rank are contiguous.
   ranges[nranks+1] = ownership ranges (PetscLayout).  recv_ranks/recv_off sized nranks(+1) by the caller. */
int HipxHaloRecvPlan(hipx_int nghost, const hipx_int *garray, int nranks, const hipx_int *ranges, int *nrecv, int *recv_ranks, hipx_int *recv_off)
{
  int      nr = 0, owner = 0;
  recv_off[0] = 0;
  for (hipx_int k = 0; k < nghost; k++) {
    while (owner < nranks && garray[k] >= ranges[owner + 1]) owner++;
    if (owner >= nranks) return HIPX_ERR_ARG;
    if (!nr || recv_ranks[nr - 1] != owner) {
      recv_ranks[nr] = owner;
      recv_off[nr]   = k;
      nr++;
    }
    recv_off[nr] = k + 1;
  }
  *nrecv = nr;
  return 0;
}

/* PetscSplitOwnership (src/sys/utils/psplit.c): n = N/size + ((N % size) > rank) */
void HipxSplitOwnership(hipx_int N, int size, hipx_int *ranges)
{
  ranges[0] = 0;
  for (int r = 0; r < size; r++) ranges[r + 1] = ranges[r] + N / size + ((N % size) > r);
}

/* accessors for ctypes callers */
hipx_int HipxMPIAIJSplitSize(void) { return (hipx_int)sizeof(HipxMPIAIJSplit); }
