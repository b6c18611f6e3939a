/*
 * hipx_ksp.c -- C host layer over libhipx.so: the reference's Krylov callers restated over device
 * vectors (see include/hipx_ksp.h for the file:line map).  Plain C11, built with gcc.
 * No CPU fallback: every numerical step is a libhipx kernel; a missing GPU fails in hipxInit().
 */
#include "hipx_ksp.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define CHK(call) \
  do { \
    int ierr_ = (call); \
    if (ierr_) return ierr_; \
  } while (0)

/* KSPConvergedReason values (include/petscksp.h) */
enum {
  KSP_CONVERGED_ITERATING        = 0,
  KSP_CONVERGED_RTOL             = 2,
  KSP_CONVERGED_ATOL             = 3,
  KSP_CONVERGED_HAPPY_BREAKDOWN  = 8,
  KSP_DIVERGED_ITS               = -3,
  KSP_DIVERGED_DTOL              = -4,
  KSP_DIVERGED_NULL              = -2,
  KSP_DIVERGED_BREAKDOWN         = -5,
  KSP_DIVERGED_INDEFINITE_PC     = -8,
  KSP_DIVERGED_NANORINF          = -9,
  KSP_DIVERGED_INDEFINITE_MAT    = -10,
  KSP_DIVERGED_PC_FAILED         = -11
};

int HipxStructSizes(int *mat, int *pc, int *ksp)
{
  *mat = (int)sizeof(HipxMat);
  *pc  = (int)sizeof(HipxPC);
  *ksp = (int)sizeof(HipxKSP);
  return 0;
}

void HipxKSPSetDefaults(HipxKSP *ksp)
{
  memset(ksp, 0, sizeof(*ksp));
  ksp->normtype         = HIPX_KSP_NORM_PRECONDITIONED;
  ksp->rtol             = 1e-5;
  ksp->abstol           = 1e-50;
  ksp->divtol           = 1e4;
  ksp->max_it           = 10000;
  ksp->pipeline         = 1;
  ksp->gmres_restart    = 30;
  ksp->gmres_haptol     = 1e-30;
  ksp->gmres_cgs_refine = 0;
}

void HipxPCSetDefaults(HipxPC *pc)
{
  memset(pc, 0, sizeof(*pc));
  pc->type      = HIPX_PC_JACOBI;
  pc->sor_flag  = 12; /* SOR_LOCAL_SYMMETRIC_SWEEP */
  pc->sor_omega = 1.0;
  pc->sor_shift = 0.0;
  pc->sor_its   = 1;
  pc->sor_lits  = 1;
}

/* MatMult_SeqAIJ (aij.c:1444) | MatMult_MPIAIJ (mpiaij.c:1047-1061) */
int HipxMatMult(HipxMat *A, const double *x, double *y)
{
  if (A->B) return hipxMatMultMPI(A->A, A->B, A->halo, x, A->lvec, y);
  return hipxMatMult(A->A, x, y);
}

/* VecTDot_Seq (bvec1.c:40) | VecTDot_MPI (pvec2.c, pvecimpl.h:105-111): local dot + SUM all-reduce */
int HipxVecDot(HipxMat *A, const double *x, const double *y, hipx_int n, double *r)
{
  if (A->nranks > 1) {
    const double *ys[1] = {y};
    return hipxVecMDotAllreduce(x, 1, ys, n, r); /* local kernel -> ncclAllReduce -> one host wait */
  }
  return hipxVecDot(x, y, n, r);
}

/* VecNorm_Seq NORM_2 = sqrt(ddot(x,x)) (bvec2.c:204) | VecNorm_MPI_Default (pvecimpl.h:150-175):
   local norm squared, SUM all-reduce, sqrt */
int HipxVecNorm2(HipxMat *A, const double *x, hipx_int n, double *r)
{
  double s;
  if (A->nranks > 1) {
    /* the reference roots the local sum, squares it again, all-reduces and roots (pvecimpl.h:150-175); summing the
       un-rooted local sums differs from that only in the last bit of each addend and saves a host round trip */
    const double *ys[1] = {x};
    CHK(hipxVecMDotAllreduce(x, 1, ys, n, &s));
  } else CHK(hipxVecDot(x, x, n, &s));
  *r = sqrt(s);
  return 0;
}

int HipxPCSetUp(HipxPC *pc, HipxMat *A)
{
  if (pc->type == HIPX_PC_NONE) { /* PCApply_None = VecCopy (none.c:6): z = r, i.e. the constant "diagonal" 1.0 of the fused kernels (r * 1.0 == r, every bit) */
    pc->dconst_valid = 1;
    pc->dconst       = 1.0;
  }
  if (pc->type == HIPX_PC_JACOBI) {
    if (!pc->dinv) CHK(hipxMalloc((void **)&pc->dinv, sizeof(double) * (size_t)(A->m ? A->m : 1)));
    CHK(hipxPCJacobiSetUp(A->A, pc->dinv)); /* MatGetDiagonal_MPIAIJ = diagonal of the diag block (mpiaij.c:1158-1167) */
    pc->dconst_valid = 0;
    if (A->m > 0) { /* constant-coefficient operators: min == max (same bits, not NaN) -> the fused update skips the dinv stream */
      double   lo, hi;
      hipx_int ilo, ihi;
      CHK(hipxVecMin(pc->dinv, A->m, &ilo, &lo));
      CHK(hipxVecMax(pc->dinv, A->m, &ihi, &hi));
      if (lo == hi && memcmp(&lo, &hi, sizeof(double)) == 0) {
        pc->dconst_valid = 1;
        pc->dconst       = lo;
      }
    }
  }
  return 0;
}

int HipxPCDestroy(HipxPC *pc)
{
  if (pc->dinv) CHK(hipxFree(pc->dinv));
  pc->dinv = NULL;
  return 0;
}

/* PCApply_None (VecCopy) | PCApply_Jacobi (jacobi.c:354-362: VecPointwiseMult(y, x, diag)) |
   PCApply_SOR (sor.c:27-36) -> MatSOR_SeqAIJ, or MatSOR_MPIAIJ (mpiaij.c:1394-1412): with zero initial guess and
   its == 1 one local sweep on the diagonal block: sor(A, bb, omega, flag, fshift, lits, 1, xx). */
int HipxPCApply(HipxPC *pc, HipxMat *A, const double *x, double *y)
{
  switch (pc->type) {
  case HIPX_PC_NONE:
    return hipxVecCopy(x, y, A->m);
  case HIPX_PC_JACOBI:
    return hipxVecPointwiseMult(y, x, pc->dinv, A->m);
  case HIPX_PC_SOR: {
    int flag = pc->sor_flag | 16; /* SOR_ZERO_INITIAL_GUESS */
    if (A->B) {
      if (pc->sor_its != 1) return HIPX_ERR_SUP; /* multi-sweep parallel SOR needs the ghost update; not on the hot path */
      return hipxMatSOR(A->A, x, pc->sor_omega, flag, pc->sor_shift, pc->sor_lits, 1, y);
    }
    return hipxMatSOR(A->A, x, pc->sor_omega, flag, pc->sor_shift, pc->sor_its, pc->sor_lits, y);
  }
  }
  return HIPX_ERR_ARG;
}

static void log_history(HipxKSP *ksp, double rnorm)
{
  if (ksp->history && ksp->hist_n < ksp->hist_len) ksp->history[ksp->hist_n] = rnorm;
  ksp->hist_n++;
}

/* iterativ.c:1490-1585 with the default context (initialrtol = mininitialrtol = convmaxits = FALSE) */
static int converged_default(HipxKSP *ksp, HipxMat *A, HipxPC *pc, hipx_int n, double rnorm, const double *b, int *reason)
{
  *reason = KSP_CONVERGED_ITERATING;
  if (ksp->normtype == HIPX_KSP_NORM_NONE || ksp->external_test) return 0;
  if (!n) {
    if (ksp->guess_nonzero) {
      double snorm = 0.0;
      if (ksp->normtype == HIPX_KSP_NORM_UNPRECONDITIONED) CHK(HipxVecNorm2(A, b, A->m, &snorm));
      else {
        double *z;
        CHK(hipxMalloc((void **)&z, sizeof(double) * (size_t)(A->m ? A->m : 1)));
        CHK(HipxPCApply(pc, A, b, z));
        if (ksp->normtype == HIPX_KSP_NORM_PRECONDITIONED) CHK(HipxVecNorm2(A, z, A->m, &snorm));
        else {
          double d;
          CHK(HipxVecDot(A, b, z, A->m, &d));
          snorm = sqrt(fabs(d));
        }
        CHK(hipxFree(z));
      }
      if (!snorm) snorm = rnorm;
      ksp->rnorm0 = snorm;
    } else ksp->rnorm0 = rnorm;
    ksp->ttol = fmax(ksp->rtol * ksp->rnorm0, ksp->abstol);
  }
  /* (iterativ.c:1546 `if (n <= ksp->chknorm) return`: chknorm = -1 by default, itcreate.c:816 -- the test runs at n == 0 too) */
  if (isnan(rnorm) || isinf(rnorm)) {
    *reason = KSP_DIVERGED_NANORINF;
    return 0;
  }
  if (n < ksp->min_it) return 0;
  if (rnorm <= ksp->ttol) *reason = (rnorm < ksp->abstol) ? KSP_CONVERGED_ATOL : KSP_CONVERGED_RTOL;
  else if (rnorm >= ksp->divtol * ksp->rnorm0) *reason = KSP_DIVERGED_DTOL;
  return 0;
}

static int ensure_cg_work(HipxKSP *ksp, hipx_int n)
{
  if (ksp->R && ksp->work_n == n) return 0;
  CHK(HipxKSPDestroyWork(ksp));
  size_t bytes = sizeof(double) * (size_t)(n ? n : 1);
  CHK(hipxMalloc((void **)&ksp->R, bytes));
  CHK(hipxMalloc((void **)&ksp->Z, bytes));
  CHK(hipxMalloc((void **)&ksp->P, bytes));
  CHK(hipxMalloc((void **)&ksp->P2, bytes));
  CHK(hipxMalloc((void **)&ksp->dscal, sizeof(double) * 16)); /* device-resident scalars of the launch-ahead loops (two-reduction form: 7, single-reduction form: 12) */
  ksp->work_n = n;
  return 0;
}

int HipxKSPDestroyWork(HipxKSP *ksp)
{
  if (ksp->R) CHK(hipxFree(ksp->R));
  if (ksp->Z) CHK(hipxFree(ksp->Z));
  if (ksp->P) CHK(hipxFree(ksp->P));
  if (ksp->P2) CHK(hipxFree(ksp->P2));
  if (ksp->dscal) CHK(hipxFree(ksp->dscal));
  if (ksp->S) CHK(hipxFree(ksp->S));
  if (ksp->W) CHK(hipxFree(ksp->W));
  ksp->S = ksp->W = NULL;
  if (ksp->gslab) CHK(hipxFree(ksp->gslab));
  ksp->gslab     = NULL;
  ksp->gslab_len = 0.0;
  if (ksp->pipe_slab) CHK(hipxFree(ksp->pipe_slab));
  ksp->pipe_slab     = NULL;
  ksp->pipe_slab_len = 0.0;
  ksp->dscal = NULL;
  ksp->R = ksp->Z = ksp->P = ksp->P2 = NULL;
  ksp->work_n = 0;
  return 0;
}

/* ---- KSPSolve_CG_SingleReduction (cg.c:364-534, KSPCGUseSingleReduction): w = A p by recurrence from s = A z, so that the iteration's sums --
   ||z||^2, delta = z . s, beta = z . r -- form ONE reduction stage (here: one 3-value reduction kernel, on several ranks one 24-byte all-reduce)
   instead of the two of the standard form.  Preconditioned norm; with PCJACOBI / PCNONE and ksp->fused the five vector updates of an
   iteration are one kernel (hipxCGSingleReductionUpdate).  Statement by statement the reference's loop: its residual history bit for bit
   when the reductions are exact (hipxSetReductionMode). */
static int sr_sums(HipxMat *A, const double *z, const double *s, const double *r, hipx_int n, double *sums3)
{
  const double *ys[3] = {z, s, r};
  if (A->nranks > 1) return hipxVecMDotAllreduce(z, 3, ys, n, sums3);
  return hipxVecMDot(z, 3, ys, n, sums3);
}

static int sr_dots(HipxMat *A, const double *z, const double *s, const double *r, hipx_int n, double *sums2)
{
  const double *ys[2] = {s, r}; /* cg.c:519-523: VecMDot(Z, 2, {S, R}) */
  if (A->nranks > 1) return hipxVecMDotAllreduce(z, 2, ys, n, sums2);
  return hipxVecMDot(z, 2, ys, n, sums2);
}

static int cg_sr_begin(HipxKSP *ksp, HipxMat *A, HipxPC *pc, const double *B, double *X)
{
  const hipx_int n = A->m;
  double        *R = ksp->R, *Z = ksp->Z, sums[3], dp = 0.0;
  const int      precond = ksp->normtype == HIPX_KSP_NORM_PRECONDITIONED, natural = ksp->normtype == HIPX_KSP_NORM_NATURAL;
  if (!ksp->S) {
    const size_t bytes = sizeof(double) * (size_t)(n ? n : 1);
    CHK(hipxMalloc((void **)&ksp->S, bytes));
    CHK(hipxMalloc((void **)&ksp->W, bytes));
  }
  if (!ksp->guess_nonzero) CHK(hipxVecSet(X, n, 0.0));
  if (ksp->guess_nonzero) {
    CHK(HipxMatMult(A, X, R));       /* cg.c:397 */
    CHK(hipxVecAYPX(R, -1.0, B, n)); /* cg.c:398 */
  } else CHK(hipxVecCopy(B, R, n));  /* cg.c:400 */
  sums[0] = sums[1] = sums[2] = 0.0;
  if (precond || natural) {          /* cg.c:403-421 */
    CHK(HipxPCApply(pc, A, R, Z));   /* cg.c:405 / 414 */
    CHK(HipxMatMult(A, Z, ksp->S));  /* cg.c:439 / 415 (preconditioned norm: moved before the norm -- one reduction for the three sums; the values are the same) */
    CHK(sr_sums(A, Z, ksp->S, R, n, sums));
    dp = precond ? sqrt(sums[0]) : sqrt(fabs(sums[2])); /* cg.c:406 / 419 */
    if (natural && (isnan(sums[2]) || isinf(sums[2]))) {  /* KSPCheckDot cg.c:418 */
      ksp->reason = KSP_DIVERGED_NANORINF;
      return 0;
    }
  } else if (ksp->normtype == HIPX_KSP_NORM_UNPRECONDITIONED) CHK(HipxVecNorm2(A, R, n, &dp)); /* cg.c:410 */
  if (isnan(dp) || isinf(dp)) {
    ksp->reason = KSP_DIVERGED_NANORINF;
    return 0;
  }
  log_history(ksp, dp);
  ksp->rnorm = dp;
  CHK(converged_default(ksp, A, pc, 0, dp, B, &ksp->reason)); /* cg.c:434 */
  if (ksp->reason) return 0;
  if (!precond && !natural) {        /* cg.c:437-443 */
    CHK(HipxPCApply(pc, A, R, Z));
    CHK(HipxMatMult(A, Z, ksp->S));
    CHK(sr_dots(A, Z, ksp->S, R, n, sums + 1));
  }
  ksp->delta = sums[1]; /* cg.c:440 */
  ksp->beta  = sums[2]; /* cg.c:441 */
  if (isnan(ksp->beta) || isinf(ksp->beta)) ksp->reason = KSP_DIVERGED_NANORINF;
  return 0;
}

static int cg_sr_step(HipxKSP *ksp, HipxMat *A, HipxPC *pc, const double *B, double *X, hipx_int nsteps)
{
  const hipx_int n = A->m;
  double        *R = ksp->R, *Z = ksp->Z, *P = ksp->P, *S = ksp->S, *W = ksp->W;
  const int      onekernel = ksp->fused && (pc->type == HIPX_PC_JACOBI || pc->type == HIPX_PC_NONE);
  const int      precond   = ksp->normtype == HIPX_KSP_NORM_PRECONDITIONED;
  for (hipx_int st = 0; st < nsteps && !ksp->reason && ksp->i < ksp->max_it; st++) {
    const hipx_int i = ksp->i;
    double         b = 0.0, dpiold, sums[3], dp;
    ksp->its = i + 1;
    if (ksp->beta == 0.0) { /* cg.c:448 */
      ksp->reason = KSP_CONVERGED_ATOL;
      break;
    } else if ((i > 0) && (ksp->beta * ksp->betaold < 0.0)) { /* cg.c:453 */
      ksp->reason = KSP_DIVERGED_INDEFINITE_PC;
      break;
    }
    dpiold = ksp->dpi;
    if (!i) {
      CHK(hipxVecCopy(Z, P, n));            /* cg.c:461 */
      CHK(HipxMatMult(A, P, W));            /* cg.c:474 */
      CHK(HipxVecDot(A, P, W, n, &ksp->dpi)); /* cg.c:475 */
    } else {
      b        = ksp->beta / ksp->betaold;  /* cg.c:464 */
      ksp->dpi = ksp->delta - ksp->beta * ksp->beta * dpiold / (ksp->betaold * ksp->betaold); /* cg.c:478 */
    }
    ksp->betaold = ksp->beta;
    if (isnan(ksp->beta) || isinf(ksp->beta)) { /* KSPCheckDot cg.c:481 */
      ksp->reason = KSP_DIVERGED_NANORINF;
      break;
    }
    if ((ksp->dpi == 0.0) || ((i > 0) && (ksp->dpi * dpiold <= 0.0))) { /* cg.c:483 */
      ksp->reason = KSP_DIVERGED_INDEFINITE_MAT;
      break;
    }
    ksp->a = ksp->beta / ksp->dpi; /* cg.c:488 */
    if (i && onekernel) CHK(hipxCGSingleReductionUpdate(P, W, X, R, Z, S, pc->type == HIPX_PC_JACOBI ? pc->dinv : NULL, b, ksp->a, n));
    else {
      if (i) {
        CHK(hipxVecAYPX(P, b, Z, n)); /* cg.c:470 */
        CHK(hipxVecAYPX(W, b, S, n)); /* cg.c:477 */
      }
      CHK(hipxVecAXPY(X, ksp->a, P, n));  /* cg.c:490 */
      CHK(hipxVecAXPY(R, -ksp->a, W, n)); /* cg.c:491 */
      CHK(HipxPCApply(pc, A, R, Z));      /* cg.c:493 | 503 | 519: z <- B r once per iteration whatever the norm type (for the unpreconditioned norm and
                                             KSP_NORM_NONE the reference applies it after the convergence test: nothing in between reads z) */
    }
    sums[0] = sums[1] = sums[2] = 0.0;
    if (precond) {
      CHK(HipxMatMult(A, Z, S));              /* cg.c:494 */
      CHK(sr_sums(A, Z, S, R, n, sums));      /* cg.c:495 + 523: VecNorm(Z) and VecMDot(Z, {S, R}) as one reduction */
      dp = sqrt(sums[0]);
    } else if (ksp->normtype == HIPX_KSP_NORM_UNPRECONDITIONED) {
      CHK(HipxVecNorm2(A, R, n, &dp));        /* cg.c:498 */
    } else if (ksp->normtype == HIPX_KSP_NORM_NATURAL) {
      CHK(HipxMatMult(A, Z, S));              /* cg.c:504 */
      CHK(sr_dots(A, Z, S, R, n, sums + 1));  /* cg.c:505 */
      if (isnan(sums[2]) || isinf(sums[2])) { /* KSPCheckDot cg.c:508 */
        ksp->reason = KSP_DIVERGED_NANORINF;
        break;
      }
      dp = sqrt(fabs(sums[2]));               /* cg.c:509 */
    } else dp = 0.0;                          /* cg.c:511 */
    if (isnan(dp) || isinf(dp)) {
      ksp->reason = KSP_DIVERGED_NANORINF;
      break;
    }
    ksp->rnorm = dp;
    log_history(ksp, dp);
    CHK(converged_default(ksp, A, pc, i + 1, dp, B, &ksp->reason)); /* cg.c:514 */
    if (ksp->reason) break;
    if (!precond && ksp->normtype != HIPX_KSP_NORM_NATURAL) { /* cg.c:518-526 */
      CHK(HipxMatMult(A, Z, S));
      CHK(sr_dots(A, Z, S, R, n, sums + 1));
    }
    ksp->delta = sums[1];
    ksp->beta  = sums[2];
    if (isnan(ksp->beta) || isinf(ksp->beta)) { /* cg.c:526 */
      ksp->reason = KSP_DIVERGED_NANORINF;
      break;
    }
    ksp->i++;
  }
  if (!ksp->reason && ksp->i >= ksp->max_it) ksp->reason = KSP_DIVERGED_ITS; /* cg.c:532 */
  return 0;
}

/* Launch-ahead form of the single-reduction loop (round 5; ksp->pipeline == 4; fused PCJACOBI / PCNONE).  Per iteration i >= 1 the device runs
     U(i): b, dpi, a from {z.s, z.r}(i-1) and {beta, dpi, a}(i-1) -- cg.c:464,478,488, on the device, in the host's expression order --, x += a_{i-1} p_{i-1} (the x
           update iteration i-1 left behind), p = z + b p, w = s + b w, r -= a w, z = r .* d                        (cg.c:470,477,490,491,493: ONE kernel)
     M(i): s = A z                                                                                                  (cg.c:494)
     R(i): {z.z, z.s, z.r} in ONE reduction (on several ranks: one 24-byte all-reduce on the stream)                (cg.c:495,523)
   and the host enqueues U(i+1), M(i+1), R(i+1) BEFORE it waits for the sums of R(i) -- its own copies of the recurrences (the same IEEE expressions) serve the
   reference's breakdown checks and the convergence test one step behind the device, as in cg_step_pipelined.  What is queued ahead when the loop stops has
   applied exactly the x update of the last completed iteration; r, z, p, w are then one iteration further (nobody reads them).  Histories and x: the
   host-synchronised loop's, bit for bit (the sums come from the same reduction kernel, the vector updates are the same operations). */
static int sr_sums_begin(HipxMat *A, const double *z, const double *s, const double *r, hipx_int n, int slot, double *dev3)
{
  const double *ys[3] = {z, s, r};
  if (A->nranks > 1) return hipxVecMDotBeginAllreduce(z, 3, ys, n, slot, dev3);
  return hipxVecMDotBegin(z, 3, ys, n, slot, dev3);
}

static int cg_sr_step_pipelined(HipxKSP *ksp, HipxMat *A, HipxPC *pc, const double *B, double *X, hipx_int nsteps)
{
  const hipx_int n = A->m;
  double        *R = ksp->R, *Z = ksp->Z, *P = ksp->P, *S = ksp->S, *W = ksp->W, *ds = ksp->dscal;
  const double  *dinv  = pc->type == HIPX_PC_JACOBI ? pc->dinv : NULL;
  int            ahead = 0; /* U, M, R of the current iteration are enqueued already */
  enum { SLOT_SR = 4 };     /* reduction slots 4, 5 by iteration parity */
#define SR_SUMS(k) (ds + 3 * (k))
#define SR_STATE(k) (ds + 6 + 3 * (k))
  for (hipx_int st = 0; st < nsteps && !ksp->reason && ksp->i < ksp->max_it; st++) {
    const hipx_int i = ksp->i;
    const int      q = (int)(i & 1);
    double         sums[3], dp, dpiold;
    ksp->its = i + 1;
    if (ksp->beta == 0.0) { /* cg.c:448 */
      ksp->reason = KSP_CONVERGED_ATOL;
      break;
    } else if ((i > 0) && (ksp->beta * ksp->betaold < 0.0)) { /* cg.c:453 */
      ksp->reason = KSP_DIVERGED_INDEFINITE_PC;
      break;
    }
    dpiold = ksp->dpi;
    if (!i) { /* the first iteration forms dpi = p . A p explicitly (cg.c:461,474-475): host-synchronised, its x update left to U(1) or to the flush */
      CHK(hipxVecCopy(Z, P, n));
      CHK(HipxMatMult(A, P, W));
      CHK(HipxVecDot(A, P, W, n, &ksp->dpi));
      ksp->betaold = ksp->beta;
      if (isnan(ksp->beta) || isinf(ksp->beta)) {
        ksp->reason = KSP_DIVERGED_NANORINF;
        break;
      }
      if (ksp->dpi == 0.0) { /* cg.c:483 */
        ksp->reason = KSP_DIVERGED_INDEFINITE_MAT;
        break;
      }
      ksp->a         = ksp->beta / ksp->dpi;
      ksp->x_pending = 1; /* cg.c:490 */
      ksp->a_pending = ksp->a;
      CHK(hipxVecAXPY(R, -ksp->a, W, n)); /* cg.c:491 */
      CHK(HipxPCApply(pc, A, R, Z));      /* cg.c:493 */
      CHK(HipxMatMult(A, Z, S));          /* cg.c:494 */
      CHK(sr_sums(A, Z, S, R, n, sums));
    } else {
      if (!ahead) { /* nothing in flight (first pass of this call): the device's scalar blocks from the host's copies */
        const double hs[3] = {0.0, ksp->delta, ksp->beta}, ht[3] = {ksp->betaold, ksp->dpi, ksp->x_pending ? ksp->a_pending : 0.0};
        CHK(hipxMemcpyHtoD(SR_SUMS(1 - q), hs, sizeof(hs)));
        CHK(hipxMemcpyHtoD(SR_STATE(1 - q), ht, sizeof(ht)));
        CHK(hipxCGSingleReductionUpdateDev(P, W, X, R, Z, S, dinv, SR_SUMS(1 - q), SR_STATE(1 - q), SR_STATE(q), n));
        CHK(HipxMatMult(A, Z, S));
        CHK(sr_sums_begin(A, Z, S, R, n, SLOT_SR + q, SR_SUMS(q)));
      }
      ahead          = 0;
      ksp->x_pending = 0; /* U(i) has applied what iteration i-1 left behind */
      /* the host's copies of the recurrences: the values U(i) formed on the device */
      ksp->dpi     = ksp->delta - ksp->beta * ksp->beta * dpiold / (ksp->betaold * ksp->betaold); /* cg.c:478 */
      ksp->betaold = ksp->beta;
      if (isnan(ksp->beta) || isinf(ksp->beta)) { /* KSPCheckDot cg.c:481 */
        ksp->reason = KSP_DIVERGED_NANORINF;
        break;
      }
      if ((ksp->dpi == 0.0) || (ksp->dpi * dpiold <= 0.0)) { /* cg.c:483 */
        ksp->reason = KSP_DIVERGED_INDEFINITE_MAT;
        break;
      }
      ksp->a         = ksp->beta / ksp->dpi; /* cg.c:488 */
      ksp->x_pending = 1;                    /* U(i) leaves x += a p to U(i+1) or to the flush */
      ksp->a_pending = ksp->a;
      if (st + 1 < nsteps && i + 1 < ksp->max_it) { /* iteration i+1 behind R(i) on the stream, before the host has seen R(i)'s sums */
        CHK(hipxCGSingleReductionUpdateDev(P, W, X, R, Z, S, dinv, SR_SUMS(q), SR_STATE(q), SR_STATE(1 - q), n));
        CHK(HipxMatMult(A, Z, S));
        CHK(sr_sums_begin(A, Z, S, R, n, SLOT_SR + (1 - q), SR_SUMS(1 - q)));
        ahead          = 1;
        ksp->x_pending = 0; /* U(i+1) applies it */
      }
      CHK(hipxRedEnd(SLOT_SR + q, 3, sums));
      if (A->nranks > 1) CHK(hipxCommCheckError());
    }
    dp = sqrt(sums[0]);
    if (isnan(dp) || isinf(dp)) {
      ksp->reason = KSP_DIVERGED_NANORINF;
      break;
    }
    ksp->rnorm = dp;
    log_history(ksp, dp);
    CHK(converged_default(ksp, A, pc, i + 1, dp, B, &ksp->reason)); /* cg.c:514 */
    if (ksp->reason) break;
    ksp->delta = sums[1];
    ksp->beta  = sums[2];
    if (isnan(ksp->beta) || isinf(ksp->beta)) { /* cg.c:526 */
      ksp->reason = KSP_DIVERGED_NANORINF;
      break;
    }
    ksp->i++;
  }
#undef SR_SUMS
#undef SR_STATE
  if (ahead) CHK(hipxStreamSynchronize()); /* stopped with iteration i+1 in flight: x is complete once U(i+1) has run */
  if (!ksp->defer_flush) CHK(HipxKSPCGFlush(ksp, A, X));
  if (!ksp->reason && ksp->i >= ksp->max_it) ksp->reason = KSP_DIVERGED_ITS; /* cg.c:532 */
  return 0;
}

/* cg.c:134-217: everything before the do-loop */
int HipxKSPCGBegin(HipxKSP *ksp, HipxMat *A, HipxPC *pc, const double *B, double *X)
{
  const hipx_int n  = A->m;
  double         dp = 0.0;
  CHK(ensure_cg_work(ksp, n));
  double *R = ksp->R, *Z = ksp->Z;
  ksp->its    = 0;
  ksp->reason = 0;
  ksp->hist_n = 0;
  ksp->i      = 0;
  ksp->dpi    = 0.0;
  ksp->a      = 1.0;
  ksp->beta   = 0.0;
  ksp->betaold = 1.0;
  ksp->x_pending = 0;
  ksp->a_pending = 0.0;
  if (ksp->single_reduction) return cg_sr_begin(ksp, A, pc, B, X);
  if (!ksp->guess_nonzero) CHK(hipxVecSet(X, n, 0.0)); /* itfunc.c:908 */
  if (ksp->guess_nonzero) {
    CHK(HipxMatMult(A, X, R));         /* cg.c:154 */
    CHK(hipxVecAYPX(R, -1.0, B, n));   /* cg.c:156 */
  } else CHK(hipxVecCopy(B, R, n));    /* cg.c:162 */

  switch (ksp->normtype) { /* cg.c:168-190 */
  case HIPX_KSP_NORM_PRECONDITIONED:
    CHK(HipxPCApply(pc, A, R, Z));
    CHK(HipxVecNorm2(A, Z, n, &dp));
    break;
  case HIPX_KSP_NORM_UNPRECONDITIONED:
    CHK(HipxVecNorm2(A, R, n, &dp));
    break;
  case HIPX_KSP_NORM_NATURAL:
    CHK(HipxPCApply(pc, A, R, Z));
    CHK(HipxVecDot(A, Z, R, n, &ksp->beta));
    dp = sqrt(fabs(ksp->beta));
    break;
  default:
    dp = 0.0;
  }
  if (isnan(dp) || isinf(dp)) { /* KSPCheckNorm, kspimpl.h:571 */
    ksp->reason = KSP_DIVERGED_NANORINF;
    return 0;
  }
  log_history(ksp, dp);
  ksp->rnorm = dp;
  CHK(converged_default(ksp, A, pc, 0, dp, B, &ksp->reason)); /* cg.c:205 */
  if (ksp->reason) return 0;
  if (ksp->normtype != HIPX_KSP_NORM_PRECONDITIONED && ksp->normtype != HIPX_KSP_NORM_NATURAL) CHK(HipxPCApply(pc, A, R, Z)); /* cg.c:214 */
  if (ksp->normtype != HIPX_KSP_NORM_NATURAL) {
    CHK(HipxVecDot(A, Z, R, n, &ksp->beta)); /* cg.c:216 */
    if (isnan(ksp->beta) || isinf(ksp->beta)) ksp->reason = KSP_DIVERGED_NANORINF;
  }
  return 0;
}

/* Launch-ahead form of the fused loop (one rank, PCJACOBI, preconditioned norm).  Per iteration i the device runs
     A(i): p = z + b p ; x += a_{i-1} p_{i-1}        (cg.c:249 + deferred cg.c:305)
     B(i): w = A p ; p.w                             (cg.c:257-258)
     C(i): r -= a w ; z = r .* d ; z.z ; z.r         (cg.c:306-309,344)
   with b and a formed ON THE DEVICE from the sums of the kernels queued before (same IEEE quotients as the host's).  The host
   enqueues C(i) right behind B(i), waits for p.w only to run the reference's breakdown checks (cg.c:262-268), then enqueues
   A(i+1), B(i+1), C(i+1) while C(i) is still running, and only then waits for the sums of C(i) to log the norm and test
   convergence (cg.c:326-328).  What has been enqueued ahead when the loop stops is harmless: A(i+1) applies exactly the x
   update of iteration i that was due anyway, B and C only touch work vectors (x is never written by C).  The arithmetic and
   its order are those of the plain loop: histories and x are bit-identical to pipeline = 0. */
/* B(i) and C(i) of the launch-ahead loop on one rank or on several (then with the all-reduce on the stream, hipx_comm.hip) */
static int mm_dot_begin(HipxMat *A, const double *p, double *w, int slot, double *dev_dot)
{
  if (A->nranks > 1) return hipxMatMultMPIDotBegin(A->A, A->B, A->halo, p, A->lvec, w, A->m, slot, dev_dot);
  return hipxMatMultDotBegin(A->A, p, w, slot, dev_dot);
}
static int fused_update_begin(HipxMat *A, double *r, double *z, const double *p, const double *w, const double *d, double dconst, const double *dev_beta, const double *dev_dpi, hipx_int n, int slot,
                              double *dev_sums2)
{
  if (A->nranks > 1) return hipxCGFusedUpdateBeginAllreduce(NULL, r, z, p, w, d, dconst, dev_beta, dev_dpi, n, slot, dev_sums2);
  return hipxCGFusedUpdateBegin(NULL, r, z, p, w, d, dconst, dev_beta, dev_dpi, n, slot, dev_sums2);
}

/* A(i) + B(i) as ONE kernel where the operator supports it (round 4: the direction update as the prologue of the march-form SpMV; constant Jacobi
   diagonal or PCNONE so that z = r * dconst is never stored and W = Z's buffer stays free; round 6: also on a rank with an off-diagonal block).  The kernel writes p_new into the
   second direction vector; on success the two swap (ksp->P is always the current direction).  *done = 0: nothing was enqueued. */
static int fused_direction_product(HipxKSP *ksp, HipxMat *A, HipxPC *pc, int dcon, double *X, double b, double a, const double *dbn, const double *dbo, const double *ddpi, int slot,
                                   double *dev_dot, int *done)
{
  *done = 0;
  if (!dcon || !ksp->P2) return 0;
  if (A->nranks > 1 || A->B) { /* round 6: a rank with an off-diagonal block runs the same two-kernel iteration (mpiaij.c:1047-1061 around the fused kernel) */
    if (!(A->nranks > 1 && A->B && A->halo) || (ksp->fused & 2)) return 0; /* (fused = 3: the round-5 sequence -- separate direction and dot kernels -- for A/B timing) */
    CHK(hipxMatMultMPICGDirectionDotBegin(A->A, A->B, A->halo, ksp->P, ksp->P2, ksp->R, pc->dconst, X, b, a, dbn, dbo, ddpi, A->lvec, ksp->Z, A->m, slot, dev_dot, done));
  } else CHK(hipxMatMultCGDirectionDotBegin(A->A, ksp->P, ksp->P2, ksp->R, pc->dconst, X, b, a, dbn, dbo, ddpi, ksp->Z, slot, dev_dot, done));
  if (*done) {
    double *t = ksp->P;
    ksp->P    = ksp->P2;
    ksp->P2   = t;
  }
  return 0;
}

static int cg_step_pipelined(HipxKSP *ksp, HipxMat *A, HipxPC *pc, const double *B, double *X, hipx_int nsteps)
{
  const hipx_int n = A->m;
  double        *R = ksp->R, *Z = ksp->Z, *W = ksp->Z;
#define P (ksp->P) /* the current direction: the fused direction + product kernel swaps the two direction vectors */
  double        *ds = ksp->dscal;
  int            ahead = 0; /* A(i), B(i), C(i) of the current iteration already enqueued */
  const int      dcon  = pc->dconst_valid && (pc->type == HIPX_PC_NONE || !getenv("HIPX_NO_DCONST")); /* constant Jacobi diagonal (or PCNONE: 1.0): multiply by the scalar */
  enum { SLOT_DOT = 1, SLOT_SUMS = 2 };
  for (hipx_int s = 0; s < nsteps && !ksp->reason && ksp->i < ksp->max_it; s++) {
    const hipx_int i  = ksp->i;
    const int      q  = (int)(i & 1);
    double        *dbeta_i = ds + 3 + 2 * (1 - q); /* z.r of iteration i-1 */
    double         sums[2], dp, dpiold;
    ksp->its = i + 1;
    if (ksp->beta == 0.0) { /* cg.c:223 */
      ksp->reason = KSP_CONVERGED_ATOL;
      break;
    } else if ((i > 0) && (ksp->beta * ksp->betaold < 0.0)) { /* cg.c:228 */
      ksp->reason = KSP_DIVERGED_INDEFINITE_PC;
      break;
    }
    if (!ahead) {
      int done = 0;
      CHK(hipxMemcpyHtoD(dbeta_i, &ksp->beta, sizeof(double)));
      if (!i) CHK(hipxVecCopy(Z, P, n)); /* cg.c:236 */
      else {
        const double b = ksp->beta / ksp->betaold;
        if (ksp->x_pending) CHK(fused_direction_product(ksp, A, pc, dcon, X, b, ksp->a_pending, NULL, NULL, NULL, SLOT_DOT, ds, &done));
        if (done) ksp->x_pending = 0;
        else if (dcon) { /* z = r * dconst is not stored in this mode */
          CHK(hipxCGAypxAxpyR(P, b, R, pc->dconst, ksp->x_pending ? X : NULL, ksp->a_pending, n));
          ksp->x_pending = 0;
        } else if (ksp->x_pending) {
          CHK(hipxCGAypxAxpy(P, b, Z, X, ksp->a_pending, n));
          ksp->x_pending = 0;
        } else CHK(hipxVecAYPX(P, b, Z, n)); /* cg.c:249 */
      }
      if (!done) CHK(mm_dot_begin(A, P, W, SLOT_DOT, ds));
      CHK(fused_update_begin(A, R, dcon ? NULL : Z, P, W, dcon ? NULL : pc->dinv, pc->dconst, dbeta_i, ds, n, SLOT_SUMS + q, ds + 2 + 2 * q));
    }
    ahead  = 0;
    dpiold = ksp->dpi;
    CHK(hipxRedEnd(SLOT_DOT, 1, &ksp->dpi));
    if (A->nranks > 1) CHK(hipxCommCheckError()); /* an all-reduce that gave up on a peer did not produce sums */
    ksp->betaold = ksp->beta;
    if (isnan(ksp->dpi) || isinf(ksp->dpi)) { /* KSPCheckDot */
      ksp->reason = KSP_DIVERGED_NANORINF;
      break;
    }
    if ((ksp->dpi == 0.0) || ((i > 0) && ((((ksp->dpi > 0) - (ksp->dpi < 0)) * ((dpiold > 0) - (dpiold < 0))) < 0.0))) { /* cg.c:262 */
      ksp->reason = KSP_DIVERGED_INDEFINITE_MAT;
      break;
    }
    ksp->a         = ksp->beta / ksp->dpi; /* cg.c:288 */
    ksp->x_pending = 1;                    /* C(i) leaves x += a p to A(i+1) or to the flush */
    ksp->a_pending = ksp->a;
    if (s + 1 < nsteps && i + 1 < ksp->max_it) { /* enqueue iteration i+1 while C(i) runs */
      double *dbeta_n = ds + 3 + 2 * q; /* z.r of iteration i, written by C(i) */
      int     done    = 0;
      /* (the fused kernel reads dpi of iteration i from ds[0] and writes dpi of iteration i + 1 there: its fold runs behind it on the stream) */
      CHK(fused_direction_product(ksp, A, pc, dcon, X, 0.0, 0.0, dbeta_n, dbeta_i, ds, SLOT_DOT, ds, &done));
      if (!done) {
        CHK(hipxCGAypxAxpyDev(P, dcon ? NULL : Z, R, pc->dconst, X, dbeta_n, dbeta_i, ds, n));
        CHK(mm_dot_begin(A, P, W, SLOT_DOT, ds));
      }
      CHK(fused_update_begin(A, R, dcon ? NULL : Z, P, W, dcon ? NULL : pc->dinv, pc->dconst, dbeta_n, ds, n, SLOT_SUMS + (1 - q), ds + 2 + 2 * (1 - q)));
      ahead          = 1;
      ksp->x_pending = 0; /* A(i+1) applies it */
    }
    CHK(hipxRedEnd(SLOT_SUMS + q, 2, sums));
    if (A->nranks > 1) CHK(hipxCommCheckError());
    dp = sqrt(sums[0]);
    if (isnan(dp) || isinf(dp)) {
      ksp->reason = KSP_DIVERGED_NANORINF;
      break;
    }
    ksp->rnorm = dp;
    log_history(ksp, dp);
    CHK(converged_default(ksp, A, pc, i + 1, dp, B, &ksp->reason));
    if (ksp->reason) break;
    ksp->beta = sums[1];
    if (isnan(ksp->beta) || isinf(ksp->beta)) {
      ksp->reason = KSP_DIVERGED_NANORINF;
      break;
    }
    ksp->i++;
  }
  if (ahead) CHK(hipxStreamSynchronize()); /* stopped with iteration i+1 in flight: x is complete once A(i+1) has run */
  if (!ksp->defer_flush) CHK(HipxKSPCGFlush(ksp, A, X));
  if (!ksp->reason && ksp->i >= ksp->max_it) ksp->reason = KSP_DIVERGED_ITS; /* cg.c:350 */
  return 0;
#undef P
}

/* nsteps passes of the loop body cg.c:220-349 (stops early when ksp->reason is set) */
int HipxKSPCGStep(HipxKSP *ksp, HipxMat *A, HipxPC *pc, const double *B, double *X, hipx_int nsteps)
{
  const hipx_int n = A->m;
  double        *R = ksp->R, *Z = ksp->Z, *P = ksp->P, *W = ksp->Z; /* W aliases Z, cg.c:145 */
  double         dp = 0.0, b, dpiold;
  /* fused update kernel (AXPY, AXPY, PCJACOBI, norm, dot): any rank count, its two sums all-reduced together;
     SpMV + dot fusion: only without an off-diagonal block (the dot needs the complete w) */
  if (ksp->single_reduction) { /* (every norm type: cg.c:403-421,492-512; round 6) */
    /* pipeline == 4: the launch-ahead form (preconditioned norm, fused PCJACOBI / PCNONE; several ranks: with the ghost exchange and the all-reduce on the streams) */
    if (ksp->pipeline == 4 && ksp->normtype == HIPX_KSP_NORM_PRECONDITIONED && ksp->fused && (pc->type == HIPX_PC_JACOBI || pc->type == HIPX_PC_NONE) && ((A->nranks <= 1 && A->m > 0) || (A->nranks > 1 && A->B && A->halo))) return cg_sr_step_pipelined(ksp, A, pc, B, X, nsteps);
    return cg_sr_step(ksp, A, pc, B, X, nsteps);
  }
  const int      fused_any = ksp->fused && (pc->type == HIPX_PC_JACOBI || (pc->type == HIPX_PC_NONE && pc->dconst_valid)) && ksp->normtype == HIPX_KSP_NORM_PRECONDITIONED;
  const int      fused_upd = fused_any && pc->type == HIPX_PC_JACOBI; /* the host-synchronised loop below streams dinv; PCNONE (round 4) takes the launch-ahead loop, whose kernels multiply by a scalar */
  const int      fused     = fused_upd && !A->B && A->nranks <= 1;
  /* launch-ahead form: one rank (SpMV + dot fused), or several ranks with a device ghost exchange (round 3: the all-reduces complete
     on the stream, see mm_dot_begin / fused_update_begin); ksp->pipeline == 2 keeps several ranks on the host-synchronised loop */
  if (ksp->pipeline && n > 0 && fused_any && ((!A->B && A->nranks <= 1) || (A->nranks > 1 && A->B && A->halo && ksp->pipeline != 2))) return cg_step_pipelined(ksp, A, pc, B, X, nsteps);
  for (hipx_int s = 0; s < nsteps && !ksp->reason && ksp->i < ksp->max_it; s++) {
    const hipx_int i = ksp->i;
    ksp->its = i + 1;
    if (ksp->beta == 0.0) { /* cg.c:223 */
      ksp->reason = KSP_CONVERGED_ATOL;
      break;
    } else if ((i > 0) && (ksp->beta * ksp->betaold < 0.0)) { /* cg.c:228 */
      ksp->reason = KSP_DIVERGED_INDEFINITE_PC;
      break;
    }
    if (!i) {
      CHK(hipxVecCopy(Z, P, n)); /* cg.c:236 */
      b = 0.0;
    } else {
      b = ksp->beta / ksp->betaold;
      if (ksp->x_pending) { /* cg.c:249 and the deferred cg.c:305 of the previous iteration in one pass over P */
        CHK(hipxCGAypxAxpy(P, b, Z, X, ksp->a_pending, n));
        ksp->x_pending = 0;
      } else CHK(hipxVecAYPX(P, b, Z, n)); /* cg.c:249 */
    }
    dpiold = ksp->dpi;
    if (fused) CHK(hipxMatMultDot(A->A, P, W, &ksp->dpi)); /* cg.c:257-258 in one pass */
    else {
      CHK(HipxMatMult(A, P, W));                 /* cg.c:257 */
      CHK(HipxVecDot(A, P, W, n, &ksp->dpi));    /* cg.c:258 */
    }
    ksp->betaold = ksp->beta;
    if (isnan(ksp->dpi) || isinf(ksp->dpi)) { /* KSPCheckDot */
      ksp->reason = KSP_DIVERGED_NANORINF;
      break;
    }
    if ((ksp->dpi == 0.0) || ((i > 0) && ((((ksp->dpi > 0) - (ksp->dpi < 0)) * ((dpiold > 0) - (dpiold < 0))) < 0.0))) { /* cg.c:262 */
      ksp->reason = KSP_DIVERGED_INDEFINITE_MAT;
      break;
    }
    ksp->a = ksp->beta / ksp->dpi; /* cg.c:288 */
    if (fused_upd) {
      /* cg.c:306-309 + cg.c:344 in one pass: r -= a w; z = r .* d; dp = ||z||; beta = z.r.  x += a p (cg.c:305) is deferred
         to the AYPX pass of the next iteration (or to the flush below): same operands, same arithmetic, p read once */
      double sums[2];
      if (A->nranks > 1) CHK(hipxCGFusedUpdateAllreduce(NULL, R, Z, P, W, pc->dinv, ksp->a, n, sums));
      else CHK(hipxCGFusedUpdate(NULL, R, Z, P, W, pc->dinv, ksp->a, n, sums));
      ksp->x_pending = 1;
      ksp->a_pending = ksp->a;
      dp = sqrt(sums[0]);
      if (isnan(dp) || isinf(dp)) {
        ksp->reason = KSP_DIVERGED_NANORINF;
        break;
      }
      ksp->rnorm = dp;
      log_history(ksp, dp);
      CHK(converged_default(ksp, A, pc, i + 1, dp, B, &ksp->reason));
      if (ksp->reason) break;
      ksp->beta = sums[1];
      if (isnan(ksp->beta) || isinf(ksp->beta)) {
        ksp->reason = KSP_DIVERGED_NANORINF;
        break;
      }
      ksp->i++;
      continue;
    }
    CHK(hipxVecAXPY(X, ksp->a, P, n));  /* cg.c:305 */
    CHK(hipxVecAXPY(R, -ksp->a, W, n)); /* cg.c:306 */
    if (ksp->normtype == HIPX_KSP_NORM_PRECONDITIONED) {
      CHK(HipxPCApply(pc, A, R, Z));     /* cg.c:308 */
      CHK(HipxVecNorm2(A, Z, n, &dp));   /* cg.c:309 */
    } else if (ksp->normtype == HIPX_KSP_NORM_UNPRECONDITIONED) {
      CHK(HipxVecNorm2(A, R, n, &dp));
    } else if (ksp->normtype == HIPX_KSP_NORM_NATURAL) {
      CHK(HipxPCApply(pc, A, R, Z));
      CHK(HipxVecDot(A, Z, R, n, &ksp->beta));
      dp = sqrt(fabs(ksp->beta));
    } else dp = 0.0;
    if (isnan(dp) || isinf(dp)) {
      ksp->reason = KSP_DIVERGED_NANORINF;
      break;
    }
    ksp->rnorm = dp;
    log_history(ksp, dp);
    CHK(converged_default(ksp, A, pc, i + 1, dp, B, &ksp->reason)); /* cg.c:328 */
    if (ksp->reason) break;
    if (ksp->normtype != HIPX_KSP_NORM_PRECONDITIONED && ksp->normtype != HIPX_KSP_NORM_NATURAL) CHK(HipxPCApply(pc, A, R, Z)); /* cg.c:342 */
    if (ksp->normtype != HIPX_KSP_NORM_NATURAL) {
      CHK(HipxVecDot(A, Z, R, n, &ksp->beta)); /* cg.c:344 */
      if (isnan(ksp->beta) || isinf(ksp->beta)) {
        ksp->reason = KSP_DIVERGED_NANORINF;
        break;
      }
    }
    ksp->i++;
  }
  if (!ksp->defer_flush) CHK(HipxKSPCGFlush(ksp, A, X)); /* X is complete whenever this function returns */
  if (!ksp->reason && ksp->i >= ksp->max_it) ksp->reason = KSP_DIVERGED_ITS; /* cg.c:350 */
  return 0;
}

int HipxKSPCGFlush(HipxKSP *ksp, HipxMat *A, double *X)
{
  if (ksp->x_pending) {
    CHK(hipxVecAXPY(X, ksp->a_pending, ksp->P, A->m)); /* cg.c:305 of the last iteration */
    ksp->x_pending = 0;
  }
  return 0;
}

int HipxKSPSolve_CG(HipxKSP *ksp, HipxMat *A, HipxPC *pc, const double *b, double *x)
{
  CHK(HipxKSPCGBegin(ksp, A, pc, b, x));
  if (ksp->reason) return 0;
  CHK(HipxKSPCGStep(ksp, A, pc, b, x, ksp->max_it));
  return 0;
}


/* ---- KSPSolve_PIPECG (pipecg.c:20-160: Ghysels & Vanroose's pipelined CG) over device vectors, launch-ahead.
   The reference's iteration i:   [dp_i, gamma_i = r.u, delta_i = w.u: ONE split-phase reduction]  ||  m = B w;  n = A m      (pipecg.c:95-113)
                                  test dp_i (i > 0);  alpha_i, beta_i;  z, q, p, s (4 VecAYPX);  x, u, w, r (4 VecAXPY)         (pipecg.c:115-150)
   Here, per iteration, on the compute stream:
     K(i)   one pass (hipxPipeCGUpdateBegin, csrc/hipx_pipe.hip): alpha_i, beta_i formed on the device from the sums of K(i-1); x += alpha_{i-1} p_{i-1} (the update
            iteration i-1 left behind -- so what is enqueued ahead when the loop stops is exactly the update that was due); the eight vector updates of
            iteration i; m = B w (PCJACOBI / PCNONE) for the product; the three sums of iteration i + 1.  9 vector reads + 9 writes.
     [several ranks: the sums' all-reduce STARTS here (hipxPipeCGUpdateBeginAllreduce) ...]
     S(i+1) n = A m (MatMult_SeqAIJ | MatMult_MPIAIJ with its ghost exchange)
     [... and ENDS here (hipxAllreduceEnd): one 24-byte all-reduce per iteration, hidden behind the product -- PetscCommSplitReductionBegin ... End of pipecg.c:103-113]
   The host enqueues K(i), S(i+1) BEFORE it waits for the sums K(i-1) produced (ksp->pipeline, default on), then logs dp_i and runs the convergence test of
   iteration i: no host round trip sits between the kernels.  Elementwise the operations and their order are the reference's (the x update is deferred, not
   reordered); the history differs from the reference's by the rounding of the reductions only (bit-identical to reference + exact BLAS in the exact reduction
   mode, tests/test_gpu_ksp.py).  Loop bound, iteration count and reason as pipecg.c:158-160 (`i <= max_it`, then KSP_DIVERGED_ITS). */
static int ensure_pipe_work(HipxKSP *ksp, hipx_int n)
{
  const double need = 9.0 * (double)(((size_t)n + 1) & ~(size_t)1) + 2.0;
  if (!ksp->pipe_slab || ksp->pipe_slab_len < need) {
    if (ksp->pipe_slab) CHK(hipxFree(ksp->pipe_slab));
    ksp->pipe_slab     = NULL;
    ksp->pipe_slab_len = 0.0;
    CHK(hipxMalloc((void **)&ksp->pipe_slab, sizeof(double) * (size_t)need));
    ksp->pipe_slab_len = need;
  }
  if (!ksp->dscal) CHK(hipxMalloc((void **)&ksp->dscal, sizeof(double) * 16));
  return 0;
}

int HipxKSPSolve_PIPECG(HipxKSP *ksp, HipxMat *A, HipxPC *pc, const double *B, double *X)
{
  const hipx_int n    = A->m;
  const size_t   npad = ((size_t)n + 1) & ~(size_t)1; /* (16-byte aligned pieces) */
  const int      nrm  = ksp->normtype == HIPX_KSP_NORM_PRECONDITIONED ? 1 : (ksp->normtype == HIPX_KSP_NORM_UNPRECONDITIONED ? 2 : 0);
  enum { SLOT_P = 6 }; /* reduction slots 6, 7 by iteration parity */
  hipxPipeCGVecs v;
  double        *R, *U, *W, *Z, *Q, *P, *S, *M, *N, *ds;
  const double  *dinv;
  double         dp = 0.0, gamma = 0.0, delta = 0.0, gammaold = 0.0, alpha = 0.0, sums[3];
  int            ahead = 0, xpend = 0; /* K(i) of the current iteration enqueued already; x += alpha p of the last K not applied yet */
  hipx_int       i;

  if (pc->type != HIPX_PC_NONE && pc->type != HIPX_PC_JACOBI) return HIPX_ERR_SUP; /* (m = B w is formed inside the update kernel: pointwise preconditioners) */
  if (n <= 0 && A->nranks <= 1) return HIPX_ERR_ARG;                               /* (an empty system; a rank without rows among several takes part in the collectives) */
  CHK(ensure_pipe_work(ksp, n));
  R = ksp->pipe_slab; U = R + npad; W = U + npad; Z = W + npad; Q = Z + npad; P = Q + npad; S = P + npad; M = S + npad; N = M + npad;
  ds = ksp->dscal; /* [0..9): the sums of three consecutive iterations, [9], [10]: alpha by iteration parity */
#define PSUMS(k) (ds + 3 * (int)((k) % 3))
#define PALPHA(k) (ds + 9 + (int)((k) & 1))
  v.z = Z; v.q = Q; v.p = P; v.s = S; v.x = X; v.u = U; v.w = W; v.r = R; v.m = M; v.n = N;
  dinv = (pc->type == HIPX_PC_JACOBI && !(pc->dconst_valid && !getenv("HIPX_NO_DCONST"))) ? pc->dinv : NULL;
  if (!dinv && pc->dconst == 1.0) M = W; /* PCNONE (or a diagonal of ones): m = B w is w itself, bit for bit -- the update kernel does not store it, the product reads w */
  ksp->its    = 0;
  ksp->reason = 0;
  ksp->hist_n = 0;
  if (!ksp->guess_nonzero) CHK(hipxVecSet(X, n, 0.0)); /* itfunc.c:908 */
  if (ksp->guess_nonzero) {
    CHK(HipxMatMult(A, X, R));       /* pipecg.c:49 */
    CHK(hipxVecAYPX(R, -1.0, B, n)); /* pipecg.c:50 */
  } else CHK(hipxVecCopy(B, R, n));  /* pipecg.c:52 */
  CHK(HipxPCApply(pc, A, R, U));     /* pipecg.c:55 */
  switch (ksp->normtype) {           /* pipecg.c:57-86 */
  case HIPX_KSP_NORM_PRECONDITIONED:
    CHK(HipxVecNorm2(A, U, n, &dp));
    break;
  case HIPX_KSP_NORM_UNPRECONDITIONED:
    CHK(HipxVecNorm2(A, R, n, &dp));
    break;
  case HIPX_KSP_NORM_NATURAL:
    CHK(HipxVecDot(A, R, U, n, &gamma));
    if (isnan(gamma) || isinf(gamma)) { /* KSPCheckDot */
      ksp->reason = KSP_DIVERGED_NANORINF;
      return 0;
    }
    dp = sqrt(fabs(gamma));
    break;
  default:
    dp = 0.0;
  }
  CHK(HipxMatMult(A, U, W)); /* w <- A u */
  log_history(ksp, dp);
  ksp->rnorm = dp;
  CHK(converged_default(ksp, A, pc, 0, dp, B, &ksp->reason)); /* pipecg.c:90 */
  if (ksp->reason) return 0;
  { /* the sums of iteration 0 (pipecg.c:100-101; u.u rides along unused): device copy for K(0), host copy for the final x update's alpha */
    const double *ys[3] = {U, R, W};
    if (A->nranks > 1) CHK(hipxVecMDotBeginAllreduce(U, 3, ys, n, SLOT_P, PSUMS(0)));
    else CHK(hipxVecMDotBegin(U, 3, ys, n, SLOT_P, PSUMS(0)));
    CHK(hipxRedEnd(SLOT_P, 3, sums));
    if (A->nranks > 1) CHK(hipxCommCheckError());
    if (ksp->normtype != HIPX_KSP_NORM_NATURAL) gamma = sums[1];
    else CHK(hipxMemcpyHtoD(PSUMS(0) + 1, &gamma, sizeof(double))); /* (natural norm: iteration 0 keeps the gamma its norm was formed from, pipecg.c:100) */
    delta = sums[2];
  }
  if (M != W) CHK(HipxPCApply(pc, A, W, M)); /* pipecg.c:104 */
  CHK(HipxMatMult(A, M, N));                 /* pipecg.c:105 */
  i = 0;
  do {
    /* top of iteration i: gamma, delta (and dp for i > 0) are the host's copies of sums_i */
    if (i > 0) {
      /* launch-ahead: K(i), S(i+1) behind K(i-1), S(i) on the stream before the host looks at sums_i */
      if (ksp->pipeline && !ahead) {
        if (A->nranks > 1) CHK(hipxPipeCGUpdateBeginAllreduce(&v, dinv, pc->dconst, nrm, 0, PSUMS(i), PSUMS(i - 1), PALPHA(i - 1), PALPHA(i), n, SLOT_P + (int)((i + 1) & 1)));
        else CHK(hipxPipeCGUpdateBegin(&v, dinv, pc->dconst, nrm, 0, PSUMS(i), PSUMS(i - 1), PALPHA(i - 1), PALPHA(i), n, SLOT_P + (int)((i + 1) & 1), PSUMS(i + 1)));
        CHK(HipxMatMult(A, M, N));
        if (A->nranks > 1) CHK(hipxAllreduceEnd(SLOT_P + (int)((i + 1) & 1), 3, PSUMS(i + 1)));
        ahead = 1;
      }
      CHK(hipxRedEnd(SLOT_P + (int)(i & 1), 3, sums)); /* sums_i, produced by K(i-1) */
      if (A->nranks > 1) CHK(hipxCommCheckError());
      gammaold = gamma;
      gamma    = sums[1];
      delta    = sums[2];
      if (ksp->normtype == HIPX_KSP_NORM_NATURAL) dp = sqrt(fabs(gamma));
      else if (ksp->normtype == HIPX_KSP_NORM_NONE) dp = 0.0;
      else dp = sqrt(sums[0]);
      ksp->rnorm = dp;
      log_history(ksp, dp);
      CHK(converged_default(ksp, A, pc, i, dp, B, &ksp->reason)); /* pipecg.c:124 */
      if (ksp->reason) break;
    }
    /* the host's copies of the scalars K(i) forms on the device (the same IEEE operations) */
    if (i == 0) alpha = gamma / delta; /* pipecg.c:129 */
    else {
      const double beta = gamma / gammaold;           /* pipecg.c:135 */
      alpha = gamma / (delta - beta / alpha * gamma); /* pipecg.c:136 */
    }
    if (!ahead) {
      const int first = (i == 0);
      if (A->nranks > 1) CHK(hipxPipeCGUpdateBeginAllreduce(&v, dinv, pc->dconst, nrm, first, PSUMS(i), first ? NULL : PSUMS(i - 1), first ? NULL : PALPHA(i - 1), PALPHA(i), n, SLOT_P + (int)((i + 1) & 1)));
      else CHK(hipxPipeCGUpdateBegin(&v, dinv, pc->dconst, nrm, first, PSUMS(i), first ? NULL : PSUMS(i - 1), first ? NULL : PALPHA(i - 1), PALPHA(i), n, SLOT_P + (int)((i + 1) & 1), PSUMS(i + 1)));
      CHK(HipxMatMult(A, M, N));
      if (A->nranks > 1) CHK(hipxAllreduceEnd(SLOT_P + (int)((i + 1) & 1), 3, PSUMS(i + 1)));
    }
    ahead = 0;
    xpend = 1; /* K(i) has applied the update of iteration i-1; x += alpha_i p_i waits for K(i+1) or for the flush below */
    i++;
    ksp->its = i;
  } while (i <= ksp->max_it);
  /* stopped by the test with K(i) enqueued ahead: it applied the update that was due (alpha_{i-1} p_{i-1}) and xpend refers to that one -- nothing is pending;
     stopped by the test without it, or by the loop bound: the last K left its own update behind */
  if (ksp->reason && ahead) xpend = 0;
  CHK(hipxStreamSynchronize());
  if (xpend && n > 0) CHK(hipxVecAXPY(X, alpha, P, n)); /* pipecg.c:142 of the last iteration that ran (its p is in P: no K has overwritten it) */
  if (!ksp->reason) ksp->reason = KSP_DIVERGED_ITS;
  return 0;
#undef PSUMS
#undef PALPHA
}

/* ---- KSPSolve_GROPPCG (groppcg.c:23-140: Gropp's asynchronous CG) over device vectors, launch-ahead (round 6).
   Per iteration i on the compute stream (csrc/hipx_pipe.hip):
     D(i)  i > 1: x += alpha_{i-1} p ; p = z + beta p ; s = Z + beta s ; t_i = p.s   (groppcg.c:132-136 of iteration i-1 + the x update it left behind + groppcg.c:87;
           i = 1: t_1 = p.s alone);  several ranks: t all-reduced on the stream
     U(i)  alpha_i = gamma_{i-1} / t_i ; r -= alpha s ; z -= alpha (B s) ; dp, gammaNew_i = r.z   (groppcg.c:96-108; S = B s re-formed per element: PCJACOBI / PCNONE);
           several ranks: the two sums START their all-reduce here ...
     S(i)  Z = A z                                                                     (groppcg.c:111)
           ... and it ENDS here (hipxAllreduceEnd): reduction 2 hidden behind the product, as PetscCommSplitReductionBegin ... VecDotEnd hide it in the reference.
   alpha and beta are formed on the device; the host enqueues iteration i + 1 before it waits for iteration i's sums (ksp->pipeline), logs dp_i and runs the
   convergence test one step behind.  The x update of iteration i is applied by D(i + 1) or by the flush at the end (alpha_i = gamma_{i-1} / t_i re-formed on the
   host: the same IEEE quotient), so what is queued ahead when the loop stops has applied exactly the updates of completed iterations.  13 vector passes + the
   product per iteration against the reference loop's 20 (one kernel per call, S stored).  Elementwise bit-identical to the reference; history equal to
   reference + exact BLAS in the exact reduction mode (tests/test_gpu_pipecg.py). */
int HipxKSPSolve_GROPPCG(HipxKSP *ksp, HipxMat *A, HipxPC *pc, const double *B, double *X)
{
  const hipx_int n    = A->m;
  const size_t   npad = ((size_t)n + 1) & ~(size_t)1;
  const int      nrm  = ksp->normtype == HIPX_KSP_NORM_PRECONDITIONED ? 1 : (ksp->normtype == HIPX_KSP_NORM_UNPRECONDITIONED ? 2 : 0);
  enum { SLOT_T = 8, SLOT_U = 10 }; /* reduction slots by iteration parity: t_i in 8 / 9, {dp, gammaNew_i} in 10 / 11 */
  double        *r, *p, *s, *z, *Z, *ds;
  const double  *dinv;
  double         dp = 0.0, gamma = 0.0, gammaNew = 0.0, gamma_im1 = 0.0, sums[2], t = 0.0;
  int            ahead = 0, xpend = 0;
  hipx_int       i;

  if (pc->type != HIPX_PC_NONE && pc->type != HIPX_PC_JACOBI) return HIPX_ERR_SUP;
  if (n <= 0 && A->nranks <= 1) return HIPX_ERR_ARG;
  CHK(ensure_pipe_work(ksp, n));
  r = ksp->pipe_slab; p = r + npad; s = p + npad; z = s + npad; Z = z + npad;
  ds = ksp->dscal; /* [0..6): {dp, gamma} of three consecutive iterations; [6], [7]: t by parity; [8], [9]: alpha by parity */
#define GSUMS(k) (ds + 2 * (int)((k) % 3))
#define GT(k) (ds + 6 + (int)((k) & 1))
#define GALPHA(k) (ds + 8 + (int)((k) & 1))
  dinv = (pc->type == HIPX_PC_JACOBI && !(pc->dconst_valid && !getenv("HIPX_NO_DCONST"))) ? pc->dinv : NULL;
  ksp->its    = 0;
  ksp->reason = 0;
  ksp->hist_n = 0;
  if (!ksp->guess_nonzero) CHK(hipxVecSet(X, n, 0.0));
  if (ksp->guess_nonzero) {
    CHK(HipxMatMult(A, X, r));       /* groppcg.c:48 */
    CHK(hipxVecAYPX(r, -1.0, B, n)); /* groppcg.c:49 */
  } else CHK(hipxVecCopy(B, r, n));  /* groppcg.c:51 */
  CHK(HipxPCApply(pc, A, r, z));     /* groppcg.c:54 */
  CHK(hipxVecCopy(z, p, n));         /* groppcg.c:55 */
  CHK(HipxVecDot(A, r, z, n, &gamma)); /* groppcg.c:56-59 */
  CHK(HipxMatMult(A, p, s));           /* groppcg.c:58 */
  switch (ksp->normtype) {             /* groppcg.c:61-80 */
  case HIPX_KSP_NORM_PRECONDITIONED:
    CHK(HipxVecNorm2(A, z, n, &dp));
    break;
  case HIPX_KSP_NORM_UNPRECONDITIONED:
    CHK(HipxVecNorm2(A, r, n, &dp));
    break;
  case HIPX_KSP_NORM_NATURAL:
    if (isnan(gamma) || isinf(gamma)) { /* KSPCheckDot */
      ksp->reason = KSP_DIVERGED_NANORINF;
      return 0;
    }
    dp = sqrt(fabs(gamma));
    break;
  default:
    dp = 0.0;
  }
  log_history(ksp, dp);
  ksp->rnorm = dp;
  CHK(converged_default(ksp, A, pc, 0, dp, B, &ksp->reason)); /* groppcg.c:84 */
  if (ksp->reason) return 0;
  CHK(hipxMemcpyHtoD(GSUMS(0) + 1, &gamma, sizeof(double))); /* gamma_0 for U(1) and D(2) */
  i = 0;
  do {
    ksp->its = i + 1;
    i++;
    gamma_im1 = gamma;
    const int had = ahead;
    ahead         = 0;
    for (int ka = had ? 1 : 0; ka < 2; ka++) { /* ka = 0: this iteration's kernels (unless queued ahead); ka = 1: the next iteration's, before the host looks at this one's sums */
      const hipx_int k = i + ka;
      if (ka == 1 && !(ksp->pipeline && k <= ksp->max_it)) break;
      if (k == 1) { /* t_1 = p.s (groppcg.c:87) */
        const double *ys[1] = {s};
        if (A->nranks > 1) CHK(hipxVecMDotBeginAllreduce(p, 1, ys, n, SLOT_T + (int)(k & 1), GT(k)));
        else CHK(hipxVecMDotBegin(p, 1, ys, n, SLOT_T + (int)(k & 1), GT(k)));
      } else if (A->nranks > 1) CHK(hipxGroppCGDirectionBeginAllreduce(p, s, X, z, Z, GSUMS(k - 1) + 1, GSUMS(k - 2) + 1, GALPHA(k - 1), n, SLOT_T + (int)(k & 1), GT(k)));
      else CHK(hipxGroppCGDirectionBegin(p, s, X, z, Z, GSUMS(k - 1) + 1, GSUMS(k - 2) + 1, GALPHA(k - 1), n, SLOT_T + (int)(k & 1), GT(k)));
      if (A->nranks > 1) CHK(hipxGroppCGUpdateBeginAllreduce(r, z, s, dinv, pc->dconst, nrm, GSUMS(k - 1) + 1, GT(k), GALPHA(k), n, SLOT_U + (int)(k & 1)));
      else CHK(hipxGroppCGUpdateBegin(r, z, s, dinv, pc->dconst, nrm, GSUMS(k - 1) + 1, GT(k), GALPHA(k), n, SLOT_U + (int)(k & 1), GSUMS(k)));
      CHK(HipxMatMult(A, z, Z)); /* groppcg.c:111 */
      if (A->nranks > 1) CHK(hipxAllreduceEnd(SLOT_U + (int)(k & 1), 2, GSUMS(k)));
      ahead = ka; /* (1 after the second pass: iteration i + 1 is queued) */
    }
    xpend = 1; /* U(i) is queued: x += alpha_i p_i waits for D(i + 1) or for the flush */
    CHK(hipxRedEnd(SLOT_U + (int)(i & 1), 2, sums)); /* {dp^2, gammaNew_i} */
    if (A->nranks > 1) CHK(hipxCommCheckError());
    gammaNew = sums[1];
    if (ksp->normtype == HIPX_KSP_NORM_NATURAL) {
      if (isnan(gammaNew) || isinf(gammaNew)) { /* KSPCheckDot, groppcg.c:120 */
        ksp->reason = KSP_DIVERGED_NANORINF;
        break;
      }
      dp = sqrt(fabs(gammaNew));
    } else if (ksp->normtype == HIPX_KSP_NORM_NONE) dp = 0.0;
    else dp = sqrt(sums[0]);
    ksp->rnorm = dp;
    log_history(ksp, dp);
    CHK(converged_default(ksp, A, pc, i, dp, B, &ksp->reason)); /* groppcg.c:129 */
    if (ksp->reason) break;
    gamma = gammaNew; /* groppcg.c:133 (beta and the p, s updates are D(i + 1)'s) */
  } while (i < ksp->max_it);
  if (ahead) xpend = 0; /* D(i + 1), queued ahead, applies x += alpha_i p_i: exactly the update that was due */
  CHK(hipxStreamSynchronize());
  if (xpend && n > 0) { /* groppcg.c:98 of the last iteration: alpha_i = gamma_{i-1} / t_i, the quotient U(i) formed */
    CHK(hipxMemcpyDtoH(&t, GT(i), sizeof(double)));
    CHK(hipxVecAXPY(X, gamma_im1 / t, p, n));
  }
  if (!ksp->reason && i >= ksp->max_it) ksp->reason = KSP_DIVERGED_ITS; /* groppcg.c:139 */
  return 0;
#undef GSUMS
#undef GT
#undef GALPHA
}

/* KSPSolve_Chebyshev_FirstKind (cheby.c:389-555), statement by statement, with the eigenvalue bounds given (-ksp_chebyshev_eigenvalues /
   KSPChebyshevSetEigenvalues: cheby.c:40-62 returns them as they are).  With no norm requested (the smoother configuration:
   KSP_NORM_NONE) and PCJACOBI or PCNONE an iteration is the SpMV plus ONE elementwise kernel (hipxVecChebyshevStep: residual, PC
   application and the three-term update in one pass, bit-identical to the three reference loops); otherwise the reference's own
   sequence of MatMult / VecAYPX / PCApply / VecNorm / VecAXPBYPCZ.  No reduction anywhere on the fused path: the solution is
   bit-identical to the CPU run on any number of ranks. */
enum { KSP_CONVERGED_ITS = 4 };
int HipxKSPSolve_Chebyshev(HipxKSP *ksp, HipxMat *A, HipxPC *pc, const double *B, double *X, double emin, double emax)
{
  const hipx_int n = A->m;
  double        *p[3], *r = NULL, rnorm = 0.0;
  int            km1 = 0, k = 1, kp1 = 2, ierr = 0;
  const size_t   bytes = sizeof(double) * (size_t)(n ? n : 1);
  const int      fast = (ksp->normtype == HIPX_KSP_NORM_NONE) && (pc->type == HIPX_PC_NONE || pc->type == HIPX_PC_JACOBI) && ksp->fused;
#define CCHK(call) \
  do { \
    ierr = (call); \
    if (ierr) goto cleanup; \
  } while (0)
  p[0] = X;
  p[1] = p[2] = NULL;
  CCHK(hipxMalloc((void **)&p[1], bytes));
  CCHK(hipxMalloc((void **)&p[2], bytes));
  CCHK(hipxMalloc((void **)&r, bytes));
  ksp->its    = 0;
  ksp->reason = 0;
  ksp->hist_n = 0;
  {
    const double scale = 2.0 / (emax + emin), alpha = 1.0 - scale * emin, Gamma = 1.0, mu = 1.0 / alpha, omegaprod = 2.0 / alpha; /* cheby.c:420-426 */
    double       c[3], omega;
    hipx_int     i;
    c[km1] = 1.0;
    c[k]   = mu;
    if (ksp->guess_nonzero) { /* cheby.c:431-436 */
      CCHK(HipxMatMult(A, X, r));
      CCHK(hipxVecAYPX(r, -1.0, B, n));
    } else {
      CCHK(hipxVecSet(X, n, 0.0)); /* KSPSolve zeroes the solution for a zero initial guess (itfunc.c) */
      CCHK(hipxVecCopy(B, r, n));
    }
    if (ksp->normtype) { /* cheby.c:439-459 */
      if (ksp->normtype == HIPX_KSP_NORM_PRECONDITIONED) {
        CCHK(HipxPCApply(pc, A, r, p[k]));
        CCHK(HipxVecNorm2(A, p[k], n, &rnorm));
      } else CCHK(HipxVecNorm2(A, r, n, &rnorm));
      ksp->rnorm = rnorm;
      log_history(ksp, rnorm);
      CCHK(converged_default(ksp, A, pc, 0, rnorm, B, &ksp->reason));
    } else ksp->reason = KSP_CONVERGED_ITERATING;
    if (ksp->reason || ksp->max_it == 0) {
      if (ksp->max_it == 0) ksp->reason = KSP_DIVERGED_ITS;
      goto cleanup;
    }
    if (ksp->normtype != HIPX_KSP_NORM_PRECONDITIONED) CCHK(HipxPCApply(pc, A, r, p[k])); /* cheby.c:464 */
    CCHK(hipxVecAYPX(p[k], scale, p[km1], n));                                             /* p[k] = scale B^{-1} r + p[km1] */
    ksp->its = 1;
    for (i = 1; i < ksp->max_it; i++) { /* cheby.c:470-517 */
      int ktmp;
      ksp->its++;
      if (fast && A->nranks == 1) { /* one kernel: the step as the SpMV's epilogue (hipxMatMultChebyshev) */
        c[kp1] = 2.0 * mu * c[k] - c[km1];
        omega  = omegaprod * c[k] / c[kp1];
        CCHK(hipxMatMultChebyshev(A->A, p[k], p[kp1], 1.0 - omega, omega, omega * Gamma * scale, p[km1], pc->type == HIPX_PC_JACOBI ? pc->dinv : NULL, B));
      } else if (fast) {
        CCHK(HipxMatMult(A, p[k], r));
        c[kp1] = 2.0 * mu * c[k] - c[km1];
        omega  = omegaprod * c[k] / c[kp1];
        CCHK(hipxVecChebyshevStep(p[kp1], 1.0 - omega, omega, omega * Gamma * scale, p[km1], p[k], pc->type == HIPX_PC_JACOBI ? pc->dinv : NULL, B, r, NULL, n));
      } else {
        CCHK(HipxMatMult(A, p[k], r));
        CCHK(hipxVecAYPX(r, -1.0, B, n));
        if (ksp->normtype) {
          if (ksp->normtype == HIPX_KSP_NORM_PRECONDITIONED) {
            CCHK(HipxPCApply(pc, A, r, p[kp1]));
            CCHK(HipxVecNorm2(A, p[kp1], n, &rnorm));
          } else CCHK(HipxVecNorm2(A, r, n, &rnorm));
          if (isnan(rnorm) || isinf(rnorm)) { /* KSPCheckNorm */
            ksp->reason = KSP_DIVERGED_NANORINF;
            break;
          }
          ksp->rnorm = rnorm;
          log_history(ksp, rnorm);
          CCHK(converged_default(ksp, A, pc, i, rnorm, B, &ksp->reason));
          if (ksp->reason) break;
          if (ksp->normtype != HIPX_KSP_NORM_PRECONDITIONED) CCHK(HipxPCApply(pc, A, r, p[kp1]));
        } else CCHK(HipxPCApply(pc, A, r, p[kp1]));
        c[kp1] = 2.0 * mu * c[k] - c[km1];
        omega  = omegaprod * c[k] / c[kp1];
        CCHK(hipxVecAXPBYPCZ(p[kp1], 1.0 - omega, omega, omega * Gamma * scale, p[km1], p[k], n));
      }
      ktmp = km1;
      km1  = k;
      k    = kp1;
      kp1  = ktmp;
    }
    if (!ksp->reason) { /* cheby.c:518-548 */
      if (ksp->normtype) {
        CCHK(HipxMatMult(A, p[k], r));
        CCHK(hipxVecAYPX(r, -1.0, B, n));
        if (ksp->normtype == HIPX_KSP_NORM_PRECONDITIONED) {
          CCHK(HipxPCApply(pc, A, r, p[kp1]));
          CCHK(HipxVecNorm2(A, p[kp1], n, &rnorm));
        } else CCHK(HipxVecNorm2(A, r, n, &rnorm));
        if (isnan(rnorm) || isinf(rnorm)) ksp->reason = KSP_DIVERGED_NANORINF;
        else {
          ksp->rnorm = rnorm;
          log_history(ksp, rnorm);
        }
      }
      if (!ksp->reason && ksp->its >= ksp->max_it) {
        if (ksp->normtype != HIPX_KSP_NORM_NONE) {
          CCHK(converged_default(ksp, A, pc, i, rnorm, B, &ksp->reason));
          if (!ksp->reason) ksp->reason = KSP_DIVERGED_ITS;
        } else ksp->reason = KSP_CONVERGED_ITS;
      }
    }
    /* cheby.c:551-552: the solution ends in the user's vector -- but KSPCheckNorm (cheby.c:491) RETURNS on a NaN/Inf norm: vec_sol keeps what it held */
    if (k && ksp->reason != KSP_DIVERGED_NANORINF) CCHK(hipxVecCopy(p[k], X, n));
  }
cleanup:
  if (p[1]) (void)hipxFree(p[1]);
  if (p[2]) (void)hipxFree(p[2]);
  if (r) (void)hipxFree(r);
#undef CCHK
  return ierr;
}

/* gmres.c:88-238 (cycle + solve), :298-345 (BuildSoln), :349-395 (UpdateHessenberg), borthog2.c:35-113,
   left preconditioning, KSPInitialResidual itres.c:35-75.  VV(0..max_k) live in one contiguous slab
   (cf. VecDuplicateVecs_Seq_GEMV bvec2.c:670) so MDot/MAXPY stream through consecutive memory. */
/* VecMDot_MPI (pvecimpl.h:97-111): all-reduces of <= 16 sums, so restarts of any length work (the reference has no limit on gmres_restart) */
/* VecMDot on one rank or several.  Several: 16 vectors per chain (local kernel -> all-reduce on the stream -> one host wait) -- in the exact reduction
   mode the ranks' sums travel as unrounded (hi, lo) pairs and are rounded once after the fold over ranks (round 4), so the orthogonalisation
   coefficients are the correctly rounded global dot products whatever the cut into ranks */
static int mdot_ranks(HipxMat *A, const double *x, hipx_int nv, const double *const *y, hipx_int n, double *out)
{
  if (A->nranks <= 1) return hipxVecMDot(x, nv, y, n, out);
  for (hipx_int k = 0; k < nv; k += 16) CHK(hipxVecMDotAllreduce(x, (nv - k) < 16 ? (nv - k) : 16, y + k, n, out + k));
  return 0;
}

int HipxKSPSolve_GMRES(HipxKSP *ksp, HipxMat *A, HipxPC *pc, const double *B, double *X)
{
  const hipx_int n = A->m, max_k = ksp->gmres_restart;
  const size_t   ld = ((size_t)(n ? n : 1) + 1) & ~(size_t)1; /* keep every basis vector 16-byte aligned */
  double        *slab = NULL, *TEMP, *TMOP;
  double       **VV   = (double **)malloc(sizeof(double *) * (size_t)(max_k + 2));
  double        *hh   = (double *)calloc((size_t)(max_k + 2) * (size_t)(max_k + 1), sizeof(double));
  double        *grs  = (double *)calloc((size_t)(max_k + 2), sizeof(double));
  double        *cc   = (double *)calloc((size_t)(max_k + 2), sizeof(double));
  double        *ss   = (double *)calloc((size_t)(max_k + 2), sizeof(double));
  double        *nrs  = (double *)calloc((size_t)(max_k + 2), sizeof(double));
  double        *lhh  = (double *)calloc((size_t)(max_k + 2), sizeof(double));
  hipx_int       itcount = 0;
  const int      guess_nonzero = ksp->guess_nonzero;
  int            ierr = 0;
#define HH(a, b) (hh + (size_t)(b) * (size_t)(max_k + 2) + (a))
#define GCHK(call) \
  do { \
    ierr = (call); \
    if (ierr) goto cleanup; \
  } while (0)
  /* the work vectors live as long as the KSP (KSPSetUp_GMRES, gmres.c:35-86: allocated once, reused by every KSPSolve): a 4.6 GB hipMalloc +
     hipFree pair per solve of the 27-pt 256^3 system cost up to 0.3 s inside the timed region of a bench leg that ran late in the process */
  if (!ksp->gslab || ksp->gslab_len < (double)ld * (double)(max_k + 4)) {
    if (ksp->gslab) GCHK(hipxFree(ksp->gslab));
    ksp->gslab     = NULL;
    ksp->gslab_len = 0.0;
    GCHK(hipxMalloc((void **)&ksp->gslab, sizeof(double) * ld * (size_t)(max_k + 4)));
    ksp->gslab_len = (double)ld * (double)(max_k + 4);
  }
  slab = ksp->gslab;
  TEMP = slab;
  TMOP = slab + ld;
  for (hipx_int k = 0; k < max_k + 2; k++) VV[k] = slab + ld * (size_t)(k + 2);
  ksp->its    = 0;
  ksp->reason = 0;
  ksp->hist_n = 0;
  ksp->rnorm  = -1.0;
  if (!ksp->guess_nonzero) GCHK(hipxVecSet(X, n, 0.0));

  while (!ksp->reason) {
    double   res = 0.0, tt, hapbnd;
    hipx_int it     = 0;
    int      hapend = 0;
    if (ksp->guess_nonzero) { /* KSPInitialResidual, PC_LEFT */
      GCHK(HipxMatMult(A, X, TEMP));
      GCHK(hipxVecCopy(B, TMOP, n));
      GCHK(hipxVecAXPY(TMOP, -1.0, TEMP, n));
      GCHK(HipxPCApply(pc, A, TMOP, VV[0]));
    } else {
      GCHK(hipxVecCopy(B, TMOP, n));
      GCHK(HipxPCApply(pc, A, B, VV[0]));
    }
    GCHK(HipxVecNorm2(A, VV[0], n, &res)); /* VecNormalize gmres.c:98 */
    if (isnan(res) || isinf(res)) {
      ksp->reason = KSP_DIVERGED_NANORINF;
      break;
    }
    if (res != 0.0) GCHK(hipxVecScale(VV[0], n, 1.0 / res));
    grs[0]     = res;
    ksp->rnorm = res;
    log_history(ksp, res);
    if (!res) {
      ksp->reason = KSP_CONVERGED_ATOL;
      break;
    }
    GCHK(converged_default(ksp, A, pc, ksp->its, res, B, &ksp->reason));
    while (!ksp->reason && it < max_k && ksp->its < ksp->max_it) {
      if (it) log_history(ksp, res);
      GCHK(HipxMatMult(A, VV[it], TMOP)); /* KSP_PCApplyBAorAB, left */
      GCHK(HipxPCApply(pc, A, TMOP, VV[it + 1]));
      { /* borthog2.c:35-113 */
        double *h      = HH(0, it);
        int     refine = (ksp->gmres_cgs_refine == 2);
        for (hipx_int j = 0; j <= it; j++) h[j] = 0.0;
        GCHK(mdot_ranks(A, VV[it + 1], it + 1, (const double *const *)VV, n, lhh));
        for (hipx_int j = 0; j <= it; j++) {
          if (isnan(lhh[j]) || isinf(lhh[j])) ksp->reason = KSP_DIVERGED_NANORINF;
          lhh[j] = -lhh[j];
        }
        if (ksp->reason) break;
        GCHK(hipxVecMAXPY(VV[it + 1], it + 1, lhh, (const double *const *)VV, n));
        for (hipx_int j = 0; j <= it; j++) h[j] -= lhh[j];
        if (ksp->gmres_cgs_refine == 1) {
          double hnrm = 0.0, wnrm;
          for (hipx_int j = 0; j <= it; j++) hnrm += lhh[j] * lhh[j];
          hnrm = sqrt(hnrm);
          GCHK(HipxVecNorm2(A, VV[it + 1], n, &wnrm));
          if (wnrm < hnrm) refine = 1;
        }
        if (refine) {
          GCHK(mdot_ranks(A, VV[it + 1], it + 1, (const double *const *)VV, n, lhh));
          for (hipx_int j = 0; j <= it; j++) lhh[j] = -lhh[j];
          GCHK(hipxVecMAXPY(VV[it + 1], it + 1, lhh, (const double *const *)VV, n));
          for (hipx_int j = 0; j <= it; j++) h[j] -= lhh[j];
        }
      }
      GCHK(HipxVecNorm2(A, VV[it + 1], n, &tt)); /* VecNormalize gmres.c:143 */
      if (isnan(tt) || isinf(tt)) {
        ksp->reason = KSP_DIVERGED_NANORINF;
        break;
      }
      if (tt != 0.0) GCHK(hipxVecScale(VV[it + 1], n, 1.0 / tt));
      *HH(it + 1, it) = tt;
      hapbnd          = fabs(tt / grs[it]);
      if (hapbnd > ksp->gmres_haptol) hapbnd = ksp->gmres_haptol;
      if (tt < hapbnd) hapend = 1;
      { /* KSPGMRESUpdateHessenberg gmres.c:349-395 */
        double *h = HH(0, it), *cp = cc, *sp = ss, t;
        for (hipx_int j = 1; j <= it; j++) {
          t  = *h;
          *h = *cp * t + *sp * *(h + 1);
          h++;
          *h = *cp++ * *h - (*sp++ * t);
        }
        if (!hapend) {
          t = sqrt(*h * *h + *(h + 1) * *(h + 1));
          if (t == 0.0) { /* gmres.c:374-378: the reason is set, the cycle still counts this iteration (it++, its++) before it leaves */
            ksp->reason = KSP_DIVERGED_NULL;
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
      ksp->rnorm = res;
      GCHK(converged_default(ksp, A, pc, ksp->its, res, B, &ksp->reason));
      if (hapend) {
        if (ksp->normtype == HIPX_KSP_NORM_NONE) ksp->reason = KSP_CONVERGED_HAPPY_BREAKDOWN;
        else if (!ksp->reason) {
          ksp->reason = KSP_DIVERGED_BREAKDOWN;
          break;
        }
      }
    }
    if (it - 1 >= 0) { /* KSPGMRESBuildSoln gmres.c:298-345 */
      hipx_int itl = it - 1;
      if (*HH(itl, itl) != 0.0) {
        nrs[itl] = grs[itl] / *HH(itl, itl);
        for (hipx_int ii = 1; ii <= itl; ii++) {
          hipx_int k = itl - ii;
          double   t = grs[k];
          for (hipx_int j = k + 1; j <= itl; j++) t = t - *HH(k, j) * nrs[j];
          nrs[k] = t / *HH(k, k);
        }
        GCHK(hipxVecMAXPBY(TEMP, itl + 1, nrs, 0.0, (const double *const *)VV, n)); /* gmres.c:337 */
        GCHK(hipxVecAXPY(X, 1.0, TEMP, n));                                         /* gmres.c:342 */
      } else ksp->reason = KSP_DIVERGED_BREAKDOWN;
    }
    if (ksp->reason == 0 && ksp->its >= ksp->max_it) ksp->reason = KSP_DIVERGED_ITS;
    if (it && ksp->reason) log_history(ksp, res);
    itcount += it;
    if (itcount >= ksp->max_it) {
      if (!ksp->reason) ksp->reason = KSP_DIVERGED_ITS;
      break;
    }
    ksp->guess_nonzero = 1; /* gmres.c:233 */
  }
cleanup:
  ksp->guess_nonzero = guess_nonzero;
  free(VV);
  free(hh);
  free(grs);
  free(cc);
  free(ss);
  free(nrs);
  free(lhh);
#undef HH
#undef GCHK
  return ierr;
}
