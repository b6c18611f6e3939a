/*
 * ksphipx.c -- KSPCGHIPX ("cghipx"): KSPCG whose solve runs the fused device kernels of libhipx when the configuration is
 * the hot path's (MATSEQAIJHIPX operator, or MATMPIAIJHIPX with the device ghost exchange; default PCJACOBI, left preconditioning, preconditioned norm, no
 * eigenvalue estimates / trust radius / single-reduction variant), and the reference's own KSPSolve_CG otherwise.
 *
 * Subclass recipe as for the Mat types: KSPCreate_CG (PETSC_EXTERN, cg.c:686) builds the object, we keep its solve op as the
 * fall-back and install ours.  The fused solve is the C host layer's stepping CG (include/hipx_ksp.h: the statement-by-
 * statement mirror of cg.c:119-352), one iteration per call, with PETSc's OWN residual history, monitors and convergence
 * test invoked between iterations exactly where KSPSolve_CG invokes them (cg.c:196-205, 326-328) -- so -ksp_monitor,
 * -ksp_converged_reason, KSPSetConvergenceTest keep working.  Per iteration: SpMV+dot, fused update (r, z, two sums),
 * AYPX+deferred AXPY = 3 kernels + 1 partial fold instead of 8 launches and 3 blocking reductions.
 */
#include "hipxplugin.h"
#include <stdlib.h>
#include <petsc/private/kspimpl.h>
#include <../src/ksp/ksp/impls/cg/cgimpl.h>
#include "hipx_ksp.h"

#define KSPCGHIPX "cghipx"

static PetscErrorCode (*parent_solve_cg)(KSP) = NULL;

static PetscBool KSPCGHIPXApplicable(KSP ksp, Mat *Aout)
{
  KSP_CG       *cg = (KSP_CG *)ksp->data;
  Mat           Amat, Pmat;
  PetscBool     isjac = PETSC_FALSE, isnone = PETSC_FALSE, useabs = PETSC_FALSE, fixdiag = PETSC_TRUE;
  PCJacobiType  jt;
  PetscMPIInt   size;

  if (ksp->calc_sings || cg->radius != 0.0 || cg->type != KSP_CG_SYMMETRIC || cg->obj_min != 0.0) return PETSC_FALSE; /* (-ksp_cg_single_reduction: round 4, HipxKSP.single_reduction) */
  if (ksp->pc_side != PC_LEFT || ksp->normtype != KSP_NORM_PRECONDITIONED || ksp->transpose_solve) return PETSC_FALSE;
  if (ksp->dscale) return PETSC_FALSE;
  if (MPI_Comm_size(PetscObjectComm((PetscObject)ksp), &size)) return PETSC_FALSE;
  if (PCGetOperators(ksp->pc, &Amat, &Pmat) || Amat != Pmat || Amat->rmap->n != Amat->cmap->n) return PETSC_FALSE;
  if (size == 1) {
    if (!MatIsSeqAIJHIPX(Amat)) return PETSC_FALSE;
  } else { /* MATMPIAIJHIPX with its ghost exchange and the scalar all-reduces on the device (RCCL or IPC transport) */
    PetscBool ismpi = PETSC_FALSE;
    hipxMat   dA, dB;
    hipxHalo  halo = NULL;
    Vec       lvec;
    if (PetscObjectTypeCompare((PetscObject)Amat, MATMPIAIJHIPX, &ismpi) || !ismpi) return PETSC_FALSE;
    if (MatMPIAIJHIPXGetDevice(Amat, &dA, &dB, &halo, &lvec) || !halo) return PETSC_FALSE;
  }
  if (PetscObjectTypeCompare((PetscObject)ksp->pc, PCJACOBI, &isjac)) return PETSC_FALSE;
  if (PetscObjectTypeCompare((PetscObject)ksp->pc, PCNONE, &isnone)) return PETSC_FALSE; /* PCApply_None = VecCopy (none.c:6): the fused kernels with the constant 1.0 */
  if (!isjac && !isnone) return PETSC_FALSE;
  if (isjac) {
    if (PCJacobiGetType(ksp->pc, &jt) || jt != PC_JACOBI_DIAGONAL) return PETSC_FALSE;
    if (PCJacobiGetUseAbs(ksp->pc, &useabs) || useabs) return PETSC_FALSE;
    if (PCJacobiGetFixDiagonal(ksp->pc, &fixdiag) || !fixdiag) return PETSC_FALSE;
  }
  if (!VecIsHIPX(ksp->vec_rhs) || !VecIsHIPX(ksp->vec_sol)) return PETSC_FALSE;
  {
    MatNullSpace nsp = NULL;
    if (MatGetNullSpace(Amat, &nsp) || nsp) return PETSC_FALSE; /* KSPSolve removes it through the PC; keep that on the reference path */
  }
  *Aout = Amat;
  return PETSC_TRUE;
}

static PetscErrorCode KSPSolve_CGHIPX(KSP ksp)
{
  Mat                Amat = NULL;
  hipxMat            dA;
  HipxMat            M;
  HipxPC             hpc;
  HipxKSP            k;
  const PetscScalar *db;
  PetscScalar       *dx;
  void              *tb, *tx, *tlv = NULL;
  PetscInt           n;
  PetscMPIInt        size;
  Vec                lvecv = NULL;
  PetscScalar       *dlv   = NULL;
  int                herr  = 0;
  char               herrmsg[512] = "";

  PetscFunctionBegin;
  if (!KSPCGHIPXApplicable(ksp, &Amat)) {
    PetscCall(PetscInfo(ksp, "KSPCGHIPX: configuration outside the fused path, running the reference KSPSolve_CG\n"));
    PetscCall((*parent_solve_cg)(ksp));
    PetscFunctionReturn(PETSC_SUCCESS);
  }
  n = Amat->rmap->n;
  PetscCallMPI(MPI_Comm_size(PetscObjectComm((PetscObject)ksp), &size));
  if (size == 1) {
    PetscCall(MatSeqAIJHIPXGetDeviceMat(Amat, &dA));
    M.m = (hipx_int)n; M.A = dA; M.B = NULL; M.halo = NULL; M.lvec = NULL; M.nranks = 1;
  } else {
    hipxMat  dB;
    hipxHalo halo;
    PetscCall(MatMPIAIJHIPXGetDevice(Amat, &dA, &dB, &halo, &lvecv));
    PetscCall(VecHIPXGetDeviceWrite(lvecv, &dlv, &tlv));
    M.m = (hipx_int)n; M.A = dA; M.B = dB; M.halo = halo; M.lvec = dlv; M.nranks = (int)size;
  }
  HipxPCSetDefaults(&hpc);
  {
    PetscBool isnone = PETSC_FALSE;
    PetscCall(PetscObjectTypeCompare((PetscObject)ksp->pc, PCNONE, &isnone));
    hpc.type = isnone ? HIPX_PC_NONE : HIPX_PC_JACOBI;
  }
  PetscCallHIPX(HipxPCSetUp(&hpc, &M)); /* 1/diag, 0 -> 1: jacobi.c:205-266 on the device */
  HipxKSPSetDefaults(&k);
  k.normtype      = HIPX_KSP_NORM_PRECONDITIONED;
  k.max_it        = (hipx_int)ksp->max_it;
  k.guess_nonzero = ksp->guess_zero ? 0 : 1;
  k.fused         = 1;
  k.single_reduction = ((KSP_CG *)ksp->data)->singlereduction ? 1 : 0; /* KSPSolve_CG_SingleReduction (cg.c:364-534): one reduction stage per iteration */
  k.external_test = 1;                                  /* PETSc's (*ksp->converged) decides */
  k.defer_flush   = ksp->numbermonitors ? 0 : 1;        /* monitors may look at the solution: keep x complete for them */
  PetscCall(VecHIPXGetDeviceRead(ksp->vec_rhs, &db, &tb));
  PetscCall(VecHIPXGetDeviceReadWrite(ksp->vec_sol, &dx, &tx));

  ksp->its = 0;
  /* Round 4: when nothing outside wants to look between iterations -- no monitors, the default convergence test with its default context
     (KSPConvergedDefault, iterativ.c:1490-1585: mirrored statement by statement in the host layer), no lag / check-norm settings -- the host layer runs
     the whole loop itself: launch-ahead (iteration i + 1 enqueued before the host has seen iteration i's sums) and the direction update as the product's
     prologue, the forms `bench.py` times.  The residual history, its, rnorm, reason, rnorm0 and ttol are handed to the KSP afterwards exactly as the
     stepwise loop below leaves them.  Anything else (-ksp_monitor, KSPSetConvergenceTest, ...) takes the stepwise loop. */
  {
    PetscBool selfdriven = PETSC_FALSE;
    if (!ksp->numbermonitors && ksp->converged == KSPConvergedDefault && ksp->cnvP && ksp->chknorm < 0 && !ksp->lagnorm && !getenv("HIPX_CGHIPX_STEPWISE")) { /* (chknorm = -1: KSPCreate's default, itcreate.c:816 -- the test at every iteration, the 0th included) */
      KSPConvergedDefaultCtx *cctx = (KSPConvergedDefaultCtx *)ksp->cnvP;
      if (!cctx->initialrtol && !cctx->mininitialrtol && !cctx->convmaxits) selfdriven = PETSC_TRUE;
    }
    if (selfdriven) {
      double  *hist = NULL;
      hipx_int hl   = (hipx_int)((ksp->max_it < 1000000 ? ksp->max_it : 1000000) + 2);
      PetscCall(PetscMalloc1((size_t)hl, &hist));
      k.external_test = 0;
      k.defer_flush   = 0;
      k.rtol          = ksp->rtol;
      k.abstol        = ksp->abstol;
      k.divtol        = ksp->divtol;
      k.min_it        = (hipx_int)ksp->min_it;
      k.history       = hist;
      k.hist_len      = hl;
      herr = HipxKSPSolve_CG(&k, &M, &hpc, db, dx); /* (an error leaves through `done`: the history is freed, the device handles go back) */
      if (!herr) {
        for (hipx_int e = 0; e < k.hist_n && e < hl; e++) PetscCall(KSPLogResidualHistory(ksp, hist[e]));
        ksp->its    = (PetscInt)k.its;
        ksp->rnorm  = k.rnorm;
        ksp->rnorm0 = k.rnorm0;
        ksp->ttol   = k.ttol;
        ksp->reason = (KSPConvergedReason)k.reason;
        if (ksp->reason == KSP_DIVERGED_NANORINF) { /* the host layer's mirror of KSPConvergedDefault knows no PC: a NaN / Inf norm after a failed
                                                       preconditioner set-up is KSP_DIVERGED_PC_FAILED (iterativ.c:1548-1559) */
          PCFailedReason pcreason;
          PetscCall(PCReduceFailedReason(ksp->pc));
          PetscCall(PCGetFailedReason(ksp->pc, &pcreason));
          if (pcreason) ksp->reason = KSP_DIVERGED_PC_FAILED;
        }
      }
      PetscCall(PetscFree(hist));
      goto done;
    }
  }
  PetscCallHIPX(HipxKSPCGBegin(&k, &M, &hpc, db, dx)); /* cg.c:134-217 */
  if (k.reason) ksp->reason = (KSPConvergedReason)k.reason; /* KSPCheckNorm: NaN/Inf */
  else {
    ksp->rnorm = k.rnorm;
    PetscCall(KSPLogResidualHistory(ksp, k.rnorm));
    PetscCall(KSPMonitor(ksp, 0, k.rnorm));
    PetscCall((*ksp->converged)(ksp, 0, k.rnorm, &ksp->reason, ksp->cnvP)); /* cg.c:205 */
  }
  while (!ksp->reason) {
    const hipx_int ibefore = k.i;
    PetscCallHIPX(HipxKSPCGStep(&k, &M, &hpc, db, dx, 1)); /* one pass of cg.c:220-349 */
    ksp->its = (PetscInt)k.its;
    if (k.i > ibefore) { /* the pass completed: history, monitors and the convergence test as in cg.c:326-328 */
      ksp->rnorm = k.rnorm;
      PetscCall(KSPLogResidualHistory(ksp, k.rnorm));
      PetscCall(KSPMonitor(ksp, (PetscInt)k.i, k.rnorm));
      PetscCall((*ksp->converged)(ksp, (PetscInt)k.i, k.rnorm, &ksp->reason, ksp->cnvP));
    }
    /* breakdown / NaN inside the pass (cg.c:223-232,262-268, KSPCheckDot/KSPCheckNorm) or max_it reached (cg.c:350) */
    if (!ksp->reason && k.reason) ksp->reason = (KSPConvergedReason)k.reason;
  }
  PetscCallHIPX(HipxKSPCGFlush(&k, &M, dx));
done:
  if (herr) PetscCall(PetscStrncpy(herrmsg, hipxGetErrorString(), sizeof(herrmsg))); /* (the clean-up calls below may overwrite the library's message) */
  (void)HipxKSPDestroyWork(&k);
  (void)HipxPCDestroy(&hpc);
  if (lvecv) PetscCall(VecHIPXRestoreDeviceWrite(lvecv, &dlv, &tlv));
  PetscCall(VecHIPXRestoreDeviceWrite(ksp->vec_sol, &dx, &tx));
  PetscCall(VecHIPXRestoreDeviceRead(ksp->vec_rhs, &db, &tb));
  PetscCall(PetscObjectStateIncrease((PetscObject)ksp->vec_sol));
  PetscCheck(herr != HIPX_ERR_SUP, PetscObjectComm((PetscObject)ksp), PETSC_ERR_SUP, "libhipx: %s", herrmsg);
  PetscCheck(!herr, PetscObjectComm((PetscObject)ksp), PETSC_ERR_GPU, "libhipx: %s", herrmsg);
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* ---- KSPCHEBYSHEVHIPX ("chebyshevhipx"): KSPCHEBYSHEV whose first-kind solve with no norm requested (the smoother configuration) runs
   one SpMV + one fused elementwise kernel per iteration (HipxKSPSolve_Chebyshev: residual, Jacobi application and the three-term
   update of cheby.c:475-511 in one pass, no reductions): bit-identical solution.  Eigenvalue estimation (the kspest machinery of
   KSPSetUp_Chebyshev), the fourth-kind polynomials, norms / monitors and every other configuration stay the parent's. */
#include <../src/ksp/ksp/impls/cheby/chebyshevimpl.h>

static PetscErrorCode (*parent_setup_cheby)(KSP) = NULL; /* KSPSetUp_Chebyshev: it (re)installs the solve op of the polynomial kind at every set-up */

static PetscBool KSPChebyshevHIPXApplicable(KSP ksp, Mat *Aout, PetscBool *jac, PetscReal *emin, PetscReal *emax)
{
  KSP_Chebyshev *cheb = (KSP_Chebyshev *)ksp->data;
  Mat            Amat, Pmat;
  PetscBool      isjac = PETSC_FALSE, isnone = PETSC_FALSE, useabs = PETSC_FALSE, fixdiag = PETSC_TRUE;
  PCJacobiType   jt;
  PetscMPIInt    size;

  if (cheb->chebykind != KSP_CHEBYSHEV_FIRST || ksp->normtype != KSP_NORM_NONE || ksp->numbermonitors || ksp->transpose_solve || ksp->dscale) return PETSC_FALSE;
  if (MPI_Comm_size(PetscObjectComm((PetscObject)ksp), &size) || size != 1) return PETSC_FALSE;
  if (PCGetOperators(ksp->pc, &Amat, &Pmat) || Amat != Pmat || Amat->rmap->n != Amat->cmap->n || !MatIsSeqAIJHIPX(Amat)) return PETSC_FALSE;
  if (PetscObjectTypeCompare((PetscObject)ksp->pc, PCJACOBI, &isjac) || PetscObjectTypeCompare((PetscObject)ksp->pc, PCNONE, &isnone) || (!isjac && !isnone)) return PETSC_FALSE;
  if (isjac) {
    if (PCJacobiGetType(ksp->pc, &jt) || jt != PC_JACOBI_DIAGONAL) return PETSC_FALSE;
    if (PCJacobiGetUseAbs(ksp->pc, &useabs) || useabs) return PETSC_FALSE;
    if (PCJacobiGetFixDiagonal(ksp->pc, &fixdiag) || !fixdiag) return PETSC_FALSE;
  }
  if (!VecIsHIPX(ksp->vec_rhs) || !VecIsHIPX(ksp->vec_sol)) return PETSC_FALSE;
  {
    MatNullSpace nsp = NULL;
    if (MatGetNullSpace(Amat, &nsp) || nsp) return PETSC_FALSE;
  }
  /* the bounds exactly as KSPChebyshevGetEigenvalues_Chebyshev (cheby.c:40-62) returns them */
  *emax = *emin = 0;
  if (cheb->emax != 0.) *emax = cheb->emax;
  else if (cheb->emax_computed != 0.) *emax = cheb->tform[2] * cheb->emin_computed + cheb->tform[3] * cheb->emax_computed;
  else if (cheb->emax_provided != 0.) *emax = cheb->tform[2] * cheb->emin_provided + cheb->tform[3] * cheb->emax_provided;
  if (cheb->emin != 0.) *emin = cheb->emin;
  else if (cheb->emin_computed != 0.) *emin = cheb->tform[0] * cheb->emin_computed + cheb->tform[1] * cheb->emax_computed;
  else if (cheb->emin_provided != 0.) *emin = cheb->tform[0] * cheb->emin_provided + cheb->tform[1] * cheb->emax_provided;
  if (*emax == 0. || *emax + *emin == 0.) return PETSC_FALSE;
  *Aout = Amat;
  *jac  = isjac;
  return PETSC_TRUE;
}

static PetscErrorCode KSPSolve_ChebyshevHIPX(KSP ksp)
{
  Mat                Amat = NULL;
  PetscBool          jac  = PETSC_FALSE;
  PetscReal          emin = 0, emax = 0;
  hipxMat            dA;
  HipxMat            M;
  HipxPC             hpc;
  HipxKSP            k;
  const PetscScalar *db;
  PetscScalar       *dx;
  void              *tb, *tx;

  PetscFunctionBegin;
  if (!KSPChebyshevHIPXApplicable(ksp, &Amat, &jac, &emin, &emax)) {
    PetscErrorCode (*psolve)(KSP) = NULL;
    PetscCall(PetscInfo(ksp, "KSPCHEBYSHEVHIPX: configuration outside the fused path, running the reference KSPSolve_Chebyshev\n"));
    PetscCall(PetscObjectQueryFunction((PetscObject)ksp, "KSPChebyshevHIPXParentSolve_C", &psolve));
    PetscCheck(psolve, PetscObjectComm((PetscObject)ksp), PETSC_ERR_ORDER, "KSPSetUp() has not run");
    PetscCall((*psolve)(ksp));
    PetscFunctionReturn(PETSC_SUCCESS);
  }
  PetscCall(MatSeqAIJHIPXGetDeviceMat(Amat, &dA));
  M.m = (hipx_int)Amat->rmap->n; M.A = dA; M.B = NULL; M.halo = NULL; M.lvec = NULL; M.nranks = 1;
  HipxPCSetDefaults(&hpc);
  hpc.type = jac ? HIPX_PC_JACOBI : HIPX_PC_NONE;
  PetscCallHIPX(HipxPCSetUp(&hpc, &M));
  HipxKSPSetDefaults(&k);
  k.normtype      = HIPX_KSP_NORM_NONE;
  k.max_it        = (hipx_int)ksp->max_it;
  k.guess_nonzero = ksp->guess_zero ? 0 : 1;
  k.fused         = 1;
  PetscCall(VecHIPXGetDeviceRead(ksp->vec_rhs, &db, &tb));
  PetscCall(VecHIPXGetDeviceReadWrite(ksp->vec_sol, &dx, &tx));
  PetscCallHIPX(HipxKSPSolve_Chebyshev(&k, &M, &hpc, db, dx, (double)emin, (double)emax));
  ksp->its    = (PetscInt)k.its;
  ksp->reason = (KSPConvergedReason)k.reason;
  PetscCallHIPX(HipxPCDestroy(&hpc));
  PetscCall(VecHIPXRestoreDeviceWrite(ksp->vec_sol, &dx, &tx));
  PetscCall(VecHIPXRestoreDeviceRead(ksp->vec_rhs, &db, &tb));
  PetscCall(PetscObjectStateIncrease((PetscObject)ksp->vec_sol));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* the parent's set-up picks the solve routine of the polynomial kind (cheby.c:757-766): keep that one as the fall-back of THIS object, put ours in front */
static PetscErrorCode KSPSetUp_ChebyshevHIPX(KSP ksp)
{
  PetscFunctionBegin;
  PetscCall((*parent_setup_cheby)(ksp));
  if (ksp->ops->solve != KSPSolve_ChebyshevHIPX) {
    PetscCall(PetscObjectComposeFunction((PetscObject)ksp, "KSPChebyshevHIPXParentSolve_C", ksp->ops->solve));
    ksp->ops->solve = KSPSolve_ChebyshevHIPX;
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

PETSC_EXTERN PetscErrorCode KSPCreate_Chebyshev(KSP); /* cheby.c:908: exported by libpetsc */

PetscErrorCode KSPCreate_ChebyshevHIPX(KSP ksp)
{
  PetscFunctionBegin;
  PetscCall(VecHIPXInitRuntime());
  PetscCall(KSPCreate_Chebyshev(ksp));
  if (!parent_setup_cheby) parent_setup_cheby = ksp->ops->setup;
  ksp->ops->setup = KSPSetUp_ChebyshevHIPX;
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* ---- KSPPIPECGHIPX ("pipecghipx", round 6): KSPPIPECG (pipecg.c:20-160) whose solve is the host layer's launch-ahead loop -- ONE fused update kernel (the eight
   vector updates, m = B w, the three sums of the next iteration; alpha and beta formed on the device) + one product per iteration, the iteration's single
   reduction (one 24-byte all-reduce on several ranks) started before the product and collected after it: PetscCommSplitReductionBegin ... End (comb.c:168-290) on
   the device.  Taken when nothing outside looks between iterations (no monitors, KSPConvergedDefault with its default context) on the hot path's operator
   (MATSEQAIJHIPX / MATMPIAIJHIPX with the device ghost exchange, PCJACOBI / PCNONE, hipx vectors); everything else runs the reference's KSPSolve_PIPECG --
   over the hipx types, where the lazy queue of vechipx.c turns its update block into one batch kernel. */
static PetscErrorCode (*parent_solve_pipecg)(KSP) = NULL;

static PetscBool KSPPIPECGHIPXApplicable(KSP ksp, Mat *Aout, PetscBool *none)
{
  Mat          Amat, Pmat;
  PetscBool    isjac = PETSC_FALSE, isnone = PETSC_FALSE, useabs = PETSC_FALSE, fixdiag = PETSC_TRUE;
  PCJacobiType jt;
  PetscMPIInt  size;

  if (ksp->calc_sings || ksp->pc_side != PC_LEFT || ksp->transpose_solve || ksp->dscale || ksp->numbermonitors || ksp->chknorm >= 0 || ksp->lagnorm) return PETSC_FALSE;
  if (ksp->converged != KSPConvergedDefault || !ksp->cnvP || getenv("HIPX_HOSTLOOP_PARENT")) return PETSC_FALSE;
  {
    KSPConvergedDefaultCtx *cctx = (KSPConvergedDefaultCtx *)ksp->cnvP;
    if (cctx->initialrtol || cctx->mininitialrtol || cctx->convmaxits) return PETSC_FALSE;
  }
  if (MPI_Comm_size(PetscObjectComm((PetscObject)ksp), &size)) return PETSC_FALSE;
  if (PCGetOperators(ksp->pc, &Amat, &Pmat) || Amat != Pmat || Amat->rmap->n != Amat->cmap->n) return PETSC_FALSE;
  if (size == 1) {
    if (!MatIsSeqAIJHIPX(Amat) || Amat->rmap->n == 0) return PETSC_FALSE;
  } else {
    PetscBool ismpi = PETSC_FALSE;
    hipxMat   dA, dB;
    hipxHalo  halo = NULL;
    Vec       lvec;
    if (PetscObjectTypeCompare((PetscObject)Amat, MATMPIAIJHIPX, &ismpi) || !ismpi) return PETSC_FALSE;
    if (MatMPIAIJHIPXGetDevice(Amat, &dA, &dB, &halo, &lvec) || !halo) return PETSC_FALSE;
  }
  if (PetscObjectTypeCompare((PetscObject)ksp->pc, PCJACOBI, &isjac) || PetscObjectTypeCompare((PetscObject)ksp->pc, PCNONE, &isnone) || (!isjac && !isnone)) return PETSC_FALSE;
  if (isjac) {
    if (PCJacobiGetType(ksp->pc, &jt) || jt != PC_JACOBI_DIAGONAL) return PETSC_FALSE;
    if (PCJacobiGetUseAbs(ksp->pc, &useabs) || useabs) return PETSC_FALSE;
    if (PCJacobiGetFixDiagonal(ksp->pc, &fixdiag) || !fixdiag) return PETSC_FALSE;
  }
  if (!VecIsHIPX(ksp->vec_rhs) || !VecIsHIPX(ksp->vec_sol)) return PETSC_FALSE;
  {
    MatNullSpace nsp = NULL;
    if (MatGetNullSpace(Amat, &nsp) || nsp) return PETSC_FALSE;
  }
  *Aout = Amat;
  *none = isnone;
  return PETSC_TRUE;
}

typedef int (*HipxHostSolve)(HipxKSP *, HipxMat *, HipxPC *, const double *, double *);
static PetscErrorCode KSPSolve_HostLoopHIPX(KSP ksp, PetscErrorCode (*parent)(KSP), HipxHostSolve solver, const char *what)
{
  Mat                Amat = NULL;
  PetscBool          isnone = PETSC_FALSE;
  hipxMat            dA;
  HipxMat            M;
  HipxPC             hpc;
  HipxKSP            k;
  const PetscScalar *db;
  PetscScalar       *dx, *dlv = NULL;
  void              *tb, *tx, *tlv = NULL;
  PetscMPIInt        size;
  Vec                lvecv = NULL;
  double            *hist = NULL;
  hipx_int           hl;
  int                herr = 0;
  char               herrmsg[512] = "";

  PetscFunctionBegin;
  if (!KSPPIPECGHIPXApplicable(ksp, &Amat, &isnone)) {
    PetscCall(PetscInfo(ksp, "%s: configuration outside the fused path, running the reference's loop\n", what));
    PetscCall((*parent)(ksp));
    PetscFunctionReturn(PETSC_SUCCESS);
  }
  PetscCallMPI(MPI_Comm_size(PetscObjectComm((PetscObject)ksp), &size));
  if (size == 1) {
    PetscCall(MatSeqAIJHIPXGetDeviceMat(Amat, &dA));
    M.m = (hipx_int)Amat->rmap->n; M.A = dA; M.B = NULL; M.halo = NULL; M.lvec = NULL; M.nranks = 1;
  } else {
    hipxMat  dB;
    hipxHalo halo;
    PetscCall(MatMPIAIJHIPXGetDevice(Amat, &dA, &dB, &halo, &lvecv));
    PetscCall(VecHIPXGetDeviceWrite(lvecv, &dlv, &tlv));
    M.m = (hipx_int)Amat->rmap->n; M.A = dA; M.B = dB; M.halo = halo; M.lvec = dlv; M.nranks = (int)size;
  }
  HipxPCSetDefaults(&hpc);
  hpc.type = isnone ? HIPX_PC_NONE : HIPX_PC_JACOBI;
  PetscCallHIPX(HipxPCSetUp(&hpc, &M));
  HipxKSPSetDefaults(&k);
  k.normtype = ksp->normtype == KSP_NORM_PRECONDITIONED ? HIPX_KSP_NORM_PRECONDITIONED : ksp->normtype == KSP_NORM_UNPRECONDITIONED ? HIPX_KSP_NORM_UNPRECONDITIONED : ksp->normtype == KSP_NORM_NATURAL ? HIPX_KSP_NORM_NATURAL : HIPX_KSP_NORM_NONE;
  k.max_it        = (hipx_int)ksp->max_it;
  k.min_it        = (hipx_int)ksp->min_it;
  k.guess_nonzero = ksp->guess_zero ? 0 : 1;
  k.rtol          = ksp->rtol;
  k.abstol        = ksp->abstol;
  k.divtol        = ksp->divtol;
  hl              = (hipx_int)((ksp->max_it < 1000000 ? ksp->max_it : 1000000) + 3);
  PetscCall(PetscMalloc1((size_t)hl, &hist));
  k.history  = hist;
  k.hist_len = hl;
  PetscCall(VecHIPXGetDeviceRead(ksp->vec_rhs, &db, &tb));
  PetscCall(VecHIPXGetDeviceReadWrite(ksp->vec_sol, &dx, &tx));
  ksp->its = 0;
  herr     = (*solver)(&k, &M, &hpc, db, dx);
  if (herr) PetscCall(PetscStrncpy(herrmsg, hipxGetErrorString(), sizeof(herrmsg)));
  else {
    for (hipx_int e = 0; e < k.hist_n && e < hl; e++) PetscCall(KSPLogResidualHistory(ksp, hist[e]));
    ksp->its    = (PetscInt)k.its;
    ksp->rnorm  = k.rnorm;
    ksp->rnorm0 = k.rnorm0;
    ksp->ttol   = k.ttol;
    ksp->reason = (KSPConvergedReason)k.reason;
    if (ksp->reason == KSP_DIVERGED_NANORINF) { /* (as in KSPSolve_CGHIPX: iterativ.c:1548-1559) */
      PCFailedReason pcreason;
      PetscCall(PCReduceFailedReason(ksp->pc));
      PetscCall(PCGetFailedReason(ksp->pc, &pcreason));
      if (pcreason) ksp->reason = KSP_DIVERGED_PC_FAILED;
    }
  }
  PetscCall(PetscFree(hist));
  (void)HipxKSPDestroyWork(&k);
  (void)HipxPCDestroy(&hpc);
  if (lvecv) PetscCall(VecHIPXRestoreDeviceWrite(lvecv, &dlv, &tlv));
  PetscCall(VecHIPXRestoreDeviceWrite(ksp->vec_sol, &dx, &tx));
  PetscCall(VecHIPXRestoreDeviceRead(ksp->vec_rhs, &db, &tb));
  PetscCall(PetscObjectStateIncrease((PetscObject)ksp->vec_sol));
  PetscCheck(herr != HIPX_ERR_SUP, PetscObjectComm((PetscObject)ksp), PETSC_ERR_SUP, "libhipx: %s", herrmsg);
  PetscCheck(!herr, PetscObjectComm((PetscObject)ksp), PETSC_ERR_GPU, "libhipx: %s", herrmsg);
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode KSPSolve_PIPECGHIPX(KSP ksp)
{
  PetscFunctionBegin;
  PetscCall(KSPSolve_HostLoopHIPX(ksp, parent_solve_pipecg, HipxKSPSolve_PIPECG, "KSPPIPECGHIPX"));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* ---- KSPGROPPCGHIPX ("groppcghipx", round 6): KSPGROPPCG (groppcg.c:23-140) on the host layer's launch-ahead loop -- two fused passes (direction + t; update + dp,
   gammaNew with S = B s re-formed per element) and one product per iteration, alpha / beta on the device, reduction 2 hidden behind the product -- under the same
   conditions as pipecghipx; the reference's KSPSolve_GROPPCG (its update blocks as batch kernels of the lazy queue) otherwise. */
static PetscErrorCode (*parent_solve_groppcg)(KSP) = NULL;
static PetscErrorCode KSPSolve_GROPPCGHIPX(KSP ksp)
{
  PetscFunctionBegin;
  PetscCall(KSPSolve_HostLoopHIPX(ksp, parent_solve_groppcg, HipxKSPSolve_GROPPCG, "KSPGROPPCGHIPX"));
  PetscFunctionReturn(PETSC_SUCCESS);
}

PETSC_EXTERN PetscErrorCode KSPCreate_GROPPCG(KSP); /* groppcg.c:166: exported by libpetsc */

PetscErrorCode KSPCreate_GROPPCGHIPX(KSP ksp)
{
  PetscFunctionBegin;
  PetscCall(VecHIPXInitRuntime());
  PetscCall(KSPCreate_GROPPCG(ksp));
  if (!parent_solve_groppcg) parent_solve_groppcg = ksp->ops->solve;
  ksp->ops->solve = KSPSolve_GROPPCGHIPX;
  PetscFunctionReturn(PETSC_SUCCESS);
}

PETSC_EXTERN PetscErrorCode KSPCreate_PIPECG(KSP); /* pipecg.c:179: exported by libpetsc */

PetscErrorCode KSPCreate_PIPECGHIPX(KSP ksp)
{
  PetscFunctionBegin;
  PetscCall(VecHIPXInitRuntime());
  PetscCall(KSPCreate_PIPECG(ksp));
  if (!parent_solve_pipecg) parent_solve_pipecg = ksp->ops->solve;
  ksp->ops->solve = KSPSolve_PIPECGHIPX;
  PetscFunctionReturn(PETSC_SUCCESS);
}

PETSC_EXTERN PetscErrorCode KSPCreate_CG(KSP); /* cg.c:686: exported by libpetsc, declared in no header */

PetscErrorCode KSPCreate_CGHIPX(KSP ksp)
{
  PetscFunctionBegin;
  PetscCall(VecHIPXInitRuntime());
  PetscCall(KSPCreate_CG(ksp)); /* PETSC_EXTERN, cg.c:686 */
  if (!parent_solve_cg) parent_solve_cg = ksp->ops->solve;
  ksp->ops->solve = KSPSolve_CGHIPX;
  PetscFunctionReturn(PETSC_SUCCESS);
}
