/*
 * hipx_ksp.h -- C host layer above the kernel ABI (hipx.h): the Krylov callers of the hot path,
 * restated in C so that the path can be driven without libpetsc (bench.py, tests, torchrun ranks).
 *
 * It mirrors the reference's caller code, statement by statement, over device vectors:
 *   KSPSolve_CG      src/ksp/ksp/impls/cg/cg.c:119-352
 *   KSPSolve_GMRES   src/ksp/ksp/impls/gmres/gmres.c:88-238,298-395 + borthog2.c:35-113
 *   KSPSolve_GROPPCG src/ksp/ksp/impls/cg/groppcg/groppcg.c:23-140
 *   KSPSolve_PIPECG  src/ksp/ksp/impls/cg/pipecg/pipecg.c:20-160 (+ the split-phase reduction of src/vec/vec/utils/comb.c:168-379)
 *   KSPConvergedDefault  src/ksp/ksp/interface/iterativ.c:1490-1585
 *   PCApply_Jacobi / PCApply_SOR / PCApply_None   jacobi.c:354, sor.c:27, pcnone
 *   MatMult_SeqAIJ / MatMult_MPIAIJ   aij.c:1444, mpiaij.c:1047
 *   VecDot_MPI / VecNorm_MPI reductions   pvecimpl.h:97-175 (local kernel + all-reduce)
 * When PETSc itself is the host (libpetschipx.so plugin), none of this is used: the reference's own
 * KSPSolve_CG/GMRES call the same kernels through the Vec/Mat ops tables.
 */
#ifndef HIPX_KSP_H
#define HIPX_KSP_H
#include "hipx.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { HIPX_PC_NONE = 0, HIPX_PC_JACOBI = 1, HIPX_PC_SOR = 2 };
enum { HIPX_KSP_NORM_NONE = 0, HIPX_KSP_NORM_PRECONDITIONED = 1, HIPX_KSP_NORM_UNPRECONDITIONED = 2, HIPX_KSP_NORM_NATURAL = 3 };

/* Mat: MATSEQAIJHIPX (A only) or MATMPIAIJHIPX (diag block A, off-diag block B, halo, lvec) */
typedef struct {
  hipx_int m;        /* local rows */
  hipxMat  A;        /* seq matrix or diagonal block */
  hipxMat  B;        /* off-diagonal block (compressed rows), NULL when sequential */
  hipxHalo halo;     /* ghost exchange plan, NULL when sequential */
  double  *lvec;     /* device ghost values, length = ghost count */
  int      nranks;   /* communicator size (1 = no reductions across ranks) */
} HipxMat;

typedef struct {
  int      type;
  double  *dinv;     /* PCJACOBI: device inverse diagonal (jacobi.c:205-266) */
  int      sor_flag; /* MatSORType, default SOR_LOCAL_SYMMETRIC_SWEEP (sor.c:442-446) */
  double   sor_omega, sor_shift;
  hipx_int sor_its, sor_lits;
  int      dconst_valid; /* PCJACOBI: every entry of dinv is the same double (constant-coefficient operator) ... */
  double   dconst;       /* ... namely this one: the fused update multiplies by it instead of streaming dinv */
} HipxPC;

typedef struct {
  /* parameters (defaults: itfunc.c KSPCreate, cg.c:696, gmres.c:906-915) */
  int      normtype;
  double   rtol, abstol, divtol;
  hipx_int max_it, min_it;
  hipx_int gmres_restart;
  double   gmres_haptol;
  int      gmres_cgs_refine; /* 0 never, 1 if needed, 2 always */
  int      guess_nonzero;
  int      fused;            /* CG only: use the fused SpMV+dot / update+PC+dots kernels (same arithmetic); bit 1 (fused = 3): on several ranks keep the separate
                                direction and dot kernels around MatMult_MPIAIJ (the round-5 sequence, for A/B timing against hipxMatMultMPICGDirectionDotBegin) */
  /* results */
  hipx_int its;
  int      reason;
  double   rnorm, rnorm0, ttol;
  double  *history;          /* host array, hist_len entries (KSPSetResidualHistory) */
  hipx_int hist_len, hist_n;
  /* CG stepping state (HipxKSPCGBegin / HipxKSPCGStep) */
  double  *R, *Z, *P;        /* device work vectors (KSPSetWorkVecs(3), cg.c:80) */
  double   beta, betaold, dpi, a;
  hipx_int i;
  hipx_int work_n;
  int      x_pending;        /* fused CG: x += a_pending * P of the last iteration not applied yet (merged into the next AYPX pass) */
  double   a_pending;
  int      defer_flush;      /* 1: HipxKSPCGStep may return with the x update pending; the caller ends with HipxKSPCGFlush */
  int      external_test;    /* 1: the caller runs its own convergence test after every step (the PETSc plugin: ksp->converged) */
  int      pipeline;         /* fused CG: enqueue iteration i+1 before the host has seen the sums of iteration i (default 1); 2: several ranks stay host-synchronised;
                                with single_reduction: 4 = the launch-ahead form of the single-reduction loop (round 5), anything else = host-synchronised */
  double  *dscal;            /* device scalars of the launch-ahead path: [0] p.w, [2+2q] z.z, [3+2q] z.r of the iterations of parity q */
  int      single_reduction; /* CG: KSPCGUseSingleReduction (cg.c:364-534): one reduction stage per iteration (three sums in one all-reduce), two more work vectors */
  double  *S, *W;            /* its work vectors S = A z and W (= A p by recurrence) */
  double   delta;            /* z . A z */
  double  *gslab;            /* GMRES: VEC_VV(0..restart+1) + VEC_TEMP + VEC_TEMP_MATOP in one slab, kept across solves as KSPSetUp_GMRES keeps its work vectors.
                                OWNERSHIP: the slab (several GB at BASELINE sizes) belongs to this HipxKSP from the first HipxKSPSolve_GMRES until
                                HipxKSPDestroyWork(ksp) -- a caller that drops the struct after a solve without that call leaks it */
  double   gslab_len;        /* its length in doubles (a double: the slab of a 512^3 problem exceeds 2^31 elements) */
  double  *P2;               /* second direction vector: hipxMatMultCGDirectionDotBegin (direction update as the product's prologue) writes p_new here
                                while other workgroups still read p; P and P2 swap roles after every fused launch (P is always the current direction) */
  double  *pipe_slab;        /* PIPECG (HipxKSPSolve_PIPECG): its nine work vectors r, u, w, z, q, p, s, m, n in one slab, kept across solves (freed by HipxKSPDestroyWork) */
  double   pipe_slab_len;    /* its length in doubles */
} HipxKSP;

/* sizeof of the three descriptor structs, so that a foreign-language mirror (petsc_amd/_lib.py) can verify its layout */
int  HipxStructSizes(int *mat, int *pc, int *ksp);
void HipxKSPSetDefaults(HipxKSP *ksp);
void HipxPCSetDefaults(HipxPC *pc);
int  HipxKSPDestroyWork(HipxKSP *ksp);
int  HipxKSPCGFlush(HipxKSP *ksp, HipxMat *A, double *x); /* applies a pending x += a p (defer_flush callers) */

int HipxMatMult(HipxMat *A, const double *x, double *y);                 /* MatMult_SeqAIJ | MatMult_MPIAIJ */
int HipxPCSetUp(HipxPC *pc, HipxMat *A);                                 /* PCSetUp_Jacobi | PCSetUp_SOR (nothing) */
int HipxPCApply(HipxPC *pc, HipxMat *A, const double *x, double *y);     /* PCApply */
int HipxPCDestroy(HipxPC *pc);
int HipxVecDot(HipxMat *A, const double *x, const double *y, hipx_int n, double *r);   /* VecTDot_Seq | _MPI */
int HipxVecNorm2(HipxMat *A, const double *x, hipx_int n, double *r);                   /* VecNorm_Seq | _MPI, NORM_2 */

int HipxKSPSolve_CG(HipxKSP *ksp, HipxMat *A, HipxPC *pc, const double *b, double *x);
int HipxKSPSolve_GMRES(HipxKSP *ksp, HipxMat *A, HipxPC *pc, const double *b, double *x);
/* replaces KSPSolve_PIPECG pipecg.c:20-160 (PCJACOBI / PCNONE; any norm type): one fused update kernel + one product per iteration, the scalars formed on the
   device, the iteration's single reduction (all-reduce on several ranks) hidden behind the product; ksp->pipeline = 0: host-synchronised (same bits) */
int HipxKSPSolve_PIPECG(HipxKSP *ksp, HipxMat *A, HipxPC *pc, const double *b, double *x);
/* replaces KSPSolve_GROPPCG groppcg.c:23-140 (PCJACOBI / PCNONE; any norm type): two fused passes + one product per iteration, scalars on the device, reduction 2
   (all-reduce on several ranks) hidden behind the product */
int HipxKSPSolve_GROPPCG(HipxKSP *ksp, HipxMat *A, HipxPC *pc, const double *b, double *x);
/* replaces KSPSolve_Chebyshev_FirstKind cheby.c:389-555 with given eigenvalue bounds (cheby.c:40-62); ksp->normtype NONE + PCJACOBI / PCNONE
   + ksp->fused: SpMV + one fused elementwise kernel per iteration, no reductions */
int HipxKSPSolve_Chebyshev(HipxKSP *ksp, HipxMat *A, HipxPC *pc, const double *b, double *x, double emin, double emax);

/* split form of KSPSolve_CG for benchmarking exactly K iterations: Begin = cg.c:134-217 (set-up, first
   residual/preconditioned norm), Step(nsteps) = nsteps passes of the loop body cg.c:220-349 */
int HipxKSPCGBegin(HipxKSP *ksp, HipxMat *A, HipxPC *pc, const double *b, double *x);
int HipxKSPCGStep(HipxKSP *ksp, HipxMat *A, HipxPC *pc, const double *b, double *x, hipx_int nsteps);

/* ---- MATMPIAIJHIPX host set-up (petsc_amd/host/hipx_mpiaij.c) ------------------------------------ */
typedef struct {
  hipx_int  m, nghost, nrows_c;
  hipx_int *Ai, *Aj; /* diagonal block, local column ids */
  double   *Aa;
  hipx_int *Bi, *Bj; /* off-diagonal block in compressed-row form: Bi[nrows_c+1], Bj = positions in garray */
  double   *Ba;
  hipx_int *ridx;    /* Mat_CompressedRow.rindex (matimpl.h:425-430) */
  hipx_int *garray;  /* sorted global ghost columns (mmaij.c:51,119) */
} HipxMPIAIJSplit;
/* MatSetValues_MPIAIJ diag/off-diag split (mpiaij.c:560-640) + MatSetUpMultiply_MPIAIJ (mmaij.c:8-125) */
int      HipxMatSetUpMultiply_MPIAIJ(hipx_int m, hipx_int cstart, hipx_int cend, const hipx_int *ai, const hipx_int *aj, const double *aa, HipxMPIAIJSplit *s);
void     HipxMPIAIJSplitFree(HipxMPIAIJSplit *s);
hipx_int HipxMPIAIJSplitSize(void);
/* receive side of the ghost plan: garray grouped by owning rank (contiguous because sorted) */
int  HipxHaloRecvPlan(hipx_int nghost, const hipx_int *garray, int nranks, const hipx_int *ranges, int *nrecv, int *recv_ranks, hipx_int *recv_off);
/* PetscSplitOwnership (src/sys/utils/psplit.c) */
void HipxSplitOwnership(hipx_int N, int size, hipx_int *ranges);

/* ---- driver assembly (petsc_amd/host/hipx_drivers.c): ex2.c:70-94, 3-D 7-point analogue, bench_kspsolve.c:115-303 */
int64_t HipxAssemble_ex2(hipx_int m, hipx_int n, hipx_int rstart, hipx_int rend, hipx_int *ai, hipx_int *aj, double *aa);
int64_t HipxAssemble_poisson7(hipx_int n, hipx_int rstart, hipx_int rend, hipx_int *ai, hipx_int *aj, double *aa);
int64_t HipxAssemble_poisson7_64(hipx_int n, hipx_int rstart, hipx_int rend, int64_t *ai, hipx_int *aj, double *aa);
/* nx x ny x nz box (x fastest); exactly one of ai (32-bit offsets) / ai64 may be non-NULL, both NULL = count only */
int64_t HipxAssemble_poisson7_box(hipx_int nx, hipx_int ny, hipx_int nz, hipx_int rstart, hipx_int rend, hipx_int *ai, int64_t *ai64, hipx_int *aj, double *aa);
int64_t HipxAssemble_bench27(hipx_int n, hipx_int rstart, hipx_int rend, hipx_int *ai, hipx_int *aj, double *aa);
int64_t HipxAssemble_bench27_64(hipx_int n, hipx_int rstart, hipx_int rend, int64_t *ai, hipx_int *aj, double *aa); /* 64-bit row offsets (512^3: 3.6e9 nonzeros) */

#ifdef __cplusplus
}
#endif
#endif
