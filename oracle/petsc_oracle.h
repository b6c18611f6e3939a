/*
 * petsc_oracle.h -- CPU restatement of the reference (PETSc 3.25) KSP hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so, and only as the
 * checker.  The shipped path (libhipx.so + libhipxksp.so + libpetschipx.so) never links it.
 *
 * Every function restates one reference routine in plain C (sequential loops, no BLAS, no FMA
 * contraction: build with -O2 -ffp-contract=off, the arithmetic the reference gets from gcc -O2 on
 * baseline x86-64).  Each cites the reference file:line it follows (paths relative to the PETSc
 * tree).  Parity pinning: see oracle/README.md (golden outputs of the reference's own tests and
 * outputs of the reference library built into oracle/_ref).
 *
 * Note on third-party arithmetic: the reference sends dot/nrm2/asum/axpy/scal to the system BLAS
 * (src/vec/vec/impls/seq/bvec1.c:27,84; bvec2.c:202-204), which is not vendored and not pinned
 * (any BLAS/LAPACK; MKL in the survey build).  The oracle restates the published BLAS level-1
 * definition (ddot = sum_i x_i*y_i accumulated left to right, nrm2 via sqrt(ddot(x,x)) exactly as
 * bvec2.c:204 calls it).  Reductions therefore agree with the reference to rounding, not bitwise.
 */
#ifndef PETSC_ORACLE_H
#define PETSC_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int    OInt;    /* PetscInt, 32-bit default (config/PETSc/options/indexTypes.py:19) */
typedef double OScalar; /* PetscScalar = PetscReal = double */

/* MatSORType bits, include/petscmat.h:1664-1671 */
enum {
  ORC_SOR_FORWARD_SWEEP         = 1,
  ORC_SOR_BACKWARD_SWEEP        = 2,
  ORC_SOR_SYMMETRIC_SWEEP       = 3,
  ORC_SOR_LOCAL_FORWARD_SWEEP   = 4,
  ORC_SOR_LOCAL_BACKWARD_SWEEP  = 8,
  ORC_SOR_LOCAL_SYMMETRIC_SWEEP = 12,
  ORC_SOR_ZERO_INITIAL_GUESS    = 16,
  ORC_SOR_EISENSTAT             = 32,
  ORC_SOR_APPLY_UPPER           = 64,
  ORC_SOR_APPLY_LOWER           = 128
};

/* NormType, include/petscvec.h */
enum { ORC_NORM_1 = 0, ORC_NORM_2 = 1, ORC_NORM_FROBENIUS = 2, ORC_NORM_INFINITY = 3, ORC_NORM_1_AND_2 = 4 };

/* KSPNormType (include/petscksp.h) */
enum { ORC_KSP_NORM_NONE = 0, ORC_KSP_NORM_PRECONDITIONED = 1, ORC_KSP_NORM_UNPRECONDITIONED = 2, ORC_KSP_NORM_NATURAL = 3 };

/* KSPConvergedReason values used on this path (include/petscksp.h) */
enum {
  ORC_KSP_CONVERGED_ITERATING      = 0,
  ORC_KSP_CONVERGED_RTOL           = 2,
  ORC_KSP_CONVERGED_ATOL           = 3,
  ORC_KSP_CONVERGED_ITS            = 4,
  ORC_KSP_CONVERGED_HAPPY_BREAKDOWN = 8,
  ORC_KSP_DIVERGED_NULL            = -2,
  ORC_KSP_DIVERGED_ITS             = -3,
  ORC_KSP_DIVERGED_DTOL            = -4,
  ORC_KSP_DIVERGED_BREAKDOWN       = -5,
  ORC_KSP_DIVERGED_INDEFINITE_PC   = -8,
  ORC_KSP_DIVERGED_NANORINF        = -9,
  ORC_KSP_DIVERGED_INDEFINITE_MAT  = -10
};

enum { ORC_PC_NONE = 0, ORC_PC_JACOBI = 1, ORC_PC_SOR = 2 };

/* ---- synthetic operators (CSR, 0-based, columns sorted, global column ids) ------------------ */
/* rows [rstart,rend) of the operator; pass ai==NULL to only count: returns nnz of the slab.    */
int64_t orc_laplace2d_5pt(OInt m, OInt n, OInt rstart, OInt rend, OInt *ai, OInt *aj, OScalar *aa);   /* ex2.c:70-94 */
int64_t orc_poisson3d_7pt(OInt n, OInt rstart, OInt rend, OInt *ai, OInt *aj, OScalar *aa);           /* 3-D analogue, SURVEY 8(d) */
int64_t orc_poisson3d_7pt_box(OInt nx, OInt ny, OInt nz, OInt rstart, OInt rend, OInt *ai, OInt *aj, OScalar *aa); /* nx x ny x nz box */
int64_t orc_poisson3d_27pt(OInt n, OInt rstart, OInt rend, OInt *ai, OInt *aj, OScalar *aa);          /* bench_kspsolve.c:115-303 */

/* ---- Mat_SeqAIJ kernels ---------------------------------------------------------------------- */
void orc_MatMult_SeqAIJ(OInt m, const OInt *ai, const OInt *aj, const OScalar *aa, const OScalar *x, OScalar *y);                       /* aij.c:1444-1502 */
void orc_MatMult_SeqAIJ_cprow(OInt m, OInt nrows, const OInt *ci, const OInt *ridx, const OInt *aj, const OScalar *aa, const OScalar *x, OScalar *y); /* aij.c:1467-1481 */
void orc_MatMultAdd_SeqAIJ(OInt m, const OInt *ai, const OInt *aj, const OScalar *aa, const OScalar *x, const OScalar *y, OScalar *z);  /* aij.c:1606-1658 */
int  orc_MatGetDiagonalMarkers_SeqAIJ(OInt m, const OInt *ai, const OInt *aj, OInt *diag);                                              /* matimpl.h:1835-1890 */
void orc_MatGetDiagonal_SeqAIJ(OInt m, const OInt *ai, const OInt *aj, const OScalar *aa, OScalar *v);                                  /* aij.c:1347-1380 */
int  orc_MatSOR_SeqAIJ(OInt m, const OInt *ai, const OInt *aj, const OScalar *aa, const OScalar *b, OScalar omega, int flag, OScalar fshift,
                       OInt its, OInt lits, OScalar *x);                                                                                /* aij.c:1797-2007 */
/* inodes: what MATSEQAIJ does with runs of rows that share their column list (blocked FEM matrices) */
void orc_MatMult_SeqAIJ_Inode(OInt m, const OInt *ai, const OInt *aj, const OScalar *aa, const OScalar *x, OScalar *y);                 /* inode.c:356-560 */
void orc_MatMultAdd_SeqAIJ_Inode(OInt m, const OInt *ai, const OInt *aj, const OScalar *aa, const OScalar *x, const OScalar *y, OScalar *z); /* inode.c:563-760 */
void orc_MatMult_SeqAIJ_dispatch(OInt m, const OInt *ai, const OInt *aj, const OScalar *aa, const OScalar *x, const OScalar *y, OScalar *z, int no_inode); /* aij.c:1459, 1617 */
OInt orc_MatSeqAIJCheckInode(OInt m, const OInt *ai, const OInt *aj, OInt limit, OInt *ns);                                             /* inode.c:3920-3985 */
int  orc_inode_invert_block(OScalar *a, int n);                                                                                         /* dgefa2.c:14 ... dgefa5.c:14 */
int  orc_MatSOR_SeqAIJ_Inode(OInt m, const OInt *ai, const OInt *aj, const OScalar *aa, OInt node_count, const OInt *ns, const OScalar *b, OScalar omega,
                             int flag, OScalar fshift, OInt its, OInt lits, OScalar *x);                                                /* inode.c:2420-3810 */
int  orc_MatSOR_SeqAIJ_dispatch(OInt m, const OInt *ai, const OInt *aj, const OScalar *aa, const OScalar *b, OScalar omega, int flag, OScalar fshift, OInt its,
                                OInt lits, OScalar *x, int no_inode);                                                                   /* aij.c:1852 */

/* ---- Mat_MPIAIJ set-up (integer work: must be bit-exact) ------------------------------------- */
/* Split rows [rstart,rend) (global columns) into diagonal block A (local cols) and off-diagonal
   block B (columns compacted to garray order).  Arrays sized by the caller: Ai/Bi m+1, Aj/Aa and
   Bj/Ba nnz, garray nnz.  Returns the ghost count ec.  mpiaij.c MatSetValues split + mmaij.c:8-125 */
OInt orc_MatSetUpMultiply_MPIAIJ(OInt m, OInt cstart, OInt cend, const OInt *ai, const OInt *aj, const OScalar *aa,
                                 OInt *Ai, OInt *Aj, OScalar *Aa, OInt *Bi, OInt *Bj, OScalar *Ba, OInt *garray);
/* compressed-row index of B (matimpl.h:425-430, MatCheckCompressedRow): rows with >=1 entry    */
OInt orc_MatCheckCompressedRow(OInt m, const OInt *bi, OInt *ci, OInt *ridx);

/* ---- Vec_Seq kernels -------------------------------------------------------------------------- */
OScalar orc_VecDot_Seq(OInt n, const OScalar *x, const OScalar *y);
/* 1: dot products / 2-norms evaluated as if in twice the working precision (Dot2, Ogita-Rump-Oishi 2005); 0: left to right in double */
void    orc_set_exact_reductions(int on);                               /* bvec1.c:10-49 (BLAS ddot) */
void    orc_VecMDot_Seq(OInt n, const OScalar *x, OInt nv, const OScalar *const *y, OScalar *z);  /* dvec2.c:83-... */
OScalar orc_VecNorm_Seq(OInt n, const OScalar *x, int type, OScalar *z2);                         /* bvec2.c:185-235 */
void    orc_VecAXPY_Seq(OInt n, OScalar *y, OScalar a, const OScalar *x);                         /* bvec1.c:70-89 */
void    orc_VecAYPX_Seq(OInt n, OScalar *y, OScalar b, const OScalar *x);                         /* dvec2.c:753-782 */
void    orc_VecAXPBY_Seq(OInt n, OScalar *y, OScalar a, OScalar b, const OScalar *x);             /* bvec1.c:91-118 */
void    orc_VecWAXPY_Seq(OInt n, OScalar *w, OScalar a, const OScalar *x, const OScalar *y);      /* dvec2.c:791-822 */
void    orc_VecAXPBYPCZ_Seq(OInt n, OScalar *z, OScalar a, OScalar b, OScalar c, const OScalar *x, const OScalar *y); /* bvec1.c:120-... */
void    orc_VecMAXPY_Seq(OInt n, OScalar *y, OInt nv, const OScalar *alpha, const OScalar *const *x); /* dvec2.c:515-590 */
void    orc_VecMAXPBY(OInt n, OScalar *y, OInt nv, const OScalar *alpha, OScalar beta, const OScalar *const *x); /* rvector.c:1394 */
void    orc_VecPointwiseMult_Seq(OInt n, OScalar *w, const OScalar *x, const OScalar *y);         /* bvec2.c:72-97 */
void    orc_VecPointwiseDivide_Seq(OInt n, OScalar *w, const OScalar *x, const OScalar *y);       /* bvec2.c:99-109 */
void    orc_VecReciprocal(OInt n, OScalar *x);                                                    /* vinv.c:1208-1229 */
void    orc_VecScale_Seq(OInt n, OScalar *x, OScalar a);                                          /* bvec2.c:167-183 */
void    orc_VecSet_Seq(OInt n, OScalar *x, OScalar a);                                            /* dvec2.c:642 */
void    orc_VecCopy_Seq(OInt n, const OScalar *x, OScalar *y);                                    /* bvec2.c:151 */

/* ---- PC ---------------------------------------------------------------------------------------- */
void orc_PCSetUp_Jacobi(OInt m, const OInt *ai, const OInt *aj, const OScalar *aa, OScalar *diag); /* jacobi.c:172-266 (DIAGONAL, fixdiag) */

/* ---- KSP ---------------------------------------------------------------------------------------- */
typedef struct {
  /* operator: either one sequential block (nranks==1) or a simulated MPIAIJ row partition */
  OInt           m;       /* local == global rows when nranks == 1 */
  const OInt    *ai, *aj; /* CSR of the whole operator, global columns */
  const OScalar *aa;
  int            nranks;  /* >1: PCSOR / PCJACOBI act per row slab like MatSOR_MPIAIJ (mpiaij.c:1394-1486) */
  const OInt    *ranges;  /* nranks+1 ownership ranges (PetscSplitOwnership) */
  /* solver parameters (defaults = Appendix B of SURVEY.md) */
  int     pc_type;
  int     sor_flag;  /* MatSORType for PCSOR; default LOCAL_SYMMETRIC_SWEEP (sor.c:442-446) */
  OScalar sor_omega, sor_shift;
  OInt    sor_its, sor_lits;
  int     normtype;
  OScalar rtol, abstol, divtol;
  OInt    max_it, min_it;
  OInt    gmres_restart;
  OScalar gmres_haptol;
  int     gmres_cgs_refine; /* 0 never (default), 2 always (ex2 goldens use refine_always) */
  int     guess_nonzero;
  /* outputs */
  OInt     its;
  int      reason;
  OScalar  rnorm;
  OScalar *history; /* caller array of length hist_len, receives rnorm per KSPLogResidualHistory */
  OInt     hist_len, hist_n;
  int      no_inode; /* 1 = -mat_no_inode: PCSOR takes the point routine on every matrix (default 0: aij.c:1852) */
  /* round 5: operators too large to hold as one CSR (27-pt 512^3: 3.6e9 nonzeros).  When mult_cb is set, y = A x is the caller's routine
     (oracle/stream_gmres.py: row slabs assembled on the fly and multiplied with orc_MatMult_SeqAIJ) and ai/aj/aa may be NULL; when pc_cb is set
     it is KSP_PCApply (there: one orc_MatSOR_SeqAIJ sweep per rank on its diagonal block, mpiaij.c:1408-1412).  The Krylov loops are unchanged. */
  void (*mult_cb)(void *user, const OScalar *x, OScalar *y);
  void (*pc_cb)(void *user, const OScalar *r, OScalar *z);
  void *user;
} OrcKSP;

void orc_KSPSetDefaults(OrcKSP *ksp);
int  orc_KSPSolve_CG(OrcKSP *ksp, const OScalar *b, OScalar *x);    /* cg.c:119-352 */
int  orc_KSPSolve_GMRES(OrcKSP *ksp, const OScalar *b, OScalar *x); /* gmres.c:88-238,298-420; borthog2.c:35-113 */
int  orc_KSPSolve_PIPECG(OrcKSP *ksp, const OScalar *b, OScalar *x);  /* pipecg.c:20-160 */
int  orc_KSPSolve_GROPPCG(OrcKSP *ksp, const OScalar *b, OScalar *x); /* groppcg.c:23-140 */
void orc_MatMult_MPIAIJ(OInt m, const OInt *ai, const OInt *aj, const OScalar *aa, int nranks, const OScalar *x, OScalar *y); /* mpiaij.c:1047-1061 on a simulated row partition */

/* PetscSplitOwnership (src/sys/utils/psplit.c): n_local = N/size + ((N % size) > rank) */
void orc_PetscSplitOwnership(OInt N, int size, OInt *ranges);

#ifdef __cplusplus
}
#endif
#endif
