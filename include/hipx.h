/*
 * hipx.h -- C ABI of libhipx.so: the MI355X (gfx950) kernels behind PETSc's Krylov inner loop.
 *
 * This is the drop-in boundary.  Plain C: raw device pointers, sizes, scalars; no PETSc types, no
 * torch types, no C++.  The PETSc-facing plugin (petsc_amd/plugin, libpetschipx.so) fills the
 * reference's ops tables (struct _VecOps include/petsc/private/vecimpl.h:18-110, struct _MatOps
 * include/petsc/private/matimpl.h:38-213, struct _PCOps include/petsc/private/pcimpl.h:11-31) with
 * thin C functions that call the entry points below; INTEGRATION.md shows that binding.
 * Each entry point cites the reference routine it replaces (paths relative to the PETSc tree).
 *
 * Conventions
 *   - one process drives one GPU (hipxInit(device)); all work is enqueued on the library's compute
 *     stream; calls that return a scalar to the host block until it is valid, all other calls may
 *     return with work enqueued (stream order keeps them correct) -- SURVEY.md 8(b) "Threading".
 *   - every function returns 0 (== PETSC_SUCCESS) or a nonzero code: HIPX_ERR_* below or
 *     HIPX_ERR_HIP_BASE + hipError_t.  hipxGetErrorString() describes the last failure.
 *   - PetscScalar = double, PetscInt = int32 (hipx_int); row offsets may be 64-bit on request.
 *   - arithmetic is IEEE fp64 without FMA contraction: elementwise kernels and the CSR row sums
 *     reproduce the reference's -O2 x86-64 results bit for bit; reductions use a fixed,
 *     run-to-run deterministic order (they cannot match a BLAS bitwise, see oracle/petsc_oracle.h).
 */
#ifndef HIPX_H
#define HIPX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t hipx_int;

#define HIPX_SUCCESS        0
#define HIPX_ERR_ARG        62   /* PETSC_ERR_ARG_WRONG      */
#define HIPX_ERR_MEM        55   /* PETSC_ERR_MEM            */
#define HIPX_ERR_SUP        56   /* PETSC_ERR_SUP            */
#define HIPX_ERR_ORDER      58   /* PETSC_ERR_ORDER (not initialised) */
#define HIPX_ERR_GPU        97   /* PETSC_ERR_GPU            */
#define HIPX_ERR_ZEROPIVOT  71   /* PETSC_ERR_MAT_LU_ZRPVT   */
#define HIPX_ERR_HIP_BASE   10000

/* ---- runtime ------------------------------------------------------------------------------- */
int         hipxInit(int device);            /* hipSetDevice + streams + reduction scratch; idempotent */
int         hipxFinalize(void);
int         hipxGetDeviceCount(int *count);  /* usable before hipxInit: one rank per GPU picks rank % count */
int         hipxIsInitialized(void);
const char *hipxGetErrorString(void);
int         hipxDeviceName(char *buf, size_t len);
int         hipxDeviceUID(unsigned long long *uid); /* hash of the current device's PCI bus id: equal across processes iff they drive the same GPU */
void       *hipxComputeStream(void);         /* hipStream_t, for callers that enqueue their own work */
void       *hipxCommStream(void);
int         hipxStreamSynchronize(void);     /* compute stream */
int         hipxDeviceSynchronize(void);

int hipxMalloc(void **dptr, size_t bytes);
int hipxFree(void *dptr);
int hipxMallocHost(void **hptr, size_t bytes); /* pinned */
int hipxFreeHost(void *hptr);
int hipxMemcpyHtoD(void *dst, const void *src, size_t bytes); /* blocking, ordered after the compute stream */
int hipxMemcpyDtoH(void *dst, const void *src, size_t bytes); /* blocking */
int hipxMemcpyDtoD(void *dst, const void *src, size_t bytes); /* async on the compute stream */
int hipxMemset(void *dst, int value, size_t bytes);           /* async on the compute stream */

/* timing helpers on the compute stream (HIP events; used by bench.py for roofline.achieved) */
int hipxEventCreate(void **ev);
int hipxEventDestroy(void *ev);
int hipxEventRecord(void *ev);
int hipxEventElapsedMs(void *start, void *stop, float *ms); /* synchronises on stop */

/* SpMV instrumentation for roofline.achieved: when enabled, every SpMV launch (hipxMatMult / MatMultDot /
   the diagonal-block product of hipxMatMultMPI) is bracketed by HIP events on the compute stream */
int hipxProfileSpMV(int enable);
int hipxProfileSpMVGet(int *count, double *total_ms); /* synchronises, returns and clears the tally */

/* the same for the other sections of the path (per-rank diagnosis of a multi-GPU run): events on the stream the section runs on */
#define HIPX_PROF_HALO      0 /* ghost exchange on the comm stream: pack/put + send/recv (hipxHaloBegin) */
#define HIPX_PROF_ALLREDUCE 1 /* scalar all-reduce on the compute stream (RCCL or IPC) */
#define HIPX_PROF_OFFDIAG   2 /* off-diagonal-block MatMultAdd incl. the wait for the ghost values (hipxMatMultMPI) */
#define HIPX_PROF_SOR       3 /* hipxMatSOR: one call = all its sweeps */
#define HIPX_PROF_CG_UPDATE 4 /* the fused CG update kernel (hipxCGFusedUpdate*: r -= a w, z, two sums) */
#define HIPX_PROF_CG_DIR    5 /* the CG direction kernel (hipxCGAypxAxpy*: p = z + b p, x += a p) when it is not the product's prologue */
#define HIPX_PROF_FOLD      6 /* fold of the SpMV epilogue's dot partials (hipxMatMultDot*, hipxMatMultCGDirectionDotBegin) */
#define HIPX_PROF_NSECTIONS 7
int hipxProfileSections(int enable);
int hipxProfileSectionGet(int id, int *count, double *total_ms); /* synchronises the device, returns and clears the tally */

/* ---- Vec BLAS-1 (device pointers; n = local length) ------------------------------------------ */
/* replaces VecSet_Seq dvec2.c:642 */                 int hipxVecSet(double *x, hipx_int n, double alpha);
/* replaces VecCopy_Seq bvec2.c:151 */                int hipxVecCopy(const double *x, double *y, hipx_int n);
/* replaces VecScale_Seq bvec2.c:167 */               int hipxVecScale(double *x, hipx_int n, double alpha);
/* replaces VecSwap_Seq bvec2.c (BLAS dswap) */       int hipxVecSwap(double *x, double *y, hipx_int n);
/* replaces VecAXPY_Seq bvec1.c:70 */                 int hipxVecAXPY(double *y, double alpha, const double *x, hipx_int n);
/* replaces VecAYPX_Seq dvec2.c:753 */                int hipxVecAYPX(double *y, double beta, const double *x, hipx_int n);
/* replaces VecAXPBY_Seq bvec1.c:91 */                int hipxVecAXPBY(double *y, double alpha, double beta, const double *x, hipx_int n);
/* replaces VecWAXPY_Seq dvec2.c:791 */               int hipxVecWAXPY(double *w, double alpha, const double *x, const double *y, hipx_int n);
/* replaces VecAXPBYPCZ_Seq bvec1.c:120 */            int hipxVecAXPBYPCZ(double *z, double alpha, double beta, double gamma, const double *x, const double *y, hipx_int n);
/* replaces VecPointwiseMult_Seq bvec2.c:72 */        int hipxVecPointwiseMult(double *w, const double *x, const double *y, hipx_int n);
/* replaces VecPointwiseDivide_Seq bvec2.c:99 */      int hipxVecPointwiseDivide(double *w, const double *x, const double *y, hipx_int n);
/* replaces VecReciprocal_Default vinv.c:1208 */      int hipxVecReciprocal(double *x, hipx_int n);
/* replaces VecAbs (vinv.c) */                        int hipxVecAbs(double *x, hipx_int n);
/* replaces VecShift (rvector.c) */                   int hipxVecShift(double *x, hipx_int n, double shift);
/* PCSetUp_Jacobi zero fix-up jacobi.c:255-266 */     int hipxVecReplaceZeros(double *x, hipx_int n, double value, hipx_int *nreplaced_host);
/* replaces VecMAXPY_Seq dvec2.c:658: y += sum_j alpha[j] x[j]; alpha on host, x = host array of nv device pointers */
int hipxVecMAXPY(double *y, hipx_int nv, const double *alpha, const double *const *x, hipx_int n);
/* replaces VecMAXPBY rvector.c:1394: y = beta y + sum_j alpha[j] x[j] */
int hipxVecMAXPBY(double *y, hipx_int nv, const double *alpha, double beta, const double *const *x, hipx_int n);

/* the vector work of one Chebyshev iteration (KSPSolve_Chebyshev_FirstKind cheby.c:475-511 with PCJACOBI or PCNONE, no norm) in one pass:
   r = b - Ap (VecAYPX, -1), z = r * dinv (PCApply_Jacobi; dinv == NULL: z = r), pnext = alpha pprev + beta pcur + gamma z (VecAXPBYPCZ_Seq
   bvec1.c:120-147, same association orders); r_out != NULL also stores r.  Element by element the operations of the three reference loops. */
int hipxVecChebyshevStep(double *pnext, double alpha, double beta, double gamma, const double *pprev, const double *pcur, const double *dinv, const double *b, const double *Ap, double *r_out,
                         hipx_int n);
/* indexed gather / scatter on the compute stream: dst[didx ? didx[k] : k] (= | +=) src[sidx ? sidx[k] : k], k < n; the index lists
   are DEVICE arrays.  mode 0 insert, 1 add (didx must then hold no duplicates).  Replaces the Pack / UnpackAndInsert / UnpackAndAdd
   loops of PetscSF (src/vec/is/sf/impls/basic/sfpack.c:706-790) for unit = one scalar. */
int hipxVecScatterIndexed(const double *src, const hipx_int *sidx, double *dst, const hipx_int *didx, hipx_int n, int mode);

/* reductions: blocking, result written to *host */
/* replaces VecDot_Seq/VecTDot_Seq bvec1.c:10-49 */   int hipxVecDot(const double *x, const double *y, hipx_int n, double *result);
/* replaces VecMDot_Seq/VecMTDot_Seq dvec2.c:83 */    int hipxVecMDot(const double *x, hipx_int nv, const double *const *y, hipx_int n, double *results);
/* replaces VecNorm_Seq bvec2.c:185; type: 0 NORM_1, 1 NORM_2, 2 FROBENIUS, 3 INFINITY, 4 NORM_1_AND_2 (results[2]).
   The returned value is the LOCAL norm (NORM_2 already square-rooted) as VecNorm_Seq returns it. */
int hipxVecNorm(const double *x, hipx_int n, int type, double *results);
/* replaces VecDotNorm2 (rvector.c): dp = x.y, nm = y.y in one pass */
int hipxVecDotNorm2(const double *x, const double *y, hipx_int n, double *dp, double *nm);
/* replaces VecSum / VecMax / VecMin (dvec2.c:592-640); idx may be NULL */
int hipxVecSum(const double *x, hipx_int n, double *result);
int hipxVecMax(const double *x, hipx_int n, hipx_int *idx, double *result);
int hipxVecMin(const double *x, hipx_int n, hipx_int *idx, double *result);

/* Reduction mode of every sum-reduction of this library (Dot/TDot/MDot/Norm 1,2/DotNorm2/Sum, the sums of the fused CG kernels, and
   the all-reduces of the multi-rank forms):
     HIPX_RED_FAST (default)  plain fp64 partial sums in a fixed order: deterministic, ~1e-15 relative from the exact value;
     HIPX_RED_EXACT           compensated sums (Dot2 / Sum2, Ogita-Rump-Oishi): every sum is carried as an unevaluated (hi, lo) pair
                              through the thread loop, the wave and workgroup folds, the fold of the workgroups' partials and the
                              fold over the ranks, and rounded ONCE: the result is the correctly rounded dot product (as if computed
                              in twice the working precision) whatever the grid shape and the rank count -- the value the reference
                              computes when its BLAS ddot/dnrm2/dasum/dgemv (bvec1.c:27, bvec2.c:202-223, dvec2.c:557) are exactly
                              rounded (oracle/exactblas.c).  Costs no extra HBM traffic; SpMV + dot fusions fall back to SpMV, then dot.
   The environment variable HIPX_REDUCTIONS=exact|fast sets the initial mode at hipxInit. */
#define HIPX_RED_FAST  0
#define HIPX_RED_EXACT 1
int hipxSetReductionMode(int mode);
int hipxGetReductionMode(int *mode);

/* split-phase reductions (enqueue now, read later): slot in [0, HIPX_MAX_RED_SLOTS) */
#define HIPX_MAX_RED_SLOTS 64
int hipxVecDotBegin(const double *x, const double *y, hipx_int n, int slot);
int hipxRedEnd(int slot, int nvals, double *results); /* synchronises the compute stream, copies nvals sums */
/* round 5.  w = x .* y (VecPointwiseMult_Seq bvec2.c:72-97; w may NOT alias x or y here) and, in the same pass, the sums w.w and w.x of the vector
   just written: what KSPSolve_CG asks for right after PCApply_Jacobi = VecPointwiseMult(z, r, diag) (jacobi.c:354-362) -- VecNorm(Z) cg.c:309 and
   VecXDot(Z, R) cg.c:344.  Enqueue only; hipxRedEnd(slot, 2, s) returns s[0] = w.w, s[1] = w.x.  The plugin's vector type keeps the two sums keyed on
   the vectors and answers those calls from them (plugin/vechipx.c, "reduction cache"): in the exact reduction mode the values are the ones hipxVecNorm /
   hipxVecDot return, bit for bit; in the fast mode they differ by the association of the partial sums (rounding). */
int hipxVecPointwiseMultDotsBegin(double *w, const double *x, const double *y, hipx_int n, int slot);
/* round 5.  y += alpha x, then w = y .* d with the sums w.w and w.y -- VecAXPY (bvec1.c:70-83) followed by VecPointwiseMult (bvec2.c:72-97): "r <- r - a w;
   z <- B r" of cg.c:306-307 with PCJACOBI -- in ONE pass over the vectors (the kernel of hipxCGFusedUpdate; element by element the operations of the two
   separate calls, the same bits).  Enqueue only; hipxRedEnd(slot, 2, s): s[0] = w.w, s[1] = w.y.  w may be x (KSPSolve_CG keeps A p in
   the vector it then overwrites with z, cg.c:145); otherwise the arrays must be distinct. */
int hipxVecAXPYPointwiseMultDotsBegin(double *y, double alpha, const double *x, double *w, const double *d, double dconst, hipx_int n, int slot); /* d == NULL: w = y * dconst (every entry of the diagonal is dconst: one stream less) */

/* fused CG kernels (same arithmetic as the separate calls, fewer HBM passes) */
/* x += a p ; r -= a w ; z = r .* d ; sums[0] = z.z ; sums[1] = z.r  (cg.c:305-309,344 with PCJACOBI) */
int hipxCGFusedUpdate(double *x, double *r, double *z, const double *p, const double *w, const double *d, double a, hipx_int n, double *sums2);
/* x == NULL in hipxCGFusedUpdate: the x update is deferred; it is then done by
   p = z + b p ; x += a p_old  (cg.c:249 of the next iteration + cg.c:305 of this one: p is read once) */
int hipxCGAypxAxpy(double *p, double b, const double *z, double *x, double a, hipx_int n);
/* Launch-ahead forms: the scalars are read from device memory (results of kernels queued earlier on the compute stream), so
   the next iteration can be enqueued before the host has seen them.  b = *dev_beta_new / *dev_beta_old (cg.c:248),
   a = *dev_beta / *dev_dpi (cg.c:288): the same IEEE quotients the host forms.  hipxRedEnd(slot, ...) collects the sums;
   dev_dot / dev_sums2 receive device copies of them. */
int hipxCGAypxAxpyDev(double *p, const double *z, const double *r, double dconst, double *x, const double *dev_beta_new, const double *dev_beta_old, const double *dev_dpi,
                      hipx_int n); /* z == NULL: z is re-formed as r * dconst (constant Jacobi diagonal, z never stored) */
/* host-scalar form of the same: p = r * dconst + b p ; x += a p_old (x == NULL: no x update) */
int hipxCGAypxAxpyR(double *p, double b, const double *r, double dconst, double *x, double a, hipx_int n);
int hipxCGFusedUpdateBegin(double *x, double *r, double *z, const double *p, const double *w, const double *d, double dconst, const double *dev_beta, const double *dev_dpi,
                           hipx_int n, int slot, double *dev_sums2); /* d == NULL: z = r * dconst (constant Jacobi diagonal), one vector pass less;
                                                                      then z == NULL as well: z is not stored (hipxCGAypxAxpyDev/R re-form it) */

/* ---- pipelined CG variants (round 6; csrc/hipx_pipe.hip) ------------------------------------
   The update blocks of KSPSolve_PIPECG (pipecg.c:132-150), KSPSolve_GROPPCG (groppcg.c:98-100,135-136) and KSPSolve_PIPECR (pipecr.c:101-117) are runs of
   VecAYPX / VecAXPY calls on up to ten vectors, followed by the sums of the next iteration (VecNormBegin / VecDotBegin, comb.c:338,379).
   hipxVecBatchAXPYDotsBegin runs such a run as ONE pass: nops operations in their order -- kind[k] 1: y += s x (VecAXPY_Seq bvec1.c:70-83), 2: y = x + s y
   (VecAYPX_Seq dvec2.c:753-780, general-beta loop; for s = 1, -1 the same bits as the reference's special cases) -- on the nvec DISTINCT device vectors
   vec[0 .. nvec), addressed by slot: slots are numbered by first appearance (for each operation y, then x).  Every operand is read once, every changed vector
   written once; element by element the operations and their order are those of the separate calls: the vectors are bit-identical.  The batch must be one of
   the compiled programs (the three loops above); otherwise nothing is enqueued and *ndots = -1 (the caller runs the calls one by one).  On success *ndots sums
   of the vectors after the batch, da[k] . db[k] (slots), are on their way to reduction slot `slot`: hipxRedEnd(slot, *ndots, sums). */
#define HIPX_BATCH_MAX_OPS  8
#define HIPX_BATCH_MAX_VECS 10
#define HIPX_BATCH_MAX_DOTS 4
int hipxVecBatchProgramKnown(int nops, const int *kind, const int *yslot, const int *xslot, int nvec); /* 1: hipxVecBatchAXPYDotsBegin would run this batch; no GPU work */
int hipxVecBatchAXPYDotsBegin(int nops, const int *kind, const int *yslot, const int *xslot, const double *s, int nvec, double *const *vec, hipx_int n, int slot, int *ndots, int *da,
                              int *db);
/* One iteration's vector work of KSPSolve_PIPECG in one pass, the scalars formed on the device (launch-ahead: HipxKSPSolve_PIPECG, hipx_ksp.h):
     first:     alpha = gamma / delta;  z = n; q = m; p = u; s = w                                                            pipecg.c:132-136
     otherwise: beta = gamma / gamma_old; alpha = gamma / (delta - beta / alpha_old * gamma);                                 pipecg.c:138-139
                x += alpha_old p  (the update the iteration before left behind: applied before p changes);  z = n + beta z; q = m + beta q; p = u + beta p; s = w + beta s
     then       u -= alpha q; w -= alpha z; r -= alpha s                                                                      pipecg.c:148-150
                m = w .* d  (PCJACOBI; d == NULL: m = w * dconst -- a constant diagonal; dconst == 1.0 = PCNONE: m is NOT written,
                the caller multiplies w itself)                                                                               pipecg.c:115
     sums of the new vectors -> slot and dev_sums_out[0..3): [0] u.u (normkind 1) | r.r (2) | 0 (0), [1] gamma = r.u, [2] delta = w.u   pipecg.c:106-113
   gamma, delta = dev_sums[1], [2]; gamma_old = dev_sums_old[1]; alpha is written to *dev_alpha_out (the host forms the same IEEE quotients for the final
   x += alpha p).  m of THIS iteration (q = m + beta q) is re-formed from w as the kernel before formed it.  v->n = A m of this iteration.  Enqueue only. */
typedef struct {
  double       *z, *q, *p, *s, *x, *u, *w, *r, *m;
  const double *n;
} hipxPipeCGVecs;
int hipxPipeCGUpdateBegin(const hipxPipeCGVecs *v, const double *d, double dconst, int normkind, int first, const double *dev_sums, const double *dev_sums_old, const double *dev_alpha_old,
                          double *dev_alpha_out, hipx_int n, int slot, double *dev_sums_out);

/* Gropp's CG (KSPSolve_GROPPCG groppcg.c:23-140) as two passes per iteration + the product, scalars formed on the device (HipxKSPSolve_GROPPCG, hipx_ksp.h).
   hipxGroppCGDirectionBegin, iteration i > 1:  x += alpha_{i-1} p  (the update the iteration before left behind: groppcg.c:98, applied before p changes);
     p = z + beta p; s = Z + beta s  with beta = *dev_gamma_new / *dev_gamma_old  (groppcg.c:132-136);  t = p . s -> slot and *dev_t_out  (groppcg.c:87).
   hipxGroppCGUpdateBegin:  alpha = *dev_gamma / *dev_t (groppcg.c:96), stored to *dev_alpha_out;  r -= alpha s;  z -= alpha (s .* d)  (S = B s of groppcg.c:92 re-formed per
     element: d == NULL -> s * dconst, dconst == 1.0 = PCNONE);  sums -> slot and dev_sums2_out: [0] z.z (normkind 1) | r.r (2) | 0, [1] gammaNew = r.z  (groppcg.c:103-108).
   Same operations per element, in the reference's order: vectors bit-identical.  Enqueue only; the ...Allreduce forms (below) reduce over the ranks on the stream. */
int hipxGroppCGDirectionBegin(double *p, double *s, double *x, const double *z, const double *Z, const double *dev_gamma_new, const double *dev_gamma_old, const double *dev_alpha_old, hipx_int n,
                              int slot, double *dev_t_out);
int hipxGroppCGUpdateBegin(double *r, double *z, const double *s, const double *d, double dconst, int normkind, const double *dev_gamma, const double *dev_t, double *dev_alpha_out, hipx_int n,
                           int slot, double *dev_sums2_out);

/* ---- Mat (CSR = Mat_SeqAIJ src/mat/impls/aij/seq/aij.h:47-78,150-168) ----------------------- */
typedef struct hipxMat_s *hipxMat;

/* Upload a host CSR matrix (m rows, n cols, 0-based, as MatAssemblyEnd_SeqAIJ leaves it; aij.c:1085).
   i has m+1 entries, j/a have i[m].  The triggering event in the plugin is MatAssemblyEnd. */
int hipxMatCreateCSR(hipx_int m, hipx_int n, const hipx_int *i, const hipx_int *j, const double *a, hipxMat *A);
int hipxMatCreateCSR64(hipx_int m, hipx_int n, const int64_t *i, const hipx_int *j, const double *a, hipxMat *A);
/* compressed-row variant (Mat_CompressedRow matimpl.h:425-430) for the MPIAIJ off-diagonal block:
   only rows ridx[0..nrows) hold entries; ci has nrows+1 offsets. */
int hipxMatCreateCSRCompressedRow(hipx_int m, hipx_int n, hipx_int nrows, const hipx_int *ci, const hipx_int *ridx, const hipx_int *j, const double *a, hipxMat *A);
int hipxMatUpdateValues(hipxMat A, const double *a);          /* same nonzero pattern, new values (host) */
/* builds NOW what the first product would build lazily (inode search, row / pattern templates, march plan, SELL / packed-column copies): the device half of
   MatAssemblyEnd_SeqAIJ's set-up work (aij.c:1085-1144: inode and compressed-row checks happen there too), so that the first MatMult is a product only */
int hipxMatSetUp(hipxMat A);
int hipxMatGetValues(hipxMat A, double *a_host);              /* the value array back to the host (CSR order) */
/* Value-only updates on the device copy (no host round trip).  Same arithmetic as the reference:
   MatScale_SeqAIJ aij.c:2604-2617 (a *= alpha), MatZeroEntries_SeqAIJ, MatDiagonalScale_SeqAIJ aij.c:2333-2371
   ((a * l_i) * r_j: left pass first; l, r device pointers of length m / n, either may be NULL). */
int hipxMatScale(hipxMat A, double alpha);
int hipxMatZeroEntries(hipxMat A);
/* replaces MatAXPY_SeqAIJ with SAME_NONZERO_PATTERN aij.c:2926-2945: Y.a += alpha X.a on the device copies (daxpy over the value
   arrays: product and sum rounded separately).  The caller guarantees identical patterns (sizes and nonzero counts are checked). */
int hipxMatAXPY(hipxMat Y, double alpha, hipxMat X);
/* one iteration of KSPSolve_Chebyshev_FirstKind cheby.c:475-511 (PCJACOBI: dinv, PCNONE: dinv == NULL; no norm): pnext = alpha pprev + beta pcur +
   gamma (dinv .* (b - A pcur)), the SpMV and hipxVecChebyshevStep in ONE kernel when the matrix takes the pair form of the template kernel
   (else the two kernels, the step in place); bit-identical to MatMult + VecAYPX + VecPointwiseMult + VecAXPBYPCZ */
int hipxMatMultChebyshev(hipxMat A, const double *pcur, double *pnext, double alpha, double beta, double gamma, const double *pprev, const double *dinv, const double *b);
int hipxMatDiagonalScale(hipxMat A, const double *l, const double *r);
/* COO assembly on the device.  replaces MatSetValuesCOO_SeqAIJ aij.c:4710-4733; jmap (nz + 1) / perm (ntot) are the maps
   MatSetPreallocationCOO_SeqAIJ leaves in MatCOOStruct_SeqAIJ (aij.c:4524-4707, aij.h:170-176): entry k of the CSR value array
   is the sum of v[perm[jmap[k] .. jmap[k+1])], added left to right.  hipxMatCreateCSR* accept a == NULL (pattern only). */
typedef struct hipxCOO_s *hipxCOO;
int hipxCOOCreate(int64_t nz, const int64_t *jmap, int64_t ntot, const int64_t *perm, hipxCOO *coo);
int hipxCOODestroy(hipxCOO *coo);
int hipxMatSetValuesCOO(hipxMat A, hipxCOO coo, const double *v, int64_t n, int v_on_device, int insert);
/* the remote part of MatSetValuesCOO_MPIAIJ mpiaij.c:6817-6822 (entries other ranks sent): maps with a target index per entry
   (Aimap2 / Ajmap2 / Aperm2 of MatCOOStruct_MPIAIJ, mpiaij.h:80-81); a[imap[k]] += v[perm[jmap[k] .. jmap[k+1])], one addition
   after the other onto the matrix value, as the reference's loop does.  v is the DEVICE receive buffer. */
int hipxCOOCreateIndexed(int64_t nz, const int64_t *imap, const int64_t *jmap, int64_t ntot, const int64_t *perm, hipxCOO *coo);
int hipxMatAddValuesCOOIndexed(hipxMat A, hipxCOO coo, const double *v_dev);
int hipxPointerIsDevice(const void *p, int *is_device);
int hipxMatDestroy(hipxMat *A);
int hipxMatGetInfo(hipxMat A, hipx_int *m, hipx_int *n, int64_t *nnz, int64_t *device_bytes);
/* replaces MatMult_SeqAIJ aij.c:1444 */              int hipxMatMult(hipxMat A, const double *x, double *y);
/* replaces MatMultAdd_SeqAIJ aij.c:1606 (z may alias y) */ int hipxMatMultAdd(hipxMat A, const double *x, const double *y, double *z);
/* replaces MatGetDiagonal_SeqAIJ aij.c:1347 */       int hipxMatGetDiagonal(hipxMat A, double *d);
/* replaces MatSOR_SeqAIJ aij.c:1842 (flag = MatSORType bits petscmat.h:1664-1671); b, x device vectors */
int hipxMatSOR(hipxMat A, const double *b, double omega, int flag, double shift, hipx_int its, hipx_int lits, double *x);
/* which schedule the last hipxMatSOR call used: 4 = plane march (round 5: zero-guess forward / backward / symmetric sweeps -- PCSOR's default
   application -- of constant-coefficient box stencils in natural ordering, 7-point ... 27-point: a workgroup owns four consecutive planes of a
   block of 64 grid lines, one line per lane, every operand of the recurrence in LDS or registers; csrc/hipx_sorbox.hip), 2 = strands (stencil
   matrices with row templates: one lane per grid line, a wave = 64 lines, neighbours through LDS windows, operands through memory), 1 =
   level-ordered dependency-driven sweep, 0 = one launch per level, 3 = the node-level sweep of a matrix with inodes (below); -1 = none yet.
   HIPX_SOR_MODE=box|strand|dep|levels forces one, HIPX_SOR_BOX=0 keeps the plane march out of the default choice (all are bit-identical to
   aij.c:1930-2002). */
int hipxMatGetSORMode(hipxMat A, int *mode);
/* INODES (Mat_SeqAIJ_Inode, aij.h:98-132).  A MATSEQAIJ matrix whose consecutive rows share one column list (blocked FEM operators:
   several unknowns per mesh node) is relaxed NODE by node by the reference: MatSOR_SeqAIJ hands such a matrix to MatSOR_SeqAIJ_Inode
   when omega == 1 and fshift == 0 (aij.c:1852; inode.c:2494-3810) -- block Gauss-Seidel with the inverses of the nodes' dense diagonal
   blocks (LINPACK dgefa/dgedi, dgefa2.c ... dgefa5.c), the rows' terms subtracted in pairs -- a DIFFERENT preconditioner from the point
   sweep, so hipxMatSOR does the same (mode 3 of hipxMatGetSORMode: node-level dependency-driven sweep, bit-identical to inode.c).  By
   default the nodes are found at the FIRST PRODUCT OR SWEEP of a SQUARE, uncompressed matrix nobody has described (hipxMatMult / MultAdd /
   MultDot* / hipxMatSOR: a row-compare pass on the device, an m-byte copy to the host, one stream synchronisation, once per matrix) exactly
   as MatSeqAIJCheckInode finds them at assembly (inode.c:3920-3985: runs of at most 5 identical rows; not used when they number more than
   0.8 m, i.e. never on scalar stencils) -- and from then on hipxMatMult / MultAdd take MatMult_SeqAIJ_Inode's PAIRWISE row sums (inode.c:356-760),
   as the reference does on such a matrix.  Rectangular matrices are never searched (round 5): the reference switches inodes OFF on the
   off-diagonal block of an MPIAIJ matrix (mpiaij.c:824), and an off-diagonal block handed over as plain CSR is rectangular in all but
   degenerate splits; a caller whose off-diagonal block happens to be square says so with hipxMatSetInodes(B, 0, NULL).
   A node whose diagonal block is singular makes hipxMatSOR return HIPX_ERR_ZEROPIVOT (x untouched), like a zero diagonal on the point path.
   SOR_APPLY_UPPER / SOR_APPLY_LOWER on a matrix with inodes return HIPX_ERR_SUP (MatSOR_SeqAIJ_Inode has no such branch; declare the matrix
   free of inodes to get the point routine's).  hipxMatSetInodes overrides the search:
   node_count > 0 with the node_count + 1 row offsets of the nodes (Mat_SeqAIJ_Inode::size_csr, a HOST array: what the PETSc plugin
   passes from the matrix it wraps), or node_count == 0 for "no inodes" (-mat_no_inode).  hipxMatGetInodes reports the count in use
   (0: none, -1: not determined yet). */
int hipxMatSetInodes(hipxMat A, hipx_int node_count, const hipx_int *size_csr);
int hipxMatGetInodes(hipxMat A, hipx_int *node_count);
/* tuning knobs (plugin option -mat_aijhipx_spmv_variant): kernel variant.  0 = auto (>= 2^20 nonzeros: packed 16-bit
   column codes, row-parallel gather for short rows, and an 8-bit value dictionary when a[] holds <= 256 distinct bit
   patterns); 1..12 = 32-bit-column stream kernel geometries; 22 / 23 = packed columns (staged / row-parallel);
   24 / 25 = 22 / 23 plus the value dictionary (falls back to 22 / 23 when the dictionary does not fit);
   26 = row templates: a matrix whose rows are <= 256 distinct (column - row, value) sequences (stencil operators in natural
   ordering) is stored as one template id per row (falls back to 25).  Auto picks 26 when the dictionary exists.
   30 = 26 with the march form of the template kernel (three planes of x resident in LDS) whenever the base template has that shape,
   however few workgroups that gives (auto and 26 take it from 192 workgroups on).
   Every variant produces the bit-identical y (same products, same left-to-right row sums as aij.c:1486-1494). */
int hipxMatSetSpMVVariant(hipxMat A, int variant);
/* name of the kernel the next hipxMatMult will launch (builds the packed formats if they are pending) */
int hipxMatGetSpMVKernel(hipxMat A, char *buf, size_t len);
/* y = A x and *dot = x.y fused in the SpMV epilogue (cg.c:257-258) */
int hipxMatMultDot(hipxMat A, const double *x, double *y, double *dot);
int hipxMatMultDotBegin(hipxMat A, const double *x, double *y, int slot, double *dev_dot); /* enqueue only; hipxRedEnd(slot, 1, &dot) waits */
/* Single-reduction CG (KSPSolve_CG_SingleReduction cg.c:364-534): the vector updates between two reductions in one pass -- p = z + b p (cg.c:470),
   w = s + b w (cg.c:477), x += a p (cg.c:490), r -= a w (cg.c:491), z = r .* d (PCApply_Jacobi; d == NULL: z = r, PCNONE) -- element by element the
   five reference loops, hence the same bits.  The caller then forms s = A z and the three sums z.z, z.s, z.r in ONE reduction (hipxVecMDot /
   hipxVecMDotAllreduce with y = {z, s, r}). */
int hipxCGSingleReductionUpdate(double *p, double *w, double *x, double *r, double *z, const double *s, const double *d, double b, double a, hipx_int n);
/* round 5: the launch-ahead form.  The scalars are formed on the device from dev_sums3 = {z.z, z.s, z.r} of the reduction queued before (hipxVecMDotBegin /
   hipxVecMDotBeginAllreduce with y = {z, s, r}) and dev_state_old = {beta, dpi, a} of the iteration before -- b = beta / betaold (cg.c:464), dpi = delta - beta *
   beta * dpiold / (betaold * betaold) (cg.c:478), a = beta / dpi (cg.c:488), the host's expressions in the host's order -- and dev_state_new <- {beta, dpi, a}.
   The x update applied is the one the iteration BEFORE left behind (x += a_old p_old, before p changes): the caller applies the last one itself. */
int hipxCGSingleReductionUpdateDev(double *p, double *w, double *x, double *r, double *z, const double *s, const double *d, const double *dev_sums3, const double *dev_state_old,
                                   double *dev_state_new, hipx_int n);
/* x . y_j for j < nv <= 16, enqueued only: hipxRedEnd(slot, nv, ...) collects the sums; dev_results receives a device copy for the kernels queued behind */
int hipxVecMDotBegin(const double *x, hipx_int nv, const double *const *y, hipx_int n, int slot, double *dev_results);
/* The CG direction update as the PROLOGUE of the product (round 4): p_new = (z * dconst) + b p_old (cg.c:248-249, VecAYPX dvec2.c:774; z = the
   preconditioned residual with dconst = 1, or the residual itself with the constant Jacobi diagonal / PCNONE), x += a p_old (cg.c:305 of the
   iteration before), w = A p_new, dot = p_new . w (cg.c:257-258) in ONE kernel -- element by element the operations of hipxCGAypxAxpyDev / R
   followed by hipxMatMultDotBegin, hence the same bits.  b and a: from the device-resident sums when dev_beta_new != NULL (b = *dev_beta_new /
   *dev_beta_old, a = *dev_beta_old / *dev_dpi), else the arguments.  p_new must be a second direction vector (other workgroups still read
   p_old), w must not alias z.  Only matrices that take the second-generation march form (stencil row templates, spmv_march2_kernel) with
   16-byte aligned vectors support it: *fused = 0 means NOTHING was enqueued and the caller runs the separate kernels. */
int hipxMatMultCGDirectionDotBegin(hipxMat A, const double *p_old, double *p_new, const double *z, double dconst, double *x, double b, double a, const double *dev_beta_new,
                                   const double *dev_beta_old, const double *dev_dpi, double *w, int slot, double *dev_dot, int *fused);

/* ---- PC ------------------------------------------------------------------------------------------ */
/* PCSetUp_Jacobi jacobi.c:205-266 (DIAGONAL, fixdiag): d = 1/diag(A), zeros -> 1 */
int hipxPCJacobiSetUp(hipxMat A, double *dinv);

/* replaces PCApply_PBJacobi / PCApplyTranspose_PBJacobi pbjacobi.c:4-124,126-241: y_i = D_i^{-1} x_i, diag = mbs inverted bs x bs blocks,
   column-major (what MatInvertBlockDiagonal leaves), all device pointers */
int hipxPCPBJacobiApply(const double *diag, hipx_int bs, hipx_int mbs, const double *x, double *y, int transpose);

/* ---- multi-GPU: MPIAIJ ghost exchange (replaces VecScatterBegin/End vscat.c:1294,1353 on this path
        and PetscSFBcast{Begin,End}_Basic sfbasic.c:352-390) and scalar all-reduces
        (VecXDot_MPI_Default pvecimpl.h:105-111, VecNorm_MPI_Default pvecimpl.h:150-175) ------------- */
#define HIPX_COMM_ID_BYTES 256 /* two ncclUniqueId: one communicator for the ghost exchange (comm stream), one for the
                                  scalar all-reduces (compute stream), so the two never serialise against each other */
int hipxCommGetUniqueId(void *id256);                         /* rank 0; broadcast the bytes yourself (MPI_Bcast / torch store) */
int hipxCommInit(const void *id256, int rank, int nranks);    /* RCCL communicators */
int hipxCommFinalize(void);
/* RCCL-free alternative for the scalar all-reduces: IPC-mapped arenas, peer stores + sequence flags, contributions added in rank
   order (the same bits on every rank).  Also works when ranks share a GPU.  Export on every rank, all-gather the 64-byte handles
   (MPI / torch.distributed), attach.  Pair it with hipxHaloIpcExport/Attach for the ghost exchange. */
int hipxCommIpcExport(int rank, int nranks, void *handle64);
int hipxCommIpcAttach(const void *all_handles /* nranks x 64 bytes, by rank */);
int hipxCommRank(int *rank, int *nranks);
int hipxCommAllreduceSum(double *host_vals, int n);           /* n <= 64 doubles, device-staged ncclAllReduce */
/* HIPX_ERR_GPU if an all-reduce of the IPC transport gave up on a peer (wait limit) since the communicator came up: the sums it
   produced are not sums.  The host-synchronised forms check this themselves; callers that collect stream-ordered reductions with
   hipxRedEnd (the launch-ahead CG on several ranks) call it after each one. */
int hipxCommCheckError(void);
/* VecTDot_MPI / VecMDot_MPI (pvecimpl.h:97-111) in one stream-ordered chain: local dot kernel(s) -> ncclAllReduce on the
   result words -> host notification; the host waits once, after the all-reduce.  nv <= 16.  Single rank: plain local dots. */
int hipxVecMDotAllreduce(const double *x, hipx_int nv, const double *const *y, hipx_int n, double *results);
/* hipxCGFusedUpdate with the two sums all-reduced over the communicator in the same chain (VecNorm_MPI + VecTDot_MPI of
   cg.c:309,344 in one 16-byte all-reduce) */
int hipxCGFusedUpdateAllreduce(double *x, double *r, double *z, const double *p, const double *w, const double *d, double a, hipx_int n, double *sums2);

typedef struct hipxHalo_s *hipxHalo;
/* nsend/nrecv neighbours; send_idx = local indices of owned entries to pack per neighbour (concatenated,
   offsets in send_off[nsend+1]); received values land contiguously in lvec (recv_off[nrecv+1]) exactly as
   mmaij.c:108-117 lays lvec[k] <-> garray[k]. */
int hipxHaloCreate(int nsend, const int *send_ranks, const hipx_int *send_off, const hipx_int *send_idx,
                   int nrecv, const int *recv_ranks, const hipx_int *recv_off, hipxHalo *h);
int hipxHaloDestroy(hipxHalo *h);
/* Second transport for the same plan: PEER STORES through HIP IPC mappings instead of RCCL send/recv.  The pack kernel of the
   sender writes x[send_idx] straight into the receiver's ghost buffer (over xGMI between GPUs; through L2 when ranks share a
   GPU, which RCCL refuses), then publishes a sequence number; the receiver's stream waits for it in a one-wave kernel; ghost
   buffers are double-buffered and acknowledged, so back-to-back products need no other synchronisation.  Set-up: every rank
   exports a blob, the host all-gathers them (MPI_Allgather / torch.distributed), every rank attaches.  Once attached,
   hipxMatMultMPI takes this path (its lvec argument is then unused). */
#define HIPX_HALO_IPC_BLOB_BYTES 1024
int hipxHaloIpcExport(hipxHalo h, int rank, int nranks, void *blob1024);
int hipxHaloIpcAttach(hipxHalo h, const void *all_blobs /* nranks x 1024 bytes, indexed by rank */);
int hipxHaloBegin(hipxHalo h, const double *x, double *lvec); /* pack on compute stream -> send/recv on comm stream */
int hipxHaloEnd(hipxHalo h);                                  /* compute stream waits for the exchange */
/* Callers of the Begin/End pair (instead of hipxMatMultMPI): where the ghost values of the exchange just ended are -- lvec
   itself with RCCL, the exchange's IPC ghost buffer with peer stores (lvec is then not written) -- and, once every kernel that
   reads them has been enqueued on the compute stream, hipxHaloRelease: with IPC it acknowledges the buffer to the senders
   (the exchange after next may overwrite it); a no-op with RCCL.  Every Begin/End needs its Release. */
int hipxHaloGhost(hipxHalo h, const double *lvec, const double **ghost);
int hipxHaloRelease(hipxHalo h);
/* replaces MatMult_MPIAIJ mpiaij.c:1047-1061: halo begin; y = Ad x (overlapped); halo end; y += Bo lvec */
int hipxMatMultMPI(hipxMat Ad, hipxMat Bo, hipxHalo h, const double *x, double *lvec, double *y);
/* replaces MatMultAdd_MPIAIJ mpiaij.c:1072-1083: halo begin; z = y + Ad x (overlapped); halo end; z += Bo lvec */
int hipxMatMultAddMPI(hipxMat Ad, hipxMat Bo, hipxHalo h, const double *x, double *lvec, const double *y, double *z);
/* Launch-ahead CG on several ranks (round 3): the reductions complete on the stream -- local kernel -> all-reduce -> publish to the host
   slot (hipxRedEnd collects it) AND to device memory -- so the next iteration's kernels (hipxCGAypxAxpyDev, ...Begin forms, which read
   their scalars from device memory) can be queued before the host has seen the sums.  hipxMatMultMPIDotBegin = MatMult_MPIAIJ +
   VecTDot_MPI (cg.c:257-258); hipxCGFusedUpdateBeginAllreduce = hipxCGFusedUpdateBegin + the 16-byte all-reduce of its two sums. */
int hipxMatMultMPIDotBegin(hipxMat Ad, hipxMat Bo, hipxHalo h, const double *x, double *lvec, double *y, hipx_int n, int slot, double *dev_dot);
int hipxVecMDotBeginAllreduce(const double *x, hipx_int nv, const double *const *y, hipx_int n, int slot, double *dev_results); /* the same, all-reduced on the stream */
int hipxCGFusedUpdateBeginAllreduce(double *x, double *r, double *z, const double *p, const double *w, const double *d, double dconst, const double *dev_beta, const double *dev_dpi,
                                    hipx_int n, int slot, double *dev_sums2);
/* Round 6: hipxMatMultCGDirectionDotBegin for a rank WITH an off-diagonal block -- MatMult_MPIAIJ (mpiaij.c:1047-1061) + the CG direction update (cg.c:248-249,
   and cg.c:305 of the iteration before) + VecTDot_MPI (cg.c:258) as: [new direction of the boundary rows formed on the way into the ghost exchange] ||
   [p_new = z d + b p, x += a p, w = Ad p_new, dot partials of the rows without off-diagonal entries: ONE kernel] -> [w += Bo ghost on the boundary rows, their
   share of the dot, the fold of all partials: one small kernel] -> all-reduce on the stream -> host slot + device copy.  Replaces the sequence hipxCGAypxAxpyDev,
   hipxMatMultMPIDotBegin (a separate direction kernel and a separate dot kernel: 7 vector passes more).  Same arguments as the one-rank form plus the blocks,
   the halo plan and lvec; n = local rows.  *fused = 0: nothing enqueued (the blocks are not a z-slab of a stencil grid in natural ordering, or the march
   kernel does not take the diagonal block): the caller runs the separate kernels.  Vectors bit-identical to those; the dot = the same products in another
   order (exact reduction mode: Dot2 over the complete vectors, order-free). */
int hipxMatMultMPICGDirectionDotBegin(hipxMat Ad, hipxMat Bo, hipxHalo h, const double *p_old, double *p_new, const double *z, double dconst, double *x, double b, double a,
                                      const double *dev_beta_new, const double *dev_beta_old, const double *dev_dpi, double *lvec, double *w, hipx_int n, int slot, double *dev_dot, int *fused);
/* Round 6: split-phase all-reduce -- PetscCommSplitReductionBegin / PetscSplitReductionEnd (comb.c:168-290: MPI_Iallreduce now, MPI_Wait at the first End) on the
   device.  hipxPipeCGUpdateBeginAllreduce = hipxPipeCGUpdateBegin whose three local sums stay in device memory and START their all-reduce (IPC transport: a
   one-wave kernel stores them into every peer's arena and raises the sequence flags; RCCL: ncclAllReduce on the comm stream behind an event); the caller enqueues
   the work the reduction hides behind (PIPECG: PCApply + MatMult_MPIAIJ), then hipxAllreduceEnd: the compute stream waits for the peers' contributions, folds them
   in rank order (plain or compensated), publishes the nvals totals to the host slot and to dev_out.  One all-reduce per PIPECG iteration, overlapped with the product. */
int hipxPipeCGUpdateBeginAllreduce(const hipxPipeCGVecs *v, const double *d, double dconst, int normkind, int first, const double *dev_sums, const double *dev_sums_old,
                                   const double *dev_alpha_old, double *dev_alpha_out, hipx_int n, int slot);
/* Gropp's CG on several ranks: the direction pass with its one sum all-reduced on the stream (reduction 1: nothing is left to hide it behind once S = B s is formed inside the
   update pass), and the update pass whose two sums START their all-reduce -- the product Z = A z runs -- hipxAllreduceEnd (reduction 2 hidden behind the product, groppcg.c:107-117) */
int hipxGroppCGDirectionBeginAllreduce(double *p, double *s, double *x, const double *z, const double *Z, const double *dev_gamma_new, const double *dev_gamma_old, const double *dev_alpha_old,
                                       hipx_int n, int slot, double *dev_t_out);
int hipxGroppCGUpdateBeginAllreduce(double *r, double *z, const double *s, const double *d, double dconst, int normkind, const double *dev_gamma, const double *dev_t, double *dev_alpha_out,
                                    hipx_int n, int slot);
int hipxVecMDotAllreduceBegin(const double *x, hipx_int nv, const double *const *y, hipx_int n, int slot); /* local x . y_j (nv <= 16) + the start of their all-reduce */
int hipxAllreduceEnd(int slot, int nvals, double *dev_out);
/* which transport the next exchange of this plan takes: 1 = IPC peer stores, 2 = RCCL send/recv, 0 = none set up */
int hipxHaloTransport(hipxHalo h, int *transport);

#ifdef __cplusplus
}
#endif
#endif
