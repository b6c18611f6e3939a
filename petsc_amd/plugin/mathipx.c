/*
 * mathipx.c -- MATSEQAIJHIPX: a Mat_SeqAIJ (src/mat/impls/aij/seq/aij.h:47-78,150-168) whose MatMult / MatMultAdd /
 * MatSOR / MatGetDiagonal run as HIP kernels.  Assembly, viewing, loading, MatSetValues, duplication ... stay with the
 * parent class (MatCreate_SeqAIJ is exported, aij.h:459); the CSR arrays are uploaded once per nonzero state and the
 * values refreshed when the object state changes (the triggering event is MatAssemblyEnd, aij.c:1085).
 * Slots of struct _MatOps (include/petsc/private/matimpl.h:38-213) owned here: mult, multadd, sor, getdiagonal,
 * assemblyend, duplicate, destroy (+ MatConvert_seqaij_seqaijhipx_C so -mat_type aijhipx works on an assembled matrix,
 * matreg.c:146-150).
 */
#include "hipxplugin.h"

typedef struct {
  hipxMat          dA;
  PetscObjectState nonzerostate; /* pattern the device copy was built from */
  PetscObjectState valuestate;   /* object state of the values on the device */
  hipxMat          dAt;          /* the transposed matrix as its own device CSR (MatMultTranspose / MatMultTransposeAdd), built at the first use */
  PetscObjectState t_nonzerostate, t_valuestate;
  PetscErrorCode (*parent_assemblyend)(Mat, MatAssemblyType);
  PetscErrorCode (*parent_destroy)(Mat);
  PetscErrorCode (*parent_duplicate)(Mat, MatDuplicateOption, Mat *);
  PetscErrorCode (*parent_prealloc_coo)(Mat, PetscCount, PetscInt[], PetscInt[]);
  PetscInt  spmv_variant;
  hipxCOO   coo;       /* device copies of the reference's COO maps (MatCOOStruct_SeqAIJ jmap / perm) */
  PetscBool dev_newer; /* the device value array is ahead of the host copy a->a (MatSetValuesCOO ran on the device) */
  PetscInt  inode_sig; /* the inode partition libhipx was told: node_count, 0 = none, -1 = nothing yet (compressed-row copies: never) */
} Mat_SeqAIJHIPX;

/* The inodes of the host matrix -> libhipx (MatSeqAIJCheckInode inode.c:3920 found them at assembly, under -mat_no_inode / -mat_inode_limit;
   MatAssemblyEnd_MPIAIJ switches them off on the off-diagonal block, mpiaij.c:824 -- AFTER a COO preallocation has already assembled that
   block once: the state is looked at again before every use).  The reference multiplies a matrix with inodes with MatMult_SeqAIJ_Inode and
   relaxes it node by node (aij.c:1459, 1852), and so does libhipx once it is told the partition.  Told in EVERY case: left alone, libhipx looks
   for inodes itself -- and would find some among the empty rows of an off-diagonal block. */
static PetscErrorCode MatSeqAIJHIPXSyncInodes(Mat A, hipxMat dA, PetscInt *sig_io)
{
  Mat_SeqAIJ *a   = (Mat_SeqAIJ *)A->data;
  PetscInt    sig = (a->inode.use && a->inode.checked && a->inode.node_count > 0 && a->inode.size_csr) ? a->inode.node_count : 0;

  PetscFunctionBegin;
  if (*sig_io == sig) PetscFunctionReturn(PETSC_SUCCESS);
  if (sig) {
    hipx_int *ns;
    PetscCall(PetscMalloc1((size_t)sig + 1, &ns));
    for (PetscInt k = 0; k <= sig; k++) ns[k] = (hipx_int)a->inode.size_csr[k];
    PetscCallHIPX(hipxMatSetInodes(dA, (hipx_int)sig, ns));
    PetscCall(PetscFree(ns));
  } else PetscCallHIPX(hipxMatSetInodes(dA, 0, NULL));
  *sig_io = sig;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode MatMult_SeqAIJHIPX(Mat, Vec, Vec);

PetscBool MatIsSeqAIJHIPX(Mat A)
{
  return (PetscBool)(A && A->ops->mult == MatMult_SeqAIJHIPX);
}

/* The CSR arrays of Mat_SeqAIJ -> libhipx.  hipx_int is 32-bit; with a 64-bit-PetscInt libpetsc (--with-64-bit-indices: systems beyond
   2^31 nonzeros, e.g. the 27-pt 512^3 operator on ONE GPU) the row offsets go through as they are (hipxMatCreateCSR64), the column
   indices -- which address a vector of < 2^31 local entries -- are narrowed into a temporary. */
static PetscErrorCode MatSeqAIJHIPXCreateDevice(Mat A, const PetscScalar *aa, PetscBool use_cprow, hipxMat *dA)
{
  Mat_SeqAIJ *a = (Mat_SeqAIJ *)A->data;

  PetscFunctionBegin;
#if defined(PETSC_USE_64BIT_INDICES)
  {
    hipx_int *j32;
    PetscCheck(A->rmap->n < PETSC_INT32_MAX && A->cmap->n < PETSC_INT32_MAX, PETSC_COMM_SELF, PETSC_ERR_SUP, "MATSEQAIJHIPX: local sizes must stay below 2^31 (only the nonzero count may exceed it)");
    PetscCall(PetscMalloc1((size_t)a->nz + 1, &j32));
    for (PetscInt k = 0; k < a->nz; k++) j32[k] = (hipx_int)a->j[k];
    if (use_cprow) {
      hipx_int *ci, *ri, nr = (hipx_int)a->compressedrow.nrows;
      PetscCheck(a->nz < PETSC_INT32_MAX, PETSC_COMM_SELF, PETSC_ERR_SUP, "compressed-row block beyond 2^31 nonzeros");
      PetscCall(PetscMalloc2((size_t)nr + 1, &ci, (size_t)nr + 1, &ri));
      for (hipx_int k = 0; k <= nr; k++) ci[k] = (hipx_int)a->compressedrow.i[k];
      for (hipx_int k = 0; k < nr; k++) ri[k] = (hipx_int)a->compressedrow.rindex[k];
      PetscCallHIPX(hipxMatCreateCSRCompressedRow((hipx_int)A->rmap->n, (hipx_int)A->cmap->n, nr, ci, ri, j32, aa, dA));
      PetscCall(PetscFree2(ci, ri));
    } else PetscCallHIPX(hipxMatCreateCSR64((hipx_int)A->rmap->n, (hipx_int)A->cmap->n, (const int64_t *)a->i, j32, aa, dA));
    PetscCall(PetscFree(j32));
  }
#else
  if (use_cprow) PetscCallHIPX(hipxMatCreateCSRCompressedRow(A->rmap->n, A->cmap->n, a->compressedrow.nrows, a->compressedrow.i, a->compressedrow.rindex, a->j, aa, dA));
  else PetscCallHIPX(hipxMatCreateCSR(A->rmap->n, A->cmap->n, a->i, a->j, aa, dA));
#endif
  PetscFunctionReturn(PETSC_SUCCESS);
}


/* device CSR, (re)built lazily */
PetscErrorCode MatSeqAIJHIPXGetDeviceMat(Mat A, hipxMat *dA)
{
  Mat_SeqAIJHIPX  *h = (Mat_SeqAIJHIPX *)A->spptr;
  Mat_SeqAIJ      *a = (Mat_SeqAIJ *)A->data;
  PetscObjectState state;

  PetscFunctionBegin;
  PetscCheck(A->assembled, PetscObjectComm((PetscObject)A), PETSC_ERR_ARG_WRONGSTATE, "Not for unassembled matrix");
  PetscCall(PetscObjectStateGet((PetscObject)A, &state));
  if (h->dev_newer && h->dA && h->nonzerostate == A->nonzerostate) { /* values were assembled on the device: nothing to upload */
    h->valuestate = state;
    if (h->inode_sig != -2) PetscCall(MatSeqAIJHIPXSyncInodes(A, h->dA, &h->inode_sig));
    *dA = h->dA;
    PetscFunctionReturn(PETSC_SUCCESS);
  }
  if (!h->dA || h->nonzerostate != A->nonzerostate) {
    const PetscScalar *aa;
    if (h->dA) PetscCallHIPX(hipxMatDestroy(&h->dA));
    PetscCall(MatSeqAIJGetArrayRead(A, &aa));
    if (a->compressedrow.use && a->compressedrow.nrows < A->rmap->n && A->rmap->n != A->cmap->n) { /* rectangular: never asked for a diagonal / SOR */
      /* Mat_CompressedRow (matimpl.h:425-430; MatAssemblyEnd_SeqAIJ checks it, aij.c:1138): the off-diagonal block of an MPIAIJ
         matrix has entries in a few rows only -- MatMult_SeqAIJ / MatMultAdd_SeqAIJ then walk the listed rows (aij.c:1463-1478,
         1624-1641), and so does the device kernel (y is not streamed for the empty rows) */
      PetscCall(MatSeqAIJHIPXCreateDevice(A, aa, PETSC_TRUE, &h->dA));
      h->inode_sig = -2; /* compressed-row copy: libhipx never looks for inodes there */
    } else {
      PetscCall(MatSeqAIJHIPXCreateDevice(A, aa, PETSC_FALSE, &h->dA));
      h->inode_sig = -1;
    }
    PetscCall(MatSeqAIJRestoreArrayRead(A, &aa));
    if (h->spmv_variant) PetscCallHIPX(hipxMatSetSpMVVariant(h->dA, (int)h->spmv_variant));
    h->nonzerostate = A->nonzerostate;
    h->valuestate   = state;
  } else if (h->valuestate != state) {
    const PetscScalar *aa;
    PetscCall(MatSeqAIJGetArrayRead(A, &aa));
    PetscCallHIPX(hipxMatUpdateValues(h->dA, aa));
    PetscCall(MatSeqAIJRestoreArrayRead(A, &aa));
    h->valuestate  = state;
  }
  if (h->inode_sig != -2) PetscCall(MatSeqAIJHIPXSyncInodes(A, h->dA, &h->inode_sig));
  *dA = h->dA;
  PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode MatSeqAIJHIPXSetUpDevice(Mat A)
{
  static PetscBool on = PETSC_TRUE, looked = PETSC_FALSE;
  hipxMat          dA;

  PetscFunctionBegin;
  if (!looked) {
    PetscCall(PetscOptionsGetBool(NULL, NULL, "-mat_hipx_setup_at_assembly", &on, NULL));
    looked = PETSC_TRUE;
  }
  if (!on || !A->assembled) PetscFunctionReturn(PETSC_SUCCESS);
  PetscCall(MatSeqAIJHIPXGetDeviceMat(A, &dA));
  PetscCallHIPX(hipxMatSetUp(dA));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* MatMult_SeqAIJ aij.c:1444-1502 */
static PetscErrorCode MatMult_SeqAIJHIPX(Mat A, Vec xx, Vec yy)
{
  Mat_SeqAIJ        *a = (Mat_SeqAIJ *)A->data;
  hipxMat            dA;
  const PetscScalar *x;
  PetscScalar       *y;
  void              *tx, *ty;

  PetscFunctionBegin;
  PetscCall(MatSeqAIJHIPXGetDeviceMat(A, &dA));
  if (A->rmap->n == A->cmap->n && !a->compressedrow.use) { /* "x += a p; p = z + b p" recorded on xx (lazy fusion, hipxplugin.h): the product kernel's prologue */
    PetscBool done;
    PetscCall(VecHIPXLazyTryCGProduct(dA, xx, yy, &done));
    if (done) {
      PetscCall(PetscLogFlops(2.0 * a->nz - a->nonzerorowcnt));
      PetscFunctionReturn(PETSC_SUCCESS);
    }
  }
  PetscCall(VecHIPXGetDeviceRead(xx, &x, &tx));
  PetscCall(VecHIPXGetDeviceWrite(yy, &y, &ty));
  {
    int mode = HIPX_RED_FAST;
    /* KSPSolve_CG asks for p . (A p) right after this product (cg.c:257-258): the product's kernels can leave that sum behind (per-wave partials in
       their epilogue) -- the reduction cache of the vector type hands it to the VecTDot that follows.  Not in the exact reduction mode (there the sum
       is a Dot2 pass over the complete vectors either way), not for compressed-row or rectangular blocks */
    PetscCallHIPX(hipxGetReductionMode(&mode));
    if (mode == HIPX_RED_FAST && !tx && !ty && xx != yy && A->rmap->n == A->cmap->n && A->rmap->n > 0 && !a->compressedrow.use && VecHIPXRedCacheWanted(HIPX_RC_MATMULT)) {
      PetscCallHIPX(hipxMatMultDotBegin(dA, x, y, VecHIPXRedCacheSlot(HIPX_RC_MATMULT), NULL));
      PetscCall(VecHIPXRedCachePut(HIPX_RC_MATMULT, xx, yy));
    } else PetscCallHIPX(hipxMatMult(dA, x, y));
  }
  PetscCall(VecHIPXRestoreDeviceWrite(yy, &y, &ty));
  PetscCall(VecHIPXRestoreDeviceRead(xx, &x, &tx));
  PetscCall(PetscLogFlops(2.0 * a->nz - a->nonzerorowcnt)); /* aij.c:1497 */
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* MatMultAdd_SeqAIJ aij.c:1606-1658: zz = yy + A xx (zz may be yy) */
static PetscErrorCode MatMultAdd_SeqAIJHIPX(Mat A, Vec xx, Vec yy, Vec zz)
{
  Mat_SeqAIJ        *a = (Mat_SeqAIJ *)A->data;
  hipxMat            dA;
  const PetscScalar *x, *y;
  PetscScalar       *z;
  void              *tx, *ty = NULL, *tz;

  PetscFunctionBegin;
  PetscCall(MatSeqAIJHIPXGetDeviceMat(A, &dA));
  PetscCall(VecHIPXGetDeviceRead(xx, &x, &tx));
  if (zz == yy) {
    PetscCall(VecHIPXGetDeviceReadWrite(zz, &z, &tz));
    y = z;
  } else {
    PetscCall(VecHIPXGetDeviceRead(yy, &y, &ty));
    PetscCall(VecHIPXGetDeviceWrite(zz, &z, &tz));
  }
  PetscCallHIPX(hipxMatMultAdd(dA, x, y, z));
  PetscCall(VecHIPXRestoreDeviceWrite(zz, &z, &tz));
  if (zz != yy) PetscCall(VecHIPXRestoreDeviceRead(yy, &y, &ty));
  PetscCall(VecHIPXRestoreDeviceRead(xx, &x, &tx));
  PetscCall(PetscLogFlops(2.0 * a->nz)); /* aij.c:1653 */
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* ---- MatMultTranspose / MatMultTransposeAdd on the device (aij.c:1383-1440).  The reference walks the ROWS of A and adds alpha v[j] into
   y[idx[j]]: column c of A receives its contributions in ascending row order, each a separately rounded product added to the running
   value (which starts at 0 for MatMultTranspose, at zz_c for MatMultTransposeAdd).  That is exactly the left-to-right row sum of the
   TRANSPOSED matrix in CSR form with its rows' entries in ascending column (= original row) order -- so the transpose is built once per
   matrix state (a counting pass over the host CSR, as cheap as the upload) and the products are hipxMatMult / hipxMatMultAdd on it:
   the same bits as the CPU loop, any of libhipx's kernel forms. */
static PetscErrorCode MatSeqAIJHIPXGetDeviceTranspose(Mat A, hipxMat *dAt)
{
  Mat_SeqAIJHIPX  *h = (Mat_SeqAIJHIPX *)A->spptr;
  Mat_SeqAIJ      *a = (Mat_SeqAIJ *)A->data;
  PetscObjectState state;

  PetscFunctionBegin;
  PetscCheck(A->assembled, PetscObjectComm((PetscObject)A), PETSC_ERR_ARG_WRONGSTATE, "Not for unassembled matrix");
  PetscCall(PetscObjectStateGet((PetscObject)A, &state));
  if (!h->dAt || h->t_nonzerostate != A->nonzerostate || h->t_valuestate != state) {
    const PetscInt     m = A->rmap->n, n = A->cmap->n, nz = a->nz, *ai = a->i, *aj = a->j;
    const PetscScalar *aa;
    PetscInt          *ti, *cur;
    hipx_int          *tj;
    PetscScalar       *ta;
    PetscCheck(m < PETSC_INT32_MAX && n < PETSC_INT32_MAX, PETSC_COMM_SELF, PETSC_ERR_SUP, "MATSEQAIJHIPX: local sizes must stay below 2^31");
    if (h->dAt) PetscCallHIPX(hipxMatDestroy(&h->dAt));
    PetscCall(MatSeqAIJGetArrayRead(A, &aa)); /* (brings the host copy up to date if the values were last assembled on the device) */
    PetscCall(PetscCalloc1((size_t)n + 1, &ti));
    PetscCall(PetscMalloc3((size_t)n + 1, &cur, (size_t)nz + 1, &tj, (size_t)nz + 1, &ta));
    for (PetscInt k = 0; k < nz; k++) ti[aj[k] + 1]++;
    for (PetscInt c = 0; c < n; c++) ti[c + 1] += ti[c];
    for (PetscInt c = 0; c < n; c++) cur[c] = ti[c];
    for (PetscInt i = 0; i < m; i++)
      for (PetscInt k = ai[i]; k < ai[i + 1]; k++) {
        const PetscInt pos = cur[aj[k]]++;
        tj[pos]            = (hipx_int)i;
        ta[pos]            = aa[k];
      }
#if defined(PETSC_USE_64BIT_INDICES)
    PetscCallHIPX(hipxMatCreateCSR64((hipx_int)n, (hipx_int)m, (const int64_t *)ti, tj, ta, &h->dAt));
#else
    PetscCallHIPX(hipxMatCreateCSR((hipx_int)n, (hipx_int)m, ti, tj, ta, &h->dAt));
#endif
    PetscCallHIPX(hipxMatSetInodes(h->dAt, 0, NULL)); /* the products with it are MatMultTranspose_SeqAIJ's left-to-right sums: never the inode order */
    PetscCall(MatSeqAIJRestoreArrayRead(A, &aa));
    PetscCall(PetscFree3(cur, tj, ta));
    PetscCall(PetscFree(ti));
    h->t_nonzerostate = A->nonzerostate;
    h->t_valuestate   = state;
  }
  *dAt = h->dAt;
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* MatMultTranspose_SeqAIJ aij.c:1434-1440 */
static PetscErrorCode MatMultTranspose_SeqAIJHIPX(Mat A, Vec xx, Vec yy)
{
  Mat_SeqAIJ        *a = (Mat_SeqAIJ *)A->data;
  hipxMat            dAt;
  const PetscScalar *x;
  PetscScalar       *y;
  void              *tx, *ty;

  PetscFunctionBegin;
  PetscCall(MatSeqAIJHIPXGetDeviceTranspose(A, &dAt));
  PetscCall(VecHIPXGetDeviceRead(xx, &x, &tx));
  PetscCall(VecHIPXGetDeviceWrite(yy, &y, &ty));
  PetscCallHIPX(hipxMatMult(dAt, x, y));
  PetscCall(VecHIPXRestoreDeviceWrite(yy, &y, &ty));
  PetscCall(VecHIPXRestoreDeviceRead(xx, &x, &tx));
  PetscCall(PetscLogFlops(2.0 * a->nz)); /* aij.c:1425 */
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* MatMultTransposeAdd_SeqAIJ aij.c:1383-1432: yy = zz + A^T xx (yy may be zz) */
static PetscErrorCode MatMultTransposeAdd_SeqAIJHIPX(Mat A, Vec xx, Vec zz, Vec yy)
{
  Mat_SeqAIJ        *a = (Mat_SeqAIJ *)A->data;
  hipxMat            dAt;
  const PetscScalar *x, *z;
  PetscScalar       *y;
  void              *tx, *tz = NULL, *ty;

  PetscFunctionBegin;
  PetscCall(MatSeqAIJHIPXGetDeviceTranspose(A, &dAt));
  PetscCall(VecHIPXGetDeviceRead(xx, &x, &tx));
  if (zz == yy) {
    PetscCall(VecHIPXGetDeviceReadWrite(yy, &y, &ty));
    z = y;
  } else {
    PetscCall(VecHIPXGetDeviceRead(zz, &z, &tz));
    PetscCall(VecHIPXGetDeviceWrite(yy, &y, &ty));
  }
  PetscCallHIPX(hipxMatMultAdd(dAt, x, z, y));
  PetscCall(VecHIPXRestoreDeviceWrite(yy, &y, &ty));
  if (zz != yy) PetscCall(VecHIPXRestoreDeviceRead(zz, &z, &tz));
  PetscCall(VecHIPXRestoreDeviceRead(xx, &x, &tx));
  PetscCall(PetscLogFlops(2.0 * a->nz));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* MatGetDiagonal_SeqAIJ aij.c:1347-1380 */
static PetscErrorCode MatGetDiagonal_SeqAIJHIPX(Mat A, Vec v)
{
  hipxMat      dA;
  PetscScalar *d;
  void        *t;

  PetscFunctionBegin;
  PetscCall(MatSeqAIJHIPXGetDeviceMat(A, &dA));
  PetscCall(VecHIPXGetDeviceWrite(v, &d, &t));
  PetscCallHIPX(hipxMatGetDiagonal(dA, d));
  PetscCall(VecHIPXRestoreDeviceWrite(v, &d, &t));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* MatSOR_SeqAIJ aij.c:1842-2007 */
static PetscErrorCode MatSOR_SeqAIJHIPX(Mat A, Vec bb, PetscReal omega, MatSORType flag, PetscReal fshift, PetscInt its, PetscInt lits, Vec xx)
{
  Mat_SeqAIJ        *a = (Mat_SeqAIJ *)A->data;
  hipxMat            dA;
  const PetscScalar *b;
  PetscScalar       *x;
  void              *tb, *tx;
  int                ierr;

  PetscFunctionBegin;
  PetscCall(MatSeqAIJHIPXGetDeviceMat(A, &dA));
  PetscCall(VecHIPXGetDeviceRead(bb, &b, &tb));
  if (flag & SOR_ZERO_INITIAL_GUESS) PetscCall(VecHIPXGetDeviceWrite(xx, &x, &tx));
  else PetscCall(VecHIPXGetDeviceReadWrite(xx, &x, &tx));
  ierr = hipxMatSOR(dA, b, omega, (int)flag, fshift, its, lits, x);
  if (ierr == HIPX_ERR_ZEROPIVOT) { /* aij.c:1818-1824 (point rows), inode.c:2460-2490 (diagonal blocks of nodes): an error when erroriffailure is set,
                                       else flag the matrix -- PCApply_SOR turns it into pc->failedreason (sor.c:34) */
    PetscCheck(!A->erroriffailure, PETSC_COMM_SELF, PETSC_ERR_MAT_LU_ZRPVT, "libhipx: %s", hipxGetErrorString());
    A->factorerrortype             = MAT_FACTOR_NUMERIC_ZEROPIVOT;
    A->factorerror_zeropivot_value = 0.0;
    ierr                           = 0;
  }
  PetscCheck(ierr != HIPX_ERR_SUP, PETSC_COMM_SELF, PETSC_ERR_SUP, "libhipx: %s", hipxGetErrorString());
  PetscCheck(!ierr, PETSC_COMM_SELF, ierr == 73 ? PETSC_ERR_ARG_WRONGSTATE : PETSC_ERR_GPU, "libhipx: %s", hipxGetErrorString());
  PetscCall(VecHIPXRestoreDeviceWrite(xx, &x, &tx));
  PetscCall(VecHIPXRestoreDeviceRead(bb, &b, &tb));
  PetscCall(PetscLogFlops(2.0 * a->nz * its * lits));
  PetscFunctionReturn(PETSC_SUCCESS);
}


/* ---- COO assembly on the device.  MatSetPreallocationCOO_SeqAIJ (aij.c:4524-4707) stays the reference's host
   routine -- integer work, once per pattern; its jmap / perm maps are mirrored on the device and MatSetValuesCOO_SeqAIJ
   (aij.c:4710-4733) becomes one kernel (same left-to-right sums -> bit-identical values).  The host copy a->a is refreshed
   lazily: the value-array accessors of Mat_SeqAIJOps copy it back when a host routine asks for it. */
static PetscErrorCode MatSeqAIJHIPXSyncValuesToHost(Mat A)
{
  Mat_SeqAIJHIPX *h = (Mat_SeqAIJHIPX *)A->spptr;
  Mat_SeqAIJ     *a = (Mat_SeqAIJ *)A->data;

  PetscFunctionBegin;
  if (h->dev_newer && h->dA) {
    PetscCallHIPX(hipxMatGetValues(h->dA, a->a));
    h->dev_newer = PETSC_FALSE; /* both copies equal; a later host write bumps the object state and is uploaded as before */
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode MatSeqAIJGetArray_SeqAIJHIPX(Mat A, PetscScalar *array[])
{
  PetscFunctionBegin;
  PetscCall(MatSeqAIJHIPXSyncValuesToHost(A));
  *array = ((Mat_SeqAIJ *)A->data)->a;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode MatSeqAIJGetArrayRead_SeqAIJHIPX(Mat A, const PetscScalar *array[])
{
  PetscFunctionBegin;
  PetscCall(MatSeqAIJHIPXSyncValuesToHost(A));
  *array = ((Mat_SeqAIJ *)A->data)->a;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode MatSeqAIJGetArrayWrite_SeqAIJHIPX(Mat A, PetscScalar *array[])
{
  PetscFunctionBegin;
  ((Mat_SeqAIJHIPX *)A->spptr)->dev_newer = PETSC_FALSE; /* the host overwrites every value: the restore bumps the object state -> uploaded at the next product */
  *array = ((Mat_SeqAIJ *)A->data)->a;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode MatSetPreallocationCOO_SeqAIJHIPX(Mat A, PetscCount n, PetscInt coo_i[], PetscInt coo_j[])
{
  Mat_SeqAIJHIPX      *h = (Mat_SeqAIJHIPX *)A->spptr;
  PetscContainer       container;
  MatCOOStruct_SeqAIJ *coo;

  PetscFunctionBegin;
  PetscCheck(h->parent_prealloc_coo, PETSC_COMM_SELF, PETSC_ERR_PLIB, "MATSEQAIJ did not provide MatSetPreallocationCOO");
  if (h->coo) PetscCallHIPX(hipxCOODestroy(&h->coo));
  if (h->dA) PetscCallHIPX(hipxMatDestroy(&h->dA)); /* new pattern */
  h->dev_newer = PETSC_FALSE;
  PetscCall((*h->parent_prealloc_coo)(A, n, coo_i, coo_j));
  PetscCall(PetscObjectQuery((PetscObject)A, "__PETSc_MatCOOStruct_Host", (PetscObject *)&container));
  PetscCheck(container, PETSC_COMM_SELF, PETSC_ERR_PLIB, "Not found MatCOOStruct on this matrix");
  PetscCall(PetscContainerGetPointer(container, &coo));
  PetscCallHIPX(hipxCOOCreate((int64_t)coo->nz, (const int64_t *)coo->jmap, (int64_t)coo->Atot, (const int64_t *)coo->perm, &h->coo));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* values v[] (host or device pointer, n entries) through device-resident COO maps into the device CSR of A; also used by
   MATMPIAIJHIPX for its two blocks (MatSetValuesCOO_MPIAIJ's local part, mpiaij.c:6803-6813) */
PetscErrorCode MatSeqAIJHIPXSetValuesCOO_Private(Mat A, hipxCOO coo, const PetscScalar v[], PetscCount n, InsertMode imode)
{
  Mat_SeqAIJHIPX *h = (Mat_SeqAIJHIPX *)A->spptr;
  Mat_SeqAIJ     *a = (Mat_SeqAIJ *)A->data;
  int             ondev = 0;

  PetscFunctionBegin;
  if (!h->dA || h->nonzerostate != A->nonzerostate) { /* device CSR of the preallocated pattern; values start at zero (aij.c:4693) */
    if (h->dA) PetscCallHIPX(hipxMatDestroy(&h->dA));
    if (imode == ADD_VALUES) PetscCall(MatSeqAIJHIPXCreateDevice(A, a->a, PETSC_FALSE, &h->dA));
    else PetscCall(MatSeqAIJHIPXCreateDevice(A, NULL, PETSC_FALSE, &h->dA));
    h->inode_sig = -1;
    PetscCall(MatSeqAIJHIPXSyncInodes(A, h->dA, &h->inode_sig));
    if (h->spmv_variant) PetscCallHIPX(hipxMatSetSpMVVariant(h->dA, (int)h->spmv_variant));
    h->nonzerostate = A->nonzerostate;
  } else if (!h->dev_newer && imode == ADD_VALUES) { /* the host copy is the current one: bring it over before adding to it */
    PetscCallHIPX(hipxMatUpdateValues(h->dA, a->a));
  }
  PetscCallHIPX(hipxPointerIsDevice(v, &ondev));
  PetscCallHIPX(hipxMatSetValuesCOO(h->dA, coo, v, (int64_t)n, ondev, imode == INSERT_VALUES ? 1 : 0));
  h->dev_newer = PETSC_TRUE;
  PetscCall(PetscObjectStateIncrease((PetscObject)A)); /* like MatSeqAIJRestoreArray: cached diagonals / norms of the block are stale */
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* the remote part of MatSetValuesCOO_MPIAIJ for one block (mpiaij.c:6817-6822): received entries (DEVICE buffer) added onto the
   device values the local part just wrote */
PetscErrorCode MatSeqAIJHIPXAddValuesCOOIndexed_Private(Mat A, hipxCOO coo2, const PetscScalar *d_recv)
{
  Mat_SeqAIJHIPX *h = (Mat_SeqAIJHIPX *)A->spptr;

  PetscFunctionBegin;
  PetscCheck(h->dA && h->dev_newer, PETSC_COMM_SELF, PETSC_ERR_ORDER, "the local part of MatSetValuesCOO has not run on the device");
  PetscCallHIPX(hipxMatAddValuesCOOIndexed(h->dA, coo2, d_recv));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode MatSetValuesCOO_SeqAIJHIPX(Mat A, const PetscScalar v[], InsertMode imode)
{
  Mat_SeqAIJHIPX      *h = (Mat_SeqAIJHIPX *)A->spptr;
  PetscContainer       container;
  MatCOOStruct_SeqAIJ *coo;

  PetscFunctionBegin;
  PetscCheck(h->coo, PETSC_COMM_SELF, PETSC_ERR_ORDER, "MatSetPreallocationCOO() has not been called");
  PetscCall(PetscObjectQuery((PetscObject)A, "__PETSc_MatCOOStruct_Host", (PetscObject *)&container));
  PetscCall(PetscContainerGetPointer(container, &coo));
  PetscCall(MatSeqAIJHIPXSetValuesCOO_Private(A, h->coo, v, coo->n, imode));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* Value-only operations on the device copy: the kernel does the arithmetic, the host array a->a is then refreshed
   by ONE device-to-host copy -- no host pass over the values and no re-upload.  (Leaving the host copy stale, as after
   MatSetValuesCOO with device values, would be unsafe here: MatSetValues_SeqAIJ, MatGetRow_SeqAIJ, MatDuplicateNoCreate_SeqAIJ ...
   read a->a directly, and a CPU build of libpetsc has none of the offload-mask checks the reference's own device types rely
   on.)  MatZeroEntries stays the
   parent's (it is the prelude of a host-side MatSetValues assembly). */
static PetscErrorCode MatSeqAIJHIPXDeviceValuesChanged(Mat A)
{
  Mat_SeqAIJHIPX  *h = (Mat_SeqAIJHIPX *)A->spptr;
  Mat_SeqAIJ      *a = (Mat_SeqAIJ *)A->data;
  PetscObjectState state;

  PetscFunctionBegin;
  PetscCallHIPX(hipxMatGetValues(h->dA, a->a));
  h->dev_newer = PETSC_FALSE;
  /* Host and device hold the same values NOW: give that moment its own object state and record it.  No guess about what the
     caller does next: MatScale()/MatDiagonalScale() bump the state once more after the op (one redundant upload at the next
     product), MatDiagonalScale_MPIAIJ (mpiaij.c:1985-1993) calls the blocks' ops directly and does not -- and any later host
     edit (MatSeqAIJRestoreArray, MatSetValues + assembly ...) moves the state past the recorded one, so it is always uploaded. */
  PetscCall(PetscObjectStateIncrease((PetscObject)A));
  PetscCall(PetscObjectStateGet((PetscObject)A, &state));
  h->valuestate = state;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode MatScale_SeqAIJHIPX(Mat A, PetscScalar alpha) /* MatScale_SeqAIJ aij.c:2604-2617 */
{
  Mat_SeqAIJHIPX *h = (Mat_SeqAIJHIPX *)A->spptr;
  Mat_SeqAIJ     *a = (Mat_SeqAIJ *)A->data;
  hipxMat         dA;

  PetscFunctionBegin;
  PetscCall(MatSeqAIJHIPXGetDeviceMat(A, &dA));
  PetscCallHIPX(hipxMatScale(dA, alpha));
  PetscCall(MatSeqAIJHIPXDeviceValuesChanged(A));
  (void)h;
  PetscCall(PetscLogFlops(a->nz));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode (*parent_axpy_seq)(Mat, PetscScalar, Mat, MatStructure);

/* MatAXPY_SeqAIJ aij.c:2926-2989: SAME_NONZERO_PATTERN is a daxpy over the value arrays -- on the device copies when both matrices
   are of this type; every other structure flag (SUBSET / DIFFERENT / UNKNOWN: pattern work) stays with the parent */
static PetscErrorCode MatAXPY_SeqAIJHIPX(Mat Y, PetscScalar a, Mat X, MatStructure str)
{
  Mat_SeqAIJ *x = (Mat_SeqAIJ *)X->data, *y = (Mat_SeqAIJ *)Y->data;
  hipxMat     dX, dY;

  PetscFunctionBegin;
  if (str != SAME_NONZERO_PATTERN || !MatIsSeqAIJHIPX(X) || !MatIsSeqAIJHIPX(Y) || x->nz != y->nz || X == Y) {
    PetscCall((*parent_axpy_seq)(Y, a, X, str));
    PetscFunctionReturn(PETSC_SUCCESS);
  }
  PetscCall(MatSeqAIJHIPXGetDeviceMat(X, &dX));
  PetscCall(MatSeqAIJHIPXGetDeviceMat(Y, &dY));
  PetscCallHIPX(hipxMatAXPY(dY, a, dX));
  PetscCall(MatSeqAIJHIPXDeviceValuesChanged(Y));
  PetscCall(PetscLogFlops(2.0 * y->nz));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode MatDiagonalScale_SeqAIJHIPX(Mat A, Vec ll, Vec rr) /* MatDiagonalScale_SeqAIJ aij.c:2333-2371 */
{
  Mat_SeqAIJHIPX    *h = (Mat_SeqAIJHIPX *)A->spptr;
  Mat_SeqAIJ        *a = (Mat_SeqAIJ *)A->data;
  hipxMat            dA;
  const PetscScalar *l = NULL, *r = NULL;
  void              *tl = NULL, *tr = NULL;
  PetscInt           m, n;

  PetscFunctionBegin;
  if (ll) {
    PetscCall(VecGetLocalSize(ll, &m));
    PetscCheck(m == A->rmap->n, PETSC_COMM_SELF, PETSC_ERR_ARG_SIZ, "Left scaling vector wrong length");
  }
  if (rr) {
    PetscCall(VecGetLocalSize(rr, &n));
    PetscCheck(n == A->cmap->n, PETSC_COMM_SELF, PETSC_ERR_ARG_SIZ, "Right scaling vector wrong length");
  }
  PetscCall(MatSeqAIJHIPXGetDeviceMat(A, &dA));
  if (ll) PetscCall(VecHIPXGetDeviceRead(ll, &l, &tl));
  if (rr) PetscCall(VecHIPXGetDeviceRead(rr, &r, &tr));
  PetscCallHIPX(hipxMatDiagonalScale(dA, l, r));
  if (rr) PetscCall(VecHIPXRestoreDeviceRead(rr, &r, &tr));
  if (ll) PetscCall(VecHIPXRestoreDeviceRead(ll, &l, &tl));
  PetscCall(MatSeqAIJHIPXDeviceValuesChanged(A));
  (void)h;
  PetscCall(PetscLogFlops((ll ? a->nz : 0) + (rr ? a->nz : 0)));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode MatAssemblyEnd_SeqAIJHIPX(Mat A, MatAssemblyType mode)
{
  Mat_SeqAIJHIPX *h = (Mat_SeqAIJHIPX *)A->spptr;

  PetscFunctionBegin;
  PetscCall((*h->parent_assemblyend)(A, mode)); /* MatAssemblyEnd_SeqAIJ aij.c:1085: compaction, nz, rmax, inode / compressed-row checks */
  /* Round 6: the device half of the set-up at set-up time -- upload + format selection (templates, march plan, inode search ...) here, not inside the first
     MatMult of a timed KSPSolve.  Square matrices only: the off-diagonal block of an MPIAIJ matrix is still being rewritten when its own assembly ends
     (MatSetUpMultiply_MPIAIJ compacts its columns, mmaij.c:8-125): MatAssemblyEnd_MPIAIJHIPX sets both blocks up when the parent has finished.
     -mat_hipx_setup_at_assembly 0: everything at the first product, as in round 5. */
  if (mode == MAT_FINAL_ASSEMBLY && A->rmap->n == A->cmap->n && A->rmap->n > 0) PetscCall(MatSeqAIJHIPXSetUpDevice(A));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode MatDestroy_SeqAIJHIPX(Mat A)
{
  Mat_SeqAIJHIPX *h = (Mat_SeqAIJHIPX *)A->spptr;
  PetscErrorCode (*pdestroy)(Mat) = h->parent_destroy;

  PetscFunctionBegin;
  if (h->dA) PetscCallHIPX(hipxMatDestroy(&h->dA));
  if (h->dAt) PetscCallHIPX(hipxMatDestroy(&h->dAt));
  if (h->coo) PetscCallHIPX(hipxCOODestroy(&h->coo));
  PetscCall(PetscObjectComposeFunction((PetscObject)A, "MatConvert_seqaij_seqaijhipx_C", NULL));
  PetscCall(PetscFree(A->spptr));
  PetscCall((*pdestroy)(A));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode MatConvert_SeqAIJ_SeqAIJHIPX(Mat, MatType, MatReuse, Mat *);

static PetscErrorCode MatDuplicate_SeqAIJHIPX(Mat A, MatDuplicateOption op, Mat *B)
{
  Mat_SeqAIJHIPX *h = (Mat_SeqAIJHIPX *)A->spptr;

  PetscFunctionBegin;
  /* MatDuplicate_SeqAIJ (aij.c:4929-4938) creates B with MatSetType(B, A's type name), i.e. through MatCreate_SeqAIJHIPX:
     B is already complete, its device CSR is uploaded lazily at the first product */
  PetscCall((*h->parent_duplicate)(A, op, B));
  if (!(MatIsSeqAIJHIPX(*B) && (*B)->spptr)) PetscCall(MatConvert_SeqAIJ_SeqAIJHIPX(*B, MATSEQAIJHIPX, MAT_INPLACE_MATRIX, B));
  ((Mat_SeqAIJHIPX *)(*B)->spptr)->spmv_variant = h->spmv_variant;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode MatSetFromOptions_SeqAIJHIPX(Mat A, PetscOptionItems PetscOptionsObject)
{
  Mat_SeqAIJHIPX *h = (Mat_SeqAIJHIPX *)A->spptr;

  PetscFunctionBegin;
  PetscOptionsHeadBegin(PetscOptionsObject, "SeqAIJHIPX options");
  PetscCall(PetscOptionsInt("-mat_aijhipx_spmv_variant", "SpMV kernel geometry / load policy (0 = auto)", "None", h->spmv_variant, &h->spmv_variant, NULL));
  PetscOptionsHeadEnd();
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* in-place conversion of a (possibly assembled) seqaij matrix: install our ops, keep the parent's for everything else */
static PetscErrorCode MatConvert_SeqAIJ_SeqAIJHIPX(Mat A, MatType mtype, MatReuse reuse, Mat *newmat)
{
  Mat             B;
  Mat_SeqAIJHIPX *h;

  PetscFunctionBegin;
  (void)mtype;
  PetscCall(VecHIPXInitRuntime());
  if (reuse == MAT_INITIAL_MATRIX) PetscCall(MatDuplicate(A, MAT_COPY_VALUES, newmat));
  else if (reuse == MAT_REUSE_MATRIX) PetscCall(MatCopy(A, *newmat, SAME_NONZERO_PATTERN));
  B = *newmat;
  if (MatIsSeqAIJHIPX(B) && B->spptr) PetscFunctionReturn(PETSC_SUCCESS);
  PetscCall(PetscNew(&h));
  h->parent_assemblyend = B->ops->assemblyend;
  h->parent_destroy     = B->ops->destroy;
  h->parent_duplicate   = B->ops->duplicate;
  B->spptr              = h;
  B->ops->mult           = MatMult_SeqAIJHIPX;
  B->ops->multadd        = MatMultAdd_SeqAIJHIPX;
  B->ops->multtranspose    = MatMultTranspose_SeqAIJHIPX;
  B->ops->multtransposeadd = MatMultTransposeAdd_SeqAIJHIPX;
  B->ops->getdiagonal    = MatGetDiagonal_SeqAIJHIPX;
  B->ops->sor            = MatSOR_SeqAIJHIPX;
  B->ops->assemblyend    = MatAssemblyEnd_SeqAIJHIPX;
  B->ops->destroy        = MatDestroy_SeqAIJHIPX;
  B->ops->duplicate      = MatDuplicate_SeqAIJHIPX;
  B->ops->setfromoptions = MatSetFromOptions_SeqAIJHIPX;
  if (!parent_axpy_seq) parent_axpy_seq = B->ops->axpy;
  B->ops->axpy           = MatAXPY_SeqAIJHIPX;
  B->ops->scale          = MatScale_SeqAIJHIPX;
  B->ops->diagonalscale  = MatDiagonalScale_SeqAIJHIPX;
  PetscCall(PetscObjectQueryFunction((PetscObject)B, "MatSetPreallocationCOO_C", &h->parent_prealloc_coo));
  PetscCall(PetscObjectComposeFunction((PetscObject)B, "MatSetPreallocationCOO_C", MatSetPreallocationCOO_SeqAIJHIPX));
  PetscCall(PetscObjectComposeFunction((PetscObject)B, "MatSetValuesCOO_C", MatSetValuesCOO_SeqAIJHIPX));
  { /* host routines that read or write the value array go through these (aij.c:4294-4400) */
    Mat_SeqAIJ *sa = (Mat_SeqAIJ *)B->data;
    sa->ops->getarray      = MatSeqAIJGetArray_SeqAIJHIPX;
    sa->ops->getarrayread  = MatSeqAIJGetArrayRead_SeqAIJHIPX;
    sa->ops->getarraywrite = MatSeqAIJGetArrayWrite_SeqAIJHIPX;
  }
  /* MatCreateVecs() hands out VECHIPX vectors, so KSP/PC work vectors live on the device (matrix.c:10069,10080) */
  PetscCall(PetscFree(B->defaultvectype));
  PetscCall(PetscStrallocpy(VECHIPX, &B->defaultvectype));
  PetscCall(PetscObjectChangeTypeName((PetscObject)B, MATSEQAIJHIPX));
  PetscCall(PetscObjectComposeFunction((PetscObject)B, "MatConvert_seqaij_seqaijhipx_C", MatConvert_SeqAIJ_SeqAIJHIPX));
  PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode MatCreate_SeqAIJHIPX(Mat B)
{
  PetscFunctionBegin;
  PetscCall(MatCreate_SeqAIJ(B));
  PetscCall(MatConvert_SeqAIJ_SeqAIJHIPX(B, MATSEQAIJHIPX, MAT_INPLACE_MATRIX, &B));
  PetscFunctionReturn(PETSC_SUCCESS);
}
