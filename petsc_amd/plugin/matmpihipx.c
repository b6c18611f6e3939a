/*
 * matmpihipx.c -- MATMPIAIJHIPX: Mat_MPIAIJ (src/mat/impls/aij/mpi/mpiaij.h:40-74) whose diagonal block A and
 * off-diagonal block B are MATSEQAIJHIPX and whose ghost vector lvec is VECSEQHIPX, so that the parent's
 * MatMult_MPIAIJ / MatMultAdd_MPIAIJ / MatSOR_MPIAIJ / MatGetDiagonal_MPIAIJ (mpiaij.c:1047-1061,1072-1083,1394-1486,
 * 1158-1167) -- which only call a->A->ops->{mult,sor,getdiagonal} and a->B->ops->multadd -- land in the HIP kernels.
 * The reference's own assembly (MatAssemblyEnd_MPIAIJ mpiaij.c:769-846 -> MatSetUpMultiply_MPIAIJ mmaij.c:8-125) builds
 * garray, the compacted B and Mvctx; we convert the pieces right after it (pattern of the reference's device
 * subclasses: mpiaijhipsparse.hip.cxx:154-157, mpiaijcupm.hpp:430-438).
 *
 * Ghost exchange in this build: the stock VecScatter (PetscSF over MPI, host buffers; a host-only libpetsc treats every
 * pointer as host memory, sfpack.c:706-721).  The RCCL/xGMI exchange of libhipx (hipxHalo*, hipxMatMultMPI) is driven by
 * the C host layer (include/hipx_ksp.h) and by bench.py; wiring it under Mvctx is listed as next step in DESIGN.md.
 */
#include "hipxplugin.h"

typedef struct {
  PetscErrorCode (*parent_assemblyend)(Mat, MatAssemblyType);
  PetscErrorCode (*parent_destroy)(Mat);
} Mat_MPIAIJHIPX;

static PetscErrorCode MatAssemblyEnd_MPIAIJHIPX(Mat A, MatAssemblyType mode)
{
  Mat_MPIAIJHIPX *h = (Mat_MPIAIJHIPX *)A->spptr;
  Mat_MPIAIJ     *a;

  PetscFunctionBegin;
  PetscCall((*h->parent_assemblyend)(A, mode));
  a = (Mat_MPIAIJ *)A->data;
  if (mode == MAT_FINAL_ASSEMBLY) {
    if (a->A) PetscCall(MatSetType(a->A, MATSEQAIJHIPX)); /* in-place subclass conversion, matreg.c:146-150 */
    if (a->B) PetscCall(MatSetType(a->B, MATSEQAIJHIPX));
    if (a->lvec && !VecIsHIPX(a->lvec)) { /* same layout, device-capable type; Mvctx only remembers sizes */
      Vec lv;
      PetscCall(MatCreateVecs(a->B, &lv, NULL));
      PetscCall(VecDestroy(&a->lvec));
      a->lvec = lv;
    }
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode MatDestroy_MPIAIJHIPX(Mat A)
{
  Mat_MPIAIJHIPX *h = (Mat_MPIAIJHIPX *)A->spptr;
  PetscErrorCode (*pdestroy)(Mat) = h->parent_destroy;

  PetscFunctionBegin;
  PetscCall(PetscFree(A->spptr));
  PetscCall((*pdestroy)(A));
  PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode MatCreate_MPIAIJHIPX(Mat B)
{
  Mat_MPIAIJHIPX *h;

  PetscFunctionBegin;
  PetscCall(VecHIPXInitRuntime());
  PetscCall(MatCreate_MPIAIJ(B)); /* exported, mpiaij.h:91 */
  PetscCall(PetscNew(&h));
  h->parent_assemblyend = B->ops->assemblyend;
  h->parent_destroy     = B->ops->destroy;
  B->spptr              = h;
  B->ops->assemblyend   = MatAssemblyEnd_MPIAIJHIPX;
  B->ops->destroy       = MatDestroy_MPIAIJHIPX;
  PetscCall(PetscFree(B->defaultvectype));
  PetscCall(PetscStrallocpy(VECHIPX, &B->defaultvectype));
  PetscCall(PetscObjectChangeTypeName((PetscObject)B, MATMPIAIJHIPX));
  PetscFunctionReturn(PETSC_SUCCESS);
}
