/*
 * pchipx.c -- PCJACOBIHIPX: point Jacobi with the set-up done on the device.
 * Mirrors PCJACOBI's default mode (jacobi.c:527-530: PC_JACOBI_DIAGONAL, fixdiag = TRUE, useabs = FALSE):
 *   PCSetUp : diag <- MatGetDiagonal(pmat); VecReciprocal; zeros -> 1     (jacobi.c:205-266)
 *   PCApply : y = x .* diag                                              (jacobi.c:354-362)
 * The stock PCJACOBI also works on HIPX types (its Vec/Mat calls land in our ops); this type only avoids the host
 * round trip of its zero fix-up loop (jacobi.c:255-266 runs VecGetArray on the diagonal).
 */
#include "hipxplugin.h"

typedef struct {
  Vec diag; /* inverse diagonal */
} PC_JacobiHIPX;

static PetscErrorCode PCSetUp_JacobiHIPX(PC pc)
{
  PC_JacobiHIPX *jac = (PC_JacobiHIPX *)pc->data;
  PetscBool      seqhipx = MatIsSeqAIJHIPX(pc->pmat);

  PetscFunctionBegin;
  if (!jac->diag) PetscCall(MatCreateVecs(pc->pmat, &jac->diag, NULL));
  if (seqhipx) { /* one fused kernel */
    hipxMat      dA;
    PetscScalar *d;
    void        *t;
    PetscCall(MatSeqAIJHIPXGetDeviceMat(pc->pmat, &dA));
    PetscCall(VecHIPXGetDeviceWrite(jac->diag, &d, &t));
    PetscCallHIPX(hipxPCJacobiSetUp(dA, d));
    PetscCall(VecHIPXRestoreDeviceWrite(jac->diag, &d, &t));
    PetscCall(PetscObjectStateIncrease((PetscObject)jac->diag));
  } else { /* any other Mat: same steps through the public interface */
    PetscScalar *d;
    void        *t;
    hipx_int     nrep;
    PetscCall(MatGetDiagonal(pc->pmat, jac->diag));
    PetscCall(VecReciprocal(jac->diag));
    PetscCall(VecHIPXGetDeviceReadWrite(jac->diag, &d, &t));
    PetscCallHIPX(hipxVecReplaceZeros(d, jac->diag->map->n, 1.0, &nrep));
    PetscCall(VecHIPXRestoreDeviceWrite(jac->diag, &d, &t));
    if (nrep) PetscCall(PetscInfo(pc, "Zero detected in diagonal of matrix, using 1 at those locations\n"));
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCApply_JacobiHIPX(PC pc, Vec x, Vec y)
{
  PC_JacobiHIPX *jac = (PC_JacobiHIPX *)pc->data;

  PetscFunctionBegin;
  PetscCall(VecPointwiseMult(y, x, jac->diag));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCReset_JacobiHIPX(PC pc)
{
  PC_JacobiHIPX *jac = (PC_JacobiHIPX *)pc->data;

  PetscFunctionBegin;
  PetscCall(VecDestroy(&jac->diag));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCDestroy_JacobiHIPX(PC pc)
{
  PetscFunctionBegin;
  PetscCall(PCReset_JacobiHIPX(pc));
  PetscCall(PetscFree(pc->data));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCView_JacobiHIPX(PC pc, PetscViewer viewer)
{
  PetscBool isascii;

  PetscFunctionBegin;
  (void)pc;
  PetscCall(PetscObjectTypeCompare((PetscObject)viewer, PETSCVIEWERASCII, &isascii));
  if (isascii) PetscCall(PetscViewerASCIIPrintf(viewer, "  type DIAGONAL (device set-up, libhipx)\n"));
  PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode PCCreate_JacobiHIPX(PC pc)
{
  PC_JacobiHIPX *jac;

  PetscFunctionBegin;
  PetscCall(PetscNew(&jac));
  pc->data                     = (void *)jac;
  pc->ops->apply               = PCApply_JacobiHIPX;
  pc->ops->applytranspose      = PCApply_JacobiHIPX;
  pc->ops->setup               = PCSetUp_JacobiHIPX;
  pc->ops->reset               = PCReset_JacobiHIPX;
  pc->ops->destroy             = PCDestroy_JacobiHIPX;
  pc->ops->view                = PCView_JacobiHIPX;
  pc->ops->setfromoptions      = NULL;
  pc->ops->applyrichardson     = NULL;
  pc->ops->applysymmetricleft  = NULL;
  pc->ops->applysymmetricright = NULL;
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* ------------------------------------------------------------------------------------------------------------------------------
 * PCPBJACOBIHIPX ("pbjacobihipx"): PCPBJACOBI whose apply runs on the device.
 * The parent does the set-up (PCSetUp_PBJacobi -> MatInvertBlockDiagonal on the host copy of the matrix, pbjacobi.c:243-304); the
 * inverted blocks are uploaded once per set-up and PCApply / PCApplyTranspose (pbjacobi.c:4-124,126-241: VecGetArray on both
 * vectors, i.e. a device-to-host and a host-to-device copy per application with device vectors) become one kernel.
 */
#include <../src/ksp/pc/impls/pbjacobi/pbjacobi.h>

static PetscErrorCode (*pbj_parent_setup)(PC), (*pbj_parent_destroy)(PC), (*pbj_parent_apply)(PC, Vec, Vec), (*pbj_parent_applytranspose)(PC, Vec, Vec);

static PetscErrorCode PCSetUp_PBJacobiHIPX(PC pc)
{
  PC_PBJacobi *jac = (PC_PBJacobi *)pc->data;
  size_t       bytes;

  PetscFunctionBegin;
  PetscCall((*pbj_parent_setup)(pc));
  PetscCall(VecHIPXInitRuntime());
  if (jac->spptr) PetscCallHIPX(hipxFree(jac->spptr));
  jac->spptr = NULL;
  bytes      = sizeof(PetscScalar) * (size_t)jac->bs * (size_t)jac->bs * (size_t)jac->mbs;
  if (bytes) {
    PetscCallHIPX(hipxMalloc(&jac->spptr, bytes));
    PetscCallHIPX(hipxMemcpyHtoD(jac->spptr, jac->diag, bytes));
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCApply_PBJacobiHIPX_Private(PC pc, Vec x, Vec y, int transpose)
{
  PC_PBJacobi       *jac = (PC_PBJacobi *)pc->data;
  const PetscScalar *xx;
  PetscScalar       *yy;
  void              *tx, *ty;

  PetscFunctionBegin;
  if (!VecIsHIPX(x) || !VecIsHIPX(y) || x == y || (!jac->spptr && jac->mbs)) { /* host vectors: the parent's loops */
    PetscCall((*(transpose ? pbj_parent_applytranspose : pbj_parent_apply))(pc, x, y));
    PetscFunctionReturn(PETSC_SUCCESS);
  }
  PetscCall(VecHIPXGetDeviceRead(x, &xx, &tx));
  PetscCall(VecHIPXGetDeviceWrite(y, &yy, &ty));
  PetscCallHIPX(hipxPCPBJacobiApply((const double *)jac->spptr, (hipx_int)jac->bs, (hipx_int)jac->mbs, xx, yy, transpose));
  PetscCall(VecHIPXRestoreDeviceWrite(y, &yy, &ty));
  PetscCall(VecHIPXRestoreDeviceRead(x, &xx, &tx));
  PetscCall(PetscLogFlops((2.0 * jac->bs * jac->bs - jac->bs) * jac->mbs)); /* pbjacobi.c:122 */
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCApply_PBJacobiHIPX(PC pc, Vec x, Vec y) { return PCApply_PBJacobiHIPX_Private(pc, x, y, 0); }
static PetscErrorCode PCApplyTranspose_PBJacobiHIPX(PC pc, Vec x, Vec y) { return PCApply_PBJacobiHIPX_Private(pc, x, y, 1); }

static PetscErrorCode PCDestroy_PBJacobiHIPX(PC pc)
{
  PC_PBJacobi *jac = (PC_PBJacobi *)pc->data;

  PetscFunctionBegin;
  if (jac->spptr) PetscCallHIPX(hipxFree(jac->spptr));
  jac->spptr = NULL;
  PetscCall((*pbj_parent_destroy)(pc));
  PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode PCCreate_PBJacobiHIPX(PC pc)
{
  PetscFunctionBegin;
  PetscCall(PCSetType(pc, PCPBJACOBI)); /* the parent's creator fills the ops table and allocates PC_PBJacobi (pbjacobi.c:357-392) */
  pbj_parent_setup          = pc->ops->setup;
  pbj_parent_destroy        = pc->ops->destroy;
  pbj_parent_apply          = pc->ops->apply;
  pbj_parent_applytranspose = pc->ops->applytranspose;
  pc->ops->setup            = PCSetUp_PBJacobiHIPX;
  pc->ops->destroy          = PCDestroy_PBJacobiHIPX;
  pc->ops->apply            = PCApply_PBJacobiHIPX;
  pc->ops->applytranspose   = PCApplyTranspose_PBJacobiHIPX;
  PetscFunctionReturn(PETSC_SUCCESS);
}
