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
