/*
 * matmpihipx.c -- MATMPIAIJHIPX: Mat_MPIAIJ (src/mat/impls/aij/mpi/mpiaij.h:40-74) whose diagonal block A and
 * off-diagonal block B are MATSEQAIJHIPX and whose ghost vector lvec is VECSEQHIPX, so that the parent's
 * MatMult_MPIAIJ / MatMultAdd_MPIAIJ / MatSOR_MPIAIJ / MatGetDiagonal_MPIAIJ (mpiaij.c:1047-1061,1072-1083,1394-1486,
 * 1158-1167) -- which only call a->A->ops->{mult,sor,getdiagonal} and a->B->ops->multadd -- land in the HIP kernels.
 * The reference's own assembly (MatAssemblyEnd_MPIAIJ mpiaij.c:769-846 -> MatSetUpMultiply_MPIAIJ mmaij.c:8-125) builds
 * garray, the compacted B and Mvctx; we convert the pieces right after it (pattern of the reference's device
 * subclasses: mpiaijhipsparse.hip.cxx:154-157, mpiaijcupm.hpp:430-438).
 *
 * MatMult / MatMultAdd run the ghost exchange ON THE DEVICE (libhipx hipxMatMultMPI: RCCL send/recv over xGMI, or peer stores
 * through HIP IPC mappings) with the plan the reference built in a->Mvctx; the stock VecScatter on host buffers (a host-only
 * libpetsc treats every pointer as host memory, sfpack.c:706-721) remains as the fall-back and for every other user of Mvctx.
 */
#include "hipxplugin.h"
#include <petscsf.h>

/* Ghost exchange of MatMult / MatMultAdd: the reference's own plan (a->Mvctx, a PetscSF built by MatSetUpMultiply_MPIAIJ,
   mmaij.c:108-125: roots = owned entries of x, leaves = lvec) is read back with PetscSFGetRootRanks / PetscSFGetLeafRanks and
   handed to libhipx, which runs it on the device:
     transport "rccl": ncclSend / ncclRecv over xGMI on the comm stream (one rank per GPU),
     transport "ipc" : peer stores through HIP IPC mappings (also when several ranks share a GPU, which RCCL refuses),
     transport "host": the stock VecScatter on host buffers (PetscSF over MPI) -- the parent's MatMult_MPIAIJ, kept as fall-back.
     "sf"            : Mvctx is re-typed to the PetscSF type hipx (sfhipx.c) and MatMult calls PetscSFBcastWithMemTypeBegin / End on
                       device pointers: the same transports behind the reference's own scatter interface.
   -mat_mpiaijhipx_halo <auto|rccl|ipc|host|sf> (or HIPX_HALO); auto = rccl when every rank drives its own device, else ipc. */
typedef struct {
  PetscErrorCode (*parent_assemblyend)(Mat, MatAssemblyType);
  PetscErrorCode (*parent_destroy)(Mat);
  PetscErrorCode (*parent_mult)(Mat, Vec, Vec);
  PetscErrorCode (*parent_multadd)(Mat, Vec, Vec, Vec);
  PetscErrorCode (*parent_prealloc_coo)(Mat, PetscCount, PetscInt[], PetscInt[]);
  PetscErrorCode (*parent_setvalues_coo)(Mat, const PetscScalar[], InsertMode);
  hipxCOO          cooA, cooB; /* device copies of Ajmap1/Aperm1 and Bjmap1/Bperm1 (MatCOOStruct_MPIAIJ, mpiaij.h:62-89) */
  PetscBool        coo_local;  /* MatSetValuesCOO runs on the device (both blocks are seqaijhipx on every rank) */
  /* entries that travel between ranks (mpiaij.c:6798-6822): packed, exchanged and added on the device */
  PetscBool        coo_travel;
  hipxCOO          cooA2, cooB2;    /* Aimap2/Ajmap2/Aperm2, Bimap2/Bjmap2/Bperm2 */
  PetscSF          coosf;           /* coo->sf's graph with the PetscSF type hipx: PetscSFReduce on device buffers */
  hipx_int        *d_cperm;         /* Cperm1 on the device */
  PetscScalar     *d_send, *d_recv; /* coo->sendlen / coo->recvlen scalars */
  PetscScalar     *d_v;             /* staging for a host-resident value array (coo->n scalars) */
  hipxHalo         halo;
  PetscSF          sf;        /* transport 3: a PetscSF of type hipx with Mvctx's graph */
  PetscInt         transport; /* 0 host, 1 ipc, 2 rccl, 3 PetscSF type hipx */
  PetscObjectState nzstate;   /* nonzero state the plan was built from */
} Mat_MPIAIJHIPX;


static PetscErrorCode MatMPIAIJHIPXBuildHalo(Mat A)
{
  Mat_MPIAIJHIPX    *h = (Mat_MPIAIJHIPX *)A->spptr;
  Mat_MPIAIJ        *a = (Mat_MPIAIJ *)A->data;
  MPI_Comm           comm = PetscObjectComm((PetscObject)A);
  PetscMPIInt        rank, size, nr, ni;
  const PetscMPIInt *ranks, *iranks;
  const PetscInt    *roff, *rmine, *rremote, *ioff, *iroot;
  char               want[16] = "auto";
  const char        *env      = getenv("HIPX_HALO");
  PetscBool          contiguous = PETSC_TRUE, allok;
  PetscInt           transport;

  PetscFunctionBegin;
  if (h->halo) PetscCallHIPX(hipxHaloDestroy(&h->halo));
  PetscCall(PetscSFDestroy(&h->sf));
  h->transport = 0;
  h->nzstate   = A->nonzerostate;
  PetscCallMPI(MPI_Comm_rank(comm, &rank));
  PetscCallMPI(MPI_Comm_size(comm, &size));
  if (env) PetscCall(PetscStrncpy(want, env, sizeof(want)));
  PetscCall(PetscOptionsGetString(((PetscObject)A)->options, ((PetscObject)A)->prefix, "-mat_mpiaijhipx_halo", want, sizeof(want), NULL));
  if (size == 1 || !a->Mvctx || !strcmp(want, "host")) PetscFunctionReturn(PETSC_SUCCESS);
  if (!strcmp(want, "sf")) { /* the exchange through the PetscSF interface: Mvctx becomes a PetscSF of type hipx (sfhipx.c), MatMult hands it
                                device pointers (PetscSFBcastWithMemTypeBegin / End around the diagonal-block product, mpiaij.c:1056-1059) */
    PetscCall(PetscSFDestroy(&h->sf));
    PetscCall(PetscSFDuplicate(a->Mvctx, PETSCSF_DUPLICATE_GRAPH, &h->sf)); /* the same star forest (roots = owned x, leaves = lvec), not set up yet ... */
    PetscCall(PetscSFSetType(h->sf, PETSCSFHIPX));                           /* ... so its type can still be chosen (Mvctx itself stays the host fall-back) */
    PetscCall(PetscSFSetUp(h->sf));
    h->transport = 3;
    PetscCall(PetscInfo(A, "MATMPIAIJHIPX ghost exchange through PetscSF type hipx\n"));
    PetscFunctionReturn(PETSC_SUCCESS);
  }
  { /* the device exchange is bootstrapped over PETSC_COMM_WORLD ranks (one RCCL communicator per process) */
    int cmp;
    PetscCallMPI(MPI_Comm_compare(comm, PETSC_COMM_WORLD, &cmp));
    if (cmp != MPI_IDENT && cmp != MPI_CONGRUENT) PetscFunctionReturn(PETSC_SUCCESS);
  }
  PetscCall(PetscSFSetUp(a->Mvctx));
  PetscCall(PetscSFGetRootRanks(a->Mvctx, &nr, &ranks, &roff, &rmine, &rremote));
  PetscCall(PetscSFGetLeafRanks(a->Mvctx, &ni, &iranks, &ioff, &iroot));
  /* lvec[k] <-> garray[k] (mmaij.c:108-117) and garray is sorted, so the leaves a rank sends are one contiguous run of lvec */
  for (PetscMPIInt k = 0; k < nr && contiguous; k++)
    for (PetscInt j = roff[k]; j < roff[k + 1]; j++)
      if (rmine && rmine[j] != j) {
        contiguous = PETSC_FALSE;
        break;
      }
  PetscCallMPI(MPIU_Allreduce(&contiguous, &allok, 1, MPIU_BOOL, MPI_LAND, comm));
  if (!allok) PetscFunctionReturn(PETSC_SUCCESS);
  {
    int      *sr, *rr;
    hipx_int *so, *si, *ro;
    PetscCall(PetscMalloc5(ni + 1, &sr, ni + 2, &so, ioff[ni] + 1, &si, nr + 1, &rr, nr + 2, &ro));
    for (PetscMPIInt k = 0; k < ni; k++) sr[k] = (int)iranks[k];
    for (PetscMPIInt k = 0; k <= ni; k++) so[k] = (hipx_int)ioff[k];
    for (PetscInt j = 0; j < ioff[ni]; j++) si[j] = (hipx_int)iroot[j];
    for (PetscMPIInt k = 0; k < nr; k++) rr[k] = (int)ranks[k];
    for (PetscMPIInt k = 0; k <= nr; k++) ro[k] = (hipx_int)roff[k];
    PetscCallHIPX(hipxHaloCreate((int)ni, sr, so, si, (int)nr, rr, ro, &h->halo));
    PetscCall(PetscFree5(sr, so, si, rr, ro));
  }
  PetscCall(HipxHaloBringUp(comm, (PetscObject)A, &h->halo, want, &transport));
  if (!transport) PetscFunctionReturn(PETSC_SUCCESS); /* back to the reference's own host scatter */
  h->transport = transport;
  PetscCall(PetscInfo(A, "MATMPIAIJHIPX ghost exchange on the device: transport %s, %d send / %d receive neighbours\n", transport == 2 ? "rccl" : "ipc", (int)ni, (int)nr));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* MatMult_MPIAIJ mpiaij.c:1047-1061 / MatMultAdd_MPIAIJ mpiaij.c:1072-1083 with the exchange on the device: pack + send on the
   comm stream while the diagonal block multiplies on the compute stream, then the off-diagonal block accumulates */
static PetscErrorCode MatMultAdd_MPIAIJHIPX_Private(Mat A, Vec xx, Vec yy, Vec zz)
{
  Mat_MPIAIJHIPX    *h = (Mat_MPIAIJHIPX *)A->spptr;
  Mat_MPIAIJ        *a = (Mat_MPIAIJ *)A->data;
  hipxMat            dA, dB;
  const PetscScalar *x, *y = NULL;
  PetscScalar       *z, *lv;
  void              *tx, *ty = NULL, *tz, *tl;

  PetscFunctionBegin;
  PetscCall(MatSeqAIJHIPXGetDeviceMat(a->A, &dA));
  PetscCall(MatSeqAIJHIPXGetDeviceMat(a->B, &dB));
  PetscCall(VecHIPXGetDeviceRead(xx, &x, &tx));
  PetscCall(VecHIPXGetDeviceWrite(a->lvec, &lv, &tl));
  if (h->transport == 3) { /* mpiaij.c:1056-1059 / 1078-1081 with the scatter = PetscSF hipx on device buffers */
    PetscCall(PetscSFBcastWithMemTypeBegin(h->sf, MPIU_SCALAR, PETSC_MEMTYPE_HIP, x, PETSC_MEMTYPE_HIP, lv, MPI_REPLACE));
    if (!yy) {
      PetscCall(VecHIPXGetDeviceWrite(zz, &z, &tz));
      PetscCallHIPX(hipxMatMult(dA, x, z));
    } else if (zz == yy) {
      PetscCall(VecHIPXGetDeviceReadWrite(zz, &z, &tz));
      PetscCallHIPX(hipxMatMultAdd(dA, x, z, z));
    } else {
      PetscCall(VecHIPXGetDeviceRead(yy, &y, &ty));
      PetscCall(VecHIPXGetDeviceWrite(zz, &z, &tz));
      PetscCallHIPX(hipxMatMultAdd(dA, x, y, z));
      PetscCall(VecHIPXRestoreDeviceRead(yy, &y, &ty));
    }
    PetscCall(PetscSFBcastEnd(h->sf, MPIU_SCALAR, x, lv, MPI_REPLACE));
    PetscCallHIPX(hipxMatMultAdd(dB, lv, z, z));
  } else if (!yy) {
    PetscCall(VecHIPXGetDeviceWrite(zz, &z, &tz));
    PetscCallHIPX(hipxMatMultMPI(dA, dB, h->halo, x, lv, z));
  } else {
    if (zz == yy) {
      PetscCall(VecHIPXGetDeviceReadWrite(zz, &z, &tz));
      y = z;
    } else {
      PetscCall(VecHIPXGetDeviceRead(yy, &y, &ty));
      PetscCall(VecHIPXGetDeviceWrite(zz, &z, &tz));
    }
    PetscCallHIPX(hipxMatMultAddMPI(dA, dB, h->halo, x, lv, y, z));
    if (zz != yy) PetscCall(VecHIPXRestoreDeviceRead(yy, &y, &ty));
  }
  PetscCall(VecHIPXRestoreDeviceWrite(zz, &z, &tz));
  PetscCall(VecHIPXRestoreDeviceWrite(a->lvec, &lv, &tl));
  PetscCall(VecHIPXRestoreDeviceRead(xx, &x, &tx));
  {
    Mat_SeqAIJ *sa = (Mat_SeqAIJ *)a->A->data, *sb = (Mat_SeqAIJ *)a->B->data;
    PetscCall(PetscLogFlops(yy ? 2.0 * sa->nz + 2.0 * sb->nz : 2.0 * sa->nz - sa->nonzerorowcnt + 2.0 * sb->nz)); /* aij.c:1497,1653 */
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode MatMult_MPIAIJHIPX(Mat, Vec, Vec);

static PetscBool MatMPIAIJHIPXDevicePath(Mat A, Vec xx, Vec zz)
{
  Mat_MPIAIJHIPX *h = (Mat_MPIAIJHIPX *)A->spptr;
  Mat_MPIAIJ     *a = (Mat_MPIAIJ *)A->data;
  return (PetscBool)(h->transport && (h->halo || h->transport == 3) && h->nzstate == A->nonzerostate && a->A && a->B && MatIsSeqAIJHIPX(a->A) && MatIsSeqAIJHIPX(a->B) && a->lvec && VecIsHIPX(a->lvec) && VecIsHIPX(xx) && VecIsHIPX(zz));
}

/* for KSPCGHIPX: the device pieces of an assembled MATMPIAIJHIPX whose ghost exchange runs on the device (NULL halo otherwise) */
PetscErrorCode MatMPIAIJHIPXGetDevice(Mat A, hipxMat *dA, hipxMat *dB, hipxHalo *halo, Vec *lvec)
{
  Mat_MPIAIJHIPX *h = (Mat_MPIAIJHIPX *)A->spptr;
  Mat_MPIAIJ     *a = (Mat_MPIAIJ *)A->data;

  PetscFunctionBegin;
  *halo = NULL;
  if (A->ops->mult == MatMult_MPIAIJHIPX && h->transport && h->halo && h->nzstate == A->nonzerostate && a->A && a->B && MatIsSeqAIJHIPX(a->A) && MatIsSeqAIJHIPX(a->B) && a->lvec && VecIsHIPX(a->lvec) &&
      HipxCommIsUp()) {
    PetscCall(MatSeqAIJHIPXGetDeviceMat(a->A, dA));
    PetscCall(MatSeqAIJHIPXGetDeviceMat(a->B, dB));
    *halo = h->halo;
    *lvec = a->lvec;
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode MatMult_MPIAIJHIPX(Mat A, Vec xx, Vec yy)
{
  Mat_MPIAIJHIPX *h = (Mat_MPIAIJHIPX *)A->spptr;

  PetscFunctionBegin;
  if (MatMPIAIJHIPXDevicePath(A, xx, yy)) PetscCall(MatMultAdd_MPIAIJHIPX_Private(A, xx, NULL, yy));
  else PetscCall((*h->parent_mult)(A, xx, yy));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode MatMultAdd_MPIAIJHIPX(Mat A, Vec xx, Vec yy, Vec zz)
{
  Mat_MPIAIJHIPX *h = (Mat_MPIAIJHIPX *)A->spptr;

  PetscFunctionBegin;
  if (MatMPIAIJHIPXDevicePath(A, xx, zz) && VecIsHIPX(yy)) PetscCall(MatMultAdd_MPIAIJHIPX_Private(A, xx, yy, zz));
  else PetscCall((*h->parent_multadd)(A, xx, yy, zz));
  PetscFunctionReturn(PETSC_SUCCESS);
}

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
    if (!h->halo || h->nzstate != A->nonzerostate) PetscCall(MatMPIAIJHIPXBuildHalo(A)); /* collective: every rank assembles */
    /* the blocks are final now (MatSetUpMultiply_MPIAIJ has compacted B's columns): their device copies and formats at set-up time (round 6) */
    if (a->A && MatIsSeqAIJHIPX(a->A)) PetscCall(MatSeqAIJHIPXSetUpDevice(a->A));
    if (a->B && MatIsSeqAIJHIPX(a->B)) PetscCall(MatSeqAIJHIPXSetUpDevice(a->B));
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* COO assembly.  The parent builds the pattern, the blocks (already seqaijhipx: it converts them to the root type,
   mpiaij.c:6725-6726) and the maps; when no entry has to travel between ranks -- every rank sets its own rows, as
   bench_kspsolve.c:301-302 does -- MatSetValuesCOO is two applications of the device kernel, one per block, and the values never
   touch the host.  Otherwise the parent's host path (PetscSFReduce of the remote entries) runs as before. */
static PetscErrorCode MatMPIAIJHIPXResetCOO(Mat_MPIAIJHIPX *h)
{
  PetscFunctionBegin;
  if (h->cooA) PetscCallHIPX(hipxCOODestroy(&h->cooA));
  if (h->cooB) PetscCallHIPX(hipxCOODestroy(&h->cooB));
  if (h->cooA2) PetscCallHIPX(hipxCOODestroy(&h->cooA2));
  if (h->cooB2) PetscCallHIPX(hipxCOODestroy(&h->cooB2));
  PetscCall(PetscSFDestroy(&h->coosf));
  if (h->d_cperm) PetscCallHIPX(hipxFree(h->d_cperm));
  if (h->d_send) PetscCallHIPX(hipxFree(h->d_send));
  if (h->d_recv) PetscCallHIPX(hipxFree(h->d_recv));
  if (h->d_v) PetscCallHIPX(hipxFree(h->d_v));
  h->d_cperm = NULL;
  h->d_send = h->d_recv = h->d_v = NULL;
  h->coo_local = h->coo_travel = PETSC_FALSE;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode MatSetPreallocationCOO_MPIAIJHIPX(Mat A, PetscCount n, PetscInt coo_i[], PetscInt coo_j[])
{
  Mat_MPIAIJHIPX      *h = (Mat_MPIAIJHIPX *)A->spptr;
  Mat_MPIAIJ          *a;
  PetscContainer       container;
  MatCOOStruct_MPIAIJ *coo;
  PetscMPIInt          mine[2], all[2];

  PetscFunctionBegin;
  PetscCall(MatMPIAIJHIPXResetCOO(h));
  PetscCall((*h->parent_prealloc_coo)(A, n, coo_i, coo_j));
  a = (Mat_MPIAIJ *)A->data;
  PetscCall(PetscObjectQuery((PetscObject)A, "__PETSc_MatCOOStruct_Host", (PetscObject *)&container));
  PetscCall(PetscContainerGetPointer(container, &coo));
  /* [0]: this rank can run the device path (MIN over ranks); [1]: some entry travels between ranks (MAX over ranks, sent negated) */
  mine[0] = (MatIsSeqAIJHIPX(a->A) && MatIsSeqAIJHIPX(a->B) && coo->n < (PetscCount)PETSC_INT32_MAX && !getenv("HIPX_COO_HOST")) ? 1 : 0;
  mine[1] = (coo->sendlen == 0 && coo->recvlen == 0 && coo->Annz2 == 0 && coo->Bnnz2 == 0) ? 1 : 0;
  PetscCallMPI(MPIU_Allreduce(mine, all, 2, MPI_INT, MPI_MIN, PetscObjectComm((PetscObject)A)));
  if (all[0]) {
    PetscCallHIPX(hipxCOOCreate((int64_t)coo->Annz, (const int64_t *)coo->Ajmap1, (int64_t)coo->Atot1, (const int64_t *)coo->Aperm1, &h->cooA));
    PetscCallHIPX(hipxCOOCreate((int64_t)coo->Bnnz, (const int64_t *)coo->Bjmap1, (int64_t)coo->Btot1, (const int64_t *)coo->Bperm1, &h->cooB));
    h->coo_local = PETSC_TRUE;
    if (!all[1]) { /* entries travel: the remote maps, the device buffers and coo->sf's graph in a PetscSF of type hipx (sfhipx.c) */
      hipx_int *cp;
      PetscCallHIPX(hipxCOOCreateIndexed((int64_t)coo->Annz2, (const int64_t *)coo->Aimap2, (const int64_t *)coo->Ajmap2, (int64_t)coo->Atot2, (const int64_t *)coo->Aperm2, &h->cooA2));
      PetscCallHIPX(hipxCOOCreateIndexed((int64_t)coo->Bnnz2, (const int64_t *)coo->Bimap2, (const int64_t *)coo->Bjmap2, (int64_t)coo->Btot2, (const int64_t *)coo->Bperm2, &h->cooB2));
      PetscCall(PetscMalloc1(coo->sendlen + 1, &cp));
      for (PetscInt i = 0; i < coo->sendlen; i++) cp[i] = (hipx_int)coo->Cperm1[i];
      PetscCallHIPX(hipxMalloc((void **)&h->d_cperm, sizeof(hipx_int) * (size_t)(coo->sendlen + 1)));
      PetscCallHIPX(hipxMemcpyHtoD(h->d_cperm, cp, sizeof(hipx_int) * (size_t)coo->sendlen));
      PetscCall(PetscFree(cp));
      PetscCallHIPX(hipxMalloc((void **)&h->d_send, sizeof(PetscScalar) * (size_t)(coo->sendlen + 1)));
      PetscCallHIPX(hipxMalloc((void **)&h->d_recv, sizeof(PetscScalar) * (size_t)(coo->recvlen + 1)));
      PetscCall(PetscSFDuplicate(coo->sf, PETSCSF_DUPLICATE_GRAPH, &h->coosf)); /* (re-typing a set-up SF in place is not supported: sf.c) */
      PetscCall(PetscSFSetType(h->coosf, PETSCSFHIPX));
      PetscCall(PetscSFSetUp(h->coosf));
      h->coo_travel = PETSC_TRUE;
    }
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode MatSetValuesCOO_MPIAIJHIPX(Mat A, const PetscScalar v[], InsertMode imode)
{
  Mat_MPIAIJHIPX      *h = (Mat_MPIAIJHIPX *)A->spptr;
  Mat_MPIAIJ          *a = (Mat_MPIAIJ *)A->data;
  PetscContainer       container;
  MatCOOStruct_MPIAIJ *coo;
  const PetscScalar   *dv = v;

  PetscFunctionBegin;
  if (!h->coo_local) {
    PetscCall((*h->parent_setvalues_coo)(A, v, imode));
    PetscFunctionReturn(PETSC_SUCCESS);
  }
  PetscCall(PetscObjectQuery((PetscObject)A, "__PETSc_MatCOOStruct_Host", (PetscObject *)&container));
  PetscCheck(container, PetscObjectComm((PetscObject)A), PETSC_ERR_PLIB, "Not found MatCOOStruct on this matrix");
  PetscCall(PetscContainerGetPointer(container, &coo));
  if (getenv("HIPX_TRACE_COO"))
    fprintf(stderr, "[petschipx] MatSetValuesCOO_MPIAIJHIPX: device path, %lld + %lld nonzeros%s\n", (long long)coo->Annz, (long long)coo->Bnnz, h->coo_travel ? ", entries travel between ranks" : "");
  if (h->coo_travel) {
    int ondev = 0;
    PetscCallHIPX(hipxPointerIsDevice(v, &ondev));
    if (!ondev && coo->n) { /* one upload serves the pack and both blocks */
      if (!h->d_v) PetscCallHIPX(hipxMalloc((void **)&h->d_v, sizeof(PetscScalar) * (size_t)coo->n));
      PetscCallHIPX(hipxMemcpyHtoD(h->d_v, v, sizeof(PetscScalar) * (size_t)coo->n));
      dv = h->d_v;
    }
    /* mpiaij.c:6798-6801: pack the entries other ranks own, start sending them to their owners */
    if (coo->sendlen) PetscCallHIPX(hipxVecScatterIndexed(dv, h->d_cperm, h->d_send, NULL, (hipx_int)coo->sendlen, 0));
    PetscCall(PetscSFReduceWithMemTypeBegin(h->coosf, MPIU_SCALAR, PETSC_MEMTYPE_HIP, h->d_send, PETSC_MEMTYPE_HIP, h->d_recv, MPI_REPLACE));
  }
  PetscCall(MatSeqAIJHIPXSetValuesCOO_Private(a->A, h->cooA, dv, coo->n, imode)); /* mpiaij.c:6804-6808 */
  PetscCall(MatSeqAIJHIPXSetValuesCOO_Private(a->B, h->cooB, dv, coo->n, imode)); /* mpiaij.c:6809-6813 */
  if (h->coo_travel) {
    PetscCall(PetscSFReduceEnd(h->coosf, MPIU_SCALAR, h->d_send, h->d_recv, MPI_REPLACE)); /* mpiaij.c:6814 */
    PetscCall(MatSeqAIJHIPXAddValuesCOOIndexed_Private(a->A, h->cooA2, h->d_recv));        /* mpiaij.c:6817-6819 */
    PetscCall(MatSeqAIJHIPXAddValuesCOOIndexed_Private(a->B, h->cooB2, h->d_recv));        /* mpiaij.c:6820-6822 */
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode MatDestroy_MPIAIJHIPX(Mat A)
{
  Mat_MPIAIJHIPX *h = (Mat_MPIAIJHIPX *)A->spptr;
  PetscErrorCode (*pdestroy)(Mat) = h->parent_destroy;

  PetscFunctionBegin;
  if (h->halo) PetscCallHIPX(hipxHaloDestroy(&h->halo));
  PetscCall(PetscSFDestroy(&h->sf));
  PetscCall(MatMPIAIJHIPXResetCOO(h));
  PetscCall(PetscObjectComposeFunction((PetscObject)A, "MatSetPreallocationCOO_C", NULL));
  PetscCall(PetscObjectComposeFunction((PetscObject)A, "MatSetValuesCOO_C", NULL));
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
  h->parent_mult        = B->ops->mult;
  h->parent_multadd     = B->ops->multadd;
  B->spptr              = h;
  B->ops->assemblyend   = MatAssemblyEnd_MPIAIJHIPX;
  B->ops->destroy       = MatDestroy_MPIAIJHIPX;
  B->ops->mult          = MatMult_MPIAIJHIPX;
  B->ops->multadd       = MatMultAdd_MPIAIJHIPX;
  PetscCall(PetscObjectQueryFunction((PetscObject)B, "MatSetPreallocationCOO_C", &h->parent_prealloc_coo));
  PetscCall(PetscObjectQueryFunction((PetscObject)B, "MatSetValuesCOO_C", &h->parent_setvalues_coo));
  PetscCall(PetscObjectComposeFunction((PetscObject)B, "MatSetPreallocationCOO_C", MatSetPreallocationCOO_MPIAIJHIPX));
  PetscCall(PetscObjectComposeFunction((PetscObject)B, "MatSetValuesCOO_C", MatSetValuesCOO_MPIAIJHIPX));
  PetscCall(PetscFree(B->defaultvectype));
  PetscCall(PetscStrallocpy(VECHIPX, &B->defaultvectype));
  PetscCall(PetscObjectChangeTypeName((PetscObject)B, MATMPIAIJHIPX));
  PetscFunctionReturn(PETSC_SUCCESS);
}
