/*
 * commhipx.c -- bringing a libhipx ghost-exchange plan (hipxHalo) up on a transport, shared by MATMPIAIJHIPX (Mvctx of MatMult)
 * and the PetscSF type "hipx".  Transports: 2 = RCCL send/recv over xGMI (one rank per GPU), 1 = IPC peer stores (any
 * rank-to-GPU mapping of one node), 0 = none (the caller keeps the reference's host path).  Collective on comm; a failure on any
 * rank makes every rank fall back together: RCCL -> IPC -> 0.
 */
#include "hipxplugin.h"

static PetscBool hipx_rccl_up = PETSC_FALSE, hipx_ipc_comm_up = PETSC_FALSE;

PetscBool HipxCommIsUp(void) { return (PetscBool)(hipx_rccl_up || hipx_ipc_comm_up); }

/* want: "auto" | "rccl" | "ipc".  *halo is destroyed (set to NULL) when no transport comes up. */
PetscErrorCode HipxHaloBringUp(MPI_Comm comm, PetscObject obj, hipxHalo *halo, const char *want, PetscInt *transport_out)
{
  PetscMPIInt rank, size;
  PetscInt    transport;

  PetscFunctionBegin;
  PetscCallMPI(MPI_Comm_rank(comm, &rank));
  PetscCallMPI(MPI_Comm_size(comm, &size));
  if (!strcmp(want, "rccl")) transport = 2;
  else if (!strcmp(want, "ipc")) transport = 1;
  else { /* auto: RCCL needs one device per rank.  Compare the real identity of the device each rank drives (PCI bus id) within
            a host: ordinals say nothing when the launcher binds one GPU per rank through HIP/ROCR_VISIBLE_DEVICES */
    unsigned long long uid = 0, *all;
    char               host[MPI_MAX_PROCESSOR_NAME];
    int                hl = 0;
    unsigned long long key = 5381;
    PetscBool          shared = PETSC_FALSE;
    PetscCallHIPX(hipxDeviceUID(&uid));
    PetscCallMPI(MPI_Get_processor_name(host, &hl));
    for (int c = 0; c < hl; c++) key = key * 33u + (unsigned char)host[c];
    PetscCall(PetscMalloc1(2 * (size_t)size, &all));
    {
      unsigned long long mine[2] = {key, uid};
      PetscCallMPI(MPI_Allgather(mine, 2, MPI_UNSIGNED_LONG_LONG, all, 2, MPI_UNSIGNED_LONG_LONG, comm));
    }
    for (int p = 0; p < size && !shared; p++)
      for (int q = p + 1; q < size; q++)
        if (all[2 * p] == all[2 * q] && all[2 * p + 1] == all[2 * q + 1]) {
          shared = PETSC_TRUE;
          break;
        }
    PetscCall(PetscFree(all));
    if (!shared && hipx_ipc_comm_up) shared = PETSC_TRUE; /* an IPC communicator is already up in this process: stay on it (hipxCommInit would refuse) */
    transport = shared ? 1 : 2;
  }
  /* Bring the transport up; a failure on ANY rank (RCCL bootstrap, hipIpcOpenMemHandle to a GPU this process cannot map, ...)
     makes every rank fall back together: RCCL -> IPC peer stores -> the stock host PetscSF scatter (transport 0). */
  for (;;) {
    int ierr = 0, anyerr = 0;
    if (transport == 2) {
      if (!hipx_rccl_up) { /* ncclUniqueId of rank 0 travels over MPI */
        char id[HIPX_COMM_ID_BYTES];
        memset(id, 0, sizeof(id));
        int  iderr = 0; /* rank 0's hipxCommGetUniqueId result travels with the id: if it failed, NO rank may enter ncclCommInitRank (the others would wait for rank 0 for ever) */
        if (!rank) iderr = hipxCommGetUniqueId(id);
        PetscCallMPI(MPI_Bcast(id, HIPX_COMM_ID_BYTES, MPI_BYTE, 0, comm));
        PetscCallMPI(MPI_Bcast(&iderr, 1, MPI_INT, 0, comm));
        if (iderr) ierr = iderr;
        else if (hipx_ipc_comm_up) ierr = HIPX_ERR_ORDER;
        else ierr = hipxCommInit(id, (int)rank, (int)size);
      }
    } else {
      char *mine, *all;
      PetscCall(PetscMalloc2(HIPX_HALO_IPC_BLOB_BYTES, &mine, (size_t)HIPX_HALO_IPC_BLOB_BYTES * size, &all));
      memset(mine, 0, HIPX_HALO_IPC_BLOB_BYTES);
      ierr = hipxHaloIpcExport((*halo), (int)rank, (int)size, mine);
      PetscCallMPI(MPI_Allgather(mine, HIPX_HALO_IPC_BLOB_BYTES, MPI_BYTE, all, HIPX_HALO_IPC_BLOB_BYTES, MPI_BYTE, comm));
      PetscCallMPI(MPI_Allreduce(&ierr, &anyerr, 1, MPI_INT, MPI_MAX, comm));
      if (!anyerr) ierr = hipxHaloIpcAttach((*halo), all);
      PetscCall(PetscFree2(mine, all));
      PetscCallMPI(MPI_Allreduce(&ierr, &anyerr, 1, MPI_INT, MPI_MAX, comm));
      if (!anyerr && !hipx_rccl_up && !hipx_ipc_comm_up) { /* scalar all-reduces of the fused solver (cghipx) through the same IPC machinery */
        char hmine[64], *hall;
        PetscCall(PetscMalloc1((size_t)64 * size, &hall));
        memset(hmine, 0, sizeof(hmine));
        ierr = hipxCommIpcExport((int)rank, (int)size, hmine);
        PetscCallMPI(MPI_Allgather(hmine, 64, MPI_BYTE, hall, 64, MPI_BYTE, comm));
        PetscCallMPI(MPI_Allreduce(&ierr, &anyerr, 1, MPI_INT, MPI_MAX, comm));
        if (!anyerr) ierr = hipxCommIpcAttach(hall);
        PetscCall(PetscFree(hall));
        PetscCallMPI(MPI_Allreduce(&ierr, &anyerr, 1, MPI_INT, MPI_MAX, comm));
        if (!anyerr) hipx_ipc_comm_up = PETSC_TRUE;
        else anyerr = 0; /* the ghost exchange itself is up; only cghipx's device all-reduce is not (it then reduces through MPI) */
        ierr = 0;
      }
    }
    PetscCallMPI(MPI_Allreduce(&ierr, &anyerr, 1, MPI_INT, MPI_MAX, comm));
    if (!anyerr) {
      if (transport == 2) hipx_rccl_up = PETSC_TRUE;
      break;
    }
    PetscCall(PetscInfo(obj, "hipx ghost exchange: transport %s could not be brought up on every rank (code %d: %s)\n", transport == 2 ? "rccl" : "ipc", anyerr, ierr ? hipxGetErrorString() : "another rank failed"));
    if (transport == 2 && !strcmp(want, "auto")) {
      transport = 1;
      continue;
    }
    PetscCallHIPX(hipxHaloDestroy(halo)); /* the caller goes back to the reference's own host scatter */
    *transport_out = 0;
    PetscFunctionReturn(PETSC_SUCCESS);
  }
  *transport_out = transport;
  PetscFunctionReturn(PETSC_SUCCESS);
}
