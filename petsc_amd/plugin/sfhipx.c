/*
 * sfhipx.c -- PetscSF type "hipx" (SURVEY.md 8(f3)): PETSCSFBASIC whose Bcast / Reduce run ON THE DEVICE when the root and leaf
 * buffers are device memory (PETSC_MEMTYPE_HIP), over libhipx's ghost-exchange transports (RCCL send/recv over xGMI, or IPC peer
 * stores); every other case is the parent's.
 *
 * Replaces, for unit = MPIU_SCALAR and op in {MPI_REPLACE, MPIU_SUM}:
 *   PetscSFBcastBegin_Basic / PetscSFBcastEnd_Basic     src/vec/is/sf/impls/basic/sfbasic.c:352-382
 *   PetscSFReduceBegin_Basic / PetscSFReduceEnd_Basic   sfbasic.c:384-430
 * (pack: sfpack.c:706-721, unpack: sfpack.c:759-790).  Same structure: pack roots -> start communication -> local (self) scatter
 * at Begin; finish communication -> unpack at End, remote contributions applied in the order of the incoming-rank list, so
 * MPIU_SUM results are bit-identical to the host path.
 *
 * Subclassing (works against a default hidden-visibility libpetsc): PetscSFSetType(sf, "hipx") has already renamed the object
 * when it calls the creator (sf.c:168-188); the creator runs the registered creator of "basic" (PetscSFCreate_Basic is
 * PETSC_INTERN, so it is looked up in PetscSFList), captures the ops table and overrides four slots + Reset/Destroy.
 * Ops table: include/petsc/private/sfimpl.h:25-49.
 *
 * Who hands it device pointers: (a) MATMPIAIJHIPX with -mat_mpiaijhipx_halo sf (MatMult's Mvctx through the SF interface),
 * (b) VecScatterBegin/End on hipx vectors when -vec_hipx_memtype is given (VecGetArray*AndMemType then return the device mirror:
 * vscat.c:41-108 passes the memtypes straight to PetscSFBcast/ReduceWithMemTypeBegin).  Device buffers of any other unit / op are
 * staged through pinned host copies and the parent's path (correct, slower).
 */
#include "hipxplugin.h"
#include <petscsf.h>
#include <petsc/private/sfimpl.h>
#include <../src/vec/is/sf/impls/basic/sfbasic.h>

typedef struct {
  struct _PetscSFOps basic; /* the parent's ops */
  /* device plan, built lazily at the first device call after (re)SetUp */
  PetscBool built, usable;
  PetscInt  transport;
  hipxHalo  fwd, rev; /* roots -> leaves, leaves -> roots (remote edges only) */
  /* self edges: leaf[lleaf[k]] <-> root[lroot[k]] */
  PetscInt  nlocal;
  hipx_int *d_lroot, *d_lleaf;
  PetscBool lroot_dups, lleaf_dups;
  /* remote edges, leaf side: positions in leafdata in message order; root side: positions in rootdata in message order */
  PetscInt  nrleaf, nrroot;
  hipx_int *d_rleaf, *d_rroot;
  PetscBool rleaf_contig, rroot_seq_ok, rleaf_seq_ok; /* seq_ok: no duplicate inside one rank's message */
  PetscBool rleaf_dups, rroot_dups;                   /* duplicates across the whole remote list */
  /* the device-or-staged choice of a collective must be the SAME on every rank (a rank on the parent's Isend/Irecv path cannot meet
     neighbours inside hipxHaloBegin): the duplicate conditions above are rank-local, these are their logical ORs over the communicator */
  PetscBool g_bcast_sum_host, g_reduce_sum_host, g_reduce_replace_host;
  PetscInt  rleaf_start;
  PetscInt  nin, nout;      /* remote incoming / outgoing ranks */
  PetscInt *inoff, *outoff; /* message offsets (host) */
  double   *d_recv_leaf, *d_recv_root;
  /* the operation in flight (one at a time on the device path) */
  int          inflight; /* 0 none, 1 device bcast, 2 device reduce, 3 staged bcast, 4 staged reduce */
  MPI_Op       op;
  const void  *src;
  void        *dst;
  void        *h_root, *h_leaf; /* staging */
  size_t       rootbytes, leafbytes;
  PetscMemType rootmtype, leafmtype;
} SF_HIPX;

static const char SFHIPX_KEY[] = "hipx_sf_ext";

static PetscErrorCode SFHIPXGet(PetscSF sf, SF_HIPX **h)
{
  PetscContainer c;

  PetscFunctionBegin;
  PetscCall(PetscObjectQuery((PetscObject)sf, SFHIPX_KEY, (PetscObject *)&c));
  PetscCheck(c, PetscObjectComm((PetscObject)sf), PETSC_ERR_PLIB, "PetscSF of type hipx without its extension");
  PetscCall(PetscContainerGetPointer(c, (void **)h));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode SFHIPXFreePlan(SF_HIPX *h)
{
  PetscFunctionBegin;
  if (h->fwd) PetscCallHIPX(hipxHaloDestroy(&h->fwd));
  if (h->rev) PetscCallHIPX(hipxHaloDestroy(&h->rev));
  if (h->d_lroot) PetscCallHIPX(hipxFree(h->d_lroot));
  if (h->d_lleaf) PetscCallHIPX(hipxFree(h->d_lleaf));
  if (h->d_rleaf) PetscCallHIPX(hipxFree(h->d_rleaf));
  if (h->d_rroot) PetscCallHIPX(hipxFree(h->d_rroot));
  if (h->d_recv_leaf) PetscCallHIPX(hipxFree(h->d_recv_leaf));
  if (h->d_recv_root) PetscCallHIPX(hipxFree(h->d_recv_root));
  PetscCall(PetscFree(h->inoff));
  PetscCall(PetscFree(h->outoff));
  h->d_lroot = h->d_lleaf = h->d_rleaf = h->d_rroot = NULL;
  h->d_recv_leaf = h->d_recv_root = NULL;
  h->built = h->usable = PETSC_FALSE;
  h->transport = 0;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode UploadIdx(const PetscInt *src, PetscInt n, hipx_int **d)
{
  hipx_int *tmp;

  PetscFunctionBegin;
  *d = NULL;
  if (!n) PetscFunctionReturn(PETSC_SUCCESS);
  PetscCall(PetscMalloc1(n, &tmp));
  for (PetscInt k = 0; k < n; k++) tmp[k] = (hipx_int)src[k];
  PetscCallHIPX(hipxMalloc((void **)d, sizeof(hipx_int) * (size_t)n));
  PetscCallHIPX(hipxMemcpyHtoD(*d, tmp, sizeof(hipx_int) * (size_t)n));
  PetscCall(PetscFree(tmp));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* does idx[0..n) hold a value twice?  (marker array of the index range) */
static PetscErrorCode HasDups(const PetscInt *idx, PetscInt n, PetscInt range, PetscBool *dups)
{
  char *seen;

  PetscFunctionBegin;
  *dups = PETSC_FALSE;
  if (n < 2) PetscFunctionReturn(PETSC_SUCCESS);
  PetscCall(PetscCalloc1(range + 1, &seen));
  for (PetscInt k = 0; k < n && !*dups; k++) {
    if (seen[idx[k]]) *dups = PETSC_TRUE;
    seen[idx[k]] = 1;
  }
  PetscCall(PetscFree(seen));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* The device plan: the parent's own rank / index lists (PetscSFSetUp_Basic, sfbasic.c:20-90) split into the self part and the remote
   part, the remote part handed to libhipx as two ghost-exchange plans (forward and reverse).  Collective. */
static PetscErrorCode SFHIPXBuildPlan(PetscSF sf, SF_HIPX *h)
{
  PetscSF_Basic     *bas  = (PetscSF_Basic *)sf->data;
  MPI_Comm           comm = PetscObjectComm((PetscObject)sf);
  PetscMPIInt        size, nr = sf->nranks, ndr = sf->ndranks, ni = bas->niranks, ndi = bas->ndiranks;
  const PetscMPIInt *ranks = sf->ranks, *iranks = bas->iranks;
  const PetscInt    *roff = sf->roffset, *rmine = sf->rmine, *rremote = sf->rremote, *ioff = bas->ioffset, *iroot = bas->irootloc;
  char               want[16] = "auto";
  const char        *env      = getenv("HIPX_HALO");
  int                cmp, ok = 1, allok;
  PetscInt           transport = 0;

  PetscFunctionBegin;
  PetscCall(SFHIPXFreePlan(h));
  h->built = PETSC_TRUE;
  PetscCallMPI(MPI_Comm_size(comm, &size));
  if (env && strcmp(env, "host") && strcmp(env, "sf")) PetscCall(PetscStrncpy(want, env, sizeof(want)));
  PetscCall(PetscOptionsGetString(((PetscObject)sf)->options, ((PetscObject)sf)->prefix, "-sf_hipx_transport", want, sizeof(want), NULL));
  PetscCall(VecHIPXInitRuntime());
  /* the device transports are bootstrapped over PETSC_COMM_WORLD ranks */
  PetscCallMPI(MPI_Comm_compare(comm, PETSC_COMM_WORLD, &cmp));
  if (size > 1 && cmp != MPI_IDENT && cmp != MPI_CONGRUENT) ok = 0;
  if (sf->nroots < 0 || sf->maxleaf >= PETSC_INT_MAX / 2) ok = 0;
  /* self edges */
  h->nlocal = ndr ? roff[ndr] - roff[0] : 0;
  if (h->nlocal) {
    PetscCheck(ndi == ndr && ioff[ndi] - ioff[0] == h->nlocal, comm, PETSC_ERR_PLIB, "self edges of the root and leaf sides differ");
    /* the leaf side lists (leaf position, root index on the owner = me): rmine / rremote of the distinguished rank */
    PetscCall(UploadIdx(rmine + roff[0], h->nlocal, &h->d_lleaf));
    PetscCall(UploadIdx(rremote + roff[0], h->nlocal, &h->d_lroot));
    PetscCall(HasDups(rmine + roff[0], h->nlocal, sf->maxleaf, &h->lleaf_dups));
    PetscCall(HasDups(rremote + roff[0], h->nlocal, sf->nroots, &h->lroot_dups));
  }
  /* remote edges */
  h->nout   = nr - ndr;
  h->nin    = ni - ndi;
  h->nrleaf = roff[nr] - roff[ndr];
  h->nrroot = ioff[ni] - ioff[ndi];
  PetscCall(PetscMalloc1(h->nout + 1, &h->outoff));
  PetscCall(PetscMalloc1(h->nin + 1, &h->inoff));
  for (PetscInt k = 0; k <= h->nout; k++) h->outoff[k] = roff[ndr + k] - roff[ndr];
  for (PetscInt k = 0; k <= h->nin; k++) h->inoff[k] = ioff[ndi + k] - ioff[ndi];
  PetscCall(UploadIdx(rmine + roff[ndr], h->nrleaf, &h->d_rleaf));
  PetscCall(UploadIdx(iroot + ioff[ndi], h->nrroot, &h->d_rroot));
  h->rleaf_contig = PETSC_TRUE;
  h->rleaf_start  = h->nrleaf ? rmine[roff[ndr]] : 0;
  for (PetscInt k = 0; k < h->nrleaf && h->rleaf_contig; k++)
    if (rmine[roff[ndr] + k] != h->rleaf_start + k) h->rleaf_contig = PETSC_FALSE;
  PetscCall(HasDups(rmine + roff[ndr], h->nrleaf, sf->maxleaf, &h->rleaf_dups));
  PetscCall(HasDups(iroot + ioff[ndi], h->nrroot, sf->nroots, &h->rroot_dups));
  h->rleaf_seq_ok = h->rroot_seq_ok = PETSC_TRUE;
  for (PetscInt k = 0; k < h->nout && h->rleaf_dups && h->rleaf_seq_ok; k++) {
    PetscBool d;
    PetscCall(HasDups(rmine + roff[ndr + k], roff[ndr + k + 1] - roff[ndr + k], sf->maxleaf, &d));
    if (d) h->rleaf_seq_ok = PETSC_FALSE;
  }
  for (PetscInt k = 0; k < h->nin && h->rroot_dups && h->rroot_seq_ok; k++) {
    PetscBool d;
    PetscCall(HasDups(iroot + ioff[ndi + k], ioff[ndi + k + 1] - ioff[ndi + k], sf->nroots, &d));
    if (d) h->rroot_seq_ok = PETSC_FALSE;
  }
  if (h->nrleaf) PetscCallHIPX(hipxMalloc((void **)&h->d_recv_leaf, sizeof(double) * (size_t)h->nrleaf));
  if (h->nrroot) PetscCallHIPX(hipxMalloc((void **)&h->d_recv_root, sizeof(double) * (size_t)h->nrroot));
  {
    int loc[4], glob[4];
    loc[0] = ok ? 0 : 1;
    loc[1] = (h->lleaf_dups || (h->rleaf_dups && !h->rleaf_seq_ok)) ? 1 : 0; /* Bcast with a sum: a leaf summed into twice inside one message */
    loc[2] = (h->lroot_dups || (h->rroot_dups && !h->rroot_seq_ok)) ? 1 : 0; /* Reduce with a sum: a root summed into twice inside one message */
    loc[3] = (h->lroot_dups || h->rroot_dups) ? 1 : 0;                       /* Reduce with REPLACE: the LAST leaf must win */
    PetscCallMPI(MPI_Allreduce(loc, glob, 4, MPI_INT, MPI_MAX, comm));
    allok                    = !glob[0];
    h->g_bcast_sum_host      = glob[1] ? PETSC_TRUE : PETSC_FALSE;
    h->g_reduce_sum_host     = glob[2] ? PETSC_TRUE : PETSC_FALSE;
    h->g_reduce_replace_host = glob[3] ? PETSC_TRUE : PETSC_FALSE;
  }
  if (!allok) PetscFunctionReturn(PETSC_SUCCESS);
  if (size > 1) {
    int      *sr, *rr;
    hipx_int *so, *si, *ro;
    /* forward: I send my roots irootloc[...] to the incoming ranks, I receive my remote leaves from the root owners */
    PetscCall(PetscMalloc5(h->nin + 1, &sr, h->nin + 2, &so, h->nrroot + 1, &si, h->nout + 1, &rr, h->nout + 2, &ro));
    for (PetscInt k = 0; k < h->nin; k++) sr[k] = (int)iranks[ndi + k];
    for (PetscInt k = 0; k <= h->nin; k++) so[k] = (hipx_int)h->inoff[k];
    for (PetscInt k = 0; k < h->nrroot; k++) si[k] = (hipx_int)iroot[ioff[ndi] + k];
    for (PetscInt k = 0; k < h->nout; k++) rr[k] = (int)ranks[ndr + k];
    for (PetscInt k = 0; k <= h->nout; k++) ro[k] = (hipx_int)h->outoff[k];
    PetscCallHIPX(hipxHaloCreate((int)h->nin, sr, so, si, (int)h->nout, rr, ro, &h->fwd));
    PetscCall(PetscFree5(sr, so, si, rr, ro));
    /* reverse: I send my remote leaves (positions rmine) to the root owners, I receive contributions to my roots */
    PetscCall(PetscMalloc5(h->nout + 1, &sr, h->nout + 2, &so, h->nrleaf + 1, &si, h->nin + 1, &rr, h->nin + 2, &ro));
    for (PetscInt k = 0; k < h->nout; k++) sr[k] = (int)ranks[ndr + k];
    for (PetscInt k = 0; k <= h->nout; k++) so[k] = (hipx_int)h->outoff[k];
    for (PetscInt k = 0; k < h->nrleaf; k++) si[k] = (hipx_int)rmine[roff[ndr] + k];
    for (PetscInt k = 0; k < h->nin; k++) rr[k] = (int)iranks[ndi + k];
    for (PetscInt k = 0; k <= h->nin; k++) ro[k] = (hipx_int)h->inoff[k];
    PetscCallHIPX(hipxHaloCreate((int)h->nout, sr, so, si, (int)h->nin, rr, ro, &h->rev));
    PetscCall(PetscFree5(sr, so, si, rr, ro));
    PetscCall(HipxHaloBringUp(comm, (PetscObject)sf, &h->fwd, want, &transport));
    if (transport) {
      PetscInt t2 = 0;
      PetscCall(HipxHaloBringUp(comm, (PetscObject)sf, &h->rev, transport == 2 ? "rccl" : "ipc", &t2));
      if (t2 != transport) { /* the second plan did not come up on the same transport: no device path */
        if (h->fwd) PetscCallHIPX(hipxHaloDestroy(&h->fwd));
        if (h->rev) PetscCallHIPX(hipxHaloDestroy(&h->rev));
        transport = 0;
      }
    } else if (h->rev) PetscCallHIPX(hipxHaloDestroy(&h->rev));
    if (!transport) PetscFunctionReturn(PETSC_SUCCESS);
  }
  h->transport = transport;
  h->usable    = PETSC_TRUE;
  PetscCall(PetscInfo(sf, "PetscSF hipx: device plan built: %" PetscInt_FMT " self edges, %" PetscInt_FMT " remote leaves from %" PetscInt_FMT " ranks, %" PetscInt_FMT " remote roots to %" PetscInt_FMT " ranks, transport %s\n",
                      h->nlocal, h->nrleaf, h->nout, h->nrroot, h->nin, transport == 2 ? "rccl" : transport == 1 ? "ipc" : "none (one rank)"));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static inline PetscBool DeviceOp(MPI_Op op) { return (PetscBool)(op == MPI_REPLACE || op == MPIU_SUM || op == MPI_SUM); }

/* can (unit, memtypes, op) run on the device plan?  Decided identically on every rank: the plan is collective and unit / op / memtypes
   are arguments every rank passes alike for a VecScatter; a rank-dependent answer is excluded by the all-reduce at plan build */
static PetscErrorCode SFHIPXDeviceCase(PetscSF sf, SF_HIPX *h, MPI_Datatype unit, PetscMemType rm, PetscMemType lm, MPI_Op op, PetscBool *yes)
{
  PetscFunctionBegin;
  *yes = PETSC_FALSE;
  if (!PetscMemTypeDevice(rm) || !PetscMemTypeDevice(lm)) PetscFunctionReturn(PETSC_SUCCESS);
  if (unit != MPIU_SCALAR && unit != MPI_DOUBLE) PetscFunctionReturn(PETSC_SUCCESS);
  if (!DeviceOp(op)) PetscFunctionReturn(PETSC_SUCCESS);
  if (!h->built) PetscCall(SFHIPXBuildPlan(sf, h));
  *yes = h->usable;
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* ---- staged path: device buffers the device plan does not cover -> pinned host copies -> the parent */
static PetscErrorCode StageIn(SF_HIPX *h, MPI_Datatype unit, PetscInt nroots, PetscInt nleafspan, PetscMemType rm, const void *root, PetscMemType lm, const void *leaf, const void **hroot, const void **hleaf)
{
  PetscMPIInt usz;

  PetscFunctionBegin;
  PetscCallMPI(MPI_Type_size(unit, &usz));
  h->rootbytes = (size_t)usz * (size_t)nroots;
  h->leafbytes = (size_t)usz * (size_t)nleafspan;
  h->h_root = h->h_leaf = NULL;
  *hroot = root;
  *hleaf = leaf;
  if (PetscMemTypeDevice(rm)) {
    PetscCallHIPX(hipxMallocHost(&h->h_root, h->rootbytes));
    if (h->rootbytes) PetscCallHIPX(hipxMemcpyDtoH(h->h_root, root, h->rootbytes));
    *hroot = h->h_root;
  }
  if (PetscMemTypeDevice(lm)) {
    PetscCallHIPX(hipxMallocHost(&h->h_leaf, h->leafbytes));
    if (h->leafbytes) PetscCallHIPX(hipxMemcpyDtoH(h->h_leaf, leaf, h->leafbytes));
    *hleaf = h->h_leaf;
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PetscSFBcastBegin_HIPX(PetscSF sf, MPI_Datatype unit, PetscMemType rootmtype, const void *rootdata, PetscMemType leafmtype, void *leafdata, MPI_Op op)
{
  SF_HIPX  *h;
  PetscBool dev;

  PetscFunctionBegin;
  PetscCall(SFHIPXGet(sf, &h));
  if (PetscMemTypeHost(rootmtype) && PetscMemTypeHost(leafmtype)) {
    PetscCall((*h->basic.BcastBegin)(sf, unit, rootmtype, rootdata, leafmtype, leafdata, op));
    PetscFunctionReturn(PETSC_SUCCESS);
  }
  PetscCheck(!h->inflight, PetscObjectComm((PetscObject)sf), PETSC_ERR_ORDER, "PetscSF hipx: one device-buffer operation at a time");
  PetscCall(SFHIPXDeviceCase(sf, h, unit, rootmtype, leafmtype, op, &dev));
  h->op  = op;
  h->src = rootdata;
  h->dst = leafdata;
  if (dev && op != MPI_REPLACE && h->g_bcast_sum_host) dev = PETSC_FALSE; /* some rank sums into a leaf twice inside one message: every rank keeps the sequential host loop */
  if (dev) {
    const double *root = (const double *)rootdata;
    double       *leaf = (double *)leafdata;
    /* sfbasic.c:360-364: pack + start the remote exchange, then the self scatter overlaps it */
    if (h->fwd) PetscCallHIPX(hipxHaloBegin(h->fwd, root, (h->rleaf_contig && op == MPI_REPLACE) ? leaf + h->rleaf_start : h->d_recv_leaf));
    if (h->nlocal) PetscCallHIPX(hipxVecScatterIndexed(root, h->d_lroot, leaf, h->d_lleaf, (hipx_int)h->nlocal, op == MPI_REPLACE ? 0 : 1));
    h->inflight = 1;
  } else {
    const void *hr, *hl;
    PetscCall(StageIn(h, unit, sf->nroots, sf->maxleaf + 1, rootmtype, rootdata, leafmtype, leafdata, &hr, &hl));
    h->rootmtype = rootmtype;
    h->leafmtype = leafmtype;
    PetscCall((*h->basic.BcastBegin)(sf, unit, PETSC_MEMTYPE_HOST, hr, PETSC_MEMTYPE_HOST, (void *)hl, op));
    h->inflight = 3;
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* remote unpack into dst[idx[..]]: one kernel when no position repeats, else message by message in rank order (sfpack.c unpacks the
   receive buffer front to back: the same order of additions) */
static PetscErrorCode UnpackRemote(const double *buf, double *dst, const hipx_int *d_idx, PetscInt n, PetscInt nmsg, const PetscInt *off, PetscBool dups, MPI_Op op)
{
  PetscFunctionBegin;
  if (!n) PetscFunctionReturn(PETSC_SUCCESS);
  if (op == MPI_REPLACE || !dups) PetscCallHIPX(hipxVecScatterIndexed(buf, NULL, dst, d_idx, (hipx_int)n, op == MPI_REPLACE ? 0 : 1)); /* REPLACE with repeats: every copy carries the same root value */
  else
    for (PetscInt k = 0; k < nmsg; k++) PetscCallHIPX(hipxVecScatterIndexed(buf + off[k], NULL, dst, d_idx + off[k], (hipx_int)(off[k + 1] - off[k]), 1));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PetscSFBcastEnd_HIPX(PetscSF sf, MPI_Datatype unit, const void *rootdata, void *leafdata, MPI_Op op)
{
  SF_HIPX *h;

  PetscFunctionBegin;
  PetscCall(SFHIPXGet(sf, &h));
  if (!h->inflight || h->src != rootdata || h->dst != leafdata) { /* a host-buffer operation: the parent keeps its own links */
    PetscCall((*h->basic.BcastEnd)(sf, unit, rootdata, leafdata, op));
    PetscFunctionReturn(PETSC_SUCCESS);
  }
  if (h->inflight == 1) {
    double *leaf = (double *)leafdata;
    if (h->fwd) {
      const double *ghost, *target = (h->rleaf_contig && op == MPI_REPLACE) ? leaf + h->rleaf_start : h->d_recv_leaf;
      PetscCallHIPX(hipxHaloEnd(h->fwd));
      PetscCallHIPX(hipxHaloGhost(h->fwd, target, &ghost));
      if (!(ghost == leaf + h->rleaf_start && h->rleaf_contig && op == MPI_REPLACE)) PetscCall(UnpackRemote(ghost, leaf, h->d_rleaf, h->nrleaf, h->nout, h->outoff, h->rleaf_dups, op));
      PetscCallHIPX(hipxHaloRelease(h->fwd));
    }
  } else {
    PetscCheck(h->inflight == 3, PetscObjectComm((PetscObject)sf), PETSC_ERR_ORDER, "PetscSFBcastEnd does not match the operation in flight");
    PetscCall((*h->basic.BcastEnd)(sf, unit, h->h_root ? h->h_root : rootdata, h->h_leaf ? h->h_leaf : leafdata, op));
    if (h->h_leaf && h->leafbytes) PetscCallHIPX(hipxMemcpyHtoD(leafdata, h->h_leaf, h->leafbytes));
    if (h->h_root) PetscCallHIPX(hipxFreeHost(h->h_root));
    if (h->h_leaf) PetscCallHIPX(hipxFreeHost(h->h_leaf));
    h->h_root = h->h_leaf = NULL;
  }
  h->inflight = 0;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PetscSFReduceBegin_HIPX(PetscSF sf, MPI_Datatype unit, PetscMemType leafmtype, const void *leafdata, PetscMemType rootmtype, void *rootdata, MPI_Op op)
{
  SF_HIPX  *h;
  PetscBool dev;

  PetscFunctionBegin;
  PetscCall(SFHIPXGet(sf, &h));
  if (PetscMemTypeHost(rootmtype) && PetscMemTypeHost(leafmtype)) {
    PetscCall((*h->basic.ReduceBegin)(sf, unit, leafmtype, leafdata, rootmtype, rootdata, op));
    PetscFunctionReturn(PETSC_SUCCESS);
  }
  PetscCheck(!h->inflight, PetscObjectComm((PetscObject)sf), PETSC_ERR_ORDER, "PetscSF hipx: one device-buffer operation at a time");
  PetscCall(SFHIPXDeviceCase(sf, h, unit, rootmtype, leafmtype, op, &dev));
  h->op  = op;
  h->src = leafdata;
  h->dst = rootdata;
  /* several leaves of one message (or several self leaves) on the same root: with MPIU_SUM the parallel kernel would race, with
     MPI_REPLACE the LAST one must win (sequential semantics): both stay on the host loop */
  if (dev && (op == MPI_REPLACE ? h->g_reduce_replace_host : h->g_reduce_sum_host)) dev = PETSC_FALSE; /* decided collectively (SFHIPXBuildPlan) */
  if (dev) {
    const double *leaf = (const double *)leafdata;
    double       *root = (double *)rootdata;
    /* sfbasic.c:390-396: pack the leaves + start the exchange, then the self part (PetscSFLinkScatterLocal) */
    if (h->rev) PetscCallHIPX(hipxHaloBegin(h->rev, leaf, h->d_recv_root));
    if (h->nlocal) PetscCallHIPX(hipxVecScatterIndexed(leaf, h->d_lleaf, root, h->d_lroot, (hipx_int)h->nlocal, op == MPI_REPLACE ? 0 : 1));
    h->inflight = 2;
  } else {
    const void *hr, *hl;
    PetscCall(StageIn(h, unit, sf->nroots, sf->maxleaf + 1, rootmtype, rootdata, leafmtype, leafdata, &hr, &hl));
    h->rootmtype = rootmtype;
    h->leafmtype = leafmtype;
    PetscCall((*h->basic.ReduceBegin)(sf, unit, PETSC_MEMTYPE_HOST, hl, PETSC_MEMTYPE_HOST, (void *)hr, op));
    h->inflight = 4;
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PetscSFReduceEnd_HIPX(PetscSF sf, MPI_Datatype unit, const void *leafdata, void *rootdata, MPI_Op op)
{
  SF_HIPX *h;

  PetscFunctionBegin;
  PetscCall(SFHIPXGet(sf, &h));
  if (!h->inflight || h->src != leafdata || h->dst != rootdata) {
    PetscCall((*h->basic.ReduceEnd)(sf, unit, leafdata, rootdata, op));
    PetscFunctionReturn(PETSC_SUCCESS);
  }
  if (h->inflight == 2) {
    if (h->rev) {
      const double *ghost;
      PetscCallHIPX(hipxHaloEnd(h->rev));
      PetscCallHIPX(hipxHaloGhost(h->rev, h->d_recv_root, &ghost));
      PetscCall(UnpackRemote(ghost, (double *)rootdata, h->d_rroot, h->nrroot, h->nin, h->inoff, h->rroot_dups, op)); /* sfbasic.c:423: PetscSFLinkUnpackRootData, buffer order */
      PetscCallHIPX(hipxHaloRelease(h->rev));
    }
  } else {
    PetscCheck(h->inflight == 4, PetscObjectComm((PetscObject)sf), PETSC_ERR_ORDER, "PetscSFReduceEnd does not match the operation in flight");
    PetscCall((*h->basic.ReduceEnd)(sf, unit, h->h_leaf ? h->h_leaf : leafdata, h->h_root ? h->h_root : rootdata, op));
    if (h->h_root && h->rootbytes) PetscCallHIPX(hipxMemcpyHtoD(rootdata, h->h_root, h->rootbytes));
    if (h->h_root) PetscCallHIPX(hipxFreeHost(h->h_root));
    if (h->h_leaf) PetscCallHIPX(hipxFreeHost(h->h_leaf));
    h->h_root = h->h_leaf = NULL;
  }
  h->inflight = 0;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PetscSFSetUp_HIPX(PetscSF sf)
{
  SF_HIPX *h;

  PetscFunctionBegin;
  PetscCall(SFHIPXGet(sf, &h));
  PetscCall(SFHIPXFreePlan(h)); /* rebuilt lazily from the new rank lists */
  PetscCall((*h->basic.SetUp)(sf));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PetscSFReset_HIPX(PetscSF sf)
{
  SF_HIPX *h;

  PetscFunctionBegin;
  PetscCall(SFHIPXGet(sf, &h));
  PetscCall(SFHIPXFreePlan(h));
  if (h->basic.Reset) PetscCall((*h->basic.Reset)(sf));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PetscSFDestroy_HIPX(PetscSF sf)
{
  SF_HIPX *h;
  PetscErrorCode (*pdestroy)(PetscSF);

  PetscFunctionBegin;
  PetscCall(SFHIPXGet(sf, &h));
  PetscCall(SFHIPXFreePlan(h));
  pdestroy = h->basic.Destroy;
  PetscCall(PetscObjectCompose((PetscObject)sf, SFHIPX_KEY, NULL)); /* frees the extension (container destroy) */
  if (pdestroy) PetscCall((*pdestroy)(sf));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode SFHIPXExtDestroy(PetscCtxRt ctx)
{
  PetscFunctionBegin;
  PetscCall(PetscFree(*(void **)ctx));
  PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode PetscSFCreate_HIPX(PetscSF sf)
{
  PetscErrorCode (*rb)(PetscSF) = NULL;
  SF_HIPX      *h;
  PetscContainer c;

  PetscFunctionBegin;
  PetscCall(PetscFunctionListFind(PetscSFList, PETSCSFBASIC, &rb));
  PetscCheck(rb && rb != PetscSFCreate_HIPX, PetscObjectComm((PetscObject)sf), PETSC_ERR_PLIB, "PETSCSFBASIC is not registered");
  PetscCall((*rb)(sf)); /* PetscSFCreate_Basic: fills sf->ops, allocates sf->data */
  PetscCall(PetscNew(&h));
  h->basic = *sf->ops;
  PetscCall(PetscContainerCreate(PETSC_COMM_SELF, &c));
  PetscCall(PetscContainerSetPointer(c, h));
  PetscCall(PetscContainerSetCtxDestroy(c, SFHIPXExtDestroy));
  PetscCall(PetscObjectCompose((PetscObject)sf, SFHIPX_KEY, (PetscObject)c));
  PetscCall(PetscContainerDestroy(&c));
  sf->ops->BcastBegin  = PetscSFBcastBegin_HIPX;
  sf->ops->BcastEnd    = PetscSFBcastEnd_HIPX;
  sf->ops->ReduceBegin = PetscSFReduceBegin_HIPX;
  sf->ops->ReduceEnd   = PetscSFReduceEnd_HIPX;
  sf->ops->SetUp       = PetscSFSetUp_HIPX;
  sf->ops->Reset       = PetscSFReset_HIPX;
  sf->ops->Destroy     = PetscSFDestroy_HIPX;
  PetscFunctionReturn(PETSC_SUCCESS);
}
