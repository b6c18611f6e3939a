/*
 * vechipx.c -- VECSEQHIPX / VECMPIHIPX / VECHIPX: PETSc vectors whose Vec BLAS-1 ops run as HIP kernels (libhipx).
 *
 * Subclassing recipe (works against a default hidden-visibility libpetsc): the creator calls
 * VecSetType(v, VECSEQ | VECMPI) (-> hidden VecCreate_Seq / exported VecCreate_MPI, src/vec/vec/impls/seq/bvec3.c:22,
 * impls/mpi/pbvec.c), keeps the parent's ops table for everything host-side (view, setvalues, load, ...), and overrides
 * the slots of struct _VecOps (include/petsc/private/vecimpl.h:18-110) that are on the Krylov path.
 *
 * Coherence: the host array stays where the parent put it (VECHEADER); the device mirror is allocated on first use.
 * v->offloadmask in {CPU, GPU, BOTH} (include/petscdevicetypes.h:239-246) says where the valid copy is:
 *   getarray      -> device-to-host if GPU is newer, then CPU      (host may write)
 *   getarrayread  -> device-to-host if GPU is newer, then BOTH
 *   getarraywrite -> no copy, CPU
 *   a kernel that reads  -> host-to-device if CPU is newer, then BOTH
 *   a kernel that writes -> GPU
 * Reductions return with the scalar valid (blocking); every other op only enqueues work on libhipx's compute stream.
 */
#include "hipxplugin.h"

static PetscBool hipx_runtime_up = PETSC_FALSE;
static PetscBool hipx_rc_on      = PETSC_TRUE; /* -hipx_reduction_cache (hipxplugin.h) */
static PetscBool hipx_lazy_on    = PETSC_TRUE; /* -hipx_lazy_fusion (hipxplugin.h) */
static PetscInt  hipx_lazy_min   = 32768;      /* -hipx_lazy_min_size: shorter vectors run their operations at once */
static PetscBool hipx_lazy_view  = PETSC_FALSE; /* -hipx_lazy_view: counts at PetscFinalize */

PetscErrorCode VecHIPXInitRuntime(void)
{
  PetscInt    dev = -1;
  PetscMPIInt rank;
  int         ndev = 0;

  PetscFunctionBegin;
  if (hipx_runtime_up) PetscFunctionReturn(PETSC_SUCCESS);
  PetscCall(PetscOptionsGetInt(NULL, NULL, "-hipx_device", &dev, NULL));
  if (dev < 0) { /* one rank per GPU: local rank modulo the visible devices */
    PetscCallMPI(MPI_Comm_rank(PETSC_COMM_WORLD, &rank));
    PetscCallHIPX(hipxGetDeviceCount(&ndev));
    dev = ndev > 0 ? rank % ndev : 0;
  }
  PetscCallHIPX(hipxInit((int)dev));
  PetscCall(PetscOptionsGetBool(NULL, NULL, "-vec_hipx_memtype", &hipx_vec_memtype_ops, NULL));
  PetscCall(PetscOptionsGetBool(NULL, NULL, "-hipx_reduction_cache", &hipx_rc_on, NULL));
  PetscCall(PetscOptionsGetBool(NULL, NULL, "-hipx_lazy_fusion", &hipx_lazy_on, NULL));
  PetscCall(PetscOptionsGetInt(NULL, NULL, "-hipx_lazy_min_size", &hipx_lazy_min, NULL));
  PetscCall(PetscOptionsGetBool(NULL, NULL, "-hipx_lazy_view", &hipx_lazy_view, NULL));
  { /* -hipx_reductions exact|fast: compensated (Dot2) sums in every reduction kernel -- the values the reference's VecDot / VecNorm /
       VecMDot return when its BLAS is exactly rounded (bvec1.c:27, bvec2.c:202-223, dvec2.c:557); default: HIPX_REDUCTIONS or fast */
    char      mode[16] = "";
    PetscBool set      = PETSC_FALSE;
    PetscCall(PetscOptionsGetString(NULL, NULL, "-hipx_reductions", mode, sizeof(mode), &set));
    if (set) {
      PetscBool ex = PETSC_FALSE, fa = PETSC_FALSE;
      PetscCall(PetscStrcmp(mode, "exact", &ex));
      PetscCall(PetscStrcmp(mode, "fast", &fa));
      PetscCheck(ex || fa, PETSC_COMM_SELF, PETSC_ERR_ARG_WRONG, "-hipx_reductions must be exact or fast, not %s", mode);
      PetscCallHIPX(hipxSetReductionMode(ex ? HIPX_RED_EXACT : HIPX_RED_FAST));
    }
  }
  hipx_runtime_up = PETSC_TRUE;
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* ------------------------------------------------------------------ reduction cache (hipxplugin.h) */
typedef struct {
  PetscObjectId a, b;    /* the two vectors of the entry */
  PetscBool     live;    /* both vectors untouched since the producing kernel was enqueued */
  PetscBool     fetched; /* v[] holds the sums (the reduction slot has been waited for) */
  PetscBool     used;    /* somebody asked for it: the producer keeps fusing */
  double        v[2];
} HipxRedCacheEntry;
static HipxRedCacheEntry hipx_rc[2];
static PetscInt          hipx_rc_miss[2] = {0, 0}, hipx_rc_calls[2] = {0, 0};
static const int         hipx_rc_slot[2] = {60, 61}; /* libhipx reduction slots of the two producers (the host layer uses 1-3, blocking reductions 0) */

int VecHIPXRedCacheSlot(int kind) { return hipx_rc_slot[kind]; }

/* Fuse the sums into the next producer of this kind?  Yes while they get used; after two unused entries in a row (GMRES never asks for x . A x) the
   producer runs its plain kernel again and tries once more every 64 calls. */
PetscBool VecHIPXRedCacheWanted(int kind)
{
  if (!hipx_rc_on) return PETSC_FALSE;
  hipx_rc_calls[kind]++;
  if (hipx_rc_miss[kind] < 2) return PETSC_TRUE;
  if (hipx_rc_calls[kind] % 64 == 0) {
    hipx_rc_miss[kind] = 1;
    return PETSC_TRUE;
  }
  return PETSC_FALSE;
}

PetscErrorCode VecHIPXRedCachePut(int kind, Vec a, Vec b)
{
  HipxRedCacheEntry *e = &hipx_rc[kind];

  PetscFunctionBegin;
  if (!hipx_rc_on) PetscFunctionReturn(PETSC_SUCCESS); /* -hipx_reduction_cache 0: no entry ever becomes live, also not from the lazy producers' fused kernels */
  if (e->a && !e->used) hipx_rc_miss[kind]++;
  else hipx_rc_miss[kind] = 0;
  e->a       = ((PetscObject)a)->id;
  e->b       = ((PetscObject)b)->id;
  e->live    = PETSC_TRUE;
  e->fetched = PETSC_FALSE;
  e->used    = PETSC_FALSE;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static void VecHIPXBatchCacheInvalidate(PetscObjectId id);
void VecHIPXRedCacheInvalidate(Vec v)
{
  const PetscObjectId id = ((PetscObject)v)->id;
  for (int k = 0; k < 2; k++)
    if (hipx_rc[k].live && (hipx_rc[k].a == id || hipx_rc[k].b == id)) hipx_rc[k].live = PETSC_FALSE;
  VecHIPXBatchCacheInvalidate(id);
}

/* the sums of a live entry on (a, b) in either order, or NULL */
static PetscErrorCode VecHIPXRedCacheGet(int kind, Vec a, Vec b, PetscBool ordered, const double **v)
{
  HipxRedCacheEntry  *e  = &hipx_rc[kind];
  const PetscObjectId ia = ((PetscObject)a)->id, ib = ((PetscObject)b)->id;

  PetscFunctionBegin;
  *v = NULL;
  if (!hipx_rc_on || !e->live || !((e->a == ia && e->b == ib) || (!ordered && e->a == ib && e->b == ia))) PetscFunctionReturn(PETSC_SUCCESS);
  if (!e->fetched) {
    PetscCallHIPX(hipxRedEnd(hipx_rc_slot[kind], kind == HIPX_RC_PWMULT ? 2 : 1, e->v));
    e->fetched = PETSC_TRUE;
  }
  e->used = PETSC_TRUE;
  *v      = e->v;
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* Round 6: the sums a recorded BATCH left behind (VecHIPXLazyTryBatch below: the update blocks of KSPSolve_PIPECG / GROPPCG / PIPECR as one kernel, with the sums
   their next VecNormBegin / VecDotBegin calls ask for, comb.c:338,379).  Up to HIPX_BATCH_MAX_DOTS pairs of vectors; a pair dies when either vector is handed
   out for writing (VecHIPXRedCacheInvalidate), exactly like the two entries above. */
static struct {
  int           n;
  PetscObjectId a[HIPX_BATCH_MAX_DOTS], b[HIPX_BATCH_MAX_DOTS];
  PetscBool     live[HIPX_BATCH_MAX_DOTS], fetched;
  double        v[HIPX_BATCH_MAX_DOTS];
} hipx_bc;
static const int hipx_bc_slot = 59;

static void VecHIPXBatchCacheInvalidate(PetscObjectId id)
{
  for (int k = 0; k < hipx_bc.n; k++)
    if (hipx_bc.live[k] && (hipx_bc.a[k] == id || hipx_bc.b[k] == id)) hipx_bc.live[k] = PETSC_FALSE;
}

/* x . y from the batch cache (either order; x == y: the square of the 2-norm), or *hit = FALSE */
static PetscErrorCode VecHIPXBatchCacheGet(Vec x, Vec y, double *val, PetscBool *hit)
{
  const PetscObjectId ia = ((PetscObject)x)->id, ib = ((PetscObject)y)->id;

  PetscFunctionBegin;
  *hit = PETSC_FALSE;
  if (!hipx_rc_on) PetscFunctionReturn(PETSC_SUCCESS);
  for (int k = 0; k < hipx_bc.n; k++) {
    if (!hipx_bc.live[k] || !((hipx_bc.a[k] == ia && hipx_bc.b[k] == ib) || (hipx_bc.a[k] == ib && hipx_bc.b[k] == ia))) continue;
    if (!hipx_bc.fetched) {
      PetscCallHIPX(hipxRedEnd(hipx_bc_slot, hipx_bc.n, hipx_bc.v));
      hipx_bc.fetched = PETSC_TRUE;
    }
    *val = hipx_bc.v[k];
    *hit = PETSC_TRUE;
    break;
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* ------------------------------------------------------------------ device mirror management */
static PetscErrorCode VecGetArray_HIPX(Vec, PetscScalar **);

PetscBool VecIsHIPX(Vec v)
{
  return (PetscBool)(v && v->ops->getarray == VecGetArray_HIPX && VecHIPXGetExt(v)->magic == VECHIPX_MAGIC);
}

static PetscErrorCode VecHIPXAllocate(Vec v)
{
  VecHIPXExt *e = VecHIPXGetExt(v);
  PetscInt    n = v->map->n;

  PetscFunctionBegin;
  if (e->d_array && e->d_n >= n) PetscFunctionReturn(PETSC_SUCCESS);
  if (e->d_array && e->d_owned) PetscCallHIPX(hipxFree(e->d_array));
  if (e->d_alt && e->d_alt_owned) PetscCallHIPX(hipxFree(e->d_alt));
  e->d_alt = NULL;
  PetscCallHIPX(hipxMalloc((void **)&e->d_array, sizeof(PetscScalar) * (size_t)(n ? n : 1)));
  e->d_n     = n;
  e->d_owned = PETSC_TRUE;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode VecHIPXCopyToDevice(Vec v)
{
  VecHIPXExt *e = VecHIPXGetExt(v);

  PetscFunctionBegin;
  PetscCall(VecHIPXAllocate(v));
  if (v->offloadmask == PETSC_OFFLOAD_CPU || v->offloadmask == PETSC_OFFLOAD_UNALLOCATED) {
    const PetscScalar *h = *(PetscScalar **)v->data; /* VECHEADER: array is the first member */
    if (v->map->n && h) PetscCallHIPX(hipxMemcpyHtoD(e->d_array, h, sizeof(PetscScalar) * (size_t)v->map->n));
    v->offloadmask = PETSC_OFFLOAD_BOTH;
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode VecHIPXCopyToHost(Vec v)
{
  VecHIPXExt *e = VecHIPXGetExt(v);

  PetscFunctionBegin;
  if (v->offloadmask == PETSC_OFFLOAD_GPU) {
    PetscScalar *h = *(PetscScalar **)v->data;
    if (v->map->n) PetscCallHIPX(hipxMemcpyDtoH(h, e->d_array, sizeof(PetscScalar) * (size_t)v->map->n));
    v->offloadmask = PETSC_OFFLOAD_BOTH;
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* Foreign vectors (any other VecType handed to one of our ops; PetscCheckSameTypeAndComm is compiled out in optimized
   builds, petscimpl.h:613-639): staged through a temporary device buffer so the op still runs on the GPU. */
typedef struct {
  PetscScalar *d;
  PetscScalar *h;
  int          mode; /* 0 read, 1 write, 2 read-write */
} HIPXTmp;

static PetscErrorCode VecHIPXForeignGet(Vec v, PetscScalar **d, void **tmp, int mode)
{
  HIPXTmp *t;
  size_t   bytes = sizeof(PetscScalar) * (size_t)(v->map->n ? v->map->n : 1);

  PetscFunctionBegin;
  PetscCall(PetscNew(&t));
  t->mode = mode;
  PetscCallHIPX(hipxMalloc((void **)&t->d, bytes));
  if (mode == 0) PetscCall(VecGetArrayRead(v, (const PetscScalar **)&t->h));
  else if (mode == 1) PetscCall(VecGetArrayWrite(v, &t->h));
  else PetscCall(VecGetArray(v, &t->h));
  if (mode != 1 && v->map->n) PetscCallHIPX(hipxMemcpyHtoD(t->d, t->h, sizeof(PetscScalar) * (size_t)v->map->n));
  *d   = t->d;
  *tmp = t;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode VecHIPXForeignRestore(Vec v, void **tmp)
{
  HIPXTmp *t = (HIPXTmp *)*tmp;

  PetscFunctionBegin;
  if (t->mode != 0 && v->map->n) PetscCallHIPX(hipxMemcpyDtoH(t->h, t->d, sizeof(PetscScalar) * (size_t)v->map->n));
  if (t->mode == 0) PetscCall(VecRestoreArrayRead(v, (const PetscScalar **)&t->h));
  else if (t->mode == 1) PetscCall(VecRestoreArrayWrite(v, &t->h));
  else PetscCall(VecRestoreArray(v, &t->h));
  PetscCallHIPX(hipxFree(t->d));
  PetscCall(PetscFree(t));
  *tmp = NULL;
  PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode VecHIPXGetDeviceRead(Vec v, const PetscScalar **d, void **tmp)
{
  PetscFunctionBegin;
  *tmp = NULL;
  PetscCall(VecHIPXLazySync(v));
  if (VecIsHIPX(v)) {
    PetscCall(VecHIPXCopyToDevice(v));
    *d = VecHIPXGetExt(v)->d_array;
  } else PetscCall(VecHIPXForeignGet(v, (PetscScalar **)d, tmp, 0));
  PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode VecHIPXRestoreDeviceRead(Vec v, const PetscScalar **d, void **tmp)
{
  PetscFunctionBegin;
  if (*tmp) PetscCall(VecHIPXForeignRestore(v, tmp));
  *d = NULL;
  PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode VecHIPXGetDeviceWrite(Vec v, PetscScalar **d, void **tmp)
{
  PetscFunctionBegin;
  *tmp = NULL;
  PetscCall(VecHIPXLazySync(v));
  VecHIPXRedCacheInvalidate(v);
  if (VecIsHIPX(v)) {
    PetscCall(VecHIPXAllocate(v));
    *d = VecHIPXGetExt(v)->d_array;
  } else PetscCall(VecHIPXForeignGet(v, d, tmp, 1));
  PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode VecHIPXGetDeviceReadWrite(Vec v, PetscScalar **d, void **tmp)
{
  PetscFunctionBegin;
  *tmp = NULL;
  PetscCall(VecHIPXLazySync(v));
  VecHIPXRedCacheInvalidate(v);
  if (VecIsHIPX(v)) {
    PetscCall(VecHIPXCopyToDevice(v));
    *d = VecHIPXGetExt(v)->d_array;
  } else PetscCall(VecHIPXForeignGet(v, d, tmp, 2));
  PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode VecHIPXRestoreDeviceWrite(Vec v, PetscScalar **d, void **tmp)
{
  PetscFunctionBegin;
  if (*tmp) PetscCall(VecHIPXForeignRestore(v, tmp));
  else v->offloadmask = PETSC_OFFLOAD_GPU;
  *d = NULL;
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* ------------------------------------------------------------------ lazy fusion (hipxplugin.h)
   The queue holds at most HIPX_LAZY_MAX recorded operations, each "y += s x" (kind 1, VecAXPY_Seq bvec1.c:70-83) or "y = x + s y" (kind 2, VecAYPX_Seq
   dvec2.c:753-780) on hipx vectors whose device copies are current.  Running them later, in the recorded order, is what running them at once would have
   been as long as nobody looks at (or changes) a vector that takes part -- every accessor calls VecHIPXLazySync first.  Recognised on the way:
     [x += a p][p = z + b p] next to each other      -> hipxCGAypxAxpy: p read once (cg_aypx_axpy_kernel)
     ... and MatMult(A, p, w) asks for the product   -> hipxMatMultCGDirectionDotBegin: the two updates are the product kernel's prologue; p is rewritten
                                                        OUT OF PLACE (other workgroups still read the old direction): the vector's two device buffers swap
     [r += s w] and VecPointwiseMult(z, r, d) follows -> hipxVecAXPYPointwiseMultDotsBegin: one pass, the sums z.z and z.r into the reduction cache
     the WHOLE queue is one of libhipx's batch programs (round 6: the update blocks of KSPSolve_PIPECG pipecg.c:140-150, KSPSolve_GROPPCG groppcg.c:98-100,
     135-136, KSPSolve_PIPECR pipecr.c:108-117) when somebody looks at one of its vectors -> hipxVecBatchAXPYDotsBegin: every operand read once, every changed
     vector written once, and the sums the loop asks for next (VecNormBegin / VecDotBegin, comb.c:338,379) into the batch cache
   Every fused kernel performs, element by element, the operations of the separate kernels in their order: the vectors come out bit-identical. */
#define HIPX_LAZY_MAX HIPX_BATCH_MAX_OPS /* (8: the update block of KSPSolve_PIPECG, pipecg.c:140-150) */
typedef struct {
  int         kind; /* 1: y += s x, 2: y = x + s y */
  Vec         y, x;
  PetscScalar s;
} HipxLazyOp;
static HipxLazyOp hipx_lazy[HIPX_LAZY_MAX];
static PetscInt   hipx_lazy_stat[7] = {0, 0, 0, 0, 0, 0, 0}; /* recorded, run alone, run as "x += a p; p = z + b p", fused into VecPointwiseMult, fused into MatMult (pairs), batches, operations in batches */
static PetscBool  hipx_lazy_fin     = PETSC_FALSE;

static PetscErrorCode VecHIPXLazyFinalize(void)
{
  PetscFunctionBegin;
  if (hipx_lazy_view)
    PetscCall(PetscPrintf(PETSC_COMM_SELF, "hipx lazy fusion: %" PetscInt_FMT " operations recorded; %" PetscInt_FMT " run alone, %" PetscInt_FMT " pairs as one direction kernel, %" PetscInt_FMT " inside VecPointwiseMult, %" PetscInt_FMT " pairs as the prologue of MatMult\n",
                          hipx_lazy_stat[0], hipx_lazy_stat[1], hipx_lazy_stat[2], hipx_lazy_stat[3], hipx_lazy_stat[4]));
  if (hipx_lazy_view && hipx_lazy_stat[5]) PetscCall(PetscPrintf(PETSC_COMM_SELF, "hipx lazy fusion: %" PetscInt_FMT " operations in %" PetscInt_FMT " batch kernels\n", hipx_lazy_stat[6], hipx_lazy_stat[5]));
  for (int k = 0; k < 7; k++) hipx_lazy_stat[k] = 0;
  hipx_lazy_fin = PETSC_FALSE;
  PetscFunctionReturn(PETSC_SUCCESS);
}
static int        hipx_nlazy    = 0;
static PetscBool  hipx_lazy_run = PETSC_FALSE; /* the queue is being run: nothing is recorded, nothing syncs */

static PetscErrorCode VecHIPXLazyRunOne(const HipxLazyOp *o)
{
  VecHIPXExt *ey = VecHIPXGetExt(o->y), *ex = VecHIPXGetExt(o->x);

  PetscFunctionBegin;
  if (o->kind == 1) PetscCallHIPX(hipxVecAXPY(ey->d_array, o->s, ex->d_array, o->y->map->n));
  else PetscCallHIPX(hipxVecAYPX(ey->d_array, o->s, ex->d_array, o->y->map->n));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* two recorded operations may change places in time unless one writes what the other reads or writes */
static PetscBool VecHIPXLazyConflict(const HipxLazyOp *a, const HipxLazyOp *b) { return (PetscBool)(a->y == b->y || a->y == b->x || b->y == a->x); }

/* run the operations marked in run[], in their order (a marked "x += a p" directly followed -- among the marked ones -- by "p = z + b p" as one kernel),
   and keep the others recorded, in their order */
static PetscErrorCode VecHIPXLazyRunMarked_Inner(const PetscBool run[])
{
  HipxLazyOp keep[HIPX_LAZY_MAX];
  int        nkeep = 0, prev = -1;

  PetscFunctionBegin;
  for (int k = 0; k <= hipx_nlazy; k++) {
    if (k < hipx_nlazy && !run[k]) {
      keep[nkeep++] = hipx_lazy[k];
      continue;
    }
    /* k == hipx_nlazy: only the operation held back in `prev` is left */
    if (prev >= 0) {
      const HipxLazyOp *o = &hipx_lazy[prev], *q = k < hipx_nlazy ? &hipx_lazy[k] : NULL;
      if (q && o->kind == 1 && q->kind == 2 && o->x == q->y && o->y != q->x && o->y != q->y && q->x != q->y) { /* x += a p; p = z + b p */
        PetscCallHIPX(hipxCGAypxAxpy(VecHIPXGetExt(q->y)->d_array, q->s, VecHIPXGetExt(q->x)->d_array, VecHIPXGetExt(o->y)->d_array, o->s, q->y->map->n));
        hipx_lazy_stat[2]++;
        prev = -1;
        continue;
      }
      PetscCall(VecHIPXLazyRunOne(o));
      hipx_lazy_stat[1]++;
    }
    prev = k < hipx_nlazy ? k : -1;
  }
  for (int k = 0; k < nkeep; k++) hipx_lazy[k] = keep[k];
  hipx_nlazy = nkeep;
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* (a kernel launch that fails in there must not leave the queue "being run" for ever -- every later sync would be a no-op on stale data -- nor keep raw
   Vec pointers recorded: the state is restored and the queue dropped before the error travels on) */
static PetscErrorCode VecHIPXLazyRunMarked(const PetscBool run[])
{
  PetscErrorCode ierr;

  PetscFunctionBegin;
  hipx_lazy_run = PETSC_TRUE;
  ierr          = VecHIPXLazyRunMarked_Inner(run);
  hipx_lazy_run = PETSC_FALSE;
  if (ierr) hipx_nlazy = 0;
  PetscCall(ierr);
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* The queue -- or its tail behind a few leading operations, which then run first, one by one, as they would have anyway (an "x += a p" nobody has looked at
   since the iteration before: KSPSolve_PIPECR) -- as ONE kernel if it is one of libhipx's batch programs (see the head of this section).  Slots: for each
   operation y, then x, numbered by first appearance: the numbering the programs are written in.  *done = FALSE: nothing ran, the queue is as it was. */
static PetscErrorCode VecHIPXLazyTryBatch(PetscBool *done)
{
  Vec      vecs[HIPX_BATCH_MAX_VECS];
  double  *ptrs[HIPX_BATCH_MAX_VECS];
  int      kind[HIPX_LAZY_MAX], ys[HIPX_LAZY_MAX], xs[HIPX_LAZY_MAX], nvec = 0, ndots = -1, da[HIPX_BATCH_MAX_DOTS], db[HIPX_BATCH_MAX_DOTS], start, nops = 0;
  double   sc[HIPX_LAZY_MAX];
  PetscInt n = 0;

  PetscFunctionBegin;
  *done = PETSC_FALSE;
  if (hipx_nlazy < 2 || hipx_lazy_run) PetscFunctionReturn(PETSC_SUCCESS);
  for (start = 0; start + 2 <= hipx_nlazy; start++) {
    PetscBool ok = PETSC_TRUE;
    nvec = 0;
    nops = hipx_nlazy - start;
    n    = hipx_lazy[start].y->map->n;
    for (int k = 0; k < nops && ok; k++) {
      Vec pair[2] = {hipx_lazy[start + k].y, hipx_lazy[start + k].x};
      int slot[2] = {0, 0};
      for (int h = 0; h < 2 && ok; h++) {
        int f = -1;
        for (int u = 0; u < nvec; u++)
          if (vecs[u] == pair[h]) f = u;
        if (f < 0) {
          if (nvec == HIPX_BATCH_MAX_VECS || pair[h]->map->n != n) ok = PETSC_FALSE;
          else {
            vecs[nvec] = pair[h];
            ptrs[nvec] = (double *)VecHIPXGetExt(pair[h])->d_array;
            for (int u = 0; u < nvec; u++)
              if (ptrs[u] == ptrs[nvec]) ok = PETSC_FALSE; /* (two vectors on one buffer: the separate kernels' business) */
            f = nvec++;
          }
        }
        slot[h] = f;
      }
      kind[k] = hipx_lazy[start + k].kind;
      ys[k]   = slot[0];
      xs[k]   = slot[1];
      sc[k]   = (double)hipx_lazy[start + k].s;
    }
    if (ok && hipxVecBatchProgramKnown(nops, kind, ys, xs, nvec)) break;
  }
  if (start + 2 > hipx_nlazy) PetscFunctionReturn(PETSC_SUCCESS); /* no tail of the queue is a compiled program */
  if (start) { /* the operations in front of the batch, in their order */
    PetscBool      run[HIPX_LAZY_MAX];
    const int      keep = nops;
    HipxLazyOp     tail[HIPX_LAZY_MAX];
    for (int k = 0; k < HIPX_LAZY_MAX; k++) run[k] = (PetscBool)(k < start);
    for (int k = 0; k < keep; k++) tail[k] = hipx_lazy[start + k];
    PetscCall(VecHIPXLazyRunMarked(run)); /* (leaves the unmarked ones -- the batch -- recorded, in order) */
    PetscCheck(hipx_nlazy == keep, PETSC_COMM_SELF, PETSC_ERR_PLIB, "lazy queue: %d operations left, expected %d", hipx_nlazy, keep);
    for (int k = 0; k < keep; k++) PetscCheck(hipx_lazy[k].y == tail[k].y && hipx_lazy[k].x == tail[k].x, PETSC_COMM_SELF, PETSC_ERR_PLIB, "lazy queue: order changed");
  }
  PetscCallHIPX(hipxVecBatchAXPYDotsBegin(nops, kind, ys, xs, sc, nvec, ptrs, (hipx_int)n, hipx_bc_slot, &ndots, da, db));
  PetscCheck(ndots >= 0, PETSC_COMM_SELF, PETSC_ERR_PLIB, "libhipx declined a batch it had accepted");
  hipx_bc.n       = ndots;
  hipx_bc.fetched = PETSC_FALSE;
  for (int k = 0; k < ndots; k++) {
    hipx_bc.a[k]    = ((PetscObject)vecs[da[k]])->id;
    hipx_bc.b[k]    = ((PetscObject)vecs[db[k]])->id;
    hipx_bc.live[k] = hipx_rc_on;
  }
  hipx_lazy_stat[5]++;
  hipx_lazy_stat[6] += nops;
  hipx_nlazy = 0;
  *done      = PETSC_TRUE;
  PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode VecHIPXLazyFlush(void)
{
  PetscBool run[HIPX_LAZY_MAX], done;

  PetscFunctionBegin;
  if (!hipx_nlazy || hipx_lazy_run) PetscFunctionReturn(PETSC_SUCCESS);
  PetscCall(VecHIPXLazyTryBatch(&done));
  if (done) PetscFunctionReturn(PETSC_SUCCESS);
  for (int k = 0; k < HIPX_LAZY_MAX; k++) run[k] = PETSC_TRUE;
  PetscCall(VecHIPXLazyRunMarked(run));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* v is about to be looked at or changed: the recorded operations that read or write it must run now -- and with them every EARLIER one that one of
   those may not overtake (VecHIPXLazyConflict).  The rest stays recorded (CG with the unpreconditioned norm: VecNorm(R) runs "r -= a w" and leaves
   "x += a p" for the direction kernel). */
PetscErrorCode VecHIPXLazySync(Vec v)
{
  PetscBool run[HIPX_LAZY_MAX], any = PETSC_FALSE, more = PETSC_TRUE;

  PetscFunctionBegin;
  if (!hipx_nlazy || hipx_lazy_run) PetscFunctionReturn(PETSC_SUCCESS);
  for (int k = 0; k < HIPX_LAZY_MAX; k++) run[k] = PETSC_FALSE;
  for (int k = 0; k < hipx_nlazy; k++)
    if (hipx_lazy[k].y == v || hipx_lazy[k].x == v) run[k] = any = PETSC_TRUE;
  if (!any) PetscFunctionReturn(PETSC_SUCCESS);
  if (hipx_nlazy >= 2) { /* the whole queue as one batch kernel? (round 6) */
    PetscBool done;
    PetscCall(VecHIPXLazyTryBatch(&done));
    if (done) PetscFunctionReturn(PETSC_SUCCESS);
  }
  while (more) {
    more = PETSC_FALSE;
    for (int k = 0; k < hipx_nlazy; k++)
      if (run[k])
        for (int j = 0; j < k; j++)
          if (!run[j] && VecHIPXLazyConflict(&hipx_lazy[j], &hipx_lazy[k])) run[j] = more = PETSC_TRUE;
  }
  PetscCall(VecHIPXLazyRunMarked(run));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* record "y += s x" / "y = x + s y" instead of running it?  Only for two different hipx vectors of one size whose current values are (or now get) on
   the device, long enough for a pass over them to matter. */
static PetscErrorCode VecHIPXLazyRecord(int kind, Vec y, PetscScalar s, Vec x, PetscBool *recorded)
{
  PetscFunctionBegin;
  *recorded = PETSC_FALSE;
  if (!hipx_lazy_on || hipx_lazy_run || x == y || !VecIsHIPX(x) || !VecIsHIPX(y) || x->map->n != y->map->n || y->map->n < hipx_lazy_min) PetscFunctionReturn(PETSC_SUCCESS);
  if (hipx_nlazy == HIPX_LAZY_MAX) PetscCall(VecHIPXLazyFlush());
  /* (a vector that takes part in a recorded operation has its current values on the device by construction: these two calls move nothing for it) */
  PetscCall(VecHIPXCopyToDevice(x));
  PetscCall(VecHIPXCopyToDevice(y));
  VecHIPXRedCacheInvalidate(y);
  hipx_lazy[hipx_nlazy].kind = kind;
  hipx_lazy[hipx_nlazy].y    = y;
  hipx_lazy[hipx_nlazy].x    = x;
  hipx_lazy[hipx_nlazy].s    = s;
  hipx_nlazy++;
  hipx_lazy_stat[0]++;
  if (!hipx_lazy_fin) {
    PetscCall(PetscRegisterFinalize(VecHIPXLazyFinalize));
    hipx_lazy_fin = PETSC_TRUE;
  }
  y->offloadmask = PETSC_OFFLOAD_GPU;
  *recorded      = PETSC_TRUE;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscBool VecHIPXLazyTouches(const HipxLazyOp *o, Vec v) { return (PetscBool)(o->y == v || o->x == v); }

/* VecPointwiseMult(w, x, y) -- or VecCopy(x, w) when y == NULL: PCApply_None -- with "x += s u" recorded and nothing else recorded on w, x, y, u: one
   kernel, sums into the reduction cache */
static PetscErrorCode VecHIPXLazyTryAXPYPointwiseMult(Vec w, Vec x, Vec y, PetscBool *done)
{
  int hit = -1;

  PetscFunctionBegin;
  *done = PETSC_FALSE;
  if (!hipx_nlazy || hipx_lazy_run || w == x || w == y || x == y || !VecIsHIPX(w) || (y && !VecIsHIPX(y)) || w->map->n != x->map->n || (y && y->map->n != x->map->n)) PetscFunctionReturn(PETSC_SUCCESS);
  for (int k = 0; k < hipx_nlazy; k++)
    if (hipx_lazy[k].kind == 1 && hipx_lazy[k].y == x) hit = k;
  /* (w may BE the recorded operation's x: KSPSolve_CG keeps A p in the vector it then overwrites with z, cg.c:145 "W = Z" -- every element is read before it
     is written, by the same thread) */
  if (hit < 0 || hipx_lazy[hit].x == y || !VecIsHIPX(hipx_lazy[hit].x)) PetscFunctionReturn(PETSC_SUCCESS);
  for (int k = 0; k < hipx_nlazy; k++)
    if (k != hit && (VecHIPXLazyTouches(&hipx_lazy[k], w) || VecHIPXLazyTouches(&hipx_lazy[k], x) || (y && VecHIPXLazyTouches(&hipx_lazy[k], y)) || VecHIPXLazyTouches(&hipx_lazy[k], hipx_lazy[hit].x)))
      PetscFunctionReturn(PETSC_SUCCESS);
  {
    const HipxLazyOp o = hipx_lazy[hit];
    PetscErrorCode   ierr;
    hipx_lazy_run      = PETSC_TRUE; /* (the accessors below must not run the queue) */
    ierr               = y ? VecHIPXCopyToDevice(y) : PETSC_SUCCESS;
    if (!ierr) ierr = VecHIPXAllocate(w);
    hipx_lazy_run = PETSC_FALSE;
    if (ierr) hipx_nlazy = 0; /* (as in VecHIPXLazyRunMarked: never leave with the flag set or operations recorded) */
    PetscCall(ierr);
    VecHIPXRedCacheInvalidate(w);
    if (!y) /* w = x: the kernel's multiplication by 1.0 returns its operand, bit for bit */
      PetscCallHIPX(hipxVecAXPYPointwiseMultDotsBegin(VecHIPXGetExt(x)->d_array, o.s, VecHIPXGetExt(o.x)->d_array, VecHIPXGetExt(w)->d_array, NULL, 1.0, x->map->n, VecHIPXRedCacheSlot(HIPX_RC_PWMULT)));
    else {
      /* a diagonal whose entries are all one (nonzero) value -- PCJACOBI on a constant-coefficient operator -- is not streamed: w = x * value, the same
         products.  Looked at once per state of y (two device reductions). */
      static PetscObjectId    cid    = 0;
      static PetscObjectState cstate = 0;
      static PetscBool        cconst = PETSC_FALSE;
      static PetscScalar      cval   = 0.0;
      PetscObjectState        st;
      PetscCall(PetscObjectStateGet((PetscObject)y, &st));
      if (cid != ((PetscObject)y)->id || cstate != st) {
        double   mn = 0.0, mx = 1.0;
        hipx_int im, ix;
        PetscCallHIPX(hipxVecMin(VecHIPXGetExt(y)->d_array, y->map->n, &im, &mn));
        PetscCallHIPX(hipxVecMax(VecHIPXGetExt(y)->d_array, y->map->n, &ix, &mx));
        cid    = ((PetscObject)y)->id;
        cstate = st;
        cconst = (PetscBool)(mn == mx && mn != 0.0);
        cval   = mn;
      }
      PetscCallHIPX(hipxVecAXPYPointwiseMultDotsBegin(VecHIPXGetExt(x)->d_array, o.s, VecHIPXGetExt(o.x)->d_array, VecHIPXGetExt(w)->d_array, cconst ? NULL : VecHIPXGetExt(y)->d_array, cval, x->map->n,
                                                      VecHIPXRedCacheSlot(HIPX_RC_PWMULT)));
    }
    for (int k = hit; k + 1 < hipx_nlazy; k++) hipx_lazy[k] = hipx_lazy[k + 1];
    hipx_nlazy--;
    hipx_lazy_stat[3]++;
    w->offloadmask = PETSC_OFFLOAD_GPU;
    PetscCall(VecHIPXRedCachePut(HIPX_RC_PWMULT, w, x));
    *done = PETSC_TRUE;
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* a second device buffer of the vector's size (kept for the next time) */
static PetscErrorCode VecHIPXAltBuffer(Vec v, PetscScalar **alt)
{
  VecHIPXExt *e = VecHIPXGetExt(v);

  PetscFunctionBegin;
  if (!e->d_alt || e->d_alt_n < v->map->n) {
    if (e->d_alt && e->d_alt_owned) PetscCallHIPX(hipxFree(e->d_alt));
    e->d_alt = NULL;
    PetscCallHIPX(hipxMalloc((void **)&e->d_alt, sizeof(PetscScalar) * (size_t)(v->map->n ? v->map->n : 1)));
    e->d_alt_n     = v->map->n;
    e->d_alt_owned = PETSC_TRUE;
  }
  *alt = e->d_alt;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static void VecHIPXSwapBuffers(Vec v)
{
  VecHIPXExt  *e  = VecHIPXGetExt(v);
  PetscScalar *t  = e->d_array;
  PetscInt     tn = e->d_n;

  const PetscBool to = e->d_owned;

  e->d_array     = e->d_alt;
  e->d_n         = e->d_alt_n;
  e->d_owned     = e->d_alt_owned;
  e->d_alt       = t;
  e->d_alt_n     = tn;
  e->d_alt_owned = to;
}

/* MatMult(A, xx = p, yy = w) with exactly "x += a p; p = z + b p" recorded: the product kernel with the two updates as its prologue.  The kernel
   rewrites p OUT OF PLACE (other workgroups still read the old direction in their halos), and w too when it is the vector z lives in (cg.c:145:
   W = Z -- the halos read z while the rows' owners would be writing w over it): the vectors' two device buffers swap afterwards. */
PetscErrorCode VecHIPXLazyTryCGProduct(hipxMat dA, Vec xx, Vec yy, PetscBool *done)
{
  PetscFunctionBegin;
  *done = PETSC_FALSE;
  if (hipx_nlazy != 2 || hipx_lazy_run || !VecIsHIPX(yy)) PetscFunctionReturn(PETSC_SUCCESS);
  {
    const HipxLazyOp o = hipx_lazy[0], q = hipx_lazy[1]; /* o: x += a p;  q: p = z + b p */
    const PetscBool  walias = (PetscBool)(yy == q.x);
    PetscScalar     *pnew, *wout;
    int              fused = 0;

    if (!(o.kind == 1 && q.kind == 2 && o.x == xx && q.y == xx && o.y != q.x && o.y != xx && q.x != xx && yy != xx && yy != o.y && yy->map->n == xx->map->n)) PetscFunctionReturn(PETSC_SUCCESS);
    /* (a VecPlaceArray-style foreign buffer cannot be swapped; a VecDuplicateVecs slab piece can: ownership travels with the buffers) */
    if ((!VecHIPXGetExt(xx)->d_owned && !VecHIPXGetExt(xx)->slab) || (walias && !VecHIPXGetExt(yy)->d_owned && !VecHIPXGetExt(yy)->slab)) PetscFunctionReturn(PETSC_SUCCESS);
    PetscCall(VecHIPXAltBuffer(xx, &pnew));
    if (walias) PetscCall(VecHIPXAltBuffer(yy, &wout));
    else {
      PetscCall(VecHIPXAllocate(yy));
      wout = VecHIPXGetExt(yy)->d_array;
    }
    PetscCallHIPX(hipxMatMultCGDirectionDotBegin(dA, VecHIPXGetExt(xx)->d_array, pnew, VecHIPXGetExt(q.x)->d_array, 1.0, VecHIPXGetExt(o.y)->d_array, q.s, o.s, NULL, NULL, NULL, wout,
                                                 VecHIPXRedCacheSlot(HIPX_RC_MATMULT), NULL, &fused));
    if (!fused) PetscFunctionReturn(PETSC_SUCCESS); /* nothing was enqueued: the caller's accessors run the queue, then the plain product */
    VecHIPXSwapBuffers(xx); /* the new direction lives in the other buffer now */
    if (walias) VecHIPXSwapBuffers(yy);
    VecHIPXRedCacheInvalidate(yy);
    hipx_nlazy = 0;
    hipx_lazy_stat[4]++;
    yy->offloadmask = PETSC_OFFLOAD_GPU;
    PetscCall(VecHIPXRedCachePut(HIPX_RC_MATMULT, xx, yy));
    *done = PETSC_TRUE;
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* ------------------------------------------------------------------ host array access (ops->getarray & co) */
static PetscErrorCode VecGetArray_HIPX(Vec v, PetscScalar **a)
{
  PetscFunctionBegin;
  PetscCall(VecHIPXLazySync(v));
  VecHIPXRedCacheInvalidate(v);
  PetscCall(VecHIPXCopyToHost(v));
  *a             = *(PetscScalar **)v->data;
  v->offloadmask = PETSC_OFFLOAD_CPU; /* the caller may write: the device copy is stale from now on */
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode VecGetArrayRead_HIPX(Vec v, const PetscScalar **a)
{
  PetscFunctionBegin;
  PetscCall(VecHIPXLazySync(v));
  PetscCall(VecHIPXCopyToHost(v));
  *a = *(PetscScalar **)v->data;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode VecGetArrayWrite_HIPX(Vec v, PetscScalar **a)
{
  PetscFunctionBegin;
  PetscCall(VecHIPXLazySync(v));
  VecHIPXRedCacheInvalidate(v);
  *a             = *(PetscScalar **)v->data;
  v->offloadmask = PETSC_OFFLOAD_CPU;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode VecRestoreArray_HIPX(Vec v, PetscScalar **a)
{
  (void)v;
  (void)a;
  return PETSC_SUCCESS;
}

static PetscErrorCode VecRestoreArrayRead_HIPX(Vec v, const PetscScalar **a)
{
  (void)v;
  (void)a;
  return PETSC_SUCCESS;
}

/* VecGetArray*AndMemType (rvector.c:2290-2560): the device mirror with PETSC_MEMTYPE_HIP.  Installed only with -vec_hipx_memtype:
   a host-only libpetsc gives device pointers to whatever PetscSF the caller uses (vscat.c:41-60), and only the PetscSF type "hipx"
   (sfhipx.c) knows what to do with them -- run with -sf_type hipx. */
PetscBool hipx_vec_memtype_ops = PETSC_FALSE;

static PetscErrorCode VecGetArrayAndMemType_HIPX(Vec v, PetscScalar **a, PetscMemType *m)
{
  PetscFunctionBegin;
  PetscCall(VecHIPXLazySync(v));
  VecHIPXRedCacheInvalidate(v);
  PetscCall(VecHIPXCopyToDevice(v));
  *a             = VecHIPXGetExt(v)->d_array;
  v->offloadmask = PETSC_OFFLOAD_GPU; /* the caller may write on the device */
  if (m) *m = PETSC_MEMTYPE_HIP;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode VecGetArrayReadAndMemType_HIPX(Vec v, const PetscScalar **a, PetscMemType *m)
{
  PetscFunctionBegin;
  PetscCall(VecHIPXLazySync(v));
  PetscCall(VecHIPXCopyToDevice(v));
  *a = VecHIPXGetExt(v)->d_array;
  if (m) *m = PETSC_MEMTYPE_HIP;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode VecGetArrayWriteAndMemType_HIPX(Vec v, PetscScalar **a, PetscMemType *m)
{
  PetscFunctionBegin;
  PetscCall(VecHIPXLazySync(v));
  VecHIPXRedCacheInvalidate(v);
  PetscCall(VecHIPXAllocate(v));
  *a             = VecHIPXGetExt(v)->d_array;
  v->offloadmask = PETSC_OFFLOAD_GPU;
  if (m) *m = PETSC_MEMTYPE_HIP;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode VecRestoreArrayAndMemType_HIPX(Vec v, PetscScalar **a)
{
  (void)v;
  (void)a;
  return PETSC_SUCCESS;
}

static PetscErrorCode VecRestoreArrayReadAndMemType_HIPX(Vec v, const PetscScalar **a)
{
  (void)v;
  (void)a;
  return PETSC_SUCCESS;
}

static PetscErrorCode VecRestoreArrayWriteAndMemType_HIPX(Vec v, PetscScalar **a, PetscMemType *m)
{
  (void)v;
  (void)a;
  (void)m;
  return PETSC_SUCCESS;
}

static PetscErrorCode (*parent_placearray)(Vec, const PetscScalar *);
static PetscErrorCode (*parent_replacearray)(Vec, const PetscScalar *);
static PetscErrorCode (*parent_resetarray)(Vec);
static PetscErrorCode (*parent_destroy_seq)(Vec);
static PetscErrorCode (*parent_destroy_mpi)(Vec);

static PetscErrorCode VecPlaceArray_HIPX(Vec v, const PetscScalar *a)
{
  PetscFunctionBegin;
  PetscCall(VecHIPXLazySync(v));
  VecHIPXRedCacheInvalidate(v);
  PetscCall(VecHIPXCopyToHost(v)); /* the original array must hold current values when it comes back (VecResetArray) */
  PetscCall((*parent_placearray)(v, a));
  v->offloadmask = PETSC_OFFLOAD_CPU;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode VecReplaceArray_HIPX(Vec v, const PetscScalar *a)
{
  PetscFunctionBegin;
  PetscCall(VecHIPXLazySync(v));
  VecHIPXRedCacheInvalidate(v);
  PetscCall((*parent_replacearray)(v, a));
  v->offloadmask = PETSC_OFFLOAD_CPU;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode VecResetArray_HIPX(Vec v)
{
  PetscFunctionBegin;
  PetscCall(VecHIPXLazySync(v));
  VecHIPXRedCacheInvalidate(v);
  /* results computed on the device while the caller's array was placed must reach that array before it is handed back
     (Place / work on the GPU / Reset: PCApply_BJacobi_Multiblock bjacobi.c:886-895; the reference's device vectors do the
     same, veccupmimpl.h:804-807) */
  PetscCall(VecHIPXCopyToHost(v));
  PetscCall((*parent_resetarray)(v));
  v->offloadmask = PETSC_OFFLOAD_CPU;
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* ------------------------------------------------------------------ BLAS-1, local part (shared by seq and mpi) */
#define RD(v, p, t)  PetscCall(VecHIPXGetDeviceRead(v, &p, &t))
#define RDX(v, p, t) PetscCall(VecHIPXRestoreDeviceRead(v, &p, &t))
#define RW(v, p, t)  PetscCall(VecHIPXGetDeviceReadWrite(v, &p, &t))
#define WR(v, p, t)  PetscCall(VecHIPXGetDeviceWrite(v, &p, &t))
#define WRX(v, p, t) PetscCall(VecHIPXRestoreDeviceWrite(v, &p, &t))

static PetscErrorCode VecSet_HIPX(Vec x, PetscScalar alpha) /* VecSet_Seq dvec2.c:642 */
{
  PetscScalar *d;
  void        *t;

  PetscFunctionBegin;
  WR(x, d, t);
  PetscCallHIPX(hipxVecSet(d, x->map->n, alpha));
  WRX(x, d, t);
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode VecCopy_HIPX(Vec x, Vec y) /* VecCopy_Seq bvec2.c:151 */
{
  const PetscScalar *dx;
  PetscScalar       *dy;
  void              *tx, *ty;

  PetscFunctionBegin;
  if (x == y) PetscFunctionReturn(PETSC_SUCCESS);
  {
    PetscBool done;
    PetscCall(VecHIPXLazyTryAXPYPointwiseMult(y, x, NULL, &done)); /* "r -= a w" recorded and z = r asked for (PCApply_None, pcnone.c): one kernel, z.z and z.r with it */
    if (done) PetscFunctionReturn(PETSC_SUCCESS);
  }
  RD(x, dx, tx);
  WR(y, dy, ty);
  PetscCallHIPX(hipxVecCopy(dx, dy, x->map->n));
  WRX(y, dy, ty);
  RDX(x, dx, tx);
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode VecScale_HIPX(Vec x, PetscScalar alpha) /* VecScale_Seq bvec2.c:167 */
{
  PetscScalar *d;
  void        *t;

  PetscFunctionBegin;
  if (alpha == (PetscScalar)0.0) PetscCall(VecSet_HIPX(x, 0.0));
  else if (alpha != (PetscScalar)1.0) {
    RW(x, d, t);
    PetscCallHIPX(hipxVecScale(d, x->map->n, alpha));
    WRX(x, d, t);
    PetscCall(PetscLogFlops(x->map->n));
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode VecSwap_HIPX(Vec x, Vec y)
{
  PetscScalar *dx, *dy;
  void        *tx, *ty;

  PetscFunctionBegin;
  if (x == y) PetscFunctionReturn(PETSC_SUCCESS);
  RW(x, dx, tx);
  RW(y, dy, ty);
  PetscCallHIPX(hipxVecSwap(dx, dy, x->map->n));
  WRX(y, dy, ty);
  WRX(x, dx, tx);
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode VecAXPY_HIPX(Vec y, PetscScalar alpha, Vec x) /* VecAXPY_Seq bvec1.c:70 */
{
  const PetscScalar *dx;
  PetscScalar       *dy;
  void              *tx, *ty;

  PetscFunctionBegin;
  if (alpha == (PetscScalar)0.0) PetscFunctionReturn(PETSC_SUCCESS); /* bvec1.c:75 */
  {
    PetscBool rec;
    PetscCall(VecHIPXLazyRecord(1, y, alpha, x, &rec));
    if (rec) {
      PetscCall(PetscLogFlops(2.0 * y->map->n));
      PetscFunctionReturn(PETSC_SUCCESS);
    }
  }
  RD(x, dx, tx);
  RW(y, dy, ty);
  PetscCallHIPX(hipxVecAXPY(dy, alpha, dx, y->map->n));
  WRX(y, dy, ty);
  RDX(x, dx, tx);
  PetscCall(PetscLogFlops(2.0 * y->map->n)); /* bvec1.c:81 */
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode VecAYPX_HIPX(Vec y, PetscScalar beta, Vec x) /* VecAYPX_Seq dvec2.c:753 */
{
  const PetscScalar *dx;
  PetscScalar       *dy;
  void              *tx, *ty;

  PetscFunctionBegin;
  if (beta == (PetscScalar)0.0) {
    PetscCall(VecCopy_HIPX(x, y));
    PetscFunctionReturn(PETSC_SUCCESS);
  }
  {
    PetscBool rec;
    PetscCall(VecHIPXLazyRecord(2, y, beta, x, &rec));
    if (rec) {
      PetscCall(PetscLogFlops((beta == (PetscScalar)-1.0 ? 1.0 : 2.0) * y->map->n));
      PetscFunctionReturn(PETSC_SUCCESS);
    }
  }
  RD(x, dx, tx);
  RW(y, dy, ty);
  PetscCallHIPX(hipxVecAYPX(dy, beta, dx, y->map->n));
  WRX(y, dy, ty);
  RDX(x, dx, tx);
  PetscCall(PetscLogFlops((beta == (PetscScalar)-1.0 ? 1.0 : 2.0) * y->map->n)); /* dvec2.c:769,776 */
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode VecAXPBY_HIPX(Vec y, PetscScalar a, PetscScalar b, Vec x) /* VecAXPBY_Seq bvec1.c:91 */
{
  const PetscScalar *dx;
  PetscScalar       *dy;
  void              *tx, *ty;

  PetscFunctionBegin;
  if (a == (PetscScalar)0.0) {
    PetscCall(VecScale_HIPX(y, b));
    PetscFunctionReturn(PETSC_SUCCESS);
  }
  RD(x, dx, tx);
  RW(y, dy, ty);
  PetscCallHIPX(hipxVecAXPBY(dy, a, b, dx, y->map->n));
  WRX(y, dy, ty);
  RDX(x, dx, tx);
  PetscCall(PetscLogFlops(3.0 * y->map->n));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode VecWAXPY_HIPX(Vec w, PetscScalar alpha, Vec x, Vec y) /* VecWAXPY_Seq dvec2.c:791 */
{
  const PetscScalar *dx, *dy;
  PetscScalar       *dw;
  void              *tx, *ty, *tw;

  PetscFunctionBegin;
  RD(x, dx, tx);
  RD(y, dy, ty);
  WR(w, dw, tw);
  PetscCallHIPX(hipxVecWAXPY(dw, alpha, dx, dy, w->map->n));
  WRX(w, dw, tw);
  RDX(y, dy, ty);
  RDX(x, dx, tx);
  PetscCall(PetscLogFlops(2.0 * w->map->n));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode VecAXPBYPCZ_HIPX(Vec z, PetscScalar a, PetscScalar b, PetscScalar c, Vec x, Vec y) /* bvec1.c:120 */
{
  const PetscScalar *dx, *dy;
  PetscScalar       *dz;
  void              *tx, *ty, *tz;

  PetscFunctionBegin;
  RD(x, dx, tx);
  RD(y, dy, ty);
  RW(z, dz, tz);
  PetscCallHIPX(hipxVecAXPBYPCZ(dz, a, b, c, dx, dy, z->map->n));
  WRX(z, dz, tz);
  RDX(y, dy, ty);
  RDX(x, dx, tx);
  PetscCall(PetscLogFlops(4.0 * z->map->n));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode VecPointwiseMult_HIPX(Vec w, Vec x, Vec y) /* VecPointwiseMult_Seq bvec2.c:72 (w may alias x or y) */
{
  const PetscScalar *dx, *dy;
  PetscScalar       *dw;
  void              *tx, *ty, *tw;

  PetscFunctionBegin;
  {
    PetscBool done;
    PetscCall(VecHIPXLazyTryAXPYPointwiseMult(w, x, y, &done)); /* "r -= a w" recorded and z = r .* d asked for: one kernel */
    if (done) {
      PetscCall(PetscLogFlops(w->map->n));
      PetscFunctionReturn(PETSC_SUCCESS);
    }
  }
  RD(x, dx, tx);
  RD(y, dy, ty);
  if (w == x || w == y) RW(w, dw, tw);
  else WR(w, dw, tw);
  if (w != x && w != y && !tx && !ty && !tw && dw != dx && dw != dy && w->map->n > 0 && VecHIPXRedCacheWanted(HIPX_RC_PWMULT)) {
    /* PCApply_Jacobi inside a Krylov loop: the sums w . w and w . x ride along (hipxplugin.h: reduction cache) */
    PetscCallHIPX(hipxVecPointwiseMultDotsBegin(dw, dx, dy, w->map->n, VecHIPXRedCacheSlot(HIPX_RC_PWMULT)));
    PetscCall(VecHIPXRedCachePut(HIPX_RC_PWMULT, w, x));
  } else PetscCallHIPX(hipxVecPointwiseMult(dw, dx, dy, w->map->n));
  WRX(w, dw, tw);
  RDX(y, dy, ty);
  RDX(x, dx, tx);
  PetscCall(PetscLogFlops(w->map->n)); /* bvec2.c:95 */
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode VecPointwiseDivide_HIPX(Vec w, Vec x, Vec y) /* bvec2.c:99 */
{
  const PetscScalar *dx, *dy;
  PetscScalar       *dw;
  void              *tx, *ty, *tw;

  PetscFunctionBegin;
  RD(x, dx, tx);
  RD(y, dy, ty);
  if (w == x || w == y) RW(w, dw, tw);
  else WR(w, dw, tw);
  PetscCallHIPX(hipxVecPointwiseDivide(dw, dx, dy, w->map->n));
  WRX(w, dw, tw);
  RDX(y, dy, ty);
  RDX(x, dx, tx);
  PetscCall(PetscLogFlops(w->map->n));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode VecReciprocal_HIPX(Vec x) /* VecReciprocal_Default vinv.c:1208 */
{
  PetscScalar *d;
  void        *t;

  PetscFunctionBegin;
  RW(x, d, t);
  PetscCallHIPX(hipxVecReciprocal(d, x->map->n));
  WRX(x, d, t);
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode VecAbs_HIPX(Vec x)
{
  PetscScalar *d;
  void        *t;

  PetscFunctionBegin;
  RW(x, d, t);
  PetscCallHIPX(hipxVecAbs(d, x->map->n));
  WRX(x, d, t);
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode VecShift_HIPX(Vec x, PetscScalar s)
{
  PetscScalar *d;
  void        *t;

  PetscFunctionBegin;
  RW(x, d, t);
  PetscCallHIPX(hipxVecShift(d, x->map->n, s));
  WRX(x, d, t);
  PetscFunctionReturn(PETSC_SUCCESS);
}

#define HIPX_MAXV 64
static PetscErrorCode VecMAXPY_HIPX(Vec y, PetscInt nv, const PetscScalar *alpha, Vec *x) /* VecMAXPY_Seq dvec2.c:658 */
{
  PetscScalar *dy;
  void        *ty;

  PetscFunctionBegin;
  RW(y, dy, ty);
  for (PetscInt s = 0; s < nv; s += HIPX_MAXV) { /* batches keep the reference's (nv & 3)-then-fours grouping only for nv <= 64 */
    const PetscScalar *dx[HIPX_MAXV];
    void              *tx[HIPX_MAXV];
    PetscInt           m = PetscMin(HIPX_MAXV, nv - s);
    for (PetscInt j = 0; j < m; j++) RD(x[s + j], dx[j], tx[j]);
    PetscCallHIPX(hipxVecMAXPY(dy, m, alpha + s, (const double *const *)dx, y->map->n));
    for (PetscInt j = 0; j < m; j++) RDX(x[s + j], dx[j], tx[j]);
  }
  WRX(y, dy, ty);
  PetscCall(PetscLogFlops(nv * 2.0 * y->map->n));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* VecMAXPBY (rvector.c:1394-1440): y = beta y + sum_j alpha[j] x[j].  One pass for up to 36 vectors (KSPGMRESBuildSoln, gmres.c:337);
   longer lists: the interface's own decomposition (VecSet / VecScale, then VecMAXPY) */
static PetscErrorCode VecMAXPBY_HIPX(Vec y, PetscInt nv, const PetscScalar *alpha, PetscScalar beta, Vec *x)
{
  PetscFunctionBegin;
  if (nv <= HIPX_MAXV) {
    PetscScalar       *dy;
    void              *ty;
    const PetscScalar *dx[HIPX_MAXV];
    void              *tx[HIPX_MAXV];
    if (beta == 0.0) WR(y, dy, ty);
    else RW(y, dy, ty);
    for (PetscInt j = 0; j < nv; j++) RD(x[j], dx[j], tx[j]);
    PetscCallHIPX(hipxVecMAXPBY(dy, nv, alpha, beta, (const double *const *)dx, y->map->n));
    for (PetscInt j = 0; j < nv; j++) RDX(x[j], dx[j], tx[j]);
    WRX(y, dy, ty);
    PetscCall(PetscLogFlops(nv * 2.0 * y->map->n));
  } else {
    if (beta == 0.0) PetscCall(VecSet(y, 0.0));
    else PetscCall(VecScale(y, beta));
    PetscCall(VecMAXPY(y, nv, alpha, x));
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode VecDotLocal_HIPX(Vec x, Vec y, PetscScalar *z) /* VecDot_Seq / VecTDot_Seq bvec1.c:10-49 (real scalars: identical) */
{
  const PetscScalar *dx, *dy;
  void              *tx, *ty;

  PetscFunctionBegin;
  PetscCall(VecHIPXLazySync(x)); /* (a recorded batch that touches x or y runs now -- and may leave exactly this sum behind) */
  PetscCall(VecHIPXLazySync(y));
  {
    PetscBool hit;
    double    val = 0.0;
    PetscCall(VecHIPXBatchCacheGet(x, y, &val, &hit));
    if (hit) {
      *z = val;
      PetscCall(PetscLogFlops(PetscMax(2.0 * x->map->n - 1, 0.0)));
      PetscFunctionReturn(PETSC_SUCCESS);
    }
  }
  if (x != y) { /* a sum the kernel that wrote one of the two vectors left behind? (reduction cache, hipxplugin.h) */
    const double *c;
    PetscCall(VecHIPXRedCacheGet(HIPX_RC_MATMULT, x, y, PETSC_FALSE, &c));
    if (c) *z = c[0];
    else {
      PetscCall(VecHIPXRedCacheGet(HIPX_RC_PWMULT, x, y, PETSC_FALSE, &c));
      if (c) *z = c[1];
    }
    if (c) {
      PetscCall(PetscLogFlops(PetscMax(2.0 * x->map->n - 1, 0.0)));
      PetscFunctionReturn(PETSC_SUCCESS);
    }
  }
  RD(x, dx, tx);
  RD(y, dy, ty);
  PetscCallHIPX(hipxVecDot(dx, dy, x->map->n, z));
  RDX(y, dy, ty);
  RDX(x, dx, tx);
  PetscCall(PetscLogFlops(PetscMax(2.0 * x->map->n - 1, 0.0))); /* bvec1.c:22 */
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode VecMDotLocal_HIPX(Vec x, PetscInt nv, const Vec y[], PetscScalar *z) /* VecMDot_Seq dvec2.c:83 */
{
  const PetscScalar *dx;
  void              *tx;

  PetscFunctionBegin;
  RD(x, dx, tx);
  for (PetscInt s = 0; s < nv; s += 32) {
    const PetscScalar *dy[32];
    void              *ty[32];
    PetscInt           m = PetscMin(32, nv - s);
    for (PetscInt j = 0; j < m; j++) RD(y[s + j], dy[j], ty[j]);
    PetscCallHIPX(hipxVecMDot(dx, m, (const double *const *)dy, x->map->n, z + s));
    for (PetscInt j = 0; j < m; j++) RDX(y[s + j], dy[j], ty[j]);
  }
  RDX(x, dx, tx);
  PetscCall(PetscLogFlops(PetscMax(nv * (2.0 * x->map->n - 1), 0.0)));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode VecNormLocal_HIPX(Vec x, NormType type, PetscReal *z) /* VecNorm_Seq bvec2.c:185 */
{
  const PetscScalar *dx;
  void              *tx;
  int                t = (type == NORM_1) ? 0 : (type == NORM_2) ? 1 : (type == NORM_FROBENIUS) ? 2 : (type == NORM_INFINITY) ? 3 : 4;

  PetscFunctionBegin;
  PetscCall(VecHIPXLazySync(x));
  if (type == NORM_2 || type == NORM_FROBENIUS) { /* x . x left behind by a batch kernel: sqrt of it, bvec2.c:204 */
    PetscBool hit;
    double    val = 0.0;
    PetscCall(VecHIPXBatchCacheGet(x, x, &val, &hit));
    if (hit) {
      *z = PetscSqrtReal(val);
      PetscCall(PetscLogFlops(PetscMax(2.0 * x->map->n - 1, 0.0)));
      PetscFunctionReturn(PETSC_SUCCESS);
    }
  }
  if ((type == NORM_2 || type == NORM_FROBENIUS) && hipx_rc[HIPX_RC_PWMULT].live && hipx_rc[HIPX_RC_PWMULT].a == ((PetscObject)x)->id) {
    HipxRedCacheEntry *e = &hipx_rc[HIPX_RC_PWMULT]; /* x was written by the fused VecPointwiseMult: sqrt(x . x), bvec2.c:204 */
    if (!e->fetched) {
      PetscCallHIPX(hipxRedEnd(hipx_rc_slot[HIPX_RC_PWMULT], 2, e->v));
      e->fetched = PETSC_TRUE;
    }
    e->used = PETSC_TRUE;
    *z      = PetscSqrtReal(e->v[0]);
    PetscCall(PetscLogFlops(PetscMax(2.0 * x->map->n - 1, 0.0)));
    PetscFunctionReturn(PETSC_SUCCESS);
  }
  RD(x, dx, tx);
  PetscCallHIPX(hipxVecNorm(dx, x->map->n, t, z));
  RDX(x, dx, tx);
  if (type == NORM_2 || type == NORM_FROBENIUS) PetscCall(PetscLogFlops(PetscMax(2.0 * x->map->n - 1, 0.0)));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode VecDotNorm2Local_HIPX(Vec s, Vec t, PetscScalar *dp, PetscScalar *nm)
{
  const PetscScalar *ds, *dt;
  void              *ts, *tt;

  PetscFunctionBegin;
  RD(s, ds, ts);
  RD(t, dt, tt);
  PetscCallHIPX(hipxVecDotNorm2(ds, dt, s->map->n, dp, nm));
  RDX(t, dt, tt);
  RDX(s, ds, ts);
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode VecSumLocal_HIPX(Vec x, PetscScalar *sum)
{
  const PetscScalar *dx;
  void              *tx;

  PetscFunctionBegin;
  RD(x, dx, tx);
  PetscCallHIPX(hipxVecSum(dx, x->map->n, sum));
  RDX(x, dx, tx);
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode VecMaxLocal_HIPX(Vec x, PetscInt *idx, PetscReal *z) /* dvec2.c:592-640 */
{
  const PetscScalar *dx;
  void              *tx;
  hipx_int           i;

  PetscFunctionBegin;
  RD(x, dx, tx);
  PetscCallHIPX(hipxVecMax(dx, x->map->n, &i, z));
  RDX(x, dx, tx);
  if (!x->map->n) *z = PETSC_MIN_REAL;
  if (idx) *idx = i;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode VecMinLocal_HIPX(Vec x, PetscInt *idx, PetscReal *z)
{
  const PetscScalar *dx;
  void              *tx;
  hipx_int           i;

  PetscFunctionBegin;
  RD(x, dx, tx);
  PetscCallHIPX(hipxVecMin(dx, x->map->n, &i, z));
  RDX(x, dx, tx);
  if (!x->map->n) *z = PETSC_MAX_REAL;
  if (idx) *idx = i;
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* ------------------------------------------------------------------ MPI wrappers (pvecimpl.h:97-175: local kernel + MPI_Allreduce) */
static PetscErrorCode VecDot_MPIHIPX(Vec x, Vec y, PetscScalar *z)
{
  PetscFunctionBegin;
  PetscCall(VecXDot_MPI_Default(x, y, z, VecDotLocal_HIPX));
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode VecMDot_MPIHIPX(Vec x, PetscInt nv, const Vec y[], PetscScalar *z)
{
  PetscFunctionBegin;
  PetscCall(VecMXDot_MPI_Default(x, nv, y, z, VecMDotLocal_HIPX));
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode VecNorm_MPIHIPX(Vec x, NormType type, PetscReal *z)
{
  PetscFunctionBegin;
  PetscCall(VecNorm_MPI_Default(x, type, z, VecNormLocal_HIPX));
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode VecDotNorm2_MPIHIPX(Vec s, Vec t, PetscScalar *dp, PetscScalar *nm)
{
  PetscFunctionBegin;
  PetscCall(VecDotNorm2_MPI_Default(s, t, dp, nm, VecDotNorm2Local_HIPX));
  PetscFunctionReturn(PETSC_SUCCESS);
}
/* ops->sum is the LOCAL sum for every type: VecSum() itself reduces over the communicator (vinv.c:1548-1558) */
static PetscErrorCode VecMax_MPIHIPX(Vec x, PetscInt *idx, PetscReal *z) /* pvec2.c:56-63 with the device local op */
{
  const MPI_Op ops[] = {MPIU_MAXLOC, MPIU_MAX};

  PetscFunctionBegin;
  PetscCall(VecMinMax_MPI_Default(x, idx, z, VecMaxLocal_HIPX, ops));
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode VecMin_MPIHIPX(Vec x, PetscInt *idx, PetscReal *z) /* pvec2.c:65-72 */
{
  const MPI_Op ops[] = {MPIU_MINLOC, MPIU_MIN};

  PetscFunctionBegin;
  PetscCall(VecMinMax_MPI_Default(x, idx, z, VecMinLocal_HIPX, ops));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* ------------------------------------------------------------------ life cycle */
static PetscErrorCode VecHIPXFreeDevice(Vec v)
{
  VecHIPXExt *e = VecHIPXGetExt(v);

  PetscFunctionBegin;
  if (e->magic == VECHIPX_MAGIC) PetscCall(VecHIPXLazySync(v));
  VecHIPXRedCacheInvalidate(v);
  if (e->magic == VECHIPX_MAGIC && e->d_array && e->d_owned) PetscCallHIPX(hipxFree(e->d_array));
  if (e->magic == VECHIPX_MAGIC && e->slab) {
    if (--e->slab->refs == 0) {
      PetscCallHIPX(hipxFree(e->slab->base));
      PetscCall(PetscFree(e->slab));
    }
    e->slab = NULL;
  }
  if (e->magic == VECHIPX_MAGIC && e->d_alt && e->d_alt_owned) PetscCallHIPX(hipxFree(e->d_alt));
  e->d_alt   = NULL;
  e->d_array = NULL;
  e->magic   = 0;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode VecDestroy_SeqHIPX(Vec v)
{
  PetscFunctionBegin;
  PetscCall(VecHIPXFreeDevice(v));
  PetscCall((*parent_destroy_seq)(v)); /* frees the host array and v->data (bvec2.c:621) */
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode VecDestroy_MPIHIPX(Vec v)
{
  PetscFunctionBegin;
  PetscCall(VecHIPXFreeDevice(v));
  PetscCall((*parent_destroy_mpi)(v));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* VecDuplicate_MPI (pbvec.c:23-66) builds the duplicate with VecCreate_MPI_Private and copies win's ops table over it, which
   would leave our ops on a plain Vec_MPI block without the VecHIPXExt tail.  Like the reference's device subclasses
   (VecDuplicate_MPICUPM), duplicate through our own creator instead; the layout object is shared, as the parent does. */
static PetscErrorCode VecDuplicate_MPIHIPX(Vec win, Vec *V)
{
  Vec v;

  PetscFunctionBegin;
  PetscCheck(!((Vec_MPI *)win->data)->localrep, PetscObjectComm((PetscObject)win), PETSC_ERR_SUP, "ghosted vectors are not supported by VECMPIHIPX");
  PetscCall(VecCreate(PetscObjectComm((PetscObject)win), &v));
  PetscCall(PetscLayoutReference(win->map, &v->map));
  PetscCall(VecSetType(v, VECMPIHIPX));
  v->ops[0]             = win->ops[0];
  v->stash.donotstash   = win->stash.donotstash;
  v->stash.ignorenegidx = win->stash.ignorenegidx;
  v->stash.bs           = win->stash.bs;
  v->bstash.bs          = win->bstash.bs;
  PetscCall(PetscObjectListDuplicate(((PetscObject)win)->olist, &((PetscObject)v)->olist));
  PetscCall(PetscFunctionListDuplicate(((PetscObject)win)->qlist, &((PetscObject)v)->qlist));
  *V = v;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode VecDuplicateVecs_HIPX(Vec w, PetscInt m, Vec *V[])
{
  static PetscBool slab_on = PETSC_TRUE, looked = PETSC_FALSE;
  static PetscInt  pad = 544; /* doubles between two vectors of a slab: 4 KiB + 256 B, so that a stride of 2^k bytes (n = 256^3) does not put element i of every vector on one HBM channel */
  const PetscInt   n   = w->map->n;
  size_t           ld;

  PetscFunctionBegin;
  if (!looked) {
    PetscCall(PetscOptionsGetBool(NULL, NULL, "-vec_hipx_duplicatevecs_slab", &slab_on, NULL));
    PetscCall(PetscOptionsGetInt(NULL, NULL, "-vec_hipx_slab_pad", &pad, NULL));
    looked = PETSC_TRUE;
  }
  ld = (((size_t)(n > 0 ? n : 1) + (size_t)(pad > 0 ? pad : 0)) + 1) & ~(size_t)1; /* every vector 16-byte aligned */
  PetscCall(PetscMalloc1(m, V));
  for (PetscInt i = 0; i < m; i++) PetscCall(VecDuplicate(w, &(*V)[i]));
  /* VecDuplicateVecs_Seq_GEMV (bvec2.c:670-720): the m vectors' storage is ONE array (there: the host array, so that VecMDot / VecMAXPY become dgemv; here: the
     device mirrors -- one hipMalloc instead of m, the basis contiguous for the wide MDot / MAXPY kernels).  The host arrays stay the parent's. */
  /* (m >= 8: a Krylov basis.  The three or five work vectors of a CG keep allocations of their own: measured on 7-pt 256^3, stock KSPCG, 400 iterations --
     separate allocations 109-112 ms, one slab with the 4352-byte skew 113-114, without skew 118, with a 527 KB skew 123: where r, z, p lie relative to each
     other moves the fused kernels by a few per cent, and the allocator's own placement is the best of those tried) */
  /* Round 6: the work vectors of a Krylov method (KSPSetWorkVecs -> VecDuplicateVecs at KSPSetUp) get their device mirrors NOW when the vector they are modelled
     on lives on the device: the hipMalloc of a 134 MB buffer costs milliseconds and would otherwise happen inside the first KSPSolve, at each vector's first use
     (measured: the first of two identical solves 30 ms longer than the second) -- the reference allocates its work vectors at set-up too (VecDuplicateVecs_Default). */
  if (!(slab_on && m >= 8) && n > 0 && VecIsHIPX(w) && VecHIPXGetExt(w)->d_array && VecIsHIPX((*V)[0]))
    for (PetscInt i = 0; i < m; i++) PetscCall(VecHIPXAllocate((*V)[i]));
  if (slab_on && m >= 8 && n > 0 && VecIsHIPX((*V)[0])) {
    VecHIPXSlab *sl;
    PetscCall(PetscNew(&sl));
    PetscCallHIPX(hipxMalloc((void **)&sl->base, sizeof(PetscScalar) * ld * (size_t)m));
    sl->refs = m;
    for (PetscInt i = 0; i < m; i++) {
      VecHIPXExt *e = VecHIPXGetExt((*V)[i]);
      if (e->d_array && e->d_owned) PetscCallHIPX(hipxFree(e->d_array));
      e->d_array = sl->base + ld * (size_t)i;
      e->d_n     = n;
      e->d_owned = PETSC_FALSE;
      e->slab    = sl;
    }
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode VecDestroyVecs_HIPX(PetscInt m, Vec V[])
{
  PetscFunctionBegin;
  for (PetscInt i = 0; i < m; i++) PetscCall(VecDestroy(&V[i]));
  PetscCall(PetscFree(V));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* re-allocate v->data as [parent struct | VecHIPXExt] */
static PetscErrorCode VecHIPXExtend(Vec v, size_t parent_size)
{
  char       *blk;
  VecHIPXExt *e;

  PetscFunctionBegin;
  PetscCall(PetscCalloc1(VECHIPX_EXT_OFF + sizeof(VecHIPXExt), &blk));
  PetscCall(PetscMemcpy(blk, v->data, parent_size));
  PetscCall(PetscFree(v->data));
  v->data  = blk;
  e        = VecHIPXGetExt(v);
  e->magic = VECHIPX_MAGIC;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static void VecHIPXInstallLocalOps(Vec v)
{
  struct _VecOps *o = v->ops;
  o->duplicatevecs    = VecDuplicateVecs_HIPX;
  o->destroyvecs      = VecDestroyVecs_HIPX;
  o->scale            = VecScale_HIPX;
  o->copy             = VecCopy_HIPX;
  o->set              = VecSet_HIPX;
  o->swap             = VecSwap_HIPX;
  o->axpy             = VecAXPY_HIPX;
  o->axpby            = VecAXPBY_HIPX;
  o->maxpy            = VecMAXPY_HIPX;
  o->aypx             = VecAYPX_HIPX;
  o->waxpy            = VecWAXPY_HIPX;
  o->axpbypcz         = VecAXPBYPCZ_HIPX;
  o->pointwisemult    = VecPointwiseMult_HIPX;
  o->pointwisedivide  = VecPointwiseDivide_HIPX;
  o->reciprocal       = VecReciprocal_HIPX;
  o->abs              = VecAbs_HIPX;
  o->shift            = VecShift_HIPX;
  o->maxpby           = VecMAXPBY_HIPX;
  o->getarray         = VecGetArray_HIPX;
  o->restorearray     = VecRestoreArray_HIPX;
  o->getarrayread     = VecGetArrayRead_HIPX;
  o->restorearrayread = VecRestoreArrayRead_HIPX;
  o->getarraywrite    = VecGetArrayWrite_HIPX;
  o->restorearraywrite = VecRestoreArray_HIPX;
  o->placearray       = VecPlaceArray_HIPX;
  o->replacearray     = VecReplaceArray_HIPX;
  o->resetarray       = VecResetArray_HIPX;
  o->dot_local        = VecDotLocal_HIPX;
  o->tdot_local       = VecDotLocal_HIPX;
  o->mdot_local       = VecMDotLocal_HIPX;
  o->mtdot_local      = VecMDotLocal_HIPX;
  o->norm_local       = VecNormLocal_HIPX;
  if (hipx_vec_memtype_ops) {
    o->getarrayandmemtype          = VecGetArrayAndMemType_HIPX;
    o->restorearrayandmemtype      = VecRestoreArrayAndMemType_HIPX;
    o->getarrayreadandmemtype      = VecGetArrayReadAndMemType_HIPX;
    o->restorearrayreadandmemtype  = VecRestoreArrayReadAndMemType_HIPX;
    o->getarraywriteandmemtype     = VecGetArrayWriteAndMemType_HIPX;
    o->restorearraywriteandmemtype = VecRestoreArrayWriteAndMemType_HIPX;
  }
}

PetscErrorCode VecCreate_SeqHIPX(Vec v)
{
  PetscFunctionBegin;
  PetscCall(VecHIPXInitRuntime());
  PetscCall(VecSetType(v, VECSEQ)); /* allocates + zeroes the host array, seeds the norm cache (bvec3.c:22-41) */
  if (!parent_destroy_seq) {
    parent_destroy_seq  = v->ops->destroy;
    parent_placearray   = v->ops->placearray;
    parent_replacearray = v->ops->replacearray;
    parent_resetarray   = v->ops->resetarray;
  }
  PetscCall(VecHIPXExtend(v, sizeof(Vec_Seq)));
  VecHIPXInstallLocalOps(v);
  v->ops->dot      = VecDotLocal_HIPX;
  v->ops->tdot     = VecDotLocal_HIPX;
  v->ops->mdot     = VecMDotLocal_HIPX;
  v->ops->mtdot    = VecMDotLocal_HIPX;
  v->ops->norm     = VecNormLocal_HIPX;
  v->ops->dotnorm2 = VecDotNorm2Local_HIPX;
  v->ops->sum      = VecSumLocal_HIPX;
  v->ops->max      = VecMaxLocal_HIPX;
  v->ops->min      = VecMinLocal_HIPX;
  v->ops->destroy  = VecDestroy_SeqHIPX;
  v->offloadmask   = PETSC_OFFLOAD_CPU;
  PetscCall(PetscObjectChangeTypeName((PetscObject)v, VECSEQHIPX));
  PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode VecCreate_MPIHIPX(Vec v)
{
  PetscFunctionBegin;
  PetscCall(VecHIPXInitRuntime());
  PetscCall(VecSetType(v, VECMPI)); /* VecCreate_MPI is exported (pvecimpl.h:74) but the registry route needs no symbol at all */
  if (!parent_destroy_mpi) {
    parent_destroy_mpi = v->ops->destroy;
    if (!parent_placearray) {
      parent_placearray   = v->ops->placearray;
      parent_replacearray = v->ops->replacearray;
      parent_resetarray   = v->ops->resetarray;
    }
  }
  PetscCall(VecHIPXExtend(v, sizeof(Vec_MPI)));
  VecHIPXInstallLocalOps(v);
  v->ops->dot      = VecDot_MPIHIPX;
  v->ops->tdot     = VecDot_MPIHIPX;
  v->ops->mdot     = VecMDot_MPIHIPX;
  v->ops->mtdot    = VecMDot_MPIHIPX;
  v->ops->norm     = VecNorm_MPIHIPX;
  v->ops->dotnorm2 = VecDotNorm2_MPIHIPX;
  v->ops->sum      = VecSumLocal_HIPX;
  v->ops->max      = VecMax_MPIHIPX;
  v->ops->min      = VecMin_MPIHIPX;
  v->ops->destroy  = VecDestroy_MPIHIPX;
  v->ops->duplicate = VecDuplicate_MPIHIPX;
  v->offloadmask   = PETSC_OFFLOAD_CPU;
  PetscCall(PetscObjectChangeTypeName((PetscObject)v, VECMPIHIPX));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* root type: dispatch on communicator size like VecCreate_Standard (pbvec.c:666-675) */
PetscErrorCode VecCreate_HIPX(Vec v)
{
  PetscMPIInt size;

  PetscFunctionBegin;
  PetscCallMPI(MPI_Comm_size(PetscObjectComm((PetscObject)v), &size));
  if (size == 1) PetscCall(VecSetType(v, VECSEQHIPX));
  else PetscCall(VecSetType(v, VECMPIHIPX));
  PetscFunctionReturn(PETSC_SUCCESS);
}
