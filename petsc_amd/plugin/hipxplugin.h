/*
 * hipxplugin.h -- internal header of libpetschipx.so, the PETSc-facing side of the drop-in boundary.
 *
 * The plugin registers VECSEQHIPX / VECMPIHIPX / VECHIPX, MATSEQAIJHIPX / MATMPIAIJHIPX / MATAIJHIPX and
 * PCJACOBIHIPX with an UNMODIFIED libpetsc (loaded with -dll_prepend, src/sys/dll/reg.c:79) and fills the
 * reference's ops tables with thin C functions that call the kernel ABI (include/hipx.h).
 * It only uses symbols a default (hidden-visibility) libpetsc exports; parent-class behaviour is obtained by
 * calling the exported parent creator and capturing its ops table (SURVEY.md Appendix B recipe).
 */
#pragma once
#include <petsc/private/vecimpl.h>
#include <petsc/private/matimpl.h>
#include <petsc/private/pcimpl.h>
#include <petsc/private/kspimpl.h>
#include <../src/vec/vec/impls/dvecimpl.h>
#include <../src/vec/vec/impls/mpi/pvecimpl.h>
#include <../src/mat/impls/aij/seq/aij.h>
#include <../src/mat/impls/aij/mpi/mpiaij.h>
#include "hipx.h"

#define VECSEQHIPX    "seqhipx"
#define VECMPIHIPX    "mpihipx"
#define VECHIPX       "hipx"
#define MATSEQAIJHIPX "seqaijhipx"
#define MATMPIAIJHIPX "mpiaijhipx"
#define MATAIJHIPX    "aijhipx"
#define PCJACOBIHIPX  "jacobihipx"

/* HIP / kernel-library failures surface as PETSC_ERR_GPU with the library's message (no silent fallback) */
#define PetscCallHIPX(call) \
  do { \
    int hipx_ierr_ = (call); \
    PetscCheck(!hipx_ierr_, PETSC_COMM_SELF, PETSC_ERR_GPU, "libhipx: %s", hipxGetErrorString()); \
  } while (0)

/* Device mirror of a vector.  It lives behind the parent's data block: v->data is re-allocated as
   [ Vec_Seq | Vec_MPI (whichever is larger) ][ VecHIPXExt ], so the parent ops keep working on the front part. */
typedef struct {
  PetscScalar *d_array;      /* device copy, NULL until first needed */
  PetscInt     d_n;          /* allocated length */
  PetscBool    d_owned;      /* PETSC_FALSE for sub-arrays of a VecDuplicateVecs slab */
  PetscInt     magic;
  PetscScalar *d_alt;        /* second device buffer (lazy fusion: the CG direction vector is rewritten out of place by the product kernel), or NULL */
  PetscInt     d_alt_n;
  PetscBool    d_alt_owned;  /* (after a swap the slab piece of a VecDuplicateVecs vector is the "second" buffer: not ours to free) */
  struct VecHIPXSlab_s *slab; /* the VecDuplicateVecs slab d_array points into (reference-counted), or NULL */
} VecHIPXExt;
/* One device allocation for the m vectors of a VecDuplicateVecs call (the reference: VecDuplicateVecs_Seq_GEMV, bvec2.c:670 -- a Krylov basis is one array of
   leading dimension lda, which is what lets VecMDot / VecMAXPY be GEMVs there).  Freed when its last vector goes. */
typedef struct VecHIPXSlab_s {
  PetscScalar *base;
  PetscInt     refs;
} VecHIPXSlab;

#define VECHIPX_MAGIC   0x48495058
#define VECHIPX_EXT_OFF ((sizeof(Vec_MPI) > sizeof(Vec_Seq) ? sizeof(Vec_MPI) : sizeof(Vec_Seq)) + 16 - ((sizeof(Vec_MPI) > sizeof(Vec_Seq) ? sizeof(Vec_MPI) : sizeof(Vec_Seq)) % 16))

static inline VecHIPXExt *VecHIPXGetExt(Vec v) { return (VecHIPXExt *)((char *)v->data + VECHIPX_EXT_OFF); }

PETSC_INTERN PetscBool      VecIsHIPX(Vec v);
PETSC_INTERN PetscErrorCode VecHIPXInitRuntime(void);
/* device access with host/device coherence driven by v->offloadmask (include/petscdevicetypes.h:239-246) */
PETSC_INTERN PetscErrorCode VecHIPXGetDeviceRead(Vec v, const PetscScalar **d, void **tmp);
PETSC_INTERN PetscErrorCode VecHIPXRestoreDeviceRead(Vec v, const PetscScalar **d, void **tmp);
PETSC_INTERN PetscErrorCode VecHIPXGetDeviceWrite(Vec v, PetscScalar **d, void **tmp);  /* contents undefined on entry */
PETSC_INTERN PetscErrorCode VecHIPXGetDeviceReadWrite(Vec v, PetscScalar **d, void **tmp);
PETSC_INTERN PetscErrorCode VecHIPXRestoreDeviceWrite(Vec v, PetscScalar **d, void **tmp);

/* Reduction cache (vechipx.c).  A kernel that WRITES a vector can form, in the same pass, the sums the Krylov method asks for next
   (MatMult: x . y of cg.c:258; VecPointwiseMult = PCApply_Jacobi: z . z and z . r of cg.c:309,344).  The sums are kept keyed on the two vectors'
   object ids; VecDot / VecTDot / VecNorm(NORM_2) on exactly those vectors return them without another pass -- as the interface layer's own norm
   cache does (rvector.c:211,232).  An entry dies the moment either vector is handed out for writing (every write access to a hipx vector goes
   through the accessors of vechipx.c), so a hit is always the value the separate kernel would have computed on the same data: bit for bit in the
   exact reduction mode, to rounding (another association of the partial sums) in the fast mode.  -hipx_reduction_cache 0 turns it off. */
#define HIPX_RC_MATMULT 0 /* entry 0: a = x, b = y of y = A x; v[0] = x . y                  */
#define HIPX_RC_PWMULT  1 /* entry 1: a = w, b = x of w = x .* y; v[0] = w . w, v[1] = w . x */
PETSC_INTERN PetscBool      VecHIPXRedCacheWanted(int kind);
PETSC_INTERN int            VecHIPXRedCacheSlot(int kind);
PETSC_INTERN PetscErrorCode VecHIPXRedCachePut(int kind, Vec a, Vec b);
PETSC_INTERN void           VecHIPXRedCacheInvalidate(Vec v);

/* Lazy fusion (vechipx.c).  VecAXPY / VecAYPX on device-resident hipx vectors are RECORDED, not run: KSPSolve_CG's "x += a p; r -= a w; z = B r" and
   "p = z + b p; w = A p" (cg.c:305-307, 248-256) then run as the two fused kernels of cghipx -- the consumers (VecPointwiseMult = PCApply_Jacobi, MatMult)
   recognise the recorded operations; ANY other access to a vector a recorded operation reads or writes (every accessor of vechipx.c, host or device)
   first runs everything recorded, in order.  -hipx_lazy_fusion 0 turns it off. */
PETSC_INTERN PetscErrorCode VecHIPXLazySync(Vec v);   /* run the recorded operations if v takes part in one */
PETSC_INTERN PetscErrorCode VecHIPXLazyFlush(void);   /* run them all */
PETSC_INTERN PetscErrorCode VecHIPXLazyTryCGProduct(hipxMat dA, Vec xx, Vec yy, PetscBool *done); /* MatMult(A, P, W) with "x += a p; p = z + b p" recorded */

PETSC_INTERN PetscErrorCode VecCreate_SeqHIPX(Vec);
PETSC_INTERN PetscErrorCode VecCreate_MPIHIPX(Vec);
PETSC_INTERN PetscErrorCode VecCreate_HIPX(Vec);
PETSC_INTERN PetscErrorCode MatCreate_SeqAIJHIPX(Mat);
PETSC_INTERN PetscErrorCode MatCreate_MPIAIJHIPX(Mat);
PETSC_INTERN PetscErrorCode PCCreate_JacobiHIPX(PC);
PETSC_INTERN PetscErrorCode PCCreate_PBJacobiHIPX(PC); /* "pbjacobihipx": PCPBJACOBI with the apply on the device */
PETSC_INTERN PetscErrorCode KSPCreate_CGHIPX(KSP); /* "cghipx": KSPCG with the fused device kernels on the hot-path configuration */
PETSC_INTERN PetscErrorCode KSPCreate_GROPPCGHIPX(KSP); /* "groppcghipx": KSPGROPPCG on two fused passes + one product per iteration, launch-ahead (round 6) */
PETSC_INTERN PetscErrorCode KSPCreate_PIPECGHIPX(KSP); /* "pipecghipx": KSPPIPECG on one fused update kernel + one product per iteration, launch-ahead (round 6) */
PETSC_INTERN PetscErrorCode KSPCreate_ChebyshevHIPX(KSP); /* "chebyshevhipx": KSPCHEBYSHEV, first kind without norms on one fused kernel per iteration */
PETSC_INTERN PetscErrorCode MatSeqAIJHIPXGetDeviceMat(Mat A, hipxMat *dA); /* uploads / refreshes the device CSR */
PETSC_INTERN PetscErrorCode MatSeqAIJHIPXSetUpDevice(Mat A); /* upload + hipxMatSetUp now (set-up phases call it; -mat_hipx_setup_at_assembly 0 turns it off) */
PETSC_INTERN PetscBool      MatIsSeqAIJHIPX(Mat A);
PETSC_INTERN PetscErrorCode MatSeqAIJHIPXSetValuesCOO_Private(Mat A, hipxCOO coo, const PetscScalar v[], PetscCount n, InsertMode imode);
PetscErrorCode MatSeqAIJHIPXAddValuesCOOIndexed_Private(Mat, hipxCOO, const PetscScalar *);
PETSC_INTERN PetscErrorCode MatMPIAIJHIPXGetDevice(Mat A, hipxMat *dA, hipxMat *dB, hipxHalo *halo, Vec *lvec); /* halo == NULL: no device exchange */
/* commhipx.c: transport bring-up of a ghost-exchange plan (collective; RCCL -> IPC -> none) */
PETSC_INTERN PetscErrorCode HipxHaloBringUp(MPI_Comm comm, PetscObject obj, hipxHalo *halo, const char *want, PetscInt *transport);
PETSC_INTERN PetscBool      HipxCommIsUp(void);
/* sfhipx.c: PetscSF type "hipx" */
#define PETSCSFHIPX "hipx"
PETSC_INTERN PetscErrorCode PetscSFCreate_HIPX(PetscSF);
PETSC_INTERN PetscBool      hipx_vec_memtype_ops; /* -vec_hipx_memtype: VecGetArray*AndMemType hand out device pointers (every PetscSF that sees these vectors must be of type hipx) */
