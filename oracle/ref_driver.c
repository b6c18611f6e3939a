/*
 * ref_driver.c -- thin PETSc driver used as (a) the strongest oracle: the REFERENCE's own KSPSolve / MatMult on the
 * BASELINE operators, on the CPU types (-mat_type aij -vec_type standard), (b) the end-to-end check of the drop-in:
 * the same binary with -dll_prepend libpetschipx.so -mat_type aijhipx -vec_type hipx, (c) the CPU baseline
 * ("kind": "reference") of bench.py.  Test infrastructure; linked against oracle/_ref/lib/libpetsc.so.
 *
 *   ref_driver -n 64 -stencil 7|27|5 [-m rows for 5-pt] [-matmult_its K] [-ksp_type cg -pc_type jacobi -ksp_rtol ...]
 * Operators: 5-pt = ex2.c:70-94, 7-pt = its 3-D analogue, 27-pt = bench_kspsolve.c:115-303.
 * b = A*1, x0 = 0.  Prints full-precision residual history (-history), iteration count, error norm, KSPSolve seconds.
 */
#include <petscksp.h>
#include <petsctime.h>
#include <../src/mat/impls/aij/mpi/mpiaij.h> /* -dump_split: the reference's own Mat_MPIAIJ pieces (garray, A, B, Mvctx) */
#include <petscsf.h>

/* -dump_split: what MatAssemblyEnd_MPIAIJ / MatSetUpMultiply_MPIAIJ (mmaij.c:8-125) left on every rank, rank by rank:
     split <rank> <rstart> <rend> <nghost>
     garray <rank> <k> <global column>
     ad <rank> <local row> <local column> <value>          diagonal block
     bo <rank> <local row> <ghost index> <value>           off-diagonal block, columns compacted to garray order */
static PetscErrorCode DumpSplit(Mat A)
{
  PetscBool   ismpi;
  PetscMPIInt rank;
  PetscInt    rs, re;

  PetscFunctionBeginUser;
  PetscCall(PetscObjectBaseTypeCompare((PetscObject)A, MATMPIAIJ, &ismpi));
  PetscCheck(ismpi, PETSC_COMM_WORLD, PETSC_ERR_SUP, "-dump_split needs an MPIAIJ matrix (np > 1)");
  PetscCallMPI(MPI_Comm_rank(PETSC_COMM_WORLD, &rank));
  PetscCall(MatGetOwnershipRange(A, &rs, &re));
  {
    Mat_MPIAIJ *a  = (Mat_MPIAIJ *)A->data;
    Mat_SeqAIJ *ad = (Mat_SeqAIJ *)a->A->data, *bo = (Mat_SeqAIJ *)a->B->data;
    PetscInt    ng = a->B->cmap->n;
    PetscCall(PetscSynchronizedPrintf(PETSC_COMM_WORLD, "split %d %" PetscInt_FMT " %" PetscInt_FMT " %" PetscInt_FMT "\n", (int)rank, rs, re, ng));
    for (PetscInt k = 0; k < ng; k++) PetscCall(PetscSynchronizedPrintf(PETSC_COMM_WORLD, "garray %d %" PetscInt_FMT " %" PetscInt_FMT "\n", (int)rank, k, a->garray[k]));
    for (PetscInt r = 0; r < re - rs; r++) {
      for (PetscInt k = ad->i[r]; k < ad->i[r + 1]; k++) PetscCall(PetscSynchronizedPrintf(PETSC_COMM_WORLD, "ad %d %" PetscInt_FMT " %" PetscInt_FMT " %.17g\n", (int)rank, r, ad->j[k], (double)ad->a[k]));
      for (PetscInt k = bo->i[r]; k < bo->i[r + 1]; k++) PetscCall(PetscSynchronizedPrintf(PETSC_COMM_WORLD, "bo %d %" PetscInt_FMT " %" PetscInt_FMT " %.17g\n", (int)rank, r, bo->j[k], (double)bo->a[k]));
    }
    { /* the ghost exchange the reference set up (a->Mvctx is a PetscSF: roots = owned entries of x, leaves = lvec):
           recv <rank> <from> <leaf = index into lvec> <root = local index on the owner>
           send <rank> <to> <position in the message> <local index of the owned entry> */
      PetscMPIInt        nr, ni;
      const PetscMPIInt *ranks, *iranks;
      const PetscInt    *roff, *rmine, *rremote, *ioff, *iroot;
      PetscCall(PetscSFSetUp(a->Mvctx));
      PetscCall(PetscSFGetRootRanks(a->Mvctx, &nr, &ranks, &roff, &rmine, &rremote));
      PetscCall(PetscSFGetLeafRanks(a->Mvctx, &ni, &iranks, &ioff, &iroot));
      for (PetscMPIInt k = 0; k < nr; k++)
        for (PetscInt j = roff[k]; j < roff[k + 1]; j++)
          PetscCall(PetscSynchronizedPrintf(PETSC_COMM_WORLD, "recv %d %d %" PetscInt_FMT " %" PetscInt_FMT "\n", (int)rank, (int)ranks[k], rmine ? rmine[j] : j, rremote[j]));
      for (PetscMPIInt k = 0; k < ni; k++)
        for (PetscInt j = ioff[k]; j < ioff[k + 1]; j++)
          PetscCall(PetscSynchronizedPrintf(PETSC_COMM_WORLD, "send %d %d %" PetscInt_FMT " %" PetscInt_FMT "\n", (int)rank, (int)iranks[k], j - ioff[k], iroot[j]));
    }
    PetscCall(PetscSynchronizedFlush(PETSC_COMM_WORLD, PETSC_STDOUT));
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscInt box_ny = 0, box_nz = 0; /* -ny / -nz: 7-point operator on an n x ny x nz box (BASELINE config 5's per-GPU share: 1024 x 1024 x 128) */

/* row Ii of the operator: columns (ascending) and values; returns the count */
static PetscInt StencilRow(PetscInt stencil, PetscInt m, PetscInt n, PetscInt Ii, PetscInt cols[27], PetscScalar vals[27])
{
  PetscInt nc = 0;
  if (stencil == 5) {
    PetscInt i = Ii / n, j = Ii - i * n;
    if (i > 0) { cols[nc] = Ii - n; vals[nc++] = -1.0; }
    if (j > 0) { cols[nc] = Ii - 1; vals[nc++] = -1.0; }
    cols[nc] = Ii; vals[nc++] = 4.0;
    if (j < n - 1) { cols[nc] = Ii + 1; vals[nc++] = -1.0; }
    if (i < m - 1) { cols[nc] = Ii + n; vals[nc++] = -1.0; }
  } else if (stencil == 7) {
    PetscInt ny = box_ny ? box_ny : n, nz = box_nz ? box_nz : n;
    PetscInt n2 = n * ny, x = Ii % n, y = (Ii / n) % ny, z = Ii / n2;
    if (z > 0) { cols[nc] = Ii - n2; vals[nc++] = -1.0; }
    if (y > 0) { cols[nc] = Ii - n; vals[nc++] = -1.0; }
    if (x > 0) { cols[nc] = Ii - 1; vals[nc++] = -1.0; }
    cols[nc] = Ii; vals[nc++] = 6.0;
    if (x < n - 1) { cols[nc] = Ii + 1; vals[nc++] = -1.0; }
    if (y < ny - 1) { cols[nc] = Ii + n; vals[nc++] = -1.0; }
    if (z < nz - 1) { cols[nc] = Ii + n2; vals[nc++] = -1.0; }
  } else {
    PetscInt    n2 = n * n, n1 = n - 1, x = Ii % n, y = (Ii / n) % n, z = Ii / n2;
    PetscScalar h = 1.0 / (n - 1), v[4];
    v[0] = 44.0 / 13 * h; v[1] = -3.0 / 13 * h; v[2] = -3.0 / 26 * h; v[3] = -1.0 / 13 * h; /* bench_kspsolve.c:122-126 */
    for (int dz = -1; dz <= 1; dz++) {
      if ((dz < 0 && z == 0) || (dz > 0 && z == n1)) continue;
      for (int dy = -1; dy <= 1; dy++) {
        if ((dy < 0 && y == 0) || (dy > 0 && y == n1)) continue;
        for (int dx = -1; dx <= 1; dx++) {
          if ((dx < 0 && x == 0) || (dx > 0 && x == n1)) continue;
          cols[nc] = Ii + dx + dy * n + dz * n2;
          vals[nc++] = v[(dx != 0) + (dy != 0) + (dz != 0)];
        }
      }
    }
  }
  return nc;
}

static PetscErrorCode Assemble(Mat A, PetscInt stencil, PetscInt m, PetscInt n, PetscInt Istart, PetscInt Iend)
{
  PetscInt    cols[27];
  PetscScalar vals[27];

  PetscFunctionBeginUser;
  for (PetscInt Ii = Istart; Ii < Iend; Ii++) {
    PetscInt nc = StencilRow(stencil, m, n, Ii, cols, vals);
    PetscCall(MatSetValues(A, 1, &Ii, nc, cols, vals, INSERT_VALUES));
  }
  PetscCall(MatAssemblyBegin(A, MAT_FINAL_ASSEMBLY));
  PetscCall(MatAssemblyEnd(A, MAT_FINAL_ASSEMBLY));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* -coo_assemble 1|2: the same operator through MatSetPreallocationCOO / MatSetValuesCOO (bench_kspsolve.c:301-302's route).
   1: every rank sets its own rows.  2: entries TRAVEL between ranks (MatSetValuesCOO_MPIAIJ's pack / PetscSFReduce / remote add,
   mpiaij.c:6798-6822): the first rows of every rank get half of each value from their owner and half from the rank before it (exact
   halves: the assembled matrix equals the MatSetValues one bit for bit, whatever the order of the two additions).  The values are set twice (ones first, then the operator: INSERT_VALUES must overwrite).
   -coo_device_values: the value array is handed over as the pointer VecGetArrayReadAndMemType gives for a -vec_type vector (device
   memory for the hipx type with -vec_hipx_memtype). */
static PetscErrorCode AssembleCOO(Mat A, PetscInt mode, PetscInt stencil, PetscInt m, PetscInt n, PetscInt Istart, PetscInt Iend)
{
  PetscMPIInt        rank, size;
  const PetscInt    *ranges;
  PetscInt           cols[27], share, nb, nbs = 0, nbe = 0, *ci, *cj;
  PetscScalar        vals[27], *cv, *ones;
  PetscCount         cnt = 0, k = 0;
  PetscBool          devvals = PETSC_FALSE;
  Vec                vv = NULL;
  const PetscScalar *vptr;

  PetscFunctionBeginUser;
  PetscCallMPI(MPI_Comm_rank(PETSC_COMM_WORLD, &rank));
  PetscCallMPI(MPI_Comm_size(PETSC_COMM_WORLD, &size));
  PetscCall(PetscOptionsGetBool(NULL, NULL, "-coo_device_values", &devvals, NULL));
  PetscCall(MatGetOwnershipRanges(A, &ranges));
  share = (mode == 2 && size > 1) ? 7 : 0; /* rows per rank that get a part of their values from the rank before */
  if (share) {
    nb  = (rank + 1) % size;
    nbs = ranges[nb];
    nbe = PetscMin(ranges[nb] + share, ranges[nb + 1]);
  }
  for (PetscInt Ii = Istart; Ii < Iend; Ii++) cnt += StencilRow(stencil, m, n, Ii, cols, vals);
  for (PetscInt Ii = nbs; Ii < nbe; Ii++) cnt += StencilRow(stencil, m, n, Ii, cols, vals);
  PetscCall(PetscMalloc4(cnt, &ci, cnt, &cj, cnt, &cv, cnt, &ones));
  for (PetscInt Ii = nbs; Ii < nbe; Ii++) { /* the travelling entries first: they are not in row order on the sender */
    PetscInt nc = StencilRow(stencil, m, n, Ii, cols, vals);
    for (PetscInt c = 0; c < nc; c++, k++) { ci[k] = Ii; cj[k] = cols[c]; cv[k] = 0.5 * vals[c]; }
  }
  for (PetscInt Ii = Istart; Ii < Iend; Ii++) {
    PetscInt    nc = StencilRow(stencil, m, n, Ii, cols, vals);
    PetscScalar w  = (share && Ii < Istart + share) ? 0.5 : 1.0;
    for (PetscInt c = 0; c < nc; c++, k++) { ci[k] = Ii; cj[k] = cols[c]; cv[k] = w * vals[c]; }
  }
  for (PetscCount q = 0; q < cnt; q++) ones[q] = 1.0;
  PetscCall(MatSetPreallocationCOO(A, cnt, ci, cj));
  PetscCall(MatSetValuesCOO(A, ones, INSERT_VALUES));
  if (devvals) {
    PetscMemType mt;
    PetscScalar *w;
    PetscCall(VecCreate(PETSC_COMM_SELF, &vv));
    PetscCall(VecSetSizes(vv, (PetscInt)cnt, (PetscInt)cnt));
    PetscCall(VecSetFromOptions(vv));
    PetscCall(VecGetArrayWrite(vv, &w));
    PetscCall(PetscArraycpy(w, cv, cnt));
    PetscCall(VecRestoreArrayWrite(vv, &w));
    PetscCall(VecGetArrayReadAndMemType(vv, &vptr, &mt));
    PetscCall(PetscPrintf(PETSC_COMM_WORLD, "coo values memtype %d\n", (int)mt));
    PetscCall(MatSetValuesCOO(A, vptr, INSERT_VALUES));
    PetscCall(VecRestoreArrayReadAndMemType(vv, &vptr));
    PetscCall(VecDestroy(&vv));
  } else PetscCall(MatSetValuesCOO(A, cv, INSERT_VALUES));
  PetscCall(PetscFree4(ci, cj, cv, ones));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* -scatter_test: general VecScatters between two parallel vectors of the type -vec_type selects -- self edges and remote edges,
   non-contiguous and out-of-order leaves, several leaves per root -- in every mode the Krylov path's callers use: forward INSERT,
   forward ADD, reverse ADD (several leaves summed into one root: the MatMultTranspose shape), reverse INSERT on a one-to-one plan.
   Prints every entry with 17 digits, rank by rank: the CPU types and the hipx types (+ -sf_type hipx) must print the same text. */
static PetscErrorCode ScatterTest(PetscInt nloc)
{
  Vec          x, y;
  IS           ix, iy;
  VecScatter   sc, one;
  PetscMPIInt  rank, size;
  PetscInt     N, rs, re, *from, *to, ne;
  PetscScalar *a;

  PetscFunctionBeginUser;
  PetscCallMPI(MPI_Comm_rank(PETSC_COMM_WORLD, &rank));
  PetscCallMPI(MPI_Comm_size(PETSC_COMM_WORLD, &size));
  PetscCall(VecCreate(PETSC_COMM_WORLD, &x));
  PetscCall(VecSetSizes(x, nloc + rank, PETSC_DECIDE)); /* uneven ownership */
  PetscCall(VecSetFromOptions(x));
  PetscCall(VecGetSize(x, &N));
  PetscCall(VecGetOwnershipRange(x, &rs, &re));
  PetscCall(VecCreate(PETSC_COMM_WORLD, &y));
  PetscCall(VecSetSizes(y, 2 * (re - rs) + 3, PETSC_DECIDE));
  PetscCall(VecSetFromOptions(y));
  /* every rank lists 2 (re - rs) edges: entry g of x goes to y positions 2 (g') and 2 (g') + 1 of a PERMUTED owner: from = a stride-7
     walk over all of x (lands on every rank incl. myself), to = my own y range backwards with a hole pattern */
  ne = 2 * (re - rs);
  PetscCall(PetscMalloc2(ne, &from, ne, &to));
  {
    PetscInt ys, ye;
    PetscCall(VecGetOwnershipRange(y, &ys, &ye));
    for (PetscInt k = 0; k < ne; k++) {
      from[k] = (7 * (rs + k / 2) + 3 * (k % 2) * (rank + 1)) % N; /* roots: some referenced twice, from every rank */
      to[k]   = ye - 1 - k - (k >= ne / 2 ? 3 : 0);                /* leaves: my y range backwards, a hole of 3 in the middle */
    }
  }
  PetscCall(ISCreateGeneral(PETSC_COMM_WORLD, ne, from, PETSC_COPY_VALUES, &ix));
  PetscCall(ISCreateGeneral(PETSC_COMM_WORLD, ne, to, PETSC_COPY_VALUES, &iy));
  PetscCall(VecScatterCreate(x, ix, y, iy, &sc));
  PetscCall(ISDestroy(&ix));
  PetscCall(ISDestroy(&iy));
  /* a one-to-one plan for the INSERT reverse leg: y position (global) p <- x entry (p * 5) mod N restricted to distinct roots */
  {
    PetscInt cnt = 0;
    for (PetscInt g = rs; g < re; g++) {
      from[cnt] = (g + N / 2 + 1) % N; /* a rotation of x: one-to-one, mostly remote */
      to[cnt]   = 2 * g;               /* into the even positions of y (global) */
      cnt++;
    }
    {
      PetscInt M;
      PetscCall(VecGetSize(y, &M));
      for (PetscInt k = 0; k < cnt; k++) to[k] = to[k] % M;
    }
    PetscCall(ISCreateGeneral(PETSC_COMM_WORLD, cnt, from, PETSC_COPY_VALUES, &ix));
    PetscCall(ISCreateGeneral(PETSC_COMM_WORLD, cnt, to, PETSC_COPY_VALUES, &iy));
    PetscCall(VecScatterCreate(x, ix, y, iy, &one));
    PetscCall(ISDestroy(&ix));
    PetscCall(ISDestroy(&iy));
  }
  PetscCall(PetscFree2(from, to));
  PetscCall(VecGetArrayWrite(x, &a));
  for (PetscInt g = rs; g < re; g++) a[g - rs] = 1.0 + (PetscReal)(g % 17) / 17.0 + 1e-3 * g;
  PetscCall(VecRestoreArrayWrite(x, &a));
  PetscCall(VecSet(y, -1.0));
  PetscCall(VecScale(x, 1.0)); /* touch x through a Vec op: with device vector types the current copy now lives on the device */
#define DUMP(tag, v) \
  do { \
    const PetscScalar *va; \
    PetscInt           vs, ve; \
    PetscCall(VecGetOwnershipRange(v, &vs, &ve)); \
    PetscCall(VecGetArrayRead(v, &va)); \
    for (PetscInt i = vs; i < ve; i++) PetscCall(PetscSynchronizedPrintf(PETSC_COMM_WORLD, "%s %" PetscInt_FMT " %.17g\n", tag, i, (double)va[i - vs])); \
    PetscCall(PetscSynchronizedFlush(PETSC_COMM_WORLD, PETSC_STDOUT)); \
    PetscCall(VecRestoreArrayRead(v, &va)); \
  } while (0)
  PetscCall(VecScatterBegin(sc, x, y, INSERT_VALUES, SCATTER_FORWARD));
  PetscCall(VecScatterEnd(sc, x, y, INSERT_VALUES, SCATTER_FORWARD));
  DUMP("fwd_insert", y);
  PetscCall(VecScale(y, 0.5));
  PetscCall(VecScatterBegin(sc, x, y, ADD_VALUES, SCATTER_FORWARD));
  PetscCall(VecScatterEnd(sc, x, y, ADD_VALUES, SCATTER_FORWARD));
  DUMP("fwd_add", y);
  PetscCall(VecScale(y, 1.25));
  PetscCall(VecScatterBegin(sc, y, x, ADD_VALUES, SCATTER_REVERSE));
  PetscCall(VecScatterEnd(sc, y, x, ADD_VALUES, SCATTER_REVERSE));
  DUMP("rev_add", x);
  PetscCall(VecScale(x, 0.75));
  PetscCall(VecScatterBegin(one, x, y, INSERT_VALUES, SCATTER_FORWARD));
  PetscCall(VecScatterEnd(one, x, y, INSERT_VALUES, SCATTER_FORWARD));
  DUMP("one_fwd", y);
  PetscCall(VecScale(y, 3.0));
  PetscCall(VecScatterBegin(one, y, x, INSERT_VALUES, SCATTER_REVERSE));
  PetscCall(VecScatterEnd(one, y, x, INSERT_VALUES, SCATTER_REVERSE));
  DUMP("one_rev", x);
  {
    PetscSFType t;
    PetscCall(PetscSFGetType(sc, &t));
    PetscCall(PetscPrintf(PETSC_COMM_WORLD, "scatter type %s\n", t));
  }
#undef DUMP
  PetscCall(VecScatterDestroy(&sc));
  PetscCall(VecScatterDestroy(&one));
  PetscCall(VecDestroy(&x));
  PetscCall(VecDestroy(&y));
  PetscFunctionReturn(PETSC_SUCCESS);
}

int main(int argc, char **argv)
{
  Mat         A;
  Vec         x, b, u;
  KSP         ksp;
  PetscInt    n = 16, m = 0, stencil = 7, N, Istart, Iend, its, mm_its = 0, nhist = 0;
  PetscReal   norm, *hist = NULL;
  PetscBool   history = PETSC_FALSE, dump_y = PETSC_FALSE;
  PetscLogDouble t0, t1;
  KSPConvergedReason reason;

  PetscFunctionBeginUser;
  PetscCall(PetscInitialize(&argc, &argv, NULL, NULL));
  {
    PetscInt stest = 0;
    PetscCall(PetscOptionsGetInt(NULL, NULL, "-scatter_test", &stest, NULL));
    if (stest > 0) {
      PetscCall(ScatterTest(stest));
      PetscCall(PetscFinalize());
      return 0;
    }
  }
  PetscCall(PetscOptionsGetInt(NULL, NULL, "-n", &n, NULL));
  PetscCall(PetscOptionsGetInt(NULL, NULL, "-stencil", &stencil, NULL));
  m = n;
  PetscCall(PetscOptionsGetInt(NULL, NULL, "-m", &m, NULL));
  PetscCall(PetscOptionsGetInt(NULL, NULL, "-matmult_its", &mm_its, NULL));
  PetscCall(PetscOptionsGetBool(NULL, NULL, "-history", &history, NULL));
  PetscCall(PetscOptionsGetBool(NULL, NULL, "-dump_y", &dump_y, NULL));
  PetscCall(PetscOptionsGetInt(NULL, NULL, "-ny", &box_ny, NULL));
  PetscCall(PetscOptionsGetInt(NULL, NULL, "-nz", &box_nz, NULL));
  PetscCheck((!box_ny && !box_nz) || stencil == 7, PETSC_COMM_WORLD, PETSC_ERR_SUP, "-ny / -nz: 7-point operator only");
  N = (stencil == 5) ? m * n : (stencil == 7 ? n * (box_ny ? box_ny : n) * (box_nz ? box_nz : n) : n * n * n);

  {
    char      file[PETSC_MAX_PATH_LEN];
    PetscBool fromfile = PETSC_FALSE; /* -f <PETSc binary Mat>: the operator comes from a file (MatLoad: BASELINE config 4's route for a SuiteSparse matrix) */
    PetscCall(PetscOptionsGetString(NULL, NULL, "-f", file, sizeof(file), &fromfile));
    if (fromfile) {
      PetscViewer viewer;
      PetscCall(PetscViewerBinaryOpen(PETSC_COMM_WORLD, file, FILE_MODE_READ, &viewer));
      PetscCall(MatCreate(PETSC_COMM_WORLD, &A));
      PetscCall(MatSetFromOptions(A));
      PetscCall(MatLoad(A, viewer));
      PetscCall(PetscViewerDestroy(&viewer));
      PetscCall(MatGetSize(A, &N, NULL));
      PetscCall(MatGetOwnershipRange(A, &Istart, &Iend));
      goto assembled;
    }
  }
  PetscCall(MatCreate(PETSC_COMM_WORLD, &A));
  PetscCall(MatSetSizes(A, PETSC_DECIDE, PETSC_DECIDE, N, N));
  PetscCall(MatSetFromOptions(A));
  {
    PetscInt bs = 1; /* -mat_block_size b: the operator declared with point blocks of b rows (PCPBJACOBI inverts the b x b diagonal blocks) */
    PetscCall(PetscOptionsGetInt(NULL, NULL, "-mat_block_size", &bs, NULL));
    if (bs > 1) PetscCall(MatSetBlockSize(A, bs));
  }
  PetscCall(MatSeqAIJSetPreallocation(A, stencil, NULL));
  PetscCall(MatMPIAIJSetPreallocation(A, stencil, NULL, stencil, NULL));
  PetscCall(MatGetOwnershipRange(A, &Istart, &Iend));
  {
    PetscInt coo = 0;
    PetscCall(PetscOptionsGetInt(NULL, NULL, "-coo_assemble", &coo, NULL));
    if (coo) {
      PetscCall(MatSetUp(A));
      PetscCall(MatGetOwnershipRange(A, &Istart, &Iend));
      PetscCall(AssembleCOO(A, coo, stencil, m, n, Istart, Iend));
    } else PetscCall(Assemble(A, stencil, m, n, Istart, Iend));
  }
assembled:
  {
    PetscBool dump_split = PETSC_FALSE;
    PetscCall(PetscOptionsGetBool(NULL, NULL, "-dump_split", &dump_split, NULL));
    if (dump_split) {
      PetscCall(DumpSplit(A));
      PetscCall(MatDestroy(&A));
      PetscCall(PetscFinalize());
      return 0;
    }
  }
  {
    PetscBool dup = PETSC_FALSE; /* -dup_mat: run everything on MatDuplicate(A) (exercises the duplicate op of Mat subclasses) */
    PetscCall(PetscOptionsGetBool(NULL, NULL, "-dup_mat", &dup, NULL));
    if (dup) {
      Mat C;
      PetscCall(MatDuplicate(A, MAT_COPY_VALUES, &C));
      PetscCall(MatDestroy(&A));
      A = C;
    }
  }
  {
    PetscBool matops = PETSC_FALSE; /* -mat_ops: A <- D_l (1.25 A) D_r with non-constant diagonal scalings, then one entry set from the host
                                       (MatScale, MatDiagonalScale, MatSetValues + assembly in a row: value-only ops of Mat subclasses) */
    PetscCall(PetscOptionsGetBool(NULL, NULL, "-mat_ops", &matops, NULL));
    if (matops) {
      Vec          l, r;
      PetscScalar *la, *ra;
      PetscInt     rs, re;
      PetscCall(MatCreateVecs(A, &r, &l));
      PetscCall(VecGetOwnershipRange(l, &rs, &re));
      PetscCall(VecGetArrayWrite(l, &la));
      PetscCall(VecGetArrayWrite(r, &ra));
      for (PetscInt i = rs; i < re; i++) {
        la[i - rs] = 1.0 + (PetscReal)(i % 7) / 7.0;
        ra[i - rs] = 1.0 + (PetscReal)(i % 5) / 5.0;
      }
      PetscCall(VecRestoreArrayWrite(l, &la));
      PetscCall(VecRestoreArrayWrite(r, &ra));
      PetscCall(MatScale(A, 1.25));
      PetscCall(MatDiagonalScale(A, l, r));
      PetscCall(MatSetValue(A, rs, rs, 7.5, ADD_VALUES)); /* host-side update right after the device-side ones */
      PetscCall(MatAssemblyBegin(A, MAT_FINAL_ASSEMBLY));
      PetscCall(MatAssemblyEnd(A, MAT_FINAL_ASSEMBLY));
      PetscCall(MatDiagonalScale(A, r, NULL));
      { /* -mat_ops_block_edit: right after a device-side value op that the MPI type forwards to its blocks WITHOUT an object-state
           increase (MatDiagonalScale_MPIAIJ mpiaij.c:1985-1993), edit one value of the diagonal block in place on the host
           (MatSeqAIJGetArray / RestoreArray: exactly one state increase): a subclass that predicts states uploads nothing here */
        PetscBool edit = PETSC_FALSE;
        PetscCall(PetscOptionsGetBool(NULL, NULL, "-mat_ops_block_edit", &edit, NULL));
        if (edit) {
          Mat          Ad = A;
          PetscScalar *arr;
          PetscMPIInt  sz;
          PetscCallMPI(MPI_Comm_size(PETSC_COMM_WORLD, &sz));
          if (sz > 1) PetscCall(MatMPIAIJGetSeqAIJ(A, &Ad, NULL, NULL));
          PetscCall(MatSeqAIJGetArray(Ad, &arr));
          arr[0] *= 1.5;
          arr[3] += 0.125;
          PetscCall(MatSeqAIJRestoreArray(Ad, &arr));
        }
      }
      PetscCall(VecDestroy(&l));
      PetscCall(VecDestroy(&r));
    }
  }
  {
    PetscBool axpy = PETSC_FALSE; /* -mat_axpy: A <- A + 0.37 C with C = a non-uniformly scaled copy of A (same nonzero pattern): MatAXPY_SeqAIJ aij.c:2926 */
    PetscCall(PetscOptionsGetBool(NULL, NULL, "-mat_axpy", &axpy, NULL));
    if (axpy) {
      Mat          C;
      Vec          l;
      PetscScalar *la;
      PetscInt     rs, re;
      PetscCall(MatDuplicate(A, MAT_COPY_VALUES, &C));
      PetscCall(MatCreateVecs(C, NULL, &l));
      PetscCall(VecGetOwnershipRange(l, &rs, &re));
      PetscCall(VecGetArrayWrite(l, &la));
      for (PetscInt i = rs; i < re; i++) la[i - rs] = 0.3 + (PetscReal)(i % 11) / 7.0;
      PetscCall(VecRestoreArrayWrite(l, &la));
      PetscCall(MatDiagonalScale(C, l, NULL));
      PetscCall(MatAXPY(A, 0.37, C, SAME_NONZERO_PATTERN));
      PetscCall(VecDestroy(&l));
      PetscCall(MatDestroy(&C));
    }
  }
  PetscCall(MatCreateVecs(A, &u, &b));
  PetscCall(VecSetFromOptions(u));
  PetscCall(VecDuplicate(u, &x));
  PetscCall(VecDestroy(&b));
  PetscCall(VecDuplicate(u, &b));
  PetscCall(VecSet(u, 1.0));
  PetscCall(MatMult(A, u, b));

  if (mm_its > 0 || dump_y) { /* SpMV leg: x_i = 1 + (i mod 17)/17 */
    PetscScalar *a;
    Vec          y;
    PetscCall(VecDuplicate(u, &y));
    PetscCall(VecGetArrayWrite(x, &a));
    for (PetscInt i = Istart; i < Iend; i++) a[i - Istart] = 1.0 + (PetscReal)(i % 17) / 17.0;
    PetscCall(VecRestoreArrayWrite(x, &a));
    PetscCall(MatMult(A, x, y));
    PetscCall(VecNorm(y, NORM_2, &norm));
    PetscCall(PetscTime(&t0));
    for (PetscInt k = 0; k < mm_its; k++) PetscCall(MatMult(A, x, y));
    PetscCall(VecNorm(y, NORM_INFINITY, &norm)); /* also drains any asynchronous work */
    PetscCall(PetscTime(&t1));
    PetscCall(VecNorm(y, NORM_2, &norm));
    PetscCall(PetscPrintf(PETSC_COMM_WORLD, "MatMult its %" PetscInt_FMT " seconds %.6e ynorm %.17g\n", mm_its, (double)(t1 - t0), (double)norm));
    if (dump_y) {
      const PetscScalar *ya;
      PetscBool          dump_yt = PETSC_FALSE; /* -dump_yt: also yt = A^T x (MatMultTranspose) and yta = y + A^T x (MatMultTransposeAdd), every entry */
      PetscCall(VecGetArrayRead(y, &ya));
      for (PetscInt i = 0; i < Iend - Istart; i++) PetscCall(PetscPrintf(PETSC_COMM_SELF, "y %" PetscInt_FMT " %.17g\n", i + Istart, (double)ya[i]));
      PetscCall(VecRestoreArrayRead(y, &ya));
      PetscCall(PetscOptionsGetBool(NULL, NULL, "-dump_yt", &dump_yt, NULL));
      if (dump_yt) {
        Vec yt, yta;
        PetscCall(VecDuplicate(u, &yt));
        PetscCall(VecDuplicate(u, &yta));
        PetscCall(MatMultTranspose(A, x, yt));
        PetscCall(MatMultTransposeAdd(A, x, y, yta));
        PetscCall(VecGetArrayRead(yt, &ya));
        for (PetscInt i = 0; i < Iend - Istart; i++) PetscCall(PetscPrintf(PETSC_COMM_SELF, "yt %" PetscInt_FMT " %.17g\n", i + Istart, (double)ya[i]));
        PetscCall(VecRestoreArrayRead(yt, &ya));
        PetscCall(VecGetArrayRead(yta, &ya));
        for (PetscInt i = 0; i < Iend - Istart; i++) PetscCall(PetscPrintf(PETSC_COMM_SELF, "yta %" PetscInt_FMT " %.17g\n", i + Istart, (double)ya[i]));
        PetscCall(VecRestoreArrayRead(yta, &ya));
        PetscCall(MatMultTransposeAdd(A, x, yt, yt)); /* in place: yt <- yt + A^T x */
        PetscCall(VecNorm(yt, NORM_2, &norm));
        PetscCall(PetscPrintf(PETSC_COMM_WORLD, "MatMultTransposeAdd in place: norm %.17g\n", (double)norm));
        PetscCall(VecDestroy(&yt));
        PetscCall(VecDestroy(&yta));
      }
    }
    PetscCall(VecDestroy(&y));
  }

  { /* -dump_sor <MatSORType bits>: x = MatSOR(A, b, omega, flag, 0, its, lits) on this operator, every entry with 17 digits -- the relaxation
       routine itself (MatSOR_SeqAIJ or, for a matrix with inodes, MatSOR_SeqAIJ_Inode; aij.c:1852), with -sor_its / -sor_lits /
       -sor_omega; without SOR_ZERO_INITIAL_GUESS (16) the sweep starts from x0_i = 0.5 + (i mod 7) / 7 */
    PetscInt  sflag = -1, sits = 1, slits = 1;
    PetscReal somega = 1.0;
    PetscCall(PetscOptionsGetInt(NULL, NULL, "-dump_sor", &sflag, NULL));
    if (sflag >= 0) {
      Vec                sx;
      PetscScalar       *xa;
      const PetscScalar *ca;
      PetscCall(PetscOptionsGetInt(NULL, NULL, "-sor_its", &sits, NULL));
      PetscCall(PetscOptionsGetInt(NULL, NULL, "-sor_lits", &slits, NULL));
      PetscCall(PetscOptionsGetReal(NULL, NULL, "-sor_omega", &somega, NULL));
      PetscCall(VecDuplicate(b, &sx));
      PetscCall(VecGetArrayWrite(sx, &xa));
      for (PetscInt i = Istart; i < Iend; i++) xa[i - Istart] = 0.5 + (PetscReal)(i % 7) / 7.0;
      PetscCall(VecRestoreArrayWrite(sx, &xa));
      PetscCall(MatSOR(A, b, somega, (MatSORType)sflag, 0.0, sits, slits, sx));
      PetscCall(VecGetArrayRead(sx, &ca));
      for (PetscInt i = 0; i < Iend - Istart; i++) PetscCall(PetscSynchronizedPrintf(PETSC_COMM_WORLD, "sor %" PetscInt_FMT " %.17g\n", i + Istart, (double)ca[i]));
      PetscCall(PetscSynchronizedFlush(PETSC_COMM_WORLD, PETSC_STDOUT));
      PetscCall(VecRestoreArrayRead(sx, &ca));
      PetscCall(VecDestroy(&sx));
    }
  }
  PetscCall(KSPCreate(PETSC_COMM_WORLD, &ksp));
  PetscCall(KSPSetOperators(ksp, A, A));
  PetscCall(KSPSetFromOptions(ksp));
  {
    PetscBool dump_pc = PETSC_FALSE; /* -dump_pc: z = PCApply(b) and z' = PCApplyTranspose(b), every entry with 17 digits */
    PetscCall(PetscOptionsGetBool(NULL, NULL, "-dump_pc", &dump_pc, NULL));
    if (dump_pc) {
      PC                 pc;
      Vec                z;
      const PetscScalar *za;
      PetscCall(KSPSetUp(ksp));
      PetscCall(KSPGetPC(ksp, &pc));
      PetscCall(VecDuplicate(b, &z));
      for (int tr = 0; tr < 2; tr++) {
        if (tr) PetscCall(PCApplyTranspose(pc, b, z));
        else PetscCall(PCApply(pc, b, z));
        PetscCall(VecGetArrayRead(z, &za));
        for (PetscInt i = 0; i < Iend - Istart; i++) PetscCall(PetscSynchronizedPrintf(PETSC_COMM_WORLD, "%s %" PetscInt_FMT " %.17g\n", tr ? "zt" : "z", i + Istart, (double)za[i]));
        PetscCall(PetscSynchronizedFlush(PETSC_COMM_WORLD, PETSC_STDOUT));
        PetscCall(VecRestoreArrayRead(z, &za));
      }
      PetscCall(VecDestroy(&z));
    }
  }
  if (history) {
    PetscCall(PetscMalloc1(100000, &hist));
    PetscCall(KSPSetResidualHistory(ksp, hist, 100000, PETSC_TRUE));
  }
  PetscCall(KSPSetUp(ksp));
  PetscCall(VecSet(x, 0.0));
  PetscCall(PetscTime(&t0));
  PetscCall(KSPSolve(ksp, b, x));
  PetscCall(VecNorm(x, NORM_INFINITY, &norm)); /* (a solver without reductions returns while a device still works: drain it inside the timed region) */
  PetscCall(PetscTime(&t1));
  PetscCall(KSPGetIterationNumber(ksp, &its));
  PetscCall(KSPGetConvergedReason(ksp, &reason));
  if (history) {
    PetscCall(KSPGetResidualHistory(ksp, NULL, &nhist));
    for (PetscInt i = 0; i < nhist; i++) PetscCall(PetscPrintf(PETSC_COMM_WORLD, "hist %" PetscInt_FMT " %.17g\n", i, (double)hist[i]));
  }
  {
    PetscBool dump_x = PETSC_FALSE; /* -dump_x: the solution, every entry with 17 digits (solvers without reductions must reproduce it bit for bit) */
    PetscCall(PetscOptionsGetBool(NULL, NULL, "-dump_x", &dump_x, NULL));
    if (dump_x) {
      const PetscScalar *xa;
      PetscCall(VecGetArrayRead(x, &xa));
      for (PetscInt i = 0; i < Iend - Istart; i++) PetscCall(PetscPrintf(PETSC_COMM_SELF, "x %" PetscInt_FMT " %.17g\n", i + Istart, (double)xa[i]));
      PetscCall(VecRestoreArrayRead(x, &xa));
    }
  }
  PetscCall(VecAXPY(x, -1.0, u));
  PetscCall(VecNorm(x, NORM_2, &norm));
  PetscCall(PetscPrintf(PETSC_COMM_WORLD, "iterations %" PetscInt_FMT " reason %d error %.17g KSPSolve_seconds %.6e\n", its, (int)reason, (double)norm, (double)(t1 - t0)));
  {
    PetscBool resolve = PETSC_FALSE; /* -resolve: the same solve once more from x = 0 -- every format and buffer a first solve builds exists: the time of the iterations alone */
    PetscCall(PetscOptionsGetBool(NULL, NULL, "-resolve", &resolve, NULL));
    if (resolve) {
      PetscCall(VecSet(x, 0.0));
      PetscCall(PetscTime(&t0));
      PetscCall(KSPSolve(ksp, b, x));
      PetscCall(VecNorm(x, NORM_INFINITY, &norm));
      PetscCall(PetscTime(&t1));
      PetscCall(KSPGetIterationNumber(ksp, &its));
      PetscCall(PetscPrintf(PETSC_COMM_WORLD, "second_solve iterations %" PetscInt_FMT " KSPSolve_seconds %.6e\n", its, (double)(t1 - t0)));
    }
  }
  PetscCall(PetscFree(hist));
  PetscCall(KSPDestroy(&ksp));
  PetscCall(VecDestroy(&u));
  PetscCall(VecDestroy(&x));
  PetscCall(VecDestroy(&b));
  PetscCall(MatDestroy(&A));
  PetscCall(PetscFinalize());
  return 0;
}
