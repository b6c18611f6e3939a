"""Readers for the two file formats BASELINE config 4 (SuiteSparse Janna/Flan_1565) comes in, so that a supplied file is one flag
away (`bench.py --matrix-file F`): MatrixMarket coordinate files (what SuiteSparse ships; the reference reads them with
src/mat/tests/mmloader.c: MatCreateFromMTX) and PETSc's binary Mat format (MatLoad_SeqAIJ_Binary, src/mat/impls/aij/seq/aij.c:
4969-5040; MatView_SeqAIJ_Binary writes it).  Both return CSR (ai, aj, aa) with ascending columns per row, int32 indices (int64
row offsets beyond 2^31 nonzeros), float64 values.  Host-side input code: nothing here is on the measured path."""
import gzip
import struct

import numpy as np

MAT_FILE_CLASSID = 1211216  # include/petscmat.h / petsc/private/matimpl.h


def _coo_to_csr(n_rows, rows, cols, vals):
    """Sort by (row, column), sum duplicates left to right in file order (stable sort), build the offsets."""
    order = np.lexsort((cols, rows))  # stable: equal (row, column) pairs keep their file order
    rows, cols, vals = rows[order], cols[order], vals[order]
    if len(rows):
        first = np.ones(len(rows), bool)
        first[1:] = (rows[1:] != rows[:-1]) | (cols[1:] != cols[:-1])
        if not first.all():
            start = np.flatnonzero(first)
            vals = np.add.reduceat(vals, start)
            rows, cols = rows[start], cols[start]
    nnz = len(rows)
    ai = np.zeros(n_rows + 1, np.int64)
    np.cumsum(np.bincount(rows, minlength=n_rows), out=ai[1:])
    if nnz < 2 ** 31 - 8:
        ai = ai.astype(np.int32)
    return ai, np.ascontiguousarray(cols, dtype=np.int32), np.ascontiguousarray(vals, dtype=np.float64)


def read_matrix_market(path):
    """MatrixMarket `matrix coordinate real|integer|pattern general|symmetric|skew-symmetric` (mmloader.c:18-118: symmetric
    files hold the lower triangle, the upper one is mirrored; `pattern` entries get the value 1)."""
    op = gzip.open if str(path).endswith(".gz") else open
    with op(path, "rt") as f:
        head = f.readline().split()
        if len(head) < 5 or head[0] != "%%MatrixMarket" or head[1].lower() != "matrix" or head[2].lower() != "coordinate":
            raise ValueError("not a MatrixMarket coordinate file: %r" % " ".join(head))
        field, symm = head[3].lower(), head[4].lower()
        if field not in ("real", "integer", "pattern", "double") or symm not in ("general", "symmetric", "skew-symmetric"):
            raise ValueError("unsupported MatrixMarket kind: %s %s" % (field, symm))
        line = f.readline()
        while line.startswith("%") or not line.strip():
            line = f.readline()
        m, n, nz = (int(t) for t in line.split()[:3])
        data = np.loadtxt(f, ndmin=2, dtype=np.float64) if nz else np.zeros((0, 3))
    if data.shape[0] != nz:
        raise ValueError("MatrixMarket file declares %d entries, holds %d" % (nz, data.shape[0]))
    rows = data[:, 0].astype(np.int64) - 1
    cols = data[:, 1].astype(np.int64) - 1
    vals = np.ones(nz) if field == "pattern" else data[:, 2].copy()
    if symm != "general":
        off = rows != cols
        sign = -1.0 if symm == "skew-symmetric" else 1.0
        rows, cols, vals = np.concatenate([rows, cols[off]]), np.concatenate([cols, rows[off]]), np.concatenate([vals, sign * vals[off]])
    if m != n:
        raise ValueError("square matrices only on this path (%d x %d)" % (m, n))
    return _coo_to_csr(m, rows, cols, vals)


def read_petsc_binary(path):
    """PETSc binary Mat (big-endian): [MAT_FILE_CLASSID, M, N, nz] int32, M row lengths, nz column indices, nz float64 values
    (aij.c:4985-5031; files written with 32-bit PetscInt)."""
    with open(path, "rb") as f:
        cid, m, n, nz = struct.unpack(">4i", f.read(16))
        if cid != MAT_FILE_CLASSID:
            raise ValueError("not a PETSc binary Mat file (class id %d)" % cid)
        if nz < 0:
            raise ValueError("dense / special-format PETSc binary matrices are not read here (nz = %d)" % nz)
        lens = np.frombuffer(f.read(4 * m), dtype=">i4").astype(np.int64)
        aj = np.frombuffer(f.read(4 * nz), dtype=">i4").astype(np.int32)
        aa = np.frombuffer(f.read(8 * nz), dtype=">f8").astype(np.float64)
    if lens.sum() != nz or len(aj) != nz or len(aa) != nz:
        raise ValueError("truncated PETSc binary Mat file")
    if m != n:
        raise ValueError("square matrices only on this path (%d x %d)" % (m, n))
    ai = np.zeros(m + 1, np.int64)
    np.cumsum(lens, out=ai[1:])
    rows = np.repeat(np.arange(m, dtype=np.int64), lens)
    return _coo_to_csr(m, rows, aj.astype(np.int64), aa)  # (sorted per row already when PETSc wrote it; duplicates cannot occur)


def write_petsc_binary(path, ai, aj, aa, n_cols=None):
    m = len(ai) - 1
    with open(path, "wb") as f:
        f.write(struct.pack(">4i", MAT_FILE_CLASSID, m, n_cols if n_cols is not None else m, int(ai[-1])))
        f.write(np.diff(np.asarray(ai, np.int64)).astype(">i4").tobytes())
        f.write(np.asarray(aj).astype(">i4").tobytes())
        f.write(np.asarray(aa).astype(">f8").tobytes())


def read_matrix(path):
    """By content: MatrixMarket text (optionally gzip-ed) or PETSc binary."""
    p = str(path)
    if p.endswith(".gz"):
        return read_matrix_market(p)
    with open(p, "rb") as f:
        magic = f.read(14)
    return read_matrix_market(p) if magic.startswith(b"%%MatrixMarket") else read_petsc_binary(p)
