"""Synthetic stand-ins for inputs the container does not hold (test + benchmark infrastructure)."""
import numpy as np


def flan_surrogate(n=80, seed=1565):
    """Surrogate for BASELINE config 4 (SuiteSparse Janna/Flan_1565: 1,564,794 rows, ~117 M nonzeros, ~75 per row, irregular;
    not in the container and no network -- SURVEY.md 8(d) asks for a documented substitute).  A hexahedral elasticity mesh like
    Flan_1565's: n^3 nodes x 3 degrees of freedom (n = 80: 1,536,000 rows), every node coupled to its 27 neighbours by a dense
    3x3 block (up to 81 entries per row, ~79 on average, 121 M nonzeros), node numbering shuffled inside windows of 512, ALL
    values distinct (standard normal), so neither a value dictionary nor row templates apply.  Returns CSR (ai, aj, aa)."""
    nn = n ** 3
    N = 3 * nn
    rng = np.random.default_rng(seed)
    perm = np.arange(nn, dtype=np.int64).reshape(-1, 512)
    perm = np.take_along_axis(perm, np.argsort(rng.random(perm.shape), axis=1), axis=1).reshape(-1)
    gx, gy, gz = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    gx, gy, gz = gx.ravel(), gy.ravel(), gz.ravel()
    u_all = gx + n * gy + n * n * gz
    rows_l, cols_l = [], []
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                ok = (gx + dx >= 0) & (gx + dx < n) & (gy + dy >= 0) & (gy + dy < n) & (gz + dz >= 0) & (gz + dz < n)
                u = perm[u_all[ok]]
                v = perm[u_all[ok] + dx + n * dy + n * n * dz]
                for a in range(3):
                    for b in range(3):
                        rows_l.append(3 * u + a)
                        cols_l.append(3 * v + b)
    rows = np.concatenate(rows_l)
    cols = np.concatenate(cols_l)
    del rows_l, cols_l
    order = np.argsort(rows * N + cols, kind="stable")
    rows, cols = rows[order], cols[order].astype(np.int32)
    del order
    lens = np.bincount(rows, minlength=N)
    ai = np.zeros(N + 1, np.int32)
    ai[1:] = np.cumsum(lens)
    aa = rng.standard_normal(int(ai[-1]))
    return ai, cols, aa


def flan_surrogate_spd(n=80, seed=1565):
    """The same pattern as flan_surrogate() with SYMMETRIC POSITIVE DEFINITE values, for BASELINE config 4's solver (KSPCG +
    PCJACOBI; Flan_1565 is SPD): a_ij = a_ji = -(0.05 + u(i, j)) with u a hash of the unordered pair in [0, 1) (all off-diagonal
    values distinct up to hash collisions: no dictionary, no templates), a_ii = 1 + sum_j |a_ij| + u(i, i) (strictly diagonally
    dominant, distinct diagonals: the constant-diagonal Jacobi shortcut does not apply).  Returns CSR (ai, aj, aa)."""
    ai, aj, _ = flan_surrogate(n, seed)
    N = len(ai) - 1
    rows = np.repeat(np.arange(N, dtype=np.int64), np.diff(ai))
    cols = aj.astype(np.int64)
    lo, hi = np.minimum(rows, cols), np.maximum(rows, cols)
    key = (lo * N + hi).astype(np.uint64)
    h = key * np.uint64(0x9E3779B97F4A7C15)
    h ^= h >> np.uint64(29)
    h *= np.uint64(0xBF58476D1CE4E5B9)
    u = (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    aa = -(0.05 + u)
    diag = rows == cols
    off_abs = np.where(diag, 0.0, -aa)
    rowsum = np.bincount(rows, weights=off_abs, minlength=N)
    aa[diag] = 1.0 + rowsum[rows[diag]] + u[diag]
    return ai, aj, aa


def inode_matrix(nnodes=60, seed=7, sizes=(1, 2, 3, 4, 5), degree=4, pivot_blocks=True, long_run=True):
    """A matrix with INODES (runs of consecutive rows sharing one column list, what MatSeqAIJCheckInode finds: inode.c:3920): `nnodes`
    mesh nodes of 1-5 unknowns each (sizes drawn from `sizes`; one run of 7 identical rows exercises the limit of 5), every node
    coupled to itself and to ~`degree` others by dense blocks, all values distinct.  Diagonal blocks are well conditioned; with
    pivot_blocks some of them have their largest entries OFF the diagonal, so the block inverses interchange rows (dgefa3.c:31-46).
    Returns CSR (ai, aj, aa) with sorted columns."""
    rng = np.random.default_rng(seed)
    sz = rng.choice(np.array(sizes), size=nnodes)
    if nnodes > 8 and long_run:
        sz[3] = 7  # 7 identical rows: nodes of 5 + 2
    start = np.concatenate([[0], np.cumsum(sz)])
    N = int(start[-1])
    nbr = [set([u]) for u in range(nnodes)]
    for u in range(nnodes):
        for v in rng.choice(nnodes, size=min(degree, nnodes), replace=False):
            nbr[u].add(int(v))
            nbr[int(v)].add(u)
    ai, aj, aa = [0], [], []
    for u in range(nnodes):
        cols = np.concatenate([np.arange(start[v], start[v + 1]) for v in sorted(nbr[u])])
        for r in range(int(sz[u])):
            row = int(start[u]) + r
            vals = rng.standard_normal(len(cols)) * 0.3
            d = np.searchsorted(cols, row)
            vals[d] = 4.0 + rng.random() + np.abs(vals).sum()
            if pivot_blocks and sz[u] > 1 and u % 3 == 1:  # the largest entry of this row of the diagonal block is NOT the diagonal one
                o = np.searchsorted(cols, int(start[u]) + (r + 1) % int(sz[u]))
                vals[o] = 2.5 * vals[d] * (1 if r % 2 else -1)
            aj.append(cols)
            aa.append(np.round(vals * 1024.0) / 1024.0)  # multiples of 2^-10: A * 1 is exact in ANY summation order (the reference forms
            ai.append(ai[-1] + len(cols))                # b = A * 1 with MatMult_SeqAIJ_Inode, whose order is not MatMult_SeqAIJ's)
    return np.array(ai, np.int32), np.concatenate(aj).astype(np.int32), np.concatenate(aa)
