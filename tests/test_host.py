"""CPU tests of the C host logic (no kernels): driver assembly, MPIAIJ split / garray (bit-exact integer work),
ghost-exchange plan, including a 2-process gloo run of the plan exchange used by bench.py for N > 1."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def host_stencil(ks, kind, n, rs, re, m=None):
    f = {"5pt": lambda *a: ks.HipxAssemble_ex2(m or n, n, *a), "7pt": lambda *a: ks.HipxAssemble_poisson7(n, *a),
         "27pt": lambda *a: ks.HipxAssemble_bench27(n, *a)}[kind]
    nz = f(rs, re, None, None, None)
    ai = np.zeros(re - rs + 1, np.int32)
    aj = np.zeros(nz, np.int32)
    aa = np.zeros(nz)
    f(rs, re, ai.ctypes.data_as(C.c_void_p), aj.ctypes.data_as(C.c_void_p), aa.ctypes.data_as(C.c_void_p))
    return ai, aj, aa


@pytest.mark.parametrize("kind,n,m", [("5pt", 7, 8), ("7pt", 9, None), ("27pt", 8, None)])
def test_driver_assembly_equals_oracle_generators(built, kind, n, m):
    from petsc_amd import _lib
    _, ks = _lib.load()
    N = (m or n) * n if kind == "5pt" else n ** 3
    for rs, re in ((0, N), (N // 3, 2 * N // 3 + 1)):
        a = host_stencil(ks, kind, n, rs, re, m)
        b = orc.stencil(kind, n, rs, re, m=m)
        for u, v in zip(a, b):
            assert np.array_equal(u, v)


@pytest.mark.parametrize("nranks", [2, 3, 5])
def test_mpiaij_split_garray_bit_exact_vs_oracle(built, nranks):
    from petsc_amd import _lib
    _, ks = _lib.load()
    n = 6
    N = n ** 3
    ranges = np.zeros(nranks + 1, np.int32)
    ks.HipxSplitOwnership(N, nranks, ranges.ctypes.data_as(C.c_void_p))
    oranges = np.zeros(nranks + 1, np.int32)
    orc.lib().orc_PetscSplitOwnership(N, nranks, orc.P(oranges))
    assert np.array_equal(ranges, oranges) and ranges[-1] == N
    for r in range(nranks):
        rs, re = int(ranges[r]), int(ranges[r + 1])
        ai, aj, aa = orc.stencil("27pt", n, rs, re)
        m, nz = re - rs, len(aj)
        s = _lib.MPIAIJSplit()
        assert ks.HipxMatSetUpMultiply_MPIAIJ(m, rs, re, orc.P(ai), orc.P(aj), orc.P(aa), C.byref(s)) == 0
        Ai, Aj, Bi, Bj, ga = (np.zeros(k, np.int32) for k in (m + 1, nz + 1, m + 1, nz + 1, nz + 1))
        Aa, Ba = np.zeros(nz + 1), np.zeros(nz + 1)
        ec = orc.lib().orc_MatSetUpMultiply_MPIAIJ(m, rs, re, orc.P(ai), orc.P(aj), orc.P(aa), orc.P(Ai), orc.P(Aj), orc.P(Aa), orc.P(Bi), orc.P(Bj), orc.P(Ba), orc.P(ga))
        assert s.nghost == ec

        def arr(ptr, cnt, dt):
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(dt)), shape=(max(cnt, 1),))[:cnt].copy()
        assert np.array_equal(arr(s.garray, ec, C.c_int32), ga[:ec])
        assert np.array_equal(arr(s.Ai, m + 1, C.c_int32), Ai)
        assert np.array_equal(arr(s.Aj, Ai[m], C.c_int32), Aj[:Ai[m]])
        assert np.array_equal(arr(s.Aa, Ai[m], C.c_double), Aa[:Ai[m]])
        assert np.array_equal(arr(s.Bj, Bi[m], C.c_int32), Bj[:Bi[m]])  # compacted column ids, bit-exact
        assert np.array_equal(arr(s.Ba, Bi[m], C.c_double), Ba[:Bi[m]])
        ci, ridx = np.zeros(m + 1, np.int32), np.zeros(m + 1, np.int32)
        nrc = orc.lib().orc_MatCheckCompressedRow(m, orc.P(Bi), orc.P(ci), orc.P(ridx))
        assert s.nrows_c == nrc
        assert np.array_equal(arr(s.Bi, nrc + 1, C.c_int32), ci[:nrc + 1]) and np.array_equal(arr(s.ridx, nrc, C.c_int32), ridx[:nrc])
        # receive plan: owners contiguous, only neighbours for a slab partition of a 27-point stencil
        nrecv = C.c_int()
        rr, ro = np.zeros(nranks, np.int32), np.zeros(nranks + 1, np.int32)
        assert ks.HipxHaloRecvPlan(ec, s.garray, nranks, ranges.ctypes.data_as(C.c_void_p), C.byref(nrecv), rr.ctypes.data_as(C.c_void_p), ro.ctypes.data_as(C.c_void_p)) == 0
        for k in range(nrecv.value):
            seg = ga[ro[k]:ro[k + 1]]
            assert np.all((seg >= ranges[rr[k]]) & (seg < ranges[rr[k] + 1])) and rr[k] != r
        assert ro[nrecv.value] == ec
        ks.HipxMPIAIJSplitFree(C.byref(s))


@pytest.mark.parametrize("world,kind,n,port", [(2, "27pt", 7, 29631), (4, "7pt", 9, 29641), (4, "27pt", 6, 29651), (8, "27pt", 9, 29661), (8, "7pt", 5, 29671), (3, "27pt", 4, 29681)])
def test_gloo_plan_exchange_and_simulated_matmult(built, tmp_path, world, kind, n, port):
    """world_size 2-8 on CPU (gloo): each rank builds its slab, split and receive plan in C, exchanges the send lists
    exactly as bench.py does for N > 1, and checks (a) send / receive symmetry over ALL ranks (ids and order), (b) the ghost
    values land as lvec[k] <-> garray[k], (c) y = A_d x_local + B_o x_ghost against the oracle's global product.  Uneven splits
    (9^3 rows over 4 or 8 ranks, 64 rows over 3), slabs thinner than a plane (8 ranks on 5^3: a rank talks to more than two
    neighbours), 7- and 27-point stencils.  (The arithmetic here is the oracle's: this covers the plan, not the kernels.)"""
    script = os.path.join(ROOT, "tests", "_gloo_plan_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PYTHONPATH=ROOT + os.pathsep + os.path.join(ROOT, "oracle"))
    procs = [subprocess.Popen([sys.executable, script, str(r), str(world), kind, str(n)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert "PLAN_OK" in o, o
