"""GPU, round 6: the two-kernel CG iteration on a rank WITH an off-diagonal block (hipxMatMultMPICGDirectionDotBegin: MatMult_MPIAIJ mpiaij.c:1047-1061
around the fused direction + product kernel, the boundary rows' share of p . w formed by the off-diagonal kernel).

One process plays one rank of an N-rank row-slab partition alone ("loop-back": petsc_amd/dist.py build_plan(..., loopback=True) -- the slab's real diagonal
and off-diagonal blocks, ghost lists and IPC self-exchange; the neighbours' planes are played by the rank's own).  Held against the SEPARATE kernels of the
same library on the same blocks (hipxCGAypxAxpyR, hipxMatMultMPI, hipxVecDot -- themselves bit-identical to the reference's MatMult_MPIAIJ / VecAYPX / VecAXPY
under mpiexec: tests/test_gpu_plugin_mpi.py, tests/test_gpu_halo.py): p_new, x and w bit for bit, the dot to rounding (default reductions) or bit for bit
(exact reductions); then the whole solver: launch-ahead fused CG on the loop-back rank against the host-synchronised unfused loop."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def hxl(hx):
    """libhipx with a one-rank IPC communicator for the duration of this module (the all-reduce kernel runs as between ranks)"""
    from petsc_amd import _lib
    from petsc_amd import dist as pdist
    pdist.comm_init_loopback()
    yield hx
    _lib.chk(hx.hipxCommFinalize())


def slab(ks, stencil, dims, world, rank):
    from petsc_amd import dist as pdist
    nx, ny, nz = dims
    N = nx * ny * nz
    ranges = pdist.split_ownership(N, world)
    rs, re = int(ranges[rank]), int(ranges[rank + 1])
    if stencil == 7:
        def f(ai, aj, aa):
            return ks.HipxAssemble_poisson7_box(nx, ny, nz, rs, re, ai, None, aj, aa)
    else:
        def f(ai, aj, aa):
            return ks.HipxAssemble_bench27(nx, rs, re, ai, aj, aa)
    nnz = f(None, None, None)
    ai, aj, aa = np.zeros(re - rs + 1, np.int32), np.zeros(nnz, np.int32), np.zeros(nnz)
    f(ai.ctypes.data_as(C.c_void_p), aj.ctypes.data_as(C.c_void_p), aa.ctypes.data_as(C.c_void_p))
    plan = pdist.build_plan(ai, aj, aa, ranges, rank, loopback=True)
    M, keep = pdist.create_device_mat(plan, world, loopback=True)
    from petsc_amd import _lib
    hx, _ = _lib.load()
    _lib.chk(hx.hipxMatSetSpMVVariant(M.A, 30))  # the march form also on these small slabs (auto keeps it for grids that fill the chip: >= 192 workgroups)
    return M, keep, plan


def free(hx, keep):
    from petsc_amd import _lib
    _lib.chk(hx.hipxDeviceSynchronize())
    A, B, halo, lvec = keep
    lvec.free()
    _lib.chk(hx.hipxHaloDestroy(C.byref(halo)))
    _lib.mat_destroy(B)
    _lib.mat_destroy(A)


# (the 27-entry template takes the CG-prologue kernel on planes of >= 32768 rows -- 2048-row tiles; smaller planes keep the separate kernels: *fused = 0)
CASES = [(7, (64, 64, 64), 4, 1), (27, (192, 192, 192), 2, 0), (7, (128, 128, 32), 2, 1), (27, (192, 192, 192), 4, 2), (7, (256, 256, 64), 8, 3)]


@pytest.mark.parametrize("stencil,dims,world,rank", CASES)
@pytest.mark.parametrize("exact", [0, 1])
def test_fused_mpi_direction_product_equals_the_separate_kernels(hxl, stencil, dims, world, rank, exact):
    from petsc_amd import _lib
    hx = hxl
    _, ks = _lib.load()
    M, keep, plan = slab(ks, stencil, dims, world, rank)
    m = plan["m"]
    S = dims[0] * dims[1]
    assert plan["nghost"] == S * ((rank > 0) + (rank < world - 1)) and plan["nrows_c"] == plan["nghost"]
    rng = np.random.default_rng(100 * stencil + rank)
    p0, r, x0 = rng.standard_normal(m), rng.standard_normal(m), rng.standard_normal(m)
    b, a, dconst = 0.731 / 1.913, 1.913 / 2.57, 0.37
    _lib.chk(hx.hipxSetReductionMode(exact))
    try:
        # the separate kernels
        P, R, X, W = _lib.DVec(m, p0), _lib.DVec(m, r), _lib.DVec(m, x0), _lib.DVec(m)
        _lib.chk(hx.hipxCGAypxAxpyR(P.ptr, b, R.ptr, dconst, X.ptr, a, m))
        _lib.chk(hx.hipxMatMultMPI(M.A, M.B, M.halo, P.ptr, M.lvec, W.ptr))
        dref = C.c_double()
        _lib.chk(hx.hipxVecDot(P.ptr, W.ptr, m, C.byref(dref)))
        pr, xr, wr = P.get(), X.get(), W.get()
        # the fused form, twice (both ghost buffers of the exchange), the second time with the scalars read from device memory
        for rep in range(2):
            P.set(p0)
            X.set(x0)
            P2, W2, dd = _lib.DVec(m, np.full(m, np.nan)), _lib.DVec(m, np.full(m, np.nan)), _lib.DVec(4, np.zeros(4))
            fused, d = C.c_int(0), C.c_double()
            if rep == 0:
                _lib.chk(hx.hipxMatMultMPICGDirectionDotBegin(M.A, M.B, M.halo, P.ptr, P2.ptr, R.ptr, dconst, X.ptr, b, a, None, None, None, M.lvec, W2.ptr, m, 5, dd.ptr, C.byref(fused)))
            else:  # b = beta_new / beta_old, a = beta_old / dpi (the launch-ahead loop's device-resident sums)
                sc = _lib.DVec(4, np.array([0.731, 1.913, 2.57, 0.0]))
                _lib.chk(hx.hipxMatMultMPICGDirectionDotBegin(M.A, M.B, M.halo, P.ptr, P2.ptr, R.ptr, dconst, X.ptr, 0.0, 0.0, sc.offset(0), sc.offset(1), sc.offset(2),
                                                              M.lvec, W2.ptr, m, 5, dd.ptr, C.byref(fused)))
            assert fused.value == 1, "the loop-back slab did not take the fused form"
            _lib.chk(hx.hipxRedEnd(5, 1, C.byref(d)))
            assert np.array_equal(P2.get(), pr), "p_new differs"
            assert np.array_equal(X.get(), xr), "x differs"
            assert np.array_equal(W2.get(), wr), "w = A p_new differs (rows %s)" % np.flatnonzero(W2.get() != wr)[:8]
            assert dd.get()[0] == d.value
            if exact:
                assert d.value == dref.value, (d.value, dref.value)
            else:
                assert abs(d.value - dref.value) <= 1e-13 * np.abs(pr * wr).sum(), (d.value, dref.value)
            for v in (P2, W2, dd):
                v.free()
            if rep:
                sc.free()
        for v in (P, R, X, W):
            v.free()
    finally:
        _lib.chk(hx.hipxSetReductionMode(0))
        free(hx, keep)


def solve(hx, ks, M, m, b_h, pipeline, its, pcname):
    from petsc_amd import _lib
    B, X = _lib.DVec(m, b_h), _lib.DVec(m, np.zeros(m))
    pc = _lib.HipxPC()
    ks.HipxPCSetDefaults(C.byref(pc))
    pc.type = {"none": 0, "jacobi": 1}[pcname]
    _lib.chk(ks.HipxPCSetUp(C.byref(pc), C.byref(M)))
    k = _lib.HipxKSP()
    ks.HipxKSPSetDefaults(C.byref(k))
    k.rtol, k.abstol, k.divtol, k.max_it, k.fused, k.pipeline = 1e-50, 1e-300, 1e300, its, 1, pipeline
    hist = np.zeros(its + 8)
    k.history, k.hist_len = hist.ctypes.data, len(hist)
    _lib.chk(ks.HipxKSPSolve_CG(C.byref(k), C.byref(M), C.byref(pc), B.ptr, X.ptr))
    out = (hist[:k.hist_n].copy(), int(k.its), int(k.reason), X.get())
    ks.HipxKSPDestroyWork(C.byref(k))
    ks.HipxPCDestroy(C.byref(pc))
    B.free()
    X.free()
    return out


@pytest.mark.parametrize("stencil,dims,world,rank,pcname", [(7, (64, 64, 64), 4, 1, "jacobi"), (27, (192, 192, 192), 2, 1, "jacobi"), (7, (128, 128, 32), 2, 0, "none")])
def test_launch_ahead_cg_on_a_rank_with_an_off_diagonal_block(hxl, stencil, dims, world, rank, pcname):
    """cg_step_pipelined on the loop-back rank (the fused direction + product kernel, PackCG exchange, off-diagonal kernel with the dot) against the
    host-synchronised loop of the same library over the separate kernels (pipeline = 2): exact reductions -- the SAME history and solution, bit for bit;
    default reductions -- the history to 1e-12 per entry over 30 iterations."""
    from petsc_amd import _lib
    hx = hxl
    _, ks = _lib.load()
    M, keep, plan = slab(ks, stencil, dims, world, rank)
    m = plan["m"]
    b_h = 1.0 + (np.arange(m) % 17) / 17.0
    try:
        for exact in (1, 0):
            _lib.chk(hx.hipxSetReductionMode(exact))
            ref = solve(hx, ks, M, m, b_h, 2, 30, pcname)
            got = solve(hx, ks, M, m, b_h, 1, 30, pcname)
            assert got[1:3] == ref[1:3] and len(got[0]) == len(ref[0]) == 31
            if exact:
                assert np.array_equal(got[0], ref[0]) and np.array_equal(got[3], ref[3])
            else:
                rel = np.abs(got[0] - ref[0]) / np.abs(ref[0])
                assert rel.max() <= 1e-12, rel.max()
    finally:
        _lib.chk(hx.hipxSetReductionMode(0))
        free(hx, keep)
