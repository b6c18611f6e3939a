"""GPU test of the ghost-exchange machinery on ONE GPU: a single-rank RCCL communicator exchanging with itself
(ncclSend/ncclRecv to self inside one group), pack kernel, comm-stream events and the overlapped
MatMult_MPIAIJ sequence (mpiaij.c:1056-1059).  The matrix is the middle slab of a 3-rank partition whose ghost
values are supplied through the self-exchange, checked against the oracle's global product."""
import ctypes as C

import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu


def test_single_rank_comm_allreduce_and_self_halo(hx, monkeypatch):
    import os
    os.environ["HIPX_FORCE_ALLREDUCE"] = "1"  # read once by libhipx at the first chained reduction
    from petsc_amd import _lib
    from petsc_amd import dist as pdist
    _, ks = _lib.load()
    idb = (C.c_char * 256)()
    _lib.chk(hx.hipxCommGetUniqueId(idb))
    _lib.chk(hx.hipxCommInit(idb, 0, 1))
    v = (C.c_double * 3)(1.5, -2.0, 7.0)
    _lib.chk(hx.hipxCommAllreduceSum(v, 3))
    assert list(v) == [1.5, -2.0, 7.0]
    # local dots -> ncclAllReduce on the pinned result words -> single host wait (HIPX_FORCE_ALLREDUCE exercises the chain
    # on this 1-rank communicator): must equal the plain local dots bit for bit
    nn = 300001
    rng0 = np.random.default_rng(9)
    xa, ya, yb = rng0.standard_normal(nn), rng0.standard_normal(nn), rng0.standard_normal(nn)
    XA, YA, YB = _lib.DVec(nn, xa), _lib.DVec(nn, ya), _lib.DVec(nn, yb)
    ptrs = (C.c_void_p * 2)(YA.ptr.value, YB.ptr.value)
    r1, r2 = (C.c_double * 2)(), (C.c_double * 2)()
    for _ in range(3):
        _lib.chk(hx.hipxVecMDotAllreduce(XA.ptr, 2, ptrs, nn, r1))
    _lib.chk(hx.hipxVecMDot(XA.ptr, 2, ptrs, nn, r2))
    assert list(r1) == list(r2) and abs(r1[0] - float(xa @ ya)) < 1e-9
    # exact reduction mode: the local kernel leaves (hi, lo) pairs, the ranks' pairs are all-gathered and folded (ncclAllGather + fold
    # kernel on this 1-rank communicator): the correctly rounded dot products, equal to the plain exact-mode MDot
    _lib.chk(hx.hipxSetReductionMode(1))
    try:
        _lib.chk(hx.hipxVecMDotAllreduce(XA.ptr, 2, ptrs, nn, r1))
        _lib.chk(hx.hipxVecMDot(XA.ptr, 2, ptrs, nn, r2))
    finally:
        _lib.chk(hx.hipxSetReductionMode(0))
    import os
    shim = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "libexactblas.so"))
    shim.exactblas_dot2.restype = C.c_double
    shim.exactblas_dot2.argtypes = [C.c_long, C.c_void_p, C.c_void_p]
    assert list(r1) == list(r2)
    assert r1[0] == shim.exactblas_dot2(nn, xa.ctypes.data_as(C.c_void_p), ya.ctypes.data_as(C.c_void_p))
    assert r1[1] == shim.exactblas_dot2(nn, xa.ctypes.data_as(C.c_void_p), yb.ctypes.data_as(C.c_void_p))
    for d in (XA, YA, YB):
        d.free()
    # periodic 1-D chain: y = A_d x + B_o x[send_idx]; the "ghosts" are this rank's own first/last entries
    m = 5000
    rng = np.random.default_rng(2)
    ai, aj, aa = orc.stencil("5pt", 50, m=100)
    ng = 64
    send_idx = np.sort(rng.choice(m, size=ng, replace=False)).astype(np.int32)
    # off-diagonal block: every 7th row references 2 ghosts
    rows = np.arange(0, m, 7, dtype=np.int32)
    ci = np.arange(0, 2 * len(rows) + 1, 2, dtype=np.int32)
    bj = np.sort(rng.integers(0, ng, size=(len(rows), 2)), axis=1).astype(np.int32).ravel()
    ba = rng.standard_normal(len(bj))
    A = _lib.mat_create_csr(m, m, ai, aj, aa)
    B = _lib.mat_create_cprow(m, ng, len(rows), ci, rows, bj, ba)
    halo = C.c_void_p()
    sr = np.zeros(1, np.int32)
    so = np.array([0, ng], np.int32)
    _lib.chk(hx.hipxHaloCreate(1, sr.ctypes.data_as(C.c_void_p), so.ctypes.data_as(C.c_void_p), send_idx.ctypes.data_as(C.c_void_p), 1, sr.ctypes.data_as(C.c_void_p),
                               so.ctypes.data_as(C.c_void_p), C.byref(halo)))
    x = rng.standard_normal(m)
    X, Y, LV = _lib.DVec(m, x), _lib.DVec(m), _lib.DVec(ng)
    for _ in range(3):
        _lib.chk(hx.hipxMatMultMPI(A, B, halo, X.ptr, LV.ptr, Y.ptr))
    y = Y.get()
    assert np.array_equal(LV.get(), x[send_idx])
    yd = orc.matmult(ai, aj, aa, x)
    bi_full = np.zeros(m + 1, np.int32)
    cnt = np.zeros(m, np.int32)
    cnt[rows] = 2
    bi_full[1:] = np.cumsum(cnt)
    z = np.zeros(m)
    lv = np.ascontiguousarray(x[send_idx])
    orc.lib().orc_MatMultAdd_SeqAIJ(m, orc.P(bi_full), orc.P(bj), orc.P(ba), orc.P(lv), orc.P(yd), orc.P(z))
    assert np.array_equal(y, z)
    # the C host layer drives the same objects (HipxMatMult with B != NULL), CG + Jacobi runs on the MPI code path
    M = _lib.HipxMat(m=m, A=A, B=B, halo=halo, lvec=LV.ptr, nranks=1)
    _lib.chk(ks.HipxMatMult(C.byref(M), X.ptr, Y.ptr))
    assert np.array_equal(Y.get(), z)
    # CG on the multi-rank code path (nranks = 2 in the descriptor: every dot/norm goes through the kernel -> ncclAllReduce ->
    # host-flag chain on this 1-rank communicator, MatMult through the halo): the fused update kernel with its 16-byte
    # all-reduce must reproduce the one-kernel-per-call path
    M2 = _lib.HipxMat(m=m, A=A, B=B, halo=halo, lvec=LV.ptr, nranks=2)
    bvec = orc.matmult(ai, aj, aa, np.ones(m))
    BV = _lib.DVec(m, bvec)
    pcj = _lib.HipxPC()
    ks.HipxPCSetDefaults(C.byref(pcj))
    _lib.chk(ks.HipxPCSetUp(C.byref(pcj), C.byref(M2)))
    sols = []
    for fused in (0, 1):
        k = _lib.HipxKSP()
        ks.HipxKSPSetDefaults(C.byref(k))
        k.rtol, k.max_it, k.fused = 1e-30, 8, fused
        hist = np.zeros(64)
        k.history, k.hist_len = hist.ctypes.data, 64
        XS = _lib.DVec(m, np.zeros(m))
        _lib.chk(ks.HipxKSPSolve_CG(C.byref(k), C.byref(M2), C.byref(pcj), BV.ptr, XS.ptr))
        sols.append((XS.get(), int(k.its), int(k.reason), hist[:k.hist_n].copy()))
        ks.HipxKSPDestroyWork(C.byref(k))
        XS.free()
    assert sols[0][1:3] == sols[1][1:3] == (8, -3)
    assert np.abs(sols[0][3] - sols[1][3]).max() <= 1e-12 * sols[0][3][0]
    assert np.abs(sols[0][0] - sols[1][0]).max() <= 1e-12 * np.abs(sols[0][0]).max()
    # the chained fused update alone: identical to the local kernel bit for bit on one rank
    rr = rng.standard_normal((6, m))
    def fused_update(fn):
        vs = [_lib.DVec(m, rr[i]) for i in range(6)]
        out = (C.c_double * 2)()
        _lib.chk(fn(vs[0].ptr, vs[1].ptr, vs[2].ptr, vs[3].ptr, vs[4].ptr, vs[5].ptr, 0.37, m, out))
        res = (vs[0].get(), vs[1].get(), vs[2].get(), list(out))
        for v in vs:
            v.free()
        return res
    ua, ub = fused_update(hx.hipxCGFusedUpdate), fused_update(hx.hipxCGFusedUpdateAllreduce)
    assert all(np.array_equal(p, q) for p, q in zip(ua[:3], ub[:3])) and ua[3] == ub[3]
    # round 6: PIPECG and Gropp's CG on the multi-rank code path over RCCL -- the SPLIT-PHASE all-reduce (hipxPipeCGUpdateBeginAllreduce: ncclAllReduce / ncclAllGather + fold on
    # its own stream between two events, the product on the compute stream meanwhile, hipxAllreduceEnd) on this 1-rank communicator must reproduce the one-rank
    # descriptor's local reductions bit for bit, plain and compensated, launch-ahead and host-synchronised
    for exact, solver, its_expected in ((0, ks.HipxKSPSolve_PIPECG, 9), (1, ks.HipxKSPSolve_PIPECG, 9), (0, ks.HipxKSPSolve_GROPPCG, 8), (1, ks.HipxKSPSolve_GROPPCG, 8)):
        _lib.chk(hx.hipxSetReductionMode(exact))
        try:
            runs = []
            for desc, pipeline in ((M, 1), (M2, 1), (M2, 0)):
                k = _lib.HipxKSP()
                ks.HipxKSPSetDefaults(C.byref(k))
                k.rtol, k.max_it, k.pipeline = 1e-30, 8, pipeline
                hist = np.zeros(64)
                k.history, k.hist_len = hist.ctypes.data, 64
                XS = _lib.DVec(m, np.zeros(m))
                _lib.chk(solver(C.byref(k), C.byref(desc), C.byref(pcj), BV.ptr, XS.ptr))
                runs.append((XS.get(), int(k.its), int(k.reason), hist[:k.hist_n].copy()))
                ks.HipxKSPDestroyWork(C.byref(k))
                XS.free()
            assert runs[0][1:3] == runs[1][1:3] == runs[2][1:3] == (its_expected, -3)
            assert np.array_equal(runs[0][3], runs[1][3]) and np.array_equal(runs[1][3], runs[2][3]), (runs[0][3], runs[1][3])
            assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[1][0], runs[2][0])
        finally:
            _lib.chk(hx.hipxSetReductionMode(0))
    ks.HipxPCDestroy(C.byref(pcj))
    BV.free()
    _lib.chk(hx.hipxHaloDestroy(C.byref(halo)))
    for d in (X, Y, LV):
        d.free()
    _lib.mat_destroy(A)
    _lib.mat_destroy(B)
    _lib.chk(hx.hipxCommFinalize())


def test_ipc_transport_self_exchange_back_to_back(hx):
    """The IPC transport (peer stores + sequence flags + acknowledged double buffers) on one process exchanging with itself:
    MatMult_MPIAIJ's composition bit-identical to the oracle, 40 products back to back with x changing every time (any lost
    acknowledgement or stale buffer shows up as a wrong y), and MatMultAdd_MPIAIJ."""
    from petsc_amd import _lib
    m = 6000
    rng = np.random.default_rng(12)
    ai, aj, aa = orc.stencil("5pt", 60, m=100)
    ng = 192
    send_idx = np.sort(rng.choice(m, size=ng, replace=False)).astype(np.int32)
    rows = np.arange(3, m, 5, dtype=np.int32)
    ci = np.arange(0, 3 * len(rows) + 1, 3, dtype=np.int32)
    bj = np.sort(rng.integers(0, ng, size=(len(rows), 3)), axis=1).astype(np.int32).ravel()
    ba = rng.standard_normal(len(bj))
    A = _lib.mat_create_csr(m, m, ai, aj, aa)
    B = _lib.mat_create_cprow(m, ng, len(rows), ci, rows, bj, ba)
    halo = C.c_void_p()
    sr = np.zeros(1, np.int32)
    so = np.array([0, ng], np.int32)
    _lib.chk(hx.hipxHaloCreate(1, sr.ctypes.data_as(C.c_void_p), so.ctypes.data_as(C.c_void_p), send_idx.ctypes.data_as(C.c_void_p), 1, sr.ctypes.data_as(C.c_void_p),
                               so.ctypes.data_as(C.c_void_p), C.byref(halo)))
    blob = (C.c_char * 1024)()
    _lib.chk(hx.hipxHaloIpcExport(halo, 0, 1, blob))
    _lib.chk(hx.hipxHaloIpcAttach(halo, blob))
    tr = C.c_int()
    _lib.chk(hx.hipxHaloTransport(halo, C.byref(tr)))
    assert tr.value == 1
    bi_full = np.zeros(m + 1, np.int32)
    cnt = np.zeros(m, np.int32)
    cnt[rows] = 3
    bi_full[1:] = np.cumsum(cnt)

    def expect(x, y0=None):
        yd = orc.matmult(ai, aj, aa, x)
        if y0 is not None:
            yd2 = np.zeros(m)
            orc.lib().orc_MatMultAdd_SeqAIJ(m, orc.P(ai), orc.P(aj), orc.P(aa), orc.P(x), orc.P(y0), orc.P(yd2))
            yd = yd2
        z = np.zeros(m)
        lv = np.ascontiguousarray(x[send_idx])
        orc.lib().orc_MatMultAdd_SeqAIJ(m, orc.P(bi_full), orc.P(bj), orc.P(ba), orc.P(lv), orc.P(yd), orc.P(z))
        return z

    xs = [rng.standard_normal(m) for _ in range(40)]
    Xs = [_lib.DVec(m, x) for x in xs]
    Ys = [_lib.DVec(m) for _ in xs]
    LV = _lib.DVec(ng)
    for X, Y in zip(Xs, Ys):  # enqueued back to back: no host synchronisation between the products
        _lib.chk(hx.hipxMatMultMPI(A, B, halo, X.ptr, LV.ptr, Y.ptr))
    for x, Y in zip(xs, Ys):
        assert np.array_equal(Y.get(), expect(x))
    y0 = rng.standard_normal(m)
    Y0, Z = _lib.DVec(m, y0), _lib.DVec(m)
    _lib.chk(hx.hipxMatMultAddMPI(A, B, halo, Xs[3].ptr, LV.ptr, Y0.ptr, Z.ptr))
    assert np.array_equal(Z.get(), expect(xs[3], y0))
    _lib.chk(hx.hipxMatMultAddMPI(A, B, halo, Xs[5].ptr, LV.ptr, Y0.ptr, Y0.ptr))  # in place
    assert np.array_equal(Y0.get(), expect(xs[5], y0))
    _lib.chk(hx.hipxHaloDestroy(C.byref(halo)))
    for d in Xs + Ys + [LV, Y0, Z]:
        d.free()
    _lib.mat_destroy(A)
    _lib.mat_destroy(B)
