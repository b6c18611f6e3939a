"""Out-of-process probe of a multi-GPU transport (RCCL or IPC peer stores) before the real ranks commit to it.

Every rank of a job starts one short-lived child (`python -m petsc_amd.commprobe ...`, no torch) that brings the transport up
through the same libhipx entry points the solver uses (hipxCommInit | hipxCommIpcExport/Attach, hipxHaloCreate ...), runs a
scalar all-reduce and a ring ghost exchange with known answers, and exits 0.  The parent waits with a timeout: a transport
that hangs, crashes or answers wrongly on ANY rank is reported as unusable and the job falls back (RCCL -> IPC) instead of
dying inside a collective.  The children rendezvous through files in a directory the ranks share (one node): the
ncclUniqueId / IPC handles never need the parent's process group.
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np


def _wait_files(paths, timeout):
    t0 = time.time()
    while time.time() - t0 < timeout:
        if all(os.path.exists(p) for p in paths):
            return True
        time.sleep(0.02)
    return False


def _publish(path, data):
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(data)
    os.rename(tmp, path)  # atomic: a reader never sees a partial file


def probe(transport, rank, world, device, d, timeout):
    os.environ.setdefault("HIPX_NO_TORCH", "1")
    os.environ.setdefault("HIPX_IPC_WAIT_SECONDS", str(max(5, int(timeout / 2))))
    from petsc_amd import _lib
    hx = _lib.init(device)
    if transport == "rccl":
        idp = os.path.join(d, "rccl_id.bin")
        if rank == 0:
            idb = (C.c_char * 256)()
            _lib.chk(hx.hipxCommGetUniqueId(idb))
            _publish(idp, bytes(idb))
        if not _wait_files([idp], timeout):
            raise RuntimeError("rank 0 never published the ncclUniqueId")
        _lib.chk(hx.hipxCommInit(open(idp, "rb").read(), rank, world))
    else:
        h = (C.c_char * 64)()
        _lib.chk(hx.hipxCommIpcExport(rank, world, h))
        _publish(os.path.join(d, "ipc_comm_%d.bin" % rank), bytes(h))
        files = [os.path.join(d, "ipc_comm_%d.bin" % r) for r in range(world)]
        if not _wait_files(files, timeout):
            raise RuntimeError("not every rank exported its all-reduce arena")
        _lib.chk(hx.hipxCommIpcAttach(b"".join(open(f, "rb").read() for f in files)))
    # scalar all-reduce with a known answer, several in a row (the IPC form is double-buffered by sequence parity)
    for k in range(5):
        v = np.array([rank + 1.0 + k, 0.5 * (rank + 1)], np.float64)
        _lib.chk(hx.hipxCommAllreduceSum(v.ctypes.data_as(C.c_void_p), 2))
        want = np.array([world * (world + 1) / 2.0 + k * world, 0.25 * world * (world + 1)])
        if not np.array_equal(v, want):
            raise RuntimeError("all-reduce returned %s, expected %s" % (v, want))
    # compensated reductions over the ranks (hipxSetReductionMode(exact): every rank's (hi, lo) pair folded in rank order, rounded once):
    # a cancelling dot product cut into one slab per rank must come out as the correctly rounded value of the WHOLE sum on every rank
    from fractions import Fraction
    L = 1024
    rng = np.random.default_rng(20260926)
    half = world * L // 2
    a, b = rng.standard_normal(half) * 10.0 ** rng.integers(-6, 6, half), rng.standard_normal(half)
    perm = rng.permutation(2 * half)
    gx, gy = np.concatenate([a, a])[perm], np.concatenate([b, -b * (1.0 + 1e-10 * rng.standard_normal(half))])[perm]
    want = float(sum(Fraction(float(u)) * Fraction(float(v)) for u, v in zip(gx, gy)))
    _lib.chk(hx.hipxSetReductionMode(1))
    X, Y = _lib.DVec(L, gx[rank * L:(rank + 1) * L]), _lib.DVec(L, gy[rank * L:(rank + 1) * L])
    ptrs, res = (C.c_void_p * 1)(Y.ptr.value), (C.c_double * 1)()
    exact_note = ""
    for k in range(3):
        _lib.chk(hx.hipxVecMDotAllreduce(X.ptr, 1, ptrs, L, res))
        if res[0] != want:  # reported, not fatal: the transport itself works (checked above); the timed runs use the default reductions
            exact_note = " (EXACT-MODE all-reduce mismatch: got %r, the correctly rounded sum is %r)" % (res[0], want)
    _lib.chk(hx.hipxSetReductionMode(0))
    X.free()
    Y.free()
    # ring ghost exchange: rank r sends x[0:n] to r+1 and receives from r-1 (a slab partition's neighbour pattern)
    n = 4096
    right, left = (rank + 1) % world, (rank - 1) % world
    sr, so, si = np.array([right], np.int32), np.array([0, n], np.int32), np.arange(n, dtype=np.int32)
    rr, ro = np.array([left], np.int32), np.array([0, n], np.int32)
    halo = C.c_void_p()
    _lib.chk(hx.hipxHaloCreate(1, sr.ctypes.data_as(C.c_void_p), so.ctypes.data_as(C.c_void_p), si.ctypes.data_as(C.c_void_p), 1, rr.ctypes.data_as(C.c_void_p),
                               ro.ctypes.data_as(C.c_void_p), C.byref(halo)))
    if transport == "ipc":
        blob = (C.c_char * 1024)()
        _lib.chk(hx.hipxHaloIpcExport(halo, rank, world, blob))
        _publish(os.path.join(d, "ipc_halo_%d.bin" % rank), bytes(blob))
        files = [os.path.join(d, "ipc_halo_%d.bin" % r) for r in range(world)]
        if not _wait_files(files, timeout):
            raise RuntimeError("not every rank exported its ghost arena")
        _lib.chk(hx.hipxHaloIpcAttach(halo, b"".join(open(f, "rb").read() for f in files)))
    X, L = _lib.DVec(n), _lib.DVec(n)
    for k in range(4):  # back-to-back exchanges (IPC: buffer acknowledgement path)
        X.set(1000.0 * rank + k + np.arange(n) / 8.0)
        L.set(np.zeros(n))
        _lib.chk(hx.hipxHaloBegin(halo, X.ptr, L.ptr))
        _lib.chk(hx.hipxHaloEnd(halo))
        g = C.c_void_p()
        _lib.chk(hx.hipxHaloGhost(halo, L.ptr, C.byref(g)))
        got = np.empty(n)
        _lib.chk(hx.hipxMemcpyDtoH(got.ctypes.data_as(C.c_void_p), g, C.c_size_t(8 * n)))
        _lib.chk(hx.hipxHaloRelease(halo))
        if not np.array_equal(got, 1000.0 * left + k + np.arange(n) / 8.0):
            raise RuntimeError("ghost exchange %d delivered wrong values" % k)
    _lib.chk(hx.hipxDeviceSynchronize())
    uid = C.c_ulonglong()
    _lib.chk(hx.hipxDeviceUID(C.byref(uid)))
    _lib.chk(hx.hipxHaloDestroy(C.byref(halo)))
    _lib.chk(hx.hipxCommFinalize())
    return uid.value, exact_note


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--transport", required=True, choices=["rccl", "ipc"])
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--device", type=int, required=True)
    ap.add_argument("--dir", required=True)
    ap.add_argument("--timeout", type=float, default=60.0)
    a = ap.parse_args()
    try:
        uid, note = probe(a.transport, a.rank, a.world, a.device, a.dir, a.timeout)
    except Exception as e:  # noqa: BLE001
        print("commprobe %s rank %d: FAILED: %s" % (a.transport, a.rank, e), flush=True)
        sys.exit(1)
    print("commprobe %s rank %d: ok device_uid %x%s" % (a.transport, a.rank, uid, note), flush=True)


if __name__ == "__main__":
    main()
