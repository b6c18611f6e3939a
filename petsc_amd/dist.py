"""MATMPIAIJHIPX set-up for one-process-per-GPU runs launched by torchrun (bench.py, tests).

The integer work is done in C (petsc_amd/host/hipx_mpiaij.c: diag/off-diag split, garray, receive plan).
Only the set-up-time exchange of request lists -- the mirror of sfbasic.c:234-247, where the reference
Isends the remote root indices once at PetscSF set-up -- goes through torch.distributed here, because the
ranks are torchrun processes; inside PETSc (the plugin) the same lists come from PetscSFGetRootRanks on
Mat_MPIAIJ.Mvctx.  The per-iteration data path (ghost values, dot/norm all-reduces) is RCCL inside libhipx.
"""
import ctypes as C

import numpy as np

from . import _lib


def _arr(ptr, cnt, dt, npdt):
    if cnt <= 0:
        return np.zeros(0, npdt)
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(dt)), shape=(cnt,)).copy()


def split_ownership(N, nranks):
    _, ks = _lib.load()
    ranges = np.zeros(nranks + 1, np.int32)
    ks.HipxSplitOwnership(int(N), int(nranks), ranges.ctypes.data_as(C.c_void_p))
    return ranges


def ng_chk(plan):
    return int(plan["nghost"]) > 0


def build_plan(ai, aj, aa, ranges, rank, dist=None, group=None, loopback=False):
    """Local slab (global column ids) -> dict with the split CSR blocks and the ghost-exchange plan."""
    _, ks = _lib.load()
    nranks = len(ranges) - 1
    rs, re = int(ranges[rank]), int(ranges[rank + 1])
    m = re - rs
    ai = np.ascontiguousarray(ai, np.int32)
    aj = np.ascontiguousarray(aj, np.int32)
    aa = np.ascontiguousarray(aa, np.float64)
    s = _lib.MPIAIJSplit()
    _lib.chk(ks.HipxMatSetUpMultiply_MPIAIJ(m, rs, re, ai.ctypes.data_as(C.c_void_p), aj.ctypes.data_as(C.c_void_p), aa.ctypes.data_as(C.c_void_p), C.byref(s)))
    nza = int(_arr(s.Ai, m + 1, C.c_int32, np.int32)[m]) if m else 0
    out = {"m": m, "rstart": rs, "nghost": int(s.nghost), "nrows_c": int(s.nrows_c)}
    out["Ai"] = _arr(s.Ai, m + 1, C.c_int32, np.int32)
    out["Aj"] = _arr(s.Aj, nza, C.c_int32, np.int32)
    out["Aa"] = _arr(s.Aa, nza, C.c_double, np.float64)
    out["Bi"] = _arr(s.Bi, s.nrows_c + 1, C.c_int32, np.int32)
    nzb = int(out["Bi"][-1]) if s.nrows_c else 0
    out["Bj"] = _arr(s.Bj, nzb, C.c_int32, np.int32)
    out["Ba"] = _arr(s.Ba, nzb, C.c_double, np.float64)
    out["ridx"] = _arr(s.ridx, s.nrows_c, C.c_int32, np.int32)
    out["garray"] = _arr(s.garray, s.nghost, C.c_int32, np.int32)
    nrecv = C.c_int()
    rr = np.zeros(max(nranks, 1), np.int32)
    ro = np.zeros(nranks + 1, np.int32)
    rg = np.ascontiguousarray(ranges, np.int32)
    _lib.chk(ks.HipxHaloRecvPlan(int(s.nghost), s.garray, nranks, rg.ctypes.data_as(C.c_void_p), C.byref(nrecv), rr.ctypes.data_as(C.c_void_p),
                                 ro.ctypes.data_as(C.c_void_p)))
    ks.HipxMPIAIJSplitFree(C.byref(s))
    out["recv_ranks"] = rr[:nrecv.value].copy()
    out["recv_off"] = ro[:nrecv.value + 1].copy()
    if loopback:
        # One process plays rank `rank` of `nranks` ALONE (bench.py's per_rank_budget leg, tests): the slab's two blocks, ghost count and receive lists are the
        # real ones; the ghost values come from the rank's own rows, wrapped (ghost g <- local row (g - rstart) mod m: the neighbour's last / first plane is
        # played by the rank's own last / first plane) through ONE self-exchange on the IPC transport -- the same kernels, launches and bytes as between two
        # neighbours, no peer needed.  The operator this defines (periodic in the slab direction) is only ever compared with itself.
        assert m > 0 and ng_chk(out), "loopback: the rank needs rows and ghosts"
        loc = (out["garray"].astype(np.int64) - rs) % m
        out["recv_ranks"] = np.zeros(1, np.int32)
        out["recv_off"] = np.asarray([0, len(loc)], np.int32)
        out["send_ranks"] = np.zeros(1, np.int32)
        out["send_off"] = np.asarray([0, len(loc)], np.int32)
        out["send_idx"] = loc.astype(np.int32)
        return out
    # requests: tell each owner which of its entries we need (global ids); it answers by packing them each MatMult
    requests = [None] * nranks
    for k in range(nrecv.value):
        requests[int(rr[k])] = out["garray"][ro[k]:ro[k + 1]].copy()
    if nranks > 1:
        assert dist is not None, "torch.distributed needed to exchange the request lists"
        gathered = [None] * nranks
        dist.all_gather_object(gathered, requests, group=group)
        incoming = [gathered[src][rank] for src in range(nranks)]
    else:
        incoming = [None]
    send_ranks, send_off, send_idx = [], [0], []
    for src in range(nranks):
        req = incoming[src]
        if req is None or len(req) == 0:
            continue
        loc = np.asarray(req, np.int64) - rs
        assert loc.min() >= 0 and loc.max() < m, "request outside the owned range"
        send_ranks.append(src)
        send_idx.append(loc.astype(np.int32))
        send_off.append(send_off[-1] + len(loc))
    out["send_ranks"] = np.asarray(send_ranks, np.int32)
    out["send_off"] = np.asarray(send_off, np.int32)
    out["send_idx"] = np.concatenate(send_idx).astype(np.int32) if send_idx else np.zeros(0, np.int32)
    return out


def comm_init(rank, nranks, dist, transport="rccl"):
    """Scalar all-reduce + ghost-exchange transport of libhipx for torchrun ranks: "rccl" (ncclUniqueId of rank 0 broadcast
    through torch.distributed) or "ipc" (IPC-mapped arenas, peer stores: also when the ranks share one GPU)."""
    hx, _ = _lib.load()
    if transport == "ipc":
        h = (C.c_char * 64)()
        _lib.chk(hx.hipxCommIpcExport(rank, nranks, h))
        allh = [None] * nranks
        dist.all_gather_object(allh, bytes(h))
        _lib.chk(hx.hipxCommIpcAttach(b"".join(allh)))
    else:
        idb = (C.c_char * 256)()
        if rank == 0:
            _lib.chk(hx.hipxCommGetUniqueId(idb))
        box = [bytes(idb)]
        dist.broadcast_object_list(box, src=0)
        _lib.chk(hx.hipxCommInit(box[0], rank, nranks))


def comm_init_loopback():
    """A one-rank IPC communicator (the all-reduce kernel and its flags run as between ranks, over one contribution)."""
    hx, _ = _lib.load()
    h = (C.c_char * 64)()
    _lib.chk(hx.hipxCommIpcExport(0, 1, h))
    _lib.chk(hx.hipxCommIpcAttach(bytes(h)))


def create_device_mat(plan, nranks, rank=0, dist=None, transport="rccl", loopback=False):
    """Uploads the blocks and creates the halo object: returns (HipxMat struct, keepalive list).  loopback: `plan` came from build_plan(..., loopback=True);
    the HipxMat says nranks = 2 so that the host layer takes its several-ranks path (all-reduces on the stream) over the one-rank communicator."""
    hx, _ = _lib.load()
    m, ng = plan["m"], plan["nghost"]
    A = _lib.mat_create_csr(m, m, plan["Ai"], plan["Aj"], plan["Aa"])
    M = _lib.HipxMat(m=m, A=A, B=None, halo=None, lvec=None, nranks=2 if loopback else nranks)
    keep = [A]
    if loopback:
        B = _lib.mat_create_cprow(m, max(ng, 1), plan["nrows_c"], plan["Bi"], plan["ridx"], plan["Bj"], plan["Ba"])
        halo = C.c_void_p()
        p = plan
        _lib.chk(hx.hipxHaloCreate(1, p["send_ranks"].ctypes.data_as(C.c_void_p), p["send_off"].ctypes.data_as(C.c_void_p), p["send_idx"].ctypes.data_as(C.c_void_p), 1,
                                   p["recv_ranks"].ctypes.data_as(C.c_void_p), p["recv_off"].ctypes.data_as(C.c_void_p), C.byref(halo)))
        blob = (C.c_char * 1024)()
        _lib.chk(hx.hipxHaloIpcExport(halo, 0, 1, blob))
        _lib.chk(hx.hipxHaloIpcAttach(halo, bytes(blob)))
        lvec = _lib.DVec(max(ng, 1))
        M.B, M.halo, M.lvec = B, halo, lvec.ptr
        keep += [B, halo, lvec]
        return M, keep
    if nranks > 1:
        B = _lib.mat_create_cprow(m, max(ng, 1), plan["nrows_c"], plan["Bi"], plan["ridx"], plan["Bj"], plan["Ba"])
        halo = C.c_void_p()
        p = plan
        _lib.chk(hx.hipxHaloCreate(len(p["send_ranks"]), p["send_ranks"].ctypes.data_as(C.c_void_p), p["send_off"].ctypes.data_as(C.c_void_p),
                                   p["send_idx"].ctypes.data_as(C.c_void_p), len(p["recv_ranks"]), p["recv_ranks"].ctypes.data_as(C.c_void_p),
                                   p["recv_off"].ctypes.data_as(C.c_void_p), C.byref(halo)))
        if transport == "ipc":
            blob = (C.c_char * 1024)()
            _lib.chk(hx.hipxHaloIpcExport(halo, rank, nranks, blob))
            allb = [None] * nranks
            dist.all_gather_object(allb, bytes(blob))
            _lib.chk(hx.hipxHaloIpcAttach(halo, b"".join(allb)))
        lvec = _lib.DVec(max(ng, 1))
        M.B, M.halo, M.lvec = B, halo, lvec.ptr
        keep += [B, halo, lvec]
    return M, keep
