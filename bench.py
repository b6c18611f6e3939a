#!/usr/bin/env python
"""bench.py -- the BASELINE.json metric: CG iterations/s (+ SpMV achieved HBM GB/s) on 3-D 7-point Poisson 256^3, fp64,
KSPCG + PCJACOBI, MATAIJHIPX/VECHIPX kernels, on N GPUs of one node (strong scaling: the 256^3 problem is split by rows).

A "step" is one Krylov iteration (CG: one pass of the loop body cg.c:220-349 -- MatMult, 2 dots, 1 norm, 2 AXPY, 1 AYPX,
PCApply; GMRES: one pass of gmres.c:123-166).  Inputs (CSR matrix, b = A*1, x0 = 0) are resident in HBM before the timed
region.  No per-launch events are recorded inside the timed region; kernel durations come from a second pass of the same K steps.

  python bench.py                                   # BASELINE config 2 on one GPU + the other single-GPU configurations
  python bench.py --gpus 8                          # self-launches 8 ranks (torch.distributed.run), one GPU per rank
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
  python bench.py --ksp gmres --pc sor --stencil 27 --grid 256 --quick               # config 3's solver alone

Prints ONE JSON line on rank 0 (contract in the task statement) with these extra objects:
  parity_gate      -- N = 1: the first 24 iterations of THIS configuration against the REFERENCE's own KSPSolve run beside it
                      with exact (twice-working-precision) BLAS reductions (oracle/_ref + oracle/libexactblas.so), entry by
                      entry, 1e-12 relative; N > 1: against the committed history of that same yardstick
                      (tests/golden/exact_histories.json).  `value` is null when the gate fails; "ungated": true when it
                      could not run.
  roofline         -- dominant kernel (the CSR SpMV the solver launches): HIP-event launch time; `frac` = HBM bytes the kernel
                      really moves (rocprofv3 PMC passes run from inside this script) / time / 8 TB/s; `effective_gbps` =
                      algorithmic CSR bytes / time
  roofline_general -- the same for the general-valued CSR kernel (what a matrix with arbitrary values gets), timed in this run
  plugin           -- the drop-in itself: the reference's executable + libpetschipx.so, its/s of KSPSolve
  cpu_baseline     -- the reference's own KSPSolve (oracle/_ref) on this host: best of P, P/2, P/4 MPI ranks, and 1 core
  other_configs    -- N = 1: BASELINE configs 3 / 4 / 5 on one GPU (GMRES(30)+PCSOR 27-pt 256^3 with `roofline_sor`; the Flan_1565
                      surrogate's SpMV with `roofline_longrow`; config 5's per-GPU share 1024 x 1024 x 128), each with its
                      own cpu_baseline;  N > 1: the north_star scaling legs (27-pt 512^3 CG+Jacobi strong, config 5 weak,
                      config 3) with per-rank SpMV / ghost-exchange / all-reduce times
  multi_gpu        -- N > 1: transports probed and timed (RCCL send/recv + all-reduce, IPC peer stores), devices per rank
"""
import argparse
import ctypes as C
import csv
import glob
import json
import os
import re
import shutil
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); measured copy peak is 6290
GATE_ITS = 24
GATE_TOL = 1e-12       # north_star: residual history within 1e-12 relative, per entry
FAST_MODE_SANITY = 1e-8  # NOT a parity bound: default (fast) reductions on GMRES(30)+SOR are reported beside the exact-mode gate; beyond this the leg is flagged
SHIM = os.path.join(ROOT, "oracle", "libexactblas.so")
GOLDEN = os.path.join(ROOT, "tests", "golden", "exact_histories.json")


# ------------------------------------------------------------------------------------------------------------ configurations
class Cfg:
    """One benchmark configuration: operator, solver, how it is split."""

    def __init__(self, stencil, dims, ksp, pc, scaling="strong", label=None, golden=None):
        self.stencil, self.dims, self.ksp, self.pc, self.scaling, self.label, self.golden = stencil, tuple(dims), ksp, pc, scaling, label, golden
        self.N = dims[0] * dims[1] * dims[2]
        self.cube = dims[0] == dims[1] == dims[2]

    def shape(self):
        if self.stencil == 5:  # 2-D (ex2.c:70-94: row Ii = i n + j, j fastest): dims = (n, m, 1)
            return "%dx%d" % self.dims[:2]
        return "%d^3" % self.dims[0] if self.cube else "%dx%dx%d" % self.dims

    def pcname(self):
        return {"jacobi": "PCJACOBI", "sor": "PCSOR", "none": "PCNONE"}[self.pc]

    def metric(self):
        return "%s iterations/sec, %d-pt Poisson %s fp64, KSP%s+%s" % (self.ksp.upper(), self.stencil, self.shape(), self.ksp.upper(), self.pcname())

    def golden_key(self):
        if self.golden:
            return self.golden
        return "%s_%s_%dpt_%s" % (self.ksp, self.pc, self.stencil, str(self.dims[0]) if self.cube else "%dx%dx%d" % self.dims)

    def driver_args(self, its):
        a = ["-stencil", str(self.stencil), "-n", str(self.dims[0]), "-ksp_type", self.ksp, "-pc_type", self.pc, "-ksp_rtol", "1e-50", "-ksp_max_it", str(its)]
        if self.stencil == 5:
            a += ["-m", str(self.dims[1])]
        elif not self.cube:
            a += ["-ny", str(self.dims[1]), "-nz", str(self.dims[2])]
        if self.ksp == "cg":
            a += ["-ksp_norm_type", "preconditioned"]
        return a


class MatrixCfg(Cfg):
    """A given matrix (BASELINE config 4: SuiteSparse Flan_1565 from a file, or its documented stand-in) solved with KSPCG + PCJACOBI on one GPU.
    `load()` returns CSR (ai, aj, aa); `binfile` (a PETSc binary Mat file of the same matrix, written on demand) is what the REFERENCE
    reads with MatLoad (`ref_driver -f`) for the parity yardstick and the CPU baseline."""

    def __init__(self, name, load, ksp="cg", pc="jacobi"):
        self.name, self._load, self.ksp, self.pc, self.scaling, self.label, self.golden = name, load, ksp, pc, "strong", None, None
        self.stencil, self.dims, self.N, self.cube = 0, (0, 0, 0), 0, False
        self.binfile = None
        self.golden_name = None  # key stem in tests/golden/exact_histories.json ("<ksp>_<pc>_<golden_name>") when the matrix is a deterministic stand-in
        self._cache = None       # the host CSR, shared by the legs of one bench run

    def load(self):
        if self._cache is None:
            self._cache = self._load()
        ai, aj, aa = self._cache
        self.N = len(ai) - 1
        self.dims = (self.N, 1, 1)
        return ai, aj, aa

    def shape(self):
        return self.name

    def metric(self):
        return "%s iterations/sec, %s fp64, KSP%s+%s" % (self.ksp.upper(), self.name, self.ksp.upper(), self.pcname())

    def golden_key(self):
        return "none:" + self.name

    def driver_args(self, its):
        a = ["-f", self.binfile, "-ksp_type", self.ksp, "-pc_type", self.pc, "-ksp_rtol", "1e-50", "-ksp_max_it", str(its)]
        if self.ksp == "cg":
            a += ["-ksp_norm_type", "preconditioned"]
        return a


def assemble(ks, stencil, dims, rs, re):
    nx, ny, nz = dims
    cube = nx == ny == nz
    if stencil == 7:
        def f(ai, ai64, aj, aa):
            return ks.HipxAssemble_poisson7_box(nx, ny, nz, rs, re, ai, ai64, aj, aa)
    elif stencil == 5:  # ex2.c:70-94 on an m x n grid (BASELINE config 1's operator at HBM size): dims = (n, m, 1)
        assert nz == 1, "the 5-point operator of ex2.c is 2-D"

        def f(ai, ai64, aj, aa):
            assert ai64 is None
            return ks.HipxAssemble_ex2(ny, nx, rs, re, ai, aj, aa)
    else:
        assert cube, "the 27-point operator of bench_kspsolve.c is defined on a cube"

        def f(ai, ai64, aj, aa):
            if ai64 is not None:
                return ks.HipxAssemble_bench27_64(nx, rs, re, ai64, aj, aa)
            return ks.HipxAssemble_bench27(nx, rs, re, ai, aj, aa)
    nnz = f(None, None, None, None)
    wide = nnz >= 2 ** 31 - 8  # 64-bit row offsets (27-pt 512^3 / 7-pt 1024^3 on one rank)
    ai = np.zeros(re - rs + 1, np.int64 if wide else np.int32)
    aj = np.zeros(nnz, np.int32)
    aa = np.zeros(nnz, np.float64)
    p = ai.ctypes.data_as(C.c_void_p)
    f(None if wide else p, p if wide else None, aj.ctypes.data_as(C.c_void_p), aa.ctypes.data_as(C.c_void_p))
    return ai, aj, aa


def physical_cores():
    """Physical cores this process may use: distinct (package, core) pairs among the CPUs of its affinity mask."""
    allowed = os.sched_getaffinity(0)
    cores = set()
    for c in allowed:
        try:
            pkg = open("/sys/devices/system/cpu/cpu%d/topology/physical_package_id" % c).read().strip()
            core = open("/sys/devices/system/cpu/cpu%d/topology/core_id" % c).read().strip()
            cores.add((pkg, core))
        except OSError:
            cores.add(("?", str(c)))
    return max(1, len(cores))


def ref_driver(np_, args, plugin=False, timeout=900, bind=False, exact=False):
    """The REFERENCE itself (oracle/_ref: libpetsc compiled from /root/reference by oracle/build_ref.py): its own MatSetValues
    assembly, KSPSolve, MatMult_SeqAIJ / _MPIAIJ, PCJACOBI / PCSOR, MKL BLAS-1 (one thread per rank).  exact=True: the same
    executable with oracle/libexactblas.so LD_PRELOADed (twice-working-precision ddot / dgemv: the parity yardstick)."""
    mp = np_ > 1
    exe = os.path.join(ROOT, "oracle", "_ref", "mpich" if mp else "", "bin", "ref_driver")
    if not os.path.exists(exe):
        return None
    cmd = (["/opt/conda/bin/mpiexec"] + (["-bind-to", "core"] if bind else []) + ["-n", str(np_)] if mp else []) + [exe] + args
    if plugin:
        so = os.path.join(ROOT, "petsc_amd", "lib", "libpetschipx_mpich.so" if mp else "libpetschipx.so")
        if not os.path.exists(so):
            return None
        cmd += ["-dll_prepend", so, "-vec_type", "hipx", "-mat_type", "aijhipx"]
    else:
        cmd += ["-mat_type", "aij", "-vec_type", "standard"]
    env = dict(os.environ, MKL_NUM_THREADS="1", OMP_NUM_THREADS="1", HIPX_NO_TORCH="1")
    if exact:
        if not os.path.exists(SHIM) or plugin:
            return None
        env["LD_PRELOAD"] = SHIM
    try:
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=timeout).stdout
        m = re.search(r"iterations (\d+) reason (-?\d+) error (\S+) KSPSolve_seconds (\S+)", out)
        hist = [float(l.split()[2]) for l in out.splitlines() if l.startswith("hist ")]
        res = {"its": int(m.group(1)), "reason": int(m.group(2)), "error": float(m.group(3)), "seconds": float(m.group(4)), "history": hist}
        m2 = re.search(r"second_solve iterations (\d+) KSPSolve_seconds (\S+)", out)  # (-resolve: the same solve again, nothing left to set up)
        if m2:
            res["second_its"], res["second_seconds"] = int(m2.group(1)), float(m2.group(2))
        return res
    except Exception:
        return None


def golden_history(key):
    try:
        e = json.load(open(GOLDEN))[key]
        return np.array([float.fromhex(v) for v in e["history_hex"]]), e["source"]
    except Exception:
        return None, None


def oracle_port_baseline(ai, aj, aa, b, budget_s, stencil, n):
    """Fallback when oracle/_ref is not on the box: the oracle's scalar C restatement.  Checker code, timed as a baseline."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    t0 = time.perf_counter()
    orc.ksp_solve("cg", ai, aj, aa, b, pc="jacobi", rtol=1e-50, max_it=2)
    t2 = time.perf_counter() - t0
    its = int(max(3, min(60, budget_s / max(t2 / 3.0, 1e-3))))
    t0 = time.perf_counter()
    _, done, _, hist = orc.ksp_solve("cg", ai, aj, aa, b, pc="jacobi", rtol=1e-50, max_it=its)
    t_its = time.perf_counter() - t0
    t0 = time.perf_counter()
    orc.ksp_solve("cg", ai, aj, aa, b, pc="jacobi", rtol=1e-50, max_it=1)
    t_one = time.perf_counter() - t0
    per_it = (t_its - t_one) / max(done - 1, 1)
    return {"value": 1.0 / per_it, "unit": "iterations/s", "cores": 1, "kind": "port",
            "sample": "%d CG+Jacobi iterations of the oracle (scalar C restatement of cg.c/aij.c, gcc -O2) on the same %d-pt %d^3 system" % (done, stencil, n)}, hist


# ------------------------------------------------------------------------------------------------------------------ one problem
class Problem:
    """One rank's share of the system on the device + the solver objects of the C host layer."""

    def __init__(self, cfg, rank, world, dist, transport="rccl", fused=1, pipeline=1, keep_host=False, loopback=False):
        """loopback: this process plays rank `rank` of `world` ALONE (per_rank_budget leg): the rank's real blocks, ghost lists and an IPC self-exchange
        (petsc_amd/dist.py build_plan(..., loopback=True)); a one-rank IPC communicator must be up (dist.comm_init_loopback)."""
        from petsc_amd import _lib
        from petsc_amd import dist as pdist
        self.lib, self.cfg, self.world, self.fused, self.pipeline = _lib, cfg, world, fused, pipeline
        self.hx, self.ks = _lib.load()
        self.dims, self.N = cfg.dims, cfg.N
        self.setup_times = {}  # where set-up time goes: host assembly / upload (+ MPI split) / device format build (first product)
        t_a = time.perf_counter()
        if isinstance(cfg, MatrixCfg):
            assert world == 1, "a matrix from a file runs on one GPU here"
            ai, aj, aa = cfg.load()
            self.dims, self.N = cfg.dims, cfg.N
            ranges, rs, re = None, 0, cfg.N
        else:
            ranges = pdist.split_ownership(self.N, world)
            rs, re = int(ranges[rank]), int(ranges[rank + 1])
            ai, aj, aa = assemble(self.ks, cfg.stencil, self.dims, rs, re)
        self.m = re - rs
        self.wide = ai.dtype == np.int64
        self.halo = self.lvec = self.Bm = None
        self.setup_times["host_assembly_s"] = time.perf_counter() - t_a
        t_a = time.perf_counter()
        if world > 1:
            plan = pdist.build_plan(ai, aj, aa, ranges, rank, dist=dist, loopback=loopback)
            self.M, keep = pdist.create_device_mat(plan, world, rank=rank, dist=dist, transport=transport, loopback=loopback)
            self.Am = keep[0]
            if len(keep) > 1:
                self.Bm, self.halo, self.lvec = keep[1], keep[2], keep[3]
            self.nnz_local = int(plan["Ai"][-1])
            self.nghost = int(plan["nghost"])
            del plan
        else:
            self.Am = _lib.mat_create_csr(self.m, self.m, ai, aj, aa)
            self.M = _lib.HipxMat(m=self.m, A=self.Am, B=None, halo=None, lvec=None, nranks=1)
            self.nnz_local = int(ai[-1])
            self.nghost = 0
        self.host_csr = (ai, aj, aa) if keep_host else None
        del ai, aj, aa
        _lib.chk(self.hx.hipxDeviceSynchronize())
        self.setup_times["upload_s"] = time.perf_counter() - t_a
        t_a = time.perf_counter()
        self.ones = _lib.DVec(self.m, np.ones(self.m))
        self.B = _lib.DVec(self.m)
        self.X = _lib.DVec(self.m)
        _lib.chk(self.hx.hipxDeviceSynchronize())
        self.setup_times["vectors_alloc_upload_s"] = time.perf_counter() - t_a  # (three device vectors, one of them filled from the host)
        t_a = time.perf_counter()
        _lib.chk(self.ks.HipxMatMult(C.byref(self.M), self.ones.ptr, self.B.ptr))  # b = A * 1 (ex2.c:139 style): the first product builds the device formats
        _lib.chk(self.hx.hipxDeviceSynchronize())
        self.setup_times["device_format_build_and_first_product_s"] = time.perf_counter() - t_a
        self.ones.free()
        self.pc = None
        self.ksp = None

    def setup(self, variant=0, no_dconst=False):
        lib, ks = self.lib, self.ks
        lib.chk(self.hx.hipxMatSetSpMVVariant(self.M.A, variant))
        if no_dconst:
            os.environ["HIPX_NO_DCONST"] = "1"
        else:
            os.environ.pop("HIPX_NO_DCONST", None)
        if self.pc is not None:
            ks.HipxKSPDestroyWork(C.byref(self.ksp))
            ks.HipxPCDestroy(C.byref(self.pc))
        self.pc = lib.HipxPC()
        ks.HipxPCSetDefaults(C.byref(self.pc))
        self.pc.type = {"none": 0, "jacobi": 1, "sor": 2}[self.cfg.pc]
        lib.chk(ks.HipxPCSetUp(C.byref(self.pc), C.byref(self.M)))
        self.ksp = lib.HipxKSP()
        ks.HipxKSPSetDefaults(C.byref(self.ksp))
        self.ksp.rtol, self.ksp.abstol, self.ksp.divtol = 1e-50, 1e-300, 1e300
        self.ksp.fused, self.ksp.pipeline = self.fused, self.pipeline
        self.ksp.single_reduction = 1 if self.pipeline in (3, 4) else 0  # --pipeline 3 / 4: KSPSolve_CG_SingleReduction (cg.c:364-534), one reduction stage per iteration (4: launch-ahead)
        kbuf = C.create_string_buffer(256)
        lib.chk(self.hx.hipxMatGetSpMVKernel(self.M.A, kbuf, 256))
        return kbuf.value.decode()

    def solve(self, its, history=False):
        """A fresh solve of exactly `its` iterations from x0 = 0 (rtol = 1e-50: never converges earlier)."""
        lib, ks = self.lib, self.ks
        self.ksp.max_it = its
        hist = np.zeros(its + 8 + its // 30)
        if history:
            self.ksp.history, self.ksp.hist_len = hist.ctypes.data, len(hist)
        else:
            self.ksp.history, self.ksp.hist_len = None, 0
        lib.chk(self.hx.hipxVecSet(self.X.ptr, self.m, 0.0))
        f = {"cg": ks.HipxKSPSolve_CG, "gmres": ks.HipxKSPSolve_GMRES, "pipecg": ks.HipxKSPSolve_PIPECG, "groppcg": ks.HipxKSPSolve_GROPPCG}[self.cfg.ksp]
        lib.chk(f(C.byref(self.ksp), C.byref(self.M), C.byref(self.pc), self.B.ptr, self.X.ptr))
        # (KSPSolve_PIPECG's loop bound is `i <= max_it`, pipecg.c:160: max_it + 1 passes, max_it + 1 history entries)
        assert self.ksp.its == its + (1 if self.cfg.ksp == "pipecg" else 0) and self.ksp.reason == -3, (self.ksp.its, self.ksp.reason)
        return hist[:min(self.ksp.hist_n, len(hist))].copy()

    def begin(self, total_its):
        self.ksp.max_it = total_its
        self.ksp.history, self.ksp.hist_len = None, 0
        self.lib.chk(self.hx.hipxVecSet(self.X.ptr, self.m, 0.0))
        self.lib.chk(self.ks.HipxKSPCGBegin(C.byref(self.ksp), C.byref(self.M), C.byref(self.pc), self.B.ptr, self.X.ptr))

    def step(self, k):
        self.lib.chk(self.ks.HipxKSPCGStep(C.byref(self.ksp), C.byref(self.M), C.byref(self.pc), self.B.ptr, self.X.ptr, k))
        assert self.ksp.reason == 0, self.ksp.reason

    def spmv_bytes(self):
        return 12 * self.nnz_local + (8 if self.wide else 4) * (self.m + 1) + 16 * self.m

    def destroy(self):
        lib, ks, hx = self.lib, self.ks, self.hx
        lib.chk(hx.hipxDeviceSynchronize())
        if self.pc is not None:
            ks.HipxKSPDestroyWork(C.byref(self.ksp))
            ks.HipxPCDestroy(C.byref(self.pc))
            self.pc = self.ksp = None
        for v in (self.B, self.X, self.lvec):
            if v is not None:
                v.free()
        if self.halo is not None:
            lib.chk(hx.hipxHaloDestroy(C.byref(self.halo)))
        for m_ in (self.Am, self.Bm):
            if m_ is not None:
                lib.mat_destroy(m_)
        self.Am = self.Bm = self.halo = self.lvec = self.B = self.X = None
        self.host_csr = None


SECTIONS = {"halo_ms": 0, "allreduce_ms": 1, "offdiag_ms": 2, "sor_ms": 3, "cg_update_ms": 4, "cg_direction_ms": 5, "dot_fold_ms": 6}  # HIPX_PROF_* of include/hipx.h


def timed_steps(P, steps, warmup, sync, dist, torch):
    """W untimed + K timed iterations (max over ranks), then the same K again with HIP events around every SpMV launch (and, on
    this second pass only, around the ghost exchange, the all-reduces, the off-diagonal product and MatSOR)."""
    hx, lib = P.hx, P.lib
    # Burn-in before the W warm-up steps: throw-away solves of the same configuration until BURN_IN_S of wall time have passed, so that the K timed steps (4-5 ms of
    # GPU work at the driver's --steps 20) see the clocks a long solve sees -- the same kernel measured 141 us and 167 us on two boxes of round 6 when the timed
    # region followed a cold start.  Untimed; reported as `burn_in_s` in the detail file.
    burn = float(os.environ.get("HIPX_BENCH_BURNIN_S", "0.4"))
    t_b = time.perf_counter()
    nb = 0
    while burn > 0 and ((dist is None and time.perf_counter() - t_b < burn) or (dist is not None and nb < 6)):  # (several ranks: a solve is collective -- a fixed count, not a clock)
        P.solve(40)
        sync()
        nb += 1
    if P.cfg.ksp == "cg":
        P.begin(warmup + 2 * steps + 10)
        P.step(warmup)
        sync()
        t0 = time.perf_counter()
        P.step(steps)
        sync()
        elapsed = time.perf_counter() - t0
        lib.chk(hx.hipxProfileSpMV(1))
        lib.chk(hx.hipxProfileSections(1))
        P.step(steps)
        sync()
    else:  # GMRES: a solve of exactly K iterations from x0 = 0 (restarts, solution update and work-vector set-up included)
        P.solve(max(warmup, 1))
        sync()
        t0 = time.perf_counter()
        P.solve(steps)
        sync()
        elapsed = time.perf_counter() - t0
        lib.chk(hx.hipxProfileSpMV(1))
        lib.chk(hx.hipxProfileSections(1))
        P.solve(steps)
        sync()
    cnt, tot_ms = C.c_int(), C.c_double()
    lib.chk(hx.hipxProfileSpMVGet(C.byref(cnt), C.byref(tot_ms)))
    lib.chk(hx.hipxProfileSpMV(0))
    sec = {}
    for name, sid in SECTIONS.items():
        c2, t2 = C.c_int(), C.c_double()
        lib.chk(hx.hipxProfileSectionGet(sid, C.byref(c2), C.byref(t2)))
        if c2.value:
            sec[name] = t2.value / c2.value
            sec[name.replace("_ms", "_calls")] = c2.value
    lib.chk(hx.hipxProfileSections(0))
    local = elapsed
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
    return {"elapsed": elapsed, "elapsed_local": local, "spmv_ms": tot_ms.value / max(cnt.value, 1), "launches": cnt.value, "rnorm": float(P.ksp.rnorm), "sections": sec}


def short_kernel_name(kn):
    """'void (anonymous namespace)::spmv_march2_kernel<7, 8, 1, true, true>(hipxMarchPlan, ...)' -> 'spmv_march2_kernel<7, 8, 1, true, true>'"""
    kn = kn.strip().strip('"')
    kn = re.sub(r"^void\s+", "", kn)
    kn = kn.replace("(anonymous namespace)::", "")
    depth, out = 0, []
    for ch in kn:  # cut at the argument list: the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).strip()


PROFILE_KEEP = os.environ.get("HIPX_BENCH_KEEP_PROFILES")  # a directory: the rocprofv3 CSVs of every counter pass are copied there (profiles/ of a round)


def pmc_suite(suite_args, tag, timeout=600):
    """HBM bytes per launch of EVERY kernel of one internal workload (`bench.py --suite ...`): two rocprofv3 --pmc passes (FETCH_SIZE,
    WRITE_SIZE: separate passes, only --kernel-trace beside them), corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section)
    prescribes for gfx950: read bytes = 2 x FETCH_SIZE KiB x 1024; write bytes = WRITE_SIZE KiB x 1024.  Every suite also launches an
    AXPY of known size: the calibration of that correction, reported with the numbers.  Returns {short kernel name: {...}}."""
    exe = shutil.which("rocprofv3")
    if not exe:
        return None
    per = {}
    tmp = tempfile.mkdtemp(prefix="hipx_pmc_")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            cmd = [exe, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__)] + suite_args
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout, cwd=tmp, env=dict(os.environ, TMPDIR=tmp, HIPX_NO_TORCH="1"))
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None
            acc = {}
            for row in csv.DictReader(open(files[0])):
                if row["Counter_Name"] != ctr:
                    continue
                key = short_kernel_name(row["Kernel_Name"])
                acc.setdefault(key, {}).setdefault(row["Dispatch_Id"], 0.0)
                acc[key][row["Dispatch_Id"]] += float(row["Counter_Value"])
            for key, v in acc.items():
                vals = sorted(v.values())
                per.setdefault(key, {})[ctr] = sum(vals) / len(vals)
                per[key]["launches_sampled"] = len(vals)
            if PROFILE_KEEP:
                os.makedirs(PROFILE_KEEP, exist_ok=True)
                with open(os.path.join(PROFILE_KEEP, "%s_pmc_%s_per_kernel.csv" % (tag, ctr)), "w") as f:
                    f.write("kernel,dispatches,avg_%s_KiB\n" % ctr)
                    for key, v in sorted(acc.items()):
                        f.write('"%s",%d,%.3f\n' % (key, len(v), sum(v.values()) / len(v)))
        res = {}
        for key, v in per.items():
            if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
                res[key] = {"bytes": int(2 * v["FETCH_SIZE"] * 1024 + v["WRITE_SIZE"] * 1024), "FETCH_SIZE_KiB": v["FETCH_SIZE"], "WRITE_SIZE_KiB": v["WRITE_SIZE"],
                            "launches_sampled": v["launches_sampled"]}
        return res or None
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pick_kernel(res, needle, exclude=()):
    """The entry of a pmc_suite() result whose kernel name contains `needle` (the one with the most dispatches if several do)."""
    if not res:
        return None, None
    best = None
    for k, v in res.items():
        if needle in k and not any(e in k for e in exclude) and (best is None or v["launches_sampled"] > res[best]["launches_sampled"]):
            best = k
    return (best, res[best]) if best else (None, None)


def calibration(res, nbytes):
    """FETCH_SIZE / WRITE_SIZE correction measured on the suite's own AXPY of `nbytes`-byte vectors (reads 2 vectors, writes 1)."""
    k, v = pick_kernel(res, "ew2_kernel")
    if not v:
        return None
    return {"kernel": "hipxVecAXPY on %d doubles (reads %d B, writes %d B)" % (nbytes // 8, 2 * nbytes, nbytes), "read_bytes_over_FETCH_SIZE_KiB_x1024": 2 * nbytes / (v["FETCH_SIZE_KiB"] * 1024),
            "write_bytes_over_WRITE_SIZE_KiB_x1024": nbytes / (v["WRITE_SIZE_KiB"] * 1024) if v["WRITE_SIZE_KiB"] else None}


def suite_mode(args):
    """Internal workloads for the counter passes (python bench.py --suite NAME ...): a few launches of the kernels of one leg plus an AXPY of
    known size.  cg: the headline solver (fused kernels) for a few iterations; spmv: y = A x with the variants listed in --suite-variants;
    sor: symmetric zero-guess sweeps (PCApply_SOR), --suite-perturb: every nonzero its own value; sell: the config-4 stand-in's SpMV;
    box: the 7-pt operator on --suite-dims (config 5's share)."""
    from petsc_amd import _lib
    hx = _lib.init(0)
    _, ks = _lib.load()
    name = args.suite
    if name == "cg":
        cdims = tuple(int(v) for v in args.suite_dims.split("x")) if args.suite_dims else (args.n, args.n, args.n)
        cfg = Cfg(args.stencil, cdims, "cg", args.pc)
        P = Problem(cfg, 0, 1, None, fused=1, pipeline=1)
        P.setup(args.variant, no_dconst=bool(args.suite_no_dconst))
        P.begin(40)
        P.step(12)
        _lib.chk(hx.hipxVecAXPY(P.X.ptr, 0.5, P.B.ptr, P.m))  # calibration
        _lib.chk(hx.hipxDeviceSynchronize())
        return
    if name == "sell":
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from surrogates import flan_surrogate
        ai, aj, aa = flan_surrogate()
        N = len(ai) - 1
    elif name == "sorflan":  # PCSOR's application on config 4's stand-in (a matrix with inodes: the node-level sweep)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from surrogates import flan_surrogate_spd
        ai, aj, aa = flan_surrogate_spd()
        N = len(ai) - 1
    else:
        dims = tuple(int(v) for v in args.suite_dims.split("x")) if args.suite_dims else (args.n, args.n, args.n)
        N = dims[0] * dims[1] * dims[2]
        ai, aj, aa = assemble(ks, args.stencil, dims, 0, N)
        if args.suite_perturb:
            aa *= 1.0 + 0.3 * np.random.default_rng(1).random(aa.size)
    A = _lib.mat_create_csr(N, N, ai, aj, aa)
    del ai, aj, aa
    X, Y = _lib.DVec(N, 1.0 + (np.arange(N) % 17) / 17.0), _lib.DVec(N)
    if name in ("sor", "sorflan"):
        for _ in range(3):
            _lib.chk(hx.hipxMatSOR(A, X.ptr, 1.0, 12 | 16, 0.0, 1, 1, Y.ptr))
    else:
        for v in [int(t) for t in (args.suite_variants or "0").split(",")]:
            _lib.chk(hx.hipxMatSetSpMVVariant(A, v))
            for _ in range(4):
                _lib.chk(hx.hipxMatMult(A, X.ptr, Y.ptr))
    _lib.chk(hx.hipxVecAXPY(Y.ptr, 0.5, X.ptr, N))  # calibration of the FETCH_SIZE / WRITE_SIZE correction: 2 N doubles read, N written
    _lib.chk(hx.hipxDeviceSynchronize())


# ------------------------------------------------------------------------------------------------------------ multi-GPU helpers
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-run this command as N ranks under torch.distributed.run (the driver's own
    recipe), one rank per GPU.  On a box with fewer GPUs than ranks the ranks share devices (IPC transport; the line says so)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    env["HIPX_SELF_LAUNCHED"] = "1"
    return subprocess.call(cmd, env=env)


def probe_transport(t, rank, world, dev, dist, timeout=90.0):
    """Every rank runs petsc_amd/commprobe.py in a child process (bring-up + all-reduce + ring ghost exchange with known answers);
    usable only if every rank's child exits 0 in time.  Returns (ok, note)."""
    box = [None]
    if rank == 0:
        box[0] = tempfile.mkdtemp(prefix="hipx_probe_%s_" % t)
    dist.broadcast_object_list(box, src=0)
    d = box[0]
    env = dict(os.environ, HIPX_NO_TORCH="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "petsc_amd.commprobe", "--transport", t, "--rank", str(rank), "--world", str(world), "--device", str(dev), "--dir", d, "--timeout", str(timeout * 0.6)]
    ok, note = False, ""
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout, cwd=ROOT, env=env)
        ok = r.returncode == 0
        note = (r.stdout.strip().splitlines() or [""])[-1][-300:]
    except subprocess.TimeoutExpired:
        note = "probe timed out after %.0f s (hang inside the transport)" % timeout
    except Exception as e:  # noqa: BLE001
        note = "probe could not run: %s" % e
    notes = [None] * world
    dist.all_gather_object(notes, (ok, note))
    dist.barrier()
    if rank == 0:
        shutil.rmtree(d, ignore_errors=True)
    allok = all(o for o, _ in notes)
    return allok, [n for _, n in notes]


def parity_vs_golden(P, its, tol):
    """History of a fresh `its`-iteration solve against the committed exact-reduction history of the same configuration, entry by entry at
    `tol` (1e-12).  KSPCG legs: the default (fast) reductions are what is gated -- they are what is timed -- and the exact mode's distance is
    reported beside it.  KSPGMRES legs (30-vector Gram-Schmidt amplifies the rounding of the reductions: ~1e-10 on the tail in fast mode):
    the gate is the EXACT reduction mode (hipxSetReductionMode: compensated sums, the value the reference's exactly rounded BLAS returns) at
    the same 1e-12; the fast mode's distance is reported, not gated."""
    key = P.cfg.golden_key()
    if P.cfg.ksp == "gmres":
        key += "_np%d" % P.world
    href, source = golden_history(key)
    if href is None:
        return {"pass": None, "ungated": True, "reference": "no committed history for %s (tests/golden/exact_histories.json)" % key}
    its = min(its, len(href) - 1)
    lib, hx = P.lib, P.hx

    orig = C.c_int(0)
    lib.chk(hx.hipxGetReductionMode(C.byref(orig)))  # (HIPX_REDUCTIONS=exact runs the whole line in exact mode: put it back afterwards)

    def dist(mode):
        lib.chk(hx.hipxSetReductionMode(mode))
        try:
            hist = P.solve(its, history=True)
        finally:
            lib.chk(hx.hipxSetReductionMode(orig.value))
        k = min(len(hist), its + 1)
        return float((np.abs(hist[:k] - href[:k]) / np.abs(href[:k])).max()), k
    rel_fast, k = dist(0)
    rel_exact, k2 = dist(1)
    gated = "exact" if P.cfg.ksp in ("gmres", "pipecg", "groppcg") else "fast"  # (PIPECG: r, u = B r and w = A u are all recurred -- every rounding of a reduction is carried forward)
    if getattr(P, "pipeline", 1) in (3, 4):  # single-reduction CG: another recurrence for w = A p -- its history leaves the standard form's by rounding, a little more every iteration;
        tol = max(tol, 1e-9)            # bit parity with the REFERENCE's own single-reduction run is what tests/test_gpu_scale_parity.py holds it to
    if P.cfg.ksp in ("pipecg", "groppcg") and P.world > 1:  # the committed history is the one-rank reference's: MatMult_MPIAIJ's association (diagonal block, then the ghost terms) differs from it
        tol = max(tol, 1e-9)                    # by rounding, which the pipelined recurrences carry forward (np > 1 bit parity: tests/test_gpu_plugin_mpi.py against the partitioned oracle)
    rel = rel_exact if gated == "exact" else rel_fast
    out = {"pass": bool(rel <= tol and k == its + 1 and k2 == its + 1), "max_rel_diff": rel, "tolerance": tol, "gated_reduction_mode": gated, "iterations": its, "entries": k,
           "max_rel_diff_fast_reductions": rel_fast, "max_rel_diff_exact_reductions": rel_exact,
           "reference": "tests/golden/exact_histories.json[%s] (%s: the reference's arithmetic with exact BLAS reductions)" % (key, source)}
    if gated == "exact" and rel_fast > FAST_MODE_SANITY:
        out["pass"] = False
        out["note"] = "fast-mode history further than %g from the yardstick" % FAST_MODE_SANITY
    return out


def run_leg(cfg, rank, world, dist, torch, transport, steps, warmup, sync, variant=0, parity_its=GATE_ITS, fused=1, pipeline=1):
    """One configuration on the current communicator: timed steps + section times per rank + parity vs the committed yardstick."""
    t0 = time.perf_counter()
    P = Problem(cfg, rank, world, dist, transport=transport, fused=fused, pipeline=pipeline)
    kname = P.setup(variant)
    t_setup = time.perf_counter() - t0
    par = parity_vs_golden(P, parity_its, GATE_TOL)
    r = timed_steps(P, steps, warmup, sync, dist, torch)
    mine = {"rank": rank, "rows": P.m, "nnz": P.nnz_local, "ghosts": P.nghost, "spmv_ms": r["spmv_ms"], "elapsed_s": r["elapsed_local"]}
    mine.update(r["sections"])
    per_rank = [mine]
    if dist is not None:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    mode = C.c_int(-1)
    if cfg.pc == "sor":
        P.lib.chk(P.hx.hipxMatGetSORMode(P.M.A, C.byref(mode)))
    out = {"metric": cfg.metric(), "iterations_per_s": steps / r["elapsed"], "ms_per_step": 1e3 * r["elapsed"] / steps, "steps": steps, "warmup": warmup,
           "scaling": cfg.scaling, "global_rows": cfg.N, "spmv_kernel": kname, "parity": par, "residual_norm_after": r["rnorm"],
           "setup_seconds": t_setup, "setup_split": dict(P.setup_times), "per_rank": per_rank}
    if cfg.pc == "sor":
        out["sor_schedule"] = {4: "plane march", 3: "inode (node-level dependency-driven)", 2: "strand", 1: "dependency-driven", 0: "levels"}.get(mode.value, str(mode.value))
    nnz_l, m_l, wide = P.nnz_local, P.m, P.wide
    out["spmv_algorithmic_bytes_rank0"] = P.spmv_bytes()
    P.destroy()
    return out, (nnz_l, m_l, wide)


# ----------------------------------------------------------------------------------------------------------- single-GPU extra legs
def cpu_baseline_for(cfg, ranks, its, what):
    r = ref_driver(ranks, cfg.driver_args(its), bind=True, timeout=1200) or ref_driver(ranks, cfg.driver_args(its), timeout=1200)
    if r is None:
        return None
    return {"value": r["its"] / r["seconds"], "unit": "iterations/s", "cores": ranks, "kind": "reference",
            "sample": "the reference's own KSPSolve (%s, MAT(MPI)AIJ, MKL one thread per rank; oracle/_ref) on the same system: %d iterations on %d MPI ranks, KSPSolve wall %.3f s; assembly excluded"
                      % (what, r["its"], ranks, r["seconds"])}


def leg_surrogate_spmv(hx, lib):
    """BASELINE config 4's stand-in (SuiteSparse Flan_1565 is not in the image; tests/surrogates.py documents the substitute): the
    SpMV of a 1.5 M-row, 121 M-nonzero, all-values-distinct FEM-like matrix with the kernel auto picks, sampled rows checked."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from surrogates import flan_surrogate
    t0 = time.perf_counter()
    ai, aj, aa = flan_surrogate()
    N, nnz = len(ai) - 1, int(ai[-1])
    A = lib.mat_create_csr(N, N, ai, aj, aa)
    xh = 1.0 + (np.arange(N) % 17) / 17.0
    X, Y = lib.DVec(N, xh), lib.DVec(N)
    kbuf = C.create_string_buffer(256)
    lib.chk(hx.hipxMatGetSpMVKernel(A, kbuf, 256))
    for _ in range(5):
        lib.chk(hx.hipxMatMult(A, X.ptr, Y.ptr))
    lib.chk(hx.hipxProfileSpMV(1))
    for _ in range(50):
        lib.chk(hx.hipxMatMult(A, X.ptr, Y.ptr))
    cnt, tot = C.c_int(), C.c_double()
    lib.chk(hx.hipxProfileSpMVGet(C.byref(cnt), C.byref(tot)))
    lib.chk(hx.hipxProfileSpMV(0))
    y = Y.get()
    rows = np.random.default_rng(4).integers(0, N, 1500)
    bad = 0
    for r in rows:  # three unknowns per node = a matrix with inodes: MatMult_SeqAIJ_Inode's row sums (inode.c:398-411: the terms in pairs, a last odd one alone)
        s, k, e = 0.0, int(ai[r]), int(ai[r + 1])
        while k + 1 < e:
            s += aa[k] * xh[aj[k]] + aa[k + 1] * xh[aj[k + 1]]
            k += 2
        if k < e:
            s += aa[k] * xh[aj[k]]
        bad += int(s != y[r])
    ms = tot.value / max(cnt.value, 1)
    byts = 12 * nnz + 4 * (N + 1) + 16 * N
    out = {"what": "config 4 surrogate (Flan_1565-like: %d rows, %d nonzeros, %.1f per row, all values distinct)" % (N, nnz, nnz / N), "kernel": kbuf.value.decode(),
           "roofline_longrow": {"bound": "hbm", "avg_launch_ms": ms, "launches": cnt.value, "algorithmic_bytes": byts, "achieved": byts / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": byts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None},
           "sampled_rows_bit_identical": bool(bad == 0), "sampled_rows": len(rows), "build_seconds": time.perf_counter() - t0}
    for v in (X, Y):
        v.free()
    lib.mat_destroy(A)
    return out


def leg_sor_arbitrary_values(hx, lib, ks, n=256):
    """PCSOR's default application (one symmetric zero-guess sweep) on config 3's operator with ARBITRARY values on its pattern (every
    nonzero its own value -- variable-coefficient operators: no row templates): the strand schedule with streamed coefficients, checked
    bit for bit against the level-ordered dependency-driven sweep of the same library (itself held to MatSOR_SeqAIJ by tests/test_gpu_sor.py)."""
    N = n ** 3
    ai, aj, aa = assemble(ks, 27, (n, n, n), 0, N)
    aa *= 1.0 + 0.3 * np.random.default_rng(1).random(aa.size)
    A = lib.mat_create_csr(N, N, ai, aj, aa)
    nnz = int(ai[-1])
    del ai, aj, aa
    B, X = lib.DVec(N, 1.0 + (np.arange(N) % 17) / 17.0), lib.DVec(N)
    e0, e1 = C.c_void_p(), C.c_void_p()
    lib.chk(hx.hipxEventCreate(C.byref(e0)))
    lib.chk(hx.hipxEventCreate(C.byref(e1)))
    res, xs = {}, {}
    old = os.environ.pop("HIPX_SOR_MODE", None)
    try:
        for mode, reps in (("strand", 5), ("dep", 2)):
            os.environ["HIPX_SOR_MODE"] = mode
            for _ in range(2):
                lib.chk(hx.hipxMatSOR(A, B.ptr, 1.0, 12 | 16, 0.0, 1, 1, X.ptr))
            lib.chk(hx.hipxEventRecord(e0))
            for _ in range(reps):
                lib.chk(hx.hipxMatSOR(A, B.ptr, 1.0, 12 | 16, 0.0, 1, 1, X.ptr))
            lib.chk(hx.hipxEventRecord(e1))
            ms = C.c_float()
            lib.chk(hx.hipxEventElapsedMs(e0, e1, C.byref(ms)))
            res[mode] = ms.value / reps
            xs[mode] = X.get()
    finally:
        os.environ.pop("HIPX_SOR_MODE", None)
        if old is not None:
            os.environ["HIPX_SOR_MODE"] = old
    ssor = 2 * 12 * nnz + 40 * N
    out = {"what": "27-pt %d^3 pattern, every nonzero its own value: one PCApply_SOR (symmetric zero-guess sweep)" % n,
           "strand_streamed_coefficients_ms": res["strand"], "level_ordered_ms": res["dep"], "bit_identical_to_level_ordered": bool(np.array_equal(xs["strand"], xs["dep"])),
           "algorithmic_bytes": ssor, "effective_gbps": ssor / (res["strand"] * 1e-3) / 1e9}
    B.free()
    X.free()
    lib.mat_destroy(A)
    return out


def leg_per_rank_budget(hx, lib, torch, sync, cases, steps=60, warmup=8):
    """What ONE rank of an 8-GPU run does per CG iteration, timed alone on this GPU (verdict r5 item 1): rank 3 of 8's slab -- its real diagonal and off-diagonal
    blocks, ghost lists, the IPC put / wait / acknowledge kernels and the on-stream all-reduce kernel of a one-rank communicator, the neighbours' planes played
    by the rank's own (loop-back) -- under the launch-ahead fused CG.  Reported: ms per iteration (everything a rank does except the wire: no xGMI latency, no
    skew between ranks), the HIP-event time of every section, and the round-5 kernel sequence (separate direction and dot kernels, HipxKSP.fused = 3) beside the
    round-6 one (hipxMatMultMPICGDirectionDotBegin).  predicted_8gpu_it_s = 1 / that time: an UPPER bound for the 8-GPU rate."""
    from petsc_amd import dist as pdist
    pdist.comm_init_loopback()
    out = {}
    try:
        for name, cfg, world, rank in cases:
            e = {"what": "rank %d of %d of %s, KSPCG + %s, alone on this GPU with a loop-back ghost exchange and a one-rank IPC all-reduce" % (rank, world, cfg.metric(), cfg.pcname())}
            for label, fused in (("round5_separate_direction_and_dot_kernels", 3), ("fused_mpi_product", 1)):
                P = Problem(cfg, rank, world, None, transport="ipc", fused=fused, pipeline=1, loopback=True)
                kname = P.setup(0)
                r = timed_steps(P, steps, warmup, sync, None, torch)
                sec = r["sections"]
                kern = {"product_us": 1e3 * r["spmv_ms"], "product_launches_per_iteration": r["launches"] / float(steps)}
                for k in ("halo_ms", "allreduce_ms", "offdiag_ms", "cg_update_ms", "cg_direction_ms", "dot_fold_ms"):
                    if k in sec:
                        kern[k.replace("_ms", "_us")] = 1e3 * sec[k]
                        kern[k.replace("_ms", "_per_iteration")] = sec[k.replace("_ms", "_calls")] / float(steps)
                e[label] = {"ms_per_iteration": 1e3 * r["elapsed"] / steps, "predicted_8gpu_it_s": steps / r["elapsed"], "sections": kern, "spmv_kernel": kname.split(" ")[0],
                            "rows": P.m, "ghosts": P.nghost, "residual_norm_after": r["rnorm"]}
                P.destroy()
            e["ms_per_iteration"] = e["fused_mpi_product"]["ms_per_iteration"]
            e["predicted_8gpu_it_s"] = e["fused_mpi_product"]["predicted_8gpu_it_s"]
            e["same_residual_both_sequences"] = bool(abs(e["fused_mpi_product"]["residual_norm_after"] - e["round5_separate_direction_and_dot_kernels"]["residual_norm_after"])
                                                     <= 1e-10 * abs(e["fused_mpi_product"]["residual_norm_after"]))
            out[name] = e
    finally:
        lib.chk(hx.hipxCommFinalize())
    return out


def leg_matrix_solver(cfg, steps, warmup, sync, torch, best_ranks=None, parity_its=10, cpu_its=10, tmpdir=None, allow_ref_run=True):
    """BASELINE config 4's solver leg: KSPCG + PCJACOBI / PCSOR on a given matrix (a file, or the SPD Flan-like stand-in) on one GPU --
    iterations/s, the SpMV kernel auto picks with its roofline on the CSR bytes, and the first iterations against the REFERENCE's own
    MatLoad + KSPSolve with exact BLAS reductions (`ref_driver -f`, oracle/libexactblas.so preloaded): the committed history of that run
    (tests/golden/exact_histories.json, written by tests/golden/make_exact_golden.py from the same deterministic stand-in) when there is
    one, else -- `allow_ref_run` -- the run itself beside this one (the matrix is written as a PETSc binary file first: tens of seconds)."""
    from petsc_amd import matio
    t0 = time.perf_counter()
    P = Problem(cfg, 0, 1, None, keep_host=True)
    kname = P.setup(0)
    t_setup = time.perf_counter() - t0
    par = {"pass": None, "reference": "no committed history and no reference run (budget, or oracle/_ref not on this box)"}
    own_tmp = None
    exe = os.path.join(ROOT, "oracle", "_ref", "bin", "ref_driver")
    hgold, gsrc = (golden_history("%s_%s_%s" % (cfg.ksp, cfg.pc, cfg.golden_name)) if cfg.golden_name else (None, None))
    if parity_its and hgold is not None and len(hgold) > parity_its:
        hist = P.solve(parity_its, history=True)
        href = hgold[:len(hist)]
        rel = float((np.abs(hist - href) / np.abs(href)).max())
        par = {"pass": bool(rel <= GATE_TOL), "max_rel_diff": rel, "tolerance": GATE_TOL, "iterations": parity_its, "entries": len(hist),
               "reference": "tests/golden/exact_histories.json[%s_%s_%s] (%s: the REFERENCE's MatLoad + KSPSolve on the same matrix with exact BLAS reductions)" % (cfg.ksp, cfg.pc, cfg.golden_name, gsrc)}
    if os.path.exists(exe) and allow_ref_run and cfg.binfile is None:
        own_tmp = tmpdir or tempfile.mkdtemp(prefix="hipx_mat_")
        cfg.binfile = os.path.join(own_tmp, "matrix.bin")
        matio.write_petsc_binary(cfg.binfile, *P.host_csr)
    if os.path.exists(exe) and parity_its and par["pass"] is None and cfg.binfile:
        hist = P.solve(parity_its, history=True)
        refx = ref_driver(1, cfg.driver_args(parity_its) + ["-history"], exact=True, timeout=1800)
        if refx is not None and len(refx["history"]) == len(hist):
            href = np.array(refx["history"])
            rel = float((np.abs(hist - href) / np.abs(href)).max())
            par = {"pass": bool(rel <= GATE_TOL), "max_rel_diff": rel, "tolerance": GATE_TOL, "iterations": parity_its, "entries": len(hist),
                   "reference": "the REFERENCE's MatLoad + KSPSolve on the same file with exact BLAS reductions (ref_driver -f, oracle/libexactblas.so)"}
        else:
            par = {"pass": None, "reference": "the reference run on the file did not return a history"}
    P.host_csr = None
    r = timed_steps(P, steps, warmup, sync, None, torch)
    byts = P.spmv_bytes()
    out = {"metric": cfg.metric(), "iterations_per_s": steps / r["elapsed"], "ms_per_step": 1e3 * r["elapsed"] / steps, "steps": steps, "warmup": warmup,
           "rows": P.m, "nnz": P.nnz_local, "spmv_kernel": kname, "parity": par, "residual_norm_after": r["rnorm"], "setup_seconds": t_setup, "setup_split": dict(P.setup_times),
           "spmv_ms": r["spmv_ms"],
           "roofline_spmv": {"bound": "hbm", "kernel": kname, "avg_launch_ms": r["spmv_ms"], "launches": r["launches"], "algorithmic_bytes": byts,
                             "achieved": byts / (r["spmv_ms"] * 1e-3) / 1e9 if r["spmv_ms"] > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": byts / (r["spmv_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS if r["spmv_ms"] > 0 else 0.0, "traffic": None}}
    if cfg.pc == "sor":  # PCSOR on a matrix without row templates: the dependency-driven (level-ordered) schedule, hipx_sor.hip
        mode = C.c_int(-1)
        P.lib.chk(P.hx.hipxMatGetSORMode(P.M.A, C.byref(mode)))
        out["sor_schedule"] = {4: "plane march", 3: "inode (node-level dependency-driven)", 2: "strand", 1: "dependency-driven", 0: "levels"}.get(mode.value, str(mode.value))
        if "sor_ms" in r["sections"]:
            ssor = 2 * 12 * P.nnz_local + 40 * P.m  # two passes over a, j + 5 vector passes
            out["roofline_sor"] = {"bound": "hbm", "kernel": "one PCApply_SOR = symmetric sweep (%s schedule)" % out["sor_schedule"], "avg_call_ms": r["sections"]["sor_ms"],
                                   "calls": r["sections"].get("sor_calls"), "algorithmic_bytes": ssor, "effective_gbps": ssor / (r["sections"]["sor_ms"] * 1e-3) / 1e9,
                                   "frac": ssor / (r["sections"]["sor_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None}
    P.destroy()
    if best_ranks and cpu_its and cfg.binfile:
        out["cpu_baseline"] = cpu_baseline_for(cfg, best_ranks, cpu_its, "KSPCG + %s, MatLoad of the same file" % cfg.pcname())
    if own_tmp and not tmpdir:
        shutil.rmtree(own_tmp, ignore_errors=True)
        cfg.binfile = None
    return out


def config4_cfg(path=None):
    """config 4's matrix: the file if one is given (--matrix-file / HIPX_FLAN_FILE: MatrixMarket or PETSc binary), else the SPD stand-in."""
    from petsc_amd import matio
    path = path or os.environ.get("HIPX_FLAN_FILE")
    if path:
        cfg = MatrixCfg("matrix file %s" % os.path.basename(path), lambda: matio.read_matrix(path))
        with open(path, "rb") as f:
            if not f.read(14).startswith(b"%%MatrixMarket") and not path.endswith(".gz"):
                cfg.binfile = path  # already a PETSc binary file: the reference reads it as it is
        return cfg
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from surrogates import flan_surrogate_spd
    cfg = MatrixCfg("Flan_1565 stand-in (hexahedral elasticity pattern, 1536000 rows, 121 M nonzeros, SPD, distinct values; tests/surrogates.py)", flan_surrogate_spd)
    cfg.golden_name = "flan_standin"
    return cfg



# --------------------------------------------------------------------------------------------------- the contract line (compact)
LINE_LIMIT = 4000  # bytes: the driver keeps a few KB of stdout; round 4's 25 KB line could not be parsed


def _num(v, digits=6):
    """Numbers of the contract line: 6 significant digits are plenty there (bench_detail.json keeps everything)."""
    if isinstance(v, bool) or v is None or isinstance(v, (int, str)):
        return v
    try:
        return float("%.*g" % (digits, float(v)))
    except (TypeError, ValueError):
        return v


def _short(s, n):
    s = "" if s is None else str(s)
    return s if len(s) <= n else s[:n - 1] + "~"


def _kshort(name):
    """'spmv_march2_kernel (CSR MatMult, row templates: ...) + the CG ...' -> 'spmv_march2_kernel+cgdir'"""
    if not name:
        return name
    base = str(name).split(" ")[0]
    return base + ("+cgdir" if "prologue" in str(name) else "")


def _leg_frac(leg):
    """the dominant kernel's roofline fraction of an other_configs leg: SOR when the leg has one (it dominates), else the product"""
    for key in ("roofline_sor", "roofline_longrow", "roofline_spmv", "roofline_cg_update"):
        r = leg.get(key)
        if isinstance(r, dict):
            f = r.get("frac", r.get("frac_counter_bytes"))
            if f is not None:
                return key.replace("roofline_", ""), f
            if r.get("frac_algorithmic_bytes") is not None:
                return key.replace("roofline_", "") + " (algorithmic bytes)", r["frac_algorithmic_bytes"]
    return None, None


def compact_line(out):
    """The ONE line the driver parses: the contract's fields, `parity_gate`, `roofline`, `cpu_baseline` and one number set per leg --
    everything else (notes, per-kernel counter detail, per-rank tables, samples) is in bench_detail.json, written beside it."""
    c = {}
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        c[k] = _num(out.get(k), 9)
    cfg = out.get("config") or {}
    c["config"] = {k: (_short(v, 170) if isinstance(v, str) else (v if k == "residual_norm_after" else _num(v))) for k, v in cfg.items()
                   if k in ("workload", "global_rows", "parallelism", "transport", "fused", "pipeline", "reduction_mode", "residual_norm_after")}
    if "error" in out:
        c["error"] = _short(out["error"], 300)
    if "scaling_legs_error" in out:
        c["scaling_legs_error"] = _short(out["scaling_legs_error"], 200)
    c["ungated"] = out.get("ungated")
    g = out.get("parity_gate") or {}
    c["parity_gate"] = {k: _num(g.get(k)) for k in ("pass", "max_rel_diff", "tolerance", "iterations", "gated_reduction_mode") if k in g}
    r = out.get("roofline")
    if r:
        rr = {"bound": r.get("bound"), "kernel": _kshort(r.get("kernel"))}
        for k in ("achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes", "avg_launch_ms", "effective_gbps", "iteration_frac", "dominant_by_time"):
            if k in r:
                rr[k] = _num(r[k])
        if r.get("by_kernel"):  # [name, us per launch, launches per iteration, HBM bytes per launch, frac]
            rr["by_kernel"] = [[_short(k.get("kernel") or k.get("match"), 60), _num(k.get("avg_launch_us"), 4), _num(k.get("launches_per_iteration"), 3), k.get("traffic"), _num(k.get("frac"), 4)]
                               for k in r["by_kernel"] if k.get("launches_per_iteration", 0) >= 0.5]
        if isinstance(r.get("general"), dict):  # SURVEY 8(d)'s literal figure: algorithmic CSR bytes / launch time of the kernel that STREAMS the values (verdict r5)
            rr["frac_csr_bytes"] = _num(r["general"].get("frac"), 4)
        for key in ("general", "unstructured"):
            if isinstance(r.get(key), dict):
                e = r[key]
                rr[key] = {"kernel": _kshort(e.get("kernel")), "frac": _num(e.get("frac"), 4), "frac_counter_bytes": _num(e.get("frac_counter_bytes"), 4),
                           "avg_launch_ms": _num(e.get("avg_launch_ms"), 4), "it_s": _num(e.get("iterations_per_s"), 5)}
        c["roofline"] = rr
    b = out.get("cpu_baseline")
    if b:
        c["cpu_baseline"] = {"value": _num(b.get("value")), "unit": b.get("unit"), "cores": b.get("cores"), "kind": b.get("kind"), "sample": _short(b.get("sample"), 200)}
        if b.get("value_1core") is not None:
            c["cpu_baseline"]["value_1core"] = _num(b["value_1core"])
    else:
        c["cpu_baseline"] = None
    if out.get("plugin"):
        c["plugin_it_s"] = {k: _num(v.get("iterations_per_s"), 5) for k, v in out["plugin"].items() if isinstance(v, dict)}
        c["plugin_second_solve_it_s"] = {k: _num(v.get("second_solve_iterations_per_s"), 5) for k, v in out["plugin"].items() if isinstance(v, dict) and v.get("second_solve_iterations_per_s")}
    oc = out.get("other_configs")
    if oc:
        legs = {}
        for name, leg in oc.items():
            if not isinstance(leg, dict):
                continue
            if "error" in leg or "skipped" in leg:
                legs[name] = {"error": _short(leg["error"], 80)} if "error" in leg else {"skipped": _short(leg["skipped"], 40)}
                continue
            if name == "per_rank_budget":  # per case: [ms per iteration of one rank alone (round-6 sequence), the round-5 sequence's, predicted 1 -> 8 GPU scaling]
                legs[name] = {k: [_num(v.get("ms_per_iteration"), 4), _num((v.get("round5_separate_direction_and_dot_kernels") or {}).get("ms_per_iteration"), 4), _num(v.get("predicted_scaling_1_to_8"), 3)]
                              for k, v in leg.items() if isinstance(v, dict)}
                continue
            e = {}
            if leg.get("iterations_per_s") is not None:
                e["it_s"] = _num(leg["iterations_per_s"], 5)
            par = leg.get("parity")
            if isinstance(par, dict):
                e["parity"] = par.get("pass")
                if par.get("max_rel_diff") is not None:
                    e["rel"] = _num(par["max_rel_diff"], 3)
            elif "sampled_rows_bit_identical" in leg:
                e["parity"] = leg["sampled_rows_bit_identical"]
            elif "bit_identical_to_level_ordered" in leg:
                e["parity"] = leg["bit_identical_to_level_ordered"]
            which, f = _leg_frac(leg)
            if f is not None:
                e["frac"], e["of"] = _num(f, 4), which
            for k in ("spmv_ms", "strand_streamed_coefficients_ms"):
                if leg.get(k) is not None:
                    e["spmv_ms" if k == "spmv_ms" else "sor_ms"] = _num(leg[k], 4)
            if isinstance(leg.get("roofline_sor"), dict) and leg["roofline_sor"].get("avg_call_ms") is not None:
                e["sor_ms"] = _num(leg["roofline_sor"]["avg_call_ms"], 4)
            if isinstance(leg.get("cpu_baseline"), dict):
                e["cpu_it_s"] = _num(leg["cpu_baseline"].get("value"), 4)
            legs[name] = e
        c["other_configs"] = legs
    mg = out.get("multi_gpu")
    if mg:
        c["multi_gpu"] = {"ranks": mg.get("ranks"), "distinct_devices": mg.get("distinct_devices"), "launcher": _short(mg.get("launcher"), 50),
                          "transports": {t: {k: (_num(v.get(k), 5) if k != "parity" else (v.get(k) or {}).get("pass")) for k in ("probe_ok", "iterations_per_s", "parity", "comm_nranks") if k in v}
                                         for t, v in (mg.get("transports") or {}).items()}}
        if mg.get("note"):
            c["multi_gpu"]["note"] = _short(mg["note"], 120)
    pr = out.get("per_rank")
    if pr:  # per rank: [rows, ghosts, product ms, ghost-exchange ms, all-reduce ms]
        c["per_rank"] = [[p_.get("rows"), p_.get("ghosts"), _num(p_.get("spmv_ms"), 4), _num(p_.get("halo_ms"), 4), _num(p_.get("allreduce_ms"), 4)] for p_ in pr[:8]]
    for k in ("wall_s", "budget_s", "detail"):
        if k in out:
            c[k] = _num(out[k], 4)
    # never longer than the driver can read: drop the optional groups, least important first
    def fits():
        return len(json.dumps(c)) <= LINE_LIMIT
    for victim in ("per_rank", "plugin_second_solve_it_s", "plugin_it_s", "multi_gpu"):
        if fits():
            break
        c.pop(victim, None)
    if not fits() and "other_configs" in c:  # one number per leg ...
        c["other_configs"] = {k: (v.get("it_s", v.get("frac")) if isinstance(v, dict) else v) for k, v in c["other_configs"].items()}
        while not fits() and c["other_configs"]:  # ... and, if a run has more legs than the line can name, the first ones only (all of them are in bench_detail.json)
            c["other_configs"].popitem()
            c["other_configs_truncated"] = True
    if not fits() and "roofline" in c:
        c["roofline"].pop("by_kernel", None)
    return c


def emit(out, t_start=None):
    """Write everything to bench_detail.json (repo root, and gpurun_out/ when that exists) and print the compact contract line LAST."""
    if t_start is not None:
        out["wall_s"] = time.time() - t_start
    out["detail"] = "bench_detail.json"
    text = json.dumps(out)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        try:
            if os.path.isdir(d):
                with open(os.path.join(d, "bench_detail.json"), "w") as f:
                    f.write(text + "\n")
        except OSError:
            pass
    line = json.dumps(compact_line(out))
    sys.stdout.flush()
    sys.stdout.write(line + "\n")
    sys.stdout.flush()
    return line


# ---------------------------------------------------------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--grid", dest="n", type=int, default=256, help="grid points per side (not --n: torchrun would read it as an abbreviation of its own options)")
    ap.add_argument("--stencil", type=int, default=7, choices=[5, 7, 27], help="5: the 2-D operator of ex2.c (BASELINE config 1) on a grid x grid mesh (--grid 4096: 16.8 M rows)")
    ap.add_argument("--ksp", default="cg", choices=["cg", "gmres", "pipecg", "groppcg"])
    ap.add_argument("--pc", default="jacobi", choices=["jacobi", "sor", "none"])
    ap.add_argument("--transport", default=os.environ.get("HIPX_TRANSPORT", "auto"), choices=["auto", "rccl", "ipc"],
                    help="multi-GPU data path: RCCL send/recv + all-reduce over xGMI, or IPC peer stores (also when ranks share a GPU); auto: probe both, time both, report the faster as `value`")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"], help="weak: every GPU owns grid x grid x grid/8 rows (config 5: --grid 1024)")
    ap.add_argument("--fused", type=int, default=1, help="1 (default): fused SpMV+dot and AXPY+AXPY+PCJACOBI+norm+dot kernels -- same arithmetic and order per element, fewer HBM passes; 0: one kernel per reference Vec/Mat call (cg.c:249-344)")
    ap.add_argument("--pipeline", type=int, default=1, help="1 (default): launch-ahead fused CG (iteration i+1 enqueued before the host has seen iteration i's sums; device-resident scalars); 0: host waits between kernels; "
                    "2: several ranks on the host-synchronised loop; 3: single-reduction CG (cg.c:364-534: ONE reduction stage -- a 24-byte all-reduce -- per iteration instead of two), host-synchronised; 4: its launch-ahead form (scalars formed on the device)")
    ap.add_argument("--variant", type=int, default=0, help="SpMV kernel variant (include/hipx.h hipxMatSetSpMVVariant): 0 auto")
    ap.add_argument("--general-variant", type=int, default=29, help="kernel of the roofline_general leg: what a matrix with arbitrary values on this pattern gets (29: pattern templates + streamed values, "
                    "the auto choice for short rows on <= 256 row patterns; 23: packed 16-bit columns, what an unstructured matrix gets)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 PMC passes (roofline.traffic then comes from profiles/spmv_traffic.json)")
    ap.add_argument("--no-plugin", action="store_true")
    ap.add_argument("--no-general", action="store_true")
    ap.add_argument("--matrix-file", default=None, help="N = 1: solve this matrix (MatrixMarket coordinate file, optionally .gz, or PETSc binary Mat) with KSPCG + PCJACOBI "
                                                        "instead of a Poisson operator: BASELINE config 4 with the real SuiteSparse file")
    ap.add_argument("--no-other", action="store_true", help="skip the other_configs legs (configs 3/4/5 on one GPU; the scaling legs on N GPUs)")
    ap.add_argument("--quick", action="store_true", help="the timed legs only: no plugin / PMC / CPU-baseline / general-kernel / other-config legs")
    ap.add_argument("--only-legs", default="", help="N > 1: run only the scaling legs whose name contains one of these comma-separated strings (tests)")
    ap.add_argument("--budget-s", type=float, default=float(os.environ.get("HIPX_BENCH_BUDGET_S", "175")),
                    help="wall-clock budget of the whole run (default 175 s): the headline (parity gate, timed steps, counter pass, CPU baseline) always runs; the optional legs (plugin rows, "
                         "other_configs, their counter passes and CPU baselines) run in order of importance while their estimated cost still fits, the rest are reported as skipped")
    ap.add_argument("--parity-its", type=int, default=GATE_ITS, help="N > 1: entries of the committed exact-reduction history the headline leg is gated on (default 24; 35 covers a GMRES(30) restart)")
    ap.add_argument("--full", action="store_true", help="no budget: every leg, every counter pass, every CPU-baseline rank count (several minutes)")
    ap.add_argument("--suite", default=None, help=argparse.SUPPRESS)  # internal workloads of the counter passes (suite_mode)
    ap.add_argument("--suite-variants", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--suite-dims", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--suite-perturb", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--suite-no-dconst", type=int, default=0, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.suite:
        return suite_mode(args)
    t_start = time.time()
    deadline = float("inf") if (args.full or args.budget_s <= 0) else t_start + args.budget_s
    timeline = []

    def room(est):
        """does a leg of estimated cost `est` seconds still fit the budget?"""
        return time.time() + est <= deadline

    class phase:  # with phase("name"): ...  -> timeline entry
        def __init__(self, name):
            self.name = name

        def __enter__(self):
            self.t = time.time()

        def __exit__(self, *a):
            timeline.append([self.name, round(time.time() - self.t, 2)])
    if args.quick:
        args.no_cpu_baseline = args.no_traffic = args.no_plugin = args.no_general = args.no_other = True
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    ndev = torch.cuda.device_count()
    share_all = os.environ.get("HIPX_ALL_RANKS_DEVICE0") == "1"
    dev = 0 if share_all else (local_rank % max(ndev, 1))
    shared = share_all or world > max(ndev, 1)  # fewer GPUs than ranks: ranks share devices, RCCL refuses that ("Duplicate GPU")
    dist = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        # gloo carries the set-up exchanges and the timing barrier only; the data path (ghost values, dot/norm sums) is libhipx's own
        dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=30))

    from petsc_amd import _lib
    hx = _lib.init(dev)
    _, ks = _lib.load()
    n = args.n
    dims = (n, n, n) if args.scaling == "strong" else (n, n, (n // 8) * world)  # weak: config 5 = n x n x n/8 rows per GPU
    if args.stencil == 5:
        dims = (n, n, 1)
    head = Cfg(args.stencil, dims, args.ksp, args.pc, args.scaling)
    N = head.N

    def sync():
        _lib.chk(hx.hipxDeviceSynchronize())
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    if world > 1:
        return main_multi(args, head, rank, world, dev, shared, dist, torch, hx, sync, t_start)

    # =========================================================================================================== one GPU
    if args.matrix_file:  # BASELINE config 4 with a supplied file (or `--matrix-file standin`: the documented stand-in): the whole line is that solve
        cfg4 = config4_cfg(None if args.matrix_file == "standin" else args.matrix_file)
        cfg4.pc = args.pc
        ranks4 = None if args.no_cpu_baseline else max(2, physical_cores() // 4)
        r4 = leg_matrix_solver(cfg4, args.steps, args.warmup, sync, torch, best_ranks=ranks4, parity_its=0 if args.no_cpu_baseline else 10)
        rf = dict(r4["roofline_spmv"], kernel=r4["spmv_kernel"], basis="algorithmic CSR bytes / launch time (no counter pass on a file matrix)")
        emit({"metric": cfg4.metric(), "value": r4["iterations_per_s"] if r4["parity"].get("pass") is not False else None, "unit": "iterations/s", "n_gpus": 1,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": r4["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                          "dtype": "f64", "data": "file: %s" % args.matrix_file,
                          "config": {"workload": "%s (N=%d rows, nnz=%d), KSPCG + %s, b = A*1, x0 = 0" % (cfg4.name, r4["rows"], r4["nnz"], cfg4.pcname()), "global_rows": r4["rows"], "parallelism": "rows1"},
                          "ungated": r4["parity"].get("pass") is None, "parity_gate": r4["parity"], "roofline": rf, "cpu_baseline": r4.get("cpu_baseline")}, t_start)
        return
    P = Problem(head, 0, 1, None, fused=args.fused, pipeline=args.pipeline, keep_host=True)
    kname = P.setup(args.variant)
    # ---- parity gate: this configuration's first iterations against the reference's own KSPSolve run
    # beside it with exact BLAS reductions (the reference minus its BLAS's rounding noise)
    gate = {"iterations": GATE_ITS, "tolerance": GATE_TOL, "max_rel_diff": None, "pass": None, "reference": None,
            "criterion": "every entry of the GPU residual history within 1e-12 (relative) of the REFERENCE's own KSPSolve (oracle/_ref: its cg.c / gmres.c, MatMult_SeqAIJ, Vec loops) run on this host "
                         "with oracle/libexactblas.so preloaded: ddot / dgemv in twice the working precision (Dot2), i.e. the reference without its BLAS's summation-order noise; the restated oracle's "
                         "exact mode and the committed golden history are bit-identical to that run (tests/test_oracle_exact.py) and are reported beside it"}
    ref1 = refx = None
    gate_on = not args.no_cpu_baseline and N <= 2 ** 25
    if gate_on:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(2) as ex:  # the reference's two runs (MKL / exact BLAS reductions) on two host cores, side by side
            f1 = ex.submit(ref_driver, 1, head.driver_args(GATE_ITS) + ["-history"])
            fx = ex.submit(ref_driver, 1, head.driver_args(GATE_ITS) + ["-history"], False, 900, False, True)
            hist = P.solve(GATE_ITS, history=True)
            ref1, refx = f1.result(), fx.result()
        tol = GATE_TOL
        gate["tolerance"] = tol
        hgold, gsrc = golden_history(head.golden_key() + ("_np1" if head.ksp == "gmres" else ""))
        hyard, what = None, None
        if refx is not None and len(refx["history"]) == len(hist):
            hyard, what = np.array(refx["history"]), "the REFERENCE's KSPSolve with exact BLAS reductions, run on this host (%d history entries)" % len(hist)
        elif hgold is not None and len(hgold) >= len(hist):
            hyard, what = hgold[:len(hist)], "tests/golden/exact_histories.json (%s; oracle/_ref not on this box)" % gsrc
        else:
            try:  # last resort: the restated oracle's exact mode
                sys.path.insert(0, os.path.join(ROOT, "oracle"))
                import oracle as orc
                ai, aj, aa = P.host_csr
                hyard = orc.ksp_solve(head.ksp, ai, aj, aa, P.B.get(), pc=head.pc, rtol=1e-50, max_it=GATE_ITS, exact=True)[3]
                what = "oracle restatement, exact mode (oracle/_ref and the golden file are not available)"
            except Exception as e:  # noqa: BLE001
                gate["reference"] = "no yardstick available: %s" % e
        if hyard is not None and len(hyard) == len(hist):
            rel = float((np.abs(hist - hyard) / np.abs(hyard)).max())
            gate.update({"max_rel_diff": rel, "pass": bool(rel <= tol), "reference": what})
            if hgold is not None and len(hgold) >= len(hist):
                gate["yardstick_vs_committed_golden_max_abs_diff"] = float(np.abs(hyard - hgold[:len(hist)]).max())
            if ref1 is not None and len(ref1["history"]) == len(hist):
                href = np.array(ref1["history"])
                gate["gpu_vs_reference_mkl_max_rel_diff"] = float((np.abs(hist - href) / np.abs(href)).max())
                gate["reference_mkl_vs_reference_exact_max_rel_diff"] = float((np.abs(href - hyard) / np.abs(hyard)).max())
        elif hyard is not None:
            gate.update({"pass": False, "reference": "history lengths differ: %d vs %d" % (len(hist), len(hyard))})
    else:
        gate["reference"] = "not run: --no-cpu-baseline / --quick / size"
    P.host_csr = None

    # ---- the timed legs
    r = timed_steps(P, args.steps, args.warmup, sync, None, torch)
    elapsed, spmv_ms, launches, rnorm = r["elapsed"], r["spmv_ms"], r["launches"], r["rnorm"]
    spmv_bytes = P.spmv_bytes()
    sec = r["sections"]
    # every kernel of the timed iteration (HIP events around each launch, second pass of the same K steps)
    by_kernel = [{"role": "product (+ CG direction update as its prologue when the kernel is spmv_march2_kernel<..., true> and the solver is the fused CG)", "match": kname.split(" ")[0],
                  "avg_launch_us": 1e3 * spmv_ms, "launches_per_iteration": launches / float(args.steps)}]
    for key, role, match in (("cg_update_ms", "fused CG update: r -= a w, z = r d, z.z, z.r (cg.c:306-309,344)", "cg_fused_"),
                             ("cg_direction_ms", "CG direction: p = z + b p, x += a p (cg.c:249,305) as its own kernel", "cg_aypx_axpy_kernel"),
                             ("dot_fold_ms", "fold of the product's dot partials (cg.c:258)", "sum_kernel")):
        if key in sec:
            by_kernel.append({"role": role, "match": match, "avg_launch_us": 1e3 * sec[key], "launches_per_iteration": sec[key.replace("_ms", "_calls")] / float(args.steps)})
    extra_lines = {}
    if not args.no_general and head.ksp == "cg":
        for key, variant, note in (("general", args.general_variant, "pattern templates + streamed values: what a matrix with arbitrary VALUES on this stencil pattern gets (variable-coefficient operators)"),
                                   ("unstructured", 23, "packed 16-bit columns, row-parallel gather: what an UNSTRUCTURED matrix with short rows gets")):
            gname = P.setup(variant, no_dconst=True)
            g = timed_steps(P, args.steps, args.warmup, sync, None, torch)
            e = {"bound": "hbm", "kernel": gname, "avg_launch_ms": g["spmv_ms"], "launches": g["launches"], "algorithmic_bytes": spmv_bytes,
                 "achieved": spmv_bytes / (g["spmv_ms"] * 1e-3) / 1e9 if g["spmv_ms"] > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s", "iterations_per_s": args.steps / g["elapsed"],
                 "basis": "algorithmic CSR bytes (12 nnz + 4 (N + 1) + 16 N, SURVEY 8(d)) / launch time; frac_counter_bytes: HBM bytes moved (PMC) / launch time",
                 "traffic": None, "variant": variant, "note": "the same solver with --variant %d and the constant-Jacobi-diagonal shortcut off: %s" % (variant, note)}
            e["frac"] = e["achieved"] / HBM_PEAK_GBS
            extra_lines[key] = e
        P.setup(args.variant)
    nnz_local = P.nnz_local
    setup_split = dict(P.setup_times)
    P.destroy()

    achieved_alg = spmv_bytes / (spmv_ms * 1e-3) / 1e9 if spmv_ms > 0 else 0.0
    value = args.steps / elapsed if gate["pass"] is not False else None
    fused_product = "march2" in kname and args.fused and args.pipeline and head.pc in ("jacobi", "none") and "cg_direction_ms" not in sec
    out = {
        "metric": head.metric(), "value": value, "unit": "iterations/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s %d-pt Poisson %dx%dx%d (N=%d rows, nnz=%d local), KSP%s + %s, b = A*1, x0 = 0; rows split over 1 rank(s)"
                               % ("2-D" if args.stencil == 5 else "3-D", args.stencil, dims[0], dims[1], dims[2], N, nnz_local, args.ksp.upper(), head.pcname()),
                   "global_rows": N, "parallelism": "rows1", "transport": None, "fused": args.fused, "pipeline": args.pipeline, "spmv_variant": args.variant,
                   "residual_norm_after": rnorm, "reduction_mode": os.environ.get("HIPX_REDUCTIONS", "fast")},
        "ungated": gate["pass"] is None,
        "parity_gate": gate,
        "setup_split": setup_split,
        "roofline": {"bound": "hbm", "kernel": kname + (" + the CG direction update as its prologue (hipxMatMultCGDirectionDotBegin)" if fused_product else ""),
                     "achieved": achieved_alg, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_alg / HBM_PEAK_GBS,
                     "basis": "algorithmic CSR bytes / launch time (no counter bytes available)", "traffic": None, "traffic_source": None,
                     "launches": launches, "avg_launch_ms": spmv_ms, "algorithmic_bytes": spmv_bytes + (48 * N if fused_product else 0),
                     "note": "achieved / frac: HBM bytes the dominant kernel really moves (rocprofv3 PMC passes run from inside this script) / its HIP-event launch time; effective_gbps = the "
                             "algorithmic bytes of the operations it replaces (CSR SpMV 12 nnz + 4 (N+1) + 16 N, SURVEY 8(d); + 48 N for the direction update when that is its prologue) / launch "
                             "time: the row-template format moves far fewer bytes than CSR, so effective_gbps exceeds the HBM peak",
                     "by_kernel": by_kernel},
    }
    rf = out["roofline"]
    rf["effective_gbps"] = rf["algorithmic_bytes"] / (spmv_ms * 1e-3) / 1e9 if spmv_ms > 0 else 0.0
    rf["effective_frac_of_peak"] = rf["effective_gbps"] / HBM_PEAK_GBS
    rf.update(extra_lines)
    tot_us = sum(k["avg_launch_us"] * k["launches_per_iteration"] for k in by_kernel)
    rf["iteration"] = {"us_in_kernels": tot_us, "us_per_step": 1e3 * out["ms_per_step"], "kernel_time_over_step_time": tot_us / (1e3 * out["ms_per_step"]) if out["ms_per_step"] else None}
    dom = max(by_kernel, key=lambda k: k["avg_launch_us"] * k["launches_per_iteration"])
    rf["dominant_by_time"] = dom["match"]

    # ---- the drop-in itself (reference executable + plugin): GPU legs, before the counter passes.  Each row is one run of the reference's
    # executable (its own MatSetValues assembly on the host + the solve): ~10 s apiece, so the default budget takes the two rows north_star
    # is about and leaves the rest to --full
    if not args.no_plugin and head.cube and args.ksp == "cg" and args.pc == "jacobi" and N <= 2 ** 25:
        plug = {}
        rows = [("reference KSPSolve_CG over hipx types", "cg", 1), ("-ksp_type cghipx (fused kernels under PETSc's monitors / convergence test)", "cghipx", 1),
                # the reference's reduction-fused / pipelined callers, unmodified, over the hipx types
                # the same with -pc_type jacobihipx: PCSetUp_Jacobi's host pass over the diagonal (jacobi.c:252-262: VecGetArray + loop, inside the first KSPSolve) becomes a device kernel
                ("reference KSPSolve_CG over hipx types, -pc_type jacobihipx", "cg+jacobihipx", 0),
                ("reference KSPSolve_PIPECG over hipx types (pipecg.c; update block = one batch kernel)", "pipecg", 0),
                ("-ksp_type pipecghipx (one fused update kernel + one product per iteration, launch-ahead)", "pipecghipx", 0),
                ("reference KSPSolve_GROPPCG over hipx types (groppcg.c; update blocks = two batch kernels)", "groppcg", 0),
                ("-ksp_type groppcghipx (two fused passes + one product per iteration, launch-ahead)", "groppcghipx", 0)]
        with phase("plugin rows"):
            for label, ksp, always in rows:
                if not (always and room(60)) and not (args.full or room(95)):
                    plug[ksp] = {"what": label, "skipped": "budget"}
                    continue
                a = [x if x != "cg" else ksp.split("+")[0] for x in head.driver_args(400)] + ["-resolve"]
                if "+" in ksp:  # "cg+jacobihipx": the plugin's PC type (its set-up runs on the device: no host pass over the diagonal inside the first KSPSolve)
                    a = [x if x != "jacobi" else ksp.split("+")[1] for x in a]
                rr = ref_driver(1, a, plugin=True)
                plug[ksp] = {"what": label, "iterations_per_s": (rr["its"] / rr["seconds"]) if rr else None, "iterations": rr["its"] if rr else None,
                             "KSPSolve_seconds": rr["seconds"] if rr else None}
                if rr and rr.get("second_seconds"):  # the second KSPSolve of the same process: formats and buffers exist
                    plug[ksp]["second_solve_iterations_per_s"] = rr["second_its"] / rr["second_seconds"]
            # Chebyshev as a smoother (first kind, no norms, bounds given: 7-pt Poisson + Jacobi has its spectrum in (0, 2)): the
            # reference's KSPSolve_Chebyshev over the hipx types (4 kernels per iteration) and -ksp_type chebyshevhipx (SpMV + one fused kernel)
            for label, ksp in (("reference KSPSolve_Chebyshev over hipx types (smoother configuration: -ksp_norm_type none)", "chebyshev"),
                               ("-ksp_type chebyshevhipx (SpMV + one fused kernel per iteration, bit-identical solution)", "chebyshevhipx")):
                if not args.full:
                    plug[ksp] = {"what": label, "skipped": "budget (--full runs it)"}
                    continue
                a = ["-stencil", str(head.stencil), "-n", str(head.dims[0]), "-ksp_type", ksp, "-pc_type", "jacobi", "-ksp_norm_type", "none", "-ksp_max_it", "400",
                     "-ksp_chebyshev_eigenvalues", "0.1,2.0"]
                rr = ref_driver(1, a, plugin=True)
                plug[ksp] = {"what": label, "iterations_per_s": (rr["its"] / rr["seconds"]) if rr else None, "iterations": rr["its"] if rr else None,
                             "KSPSolve_seconds": rr["seconds"] if rr else None, "error_norm": rr["error"] if rr else None}
        out["plugin"] = plug
    else:
        out["plugin"] = None

    # ---- BASELINE configs 3 / 4 / 5 on this GPU: the timed part (their counter passes and CPU baselines follow below), most important first;
    # a leg runs when its estimated cost (seconds, measured on the round-4/5 boxes) still fits the budget with `reserve` left for the headline's
    # own counter pass + CPU baseline
    other, leg_cfgs = {}, {}
    reserve = 32.0
    cfg4 = cfg4s = None
    tmp4 = None
    if not args.no_other:
        tmp4 = tempfile.mkdtemp(prefix="hipx_mat_")
        legs = [("config3_solver_gmres30_sor_27pt_256", Cfg(27, (256, 256, 256), "gmres", "sor", golden="gmres_sor_27pt_256"), 60, 5, 35, 10, 8),
                ("config5_share_cg_none_7pt_1024x1024x128", Cfg(7, (1024, 1024, 128), "cg", "none", scaling="weak"), 50, 5, 12, 10, 9),
                # the 1-GPU point of north_star's >= 6x target (27-pt 512^3: 3.6e9 nonzeros, 64-bit row offsets, 46 GB of CSR in HBM)
                ("cg_jacobi_27pt_512_strong", Cfg(27, (512, 512, 512), "cg", "jacobi"), 30, 3, 16, 0, 24),
                # north_star: "5-/7-/27-point Poisson stencils reported" -- BASELINE config 1's operator (ex2.c:70-94) at HBM size: 4096 x 4096 = the headline's row count
                ("cg_jacobi_5pt_4096x4096", Cfg(5, (4096, 4096, 1), "cg", "jacobi"), 100, 10, 16, 10, 7),
                # KSPPIPECG (pipecg.c) on the headline's system through the host layer's launch-ahead loop (one fused update kernel + one product per iteration;
                # a timed "step" here is one pass of a complete K-iteration solve, set-up included)
                ("pipecg_jacobi_7pt_256", Cfg(7, (256, 256, 256), "pipecg", "jacobi"), 100, 10, 24, 0, 7),
                # ... and KSPGROPPCG (groppcg.c) the same way: two fused passes + one product per iteration
                ("groppcg_jacobi_7pt_256", Cfg(7, (256, 256, 256), "groppcg", "jacobi"), 100, 10, 24, 0, 6)]
        for name, cfg, st, wu, pits, cpu_its, est in legs:
            if not room(est + reserve):
                other[name] = {"skipped": "budget"}
                continue
            try:
                with phase(name):
                    res, (nnz_l, m_l, wide_l) = run_leg(cfg, 0, 1, None, torch, None, st, wu, sync, parity_its=pits)
                pr = res.pop("per_rank")[0]
                res["spmv_ms"] = pr["spmv_ms"]
                byts = 12 * nnz_l + (8 if wide_l else 4) * (m_l + 1) + 16 * m_l
                res["roofline_spmv"] = {"bound": "hbm", "kernel": res["spmv_kernel"], "avg_launch_ms": pr["spmv_ms"], "algorithmic_bytes": byts, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "effective_gbps": byts / (pr["spmv_ms"] * 1e-3) / 1e9 if pr["spmv_ms"] > 0 else None, "traffic": None}
                for k2 in ("cg_update_ms", "cg_direction_ms", "dot_fold_ms"):
                    if k2 in pr:
                        res[k2] = pr[k2]
                if cfg.pc == "sor" and "sor_ms" in pr:
                    ssor = 2 * 12 * nnz_l + 40 * m_l  # two passes over a, j + 5 vector passes
                    res["roofline_sor"] = {"bound": "hbm", "kernel": "%s forward + backward (one PCApply_SOR = symmetric sweep)" % ("sor_box_kernel" if res.get("sor_schedule") == "plane march" else "sor_strand_kernel"),
                                           "avg_call_ms": pr["sor_ms"], "calls": pr.get("sor_calls"),
                                           "algorithmic_bytes": ssor, "effective_gbps": ssor / (pr["sor_ms"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None,
                                           "frac_algorithmic_bytes": ssor / (pr["sor_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
                other[name] = res
                leg_cfgs[name] = (cfg, cpu_its, m_l)
            except Exception as e:  # noqa: BLE001
                other[name] = {"error": str(e)[:400]}
        # config 4 (the stand-in, or HIPX_FLAN_FILE): the matrix is generated once and shared by its legs; parity against the committed
        # history of the REFERENCE's MatLoad + KSPSolve + exact BLAS on the same (deterministic) matrix when there is one, else that run itself
        for name, pc, st, wu, pits, est in (("config4_solver_cg_jacobi", "jacobi", 100, 10, 10, 22), ("config4_solver_cg_sor", "sor", 30, 3, 5, 24)):
            if not room(est + reserve):
                other[name] = {"skipped": "budget"}
                continue
            try:
                with phase(name):
                    c4 = config4_cfg()
                    c4.pc = pc
                    if cfg4 is not None:
                        c4._cache, c4.binfile = cfg4._cache, cfg4.binfile
                    other[name] = leg_matrix_solver(c4, st, wu, sync, torch, best_ranks=None, parity_its=pits, tmpdir=tmp4, allow_ref_run=args.full or room(60 + reserve))
                    if pc == "jacobi":
                        cfg4 = c4
                    else:
                        cfg4s = c4
            except Exception as e:  # noqa: BLE001
                other[name] = {"error": str(e)[:400]}
        for name, est, fn in (("config3_sor_arbitrary_values_27pt_256", 12, lambda: leg_sor_arbitrary_values(hx, _lib, ks)),
                              ("config4_surrogate_spmv", 20, lambda: leg_surrogate_spmv(hx, _lib))):
            if not room(est + reserve):
                other[name] = {"skipped": "budget"}
                continue
            try:
                with phase(name):
                    other[name] = fn()
            except Exception as e:  # noqa: BLE001
                other[name] = {"error": str(e)[:400]}
        if room(14 + reserve):
            try:
                with phase("per_rank_budget"):
                    prb = leg_per_rank_budget(hx, _lib, torch, sync, [("27pt_512_rank3of8", Cfg(27, (512, 512, 512), "cg", "jacobi"), 8, 3), ("7pt_256_rank3of8", Cfg(7, (256, 256, 256), "cg", "jacobi"), 8, 3)])
                one = other.get("cg_jacobi_27pt_512_strong", {}).get("iterations_per_s")
                if one and "27pt_512_rank3of8" in prb:
                    prb["27pt_512_rank3of8"]["one_gpu_it_s_this_run"] = one
                    prb["27pt_512_rank3of8"]["predicted_scaling_1_to_8"] = prb["27pt_512_rank3of8"]["predicted_8gpu_it_s"] / one
                if value and "7pt_256_rank3of8" in prb:
                    prb["7pt_256_rank3of8"]["one_gpu_it_s_this_run"] = value
                    prb["7pt_256_rank3of8"]["predicted_scaling_1_to_8"] = prb["7pt_256_rank3of8"]["predicted_8gpu_it_s"] / value
                other["per_rank_budget"] = prb
            except Exception as e:  # noqa: BLE001
                other["per_rank_budget"] = {"error": str(e)[:400]}
        else:
            other["per_rank_budget"] = {"skipped": "budget"}
        for c4 in (cfg4, cfg4s):
            if c4 is not None:
                c4._cache = None
        out["other_configs"] = other
    _lib.chk(hx.hipxDeviceSynchronize())

    # ---- counter passes (GPU, rocprofv3 child processes: the device must be theirs alone) in a thread, the CPU baselines (host cores only) beside them
    pmc = {}
    ran = lambda nm: nm in other and "error" not in other[nm] and "skipped" not in other[nm]  # noqa: E731

    def counter_passes():
        src = "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes launched by this run (bench.py --suite %s), gfx950 correction read = 2 x FETCH_SIZE"
        todo = []
        if head.cube:
            todo.append(("cg", ["--suite", "cg", "--grid", str(n), "--stencil", str(args.stencil), "--pc", args.pc, "--variant", str(args.variant)], "headline_cg", src % "cg", 0))
        if not args.no_other:
            if ran("config3_solver_gmres30_sor_27pt_256"):
                todo.append(("sor27", ["--suite", "sor", "--grid", "256", "--stencil", "27"], "27pt_256_sor", src % "sor (27-pt 256^3)", 14))
        if head.cube and extra_lines:
            todo.append(("spmv", ["--suite", "spmv", "--grid", str(n), "--stencil", str(args.stencil), "--suite-variants", ",".join(str(e["variant"]) for e in extra_lines.values())],
                         "headline_general", src % "spmv", 14))
        if not args.no_other:
            if ran("config3_solver_gmres30_sor_27pt_256") or ran("cg_jacobi_27pt_512_strong"):
                todo.append(("spmv27", ["--suite", "spmv", "--grid", "256", "--stencil", "27", "--suite-variants", "0"], "27pt_256_spmv", src % "spmv (27-pt 256^3)", 14))
            if ran("cg_jacobi_27pt_512_strong"):  # the timed instantiation itself (spmv_march2_kernel<27, 8, 3, true, true> at 512^3), not an estimate from 256^3
                todo.append(("cg27_512", ["--suite", "cg", "--grid", "512", "--stencil", "27", "--pc", "jacobi"], "27pt_512_cg", src % "cg --grid 512 --stencil 27", 36))
            if ran("cg_jacobi_5pt_4096x4096"):
                todo.append(("cg5", ["--suite", "cg", "--stencil", "5", "--pc", "jacobi", "--suite-dims", "4096x4096x1"], "5pt_4096_cg", src % "cg --stencil 5 --suite-dims 4096x4096x1", 12))
            if ran("config5_share_cg_none_7pt_1024x1024x128"):
                todo.append(("box", ["--suite", "cg", "--stencil", "7", "--pc", "none", "--suite-dims", "1024x1024x128"], "config5_share_cg", src % "cg --suite-dims 1024x1024x128 --pc none", 22))
            if ran("config3_sor_arbitrary_values_27pt_256"):
                todo.append(("sor27var", ["--suite", "sor", "--grid", "256", "--stencil", "27", "--suite-perturb", "1"], "27pt_256_sor_arbitrary_values", src % "sor --suite-perturb 1", 20))
            if ran("config4_surrogate_spmv") or ran("config4_solver_cg_jacobi"):
                todo.append(("sell", ["--suite", "sell"], "config4_standin_spmv", src % "sell", 40))
            if ran("config4_solver_cg_sor"):
                todo.append(("sorflan", ["--suite", "sorflan"], "config4_standin_sor", src % "sorflan", 45))
        for key, sargs, tag, source, est in todo:
            if est and not room(est):
                pmc[key] = (None, "counter pass skipped: budget")
                continue
            t_ = time.time()
            pmc[key] = (pmc_suite(sargs, tag), source)
            timeline.append(["pmc " + key, round(time.time() - t_, 2)])

    import threading
    th = None
    if not args.no_traffic:
        th = threading.Thread(target=counter_passes)
        th.start()

    best_ranks = None
    if not args.no_cpu_baseline:
        cores = physical_cores()
        base = None
        t_cpu = time.time()
        if N <= 2 ** 25:
            # the box's host cores.  A memory-bound solve peaks well below P = cores ranks (round 4's boxes, 128 cores: 32 ranks 28.5 it/s, 64 ranks
            # 16.7, 128 ranks 19.0): the default run takes P/4 ranks only; --full tries P, P/2, P/4 and reports the best
            its_p = 40 if n >= 200 else 200
            tried, rp = [], None
            counts = [cores, cores // 2, cores // 4] if args.full else [max(2, cores // 4)]
            for p_ in [c for c in dict.fromkeys(counts) if c > 1]:
                rr = ref_driver(p_, head.driver_args(its_p), bind=True) or ref_driver(p_, head.driver_args(its_p))
                if rr is not None:
                    rr["ranks"] = p_
                    tried.append({"ranks": p_, "iterations_per_s": rr["its"] / rr["seconds"]})
                    if rp is None or rr["its"] / rr["seconds"] > rp["its"] / rp["seconds"]:
                        rp = rr
            r1 = ref1 if ref1 is not None else ref_driver(1, head.driver_args(GATE_ITS))
            if rp is not None or r1 is not None:
                best = rp if rp is not None else r1
                best_ranks = rp["ranks"] if rp is not None else None
                base = {"value": best["its"] / best["seconds"], "unit": "iterations/s", "cores": rp["ranks"] if rp is not None else 1, "kind": "reference",
                        "physical_cores": cores, "ranks_tried": tried, "value_1core": (r1["its"] / r1["seconds"]) if r1 else None,
                        "sample": "reference KSPSolve (KSP%s+%s, MPIAIJ, MKL 1 thread/rank, gcc -O2; oracle/_ref) on the same %d-pt %s system: %s its on %d MPI ranks of %d physical cores "
                                  "(KSPSolve wall %s s; rank counts tried: %s), %s its on 1 core (%s s); assembly excluded"
                                  % (args.ksp.upper(), head.pcname(), args.stencil, head.shape(), rp["its"] if rp else "-", rp["ranks"] if rp else 0, cores, "%.3f" % rp["seconds"] if rp else "-",
                                     ",".join(str(t_["ranks"]) for t_ in tried), r1["its"] if r1 else "-", "%.3f" % r1["seconds"] if r1 else "-")}
        if base is None and head.cube and args.ksp == "cg" and args.pc == "jacobi" and N <= 2 ** 25:
            ai, aj, aa = assemble(ks, args.stencil, dims, 0, N)
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle as orc
            base, _ = oracle_port_baseline(ai, aj, aa, orc.matmult(ai, aj, aa, np.ones(N)), 20.0, args.stencil, n)
            del ai, aj, aa
        out["cpu_baseline"] = base
        timeline.append(["cpu baseline (headline)", round(time.time() - t_cpu, 2)])
        if best_ranks:
            for name, (cfg, cpu_its, _) in leg_cfgs.items():
                if cpu_its and ran(name):
                    if not room(12):
                        other[name]["cpu_baseline"] = {"skipped": "budget"}
                        continue
                    t_ = time.time()
                    other[name]["cpu_baseline"] = cpu_baseline_for(cfg, best_ranks, cpu_its, "KSP%s + %s" % (cfg.ksp.upper(), cfg.pcname()))
                    timeline.append(["cpu baseline " + name, round(time.time() - t_, 2)])
            for name, c4, its4 in (("config4_solver_cg_jacobi", cfg4, 10), ("config4_solver_cg_sor", cfg4s, 5)):
                if not args.no_other and c4 is not None and c4.binfile and ran(name) and room(40):
                    t_ = time.time()
                    other[name]["cpu_baseline"] = cpu_baseline_for(c4, best_ranks, its4, "KSPCG + %s, MatLoad of the same file" % c4.pcname())
                    timeline.append(["cpu baseline " + name, round(time.time() - t_, 2)])
    else:
        out["cpu_baseline"] = None
    if tmp4:
        shutil.rmtree(tmp4, ignore_errors=True)
    if th is not None:
        th.join()

    # ---- fold the counter bytes into the roofline lines
    def put_traffic(line, res_src, needles, time_ms, scale=1.0, note=None):
        """line['traffic'] = sum of the PMC bytes of the kernels matching `needles` (x scale), achieved / frac on them"""
        res, source = res_src if res_src else (None, None)
        found = [pick_kernel(res, nd) for nd in needles]
        if not res or any(v is None for _, v in found):
            line["traffic_source"] = source if (not res and source) else "counter pass did not complete on this box"
            return False
        tb = int(sum(v["bytes"] for _, v in found) * scale)
        line.update({"traffic": tb, "traffic_detail": {k: v for k, v in found}, "traffic_source": source + (" -- " + note if note else "")})
        if time_ms and time_ms > 0:
            line["achieved_on_counter_bytes"] = tb / (time_ms * 1e-3) / 1e9
            line["frac_counter_bytes"] = line["achieved_on_counter_bytes"] / HBM_PEAK_GBS
        return True

    if "cg" in pmc and pmc["cg"][0]:
        res, source = pmc["cg"]
        cal = calibration(res, 8 * N)
        tot_b = 0
        for k in by_kernel:
            kn_, v = pick_kernel(res, k["match"])
            if v:
                k.update({"kernel": kn_, "traffic": v["bytes"], "achieved": v["bytes"] / (k["avg_launch_us"] * 1e-6) / 1e9, "launches_sampled": v["launches_sampled"]})
                k["frac"] = k["achieved"] / HBM_PEAK_GBS
                tot_b += v["bytes"] * k["launches_per_iteration"]
        d0 = by_kernel[0]
        if "traffic" in d0:
            rf.update({"traffic": d0["traffic"], "achieved": d0["achieved"], "frac": d0["frac"], "basis": "HBM bytes moved (PMC counters) / launch time", "traffic_source": source,
                       "calibration": cal, "frac_of_measured_copy_peak_6290": d0["achieved"] / 6290.0})
        if tot_b and out["ms_per_step"]:
            rf["iteration"].update({"traffic_bytes": int(tot_b), "achieved": tot_b / (out["ms_per_step"] * 1e-3) / 1e9})
            rf["iteration_frac"] = rf["iteration"]["achieved"] / HBM_PEAK_GBS
    for key, e in extra_lines.items():
        kg = e["kernel"].split(" ")[0]
        if put_traffic(e, pmc.get("spmv"), [kg], e["avg_launch_ms"]):
            e["calibration"] = calibration(pmc["spmv"][0], 8 * N)
    if not args.no_other:
        c3 = other.get("config3_solver_gmres30_sor_27pt_256", {})
        if "roofline_sor" in c3:
            needles = ["sor_box_kernel<false", "sor_box_kernel<true"] if c3.get("sor_schedule") == "plane march" else ["sor_strand_kernel<0", "sor_strand_kernel<1"]
            if put_traffic(c3["roofline_sor"], pmc.get("sor27"), needles, c3["roofline_sor"]["avg_call_ms"]):
                c3["roofline_sor"]["achieved"], c3["roofline_sor"]["frac"] = c3["roofline_sor"]["achieved_on_counter_bytes"], c3["roofline_sor"]["frac_counter_bytes"]
        if "roofline_spmv" in c3:
            put_traffic(c3["roofline_spmv"], pmc.get("spmv27"), [c3["roofline_spmv"]["kernel"].split(" ")[0]], c3["roofline_spmv"]["avg_launch_ms"])
        c5 = other.get("config5_share_cg_none_7pt_1024x1024x128", {})
        if "roofline_spmv" in c5:
            put_traffic(c5["roofline_spmv"], pmc.get("box"), [c5["roofline_spmv"]["kernel"].split(" ")[0]], c5["roofline_spmv"]["avg_launch_ms"],
                        note="counter pass on the same solver at the same size (1024 x 1024 x 128); the product kernel carries the CG direction update as its prologue")
            if pmc.get("box") and pmc["box"][0] and "cg_update_ms" in c5:
                _, vu = pick_kernel(pmc["box"][0], "cg_fused_")
                if vu:
                    c5["roofline_cg_update"] = {"bound": "hbm", "kernel": "cg_fused_kernel", "avg_launch_ms": c5["cg_update_ms"], "traffic": int(vu["bytes"]), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                "frac_counter_bytes": vu["bytes"] / (c5["cg_update_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
        for nm, key in (("cg_jacobi_27pt_512_strong", "cg27_512"), ("cg_jacobi_5pt_4096x4096", "cg5")):
            cl = other.get(nm, {})
            if "roofline_spmv" in cl:  # the counter pass ran the same solver at the same size: the product kernel's own bytes (its CG prologue included) over its own launch time
                if put_traffic(cl["roofline_spmv"], pmc.get(key), [cl["roofline_spmv"]["kernel"].split(" ")[0]], cl["roofline_spmv"]["avg_launch_ms"],
                               note="counter pass on the same solver at the same size; the product kernel carries the CG direction update as its prologue"):
                    cl["roofline_spmv"]["achieved"], cl["roofline_spmv"]["frac"] = cl["roofline_spmv"]["achieved_on_counter_bytes"], cl["roofline_spmv"]["frac_counter_bytes"]
                if pmc.get(key) and pmc[key][0] and "cg_update_ms" in cl:
                    _, vu = pick_kernel(pmc[key][0], "cg_fused_")
                    if vu:
                        cl["roofline_cg_update"] = {"bound": "hbm", "kernel": "cg_fused_kernel", "avg_launch_ms": cl["cg_update_ms"], "traffic": int(vu["bytes"]), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                    "frac_counter_bytes": vu["bytes"] / (cl["cg_update_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
        sv = other.get("config3_sor_arbitrary_values_27pt_256", {})
        if "strand_streamed_coefficients_ms" in sv:
            line = {"bound": "hbm", "kernel": "sor_strand_kernel with streamed coefficients, forward + backward", "avg_call_ms": sv["strand_streamed_coefficients_ms"], "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "algorithmic_bytes": sv.get("algorithmic_bytes"), "traffic": None}
            if put_traffic(line, pmc.get("sor27var"), ["sor_strand_kernel<0", "sor_strand_kernel<1"], line["avg_call_ms"]):
                line["achieved"], line["frac"] = line["achieved_on_counter_bytes"], line["frac_counter_bytes"]
            sv["roofline_sor"] = line
        c4s = other.get("config4_solver_cg_sor", {})
        if "roofline_sor" in c4s:
            kin = "sor_inode_coop_kernel" if pmc.get("sorflan") and pmc["sorflan"][0] and pick_kernel(pmc["sorflan"][0], "sor_inode_coop_kernel<0")[1] else "sor_inode_kernel"
            if put_traffic(c4s["roofline_sor"], pmc.get("sorflan"), [kin + "<0", kin + "<1"], c4s["roofline_sor"]["avg_call_ms"],
                           note="forward (kind 0) + backward (kind 1) launch of the node-level sweep; the sweep is bound by the dependency levels of the node graph (one hop of ~2.7 us per level, 1344 levels on this stand-in), not by bytes"):
                c4s["roofline_sor"]["achieved"], c4s["roofline_sor"]["frac"] = c4s["roofline_sor"]["achieved_on_counter_bytes"], c4s["roofline_sor"]["frac_counter_bytes"]
        for nm in ("config4_surrogate_spmv", "config4_solver_cg_jacobi", "config4_solver_cg_sor"):
            c4 = other.get(nm, {})
            line = c4.get("roofline_longrow") or c4.get("roofline_spmv")
            if line:
                kk = (c4.get("kernel") or c4.get("spmv_kernel") or "").split(" ")[0]
                put_traffic(line, pmc.get("sell"), [kk or "spmv_sell_kernel"], line["avg_launch_ms"],
                            note=None if nm == "config4_surrogate_spmv" else "counter pass on the stand-in with the same pattern (other values)")
    out["budget_s"] = None if deadline == float("inf") else args.budget_s
    out["timeline"] = timeline
    emit(out, t_start)


def main_multi(args, head, rank, world, dev, shared, dist, torch, hx, sync, t_start=None):
    """N > 1 ranks: probe the transports out of process, time the headline configuration on each usable one (RCCL first: the
    north_star's transport; IPC peer stores as the alternative / fallback), report the faster as `value`, then run the scaling
    legs on it.  Every leg checks its first iterations against the committed exact-reduction history of the same system."""
    from petsc_amd import _lib
    from petsc_amd import dist as pdist
    uid = C.c_ulonglong()
    _lib.chk(hx.hipxDeviceUID(C.byref(uid)))
    devs = [None] * world
    dist.all_gather_object(devs, {"rank": rank, "device": dev, "device_uid": "%016x" % uid.value})
    distinct = len({d["device_uid"] for d in devs})
    shared = shared or distinct < world
    want = [args.transport] if args.transport != "auto" else (["ipc"] if shared else ["rccl", "ipc"])
    multi = {"ranks": world, "distinct_devices": distinct, "devices": devs, "transports": {}, "launcher": "self (bench.py -> torch.distributed.run)" if os.environ.get("HIPX_SELF_LAUNCHED") else "external (torch.distributed.run)"}
    if shared:
        multi["note"] = "ranks share GPUs on this box (%d device(s) for %d ranks): RCCL refuses that, the IPC transport runs; this is a functional run, not a scaling measurement" % (distinct, world)
    legs, best = {}, None
    for t in want:
        ok, notes = probe_transport(t, rank, world, dev, dist)
        multi["transports"][t] = {"probe_ok": ok, "probe": notes[:2] + (["..."] if world > 2 else [])}
        if not ok:
            continue
        try:
            pdist.comm_init(rank, world, dist, t)
            res, _ = run_leg(head, rank, world, dist, torch, t, args.steps, args.warmup, sync, variant=args.variant, parity_its=args.parity_its, fused=args.fused, pipeline=args.pipeline)
            nr = C.c_int()
            _lib.chk(hx.hipxCommRank(None, C.byref(nr)))
            res["comm_nranks"] = nr.value
            legs[t] = res
            multi["transports"][t].update({"iterations_per_s": res["iterations_per_s"], "parity": res["parity"], "comm_nranks": nr.value})
            _lib.chk(hx.hipxCommFinalize())
            if res["parity"]["pass"] is not False and (best is None or res["iterations_per_s"] > legs[best]["iterations_per_s"]):
                best = t
        except Exception as e:  # noqa: BLE001  (a rank-local failure here cannot be recovered collectively: say what happened on stdout -- one line, the
            # contract's shape, value null -- and end the job: the launcher tears the other ranks down instead of leaving them in a collective)
            multi["transports"][t]["error"] = str(e)[:300]
            print(json.dumps(compact_line({"metric": head.metric(), "value": None, "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
                                           "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic", "config": {"workload": head.metric()},
                                           "multi_gpu": multi, "error": "rank %d failed on transport %s: %s" % (rank, t, str(e)[:300])})))
            sys.stdout.flush()
            os._exit(3)
    out = None
    if rank == 0:
        pcname = head.pcname()
        if best is None:
            out = {"metric": head.metric(), "value": None, "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
                   "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic", "config": {"workload": head.metric()}, "multi_gpu": multi,
                   "error": "no transport came up on every rank (or parity failed on all)"}
        else:
            res = legs[best]
            out = {"metric": head.metric(), "value": res["iterations_per_s"] if res["parity"]["pass"] is not False else None, "unit": "iterations/s", "n_gpus": world,
                   "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
                   "dtype": "f64", "data": "synthetic",
                   "config": {"workload": "3-D %d-pt Poisson %dx%dx%d (N=%d rows), KSP%s + %s, b = A*1, x0 = 0; rows split over %d rank(s)"
                                          % (head.stencil, head.dims[0], head.dims[1], head.dims[2], head.N, head.ksp.upper(), pcname, world),
                              "global_rows": head.N, "parallelism": "rows%d" % world, "transport": best, "fused": args.fused, "pipeline": args.pipeline, "spmv_variant": args.variant,
                              "residual_norm_after": res["residual_norm_after"]},
                   "ungated": res["parity"]["pass"] is None, "parity_gate": res["parity"],
                   "roofline": {"bound": "hbm", "kernel": res["spmv_kernel"], "avg_launch_ms": res["per_rank"][0]["spmv_ms"], "algorithmic_bytes": res["spmv_algorithmic_bytes_rank0"],
                                "achieved": res["spmv_algorithmic_bytes_rank0"] / (res["per_rank"][0]["spmv_ms"] * 1e-3) / 1e9 if res["per_rank"][0]["spmv_ms"] > 0 else None,
                                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                                "note": "rank 0's diagonal-block SpMV on the algorithmic CSR bytes of its slab (the N = 1 line carries the counter-based figure)"},
                   "per_rank": res["per_rank"], "cpu_baseline": None, "multi_gpu": multi}
            if out["roofline"]["achieved"]:
                out["roofline"]["frac"] = out["roofline"]["achieved"] / HBM_PEAK_GBS
    # ---- the north_star scaling legs on the faster transport, most important first; rank 0 decides (and tells the others) whether a leg's
    # estimated cost still fits the budget -- a leg is collective, every rank must take the same decision
    if best is not None and not args.no_other:
        pdist.comm_init(rank, world, dist, best)
        other = {}
        deadline = float("inf") if (args.full or args.budget_s <= 0 or t_start is None) else t_start + args.budget_s
        scaling_legs = [("cg_jacobi_27pt_512_strong", Cfg(27, (512, 512, 512), "cg", "jacobi", "strong"), 60, 5, 16, 20 + 60 // world),
                        ("config3_gmres30_sor_27pt_512_strong", Cfg(27, (512, 512, 512), "gmres", "sor", "strong", golden="gmres_sor_27pt_512"), 60, 5, 16, 25 + 60 // world),
                        ("config5_cg_none_7pt_1024x1024x%d_weak" % (128 * world), Cfg(7, (1024, 1024, 128 * world), "cg", "none", "weak"), 60, 5, 12, 16),
                        ("config3_solver_gmres30_sor_27pt_256_parity", Cfg(27, (256, 256, 256), "gmres", "sor", "strong", golden="gmres_sor_27pt_256"), 60, 5, 35, 12)]
        scaling_legs = [leg + (1,) for leg in scaling_legs]
        # A leg is a collective job on transports no box of five rounds could exercise across GPUs: should one stall, the measured headline must not
        # be lost with it.  Rank 0 arms a timer (budget + grace); if it fires, the line is printed with what has been collected and the process leaves.
        watchdog = None
        if rank == 0 and out is not None and deadline != float("inf"):
            import threading

            def _give_up():
                out["other_configs"] = dict(other)
                out["budget_s"] = args.budget_s
                out["scaling_legs_error"] = "a scaling leg did not return within the budget + 120 s: the line is printed without it"
                try:
                    emit(out, t_start)
                    sys.stdout.flush()
                finally:
                    # a stalled collective is a failure: leave with a non-zero status -- the launcher (torch.distributed.run) then tears the
                    # other ranks down instead of leaving them in the collective with their GPUs held; the line above carries `scaling_legs_error`
                    os._exit(4)
            watchdog = threading.Timer(max(30.0, deadline - time.time()) + 120.0, _give_up)
            watchdog.daemon = True
            watchdog.start()
        if head.ksp == "cg" and head.pc in ("jacobi", "none") and args.pipeline == 1 and args.fused:
            # the one-all-reduce form beside the two-all-reduce one the line's `value` is measured with (round 5: launch-ahead single-reduction CG, cg.c:364-534 -- the same
            # KSPCG with KSPCGUseSingleReduction; its history is gated against the standard form's yardstick at 1e-9, not 1e-12: another recurrence for A p)
            scaling_legs.insert(0, ("headline_single_reduction_launch_ahead", head, args.steps, args.warmup, args.parity_its, 10 + 20 // world, 4))
            scaling_legs.insert(2, ("cg_jacobi_27pt_512_strong_single_reduction_launch_ahead", Cfg(27, (512, 512, 512), "cg", "jacobi", "strong"), 60, 5, 16, 20 + 60 // world, 4))
            # round 6: KSPPIPECG (pipecg.c) on the headline's system -- one fused update kernel + one product per iteration and rank, its ONE all-reduce started before the product
            # and collected after it (hipxPipeCGUpdateBeginAllreduce ... hipxAllreduceEnd); a timed "step" is one pass of a complete K-iteration solve
            if head.cube:
                scaling_legs.append(("headline_pipecg_launch_ahead", Cfg(head.stencil, head.dims, "pipecg", head.pc, head.scaling), args.steps, args.warmup, args.parity_its, 10 + 20 // world, 1))
                scaling_legs.append(("headline_groppcg_launch_ahead", Cfg(head.stencil, head.dims, "groppcg", head.pc, head.scaling), args.steps, args.warmup, args.parity_its, 10 + 20 // world, 1))
        if args.only_legs:
            scaling_legs = [leg for leg in scaling_legs if any(w and w in leg[0] for w in args.only_legs.split(","))]
        for name, cfg, st, wu, pits, est, pipe in scaling_legs:
            go = [time.time() + est <= deadline]
            dist.broadcast_object_list(go, src=0)
            if not go[0]:
                other[name] = {"skipped": "budget"}
                continue
            try:
                t_ = time.time()
                res, _ = run_leg(cfg, rank, world, dist, torch, best, st, wu, sync, parity_its=pits, pipeline=pipe)
                res["leg_seconds"] = time.time() - t_
                res["pipeline"] = pipe
                other[name] = res
            except Exception as e:  # noqa: BLE001
                other[name] = {"error": str(e)[:400]}
        if watchdog is not None:
            watchdog.cancel()
        if out is not None:
            out["other_configs"] = other
            out["budget_s"] = None if deadline == float("inf") else args.budget_s
        _lib.chk(hx.hipxCommFinalize())
    if rank == 0:
        emit(out, t_start)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
