#!/usr/bin/env python
"""bench.py -- the BASELINE.json metric: CG iterations/s (+ SpMV achieved HBM GB/s) on 3-D 7-point Poisson 256^3, fp64,
KSPCG + PCJACOBI, MATAIJHIPX/VECHIPX kernels, on N GPUs of one node (strong scaling: the 256^3 problem is split by rows).

A "step" is one CG iteration (one pass of the loop body cg.c:220-349: MatMult, 2 dots, 1 norm, 2 AXPY, 1 AYPX, PCApply).
Inputs (CSR matrix, b = A*1, x0 = 0) are resident in HBM before the timed region.

  python bench.py --gpus 1 --steps 200 --warmup 20
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- dominant kernel (CSR SpMV): algorithmic bytes per launch / mean launch duration measured with HIP events
                  on the compute stream inside the timed region, against the 8 TB/s HBM3E peak
  cpu_baseline -- the CPU oracle (scalar restatement of the reference path, 1 core) timed on a bounded sample of the same
                  workload on this host (rank 0, N = 1 only)
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); measured copy peak is 6290


def assemble(ks, stencil, n, rs, re):
    f = {7: ks.HipxAssemble_poisson7, 27: ks.HipxAssemble_bench27}[stencil]
    nz = f(n, rs, re, None, None, None)
    wide = nz >= 2 ** 31 - 8  # 64-bit row offsets (27-pt 512^3 / 7-pt 1024^3 on one rank)
    if wide:
        f = {7: ks.HipxAssemble_poisson7_64, 27: ks.HipxAssemble_bench27_64}[stencil]
    ai = np.zeros(re - rs + 1, np.int64 if wide else np.int32)
    aj = np.zeros(nz, np.int32)
    aa = np.zeros(nz, np.float64)
    f(n, rs, re, ai.ctypes.data_as(C.c_void_p), aj.ctypes.data_as(C.c_void_p), aa.ctypes.data_as(C.c_void_p))
    return ai, aj, aa


def cpu_baseline_reference(stencil, n, its):
    """The REFERENCE itself (oracle/_ref: libpetsc compiled from /root/reference by oracle/build_ref.py) on the host:
    its own MatSetValues assembly, KSPSolve_CG, MatMult_SeqAIJ, PCJACOBI, MKL BLAS-1 -- one rank (MPIUNI), one thread."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "bin", "ref_driver")
    if not os.path.exists(exe):
        return None
    env = dict(os.environ, MKL_NUM_THREADS="1", OMP_NUM_THREADS="1")
    cmd = [exe, "-stencil", str(stencil), "-n", str(n), "-ksp_type", "cg", "-pc_type", "jacobi", "-ksp_rtol", "1e-50", "-ksp_max_it", str(its),
           "-mat_type", "aij", "-vec_type", "standard"]
    try:
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=600).stdout
        m = re.search(r"iterations (\d+) reason (-?\d+) error \S+ KSPSolve_seconds (\S+)", out)
        done, secs = int(m.group(1)), float(m.group(3))
    except Exception:
        return None
    return {"value": done / secs, "unit": "CG iterations/s", "cores": 1, "kind": "reference",
            "sample": "%d iterations of the reference's own KSPSolve (KSPCG + PCJACOBI, MATSEQAIJ, VECSEQ, MKL BLAS single-threaded, gcc -O2; oracle/_ref) "
                      "on the same %d-pt %d^3 system; KSPSolve wall %.3f s, assembly excluded" % (done, stencil, n, secs)}


def cpu_baseline(ai, aj, aa, b, budget_s, stencil, n):
    """Oracle CG + Jacobi on the same system, bounded to ~budget_s seconds of CPU work.  Checker code, timed as a baseline."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    t0 = time.perf_counter()
    orc.ksp_solve("cg", ai, aj, aa, b, pc="jacobi", rtol=1e-50, max_it=2)
    t2 = time.perf_counter() - t0  # includes set-up + 2 iterations
    its = int(max(3, min(60, budget_s / max(t2 / 3.0, 1e-3))))
    t0 = time.perf_counter()
    _, done, _, _ = orc.ksp_solve("cg", ai, aj, aa, b, pc="jacobi", rtol=1e-50, max_it=its)
    t_its = time.perf_counter() - t0
    t0 = time.perf_counter()
    orc.ksp_solve("cg", ai, aj, aa, b, pc="jacobi", rtol=1e-50, max_it=1)
    t_one = time.perf_counter() - t0
    per_it = (t_its - t_one) / max(done - 1, 1)
    return {"value": 1.0 / per_it, "unit": "CG iterations/s", "cores": 1, "kind": "port",
            "sample": "%d CG+Jacobi iterations of the oracle (scalar C restatement of cg.c/aij.c, gcc -O2) on the same %d-pt %d^3 system" % (done, stencil, n)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--grid", dest="n", type=int, default=256, help="grid points per side (not --n: torchrun would read it as an abbreviation of its own options)")
    ap.add_argument("--stencil", type=int, default=7, choices=[7, 27])
    ap.add_argument("--fused", type=int, default=1, help="1 (default): fused SpMV+dot and AXPY+AXPY+PCJACOBI+norm+dot kernels -- same arithmetic and order per element, fewer HBM passes; 0: one kernel per reference Vec/Mat call (cg.c:249-344)")
    ap.add_argument("--pipeline", type=int, default=1, help="1 (default): launch-ahead fused CG (iteration i+1 enqueued before the host has seen iteration i's sums; device-resident scalars); 0: host waits between kernels")
    ap.add_argument("--variant", type=int, default=0, help="SpMV kernel variant (include/hipx.h hipxMatSetSpMVVariant): 0 auto, 1 32-bit-column stream kernel, 22/23 packed 16-bit columns, 24/25 packed columns + 8-bit value dictionary")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=20.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("HIPX_ALL_RANKS_DEVICE0") == "1":  # debugging aid on a 1-GPU box: every rank drives GPU 0
        local_rank = 0
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="cpu:gloo,cuda:nccl", rank=rank, world_size=world)
    assert world == args.gpus, "--gpus must equal WORLD_SIZE (launch N > 1 with torch.distributed.run)"

    from petsc_amd import _lib
    from petsc_amd import dist as pdist
    hx = _lib.init(local_rank)
    _, ks = _lib.load()

    n, N = args.n, args.n ** 3
    ranges = pdist.split_ownership(N, world)
    rs, re = int(ranges[rank]), int(ranges[rank + 1])
    ai, aj, aa = assemble(ks, args.stencil, n, rs, re)
    m = re - rs
    if world > 1:
        idb = (C.c_char * 256)()
        if rank == 0:
            _lib.chk(hx.hipxCommGetUniqueId(idb))
        box = [bytes(idb)]
        dist.broadcast_object_list(box, src=0)
        _lib.chk(hx.hipxCommInit(box[0], rank, world))
        plan = pdist.build_plan(ai, aj, aa, ranges, rank, dist=dist)
        M, keep = pdist.create_device_mat(plan, world)
        nnz_local = int(plan["Ai"][-1])
    else:
        A = _lib.mat_create_csr(m, m, ai, aj, aa)
        M, keep = _lib.HipxMat(m=m, A=A, B=None, halo=None, lvec=None, nranks=1), [A]
        nnz_local = int(ai[-1])
    _lib.chk(hx.hipxMatSetSpMVVariant(M.A, args.variant))

    # b = A * 1 (ex2.c:139 style), x0 = 0
    ones = _lib.DVec(m, np.ones(m))
    B = _lib.DVec(m)
    X = _lib.DVec(m)
    _lib.chk(ks.HipxMatMult(C.byref(M), ones.ptr, B.ptr))
    pc = _lib.HipxPC()
    ks.HipxPCSetDefaults(C.byref(pc))
    _lib.chk(ks.HipxPCSetUp(C.byref(pc), C.byref(M)))
    ksp = _lib.HipxKSP()
    ks.HipxKSPSetDefaults(C.byref(ksp))
    ksp.rtol, ksp.abstol, ksp.divtol = 1e-50, 1e-300, 1e300
    ksp.max_it = args.warmup + args.steps + 10
    ksp.fused = args.fused
    ksp.pipeline = args.pipeline
    _lib.chk(ks.HipxKSPCGBegin(C.byref(ksp), C.byref(M), C.byref(pc), B.ptr, X.ptr))
    _lib.chk(ks.HipxKSPCGStep(C.byref(ksp), C.byref(M), C.byref(pc), B.ptr, X.ptr, args.warmup))
    assert ksp.reason == 0 and ksp.its == args.warmup, (ksp.reason, ksp.its)

    def sync():
        _lib.chk(hx.hipxDeviceSynchronize())
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    _lib.chk(hx.hipxProfileSpMV(1))
    sync()
    t0 = time.perf_counter()
    _lib.chk(ks.HipxKSPCGStep(C.byref(ksp), C.byref(M), C.byref(pc), B.ptr, X.ptr, args.steps))
    sync()
    elapsed = time.perf_counter() - t0
    assert ksp.reason == 0 and ksp.its == args.warmup + args.steps, (ksp.reason, ksp.its)
    cnt, tot_ms = C.c_int(), C.c_double()
    _lib.chk(hx.hipxProfileSpMVGet(C.byref(cnt), C.byref(tot_ms)))
    _lib.chk(hx.hipxProfileSpMV(0))
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
    rnorm = float(ksp.rnorm)

    # roofline of the dominant kernel (the diagonal-block / sequential SpMV launch of this rank)
    spmv_bytes = 12 * nnz_local + (8 if ai.dtype == np.int64 else 4) * (m + 1) + 16 * m  # val + col + row offsets + x + y
    spmv_ms = tot_ms.value / max(cnt.value, 1)
    achieved = spmv_bytes / (spmv_ms * 1e-3) / 1e9 if spmv_ms > 0 else 0.0

    kbuf = C.create_string_buffer(256)
    _lib.chk(hx.hipxMatGetSpMVKernel(M.A, kbuf, 256))
    kname = kbuf.value.decode()
    # HBM traffic of the SpMV launch cannot be counted from inside this process; it comes from the committed rocprofv3
    # PMC passes of this same command (profiles/README.md, scripts/pmc_summary.py), matched on kernel and workload
    traffic = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "spmv_traffic.json")))
        key = "%dpt_%d_%s_g%d" % (args.stencil, n, kname.split(" ")[0], world)
        if key in tj:
            traffic = tj[key]["traffic_bytes"]
    except Exception:
        pass
    out = None
    if rank == 0:
        value = args.steps / elapsed
        out = {
            "metric": "CG iterations/sec, %d-pt Poisson %d^3 fp64, KSPCG+PCJACOBI" % (args.stencil, n),
            "value": value, "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "3-D %d-pt Poisson %d^3 (N=%d rows, nnz=%d local), KSPCG + PCJACOBI, b = A*1, x0 = 0; rows split over %d rank(s)"
                                   % (args.stencil, n, N, nnz_local, world),
                       "global_rows": N, "parallelism": "rows%d" % world, "fused": args.fused, "pipeline": args.pipeline, "spmv_variant": args.variant,
                       "residual_norm_after": rnorm},
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "launches": cnt.value, "avg_launch_ms": spmv_ms, "algorithmic_bytes": spmv_bytes,
                         "frac_of_measured_copy_peak_6290": achieved / 6290.0,
                         "note": "achieved = CSR algorithmic bytes (12 nnz + 4 (N+1) + 16 N, SURVEY 8(d)) / launch time; the packed kernels move fewer bytes "
                                 "than that (16-bit column codes; 8-bit value codes when a[] has <= 256 distinct values), see traffic"},
        }
        if world == 1 and not args.no_cpu_baseline:
            ref = cpu_baseline_reference(args.stencil, n, 24 if n >= 200 else 100) if N <= 2 ** 25 else {
                "value": None, "unit": "CG iterations/s", "cores": 1, "kind": "reference", "sample": "not timed: the bounded-sample rule (10-30 s of CPU work) cannot hold at this size"}
            if ref is None:
                bh = B.get()
                ref = cpu_baseline(ai, aj, aa, bh, args.cpu_baseline_seconds, args.stencil, n)
            out["cpu_baseline"] = ref
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        _lib.chk(hx.hipxCommFinalize())
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
