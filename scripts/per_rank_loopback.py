#!/usr/bin/env python
"""One rank of an 8-rank CG + PCJACOBI run, alone on the GPU with a loop-back ghost exchange (bench.py's per_rank_budget leg) -- as a stand-alone workload for
`rocprofv3 --kernel-trace`: the kernel timeline of a rank's iteration (which kernel waits for which; what the comm stream's put kernel does beside the product).

    python scripts/per_rank_loopback.py [--stencil 27 --grid 512 --world 8 --rank 3 --fused 1 --its 40]
    rocprofv3 --kernel-trace --output-format csv -d out -o t -- python scripts/per_rank_loopback.py ...;  python scripts/trace_timeline.py out/.../t_kernel_trace.csv
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stencil", type=int, default=27)
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=3)
    ap.add_argument("--fused", type=int, default=1)
    ap.add_argument("--pipeline", type=int, default=1)
    ap.add_argument("--its", type=int, default=40)
    a = ap.parse_args()
    from petsc_amd import _lib
    from petsc_amd import dist as pdist
    hx = _lib.init(0)
    pdist.comm_init_loopback()
    cfg = bench.Cfg(a.stencil, (a.grid, a.grid, a.grid), "cg", "jacobi")
    P = bench.Problem(cfg, a.rank, a.world, None, transport="ipc", fused=a.fused, pipeline=a.pipeline, loopback=True)
    print("kernel:", P.setup(0))
    P.begin(a.its + 30)
    P.step(10)
    _lib.chk(hx.hipxDeviceSynchronize())
    t0 = time.perf_counter()
    P.step(a.its)
    _lib.chk(hx.hipxDeviceSynchronize())
    dt = time.perf_counter() - t0
    print("rows %d ghosts %d: %.4f ms per iteration (%d iterations), rnorm %.17g" % (P.m, P.nghost, 1e3 * dt / a.its, a.its, P.ksp.rnorm))
    P.destroy()
    _lib.chk(hx.hipxCommFinalize())


if __name__ == "__main__":
    main()
