"""GPU: the MULTI-RANK path of bench.py (torchrun ranks, C host layer, MATMPIAIJ split, ghost exchange, all-reduced dot/norm
sums, fused + launch-ahead CG) executed with world_size 2-4 on this box's ONE GPU.  RCCL refuses several ranks per device, so
the ranks use libhipx's IPC transport (peer stores + sequence flags through IPC-mapped arenas; --transport ipc); everything
above the transport -- plan, kernels, solver loop, timing protocol, JSON line -- is the code the 8-GPU run executes.
Checks: the residual after K iterations equals the 1-rank run's (the partition changes only the rounding of the reductions),
and KSPGMRES(30)+PCSOR (config 3's solver, per-rank local sweeps) runs on the strands schedule on every rank."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(nranks, extra, port):
    env = dict(os.environ, HIPX_ALL_RANKS_DEVICE0="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    base = [os.path.join(ROOT, "bench.py"), "--gpus", str(nranks), "--quick", "--transport", "ipc"] + extra
    if nranks == 1:
        cmd = [sys.executable] + base
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks), "--master-addr", "127.0.0.1", "--master-port", str(port)] + base
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = r.stdout.splitlines()[-1]  # the compact contract line, last on stdout
    assert len(line) < 4096
    return json.loads(line)


@pytest.mark.parametrize("stencil,grid", [(7, 64), (27, 40)])
def test_cg_jacobi_strong_scaling_path_matches_one_rank(stencil, grid):
    extra = ["--stencil", str(stencil), "--grid", str(grid), "--steps", "30", "--warmup", "5"]
    one = run_bench(1, extra, 0)
    r1 = one["config"]["residual_norm_after"]
    assert one["n_gpus"] == 1 and r1 > 0
    for n in (2, 3, 4):
        d = run_bench(n, extra, 29500 + 7 * n + stencil)
        assert d["n_gpus"] == n and d["steps"] == 30 and d["scaling"] == "strong" and d["config"]["transport"] == "ipc" and d["value"] > 0
        rn = d["config"]["residual_norm_after"]
        assert abs(rn - r1) <= 1e-10 * r1, (n, rn, r1)


def test_weak_scaling_and_gmres_sor_paths_run():
    d = run_bench(2, ["--grid", "64", "--scaling", "weak", "--pc", "none", "--steps", "20", "--warmup", "5"], 29611)   # config 5's shape: 64 x 64 x 8 per rank
    assert d["scaling"] == "weak" and d["config"]["global_rows"] == 64 * 64 * 16 and d["value"] > 0
    d = run_bench(2, ["--stencil", "27", "--grid", "32", "--ksp", "gmres", "--pc", "sor", "--steps", "40", "--warmup", "3"], 29633)  # config 3's solver
    assert "GMRES" in d["metric"] and d["value"] > 0 and d["config"]["residual_norm_after"] < 1.0
