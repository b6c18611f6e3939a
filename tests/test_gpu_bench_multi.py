"""GPU: bench.py's multi-rank path as the driver will run it -- `python bench.py --gpus N` WITHOUT a launcher (self-launch) and
under torch.distributed.run -- on this one-GPU box: the ranks share the device, so the transport is the IPC one (peer stores);
the out-of-process transport probe, the slab split, the device ghost exchange, the all-reduces and the parity gate against the
committed exact-reduction history all run with 2 (and 3) real ranks.  VERDICT r2 item 1."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def clean_env():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "HIPX_ALL_RANKS_DEVICE0"):
        env.pop(k, None)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return env


def last_json(out):
    """The contract line is the LAST stdout line and compact (< 4 KB: what the driver parses); the full result of the same run is in
    bench_detail.json beside bench.py.  Returns the full result after checking that the line is its summary."""
    lines = out.splitlines()
    assert lines and lines[-1].startswith("{"), out[-3000:]
    assert len(lines[-1]) < 4096, len(lines[-1])
    c = json.loads(out[-8000:].splitlines()[-1])
    d = json.load(open(os.path.join(ROOT, c["detail"])))
    for k in ("metric", "n_gpus", "steps", "warmup", "unit", "scaling"):
        assert c[k] == d[k], k
    assert (c["value"] is None and d["value"] is None) or abs(c["value"] - d["value"]) <= 1e-6 * abs(d["value"])
    assert c["parity_gate"].get("pass") == d["parity_gate"].get("pass")
    return d


def test_self_launch_two_ranks_headline_shape_with_parity():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--grid", "64", "--steps", "10", "--warmup", "3", "--quick"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900, env=clean_env(), cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:]
    d = last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["unit"] == "iterations/s" and d["scaling"] == "strong"
    mg = d["multi_gpu"]
    assert mg["ranks"] == 2 and mg["launcher"].startswith("self")
    assert mg["distinct_devices"] == 1 and "ipc" in mg["transports"] and "rccl" not in mg["transports"]  # one GPU here: shared, IPC only
    assert mg["transports"]["ipc"]["probe_ok"] is True and mg["transports"]["ipc"]["comm_nranks"] == 2
    assert d["config"]["transport"] == "ipc" and d["config"]["parallelism"] == "rows2"
    g = d["parity_gate"]  # vs tests/golden/exact_histories.json[cg_jacobi_7pt_64]: the reference with exact BLAS reductions
    assert g["pass"] is True and g["max_rel_diff"] <= 1e-12 and d["ungated"] is False
    assert len(d["per_rank"]) == 2 and all(p["rows"] == 64 ** 3 // 2 for p in d["per_rank"])
    assert all(p["halo_ms"] >= 0 and p["allreduce_ms"] > 0 and p["spmv_ms"] > 0 for p in d["per_rank"])


def test_torchrun_three_ranks_weak_box():
    """The driver's own recipe (python -m torch.distributed.run ... bench.py --gpus N), 3 ranks, weak mode: 64 x 64 x 8 rows per
    rank -- uneven nothing, but a different box per rank count; --pc none; history vs nothing committed -> ungated but finite."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1", "--master-port", "29631",
           os.path.join(ROOT, "bench.py"), "--gpus", "3", "--grid", "64", "--scaling", "weak", "--pc", "none", "--steps", "8", "--warmup", "2", "--quick"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900, env=clean_env(), cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:]
    d = last_json(r.stdout)
    assert d["n_gpus"] == 3 and d["scaling"] == "weak" and d["config"]["global_rows"] == 64 * 64 * 24
    assert d["multi_gpu"]["launcher"].startswith("external") and d["ungated"] is True and d["value"] > 0


def test_self_launch_two_ranks_weak_with_golden():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--grid", "64", "--scaling", "weak", "--pc", "none", "--steps", "8", "--warmup", "2", "--quick"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900, env=clean_env(), cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:]
    d = last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["parity_gate"]["pass"] is True and d["parity_gate"]["max_rel_diff"] <= 1e-12  # cg_none_7pt_64x64x16


def test_launch_ahead_cg_on_two_ranks_equals_host_synchronised_loop():
    """The fused CG runs launch-ahead on several ranks too (the all-reduces complete on the stream and feed device-resident scalars:
    hipxMatMultMPIDotBegin, hipxCGFusedUpdateBeginAllreduce), and (round 4) as the single-reduction form (--pipeline 3: one 24-byte
    all-reduce per iteration).  Launch-ahead and host-synchronised loop (--pipeline 2) run the same arithmetic; their reduction kernels walk the
    vectors in different orders (round 4), so with the default reductions the residuals agree to rounding, and with EXACT reductions
    (HIPX_REDUCTIONS=exact: every sum rounded once, whatever the order, the kernel and the cut into ranks) they are the same double."""
    out = {}
    for mode in ("fast", "exact"):
        for pipe in ("1", "2", "3", "4"):
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--grid", "64", "--steps", "25", "--warmup", "4", "--quick", "--pipeline", pipe],
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900, env=dict(clean_env(), HIPX_REDUCTIONS=mode), cwd=ROOT)
            assert r.returncode == 0, r.stdout[-3000:]
            out[mode, pipe] = last_json(r.stdout)
            assert out[mode, pipe]["parity_gate"]["pass"] is True, (mode, pipe, out[mode, pipe]["parity_gate"])
    a, b = out["fast", "1"], out["fast", "2"]
    assert abs(a["config"]["residual_norm_after"] - b["config"]["residual_norm_after"]) <= 1e-12 * b["config"]["residual_norm_after"]
    a, b = out["exact", "1"], out["exact", "2"]
    assert a["config"]["residual_norm_after"] == b["config"]["residual_norm_after"]
    assert a["parity_gate"]["max_rel_diff_exact_reductions"] == b["parity_gate"]["max_rel_diff_exact_reductions"]
    c = out["exact", "3"]  # another recurrence: close to, not equal to, the standard form
    assert abs(c["config"]["residual_norm_after"] - b["config"]["residual_norm_after"]) <= 1e-9 * b["config"]["residual_norm_after"]
    # round 5: the launch-ahead single-reduction form (--pipeline 4: scalars formed on the device, the 24-byte all-reduce on the stream) = the host-synchronised
    # one (--pipeline 3), bit for bit, in both reduction modes
    for mode in ("fast", "exact"):
        assert out[mode, "4"]["config"]["residual_norm_after"] == out[mode, "3"]["config"]["residual_norm_after"], mode


def test_two_rank_line_times_the_single_reduction_form_beside_the_headline():
    """Round 5: on several ranks the default line (`value`: the two-all-reduce launch-ahead loop, gated at 1e-12) also times the launch-ahead single-reduction
    form (one 24-byte all-reduce per iteration; gated at 1e-9 against the standard form's yardstick) as a leg of its own -- within the wall-clock budget, decided
    by rank 0 for all ranks; legs that do not fit the budget say so."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--grid", "64", "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--no-traffic", "--no-plugin",
                        "--no-general", "--budget-s", "55"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900, env=clean_env(), cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:]
    d = last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["pipeline"] == 1 and d["parity_gate"]["pass"] is True and d["parity_gate"]["max_rel_diff"] <= 1e-12
    sr = d["other_configs"]["headline_single_reduction_launch_ahead"]
    assert sr["pipeline"] == 4 and sr["iterations_per_s"] > 0 and sr["parity"]["pass"] is True and sr["parity"]["max_rel_diff"] <= 1e-9, sr
    legs = d["other_configs"]
    assert any(v == {"skipped": "budget"} for v in legs.values()), list(legs)  # (the budget cuts the list somewhere; what ran is gated)
    assert all(v == {"skipped": "budget"} or v.get("parity", {}).get("pass") is not False for v in legs.values()), {k: v.get("parity") for k, v in legs.items() if isinstance(v, dict)}


def test_pipecg_on_two_ranks_through_bench_py():
    """Round 6: `bench.py --gpus 2 --ksp pipecg` -- the host layer's launch-ahead PIPECG on two ranks sharing the GPU: per iteration and rank one fused update
    kernel whose three sums START their all-reduce (IPC post phase), the product with its ghost exchange, the all-reduce's END.  Gated in the exact reduction mode
    against the committed history of the REFERENCE's KSPSolve_PIPECG + exact BLAS (one rank: the two-rank product's association differs from it by rounding,
    hence 1e-9 instead of equality -- equality against the partitioned oracle: tests/test_gpu_plugin_mpi.py); the CG headline's line carries it as a leg too."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--ksp", "pipecg", "--grid", "64", "--steps", "30", "--warmup", "3", "--quick"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900, env=clean_env(), cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:]
    d = last_json(r.stdout)
    assert d["n_gpus"] == 2 and "PIPECG" in d["metric"] and d["value"] > 0
    g = d["parity_gate"]
    assert g["pass"] is True and g["gated_reduction_mode"] == "exact" and g["max_rel_diff"] <= 1e-9 and g["max_rel_diff_fast_reductions"] <= 1e-8, g
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--grid", "64", "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--no-traffic", "--no-plugin",
                        "--no-general", "--budget-s", "300", "--only-legs", "pipecg,groppcg"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1200, env=clean_env(), cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:]
    for name in ("headline_pipecg_launch_ahead", "headline_groppcg_launch_ahead"):
        leg = last_json(r.stdout)["other_configs"][name]
        assert leg.get("iterations_per_s", 0) > 0 and leg["parity"]["pass"] is True, (name, leg)


def test_gmres_sor_on_two_and_four_ranks_follows_the_exact_yardstick_at_1e12():
    """Config 3's solver on 2 and 4 ranks sharing the GPU (IPC transport): in the exact reduction mode -- the ranks' sums folded as unrounded
    pairs, GMRES's MDot included (round 4) -- the first 35 residual norms sit within 1e-12 of the committed exact-reduction history of the same
    per-rank local sweeps (tests/golden/exact_histories.json gmres_sor_27pt_128_np{2,4}); bench.py gates GMRES legs in that mode."""
    for n in (2, 4):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--ksp", "gmres", "--pc", "sor", "--stencil", "27", "--grid", "128", "--steps", "40", "--warmup", "3", "--quick"],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900, env=clean_env(), cwd=ROOT)
        assert r.returncode == 0, r.stdout[-3000:]
        d = last_json(r.stdout)
        g = d["parity_gate"]
        assert d["n_gpus"] == n and g["pass"] is True and g["gated_reduction_mode"] == "exact" and g["max_rel_diff"] <= 1e-12, g


def _mem_available_gb():
    for ln in open("/proc/meminfo"):
        if ln.startswith("MemAvailable:"):
            return int(ln.split()[1]) / 1048576.0
    return 0.0


def test_config3_at_its_real_shape_eight_ranks_on_this_gpu():
    """BASELINE config 3 itself -- 27-pt 512^3, KSPGMRES(30) + PCSOR over EIGHT ranks (16.8 M-row slabs of 451 M nonzeros each: 8 x 5.8 GB of CSR fit this
    one MI355X) -- run functionally: the 8-rank split, garray / ghost lists of 2 x 512^2 values per interior rank, MatMult_MPIAIJ's order (diagonal block, then the
    ghost terms added), per-rank local symmetric sweeps, GMRES's 30-vector orthogonalisation with the ranks' sums folded as unrounded pairs.  The yardstick
    (tests/golden/exact_histories.json[gmres_sor_27pt_512_np8], round 5) is the C oracle's GMRES loop with exact reductions over products and local sweeps
    streamed slab by slab (oracle/stream_gmres.py: 3.6e9 nonzeros do not fit the build container as one matrix); 35 iterations = one restart.  VERDICT r4 item 2.
    The ranks time-slice one device over the IPC transport: a functional run, not a scaling measurement."""
    if _mem_available_gb() < 170:
        pytest.skip("the 8 ranks assemble 8 x 5.4 GB slabs (and split them) on the host: needs ~170 GB of RAM, this box has %.0f GB available" % _mem_available_gb())
    env = dict(clean_env(), HIPX_ALL_RANKS_DEVICE0="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--ksp", "gmres", "--pc", "sor", "--stencil", "27", "--grid", "512", "--steps", "35", "--warmup", "2",
                        "--quick", "--transport", "ipc", "--parity-its", "35"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:]
    d = last_json(r.stdout)
    g = d["parity_gate"]
    assert d["n_gpus"] == 8 and d["config"]["global_rows"] == 512 ** 3 and d["config"]["transport"] == "ipc"
    assert g["pass"] is True and g["gated_reduction_mode"] == "exact" and g["iterations"] == 35 and g["entries"] == 36 and g["max_rel_diff"] <= 1e-12, g
    assert g["max_rel_diff_fast_reductions"] <= 1e-8
    assert len(d["per_rank"]) == 8 and all(p["rows"] == 512 ** 3 // 8 for p in d["per_rank"])
    assert [p["ghosts"] for p in d["per_rank"]] == [512 * 512] + [2 * 512 * 512] * 6 + [512 * 512]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump({"what": "config 3 at its real shape: 8 ranks time-slicing ONE MI355X (IPC transport)", "parity_gate": g, "iterations_per_s": d["value"], "ms_per_step": d["ms_per_step"],
               "per_rank": d["per_rank"]}, open(os.path.join(ROOT, "gpurun_out", "config3_real_shape_np8_one_gpu.json"), "w"), indent=1)


def test_headline_256_on_eight_ranks_on_this_gpu():
    """The headline system (7-pt 256^3, KSPCG + PCJACOBI) on eight ranks sharing this GPU: the 8-rank plan at BASELINE shape against the committed history of
    the reference + exact BLAS (cg_jacobi_7pt_256), in the default reduction mode at 1e-12."""
    env = dict(clean_env(), HIPX_ALL_RANKS_DEVICE0="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "60", "--warmup", "5", "--quick", "--transport", "ipc", "--parity-its", "50"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:]
    d = last_json(r.stdout)
    g = d["parity_gate"]
    assert d["n_gpus"] == 8 and g["pass"] is True and g["iterations"] == 50 and g["max_rel_diff"] <= 1e-12, g
    assert [p["ghosts"] for p in d["per_rank"]] == [256 * 256] + [2 * 256 * 256] * 6 + [256 * 256]
