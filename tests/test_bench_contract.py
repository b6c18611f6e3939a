"""CPU: the bench.py contract (the driver parses ONE JSON line).  Checks the committed line of the last GPU run
(profiles/r01f_bench_default.json, produced by `python bench.py` on an MI355X) and the command-line defaults."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_every_contract_field():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_default.json")))
    assert files, "no committed bench line under profiles/"
    d = json.loads(open(files[-1]).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "7-pt Poisson 256^3" in d["metric"] and d["unit"] == "iterations/s" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["ms_per_step"] * d["value"] - 1000.0) < 1e-6 * 1000.0
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    if "effective_gbps" in r:  # round 2 on: achieved / frac are on the HBM bytes the kernel really moves, the algorithmic figure is effective_gbps
        assert abs(r["effective_gbps"] - r["algorithmic_bytes"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["effective_gbps"]
        if r["traffic"]:
            assert abs(r["achieved"] - r["traffic"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"] and r["frac"] < 1.0
            assert r["traffic_source"]
        assert d["value"] is None or d["parity_gate"]["pass"] is not False
    else:
        assert abs(r["achieved"] - r["algorithmic_bytes"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert r["algorithmic_bytes"] == 12 * 117047296 + 4 * (16777216 + 1) + 16 * 16777216
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0


def test_bench_cli_defaults():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120,
                         env=dict(os.environ, HIPX_NO_TORCH="1")).stdout
    for flag in ("--gpus", "--steps", "--warmup", "--grid", "--stencil", "--fused", "--pipeline", "--variant", "--ksp", "--pc", "--scaling"):
        assert flag in out, flag


def test_bench_goldens_cover_the_configurations_it_gates():
    """bench.py's N > 1 parity gate and its other_configs legs look their yardstick up in tests/golden/exact_histories.json by
    these keys; every history has its exact hex form beside the decimals."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "exact_histories.json")))
    head = b.Cfg(7, (256, 256, 256), "cg", "jacobi")
    assert head.golden_key() == "cg_jacobi_7pt_256" and head.metric() == "CG iterations/sec, 7-pt Poisson 256^3 fp64, KSPCG+PCJACOBI"
    need = [head.golden_key(), b.Cfg(27, (512, 512, 512), "cg", "jacobi").golden_key(), b.Cfg(7, (1024, 1024, 128), "cg", "none").golden_key(),
            b.Cfg(7, (1024, 1024, 256), "cg", "none").golden_key(), "gmres_sor_27pt_256_np1", "gmres_sor_27pt_256_np8", "cg_jacobi_7pt_64", "cg_none_7pt_64x64x16"]
    for k in need:
        assert k in g, k
        e = g[k]
        assert len(e["history"]) == len(e["history_hex"]) >= 13 and all(float.fromhex(h) == v for h, v in zip(e["history_hex"], e["history"]))
    assert b.Cfg(7, (1024, 1024, 128), "cg", "none").driver_args(5)[:4] == ["-stencil", "7", "-n", "1024"] and "-nz" in b.Cfg(7, (1024, 1024, 128), "cg", "none").driver_args(5)


def test_bench_self_launch_command(monkeypatch):
    """`python bench.py --gpus N` without RANK/WORLD_SIZE re-runs itself under torch.distributed.run with the driver's own recipe."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    seen = {}
    monkeypatch.setattr(b.subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7"])

    class A:
        gpus = 4
    assert b.self_launch(A) == 0
    c = seen["cmd"]
    assert c[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and c[c.index("--nproc-per-node") + 1] == "4" and c[c.index("--master-addr") + 1] == "127.0.0.1"
    assert c[-4:] == ["--gpus", "4", "--steps", "7"] and seen["env"]["HIPX_SELF_LAUNCHED"] == "1" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
