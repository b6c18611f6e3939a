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
