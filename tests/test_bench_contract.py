"""CPU: the bench.py contract (the driver parses ONE JSON line).  Checks the committed line of the last GPU run
(profiles/r01f_bench_default.json, produced by `python bench.py` on an MI355X) and the command-line defaults."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_module(name="bench_mod0"):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    return b


def _check_contract_line(d):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "7-pt Poisson 256^3" in d["metric"] and d["unit"] == "iterations/s" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["ms_per_step"] * d["value"] - 1000.0) < 1e-5 * 1000.0
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5
    assert abs(r["effective_gbps"] - r["algorithmic_bytes"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-4 * r["effective_gbps"]
    if r["traffic"]:  # achieved / frac are on the HBM bytes the kernel really moves (PMC), the algorithmic figure is effective_gbps
        assert abs(r["achieved"] - r["traffic"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-4 * r["achieved"] and r["frac"] < 1.0
    assert d["value"] is None or d["parity_gate"]["pass"] is not False
    assert r["algorithmic_bytes"] in (12 * 117047296 + 4 * (16777216 + 1) + 16 * 16777216, 12 * 117047296 + 4 * (16777216 + 1) + 16 * 16777216 + 48 * 16777216)  # SURVEY.md 8(d), config 2 (+ the direction update when it is the product's prologue)
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0


def test_committed_bench_line_is_compact_and_has_every_contract_field():
    """The line of the last GPU run as the driver sees it: the LAST stdout line, < 4 KB (round 4's 25 KB line left BENCH_r04.parsed null)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_default.json")))
    assert files, "no committed bench line under profiles/"
    text = open(files[-1]).read()
    last = text.strip().splitlines()[-1]
    d = json.loads(text[-8000:].splitlines()[-1]) if os.path.basename(files[-1]) >= "r05" else json.loads(last)
    if os.path.basename(files[-1]) >= "r05":
        assert len(last) < 4096, len(last)
        _check_contract_line(d)
    else:  # an older (uncompacted) line: what compact_line makes of it is what the driver would get today
        c = _bench_module().compact_line(d)
        assert len(json.dumps(c)) < 4096
        _check_contract_line(c)


def test_compact_line_stays_under_the_limit_and_emit_prints_it_last(tmp_path, capsys, monkeypatch):
    """compact_line() of a full result (round 4's 25 KB line, every leg and counter detail present) is < 4 KB and keeps the contract; emit()
    writes the full result to bench_detail.json and the compact line as the last stdout line, parseable from the last 8000 bytes."""
    b = _bench_module("bench_mod_c")
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_default.json")))
    assert len(json.dumps(full)) > 20000
    c = b.compact_line(full)
    assert len(json.dumps(c)) < 4096
    _check_contract_line(c)
    assert set(c["other_configs"]) == set(full["other_configs"]) and all("it_s" in v or "frac" in v for v in c["other_configs"].values())
    assert [k[0] for k in c["roofline"]["by_kernel"]][:2] == ["spmv_march2_kernel<7, 8, 1, true, true, 256>", "cg_fused_kernel<false, true, true, false, true>"]
    # a pathological result (hundreds of legs) still fits: optional groups are dropped, the contract fields never
    fat = dict(full, other_configs={("leg%03d" % i): {"iterations_per_s": 1.0 + i, "parity": {"pass": True, "max_rel_diff": 1e-15}} for i in range(400)},
               per_rank=[{"rows": 1, "ghosts": 2, "spmv_ms": 0.1}] * 8)
    cf = b.compact_line(fat)
    assert len(json.dumps(cf)) < 4096
    _check_contract_line(cf)
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    print("noise before the line " * 50)
    b.emit(dict(full), None)
    outp = capsys.readouterr().out
    d = json.loads(outp[-8000:].splitlines()[-1])
    _check_contract_line(d)
    assert d["detail"] == "bench_detail.json" and json.load(open(tmp_path / "bench_detail.json"))["roofline"]["traffic_source"]


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
            b.Cfg(7, (1024, 1024, 256), "cg", "none").golden_key(), b.Cfg(5, (4096, 4096, 1), "cg", "jacobi").golden_key(), "gmres_sor_27pt_256_np1", "gmres_sor_27pt_256_np8", "cg_jacobi_7pt_64", "cg_none_7pt_64x64x16"]
    for k in need:
        assert k in g, k
        e = g[k]
        assert len(e["history"]) == len(e["history_hex"]) >= 13 and all(float.fromhex(h) == v for h, v in zip(e["history_hex"], e["history"]))
    c5 = b.Cfg(5, (4096, 4096, 1), "cg", "jacobi")  # round 6: north_star's 5-point leg = ex2.c's operator (config 1) at HBM size
    assert c5.golden_key() == "cg_jacobi_5pt_4096x4096x1" and c5.shape() == "4096x4096" and c5.driver_args(3)[:4] == ["-stencil", "5", "-n", "4096"] and "-m" in c5.driver_args(3) and "-nz" not in c5.driver_args(3)
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
