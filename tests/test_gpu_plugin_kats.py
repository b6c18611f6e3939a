"""GPU: the reference's OWN known-answer tests for this path, run unmodified with the plugin types and held to the
reference's golden outputs (tests/golden/kats.json, copied from src/*/output/*.out by tests/golden/make_golden.py).
This is exactly how the reference tests its device back ends (same test, `-vec_type hip`, same golden):
  src/vec/vec/tutorials/ex1.c ("exact numbers are critical", diff_args -j), tests/ex43.c (MDot/Dot/MTDot/TDot),
  ex34.c (norm cache), ex60.c (PlaceArray/Reciprocal), ex21.c (Max with index), ex28/31/52/63, src/mat/tests/ex5.c
  (MatMult / MultAdd / MultTranspose, seq and mpi types on one rank)."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "bin")
PLUGIN = os.path.join(ROOT, "petsc_amd", "lib", "libpetschipx.so")
K = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kats.json")))


def apply_filter(kind, text):
    lines = text.splitlines()
    if kind == "notype":  # TEST block: filter: grep -v type; no `diff_args: -j`, so the harness (lib/petsc/bin/petscdiff) ignores
        # white-space differences: compare token streams per line
        lines = [" ".join(ln.split()) for ln in lines if "type" not in ln]
    elif kind == "ex21":  # grep -v type | grep -v " MPI process" | grep -v Process
        lines = [ln for ln in lines if "type" not in ln and " MPI process" not in ln and "Process" not in ln]
    elif kind == "seqname":  # sed -e 's/seqhip/seq/' in the reference; our type is seqhipx
        lines = [ln.replace("seqhipx", "seq") for ln in lines]
    elif kind == "ex123":  # grep -v type | grep -v "Mat Object"; diff_args -j = plain `diff -w` (lib/petsc/bin/petscdiff:73): numbers exact,
        # white space not significant (MatView of MPIAIJ indents its rows, the golden was written by SeqAIJ)
        lines = [" ".join(ln.split()) for ln in lines if "type" not in ln and "Mat Object" not in ln]
    elif kind == "monitor":  # the golden was written with %g; today's monitor prints 14 digits: reformat the numbers
        import re
        out = []
        for ln in lines:
            m = re.match(r"(\s*\d+ KSP Residual norm )(\S+)\s*$", ln)
            out.append(m.group(1) + "%g" % float(m.group(2)) if m else ln)
        lines = out
    return "\n".join(lines).strip()


@pytest.mark.parametrize("name", sorted(K))
def test_reference_kat_with_hipx_types(name):
    k = K[name]
    exe = os.path.join(BIN, k["exe"])
    assert os.path.exists(exe) and os.path.exists(PLUGIN), "oracle/_ref or the plugin is not built"
    args = k["args"].replace("-mat_type seqaij", "-mat_type seqaijhipx").replace("-mat_type mpiaij", "-mat_type mpiaijhipx").split()
    cmd = [exe] + args + ["-dll_prepend", PLUGIN, "-vec_type", "hipx"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120, env=dict(os.environ, HIPX_NO_TORCH="1"))
    assert r.returncode == 0, r.stdout[-2000:]
    got = apply_filter(k["filter"], r.stdout)
    want = apply_filter(k["filter"], k["golden"])
    assert got == want, "KAT %s (%s) differs from %s:\n--- got\n%s\n--- want\n%s" % (name, " ".join(cmd), k["golden_file"], got[-1500:], want[-1500:])
