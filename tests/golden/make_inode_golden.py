#!/usr/bin/env python
"""Generates tests/golden/inode_sor.json: outputs of the REFERENCE's own MatSOR on a matrix with inodes (aij.c:1852 ->
MatSOR_SeqAIJ_Inode, inode.c:2494-3810) and its KSPCG + PCSOR history on a small blocked operator, from oracle/_ref/bin/ref_driver
(-dump_sor / -history, the latter with oracle/libexactblas.so preloaded).  Run in the build container (CPU only, needs oracle/_ref):

    python tests/golden/make_inode_golden.py

The matrices come from tests/surrogates.py (inode_matrix: nodes of 1-5 rows, one run of 7 identical rows, diagonal blocks that need row
interchanges; flan_surrogate_spd(n=8): 3 unknowns per node), written to PETSc binary files for MatLoad; b = A * 1 (exact in any
summation order for inode_matrix: its values are multiples of 2^-10).  Values are stored as C99 hex floats.
tests/test_oracle.py holds the oracle to these bit for bit; tests/test_gpu_inode.py holds the HIP path to the oracle."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
from surrogates import flan_surrogate_spd, inode_matrix  # noqa: E402
from petsc_amd import matio  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "inode_sor.json")
REF = os.path.join(ROOT, "oracle", "_ref", "bin", "ref_driver")
SHIM = os.path.join(ROOT, "oracle", "libexactblas.so")
CASES = [(16 | 1, 1, 1), (16 | 2, 1, 1), (16 | 3, 1, 1), (16 | 12, 1, 1), (16 | 4, 1, 1), (16 | 8, 1, 1), (3, 1, 1), (1, 1, 1), (2, 1, 1), (12, 2, 1), (16 | 3, 3, 1),
         (16 | 1, 2, 1), (16 | 2, 2, 1), (16 | 12, 1, 2), (32, 1, 1), (2, 2, 3)]  # (MatSORType bits, its, lits)


def run(args, exact=False):
    env = dict(os.environ, MKL_NUM_THREADS="1", OMP_NUM_THREADS="1")
    if exact:
        env["LD_PRELOAD"] = SHIM
    return subprocess.run([REF] + args + ["-mat_type", "aij", "-vec_type", "standard"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=600).stdout


def main():
    tmp = tempfile.mkdtemp(prefix="inode_golden_")
    out = {"what": "the reference's MatSOR / KSPCG+PCSOR on matrices with inodes (oracle/_ref/bin/ref_driver); see make_inode_golden.py", "sor": {}, "ksp": {}}
    ai, aj, aa = inode_matrix()
    f = os.path.join(tmp, "inode.bin")
    matio.write_petsc_binary(f, ai, aj, aa)
    for flag, its, lits in CASES:
        for noin in (0, 1):
            if noin and (flag, its, lits) != (16 | 12, 1, 1):
                continue
            o = run(["-f", f, "-dump_sor", str(flag), "-sor_its", str(its), "-sor_lits", str(lits), "-ksp_max_it", "1"] + (["-mat_no_inode"] if noin else []))
            x = [float(l.split()[2]) for l in o.splitlines() if l.startswith("sor ")]
            assert len(x) == len(ai) - 1, o[-2000:]
            out["sor"]["flag%d_its%d_lits%d%s" % (flag, its, lits, "_noinode" if noin else "")] = [v.hex() for v in x]
    out["sor_matrix"] = {"generator": "tests/surrogates.py inode_matrix()", "rows": int(len(ai) - 1), "nnz": int(ai[-1]), "x0": "0.5 + (i mod 7) / 7 (used without SOR_ZERO_INITIAL_GUESS)", "b": "A * 1"}
    ai, aj, aa = flan_surrogate_spd(n=8)
    f = os.path.join(tmp, "flan8.bin")
    matio.write_petsc_binary(f, ai, aj, aa)
    for key, extra in (("cg_sor_flan8", []), ("cg_sor_flan8_noinode", ["-mat_no_inode"]), ("gmres_sor_flan8", [])):
        ksp = "gmres" if key.startswith("gmres") else "cg"
        a = ["-f", f, "-ksp_type", ksp, "-pc_type", "sor", "-ksp_rtol", "1e-50", "-ksp_max_it", "12", "-history"] + (["-ksp_norm_type", "preconditioned"] if ksp == "cg" else []) + extra
        o = run(a, exact=True)
        h = [float(l.split()[2]) for l in o.splitlines() if l.startswith("hist ")]
        assert len(h) == 13, o[-2000:]
        out["ksp"][key] = {"history_hex": [v.hex() for v in h], "source": "reference+shim", "matrix": "flan_surrogate_spd(n=8): %d rows" % (len(ai) - 1)}
    json.dump(out, open(OUT, "w"), indent=0)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
