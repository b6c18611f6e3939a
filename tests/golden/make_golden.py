"""Regenerates tests/golden/*.json.  Run in the build container (needs /root/reference):
      python tests/golden/make_golden.py
  * ref_outputs.json  -- known-answer outputs of the reference's own tests for this path, parsed from
    /root/reference/src/ksp/ksp/tutorials/output/{ex2_3,bench_kspsolve_matmult,bench_kspsolve_ksp}.out
    (residual histories / iteration counts / error norms printed with %g, i.e. 6 significant digits).
  * ref_runs.json     -- outputs of the reference library itself (oracle/_ref, built by oracle/build_ref.py)
    run here with -ksp_monitor at full precision, when oracle/_ref exists (see oracle/README.md).
"""
import json
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src/ksp/ksp/tutorials/output"


def parse_monitor(path):
    hist, tail = [], None
    for line in open(path):
        m = re.match(r"\s*(\d+) KSP Residual norm (\S+)", line)
        if m:
            hist.append(float(m.group(2)))
        m = re.match(r"Norm of error (\S+) iterations (\d+)", line)
        if m:
            tail = {"error": float(m.group(1)), "iterations": int(m.group(2))}
    return hist, tail


def main():
    out = {}
    # ex2 suffix 3: -pc_type sor -pc_sor_symmetric -ksp_monitor -ksp_gmres_cgs_refinement_type refine_always (ex2.c TEST block),
    # default m = 8, n = 7, GMRES(30), rtol = 1e-2/((m+1)(n+1))
    h, t = parse_monitor(os.path.join(REF, "ex2_3.out"))
    out["ex2_3"] = {"args": "-pc_type sor -pc_sor_symmetric -ksp_monitor -ksp_gmres_cgs_refinement_type refine_always", "m": 8, "n": 7, "history": h, **t}
    for name in ("bench_kspsolve_matmult", "bench_kspsolve_ksp"):
        txt = open(os.path.join(REF, name + ".out")).read()
        out[name] = {"dofs": int(re.search(r"DoFs = (\d+)", txt).group(1)), "nnz": int(re.search(r"Number of nonzeros = (\d+)", txt).group(1)),
                     "n": int(re.search(r"-n (\d+)", txt).group(1))}
        m = re.search(r"Error norm:\s+(\S+)", txt)
        if m:
            out[name]["error_norm"] = float(m.group(1))
        m = re.search(r"KSP iters:\s+(\d+)", txt)
        if m:
            out[name]["iterations"] = int(m.group(1))
    json.dump(out, open(os.path.join(HERE, "ref_outputs.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
