#!/bin/bash
# rocprofv3 evidence for the device-to-device ghost exchange under the plugin: two MPI ranks (sharing the box's GPU, IPC transport),
# the reference's MatMult_MPIAIJ call path over MATMPIAIJHIPX, 10 vs 110 products: the number and size of memory copies must not grow
# with the number of products.  Usage: bash scripts/gpu_np2_trace.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 HIPX_NO_TORCH=1
R="$GRAFT_REPO_ROOT"; T=${1:-np2}
MPIEXEC=/opt/conda/bin/mpiexec; BIN="$R/oracle/_ref/mpich/bin"; PLUG="$R/petsc_amd/lib/libpetschipx_mpich.so"
cd /tmp
for its in 10 110; do
  rm -rf "$R/gpurun_out/${T}_its$its"
  timeout 300 $MPIEXEC -n 2 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d "$R/gpurun_out/${T}_its$its" -o r -- \
    "$BIN/ref_driver" -stencil 7 -n 128 -matmult_its $its -ksp_max_it 1 -dll_prepend "$PLUG" -vec_type hipx -mat_type aijhipx > "$R/gpurun_out/${T}_its$its.log" 2>&1
  find "$R/gpurun_out/${T}_its$its" -name "*_trace.csv" -delete
done
cd "$R"
python - "$T" <<'PY'
import csv, glob, json, os, sys
T = sys.argv[1] if len(sys.argv) > 1 else "np2"
out = {}
for its in (10, 110):
    tot = {}
    for f in glob.glob("gpurun_out/%s_its%d/**/*memory_copy_stats.csv" % (T, its), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Name"]
            tot.setdefault(k, [0, 0.0])
            tot[k][0] += int(row["Calls"])
            tot[k][1] += float(row["TotalDurationNs"]) / 1e6
    ker = {}
    for f in glob.glob("gpurun_out/%s_its%d/**/*kernel_stats.csv" % (T, its), recursive=True):
        for row in csv.DictReader(open(f)):
            nm = row["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            nm = nm.split("(")[0].split("<")[0]
            if any(s in nm for s in ("ipc_", "halo", "spmv", "pack")):
                ker[nm] = ker.get(nm, 0) + int(row["Calls"])
    out["matmult_its_%d" % its] = {"memory_copies (both ranks): name -> [calls, total ms]": tot, "kernels (both ranks): calls": ker}
json.dump(out, open("gpurun_out/%s_summary.json" % T, "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
