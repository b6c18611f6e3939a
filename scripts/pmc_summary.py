"""Summarise rocprofv3 PMC passes of bench.py into profiles/spmv_traffic.json (what bench.py reports as roofline.traffic).
  python scripts/pmc_summary.py <fetch_dir> <write_dir> <key> [--kernel spmv_vd_kernel]
<fetch_dir>/<write_dir> hold pmc_counter_collection.csv of `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs
(separate passes, --kernel-trace only).  Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md:
FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE counts 64 B per 128 B request on gfx950 for wide streams (calibrated on
our AXPY: 2 x 134 MB read -> FETCH_SIZE = half), so read bytes = 2 x FETCH_SIZE x 1024."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_launch(path, counter, kernel):
    vals = {}
    for row in csv.DictReader(open(os.path.join(path, "pmc_counter_collection.csv"))):
        if row["Counter_Name"] == counter and kernel in row["Kernel_Name"]:
            vals.setdefault(row["Dispatch_Id"], 0.0)
            vals[row["Dispatch_Id"]] += float(row["Counter_Value"])
    v = list(vals.values())
    return (sum(v) / len(v), len(v)) if v else (None, 0)


def sor_main(fetch_dir, write_dir):
    """--sor: bytes per launch of the strand SOR kernels (forward = KIND 0, backward = KIND 1 instantiations) and of the AXPY that
    calibrates the FETCH_SIZE unit in the same pass."""
    def find(d):
        for root, _, files in os.walk(d):
            for fn in files:
                if fn.endswith("counter_collection.csv"):
                    return os.path.join(root, fn)
        raise SystemExit("no counter_collection.csv under " + d)
    out = {}
    for counter, d in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir)):
        acc = {}
        for row in csv.DictReader(open(find(d))):
            if row["Counter_Name"] != counter:
                continue
            kn = row["Kernel_Name"]
            if "sor_strand_kernel<0" in kn or "sor_strand_kernelILi0" in kn:
                key = "sor_strand_kernel forward (KIND 0)"
            elif "sor_strand_kernel<1" in kn or "sor_strand_kernelILi1" in kn:
                key = "sor_strand_kernel backward (KIND 1)"
            elif "sor_dep_kernel" in kn:
                key = "sor_dep_kernel"
            elif "sor_fill_kernel" in kn:
                key = "sor_fill_kernel"
            else:
                continue
            acc.setdefault(key, {}).setdefault(row["Dispatch_Id"], 0.0)
            acc[key][row["Dispatch_Id"]] += float(row["Counter_Value"])
        for key, v in acc.items():
            vals = list(v.values())
            out.setdefault(key, {})[counter + "_KiB_per_launch"] = sum(vals) / len(vals)
            out[key]["launches_sampled"] = len(vals)
    for key, v in out.items():
        f, w = v.get("FETCH_SIZE_KiB_per_launch"), v.get("WRITE_SIZE_KiB_per_launch")
        if f is not None and w is not None:
            v["traffic_bytes_per_launch"] = int(2 * f * 1024 + w * 1024)  # same correction as for the SpMV (see the header)
    print(json.dumps(out, indent=1))


def table_main(path):
    """--table <counter_collection.csv>: mean per launch of every counter of every kernel of one rocprofv3 --pmc pass (what scripts/gpu_run.sh
    prints for its `pmc:` steps)."""
    acc, names = {}, set()
    for row in csv.DictReader(open(path)):
        kn = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        kn = kn.split("(")[0][:90]
        acc.setdefault(kn, {}).setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
        acc[kn][row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
        names.add(row["Counter_Name"])
    names = sorted(names)
    print("%-92s %6s " % ("kernel", "n") + " ".join("%18s" % c[:18] for c in names))
    for kn in sorted(acc):
        n = max(len(v) for v in acc[kn].values())
        print("%-92s %6d " % (kn, n) + " ".join("%18.1f" % (sum(acc[kn][c].values()) / len(acc[kn][c])) if c in acc[kn] else "%18s" % "-" for c in names))


def main():
    if sys.argv[1] == "--table":
        return table_main(sys.argv[2])
    if sys.argv[1] == "--sor":
        return sor_main(sys.argv[2], sys.argv[3])
    fetch_dir, write_dir, key = sys.argv[1:4]
    kernel = sys.argv[sys.argv.index("--kernel") + 1] if "--kernel" in sys.argv else "spmv_"
    f, nf = per_launch(fetch_dir, "FETCH_SIZE", kernel)
    w, nw = per_launch(write_dir, "WRITE_SIZE", kernel)
    p = os.path.join(ROOT, "profiles", "spmv_traffic.json")
    tj = json.load(open(p)) if os.path.exists(p) else {}
    tj[key] = {"kernel": kernel, "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "launches_sampled": min(nf, nw),
               "traffic_bytes": int(2 * f * 1024 + w * 1024)}
    json.dump(tj, open(p, "w"), indent=1)
    print(key, tj[key])


if __name__ == "__main__":
    main()
