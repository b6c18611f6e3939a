#!/usr/bin/env python
"""Kernel timeline of the last iterations of a rocprofv3 --kernel-trace CSV: per kernel launch its start (us after the first kernel shown), duration and queue.
    python scripts/trace_timeline.py <kernel_trace.csv> [--last 60] [--stats]"""
import argparse
import csv
import re
from collections import defaultdict


def short(kn):
    kn = re.sub(r"^void\s+", "", kn.strip().strip('"')).replace("(anonymous namespace)::", "")
    depth, out = 0, []
    for ch in kn:
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).strip()[:70]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--last", type=int, default=60)
    ap.add_argument("--stats", action="store_true")
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.csv)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    if a.stats:
        acc = defaultdict(list)
        for r in rows[len(rows) // 3:]:
            acc[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            print("%-72s n %5d  avg %9.2f us  min %9.2f  max %9.2f" % (k, len(v), sum(v) / len(v), min(v), max(v)))
    sel = rows[-a.last:]
    t0 = int(sel[0]["Start_Timestamp"])
    prev_end = {}
    for r in sel:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        q = r.get("Queue_Id", "?")
        gap = (s - prev_end[q]) / 1e3 if q in prev_end else 0.0
        prev_end[q] = e
        print("%10.2f us  +%8.2f us  (queue %s, gap %7.2f)  %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, gap, short(r["Kernel_Name"])))


if __name__ == "__main__":
    main()
