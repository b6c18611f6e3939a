"""Measured parity margins.  Every history comparison records the largest per-entry relative difference it saw, so the
tolerances written in the tests can be checked against what the hardware actually produced: the records go to
gpurun_out/parity_measured.json (merged back from the GPU box) and are summarised in profiles/README.md."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "gpurun_out", "parity_measured.json")


def record(name, value, bound):
    try:
        os.makedirs(os.path.dirname(PATH), exist_ok=True)
        d = json.load(open(PATH)) if os.path.exists(PATH) else {}
        d[name] = {"max_rel_diff": float(value), "asserted_bound": float(bound)}
        json.dump(d, open(PATH, "w"), indent=1, sort_keys=True)
    except OSError:
        pass
