"""Derive profiles/traffic.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of `python3 bench.py
--no-cpu-baseline`.  usage: python tools/derive_traffic.py <fetch counter_collection.csv> <write ...csv> [out.json]

hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per dispatch: the counters are KB per dispatch and gfx950 tallies
128-byte read requests at 64 B (MI355X_MICROARCH.md, HBM / rocprofv3 section).  reset_kernel and
fill_actions_kernel move known byte counts and are kept in the output as the calibration of that rule.
"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
E, A, T = 65536, 8, 256   # bench.py defaults
BYTES_PER_AGENT_STEP = 5.0 + 18.0 / A   # SURVEY.md 8(d), same figure as bench.py bytes_per_agent_step
ALGORITHMIC = {"rollout_kernel": int(T * E * A * BYTES_PER_AGENT_STEP), "lg_step_kernel": int(E * A * BYTES_PER_AGENT_STEP)}


def per_kernel(path, counter):
    """{kernel key: (mean counter value, dispatches, full name)}.  The bench also launches one short rollout (the
    8-step parity leg): dispatches whose value is below half of the kernel's largest are left out of the mean."""
    values, names = collections.defaultdict(list), {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"]
            for key in ("rollout_kernel", "lg_step_kernel", "reset_kernel", "fill_actions_kernel"):
                if key in name:   # "rollout_kernel" matches the quad-lane (lq_) and the pair (lg_) layout
                    values[key].append(float(r["Counter_Value"]))
                    names[key] = name
    out = {}
    for k, v in values.items():
        full = [x for x in v if x >= 0.5 * max(v)] if k == "rollout_kernel" else v
        out[k] = (sum(full) / len(full), len(full), names[k])
    return out


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"_how": __doc__.strip().split("\n\n", 1)[1].replace("\n", " "), "kernels": {}}
    for k in ("rollout_kernel", "lg_step_kernel", "reset_kernel", "fill_actions_kernel"):
        if k not in fetch or k not in write:
            continue
        f_kb, n, name = fetch[k]
        w_kb = write[k][0]
        entry = {"instance": name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", ""), "dispatches": n,
                 "FETCH_SIZE_KB": round(f_kb, 2), "WRITE_SIZE_KB": round(w_kb, 2),
                 "hbm_bytes_per_launch": int(round((2 * f_kb + w_kb) * 1024))}
        if k in ALGORITHMIC:
            entry["algorithmic_bytes_per_launch"] = ALGORITHMIC[k]
        if k == "rollout_kernel":
            entry["steps_per_launch"] = T
        out["kernels"][k] = entry
    dst = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles", "traffic.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out["kernels"], indent=1))


if __name__ == "__main__":
    main()
