#!/usr/bin/env python3
"""profiles/valu.json from one rocprofv3 SQ counter pass of the bench command (tools/refresh_profiles.sh, pass sq1:
SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU ... with --kernel-trace): what bench.py's `roofline.valu_frac`
is computed from.

usage: python tools/derive_valu.py <label> <E> <A> <T> <counter_collection.csv> <kernel_trace.csv> [...]   (groups of six)
       label = the rollout kernel as the library reports it for that batch (bench line `roofline.kernel`)

Per kernel instance and batch: VALU wave-instructions per launch (SQ_INSTS_VALU), the share of a SIMD's cycles during
which its resident waves had a vector instruction in flight -- SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (both quad-cycles
summed over waves) x waves resident per SIMD (SQ_WAVES / 1024 SIMDs, at most 8) -- and the mean duration of the launches
that pass measured.  The same instructions in a shorter live launch fill proportionally more of the issue slots, so
bench.py reports valu_frac = busy_share x pass duration / live duration."""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMDS = 256 * 4


def rollout_means(path):
    vals = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if "rollout_kernel" in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for c, v in vals.items():
        full = [x for x in v if x >= 0.5 * max(v)]          # (the bench also launches one short rollout: the parity leg)
        out[c] = sum(full) / len(full)
    return out


def rollout_duration_ms(path):
    d = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if "rollout_kernel" in r["Kernel_Name"]:
                d.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    full = [x for x in d if x >= 0.5 * max(d)]
    return sum(full) / len(full), len(full)


def main():
    args = sys.argv[1:]
    dst = os.path.join(ROOT, "profiles", "valu.json")
    try:
        with open(dst) as f:
            doc = json.load(f)
    except (OSError, ValueError):
        doc = {"kernels": []}
    doc["_how"] = " ".join(__doc__.strip().split("\n\n")[2].split())
    while len(args) >= 6:
        label, E, A, T, counters, trace = args[:6]
        args = args[6:]
        E, A, T = int(E), int(A), int(T)
        m = rollout_means(counters)
        ms, n = rollout_duration_ms(trace)
        per_simd = min(8.0, m["SQ_WAVES"] / SIMDS)
        entry = {"kernel": label, "n_envs": E, "n_agents": A, "env_steps_per_launch": T,
                 "valu_insts_per_launch": m["SQ_INSTS_VALU"], "valu_insts_per_wave_step": m["SQ_INSTS_VALU"] / m["SQ_WAVES"] / T,
                 "waves": m["SQ_WAVES"], "waves_per_simd": per_simd,
                 "valu_active_share_of_wave_cycles": m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"],
                 "valu_busy_share_of_simd_cycles": m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"] * per_simd,
                 "launch_ms_in_pass": ms, "launches_in_pass": n}
        doc["kernels"] = [k for k in doc["kernels"] if not (k["kernel"] == label and k["n_envs"] == E and k["n_agents"] == A)] + [entry]
        print(json.dumps(entry, indent=1))
    with open(dst, "w") as f:
        json.dump(doc, f, indent=1)


if __name__ == "__main__":
    main()
