"""Derive profiles/traffic.json from rocprofv3 PMC passes of `python3 bench.py --no-cpu-baseline` run at TWO launch
lengths (--rollout-steps T1 and T2): per kernel instance and batch, HBM bytes per launch = fixed + per_env_step * T.

usage: python tools/derive_traffic.py [--merge] <label> <E> <A> <T1> <fetch1.csv> <write1.csv> <T2> <fetch2.csv> <write2.csv> [...]
       (label = "<rollout kernel>||<single-step kernel>" as the library reports them for that batch: bench line
        `roofline.kernel` and `single_step_launches.kernel`; a group of nine arguments per configuration)

hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per dispatch: the counters are KB per dispatch and gfx950 tallies
128-byte read requests at 64 B (MI355X_MICROARCH.md, HBM / rocprofv3 section).  reset_kernel and fill_actions_kernel
move known byte counts and are kept in the output as the calibration of that rule.
"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def per_kernel(path, counter):
    """{kernel key: (mean counter value over the full-length dispatches, dispatches, full name)}.  The bench also
    launches one short rollout (the 8-step parity leg): rollout dispatches below half of the largest are left out."""
    values, names = collections.defaultdict(list), {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"]
            for key in ("rollout_kernel", "step_kernel", "reset_kernel", "fill_actions_kernel"):
                if key in name and not (key == "step_kernel" and "lg_step_kernel" not in name and "mapf::step_kernel<2," in name):   # (the scalar_env leg runs step_kernel<2, true> on ONE env: not a batch kernel)
                    values[key].append(float(r["Counter_Value"]))
                    names[key] = name
                    break
    out = {}
    for k, v in values.items():
        full = [x for x in v if x >= 0.5 * max(v)] if k == "rollout_kernel" else v
        out[k] = (sum(full) / len(full), len(full), names[k])
    return out


def hbm_bytes(fetch, write, key):
    return (2.0 * fetch[key][0] + write[key][0]) * 1024.0


def main():
    import bench
    csrc = bench.csrc_hash()          # the kernel sources these passes ran (bench.py uses an entry only for the same sources)
    args = sys.argv[1:]
    merge = bool(args) and args[0] == "--merge"   # keep what profiles/traffic.json holds for batches not named in this call
    if merge:
        args = args[1:]
    out = {"_how": __doc__.strip().split("\n\n")[2].replace("\n", " "), "kernels": [], "calibration": {}}
    old = {"kernels": [], "calibration": {}}
    if merge:
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                old = json.load(f)
        except (OSError, ValueError):
            pass
    while len(args) >= 9:
        label, E, A, T1, f1, w1, T2, f2, w2 = args[:9]
        label, _, step_label = label.partition("||")
        args = args[9:]
        E, A, T1, T2 = int(E), int(A), int(T1), int(T2)
        F1, W1, F2, W2 = per_kernel(f1, "FETCH_SIZE"), per_kernel(w1, "WRITE_SIZE"), per_kernel(f2, "FETCH_SIZE"), per_kernel(w2, "WRITE_SIZE")
        b1, b2 = hbm_bytes(F1, W1, "rollout_kernel"), hbm_bytes(F2, W2, "rollout_kernel")
        per_step = (b1 - b2) / (T1 - T2)
        alg = E * A * (5.0 + 18.0 / A)
        out["kernels"].append({
            "kernel": label, "csrc_hash": csrc, "instance": F1["rollout_kernel"][2].split("(mapf::")[0].replace("void ", "").strip(),
            "n_envs": E, "n_agents": A, "bytes_per_env_step_launch": round(per_step, 1), "fixed_bytes": round(b1 - per_step * T1, 1),
            "algorithmic_bytes_per_env_step_launch": alg,
            "measured": {str(T1): {"FETCH_SIZE_KB": round(F1["rollout_kernel"][0], 2), "WRITE_SIZE_KB": round(W1["rollout_kernel"][0], 2),
                                   "hbm_bytes_per_launch": int(b1), "dispatches": F1["rollout_kernel"][1]},
                         str(T2): {"FETCH_SIZE_KB": round(F2["rollout_kernel"][0], 2), "WRITE_SIZE_KB": round(W2["rollout_kernel"][0], 2),
                                   "hbm_bytes_per_launch": int(b2), "dispatches": F2["rollout_kernel"][1]}}})
        if "step_kernel" in F1 and "step_kernel" in W1:
            name = F1["step_kernel"][2]
            out["kernels"].append({
                "kernel": step_label or None, "csrc_hash": csrc, "instance": name.split("(mapf::")[0].replace("void ", "").strip(), "single_step_of": label,
                "n_envs": E, "n_agents": A, "bytes_per_env_step_launch": round(hbm_bytes(F1, W1, "step_kernel"), 1), "fixed_bytes": 0.0,
                "algorithmic_bytes_per_env_step_launch": alg,
                "measured": {"1": {"FETCH_SIZE_KB": round(F1["step_kernel"][0], 2), "WRITE_SIZE_KB": round(W1["step_kernel"][0], 2),
                                   "dispatches": F1["step_kernel"][1]}}})
        for k in ("reset_kernel", "fill_actions_kernel"):
            if k in F1 and k in W1:
                out["calibration"]["%s E=%d A=%d" % (k, E, A)] = {
                    "FETCH_SIZE_KB": round(F1[k][0], 2), "WRITE_SIZE_KB": round(W1[k][0], 2), "hbm_bytes_per_launch": int(hbm_bytes(F1, W1, k)),
                    "known_bytes": (E * A * 2 * 2) if k == "reset_kernel" else None}
    fresh = {(k["kernel"], k["n_envs"], k["n_agents"]) for k in out["kernels"]}
    out["kernels"] = [k for k in old.get("kernels", []) if (k.get("kernel"), k.get("n_envs"), k.get("n_agents")) not in fresh] + out["kernels"]
    out["calibration"] = dict(old.get("calibration", {}), **out["calibration"])
    dst = os.path.join(ROOT, "profiles", "traffic.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out["kernels"], indent=1))


if __name__ == "__main__":
    main()
