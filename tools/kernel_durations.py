"""Per-dispatch durations of the bench's kernels from a rocprofv3 kernel_trace.csv: the 256-step rollout launches
apart from the short parity launch (the aggregated kernel_stats.csv averages them together).
usage: python tools/kernel_durations.py <kernel_trace.csv>"""
import collections
import csv
import sys

d = collections.defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        d[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for name, v in d.items():
    if "rollout_kernel" in name:
        full = [x for x in v if x >= 0.5 * max(v)]
        short = [x for x in v if x < 0.5 * max(v)]
        print("%s\n  %d full launches: mean %.1f us, min %.1f, max %.1f; %d short launch(es) (parity leg): %s us"
              % (name.split("(mapf::")[0], len(full), sum(full) / len(full), min(full), max(full), len(short),
                 ", ".join("%.1f" % x for x in short)))
    elif "step_kernel" in name:
        print("%s\n  %d launches: mean %.2f us, min %.2f, max %.2f" % (name.split("(mapf::")[0], len(v), sum(v) / len(v), min(v), max(v)))
