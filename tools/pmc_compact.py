"""Per-kernel summary of a rocprofv3 counter_collection.csv (one row per kernel and counter: dispatches, mean, min,
max of the per-dispatch value) -- the raw file has one row per dispatch."""
import collections
import csv
import sys

rows = collections.defaultdict(list)
meta = {}
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        key = (r["Kernel_Name"], r["Counter_Name"])
        rows[key].append(float(r["Counter_Value"]))
        meta[key] = (r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["SGPR_Count"])
w = csv.writer(sys.stdout)
w.writerow(["Kernel_Name", "Counter_Name", "Dispatches", "Mean", "Min", "Max", "Grid_Size", "Workgroup_Size",
            "LDS_Block_Size", "VGPR_Count", "SGPR_Count"])
for (k, c), v in rows.items():
    w.writerow([k, c, len(v), "%.2f" % (sum(v) / len(v)), "%.2f" % min(v), "%.2f" % max(v), *meta[(k, c)]])
