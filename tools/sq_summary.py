"""Per-kernel summary of rocprofv3 SQ / GRBM counter passes (counter_collection.csv files): mean per full-length
dispatch of every counter, plus the ratios the VALU-bound argument in DESIGN.md rests on.
usage: python tools/sq_summary.py <pass1.csv> [<pass2.csv> ...]

Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves;
SQ_INSTS_* count wave-instructions; SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE are summed over the XCDs (8) -- and SQ_BUSY over
shader engines too -- so only ratios between SQ counters of the same kind are quoted."""
import collections
import csv
import sys

vals = collections.defaultdict(lambda: collections.defaultdict(list))
paths = sys.argv[1:]
match = ("rollout_kernel", "step_kernel")
if paths and paths[0] == "--match":          # --match <substring>: summarise the kernels whose name contains it
    match, paths = (paths[1],), paths[2:]
for path in paths:
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            if any(m in name for m in match):
                key = name.split("(mapf::")[0].replace("void ", "").strip() + "  grid=" + r["Grid_Size"]
                vals[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kern, counters in vals.items():
    print(kern)
    mean = {}
    for c, v in sorted(counters.items()):
        full = [x for x in v if x >= 0.5 * max(v)] if "rollout" in kern else v
        mean[c] = sum(full) / len(full)
        print("  %-24s %16.1f   (mean of %d dispatches)" % (c, mean[c], len(full)))

    def ratio(a, b, label):
        if a in mean and b in mean and mean[b]:
            print("  %-58s %.3f" % (label, mean[a] / mean[b]))
    ratio("SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "VALU active / wave cycles")
    ratio("SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES", "any instruction active / wave cycles")
    ratio("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", "issue stalls (WAIT_INST_ANY) / wave cycles")
    ratio("SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "parked in s_waitcnt / barrier (WAIT_ANY) / wave cycles")
    ratio("SQ_ACTIVE_INST_LDS", "SQ_WAVE_CYCLES", "LDS active / wave cycles")
    ratio("SQ_ACTIVE_INST_SCA", "SQ_WAVE_CYCLES", "scalar active / wave cycles")
    ratio("SQ_ACTIVE_INST_VMEM", "SQ_WAVE_CYCLES", "vector memory active / wave cycles")
    ratio("SQ_INSTS_VALU", "SQ_WAVES", "VALU wave-instructions per wave")
    ratio("SQ_INSTS_SALU", "SQ_WAVES", "SALU wave-instructions per wave")
    ratio("SQ_INSTS_LDS", "SQ_WAVES", "LDS wave-instructions per wave")
    ratio("SQ_INSTS_VMEM_WR", "SQ_WAVES", "vector stores per wave")
    ratio("SQ_INSTS_VMEM_RD", "SQ_WAVES", "vector loads per wave")
    ratio("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "LDS bank-conflict cycles / LDS active cycles")
    ratio("SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "quad-cycles per VALU wave-instruction (x4 = cycles)")
    ratio("SQ_ACTIVE_INST_MISC", "SQ_WAVE_CYCLES", "misc (branch, waitcnt, nop, sendmsg) active / wave cycles")
    ratio("SQ_INST_CYCLES_VMEM_WR", "SQ_WAVE_CYCLES", "vector-store issue cycles / wave cycles")
    ratio("SQ_INST_CYCLES_VMEM_RD", "SQ_WAVE_CYCLES", "vector-load issue cycles / wave cycles")
    ratio("SQ_INST_CYCLES_SALU", "SQ_WAVE_CYCLES", "SALU issue cycles / wave cycles")
    ratio("SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES", "LDS issue stalls / wave cycles")
    ratio("SQ_INSTS_BRANCH", "SQ_WAVES", "branch instructions per wave")
