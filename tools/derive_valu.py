#!/usr/bin/env python3
"""profiles/valu.json from the rocprofv3 SQ counter passes of the bench command (tools/refresh_profiles.sh: pass sq1 --
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU ... -- and pass sq3 -- SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_MUL_F64
SQ_INSTS_VALU_ADD_F64 ... -- both with --kernel-trace): what bench.py's `roofline.valu_frac` is computed from.

usage: python tools/derive_valu.py <label> <E> <A> <T> <sq1 counter_collection.csv> <sq1 kernel_trace.csv> <sq3 counter_collection.csv> [...]
       (groups of seven; label = the rollout kernel as the library reports it for that batch: bench line `roofline.kernel`)

Per kernel instance and batch: the launch's vector wave-instructions by kind -- 64-bit integer, float64 multiply, float64
add as the hardware counts them, and the 32-bit rest split by the kernel's STATIC mix (profiles/valu_mix.json,
tools/valu_mix.py) into the fast kind (32-bit-encoded VOP1 / VOP2 on vector registers only, v_bitop3: ~1.2 ns per
wave-instruction per SIMD) and the ordinary kind (VOP3, SDWA, DPP, packed, dot, perm, multiplies, compares / selects, any
scalar or literal operand: ~1.9 ns) -- and `valu_ms_per_simd` = the time ONE SIMD's vector ALU needs for its 1/1024 share
of them at the measured issue rates (profiles/r04_valu_issue_cost.txt and r04_valu_issue_cost32.txt,
tools/microbench/valu_issue_cost64.hip and valu_issue_cost.hip: eight waves per SIMD, independent registers).  bench.py
divides it by the live launch time.  Also kept, from the counters alone: the launch's length in shader cycles
(SQ_BUSY_CYCLES is summed over the 32 shader engines), the waves resident per SIMD on average (wave cycles / that length --
NOT waves launched / 1024: a 131072-env launch of a kernel that fits two waves per SIMD keeps two resident, not eight) and
the share of its own cycles a wave spends with a vector instruction in flight.  A share above 1 is a broken derivation and is refused."""
import collections
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SIMDS, SHADER_ENGINES = 256 * 4, 32


def issue_costs_ns():
    """{kind: ns per wave-instruction per SIMD} from the two committed microbenchmark outputs: every kind as its ratio to
    v_xor_b32 IN THE SAME RUN (the boxes' clocks differ by a few percent) times the mean v_xor_b32 time of the two runs;
    'ordinary' = the median ratio of the VOP3 / SDWA / DPP / multiply / select / scalar-operand forms measured."""
    def parse(name):
        out = {}
        with open(os.path.join(ROOT, 'profiles', name)) as f:
            for line in f:
                m = re.match(r'(.+?)\s+[0-9.]+ ms\s+([0-9.]+) ns/wave-instr/SIMD', line)
                if m:
                    out[m.group(1).strip()] = float(m.group(2))          # (a repeated name: the last, warmed line wins)
        return out
    wide, narrow = parse('r04_valu_issue_cost.txt'), parse('r04_valu_issue_cost32.txt')
    base = 0.5 * (wide['v_xor_b32'] + narrow['v_xor_b32'])
    ordinary = sorted(narrow[k] / narrow['v_xor_b32'] for k in (
        'v_min3_u32', 'v_perm_b32', 'v_bfe_u32', 'v_and_or_b32', 'v_lshl_or_b32', 'v_cmp+v_cndmask', 'v_mov_b32_dpp', 'v_xor_b32_dpp',
        'v_xor_b32_sdwa', 'v_mul_lo_u32', 'v_mul_u32_u24', 'v_mad_u32_u24', 'v_add3_u32', 'v_xor_b32 (sgpr)'))
    return {'fast': base, 'ordinary': base * ordinary[len(ordinary) // 2],
            'int64': base * wide['v_mad_u64_u32'] / wide['v_xor_b32'], 'mul_f64': base * wide['v_mul_f64'] / wide['v_xor_b32'],
            'add_f64': base * wide['v_add_f64'] / wide['v_xor_b32']}


def fast_share(instance):
    """share of the fast kind among `instance`'s 32-bit vector instructions (static mix of THESE kernel sources)"""
    import bench
    with open(os.path.join(ROOT, 'profiles', 'valu_mix.json')) as f:
        doc = json.load(f)
    if doc.get('csrc_hash') != bench.csrc_hash():
        raise SystemExit('derive_valu: profiles/valu_mix.json was made from other kernel sources -- run tools/valu_mix.py first')
    for name, mix in doc['instances'].items():
        if name in instance:
            return mix['fast_share_of_32bit']
    raise SystemExit('derive_valu: no static mix for %r in profiles/valu_mix.json (add the instance to tools/valu_mix.py)' % instance)


def rollout_means(path):
    vals = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if "rollout_kernel" in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for c, v in vals.items():
        full = [x for x in v if x >= 0.5 * max(v)] if max(v) > 0 else v   # (the bench also launches one short rollout: the parity leg)
        out[c] = sum(full) / len(full)
    return out


def rollout_instance(path):
    """the rollout kernel's name as rocprofv3 prints it (of the full-length dispatches: the most frequent one)"""
    names = collections.Counter()
    with open(path) as f:
        for r in csv.DictReader(f):
            if "rollout_kernel" in r["Kernel_Name"]:
                names[r["Kernel_Name"]] += 1
    return names.most_common(1)[0][0]


def rollout_duration_ms(path):
    d = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if "rollout_kernel" in r["Kernel_Name"]:
                d.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    full = [x for x in d if x >= 0.5 * max(d)]
    return sum(full) / len(full), len(full)


def price(counts, share_fast, cost):
    """vector-ALU time in ns of a launch: counts = (32-bit rest, 64-bit integer, float64 multiply, float64 add)"""
    n_plain, n_int64, n_mul, n_add = counts
    return n_plain * (share_fast * cost['fast'] + (1.0 - share_fast) * cost['ordinary']) + \
        n_int64 * cost['int64'] + n_mul * cost['mul_f64'] + n_add * cost['add_f64']


def main():
    import bench
    args = sys.argv[1:]
    dst = os.path.join(ROOT, "profiles", "valu.json")
    try:
        with open(dst) as f:
            doc = json.load(f)
    except (OSError, ValueError):
        doc = {"kernels": []}
    doc["_how"] = " ".join(" ".join(__doc__.strip().split("\n\n")[2:]).split())
    cost = issue_costs_ns()
    doc["issue_cost_ns_per_wave_instruction_per_simd"] = cost
    while len(args) >= 7:
        label, E, A, T, counters, trace, counters3 = args[:7]
        args = args[7:]
        E, A, T = int(E), int(A), int(T)
        m = rollout_means(counters)
        m3 = rollout_means(counters3)
        ms, n = rollout_duration_ms(trace)
        n_valu = m["SQ_INSTS_VALU"]
        n_int64, n_mul, n_add = m3["SQ_INSTS_VALU_INT64"], m3["SQ_INSTS_VALU_MUL_F64"], m3["SQ_INSTS_VALU_ADD_F64"]
        n_plain = n_valu - n_int64 - n_mul - n_add
        assert n_plain > 0, (n_valu, n_int64, n_mul, n_add)
        instance = rollout_instance(counters)
        share_fast = fast_share(instance)
        valu_ns = price((n_plain, n_int64, n_mul, n_add), share_fast, cost)
        valu_ms_per_simd = valu_ns / SIMDS / 1e6
        launch_cycles = m["SQ_BUSY_CYCLES"] / SHADER_ENGINES
        share = valu_ms_per_simd / ms
        if not 0.0 < share <= 1.0:
            raise SystemExit('derive_valu: %s at %d envs: vector-ALU time %.4f ms per SIMD against a launch of %.4f ms -- a share of %.3f cannot be' %
                             (label, E, valu_ms_per_simd, ms, share))
        entry = {"kernel": label, "n_envs": E, "n_agents": A, "env_steps_per_launch": T, "csrc_hash": bench.csrc_hash(),
                 "valu_insts_per_launch": n_valu, "valu_insts_per_wave_step": n_valu / m["SQ_WAVES"] / T,
                 "valu_insts_by_kind": {"int64": n_int64, "mul_f64": n_mul, "add_f64": n_add, "other_32bit": n_plain,
                                        "fast_share_of_other_32bit": share_fast},
                 "instance": instance.split("(mapf::")[0].replace("void ", "").strip(),
                 "valu_ms_per_simd": valu_ms_per_simd, "valu_share_of_launch_in_pass": share,
                 "waves": m["SQ_WAVES"], "launch_shader_cycles": launch_cycles,
                 "shader_clock_ghz_in_pass": launch_cycles / (ms * 1e6),
                 "waves_resident_per_simd": m["SQ_WAVE_CYCLES"] * 4.0 / (SIMDS * launch_cycles),
                 "valu_active_share_of_wave_cycles": m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"],
                 "launch_ms_in_pass": ms, "launches_in_pass": n}
        doc["kernels"] = [k for k in doc["kernels"] if not (k["kernel"] == label and k["n_envs"] == E and k["n_agents"] == A)] + [entry]
        print(json.dumps(entry, indent=1))
    with open(dst, "w") as f:
        json.dump(doc, f, indent=1)


if __name__ == "__main__":
    main()
