#!/usr/bin/env python3
"""Static instruction mix of the bench's rollout kernel instances (no GPU needed: hipcc -S): which share of a kernel's
32-bit vector instructions is of the FAST kind -- bitwise logic, add / sub, right shifts, moves and v_bitop3 on vector
registers (inline constants and literals allowed), which issue every ~1.2 ns per SIMD -- and which is of the ordinary kind
(left shifts, min / max, compares / selects, multiplies, perm, alignbit, the fused shift-add / or3 / add3 forms, packed,
dot, SDWA, DPP, anything with a scalar-register operand: ~1.9 ns; profiles/r04_valu_issue_cost32.txt and
r04_valu_issue_cost_forms.txt).  64-bit integer and float64 instructions are counted by the
hardware (SQ_INSTS_VALU_INT64 / _MUL_F64 / _ADD_F64) and are left out here.  Writes profiles/valu_mix.json, keyed by the
instance's template arguments as rocprofv3 prints them, with the hash of the kernel sources; tools/derive_valu.py prices a
launch's counted vector instructions with it.

    python tools/valu_mix.py          # ~5 minutes (two translation units)
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, 'gym-mapf_amd', 'csrc')
# measured fast (profiles/r04_valu_issue_cost_forms.txt): bitwise logic, add / sub, right shifts and moves -- in the 32-bit or
# the VOP3 encoding, with vector registers, inline constants or a literal -- and v_bitop3 on vector registers.  Measured
# ordinary (1.5-1.65 x): LEFT shifts, v_min / v_max, compares, selects, multiplies, v_perm, v_alignbit, v_lshl_add / _or,
# v_or3, v_add3, packed 16-bit, dot, SDWA, DPP, and ANY instruction with a scalar-register operand (v_bitop3 included).
FAST = re.compile(r'^v_(xor|and|or|add|sub|subrev|lshrrev|mov|not)_[a-z0-9]+(_e32|_e64)?$|^v_mov_b64_e32$|^v_bitop3_b32$')
WIDE = re.compile(r'^v_(mad_u64_u32|mad_i64_i32|lshl_add_u64|lshlrev_b64|lshrrev_b64|add_f64|mul_f64|fma_f64|cvt_f64|cmp_\w+_f64|cmp_\w+_u64)')


def classify(line):
    """'fast' | 'ordinary' | 'wide' | None for one line of a -S listing"""
    text = line.strip()
    if not text.startswith('v_'):
        return None
    op = text.split()[0]
    if WIDE.match(op):
        return 'wide'
    operands = text[len(op):]
    if FAST.match(op) and not re.search(r'\bs\d+\b|\bs\[|\bvcc\b|\bexec\b|\bsdwa\b|dst_sel|quad_perm|row_', operands):
        return 'fast'
    return 'ordinary'


_LISTINGS = {}


def mix_of(unit_flags, instance_re):
    key = tuple(unit_flags)
    if key not in _LISTINGS:                                     # one hipcc -S per translation unit, not per instance
        with tempfile.TemporaryDirectory() as tmp:
            out = os.path.join(tmp, 'k.s')
            subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off',
                                   '-I' + os.path.join(ROOT, 'include'), '-S', '--cuda-device-only'] + unit_flags +
                                  [os.path.join(CSRC, 'mapf_lq_rollout.hip'), '-o', out], stderr=subprocess.DEVNULL)
            _LISTINGS[key] = open(out).read()
    text = _LISTINGS[key]
    m = re.search(r'^(_ZN4mapf\S*lq_rollout_kernel' + instance_re + r'\S*):', text, re.M)
    body = text[m.end():text.index('.Lfunc_end', m.end())].split('\n')
    counts = {'fast': 0, 'ordinary': 0, 'wide': 0}
    for line in body:
        c = classify(line)
        if c:
            counts[c] += 1
    return counts


def main():
    import bench
    # (rocprofv3's spelling of the instance, -D flags of its translation unit, mangled template arguments)
    instances = [
        ('lq_rollout_kernel<2, 4, true, true, false, false, false, 0>', ['-DMAPF_LQ_K=4', '-DMAPF_LQ_RECORD=1'], 'ILi2ELi4ELb1ELb1ELb0ELb0ELb0ELi0E'),
        ('lq_rollout_kernel<2, 4, true, false, false, false, false, 0>', ['-DMAPF_LQ_K=4', '-DMAPF_LQ_RECORD=1'], 'ILi2ELi4ELb1ELb0ELb0ELb0ELb0ELi0E'),   # ... with the in-kernel policy
        ('lq_rollout_kernel<8, 4, true, true, false, true, false, 2>', ['-DMAPF_LQ_K=4', '-DMAPF_LQ_RECORD=1'], 'ILi8ELi4ELb1ELb1ELb0ELb1ELb0ELi2E'),
        ('lq_rollout_kernel<8, 4, true, true, false, true, false, 1>', ['-DMAPF_LQ_K=4', '-DMAPF_LQ_RECORD=1'], 'ILi8ELi4ELb1ELb1ELb0ELb1ELb0ELi1E'),
        ('lq_rollout_kernel<8, 4, true, true, false, true, false, 3>', ['-DMAPF_LQ_K=4', '-DMAPF_LQ_RECORD=1'], 'ILi8ELi4ELb1ELb1ELb0ELb1ELb0ELi3E'),
        ('lq_rollout_kernel<4, 8, true, true, false, true, false, 0>', ['-DMAPF_LQ_K=8', '-DMAPF_LQ_RECORD=1'], 'ILi4ELi8ELb1ELb1ELb0ELb1ELb0ELi0E'),
        ('lq_rollout_kernel<1, 8, true, true, false, false, false, 0>', ['-DMAPF_LQ_K=8', '-DMAPF_LQ_RECORD=1'], 'ILi1ELi8ELb1ELb1ELb0ELb0ELb0ELi0E'),
    ]
    doc = {'_how': ' '.join(__doc__.split('\n\n')[0].split()), 'csrc_hash': bench.csrc_hash(), 'instances': {}}
    for name, flags, mangled in instances:
        c = mix_of(flags, mangled)
        n32 = c['fast'] + c['ordinary']
        doc['instances'][name] = dict(c, fast_share_of_32bit=c['fast'] / n32)
        print('%-70s fast %5d ordinary %5d wide %5d -> fast share of the 32-bit ones %.3f' % (name, c['fast'], c['ordinary'], c['wide'], c['fast'] / n32))
    with open(os.path.join(ROOT, 'profiles', 'valu_mix.json'), 'w') as f:
        json.dump(doc, f, indent=1)


if __name__ == '__main__':
    main()
