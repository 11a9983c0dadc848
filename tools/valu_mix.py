#!/usr/bin/env python3
"""Static instruction mix of the bench's rollout kernel instances (no GPU needed: hipcc -S): which share of a kernel's
32-bit vector instructions is of the FAST kind -- VOP1 / VOP2 in the 32-bit encoding with vector-register operands only
(v_xor, v_and, v_or, v_add, shifts, v_mov) and v_bitop3, which issue every ~1.2 ns per SIMD -- and which is of the ordinary
kind (VOP3 encodings, SDWA, DPP, packed, dot, perm, multiplies, compares / selects, anything with a scalar or literal
operand: ~1.9 ns; profiles/r04_valu_issue_cost32.txt).  64-bit integer and float64 instructions are counted by the
hardware (SQ_INSTS_VALU_INT64 / _MUL_F64 / _ADD_F64) and are left out here.  Writes profiles/valu_mix.json, keyed by the
instance's template arguments as rocprofv3 prints them, with the hash of the kernel sources; tools/derive_valu.py prices a
launch's counted vector instructions with it.

    python tools/valu_mix.py          # ~1 minute (three translation units)
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
FAST = re.compile(r'^v_(xor|and|or|add|sub|subrev|lshlrev|lshrrev|ashrrev|mov|not|min|max)_[a-z0-9]+_e32$|^v_mov_b64_e32$|^v_bitop3_b32$')
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
    if FAST.match(op):
        # a scalar register, a literal or an inline constant other than a small integer keeps the 32-bit encoding but
        # measured like the ordinary kind (v_xor_b32 with an SGPR: 1.57 x)
        if re.search(r'\bs\d+\b|\bs\[|\bvcc\b|\bexec\b|0x[0-9a-f]+', operands):
            return 'ordinary'
        return 'fast'
    return 'ordinary'


def mix_of(unit_flags, instance_re):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, 'k.s')
        subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off',
                               '-I' + os.path.join(ROOT, 'include'), '-S', '--cuda-device-only'] + unit_flags +
                              [os.path.join(CSRC, 'mapf_lq_rollout.hip'), '-o', out], stderr=subprocess.DEVNULL)
        text = open(out).read()
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
        ('lq_rollout_kernel<2, 4, true, true, false, false, false, false>', ['-DMAPF_LQ_K=4', '-DMAPF_LQ_RECORD=1'], 'ILi2ELi4ELb1ELb1ELb0ELb0ELb0ELb0E'),
        ('lq_rollout_kernel<8, 4, true, true, false, true, false, true>', ['-DMAPF_LQ_K=4', '-DMAPF_LQ_RECORD=1'], 'ILi8ELi4ELb1ELb1ELb0ELb1ELb0ELb1E'),
        ('lq_rollout_kernel<4, 8, true, true, false, true, false, false>', ['-DMAPF_LQ_K=8', '-DMAPF_LQ_RECORD=1'], 'ILi4ELi8ELb1ELb1ELb0ELb1ELb0ELb0E'),
        ('lq_rollout_kernel<1, 8, true, true, false, false, false, false>', ['-DMAPF_LQ_K=8', '-DMAPF_LQ_RECORD=1'], 'ILi1ELi8ELb1ELb1ELb0ELb0ELb0ELb0E'),
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
