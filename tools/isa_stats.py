#!/usr/bin/env python3
"""Instruction histogram of one kernel group: python isa_stats.py <group> [substring ...]"""
import subprocess, sys, collections, os
g = sys.argv[1]
here = os.path.dirname(os.path.abspath(__file__))
out = '/tmp/mapf_g%s.s' % g
subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off',
                       '-I' + os.path.join(here, '..', '..', 'include'), '-DMAPF_GROUP=' + g, '-S', '--cuda-device-only',
                       os.path.join(here, 'mapf_kernels.hip'), '-o', out], stderr=subprocess.DEVNULL)
s = open(out).read()
import re
for m in re.finditer(r'^(_ZN4mapf\w+):', s, re.M):
    name = m.group(1)
    if len(sys.argv) > 2 and not any(x in name for x in sys.argv[2:]):
        continue
    body = s[m.end():s.index('.Lfunc_end', m.end())].split('\n')
    ins = [l.split()[0] for l in body if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    c = collections.Counter(ins)
    valu = sum(v for k, v in c.items() if k.startswith('v_'))
    salu = sum(v for k, v in c.items() if k.startswith('s_'))
    vg = re.search(r'\.vgpr_count:\s+(\d+)', s[s.index('.name:           ' + name):] if ('.name:           ' + name) in s else '')
    print('%s total=%d valu=%d salu=%d mem=%d vgpr=%s' % (name, len(ins), valu, salu, len(ins) - valu - salu, vg.group(1) if vg else '?'))
    print('   ', sorted(c.items(), key=lambda x: -x[1])[:28])
