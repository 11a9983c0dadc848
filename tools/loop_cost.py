#!/usr/bin/env python3
"""Weighted VALU cost of the main path of a rollout kernel's step loop (weights: profiles/r01_instruction_costs.txt).
usage: python loop_cost.py /tmp/ro.s [mangled-name-substring]   (default: the L=4 DENSE recording kernel)"""
import re, sys, collections
s = open(sys.argv[1]).read()
key = sys.argv[2] if len(sys.argv) > 2 else 'lg_rollout_kernelILi4ELb1ELb1ELb1ELb1ELb1EE'
m = re.search(r'^(_ZN4mapf\w*%s\w*):' % re.escape(key), s, re.M)
body = s[m.end():s.index('.Lfunc_end', m.end())].split('\n')
hdr = [i for i, l in enumerate(body) if 'Inner Loop Header' in l and 'Depth=1' in l][-1]
# loop = from the back-edge target's predecessor block label to the loop exit branch; take everything marked "in Loop"
start = hdr
while start > 0 and not body[start - 1].strip().startswith('s_branch'):
    start -= 1
end = max(i for i, l in enumerate(body) if 'in Loop: Header' in l)
while end + 1 < len(body) and not body[end + 1].startswith('.LBB'):
    end += 1
def weight(op, line):
    if not op.startswith('v_'):
        return 0.0
    if 'mad_u64' in op: return 3.0
    if '_f64' in op or 'lshl_add_u64' in op: return 2.0
    if op.endswith('_e32') and 'dpp' not in line and 'sdwa' not in line: return 1.0
    return 1.6
blocks = collections.OrderedDict(); cur = 'entry'
philox = False
for l in body[start:end + 1]:
    t = l.strip()
    if t.startswith('.LBB'):
        cur = t.split(':')[0]
    if not l.startswith('\t') or t.startswith((';', '.')):
        continue
    op = t.split()[0]
    b = blocks.setdefault(cur, dict(valu=0, w=0.0, salu=0, lds=0, vmem=0, mad=0))
    if op.startswith('v_'):
        b['valu'] += 1; b['w'] += weight(op, t); b['mad'] += 'mad_u64' in op
    elif op.startswith('s_'): b['salu'] += 1
    elif op.startswith('ds_'): b['lds'] += 1
    elif op.startswith(('global_', 'buffer_', 'flat_')): b['vmem'] += 1
tot = dict(valu=0, w=0.0, salu=0, lds=0, vmem=0)
for k, b in blocks.items():
    rare = b['mad'] >= 8                      # Philox blocks: refresh (1 step in 4) or tie redo (rare)
    print('%-12s valu %3d  weighted %6.1f  salu %3d  lds %2d  vmem %2d %s' % (k, b['valu'], b['w'], b['salu'], b['lds'], b['vmem'], '(philox)' if rare else ''))
    if not rare:
        for x in tot: tot[x] += b[x]
print('main path (philox blocks excluded): valu %d, weighted %.1f, salu %d, lds %d, vmem %d' % (tot['valu'], tot['w'], tot['salu'], tot['lds'], tot['vmem']))
