#!/usr/bin/env python3
"""Per-kernel register/spill metadata from a hipcc -S listing: python kernel_meta.py file.s"""
import re, sys

def kernels(text):
    out = []
    for blk in re.split(r'\n  - \.agpr_count:', text)[1:]:
        name = re.search(r'\.name:\s+(\S+)', blk)
        f = lambda k: int(re.search(r'\.%s:\s+(\d+)' % k, blk).group(1))
        out.append(dict(name=name.group(1), sgpr=f('sgpr_count'), vgpr=f('vgpr_count'), sgpr_spill=f('sgpr_spill_count'),
                        vgpr_spill=f('vgpr_spill_count'), scratch=f('private_segment_fixed_size'), lds=f('group_segment_fixed_size')))
    return out

if __name__ == '__main__':
    for k in kernels(open(sys.argv[1]).read()):
        flag = ' <-- SPILL' if k['sgpr_spill'] or k['vgpr_spill'] or k['scratch'] else ''
        print('%-70s sgpr %3d vgpr %3d sspill %3d vspill %3d scratch %3d%s' % (k['name'][:70], k['sgpr'], k['vgpr'], k['sgpr_spill'], k['vgpr_spill'], k['scratch'], flag))
