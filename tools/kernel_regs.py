#!/usr/bin/env python3
"""Register / LDS / spill table of the kernels in a `hipcc -S --cuda-device-only` listing: python tools/kernel_regs.py file.s [substring]"""
import re
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ''
for m in re.finditer(r'- \.agpr_count:\s+(\d+).*?\.group_segment_fixed_size: (\d+).*?\.name:\s+(\S+).*?\.sgpr_spill_count: (\d+).*?'
                     r'\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count: (\d+)', txt, re.S):
    ag, lds, name, ss, vg, vs = m.groups()
    t = re.search(r'N_1\d+(\w+?)I(.*?)EEv', name)
    label = (t.group(1) + '<' + t.group(2).replace('Li', '').replace('ELb', ',b').replace('E', ',') + '>') if t else name
    if pat in label:
        print('%-44s vgpr %3s agpr %3s static-lds %6s sgpr-spill %3s vgpr-spill %3s' % (label, vg, ag, lds, ss, vs))
