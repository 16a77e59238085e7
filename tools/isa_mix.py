#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -S listing: python tools/isa_mix.py file.s <substring of the mangled name>"""
import collections
import sys

lines = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
start = None
for i, l in enumerate(lines):
    if l.startswith('_ZN') and pat in l.split(':')[0] and l.rstrip().split(';')[0].strip().endswith(':'):
        start = i
        break
cnt = collections.Counter()
for l in lines[start + 1:]:
    l = l.strip()
    if l.startswith('.end_amdhsa_kernel') or l.startswith('s_endpgm'):
        break
    if not l or l[0] in '.;/' or l.endswith(':'):
        continue
    op = l.split()[0]
    if op.startswith('v_mfma'):
        cnt['MFMA'] += 1
    elif op.startswith('scratch_'):
        cnt['SCRATCH ' + op] += 1
    elif op.startswith('v_'):
        cnt['VALU'] += 1
        cnt['  ' + op] += 1
    elif op.startswith('ds_'):
        cnt['LDS'] += 1
        cnt['  ' + op] += 1
    elif op.startswith('buffer_') or op.startswith('global_'):
        cnt['VMEM'] += 1
        cnt['  ' + op] += 1
    elif op.startswith('s_waitcnt'):
        cnt['s_waitcnt'] += 1
    elif op.startswith('s_'):
        cnt['SALU'] += 1
for k, v in sorted(cnt.items(), key=lambda t: (t[0].startswith('  '), -t[1])):
    if v >= int(sys.argv[3]) if len(sys.argv) > 3 else 1:
        print('%6d %s' % (v, k))
