#!/usr/bin/env python3
"""One character per instruction of a kernel's main loop (the one with the most MFMAs) in a `hipcc -S --cuda-device-only` listing:
M = MFMA, r / w = LDS read / write, L / S = buffer load / store, . = other VALU, |n = s_waitcnt lgkmcnt(n) (|v: vmcnt only).
Shows at a glance whether operand reads are issued ahead of the MFMAs that hide them.   python tools/isa_seq.py file.s <kernel substring>"""
import sys

lines = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
start = [i for i, l in enumerate(lines) if l.startswith('_Z') and pat in l and l.split(';')[0].rstrip().endswith(':')][0]
end = [i for i in range(start, len(lines)) if 's_endpgm' in lines[i]][0]
body = lines[start:end + 1]
import re
labs = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        labs[m.group(1)] = i
best = None
for i, l in enumerate(body):                      # back edges: a branch to a label above it
    t = l.strip().split()
    if t and (t[0].startswith('s_cbranch') or t[0] == 's_branch') and t[-1] in labs and labs[t[-1]] < i:
        n = sum(1 for b in body[labs[t[-1]]:i + 1] if b.strip().startswith('v_mfma'))
        if best is None or (n, i - labs[t[-1]]) > (best[2], best[1] - best[0]):
            best = (labs[t[-1]], i, n)
out = []
for l in body[best[0]:best[1] + 1]:
    t = l.strip().split()
    if not t:
        continue
    op = t[0]
    if op.startswith('v_mfma'):
        out.append('M')
    elif op.startswith('ds_read'):
        out.append('r')
    elif op.startswith('ds_write'):
        out.append('w')
    elif op.startswith('s_waitcnt'):
        out.append('|' + (l.split('lgkmcnt(')[1].split(')')[0] if 'lgkmcnt' in l else 'v'))
    elif op.startswith('buffer_load'):
        out.append('L')
    elif op.startswith('buffer_store'):
        out.append('S')
    elif op.startswith('scratch'):
        out.append('X')
    elif op.startswith('v_'):
        out.append('.')
print(''.join(out))
