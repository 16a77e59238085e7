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
best = None
for hi, l in enumerate(body):
    if 'Loop Header: Depth=' not in l:
        continue
    lab = l.split(':')[0]
    back = [i for i, b in enumerate(body) if ('s_cbranch' in b or 's_branch' in b) and b.split()[-1] == lab and i > hi]
    if back:
        n = sum(1 for b in body[hi:back[-1] + 1] if b.strip().startswith('v_mfma'))
        if best is None or n > best[2]:
            best = (hi, back[-1], n)
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
