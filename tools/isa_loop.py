#!/usr/bin/env python3
"""Instruction mix of the main loop (the one with the most MFMAs) of one kernel in a `hipcc -S --cuda-device-only` listing:
    python tools/isa_loop.py file.s <substring of the mangled kernel name> [top N opcodes]"""
import collections
import sys

lines = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
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
        n_mfma = sum(1 for b in body[hi:back[-1] + 1] if b.strip().startswith('v_mfma'))
        if best is None or (n_mfma, back[-1] - hi) > (best[2], best[1] - best[0]):
            best = (hi, back[-1], n_mfma)
seg = body[best[0]:best[1] + 1]
cnt = collections.Counter()
for l in seg:
    l = l.strip()
    if not l or l[0] in '.;/' or l.endswith(':'):
        continue
    cnt[l.split()[0]] += 1
tot = collections.Counter()
for op, v in cnt.items():
    k = ('MFMA' if op.startswith('v_mfma') else 'VALU' if op.startswith('v_') else 'LDS' if op.startswith('ds_') else
         'VMEM' if op.startswith(('buffer', 'global')) else 'SCRATCH' if op.startswith('scratch') else
         'WAIT/NOP' if op.startswith(('s_waitcnt', 's_nop')) else 'SALU')
    tot[k] += v
print('%s: loop of %d lines: %s' % (body[0].split(':')[0][:60], len(seg), dict(tot)))
for op, v in sorted(cnt.items(), key=lambda t: -t[1])[:top]:
    print('   %5d %s' % (v, op))
