#!/usr/bin/env python3
"""Per-case table of the gradient fixture: the reference's own fp32 errors (8 threads, 1 thread), its flipped decisions, and
the error of every engine variant next to them, with the verdict of the OLD single-case gate (flat error <= 2 x the
reference's 8-thread error + 1e-6) for each evaluation -- including the reference's own 1-thread run.  Shows that the old
gate is a statement about which seeds were committed, for the fp32-MFMA engine as much as for x3.
usage (GPU box): python tests/diag/gpu_gradgate_table.py > gpurun_out/gradgate_table.txt"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import gradgate as GG                               # noqa: E402
from test_gpu_grad_gate import run_cases            # noqa: E402

groups = GG.load_groups()
modes = sys.argv[1:] or ['f32', 'x3']
res = {m: run_cases(m, groups) for m in modes}
fails = {m: 0 for m in modes}
fails['ref-1t'] = 0
total = 0
print('# class: D degenerate, S safe (margins > %.1f E), N near-tie.  old gate = err <= 2 err8 + 1e-6; X = fails it' % GG.TAU)
print('%-3s %4s %1s %9s %9s %5s %5s ' % ('grp', 'n', 'c', 'ref err8', 'ref err1', 'flip8', 'flip1') + ' '.join('%9s' % m for m in modes) + '   old gate: ref-1t ' + ' '.join(modes))
for tag, g in groups.items():
    deg, safe, near = GG.classes(g)
    for i in range(len(g['n'])):
        cls = 'D' if deg[i] else ('S' if safe[i] else 'N')
        f8 = int(g['relu'][i, :, 7].sum() + g['pool'][i, :, 7].sum())
        f1 = int(g['relu'][i, :, 8].sum() + g['pool'][i, :, 8].sum())
        row = '%-3s %4d %1s %9.2e %9.2e %5d %5d ' % (tag, g['n'][i], cls, g['err8'][i] if not deg[i] else 0, g['err1'][i] if not deg[i] else 0, f8, f1)
        row += ' '.join('%9.2e' % res[m][0][tag][i] for m in modes)
        if not deg[i]:
            total += 1
            marks = []
            ok = GG.old_single_case_gate(g['err1'][i], g['err8'][i])
            fails['ref-1t'] += not ok
            marks.append('.' if ok else 'X')
            for m in modes:
                ok = GG.old_single_case_gate(res[m][0][tag][i], g['err8'][i])
                fails[m] += not ok
                marks.append('.' if ok else 'X')
            row += '        ' + '      '.join(marks)
        print(row)
print('# non-degenerate cases: %d; failing the OLD single-case gate: %s' % (total, ', '.join('%s %d' % kv for kv in fails.items())))
for m in modes:
    try:
        print('# new gate, %s: %s' % (m, GG.check(groups, res[m][0], res[m][1], label=m)))
    except AssertionError as exc:
        print('# new gate, %s: FAILED %s' % (m, exc))
