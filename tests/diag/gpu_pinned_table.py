#!/usr/bin/env python3
"""Per-tensor table of the decision-pinned gradient comparison (tests/test_gpu_grad_pinned.py): for the three benchmarked
shapes and every engine mode, the max-norm relative error of each gradient tensor against the fp64 evaluation of the branch the
engine took, next to the reference's own fp32 figures on ITS branch (tests/golden/pinned_reference_errors.npz).
usage (GPU box): python tests/diag/gpu_pinned_table.py [json out] > gpurun_out/pinned_table.txt"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
from test_gpu_grad_pinned import run_pinned            # noqa: E402

ref = np.load(os.path.join(ROOT, 'tests', 'golden', 'pinned_reference_errors.npz'))
names = [str(n) for n in ref['names']]
out = {}
for case in ('cfg2', 'cfg4', 'cfg5'):
    modes = ['f32', 'f32s'] if case == 'cfg5' else ['f32', 'x3', 'f32s', 'x3s']
    res = {m: run_pinned(case, m) for m in modes}
    out[case] = {m: {'errs': res[m][0], 'loss_err': res[m][1], 'decisions': res[m][2]} for m in modes}
    e8, e1 = ref[case + '/err8'], ref[case + '/err1']
    print('# %s: %d decisions pinned.  columns: reference fp32 (8 threads, 1 thread) on its own branch | engine modes on theirs' % (case, res[modes[0]][2]))
    print('%-36s %9s %9s  ' % ('tensor', 'ref 8t', 'ref 1t') + ' '.join('%9s' % m for m in modes))
    for i, n in enumerate(names):
        if n.endswith('convs.2.bias'):
            continue
        print('%-36s %9.2e %9.2e  ' % (n, e8[i], e1[i]) + ' '.join('%9.2e' % res[m][0][n] for m in modes))
    live = [i for i, n in enumerate(names) if not n.endswith('convs.2.bias')]
    print('%-36s %9.2e %9.2e  ' % ('# worst', e8[live].max(), e1[live].max()) + ' '.join('%9.2e' % max(res[m][0].values()) for m in modes))
    print('%-36s %9.2e %9.2e  ' % ('# median', np.median(e8[live]), np.median(e1[live])) + ' '.join('%9.2e' % float(np.median(list(res[m][0].values()))) for m in modes))
    worst_ratio = {m: max(res[m][0][n] / max(1e-5, 2 * max(e8[i], e1[i])) for i, n in enumerate(names) if not n.endswith('convs.2.bias')) for m in modes}
    print('# worst of err / max(1e-5, 2 x the reference\'s own error on that tensor): ' + ', '.join('%s %.2f' % (m, worst_ratio[m]) for m in modes))
    print()
if len(sys.argv) > 1:
    with open(sys.argv[1], 'w') as f:
        json.dump(out, f)
