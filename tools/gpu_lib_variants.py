#!/usr/bin/env python3
"""Time bench.py (cfg2, --mfma x3 unless told otherwise) with alternative builds of the library, one fresh process per build:
python tools/gpu_lib_variants.py name[:bench args] ...   (a name = graph_neural_net_amd/_dbg/libfgnn_hip_<name>.so; 'main' = the shipped library)
Prints ms/step and the event-timed launch averages of the MLP kernels."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for spec in sys.argv[1:]:
    name, _, extra = spec.partition(':')
    args = ['--no-cpu-baseline', '--no-extra-configs'] + (extra.split() if extra else ['--mfma', 'x3'])
    if name == 'main':
        cmd = [sys.executable, os.path.join(ROOT, 'bench.py')] + args
    else:
        cmd = [sys.executable, os.path.join(ROOT, 'tools', 'gpu_bench_altlib.py'), 'graph_neural_net_amd/_dbg/libfgnn_hip_%s.so' % name] + args
    r = subprocess.run(cmd, capture_output=True, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    if not lines:
        print(spec, 'FAILED', r.stderr[-400:])
        continue
    d = json.loads(lines[0])
    ks = {k: v['avg_ms'] * 1e3 for k, v in d['kernels'].items() if k.startswith('mlp_')}
    print('%-28s %.4f ms/step (min %.4f)  ' % (spec, d['ms_per_step'], d['ms_per_step_min']) + '  '.join('%s=%.1f' % (k.replace('mlp_', ''), v) for k, v in sorted(ks.items())), flush=True)
