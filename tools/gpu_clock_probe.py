#!/usr/bin/env python3
"""What the chip clocks to under a SUSTAINED replay loop of the step (no profiler): bench.py runs as a child process for ~10 s per
variant while this process -- which never touches the GPU -- polls rocm-smi (sclk, power, temperature) twice a second.
usage (GPU box): python tools/gpu_clock_probe.py "<bench args of variant 1>" "<bench args of variant 2>" ..."""
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def smi():
    try:
        out = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--showtemp', '--json'], capture_output=True, text=True, timeout=20).stdout
        d = json.loads(out)
        card = d[sorted(d)[0]]
        pick = lambda pat: next((v for k, v in card.items() if re.search(pat, k, re.I)), None)
        return {'sclk': pick(r'sclk clock speed'), 'power': pick(r'(average|current).*power'), 'temp': pick(r'junction|hotspot|edge')}
    except Exception as exc:        # noqa: BLE001
        return {'error': str(exc)[:200]}


for variant in sys.argv[1:] or ['']:
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '4000', '--windows', '3', '--warmup', '10', '--no-cpu-baseline', '--no-extra-configs',
           '--profile-steps', '0'] + variant.split()
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    samples = []
    while p.poll() is None:
        samples.append(smi())
        time.sleep(0.5)
    line = [l for l in p.stdout.read().splitlines() if l.startswith('{')]
    ms = json.loads(line[0])['ms_per_step'] if line else None
    print('%-50s ms/step %s' % (variant or '(default)', ms))
    for s in samples:
        print('    ', s)
