"""Short view of a bench.py JSON line: python tools/show_bench.py file.json"""
import json, sys
d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')][-1])
print({k: d.get(k) for k in ('value', 'ms_per_step', 'ms_per_step_min', 'ms_per_step_max', 'timed_windows', 'dtype', 'n_gpus', 'allreduce_ms')})
print('roofline', {k: v for k, v in (d.get('roofline') or {}).items() if k in ('kernel', 'bound', 'achieved', 'peak', 'frac', 'traffic', 'avg_launch_ms')})
if d.get('cpu_baseline'):
    print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['kind'])
for k, v in (d.get('extra_configs') or {}).items():
    print(k, round(v['value'], 1), round(v['ms_per_step'], 4), v['dtype'], v['roofline'], v['wall_s'])
