"""Event-timed launches of the t16 kernels with ablated builds (graph_neural_net_amd/_dbg/libfgnn_hip_<name>.so, tools/build_variant.sh):
one fresh process per build.  usage: python tools/gpu_t16_ablate.py main abl1 abl2 ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 or (len(sys.argv) == 2 and sys.argv[1] != '--child'):
    for name in sys.argv[1:]:
        env = dict(os.environ)
        if name != 'main':
            env['FGNN_LIB'] = os.path.join(ROOT, 'graph_neural_net_amd', '_dbg', 'libfgnn_hip_%s.so' % name)
        out = subprocess.run([sys.executable, __file__, '--child'], env=env, capture_output=True, text=True)
        print('%-8s %s' % (name, out.stdout.strip().split('\n')[-1] if out.returncode == 0 else 'FAILED ' + out.stderr[-300:]))
    sys.exit(0)
import torch
sys.path.insert(0, ROOT)
from graph_neural_net_amd import _lib, synthetic
from graph_neural_net_amd.engine import FgnnEngine, ParamLayout
B, N = 32, 50
dev = torch.device('cuda:0')
lay = ParamLayout(2, 4, 32, 32, 3)
params = lay.init_flat(0, dev)
x1, x2 = synthetic.make_batch(2000, B, N, 'Regular', 0.2, 0.1)
x = torch.cat([x1, x2]).contiguous().to(dev)
eng = FgnnEngine(lay, 2 * B, N, dev, mfma='f32')
g = torch.zeros_like(params)
for _ in range(3): eng.step(params, g, x)
torch.cuda.synchronize()
_lib.PROFILE = []
for _ in range(20): eng.step(params, g, x)
torch.cuda.synchronize()
rec, _lib.PROFILE = _lib.PROFILE, None
acc = {}
for tag, e0, e1, *_ in rec:
    a = acc.setdefault(tag, []); a.append(e0.elapsed_time(e1) * 1e3)
print(' '.join('%s=%.1f' % (k, sorted(v)[len(v) // 2]) for k, v in sorted(acc.items()) if 'mlp_bwd' in k))
