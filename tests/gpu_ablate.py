import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, torch
sys.path.insert(0, %r)
from graph_neural_net_amd import _lib, synthetic
from graph_neural_net_amd.engine import FgnnEngine, ParamLayout
lay = ParamLayout(2, 4, 32, 32, 3); dev = torch.device('cuda:0')
params = lay.init_flat(0, dev); grads = torch.zeros_like(params)
x1, x2 = synthetic.make_batch(1, 32, 50); x = torch.cat([x1, x2]).contiguous().to(dev)
eng = FgnnEngine(lay, 64, 50, dev)
for _ in range(3): eng.step(params, grads, x)
torch.cuda.synchronize(); _lib.PROFILE = []
for _ in range(10): eng.step(params, grads, x)
torch.cuda.synchronize(); rec, _lib.PROFILE = _lib.PROFILE, None
acc = {}
for tag, e0, e1 in rec:
    a = acc.setdefault(tag, [0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1)
print(' '.join('%%s=%%.1f' %% (k.replace('mlp_bwd', 'B'), v[1] / v[0] * 1e3) for k, v in sorted(acc.items()) if k.startswith('mlp_bwd')))
''' % ROOT
for ab in (0, 1, 2, 4, 8, 16, 3, 7, 31):
    out = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, FGNN_ABLATE=str(ab)), capture_output=True, text=True)
    print('ablate=%2d  %s' % (ab, out.stdout.strip().split('\n')[-1] if out.stdout.strip() else out.stderr[-300:]))
