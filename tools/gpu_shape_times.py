"""Per-kernel times of one engine step at an arbitrary shape: python tools/gpu_shape_times.py B N [ragged_lo]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graph_neural_net_amd import _lib, synthetic
from graph_neural_net_amd.engine import FgnnEngine, ParamLayout

B, N = int(sys.argv[1]), int(sys.argv[2])
lo = int(sys.argv[3]) if len(sys.argv) > 3 else None
lay = ParamLayout(2, 4, 32, 32, 3)
dev = torch.device('cuda:0')
params = lay.init_flat(0, dev); grads = torch.zeros_like(params)
nv = None
if lo is None:
    x1, x2 = synthetic.make_batch(1, B, N, 'ErdosRenyi', 0.5, 0.1)
else:
    xs, ys = synthetic.make_ragged_batch(1, B, lo, N)
    nmax = max(x.shape[-1] for x in xs)
    pad = lambda t: torch.nn.functional.pad(t, (0, nmax - t.shape[-1], 0, nmax - t.shape[-1]))
    x1, x2 = torch.stack([pad(x) for x in xs]), torch.stack([pad(y) for y in ys])
    n1 = torch.tensor([x.shape[-1] for x in xs], dtype=torch.int32)
    N = x1.shape[-1]
    nv = torch.cat([n1, n1]).to(dev)
x = torch.cat([x1, x2]).contiguous().to(dev)
eng = FgnnEngine(lay, 2 * B, N, dev, ragged=nv is not None)
for _ in range(3): eng.step(params, grads, x, nvalid=nv)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(10): eng.step(params, grads, x, nvalid=nv)
torch.cuda.synchronize()
print('B=%d N=%d ragged=%s: %.3f ms/step eager' % (B, N, nv is not None, (time.perf_counter() - t0) * 100))
_lib.PROFILE = []
for _ in range(5): eng.step(params, grads, x, nvalid=nv)
torch.cuda.synchronize()
rec, _lib.PROFILE = _lib.PROFILE, None
acc = {}
for tag, e0, e1, *_ in rec:
    a = acc.setdefault(tag, [0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1)
tot = sum(v[1] for v in acc.values())
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print('  %-28s x%-3d avg %8.1f us  %5.1f%%' % (k, v[0] // 5, v[1] / v[0] * 1e3, 100 * v[1] / tot))
