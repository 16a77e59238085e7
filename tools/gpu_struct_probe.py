"""Event-timed launches of the structured block-1 kernels (and alt builds): python tools/gpu_struct_probe.py [lib names...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] != '--child':
    for name in sys.argv[1:]:
        r = subprocess.run([sys.executable, __file__, '--child', name], capture_output=True, text=True)
        print(name, r.stdout.strip() or r.stderr[-300:], flush=True)
    raise SystemExit(0)
sys.path.insert(0, ROOT)
import numpy as np, torch
from graph_neural_net_amd import _lib, synthetic
name = sys.argv[2]
if name != 'main':
    _lib.LIB_PATH = os.path.join(ROOT, 'graph_neural_net_amd', '_dbg', 'libfgnn_hip_%s.so' % name)
from graph_neural_net_amd.engine import FgnnEngine, ParamLayout
B, N = 32, 50
dev = torch.device('cuda:0')
lay = ParamLayout(2, 4, 32, 32, 3)
params = lay.init_flat(0, dev)
x1, x2 = synthetic.make_batch(2000, B, N, 'Regular', 0.2, 0.1)
bits = torch.from_numpy(synthetic.pack_adjacency(torch.cat([x1, x2])[:, 0].numpy()).view(np.int32)).to(dev)
eng = FgnnEngine(lay, 2 * B, N, dev, block1='structured')
g = torch.zeros_like(params)
eng.step(params, g, None, bits=bits)
torch.cuda.synchronize()
out = []
for fn in (lambda: eng._struct_fwd(params), lambda: eng._struct_bwd(params)):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(50): fn()
    e1.record()
    torch.cuda.synchronize()
    out.append(e0.elapsed_time(e1) / 50 * 1e3)
print('tables+fwd %.1f us   bwd (reduce + params) %.1f us' % tuple(out))
