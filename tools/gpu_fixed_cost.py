"""Measure the fixed (prologue + epilogue) cost of the MLP kernels: tiny problem vs the bench problem."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graph_neural_net_amd import _lib, synthetic
from graph_neural_net_amd.engine import FgnnEngine, ParamLayout

def run(B, N, reps=20):
    lay = ParamLayout(2, 4, 32, 32, 3)
    dev = torch.device('cuda:0')
    params = lay.init_flat(0, dev); grads = torch.zeros_like(params)
    x1, x2 = synthetic.make_batch(1, B, N, 'ErdosRenyi', 0.3, 0.1)
    x = torch.cat([x1, x2]).contiguous().to(dev)
    eng = FgnnEngine(lay, 2 * B, N, dev)
    for _ in range(3): eng.step(params, grads, x)
    torch.cuda.synchronize()
    _lib.PROFILE = []
    for _ in range(reps): eng.step(params, grads, x)
    torch.cuda.synchronize()
    rec, _lib.PROFILE = _lib.PROFILE, None
    acc = {}
    for tag, e0, e1, *_ in rec:
        a = acc.setdefault(tag, [0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1)
    print('B=%d N=%d (tiles=%d)' % (B, N, 2 * B * ((N * N + 31) // 32)))
    print('   ' + '  '.join('%s=%.1f' % (k.replace('fgnn_','').replace('chan_',''), v[1] / v[0] * 1e3) for k, v in sorted(acc.items()) if 'mlp' in k or 'matmul' in k))

for b in ([int(a) for a in sys.argv[1:]] or (1, 8, 16, 32, 64, 128)):
    run(b, 50, reps=10)
