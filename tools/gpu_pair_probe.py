"""mlp1 + mlp2 backward as one launch (fgnn_mlp_bwd_pair) vs two launches: agreement and captured step time."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graph_neural_net_amd import _lib, synthetic
from graph_neural_net_amd.engine import FgnnEngine, ParamLayout
B, N = 32, 50
dev = torch.device('cuda:0')
lay = ParamLayout(2, 4, 32, 32, 3)
params = lay.init_flat(0, dev)
x1, x2 = synthetic.make_batch(2000, B, N, 'Regular', 0.2, 0.1)
x = torch.cat([x1, x2]).contiguous().to(dev)
res = {}
for pair in (False, True):
    FgnnEngine.PAIR_BWD = pair
    eng = FgnnEngine(lay, 2 * B, N, dev, mfma='f32')
    g = torch.zeros_like(params)
    sc, loss = eng.step(params, g, x)
    torch.cuda.synchronize()
    W = eng._bwd
    res[pair] = (sc.clone(), loss.clone(), g.clone(), W['dy'][0].clone(), W['dy'][1].clone())
    eng.step(params, g, x); torch.cuda.synchronize()
    assert torch.equal(g, res[pair][2]), 'not reproducible'
    _lib.PROFILE = []
    for _ in range(10):
        eng.step(params, g, x)
    torch.cuda.synchronize()
    rec, _lib.PROFILE = _lib.PROFILE, None
    acc = {}
    for tag, e0, e1, *_ in rec:
        a = acc.setdefault(tag, [0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1)
    print(pair, ' '.join('%s=%.1f' % (k, v[1] / v[0] * 1e3) for k, v in sorted(acc.items()) if 'mlp_bwd' in k))
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        eng.step(params, g, x)
    for _ in range(64): gr.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(200): gr.replay()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 200 * 1e3)
    print(pair, 'captured ms/step', ['%.4f' % t for t in sorted(ts)])
a, b = res[False], res[True]
print('scores equal', torch.equal(a[0], b[0]), ' loss', a[1].item(), b[1].item())
print('d_in slabs bit-identical:', torch.equal(a[3], b[3]), torch.equal(a[4], b[4]))
print('grads rel L2 diff %.3e  max-norm %.3e' % (((a[2] - b[2]).norm() / a[2].norm()).item(), ((a[2] - b[2]).abs().max() / a[2].abs().max()).item()))
