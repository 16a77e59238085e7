"""fp32-MFMA engine vs the x3 (split-bf16 MFMA) engine: agreement, per-kernel times, captured step time.
    python tools/gpu_x3_probe.py [B] [N]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graph_neural_net_amd import _lib, synthetic
from graph_neural_net_amd.engine import FgnnEngine, ParamLayout

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device('cuda:0')
lay = ParamLayout(2, 4, 32, 32, 3)
params = lay.init_flat(0, dev)
x1, x2 = synthetic.make_batch(2000, B, N, 'Regular', 0.2, 0.1)
x = torch.cat([x1, x2]).contiguous().to(dev)


def capture(work):
    work(); work()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        work()
    return g


def timeit(g, reps=200):
    for _ in range(64):
        g.replay()
    torch.cuda.synchronize()
    out = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / reps * 1e3)
    return sorted(out)


def profile(eng, grads, reps=10):
    _lib.PROFILE = []
    for _ in range(reps):
        eng.step(params, grads, x)
    torch.cuda.synchronize()
    rec, _lib.PROFILE = _lib.PROFILE, None
    acc = {}
    for tag, e0, e1, *_ in rec:
        a = acc.setdefault(tag, [0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1)
    return {k: v[1] / v[0] * 1e3 for k, v in acc.items()}


res = {}
for mode in ('f32', 'x3'):
    eng = FgnnEngine(lay, 2 * B, N, dev, mfma=mode)
    g = torch.zeros_like(params)
    sc, loss = eng.step(params, g, x)
    torch.cuda.synchronize()
    res[mode] = (sc.clone(), loss.clone(), g.clone(), {k: eng.unpadded(v).clone() for k, v in eng.z.items() if k[0] in (1, 4)})
    prof = profile(eng, g)
    print(mode, ' '.join('%s=%.1f' % (k.replace('fgnn_', '').replace('chan_', ''), v) for k, v in sorted(prof.items()) if 'mlp' in k or 'pack' in k))
    gr = capture(lambda: eng.step(params, g, x))
    print(mode, 'captured ms/step', ['%.4f' % v for v in timeit(gr)])
a, b = res['f32'], res['x3']
rel = lambda u, v: ((u - v).abs().max() / u.abs().max()).item()
print('scores max-norm rel diff %.3e   loss %.7f / %.7f   grads rel L2 %.3e' % (rel(a[0], b[0]), a[1].item(), b[1].item(), ((a[2] - b[2]).norm() / a[2].norm()).item()))
for k in sorted(a[3]):
    print('  z%s max-norm rel diff %.3e' % (k, rel(a[3][k], b[3][k])))
