"""16-pixel-tile MLP kernels (csrc/*_t16.hip) against the 32-pixel kernels: agreement of one step + per-launch and captured step time.
usage: python tools/gpu_t16_probe.py [which ...]   which = comma lists for FGNN_T16, e.g. pair  pair,bwd  pair,bwd,fwd"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graph_neural_net_amd import _lib, synthetic
from graph_neural_net_amd.engine import FgnnEngine, ParamLayout
B, N = int(os.environ.get('PROBE_B', 32)), int(os.environ.get('PROBE_N', 50))
variants = ['0'] + (sys.argv[1:] or ['pair'])
dev = torch.device('cuda:0')
lay = ParamLayout(2, 4, 32, 32, 3)
params = lay.init_flat(0, dev)
x1, x2 = synthetic.make_batch(2000, B, N, 'Regular', 0.2, 0.1)
x = torch.cat([x1, x2]).contiguous().to(dev)
res = {}
for var in variants:
    FgnnEngine.T16 = var
    eng = FgnnEngine(lay, 2 * B, N, dev, mfma='f32')
    g = torch.zeros_like(params)
    sc, loss = eng.step(params, g, x)
    torch.cuda.synchronize()
    W = eng._bwd
    res[var] = dict(sc=sc.clone(), loss=loss.clone(), g=g.clone(), dy0=W['dy'][0].clone(), dy1=W['dy'][1].clone(), s12part=W['s12part'].clone(),
                    z=[eng.z[k].clone() for k in sorted(eng.z)], nrm=[eng.nrm[k].clone() for k in sorted(eng.nrm)])
    eng.step(params, g, x); torch.cuda.synchronize()
    print(var, 'reproducible:', torch.equal(g, res[var]['g']))
    _lib.PROFILE = []
    for _ in range(10):
        eng.step(params, g, x)
    torch.cuda.synchronize()
    rec, _lib.PROFILE = _lib.PROFILE, None
    acc = {}
    for tag, e0, e1, *_ in rec:
        a = acc.setdefault(tag, [0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1)
    print(var, ' '.join('%s=%.1f' % (k, v[1] / v[0] * 1e3) for k, v in sorted(acc.items()) if 'mlp_' in k))
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        eng.step(params, g, x)
    for _ in range(64): gr.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(200): gr.replay()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 200 * 1e3)
    print(var, 'captured ms/step', ['%.4f' % t for t in sorted(ts)])
    del gr, eng
a = res['0']
def rel(u, v):
    return ((u - v).norm() / v.norm().clamp_min(1e-30)).item(), ((u - v).abs().max() / v.abs().max().clamp_min(1e-30)).item()
for var in variants[1:]:
    b = res[var]
    print('== %s vs 32-pixel kernels' % var)
    print('  scores equal', torch.equal(a['sc'], b['sc']), ' loss', a['loss'].item(), b['loss'].item())
    print('  forward z bit-identical:', [bool(torch.equal(u, v)) for u, v in zip(a['z'], b['z'])].count(True), 'of', len(a['z']), ' max rel L2 %.2e' % max(rel(v, u)[0] for u, v in zip(a['z'], b['z'])),
          ' nrm bit-identical:', [bool(torch.equal(u, v)) for u, v in zip(a['nrm'], b['nrm'])].count(True), 'of', len(a['nrm']), ' max rel %.2e' % max(rel(v, u)[0] for u, v in zip(a['nrm'], b['nrm'])))
    print('  scores rel L2 / max: %.3e %.3e' % rel(b['sc'], a['sc']))
    print('  d_in slabs rel L2 / max: %.3e %.3e | %.3e %.3e' % (rel(b['dy0'], a['dy0']) + rel(b['dy1'], a['dy1'])))
    print('  s12part rel L2 / max: %.3e %.3e' % rel(b['s12part'], a['s12part']))
    print('  grads rel L2 / max: %.3e %.3e   finite: %s' % (rel(b['g'], a['g']) + (bool(torch.isfinite(b['g']).all()),)))
    L = lay
    worst = []
    for kj, r in L.mlp.items():
        lo, hi = r['off'], r['off'] + r['count']
        worst.append((rel(b['g'][lo:hi], a['g'][lo:hi])[0], kj))
    worst.sort(reverse=True)
    print('  worst tensors:', ['%s %.2e' % (kj, e) for e, kj in worst[:4]])
