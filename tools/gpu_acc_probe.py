"""Accuracy of one MlpBlock forward (block 1 of cfg2) against fp64: fp32-MFMA engine, x3 engine, torch fp32 oracle."""
import os, sys, torch
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path.insert(0, ROOT)
from graph_neural_net_amd import synthetic
from graph_neural_net_amd.engine import FgnnEngine, ParamLayout
from oracle import fgnn_oracle as O
B, N = 8, 50
dev = torch.device('cuda:0')
lay = ParamLayout(2, 4, 32, 32, 3)
params = lay.init_flat(0, dev)
x1, x2 = synthetic.make_batch(2000, B, N, 'Regular', 0.2, 0.1)
sd = {k: v.clone() for k, v in lay.unflatten(params.cpu()).items()}
sd64 = {k: v.double() for k, v in sd.items()}
s64, l64, g64 = O.step_fwd_bwd(x1.double(), x2.double(), sd64)
s32, l32, g32 = O.step_fwd_bwd(x1, x2, sd)
keys = [k for k in g64 if not k.endswith('convs.2.bias')]
flat = lambda g: torch.cat([g[k].reshape(-1).double().cpu() for k in keys])
rel = lambda a, b: ((a.double().cpu() - b.double().cpu()).abs().max() / b.abs().max()).item()
print('oracle fp32: scores %.3e  grads L2 %.3e' % (rel(s32, s64), ((flat(g32) - flat(g64)).norm() / flat(g64).norm()).item()))
x = torch.cat([x1, x2]).contiguous().to(dev)
for mode in ('f32', 'x3'):
    eng = FgnnEngine(lay, 2 * B, N, dev, mfma=mode)
    g = torch.zeros_like(params)
    sc, loss = eng.step(params, g, x)
    torch.cuda.synchronize()
    gd = {k: v for k, v in lay.unflatten(g).items()}
    print('%s engine : scores %.3e  grads L2 %.3e' % (mode, rel(sc, s64), ((flat(gd) - flat(g64)).norm() / flat(g64).norm()).item()))
