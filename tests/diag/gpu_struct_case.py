#!/usr/bin/env python3
"""One dense directed batch at N = 256 through generic / structured block 1 and the fp64 oracle, per-tensor errors:
python tests/diag/gpu_struct_case.py [N=256] [blocks=2] [dens=0.95]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
from graph_neural_net_amd import synthetic
from graph_neural_net_amd.engine import FgnnEngine, ParamLayout
from oracle import fgnn_oracle as O
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dens = float(sys.argv[3]) if len(sys.argv) > 3 else 0.95
B, DEV = 2, 'cuda:0'
rng = np.random.default_rng(46)
ws = (rng.random((2 * B, N, N)) < dens).astype(np.float32)
x = torch.zeros(2 * B, 2, N, N)
x[:, 0] = torch.from_numpy(ws)
for g in range(2 * B):
    x[g, 1] = torch.diag(x[g, 0].sum(-1))
bits = torch.from_numpy(synthetic.pack_adjacency(ws).view(np.int32)).to(DEV)
lay = ParamLayout(2, nblk, 32, 32, 3)
torch.manual_seed(3)
sd = O.init_state_dict(num_blocks=nblk)
params = lay.flatten(sd, DEV)
res = {}
for mode in ('generic', 'structured'):
    eng = FgnnEngine(lay, 2 * B, N, DEV, block1=mode)
    g = torch.zeros_like(params)
    s, l = eng.step(params, g, None, bits=bits)
    torch.cuda.synchronize()
    res[mode] = (s.cpu(), l.item(), lay.unflatten(g.cpu()))
_, _, g64 = O.step_fwd_bwd(x[:B].double(), x[B:].double(), {k: v.double() for k, v in sd.items()})
_, _, g32 = O.step_fwd_bwd(x[:B], x[B:], sd)
rel = lambda a, b: ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()
print('%-40s %10s %10s %10s' % ('tensor', 'generic', 'structured', 'oracle32'))
for k in g64:
    print('%-40s %10.2e %10.2e %10.2e' % (k, rel(res['generic'][2][k], g64[k]), rel(res['structured'][2][k], g64[k]), rel(g32[k], g64[k])))
