"""Structured vs generic block 1 against the fp64 oracle (one block, B pairs of N=50 regular graphs): who is closer."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
from graph_neural_net_amd import synthetic
from graph_neural_net_amd.engine import FgnnEngine, ParamLayout
from oracle import fgnn_oracle as O
from util import is_zero_grad, load_golden, rel, sub
DEV = 'cuda:0'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = 50
sd = sub(load_golden('cfg1_er_n20_b4_1blk.npz'), 'sd/')
lay = ParamLayout(2, 1, 32, 32, 3)
params = lay.flatten(sd, DEV)
x1, x2 = synthetic.make_batch(4200 + N, B, N, 'Regular', 0.2, 0.1)
bits = torch.from_numpy(synthetic.pack_adjacency(torch.cat([x1, x2])[:, 0].numpy()).view(np.int32)).to(DEV)
_, _, g64 = O.step_fwd_bwd(x1.double(), x2.double(), {k: v.double() for k, v in sd.items()})
_, _, g32 = O.step_fwd_bwd(x1, x2, sd)
res = {}
for mode in ('generic', 'structured'):
    eng = FgnnEngine(lay, 2 * B, N, DEV, block1=mode)
    g = torch.zeros_like(params)
    eng.step(params, g, None, bits=bits)
    torch.cuda.synchronize()
    res[mode] = (lay.unflatten(g.cpu().clone()), eng._bwd['s12'][(1, 1)].cpu().clone().view(-1, 32, 2), eng._bwd['s12'][(1, 2)].cpu().clone().view(-1, 32, 2))
for k in g64:
    if is_zero_grad(k): continue
    print('%-34s oracle32 %.2e  generic %.2e  structured %.2e   struct-vs-generic %.2e' % (k, rel(g32[k], g64[k]), rel(res['generic'][0][k], g64[k]), rel(res['structured'][0][k], g64[k]), rel(res['structured'][0][k], res['generic'][0][k])))
for m in (1, 2):
    a, b = res['generic'][m], res['structured'][m]
    d = (a - b).abs()
    i = d[..., 1].argmax()
    print('s12 mlp%d: S1 rel %.2e  S2 rel %.2e; worst S2 entry: generic %.6e structured %.6e (max |S2| %.3e)' % (m, rel(b[..., 0], a[..., 0]), rel(b[..., 1], a[..., 1]), a[..., 1].flatten()[i], b[..., 1].flatten()[i], a[..., 1].abs().max()))
