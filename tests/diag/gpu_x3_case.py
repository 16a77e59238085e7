"""Diagnose a parity case per pair and per tensor: x3 engine, fp32-MFMA engine, oracle fp32, all against the fp64 oracle.
    python tests/diag/gpu_x3_case.py B N [blocks] [seed]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from graph_neural_net_amd import synthetic
from graph_neural_net_amd.engine import FgnnEngine, ParamLayout
from oracle import fgnn_oracle as O

B, N = int(sys.argv[1]), int(sys.argv[2])
K = int(sys.argv[3]) if len(sys.argv) > 3 else 2
DEV = torch.device('cuda:0')
torch.manual_seed(100 + N)
sd = O.init_state_dict(num_blocks=K)
x1, x2 = synthetic.make_batch(9000 + N, B, N, 'ErdosRenyi', 0.5, 0.1)
lay = ParamLayout(2, K, 32, 32, 3)
params = lay.flatten(sd, DEV)
sd64 = {k: v.double() for k, v in sd.items()}


def run(mode, a, b):
    eng = FgnnEngine(lay, 2 * a.shape[0], N, DEV, mfma=mode)
    g = torch.zeros_like(params)
    sc, loss = eng.step(params, g, torch.cat([a, b]).contiguous().to(DEV))
    torch.cuda.synchronize()
    return sc.cpu(), {k: v.cpu() for k, v in lay.unflatten(g).items()}


keys = [k for k in sd if not k.endswith('convs.2.bias')]
flat = lambda g: torch.cat([g[k].reshape(-1).double() for k in keys])
for b in list(range(B)) + [None]:
    a1, a2 = (x1, x2) if b is None else (x1[b:b + 1], x2[b:b + 1])
    s64, l64, g64 = O.step_fwd_bwd(a1.double(), a2.double(), sd64)
    s32, l32, g32 = O.step_fwd_bwd(a1, a2, sd)
    t = flat(g64)
    row = ['pair %s' % ('all' if b is None else b), 'oracle32 %.2e' % ((flat(g32) - t).norm() / t.norm()).item()]
    for mode in ('f32', 'x3'):
        sc, g = run(mode, a1, a2)
        row.append('%s grads %.2e scores %.2e' % (mode, ((flat(g) - t).norm() / t.norm()).item(), ((sc.double() - s64).abs().max() / s64.abs().max()).item()))
        if b is None:
            worst = sorted(((g[k].double() - g64[k]).norm().item() / (g64[k].norm().item() + 1e-30), k) for k in keys)[-3:]
            row.append(str([(round(v, 5), k) for v, k in worst]))
    print('  '.join(row))
