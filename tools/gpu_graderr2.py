import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from util import *
from graph_neural_net_amd.engine import FgnnEngine, ParamLayout
DEV='cuda:0'
d = load_golden('cfg2_reg_n50_b32_4blk.npz')
sd = sub(load_golden('cfg2_reg_n50_b2_4blk.npz'), 'sd/')
n = int(d['n']); x1, x2 = unpack_pairs(d['bits1'], n), unpack_pairs(d['bits2'], n)
lay = ParamLayout(2, 4, 32, 32, 3)
params = lay.flatten(sd, DEV)
keys = [k for k in sub(d, 'grad/') if not is_zero_grad(k)]
g64 = flat_of(sub(d, 'grad64/'), keys)
print('theirs %.3e' % l2rel(flat_of(sub(d, 'grad/'), keys), g64))
for sb in (32, 16, 8, 4, 2, 1):
    eng = FgnnEngine(lay, 2 * sb, n, DEV)
    acc = torch.zeros_like(params, dtype=torch.float64)
    for lo in range(0, 32, sb):
        g = torch.zeros_like(params)
        eng.step(params, g, torch.cat([x1[lo:lo+sb], x2[lo:lo+sb]]).contiguous().to(DEV), total_nodes=32 * n)
        acc += g.double()
    got = lay.unflatten(acc.cpu())
    print('shards of %2d: flat L2 vs fp64 %.3e' % (sb, l2rel(flat_of(got, keys), g64)))
