import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from util import *
from graph_neural_net_amd.engine import FgnnEngine, ParamLayout
from oracle import fgnn_oracle as O
DEV='cuda:0'
d = load_golden('cfg2_reg_n50_b32_4blk.npz')
sd = sub(load_golden('cfg2_reg_n50_b2_4blk.npz'), 'sd/')
sd64 = {k: v.double() for k, v in sd.items()}
n = int(d['n']); x1, x2 = unpack_pairs(d['bits1'], n), unpack_pairs(d['bits2'], n)
lay = ParamLayout(2, 4, 32, 32, 3)
params = lay.flatten(sd, DEV)
torch.set_num_threads(16)
keys = [k for k in sub(d, 'grad/') if not is_zero_grad(k)]
fresh = len(sys.argv) > 1
eng = FgnnEngine(lay, 2, n, DEV)
for b in range(10):
    if fresh: eng = FgnnEngine(lay, 2, n, DEV)
    g = torch.zeros_like(params)
    eng.step(params, g, torch.cat([x1[b:b+1], x2[b:b+1]]).contiguous().to(DEV)); torch.cuda.synchronize()
    got = lay.unflatten(g.cpu())
    _, _, g64 = O.step_fwd_bwd(x1[b:b+1].double(), x2[b:b+1].double(), sd64)
    _, _, g32 = O.step_fwd_bwd(x1[b:b+1], x2[b:b+1], sd)
    k = 'ne_bm_block2_mlp1.gn.bias'
    print('pair %d: flat ours %.2e o32 %.2e | %s ours %.2e o32 %.2e' % (b, l2rel(flat_of(got, keys), flat_of(g64, keys)), l2rel(flat_of(g32, keys), flat_of(g64, keys)),
          k, l2rel(got[k], g64[k]), l2rel(g32[k], g64[k])))
