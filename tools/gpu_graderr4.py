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
eng = FgnnEngine(lay, 2, n, DEV)
torch.set_num_threads(16)
l2 = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm()).item()
for b in range(6):
    g = torch.zeros_like(params)
    x = torch.cat([x1[b:b+1], x2[b:b+1]]).contiguous()
    eng.step(params, g, x.to(DEV)); torch.cuda.synchronize()
    k64, k32 = {}, {}
    O.node_embedding(x.double(), sd64, k64); O.node_embedding(x, sd, k32)
    y64, y32 = k64['ne/bm/block4/mlp3'], k32['ne/bm/block4/mlp3']
    i64, i32 = y64.max(-1)[1], y32.max(-1)[1]
    ours_y = eng.normalized(4, 3, params).cpu()
    top2 = y64.topk(2, dim=-1)[0]; gap = (top2[..., 0] - top2[..., 1])
    mis_o = (eng.idx.cpu().long() != i64); mis_r = (i32 != i64)
    print('pair %d: y4 err ours %.2e oracle32 %.2e | argmax mismatches vs fp64: ours %d oracle32 %d of %d | min gap %.2e; gaps at our mismatches: %s'
          % (b, l2(ours_y, y64), l2(y32, y64), int(mis_o.sum()), int(mis_r.sum()), i64.numel(), gap.min().item(),
             ['%.1e' % v for v in gap[mis_o].tolist()[:6]]))
