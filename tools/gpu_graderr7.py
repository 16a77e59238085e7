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
eng = FgnnEngine(lay, 64, n, DEV)
torch.set_num_threads(32)
x = torch.cat([x1, x2]).contiguous()
eng.embed(params, x.to(DEV)); torch.cuda.synchronize()
k64, k32 = {}, {}
O.node_embedding(x.double(), sd64, k64); O.node_embedding(x, sd, k32)
y64, y32 = k64['ne/bm/block4/mlp3'], k32['ne/bm/block4/mlp3']
i64, i32 = y64.max(-1)[1], y32.max(-1)[1]
io = eng.idx.cpu().long()
top2 = y64.topk(2, dim=-1)[0]; gap = (top2[..., 0] - top2[..., 1]) / y64.abs().max()
mo, mr = (io != i64), (i32 != i64)
print('argmax mismatches vs fp64 over the 64 graphs: ours %d, oracle fp32 %d (of %d rows)' % (int(mo.sum()), int(mr.sum()), i64.numel()))
print('ours: graphs', sorted(set(mo.nonzero()[:, 0].tolist())), 'gaps', ['%.1e' % v for v in gap[mo].tolist()])
print('oracle32: graphs', sorted(set(mr.nonzero()[:, 0].tolist())), 'gaps', ['%.1e' % v for v in gap[mr].tolist()])
print('rows with relative gap < 1e-5: %d; < 1e-6: %d' % (int((gap < 1e-5).sum()), int((gap < 1e-6).sum())))
