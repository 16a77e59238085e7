"""bisect the backward: gradient w.r.t. every intermediate (HIP fp32 engine, snapshots after each launch) vs fp64 autograd,
next to the fp32 oracle's own error"""
import sys, torch, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from util import *
from graph_neural_net_amd import _lib
from graph_neural_net_amd.engine import FgnnEngine, ParamLayout
from oracle import fgnn_oracle as O
DEV='cuda:0'
d = load_golden('cfg2_reg_n50_b32_4blk.npz')
sd = sub(load_golden('cfg2_reg_n50_b2_4blk.npz'), 'sd/')
n = int(d['n']); x1, x2 = unpack_pairs(d['bits1'], n), unpack_pairs(d['bits2'], n)
b = int(sys.argv[1]) if len(sys.argv) > 1 else 0
x = torch.cat([x1[b:b+1], x2[b:b+1]]).contiguous()
lay = ParamLayout(2, 4, 32, 32, 3)
params = lay.flatten(sd, DEV)
eng = FgnnEngine(lay, 2, n, DEV)
l2 = lambda a, b_: ((a.double() - b_.double()).norm() / b_.double().norm()).item()

def oracle_grads(dtype):
    s = {k: v.to(dtype).requires_grad_(True) for k, v in sd.items()}
    keep = {}
    e = O.node_embedding(x.to(dtype), s, keep)
    for v in keep.values(): v.retain_grad()
    sc = torch.matmul(e[:1].transpose(1, 2), e[1:])
    loss = O.triplet_loss_mean(sc); loss.backward()
    return {k: v.grad for k, v in keep.items()}, {k: v.grad for k, v in s.items()}
k64, p64 = oracle_grads(torch.float64); k32, p32 = oracle_grads(torch.float32)

snaps = {}
orig_call = _lib.call
g = torch.zeros_like(params)
eng.forward(params, x.to(DEV), defer_loss=True)
W = eng._alloc_bwd()
state = {'k': 4}
def call(name, *args, tag=None):
    orig_call(name, *args, tag=tag)
    torch.cuda.synchronize()
    k = state['k']
    if name == 'fgnn_mlp_bwd' and tag.startswith('mlp_bwd[cin=64') or (name == 'fgnn_mlp_bwd' and tag.startswith('mlp_bwd[cin=34')):
        snaps[('dmult', k)] = eng.unpadded(W['dmult']).cpu()
    if name == 'fgnn_chan_matmul_bwd':
        snaps[('dy1', k)] = eng.unpadded(W['dy1']).cpu(); snaps[('dy2', k)] = eng.unpadded(W['dy2']).cpu()
        state['n12'] = 0
    if name == 'fgnn_mlp_bwd' and (tag.startswith('mlp_bwd[cin=32,dx=32') or tag.startswith('mlp_bwd[cin=2,')):
        state['n12'] += 1
        if state['n12'] == 2:
            if k > 1: snaps[('din', k)] = eng.unpadded(W['dy'][(4 - k + 1) % 2]).cpu()
            state['k'] = k - 1
_lib.call = call
import graph_neural_net_amd.engine as E
eng.backward(params, g)
_lib.call = orig_call
for k in (4, 3, 2, 1):
    pfx = 'ne/bm/block%d/' % k
    row = 'blk %d: ' % k
    row += 'dmult ours %.1e o32 %.1e | ' % (l2(snaps[('dmult', k)], k64[pfx + 'mult']), l2(k32[pfx + 'mult'], k64[pfx + 'mult']))
    row += 'dy1 ours %.1e o32 %.1e | dy2 ours %.1e o32 %.1e | ' % (l2(snaps[('dy1', k)], k64[pfx + 'mlp1']), l2(k32[pfx + 'mlp1'], k64[pfx + 'mlp1']),
                                                                l2(snaps[('dy2', k)], k64[pfx + 'mlp2']), l2(k32[pfx + 'mlp2'], k64[pfx + 'mlp2']))
    if k > 1:
        q = 'ne/bm/block%d/mlp3' % (k - 1)
        row += 'din ours %.1e o32 %.1e' % (l2(snaps[('din', k)], k64[q]), l2(k32[q], k64[q]))
    print(row)
got = lay.unflatten(g.cpu())
for kk in ('ne_bm_block4_mlp3.convs.0.weight', 'ne_bm_block4_mlp1.gn.bias', 'ne_bm_block2_mlp1.gn.bias', 'ne_bm_block2_mlp2.convs.1.bias'):
    print('%-36s ours %.2e o32 %.2e' % (kk, l2(got[kk], p64[kk]), l2(p32[kk], p64[kk])))
