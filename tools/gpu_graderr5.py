import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from util import *
from graph_neural_net_amd.engine import FgnnEngine, ParamLayout
from oracle import fgnn_oracle as O
DEV='cuda:0'
d = load_golden('cfg2_reg_n50_b32_4blk.npz')
sd = sub(load_golden('cfg2_reg_n50_b2_4blk.npz'), 'sd/')
n = int(d['n']); x1, x2 = unpack_pairs(d['bits1'], n), unpack_pairs(d['bits2'], n)
lay = ParamLayout(2, 4, 32, 32, 3)
params = lay.flatten(sd, DEV)
eng = FgnnEngine(lay, 2, n, DEV)
torch.set_num_threads(16)
l2 = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm()).item()
def dE_of(dtype, b):
    s = {k: v.to(dtype) for k, v in sd.items()}
    e1 = O.node_embedding(x1[b:b+1].to(dtype), s).detach().requires_grad_(True)
    e2 = O.node_embedding(x2[b:b+1].to(dtype), s).detach().requires_grad_(True)
    sc = torch.matmul(e1.transpose(1, 2), e2)
    loss = O.triplet_loss_mean(sc)
    g1, g2 = torch.autograd.grad(loss, [e1, e2])
    return torch.cat([g1, g2]), sc.detach(), loss.detach()
for b in range(4):
    g = torch.zeros_like(params)
    x = torch.cat([x1[b:b+1], x2[b:b+1]]).contiguous()
    sc, loss = eng.step(params, g, x.to(DEV)); torch.cuda.synchronize()
    d64, s64, l64 = dE_of(torch.float64, b); d32, s32, l32 = dE_of(torch.float32, b)
    print('pair %d: scores err ours %.2e o32 %.2e | loss ours %.2e o32 %.2e | dE err ours %.2e oracle32 %.2e'
          % (b, l2(sc.cpu(), s64), l2(s32, s64), abs(loss.item() - l64.item()) / l64.item(), abs(l32.item() - l64.item()) / l64.item(),
             l2(eng._bwd['dE'].cpu(), d64), l2(d32, d64)))
