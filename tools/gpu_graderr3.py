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
names = ['ne_bm_block2_mlp1.gn.bias', 'ne_bm_block2_mlp2.convs.1.bias', 'ne_bm_block1_mlp3.convs.1.weight', 'ne_bm_block4_mlp3.gn.bias', 'ne_bm_block4_mlp3.convs.0.weight']
errs = {k: [] for k in names}; errs32 = {k: [] for k in names}; mags = {k: [] for k in names}
torch.set_num_threads(16)
for b in range(12):
    g = torch.zeros_like(params)
    eng.step(params, g, torch.cat([x1[b:b+1], x2[b:b+1]]).contiguous().to(DEV))
    got = lay.unflatten(g.cpu())
    _, _, g64 = O.step_fwd_bwd(x1[b:b+1].double(), x2[b:b+1].double(), sd64)
    _, _, g32 = O.step_fwd_bwd(x1[b:b+1], x2[b:b+1], sd)
    for k in names:
        errs[k].append((got[k].double() - g64[k]).reshape(-1))
        errs32[k].append((g32[k].double() - g64[k]).reshape(-1))
        mags[k].append(g64[k].reshape(-1))
for k in names:
    e = torch.stack(errs[k]); e32 = torch.stack(errs32[k]); m = torch.stack(mags[k])
    # bias = norm of the mean error; noise = rms of the per-pair deviation
    print('%-36s |g| %.2e | ours: |mean err| %.2e rms err %.2e | oracle fp32: |mean err| %.2e rms %.2e' % (
        k, m.mean(0).norm().item(), e.mean(0).norm().item(), e.norm(dim=1).pow(2).mean().sqrt().item(),
        e32.mean(0).norm().item(), e32.norm(dim=1).pow(2).mean().sqrt().item()))
