import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from util import *
from graph_neural_net_amd.engine import FgnnEngine, ParamLayout
DEV='cuda:0'
def run(name, B=None):
    d = load_golden(name)
    sd = sub(load_golden('cfg2_reg_n50_b2_4blk.npz'), 'sd/')
    if 'bits1' in d:
        n = int(d['n']); x1, x2 = unpack_pairs(d['bits1'], n), unpack_pairs(d['bits2'], n)
    else:
        x1, x2 = d['x1'], d['x2']
    lay = ParamLayout(2, 4, 32, 32, 3)
    params = lay.flatten(sd, DEV); grads = torch.zeros_like(params)
    eng = FgnnEngine(lay, 2 * x1.shape[0], x1.shape[-1], DEV)
    eng.step(params, grads, torch.cat([x1, x2]).contiguous().to(DEV)); torch.cuda.synchronize()
    got = lay.unflatten(grads.cpu())
    keys = [k for k in sub(d, 'grad/') if not is_zero_grad(k)]
    g64 = flat_of(sub(d, 'grad64/'), keys)
    print(name, 'flat L2: ours %.3e theirs %.3e' % (l2rel(flat_of(got, keys), g64), l2rel(flat_of(sub(d, 'grad/'), keys), g64)))
    rows = []
    for k in keys:
        r64 = d['grad64/' + k].double()
        eo = (got[k].double() - r64).norm().item(); et = (d['grad/' + k].double() - r64).norm().item()
        rows.append((eo**2, et**2, k, r64.norm().item()))
    tot_o = sum(r[0] for r in rows); tot_t = sum(r[1] for r in rows)
    rows.sort(reverse=True)
    for eo2, et2, k, nr in rows[:10]:
        print('   %-44s ours %.2e (%.0f%%) theirs %.2e (%.0f%%) |g| %.2e' % (k, eo2**.5, 100*eo2/tot_o, et2**.5, 100*et2/tot_t, nr))
run('cfg2_reg_n50_b32_4blk.npz')
run('cfg2_reg_n50_b2_4blk.npz')
