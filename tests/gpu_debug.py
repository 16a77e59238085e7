"""Stage-by-stage parity report of the HIP path against the oracle (run on the GPU box)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graph_neural_net_amd.engine import FgnnEngine, ParamLayout
from oracle import fgnn_oracle as O

def rel(a, b):
    return O.max_rel_err(a, b)

def run_case(name, nblk):
    d = np.load(os.path.join(ROOT, 'tests', 'golden', name))
    sd = {k[3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith('sd/')}
    x1, x2 = torch.from_numpy(d['x1']), torch.from_numpy(d['x2'])
    B, _, N, _ = x1.shape
    lay = ParamLayout(2, nblk, 32, 32, 3)
    dev = torch.device('cuda:0')
    params = lay.flatten(sd, dev)
    grads = torch.zeros_like(params)
    eng = FgnnEngine(lay, 2 * B, N, dev)
    x = torch.cat([x1, x2]).contiguous().to(dev)
    scores, loss = eng.forward(params, x)
    torch.cuda.synchronize()
    keep = {}
    O.node_embedding(torch.cat([x1, x2]), sd, keep)
    for k in range(1, nblk + 1):
        for j in (1, 2, 3):
            y = eng.normalized(k, j, params).cpu()
            print('%s blk%d mlp%d  rel=%.3e' % (name, k, j, rel(y, keep['ne/bm/block%d/mlp%d' % (k, j)])))
        m = eng.unpadded(eng.mult[k]).cpu()
        print('%s blk%d mult  rel=%.3e' % (name, k, rel(m, keep['ne/bm/block%d/mult' % k])))
    print('%s suffix rel=%.3e' % (name, rel(eng.E.cpu(), keep['ne/suffix'])))
    print('%s scores rel=%.3e (golden fp32) %.3e (fp64)' % (name, rel(scores.cpu(), torch.from_numpy(d['scores'])), rel(scores.cpu(), torch.from_numpy(d['scores64']))))
    print('%s loss %.8f golden %.8f' % (name, loss.item(), float(d['loss'])))
    eng.backward(params, grads)
    torch.cuda.synchronize()
    g = lay.unflatten(grads.cpu())
    worst = 0
    for kname, off, shape in lay.entries:
        ref = torch.from_numpy(d['grad/' + kname]); ref64 = torch.from_numpy(d['grad64/' + kname])
        e = rel(g[kname], ref); e64 = rel(g[kname], ref64); eref = rel(ref, ref64)
        flag = '' if e < 1e-4 or ref64.abs().max() < 1e-7 else '  <<<<'
        print('  grad %-40s rel32=%.2e rel64=%.2e ref32v64=%.2e |ref|=%.2e%s' % (kname, e, e64, eref, ref64.abs().max().item(), flag))

if __name__ == '__main__':
    torch.manual_seed(0)
    run_case('cfg1_er_n20_b4_1blk.npz', 1)
    run_case('cfg2_reg_n50_b2_4blk.npz', 4)
    # timing at cfg2 full batch
    from graph_neural_net_amd import synthetic
    lay = ParamLayout(2, 4, 32, 32, 3)
    dev = torch.device('cuda:0')
    sd = O.init_state_dict()
    params = lay.flatten(sd, dev); grads = torch.zeros_like(params)
    x1, x2 = synthetic.make_batch(2001, 32, 50)
    x = torch.cat([x1, x2]).contiguous().to(dev)
    eng = FgnnEngine(lay, 64, 50, dev)
    for _ in range(3): eng.step(params, grads, x)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(20): eng.step(params, grads, x)
    torch.cuda.synchronize(); dt = (time.time() - t) / 20
    print('cfg2 B=32 step %.3f ms -> %.0f pairs/s' % (dt * 1e3, 32 / dt))
    for _ in range(3): eng.forward(params, x)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(20): eng.forward(params, x)
    torch.cuda.synchronize(); dt = (time.time() - t) / 20
    print('cfg2 B=32 fwd-only %.3f ms' % (dt * 1e3))
