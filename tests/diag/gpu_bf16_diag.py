#!/usr/bin/env python3
"""Stage-by-stage comparison of the bf16 HIP engine with oracle/fgnn_oracle_bf16.py (diagnostic; the oracle is only the
checker here).  usage: python tests/diag/gpu_bf16_diag.py [N] [B] [blocks]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from graph_neural_net_amd import synthetic                      # noqa: E402
from graph_neural_net_amd.engine import ParamLayout             # noqa: E402
from graph_neural_net_amd.engine16 import FgnnEngineBF16        # noqa: E402
from oracle import fgnn_oracle as O, fgnn_oracle_bf16 as OB     # noqa: E402


def err(a, b):
    return O.max_rel_err(a.float().cpu(), b)


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    sd = O.init_state_dict(num_blocks=K)
    g = torch.Generator().manual_seed(1)
    for k, v in sd.items():
        if k.endswith('.bias') and v.dim() == 1:
            v.add_(0.1 * torch.randn(v.shape, generator=g))
        elif k.endswith('gn.weight'):
            v.mul_(1 + 0.2 * torch.randn(v.shape, generator=g))
        elif k.endswith('gn.bias'):
            v.add_(0.05 * torch.randn(v.shape, generator=g))
    x1, x2 = synthetic.make_batch(3, B, N, 'ErdosRenyi', 0.3 if N < 100 else 0.5, 0.1)
    keep = {}
    t0 = time.time()
    s_ref, l_ref, g_ref = OB.step_fwd_bwd(x1, x2, sd, keep=keep)
    print('oracle %.1fs' % (time.time() - t0))
    s32, l32, g32 = O.step_fwd_bwd(x1, x2, sd)
    lay = ParamLayout(2, K, 32, 32, 3)
    params = lay.flatten(sd, dev)
    grads = torch.zeros_like(params)
    eng = FgnnEngineBF16(lay, 2 * B, N, dev)
    x = torch.cat([x1, x2]).contiguous().to(dev)
    scores, loss = eng.forward(params, x)
    torch.cuda.synchronize()
    for k in range(1, K + 1):
        for j, nm in ((1, 'z1'), (2, 'z2'), (3, 'z3')):
            print('blk %d %s  %.3e' % (k, nm, err(eng.dense(eng.z[(k, j)]), keep[(k, nm)])), end='   ')
            if j == 2:
                print('mult %.3e' % err(eng.dense(eng.mult[k]), keep[(k, 'mult')]), end='   ')
        print()
    print('E %.3e  idx mismatches %d / %d' % (err(eng.E, keep['E']), int((eng.idx.cpu().long() != keep['idx']).sum()), keep['idx'].numel()))
    print('scores: hip vs bf16-oracle %.3e | bf16-oracle vs fp32-oracle %.3e | hip vs fp32-oracle %.3e'
          % (err(scores, s_ref), O.max_rel_err(s_ref, s32), err(scores, s32)))
    print('loss hip %.6f oracle16 %.6f oracle32 %.6f' % (loss.item(), l_ref.item(), l32.item()))
    eng.backward(params, grads)
    torch.cuda.synchronize()
    W = eng._bwd
    print('dE %.3e' % err(W['dE'], keep['dE']))
    got = lay.unflatten(grads.cpu())
    names = [n for n, _, _ in lay.entries]
    f_h = torch.cat([got[n].reshape(-1) for n in names])
    f_o = OB.flat(g_ref, names)
    f_32 = OB.flat(g32, names)
    print('flat grad L2 rel: hip vs bf16-oracle %.3e | bf16-oracle vs fp32 %.3e | hip vs fp32 %.3e'
          % (((f_h - f_o).norm() / f_o.norm()).item(), ((f_o - f_32).norm() / f_32.norm()).item(),
             ((f_h - f_32).norm() / f_32.norm()).item()))
    worst = []
    for n in names:
        if n.endswith('convs.2.bias'):
            continue
        worst.append((O.max_rel_err(got[n], g_ref[n]), n))
    worst.sort(reverse=True)
    for e, n in worst[:8]:
        print('  %.3e %s' % (e, n))
    if len(sys.argv) > 4:
        for e, n in sorted(worst, key=lambda t: t[1]):
            print('  %.3e %s' % (e, n))


if __name__ == '__main__':
    main()
