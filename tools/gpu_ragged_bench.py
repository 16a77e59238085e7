#!/usr/bin/env python3
"""BASELINE config 5 shape (variable-N pairs, n in [30, 120]) on one GPU: ms/step of
  (a) the module path, one MaskedTensor batch padded to the largest graph (what a user of the reference writes),
  (b) FgnnTrainer.prepare_ragged + model_step_prepared (one engine pass per size bucket; staging not timed).
usage: python tools/gpu_ragged_bench.py [pairs=8] [n_lo=30] [n_hi=120] [steps=30]"""
import sys
import time

import torch

from graph_neural_net_amd import synthetic
from graph_neural_net_amd.engine import ParamLayout
from graph_neural_net_amd.masked import from_list
from graph_neural_net_amd.siamese import Siamese_Node_Exp
from graph_neural_net_amd.trainer import FgnnTrainer


def timed(fn, steps):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    lo = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    hi = int(sys.argv[3]) if len(sys.argv) > 3 else 120
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 30
    dev = torch.device('cuda', 0)
    xs, ys = synthetic.make_ragged_batch(5000, B, lo, hi, 'ErdosRenyi', 0.2, 0.1)
    xs, ys = [x.to(dev) for x in xs], [y.to(dev) for y in ys]
    sizes = [int(x.shape[-1]) for x in xs]
    work = sum(n * n for n in sizes) / (len(sizes) * max(sizes) ** 2)
    print('pairs %d  sizes %s  sum n^2 / (B nmax^2) = %.2f' % (B, sizes, work))
    ne = dict(type='node_embedding', block_init='block_emb', block_inside='block', num_blocks=4, in_features=32,
              out_features=32, depth_of_mlp=3, constant_n_vertices=False)
    model = Siamese_Node_Exp(2, ne, metric='max').to(dev)
    m1 = from_list(xs, dims=(1, 2), base_name='N')
    m2 = from_list(ys, dims=(1, 2), base_name='M')

    def module_step():
        for p in model.parameters():
            p.grad = None
        model.loss(model(m1, m2)).backward()

    ms = timed(module_step, steps)
    print('module path (MaskedTensor batch):    %.3f ms/step  %.0f pairs/s' % (ms, B / ms * 1e3))
    lay = ParamLayout(2, 4, 32, 32, 3)
    tr = FgnnTrainer(lay, lay.init_flat(0, dev))
    for gran in (16, 32, max(sizes)):
        batch = tr.prepare_ragged(xs, ys, granule=gran)       # loader work, once per batch
        shape = ' '.join('%dx%d' % (b['pairs'], b['npad']) for b in batch['buckets'])
        ms = timed(lambda: tr.model_step_prepared(batch, want_scores=False), steps)
        print('trainer, buckets of %3d [%s]: %.3f ms/step  %.0f pairs/s' % (gran, shape, ms, B / ms * 1e3))


if __name__ == '__main__':
    main()
