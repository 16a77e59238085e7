#!/usr/bin/env python3
"""ms/step of the 64-feature model on a RAGGED batch (the cfg5 batch: 8 pairs, n in [30, 120], one padded MaskedTensor per side) through the
module path (forward, loss, backward; eager launches): what the padding-tile skipping of csrc/mlp64.hip is worth.
usage: python tools/gpu_width_ragged.py"""
import time

import torch

from graph_neural_net_amd import synthetic
from graph_neural_net_amd.masked import from_list
from graph_neural_net_amd.siamese import Siamese_Node_Exp


def main():
    dev = torch.device('cuda', 0)
    xs, ys = synthetic.make_ragged_batch(5000, 8, 30, 120, 'ErdosRenyi', 0.2, 0.1)
    m1 = from_list([t.to(dev) for t in xs], dims=(1, 2), base_name='N')
    m2 = from_list([t.to(dev) for t in ys], dims=(1, 2), base_name='M')
    ne = dict(type='node_embedding', block_init='block_emb', block_inside='block', num_blocks=4, in_features=64, out_features=64, depth_of_mlp=3,
              constant_n_vertices=False)
    model = Siamese_Node_Exp(2, ne, metric='max').to(dev)

    def step():
        for p in model.parameters():
            p.grad = None
        model.loss(model(m1, m2)).backward()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print('64-feature model, 8 ragged pairs (n in [30, 120], padded to %d): %.3f ms/step  %.0f pairs/s' % (max(t.shape[-1] for t in xs), ms, 8 / ms * 1e3))


if __name__ == '__main__':
    main()
