#!/usr/bin/env python3
"""ms/step of the module path (Siamese_Node_Exp.forward -> loss -> backward) for channel widths outside the fused engine
(generic conv.hip kernels), next to the fused 32-wide configuration.  usage: python tools/gpu_width_bench.py [B=32] [N=50] [only this width]"""
import sys
import time

import torch

from graph_neural_net_amd import synthetic
from graph_neural_net_amd.siamese import Siamese_Node_Exp


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    dev = torch.device('cuda', 0)
    x1, x2 = synthetic.make_batch(2000, B, N, 'Regular', 0.2, 0.1)
    x1, x2 = x1.to(dev), x2.to(dev)
    only = int(sys.argv[3]) if len(sys.argv) > 3 else None          # e.g. 64: just the in = out = 64 model (profiling)
    for c0, cin, cout, depth in ((2, 32, 32, 3), (2, 16, 16, 3), (2, 64, 64, 3), (2, 48, 24, 2), (3, 32, 32, 3)):
        if only is not None and (cin != only or c0 != 2):
            continue
        ne = dict(type='node_embedding', block_init='block_emb', block_inside='block', num_blocks=4, in_features=cin,
                  out_features=cout, depth_of_mlp=depth)
        model = Siamese_Node_Exp(c0, ne, metric='max').to(dev)
        a, b = (x1, x2) if c0 == 2 else (torch.cat([x1, x1[:, :1]], 1), torch.cat([x2, x2[:, :1]], 1))

        def step():
            for p in model.parameters():
                p.grad = None
            model.loss(model(a, b)).backward()

        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        fused = model.node_embedder._standard_layout() is not None and model.node_embedder._pad is None
        padded = model.node_embedder._standard_layout() is not None and not fused
        cap = ''
        if not padded:                              # one replayed HIP graph: the engine's step, or the captured module path
            for _ in range(3):
                model.fused_step(a, b)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                model.fused_step(a, b)
            torch.cuda.synchronize()
            ms_c = (time.perf_counter() - t0) / 20 * 1e3
            cap = ' | fused_step (captured) %7.3f ms  %6.0f pairs/s' % (ms_c, B / ms_c * 1e3)
        print('c0 %d  in %2d  out %2d  depth %d: eager %7.3f ms/step  %6.0f pairs/s%s  (%s)'
              % (c0, cin, cout, depth, ms, B / ms * 1e3, cap,
                 'fused engine' if fused else ('fused engine, zero-padded' if padded else 'per-layer modules, conv.hip')))


if __name__ == '__main__':
    main()
