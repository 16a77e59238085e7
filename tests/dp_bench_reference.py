#!/usr/bin/env python3
"""Worker of tests/test_00_gpu_two_ranks.py::test_bench_eight_rank_dress_rehearsal: the SINGLE-PROCESS counterpart of
`bench.py --gpus W [--config cfg5]` -- the W shards bench.py's ranks build (same seeds) as one concatenated batch through one
engine, normalised by the global node count (toolbox/losses.py:27-34), plus, for the ragged configuration, the W shards one after
the other on their own padded geometry (what the ranks compute, summed on one device).
usage: dp_bench_reference.py <cfg2|cfg5> <world> <out.pt>"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graph_neural_net_amd import synthetic                       # noqa: E402
from graph_neural_net_amd.engine import FgnnEngine, ParamLayout  # noqa: E402


def _bits(x, dev):
    return torch.from_numpy(synthetic.pack_adjacency(x[:, 0].numpy()).view(np.int32)).to(dev)


def main():
    cfg, world, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    dev = torch.device('cuda', 0)
    lay = ParamLayout(2, 4, 32, 32, 3)
    params = lay.init_flat(0, dev)
    res = {}
    if cfg == 'cfg2':
        B, N = 32, 50
        shards = [synthetic.make_batch(2000 + r, B, N, 'Regular', 0.2, 0.1) for r in range(world)]
        x1 = torch.cat([s[0] for s in shards])
        x2 = torch.cat([s[1] for s in shards])
        eng = FgnnEngine(lay, 2 * B * world, N, dev, block1='structured')
        g = torch.zeros_like(params)
        _, loss = eng.step(params, g, None, total_nodes=float(B * N * world), bits=_bits(torch.cat([x1, x2]), dev))
        torch.cuda.synchronize()
        res.update(concat=g.cpu(), concat_loss=loss.item(), total_nodes=float(B * N * world))
    else:
        B = 8
        shards = [synthetic.make_ragged_batch(5000 + r, B, 30, 120, 'ErdosRenyi', 0.2, 0.1) for r in range(world)]
        sizes = [[int(t.shape[-1]) for t in xs] for xs, _ in shards]
        total = float(sum(sum(s) for s in sizes))

        def run(xs, ys, ns):
            N = max(ns)
            pad = lambda lst: torch.stack([torch.nn.functional.pad(t, (0, N - t.shape[-1], 0, N - t.shape[-1])) for t in lst])
            eng = FgnnEngine(lay, 2 * len(ns), N, dev, ragged=True, block1='structured')
            g = torch.zeros_like(params)
            nv = torch.tensor(ns * 2, dtype=torch.int32, device=dev)
            _, loss = eng.step(params, g, None, nvalid=nv, total_nodes=total, bits=_bits(torch.cat([pad(xs), pad(ys)]), dev))
            torch.cuda.synchronize()
            return g.cpu(), loss.item()
        gs, ls = zip(*[run(xs, ys, ns) for (xs, ys), ns in zip(shards, sizes)])
        res.update(shards=torch.stack(gs).double().sum(0).float(), shards_loss=float(sum(ls)))
        xs = [t for s in shards for t in s[0]]
        ys = [t for s in shards for t in s[1]]
        g, l = run(xs, ys, [n for s in sizes for n in s])
        res.update(concat=g, concat_loss=l, total_nodes=total, sizes=sizes)
    torch.save(res, out)


if __name__ == '__main__':
    main()
