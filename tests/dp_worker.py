#!/usr/bin/env python3
"""Worker of tests/test_00_gpu_two_ranks.py: one rank of a data-parallel FgnnTrainer run, or the single-process run on the
concatenated batch (WORLD_SIZE=1).  Backend and device follow the node (graph_neural_net_amd.dp.pick_backend): with at least
WORLD_SIZE GPUs every rank takes cuda:LOCAL_RANK over RCCL ('nccl') -- the form the 8-GPU run uses; on a one-GPU box all ranks
share cuda:0 over gloo (RCCL refuses two ranks on one device).  FGNN_TEST_BACKEND overrides.  Writes the parameters after each
mode's steps and the per-step losses to <out>.pt.   usage: dp_worker.py <out-prefix>"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graph_neural_net_amd import dp, synthetic                   # noqa: E402
from graph_neural_net_amd.engine import ParamLayout              # noqa: E402
from graph_neural_net_amd.trainer import FgnnTrainer             # noqa: E402

STEPS = 3

# ---- runtime count of the collectives each train step issues (every entry point of torch.distributed that moves data) ----
COLLECTIVES = ('all_reduce', 'all_gather', 'all_gather_into_tensor', 'all_gather_object', 'broadcast', 'broadcast_object_list', 'reduce',
               'reduce_scatter', 'reduce_scatter_tensor', 'all_to_all', 'all_to_all_single', 'gather', 'scatter', 'send', 'recv',
               'isend', 'irecv', 'barrier')
CALLS = []


def _count_collectives():
    import torch.distributed as dist
    for name in COLLECTIVES:
        fn = getattr(dist, name, None)
        if fn is None:
            continue

        def wrapped(*a, _fn=fn, _name=name, **k):
            CALLS.append(_name)
            return _fn(*a, **k)
        setattr(dist, name, wrapped)


def _counted(fn, *a, **k):
    """-> (result, names of the collectives issued inside fn)."""
    n0 = len(CALLS)
    r = fn(*a, **k)
    return r, CALLS[n0:]


def main():
    out = sys.argv[1]
    world = dp.env_rank()[2]
    backend, index = dp.pick_backend(world)               # device_count() only: the GPU is not initialised yet
    backend = os.environ.get('FGNN_TEST_BACKEND') or backend
    if backend == 'gloo':
        index = 0
    torch.cuda.set_device(index)
    rank, _, world = dp.init_process_group(backend)
    _count_collectives()
    dev = torch.device('cuda', index)
    lay = ParamLayout(2, 2, 32, 32, 3)
    p0 = lay.init_flat(11, dev)
    res = {}
    # ---- constant N: global batch of 8 pairs, N = 18; eager and captured ----
    batches = [synthetic.make_batch(7100 + s, 8, 18, 'ErdosRenyi', 0.3, 0.05) for s in range(STEPS)]
    for mode in ('eager', 'capture'):
        tr = FgnnTrainer(lay, p0.clone(), lr=2e-3, capture=(mode == 'capture'))
        losses = []
        res[mode + '_collectives'] = []
        for s in range(STEPS):
            x1, x2 = batches[s]
            lo, hi = dp.shard_range(8, rank, world)
            (loss, _), calls = _counted(tr.train_step, x1[lo:hi].to(dev), x2[lo:hi].to(dev))
            res[mode + '_collectives'].append(calls)
            losses.append(loss.item())
            if s == 0:
                res[mode + '_comm'] = tr.comm.cpu().clone()      # [summed gradients | loss sum | node count] after the all-reduce
        res[mode] = (tr.params.cpu().clone(), losses, tr.opt.t)
        res[mode + '_in_graph'] = tr.allreduce_in_graph
    res['backend'] = dp.backend()
    res['device_index'] = index
    # ---- ragged: 6 pairs with different sizes, ranks hold different node counts ----
    tr = FgnnTrainer(lay, p0.clone(), lr=2e-3)
    losses = []
    res['ragged_collectives'] = []
    for s in range(STEPS):
        xs, ys = synthetic.make_ragged_batch(7200 + s, 6, 9, 40)
        lo, hi = dp.shard_range(6, rank, world)
        (loss, _), calls = _counted(tr.train_step_ragged, [x.to(dev) for x in xs[lo:hi]], [y.to(dev) for y in ys[lo:hi]], granule=16)
        res['ragged_collectives'].append(calls)
        losses.append(loss.item())
        if s == 0:
            res['ragged_comm'] = tr.comm.cpu().clone()
    res['ragged'] = (tr.params.cpu().clone(), losses, tr.opt.t)
    res['p0'] = p0.cpu().clone()
    # ---- padded ragged batch through train_step(nvalid=...) ----
    tr = FgnnTrainer(lay, p0.clone(), lr=2e-3)
    losses = []
    from oracle import fgnn_oracle as O                          # (only pad_graph_list: test-side batch assembly)
    for s in range(STEPS):
        xs, ys = synthetic.make_ragged_batch(7300 + s, 4, 10, 30)
        lo, hi = dp.shard_range(4, rank, world)
        nmax = max(x.shape[-1] for x in xs)
        pad = lambda t: torch.nn.functional.pad(t, (0, nmax - t.shape[-1], 0, nmax - t.shape[-1]))
        x1 = torch.stack([pad(x) for x in xs[lo:hi]]).to(dev)
        x2 = torch.stack([pad(y) for y in ys[lo:hi]]).to(dev)
        nv = torch.tensor([x.shape[-1] for x in xs[lo:hi]], dtype=torch.int32, device=dev)
        (loss, _), calls = _counted(tr.train_step, x1, x2, nvalid=nv)
        res.setdefault('padded_collectives', []).append(calls)
        losses.append(loss.item())
        if s == 0:
            res['padded_comm'] = tr.comm.cpu().clone()
    res['padded'] = (tr.params.cpu().clone(), losses, tr.opt.t)
    torch.cuda.synchronize()
    if rank == 0:
        torch.save(res, out + '.pt')
    if world > 1:
        dp.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
