"""CPU, world_size 2 over gloo: the data-parallel reduction reproduces the single-process
gradients of the concatenated batch, incl. the ragged loss normaliser."""
import os
import sys

import torch
import torch.multiprocessing as mp

from util import ROOT


def _worker(rank, world, port, ragged, ret):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    torch.set_num_threads(2)
    from graph_neural_net_amd import dp, synthetic
    from graph_neural_net_amd.engine import ParamLayout
    from oracle import fgnn_oracle as O
    dp.init_process_group('gloo')
    torch.manual_seed(0)
    sd = O.init_state_dict(num_blocks=2)
    lay = ParamLayout(2, 2, 32, 32, 3)
    if ragged:
        xs, ys = synthetic.make_ragged_batch(11, 6, 5, 12)
    else:
        a, b = synthetic.make_batch(11, 6, 10, 'ErdosRenyi', 0.3, 0.1)
        xs, ys = list(a), list(b)
    lo, hi = dp.shard_range(len(xs), rank, world)
    # local loss is normalised by the GLOBAL node count, gradients are SUMMED
    local_nodes = sum(x.shape[-1] for x in xs[lo:hi])
    total = dp.global_node_count(local_nodes)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    scores = O.siamese_scores_ragged(xs[lo:hi], ys[lo:hi], params)
    loss = O.triplet_loss_mean(scores) * (local_nodes / total)
    gs = torch.autograd.grad(loss, list(params.values()))
    flat = lay.flatten(dict(zip(params.keys(), gs)), 'cpu')
    dp.allreduce_sum_(flat)
    if rank == 0:
        _, _, full = O.step_fwd_bwd_ragged(xs, ys, sd)
        ref = lay.flatten(full, 'cpu')
        ret['err'] = ((flat - ref).abs().max() / ref.abs().max()).item()
        ret['total'] = total
        ret['expect_total'] = float(sum(x.shape[-1] for x in xs))
    dp.barrier()
    torch.distributed.destroy_process_group()


def _run(ragged):
    ctx = mp.get_context('spawn')
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        port = 29600 + (os.getpid() % 200) + (50 if ragged else 0)
        procs = [ctx.Process(target=_worker, args=(r, 2, port, ragged, ret)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
            assert p.exitcode == 0
        return dict(ret)


def test_dp_constant_n_matches_single_process():
    r = _run(False)
    assert r['total'] == r['expect_total']
    assert r['err'] < 1e-5, r


def test_dp_ragged_loss_normaliser():
    r = _run(True)
    assert r['total'] == r['expect_total']
    assert r['err'] < 1e-5, r
