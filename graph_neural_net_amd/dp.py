"""Data-parallel glue: one process per GPU, one RCCL all-reduce per step.

The reference has no distributed code; graph pairs are independent,
so the batch is sharded over ranks and the only exchange is the sum of the flat
gradient buffer (40 000 fp32 = 160 kB for the default model) -- latency-bound on
xGMI, so a single un-bucketed call is the right shape.

``triplet_loss('mean')`` divides by the number of nodes in the *whole* batch
(toolbox/losses.py:27-34).  To reproduce a single-process run on the concatenated
batch without a second collective, every rank back-propagates the UN-normalised sum of
its pair losses and appends two floats to the flat gradient buffer -- its loss sum and
its node count -- so that ONE all-reduce delivers the gradient sum,
the global loss sum and the global normaliser; the division happens afterwards on the
device (``FlatAdam.set_grad_scale_reciprocal``), including the ragged case where ranks
hold different node counts.  ``global_node_count`` (a collective of its own) remains for
callers that need the normaliser on the host.
"""
import os

import torch
import torch.distributed as dist


def env_rank():
    return int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))


def init_process_group(backend=None):
    """Initialise torch.distributed from the torchrun environment (no-op for world size 1)."""
    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'   # 'nccl' is RCCL on ROCm
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of n_items for `rank` (first n_items % world ranks get one more)."""
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def global_node_count(local_nodes, device=None):
    """Sum of per-rank node counts (the loss normaliser of the concatenated batch)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(local_nodes)
    t = torch.tensor([float(local_nodes)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def world_size():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def allreduce_sum_(flat):
    """In-place sum of a flat buffer over all ranks (single call, single bucket)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value, device=None):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
