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


def allreduce_sum_(flat, force=False):
    """In-place sum of a flat buffer over all ranks (single call, single bucket).
    force: issue the collective also with ONE rank (tests of the RCCL path on a one-GPU box)."""
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force):
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def backend():
    return dist.get_backend() if (dist.is_available() and dist.is_initialized()) else None


def collective_captures():
    """Can the gradient all-reduce be recorded into a HIP graph together with the kernels around it?  RCCL ('nccl') enqueues
    its kernels on the current stream and supports stream capture; gloo stages through the host and does not.
    FGNN_ALLREDUCE_IN_GRAPH=0 forces the eager call."""
    return backend() == 'nccl' and os.environ.get('FGNN_ALLREDUCE_IN_GRAPH', '1') != '0'


def pick_backend(world):
    """Backend and device index of this rank for a `world`-rank job on this node: one GPU per rank over RCCL when the node has
    at least `world` GPUs, else all ranks on cuda:0 over gloo (RCCL refuses two ranks on one device) -- the functional form the
    one-GPU test box can run.  torch.cuda.device_count() does not initialise the GPU."""
    _, local_rank, _ = env_rank()
    if world > 1 and torch.cuda.device_count() >= world:
        return 'nccl', local_rank
    return ('gloo' if world > 1 else None), 0


def warm_up_collective(device, force=False):
    """The first collective of a process group creates the communicator (RCCL: rings over xGMI, IPC handles) -- seconds, and
    not something to do inside a stream capture or a timed step.  Called once at set-up, outside any step."""
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force):
        t = torch.zeros(8, dtype=torch.float32, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        if t.is_cuda:
            torch.cuda.synchronize(device)


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value, device=None):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
