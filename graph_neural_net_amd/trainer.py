"""Minimal data-parallel training step around the fused engine (the Lightning shell of the reference,
models/trainers.py:70-104, is out of scope): forward + loss + backward (HIP), ONE all-reduce of the flat
gradient buffer (RCCL), fused Adam."""
import torch

from . import dp
from .engine import FgnnEngine, ParamLayout
from .optim import FlatAdam


class FgnnTrainer:
    def __init__(self, layout, params_flat, lr=1e-3):
        self.layout = layout
        self.params = params_flat
        self.grads = torch.zeros_like(params_flat)
        self.opt = FlatAdam(params_flat, lr=lr)
        self._engines = {}

    def _engine(self, G, N, ragged):
        key = (G, N, ragged)
        if key not in self._engines:
            self._engines[key] = FgnnEngine(self.layout, G, N, self.params.device, ragged=ragged)
        return self._engines[key]

    def train_step(self, x1, x2, nvalid=None):
        """x1, x2: (B, c0, N, N) local shard on the GPU.  Returns (loss of the global batch as a device
        scalar, scores of the local shard)."""
        B, _, N, _ = x1.shape
        eng = self._engine(2 * B, N, nvalid is not None)
        local_nodes = B * N if nvalid is None else int(nvalid.sum().item())
        total = dp.global_node_count(local_nodes, self.params.device)
        x = torch.cat([x1, x2]).contiguous()
        nv = None if nvalid is None else torch.cat([nvalid, nvalid])
        scores, loss = eng.step(self.params, self.grads, x, nvalid=nv, total_nodes=total)
        loss = loss.clone()
        dp.allreduce_sum_(self.grads)          # gradients of the concatenated global batch
        dp.allreduce_sum_(loss)
        self.opt.step(self.grads)
        return loss, scores
