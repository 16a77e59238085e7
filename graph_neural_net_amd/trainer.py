"""Minimal data-parallel training step around the fused engine (the Lightning shell of the reference,
models/trainers.py:70-104, is out of scope): forward + loss + backward (HIP), ONE all-reduce of the flat
gradient buffer (RCCL), fused Adam."""
import torch
import torch.nn.functional as F

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

    # ------------------------------------------------------------------ ragged batches, bucketed by size
    @staticmethod
    def bucket_by_size(sizes, granule=16):
        """Group graph indices by padded size ceil(n / granule) * granule (a batch
        padded to its global Nmax wastes up to (Nmax / n)^2 of the work on the small graphs).
        -> list of (padded_n, [indices]) in increasing size."""
        buckets = {}
        for i, n in enumerate(sizes):
            buckets.setdefault(-(-int(n) // granule) * granule, []).append(i)
        return sorted(buckets.items())

    def model_step_ragged(self, xs, ys, granule=16):
        """Forward + loss + backward of a ragged list of pairs (xs[i], ys[i]: (c0, n_i, n_i) device tensors),
        one fused-engine pass per size bucket; gradients and loss of the buckets are summed, both normalised
        by the node count of the WHOLE (global) batch, so the result equals the single padded batch.
        Returns (loss, [scores_i of shape (n_i, n_i)]); self.grads holds the local gradient sum."""
        sizes = [int(x.shape[-1]) for x in xs]
        total = dp.global_node_count(sum(sizes), self.params.device)
        loss = torch.zeros((), dtype=torch.float32, device=self.params.device)
        scores = [None] * len(xs)
        acc = torch.zeros_like(self.grads)
        for npad, idx in self.bucket_by_size(sizes, granule):
            nmax = max(sizes[i] for i in idx)          # pad to the bucket's own maximum, not to the granule
            pad = lambda t: F.pad(t, (0, nmax - t.shape[-1], 0, nmax - t.shape[-1]))
            x = torch.stack([pad(xs[i]) for i in idx] + [pad(ys[i]) for i in idx]).contiguous()
            nv = torch.tensor([sizes[i] for i in idx] * 2, dtype=torch.int32, device=x.device)
            eng = self._engine(2 * len(idx), nmax, True)
            sc, l = eng.step(self.params, self.grads, x, nvalid=nv, total_nodes=total)
            acc += self.grads
            loss = loss + l
            for k, i in enumerate(idx):
                scores[i] = sc[k, :sizes[i], :sizes[i]].clone()
        self.grads.copy_(acc)
        return loss, scores

    def train_step_ragged(self, xs, ys, granule=16):
        loss, scores = self.model_step_ragged(xs, ys, granule)
        dp.allreduce_sum_(self.grads)
        dp.allreduce_sum_(loss)
        self.opt.step(self.grads)
        return loss, scores

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
