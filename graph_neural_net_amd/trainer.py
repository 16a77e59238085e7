"""Minimal data-parallel training step around the fused engine (the Lightning shell of the reference,
models/trainers.py:70-104, is out of scope): forward + loss + backward (HIP), ONE all-reduce of the flat
gradient buffer (RCCL), fused Adam."""
import torch
import torch.nn.functional as F

from . import dp
from .engine import FgnnEngine, ParamLayout
from .optim import FlatAdam


class FgnnTrainer:
    def __init__(self, layout, params_flat, lr=1e-3, capture=False):
        """capture=True: constant-shape steps are captured in a HIP graph (model work + fused Adam; the gradient
        all-reduce, when there is more than one rank, stays an eager RCCL call between two captured halves) and
        replayed -- the launch overhead of ~40 kernels per step disappears."""
        self.layout = layout
        self.params = params_flat
        self.grads = torch.zeros_like(params_flat)
        self.opt = FlatAdam(params_flat, lr=lr)
        self.capture = capture
        self._engines = {}
        self._graphs = {}

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

    # ------------------------------------------------------------------ captured constant-shape step
    def _captured_step(self, x1, x2):
        B, _, N, _ = x1.shape
        world = dp.env_rank()[2] if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 1
        key = (B, N)
        st = self._graphs.get(key)
        if st is None:
            eng = self._engine(2 * B, N, False)
            total = dp.global_node_count(B * N, self.params.device)
            xs = torch.cat([x1, x2]).contiguous().clone()
            self.opt.sync_hyper_parameters()
            # two eager steps on a side stream (allocations, kernel attributes), with the optimizer state restored after
            snap = [t.clone() for t in (self.params, self.opt.exp_avg, self.opt.exp_avg_sq)]
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    eng.step(self.params, self.grads, xs, total_nodes=total)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            for t, s0 in zip((self.params, self.opt.exp_avg, self.opt.exp_avg_sq), snap):
                t.copy_(s0)
            g_model, g_opt = torch.cuda.CUDAGraph(), None
            with torch.cuda.graph(g_model):
                scores, loss = eng.step(self.params, self.grads, xs, total_nodes=total)
                if world == 1:
                    self.opt.step_dev(self.grads)
            if world > 1:
                g_opt = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_opt):
                    self.opt.step_dev(self.grads)
            self.opt.t -= 1                              # the capture itself did not execute an update
            st = self._graphs[key] = (xs, g_model, g_opt, scores, loss, B)
        xs, g_model, g_opt, scores, loss, B = st
        self.opt.sync_hyper_parameters()
        xs[:B].copy_(x1)
        xs[B:].copy_(x2)
        g_model.replay()
        if g_opt is not None:
            dp.allreduce_sum_(self.grads)
            dp.allreduce_sum_(loss)
            g_opt.replay()
        self.opt.t += 1
        return loss, scores

    def train_step(self, x1, x2, nvalid=None):
        """x1, x2: (B, c0, N, N) local shard on the GPU.  Returns (loss of the global batch as a device
        scalar, scores of the local shard)."""
        if self.capture and nvalid is None:
            return self._captured_step(x1, x2)
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
