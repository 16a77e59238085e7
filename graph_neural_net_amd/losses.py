"""``triplet_loss`` (toolbox/losses.py:8-34) on the GPU: per graph, cross-entropy (sum) of the
n x n score rows against the identity matching, divided by the total number of nodes
('mean') or averaged per graph ('mean_of_mean')."""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from .masked import MaskedTensor


class _CeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scores, nvalid, weights):
        if not scores.is_cuda:
            raise RuntimeError('triplet_loss: scores are on %s; no CPU path' % (scores.device,))
        scores = scores.contiguous()
        B, N, _ = scores.shape
        f32 = dict(dtype=torch.float32, device=scores.device)
        lse = torch.empty(B, N, **f32)
        pl = torch.empty(B, **f32)
        _lib.call('fgnn_ce_fwd', _lib.ptr(scores), _lib.ptr(nvalid) if nvalid is not None else None, B, N,
                  _lib.ptr(lse), _lib.ptr(pl), _lib.stream_ptr())
        ctx.save_for_backward(scores, lse, nvalid, weights)
        return (pl * weights).sum()

    @staticmethod
    def backward(ctx, g):
        scores, lse, nvalid, weights = ctx.saved_tensors
        B, N, _ = scores.shape
        d = torch.empty_like(scores)
        one = torch.ones(1, dtype=torch.float32, device=scores.device)
        _lib.call('fgnn_ce_bwd', _lib.ptr(scores), _lib.ptr(lse), _lib.ptr(nvalid) if nvalid is not None else None,
                  _lib.ptr(one), B, N, _lib.ptr(d), _lib.stream_ptr())
        return d * (g * weights).view(B, 1, 1), None, None


class triplet_loss(nn.Module):
    def __init__(self, loss_reduction='mean'):
        super().__init__()
        if loss_reduction not in ('mean', 'mean_of_mean'):
            raise ValueError('Unknown loss_reduction parameters {}'.format(loss_reduction))
        self.loss_reduction = loss_reduction

    def forward(self, raw_scores):
        """raw_scores: (bs, n, n) tensor or MaskedTensor."""
        if isinstance(raw_scores, MaskedTensor):
            s, nvalid = raw_scores.tensor, raw_scores.nvalid
            n = nvalid.to(torch.float32)
        else:
            s, nvalid = raw_scores, None
            n = torch.full((s.shape[0],), float(s.shape[1]), dtype=torch.float32, device=s.device)
        if self.loss_reduction == 'mean':
            w = torch.ones_like(n) / n.sum()
        else:
            w = 1.0 / (n * n.numel())
        return _CeFn.apply(s, nvalid, w)
