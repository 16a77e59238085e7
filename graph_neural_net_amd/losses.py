"""``triplet_loss`` (toolbox/losses.py:8-34) on the GPU: per graph, cross-entropy (sum) of the
n x n score rows against the identity matching, divided by the total number of nodes
('mean') or averaged per graph ('mean_of_mean')."""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from .masked import MaskedTensor


class _CeFn(torch.autograd.Function):
    """weights: per-pair weights (B,) for ragged batches, or a python float when every pair has the same weight (dense
    batches): then the weighted sum is one fgnn_sum_scale launch and the backward scale goes into the kernel's gscale
    (no elementwise pass over the B x N x N score gradient)."""

    @staticmethod
    def forward(ctx, scores, nvalid, weights):
        if not scores.is_cuda:
            raise RuntimeError('triplet_loss: scores are on %s; no CPU path' % (scores.device,))
        scores = scores.contiguous()
        B, N, _ = scores.shape
        f32 = dict(dtype=torch.float32, device=scores.device)
        lse = torch.empty(B, N, **f32)
        pl = torch.empty(B, **f32)
        st = _lib.stream_ptr()
        _lib.call('fgnn_ce_fwd', _lib.ptr(scores), _lib.ptr(nvalid) if nvalid is not None else None, B, N,
                  _lib.ptr(lse), _lib.ptr(pl), st)
        ctx.uniform = isinstance(weights, float)
        if ctx.uniform:
            ctx.w = weights
            ctx.save_for_backward(scores, lse, nvalid)
            out = torch.empty(1, **f32)
            _lib.call('fgnn_sum_scale', _lib.ptr(pl), B, 1, weights, _lib.ptr(out), st)
            return out.reshape(())
        ctx.save_for_backward(scores, lse, nvalid, weights)
        return (pl * weights).sum()

    @staticmethod
    def backward(ctx, g):
        scores, lse, nvalid = ctx.saved_tensors[:3]
        B, N, _ = scores.shape
        d = torch.empty_like(scores)
        if ctx.uniform:
            gs = (g.to(torch.float32) * ctx.w).reshape(1).contiguous()
        else:
            gs = torch.ones(1, dtype=torch.float32, device=scores.device)
        _lib.call('fgnn_ce_bwd', _lib.ptr(scores), _lib.ptr(lse), _lib.ptr(nvalid) if nvalid is not None else None,
                  _lib.ptr(gs), B, N, _lib.ptr(d), _lib.stream_ptr())
        if ctx.uniform:
            return d, None, None
        return d * (g * ctx.saved_tensors[3]).view(B, 1, 1), None, None


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
            # equal-size graphs: 'mean' and 'mean_of_mean' coincide (test_losses.py:27-35): every pair weighs 1 / (B n)
            return _CeFn.apply(raw_scores, None, 1.0 / float(raw_scores.shape[0] * raw_scores.shape[1]))
        if self.loss_reduction == 'mean':
            w = torch.ones_like(n) / n.sum()
        else:
            w = 1.0 / (n * n.numel())
        return _CeFn.apply(s, nvalid, w)
