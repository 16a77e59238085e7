"""accuracy_max (toolbox/metrics.py:119-141) on the device: argmax over each score row compared with
the identity matching, plus the Hungarian metric of the reference (toolbox/metrics.py:92-116) as a host-side
evaluation metric: ONE device->host copy of the log-softmax scores per call, SciPy assignment per graph --
meant for validation, not for the per-step path."""
import numpy as np
import torch

from . import _lib
from .masked import MaskedTensor


def accuracy_max(weights, aggregate_score=True):
    """weights: (bs, n, n) device tensor or MaskedTensor.  Returns (n_correct, n_vertices), or the list
    of per-graph accuracies with aggregate_score=False."""
    if isinstance(weights, MaskedTensor):
        s, nvalid = weights.tensor, weights.nvalid
        sizes = nvalid.to(torch.int64)
    else:
        s, nvalid = weights, None
        sizes = torch.full((s.shape[0],), s.shape[1], dtype=torch.int64, device=s.device)
    if not s.is_cuda:
        raise RuntimeError('accuracy_max: scores are on %s; no CPU path' % (s.device,))
    s = s.contiguous()
    B, N, _ = s.shape
    correct = torch.empty(B, dtype=torch.int32, device=s.device)
    _lib.call('fgnn_accuracy_max', _lib.ptr(s), _lib.ptr(nvalid) if nvalid is not None else None, B, N,
              _lib.ptr(correct), _lib.stream_ptr())
    if aggregate_score:
        return int(correct.sum().item()), int(sizes.sum().item())
    return (correct.to(torch.float64) / sizes.to(torch.float64)).tolist()


def accuracy_linear_assignment(rawscores, aggregate_score=True):
    """rawscores: (bs, n, n) tensor or MaskedTensor.  Maximum-weight matching on log_softmax(scores) per graph,
    counted against the identity matching: (n_correct, n_vertices) or the list of per-graph accuracies."""
    from scipy.optimize import linear_sum_assignment
    if isinstance(rawscores, MaskedTensor):
        s, sizes = rawscores.tensor, rawscores.sizes()
    else:
        s, sizes = rawscores, [rawscores.shape[1]] * rawscores.shape[0]
    s = s.detach()
    if isinstance(rawscores, MaskedTensor):      # padding columns must not take part in the row softmax
        col = torch.arange(s.shape[-1], device=s.device)[None, None, :] < rawscores.nvalid.to(s.device)[:, None, None]
        s = s.masked_fill(~col, float('-inf'))
    cost = (-torch.log_softmax(s, -1)).cpu().numpy()
    acc, total, all_acc = 0, 0, []
    for b, n in enumerate(sizes):
        _, preds = linear_sum_assignment(cost[b, :n, :n])
        hit = int(np.sum(preds == np.arange(n)))
        acc += hit
        total += n
        all_acc.append(hit / n)
    return (acc, total) if aggregate_score else all_acc
