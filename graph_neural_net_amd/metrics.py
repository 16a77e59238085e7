"""The two matching accuracies of the reference on the device: accuracy_max (toolbox/metrics.py:119-141: argmax over each
score row compared with the identity matching) and accuracy_linear_assignment (toolbox/metrics.py:92-116: the minimum-cost
matching of -log_softmax(scores), SciPy's assignment reproduced by csrc/lsap.hip).  Neither copies the scores to the host;
the only synchronisation is the final count handed back as Python numbers, as the reference's return type demands.
Scores that live on the host (saved scores, evaluation scripts), and graphs beyond the device solver's FGNN_LSAP_MAX_N, take
the reference's own route: a host loop over the graphs with scipy.optimize.linear_sum_assignment (toolbox/metrics.py:104-112)."""
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
    if not s.is_cuda:           # host scores: the reference's arg-max comparison (toolbox/metrics.py:127-137), per graph
        n_ok = [int((s[b, :int(n), :int(n)].argmax(-1) == torch.arange(int(n))).sum()) for b, n in enumerate(sizes.tolist())]
        if aggregate_score:
            return sum(n_ok), int(sizes.sum().item())
        return [c / int(n) for c, n in zip(n_ok, sizes.tolist())]
    s = s.contiguous()
    B, N, _ = s.shape
    correct = torch.empty(B, dtype=torch.int32, device=s.device)
    _lib.call('fgnn_accuracy_max', _lib.ptr(s), _lib.ptr(nvalid) if nvalid is not None else None, B, N,
              _lib.ptr(correct), _lib.stream_ptr())
    if aggregate_score:
        return int(correct.sum().item()), int(sizes.sum().item())
    return (correct.to(torch.float64) / sizes.to(torch.float64)).tolist()


def accuracy_linear_assignment(rawscores, aggregate_score=True):
    """rawscores: (bs, n, n) device tensor or MaskedTensor.  Minimum-cost matching on -log_softmax(scores) per graph (the
    assignment scipy.optimize.linear_sum_assignment returns, computed on the device by fgnn_lsap_accuracy: no copy of the
    scores to the host, no host loop), counted against the identity matching: (n_correct, n_vertices) or the list of
    per-graph accuracies."""
    if isinstance(rawscores, MaskedTensor):
        s, nvalid = rawscores.tensor, rawscores.nvalid
        sizes = nvalid.to(torch.int64)
    else:
        s, nvalid = rawscores, None
        sizes = torch.full((s.shape[0],), s.shape[1], dtype=torch.int64, device=s.device)
    s = s.detach()
    if not s.is_cuda or s.shape[-1] > _lib.FGNN_LSAP_MAX_N:
        return _accuracy_lsap_host(s, sizes, aggregate_score)
    if nvalid is not None:          # padding columns must not take part in the row softmax
        col = torch.arange(s.shape[-1], device=s.device)[None, None, :] < nvalid.to(s.device)[:, None, None]
        s = s.masked_fill(~col, float('-inf'))
    cost = (-torch.log_softmax(s.float(), -1)).contiguous()
    B, N, _ = cost.shape
    correct = torch.empty(B, dtype=torch.int32, device=s.device)
    nv32 = nvalid.to(device=s.device, dtype=torch.int32).contiguous() if nvalid is not None else None
    _lib.call('fgnn_lsap_accuracy', _lib.ptr(cost), N * N, N, _lib.ptr(nv32), B, N, _lib.ptr(correct), None, _lib.stream_ptr())
    if aggregate_score:
        return int(correct.sum().item()), int(sizes.sum().item())
    return (correct.to(torch.float64) / sizes.to(torch.float64).to(correct.device)).tolist()


def _accuracy_lsap_host(s, sizes, aggregate_score):
    """The reference's host loop (toolbox/metrics.py:104-112): -log_softmax of each graph's valid n x n scores -> SciPy's
    assignment -> matches with the identity.  Used for scores that are not on the GPU and for n > FGNN_LSAP_MAX_N."""
    from scipy.optimize import linear_sum_assignment
    n_ok = []
    for b, n in enumerate(sizes.tolist()):
        n = int(n)
        cost = -torch.log_softmax(s[b, :n, :n].float(), -1).cpu().numpy()
        row, col = linear_sum_assignment(cost)
        n_ok.append(int((row == col).sum()))
    if aggregate_score:
        return sum(n_ok), int(sizes.sum().item())
    return [c / int(n) for c, n in zip(n_ok, sizes.tolist())]
