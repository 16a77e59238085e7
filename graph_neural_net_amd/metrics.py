"""accuracy_max (toolbox/metrics.py:119-141) on the device: argmax over each score row compared with
the identity matching.  The Hungarian metric of the reference (SciPy on the host) stays out of scope."""
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
