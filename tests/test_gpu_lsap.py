"""GPU: fgnn_lsap_accuracy (the device-side assignment of accuracy_linear_assignment, toolbox/metrics.py:92-116) returns the
matching scipy.optimize.linear_sum_assignment returns -- row for row, ties included -- and the metric built on it equals the
reference's count."""
import numpy as np
import pytest
import torch
from scipy.optimize import linear_sum_assignment

from graph_neural_net_amd import _lib
from graph_neural_net_amd.masked import MaskedTensor
from graph_neural_net_amd.metrics import accuracy_linear_assignment
from oracle import fgnn_lsap_oracle as lo

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _solve(cost, nv=None):
    B, N, _ = cost.shape
    cd = torch.as_tensor(cost, dtype=torch.float32).to(DEV).contiguous()
    nvd = torch.as_tensor(nv, dtype=torch.int32).to(DEV) if nv is not None else None
    correct = torch.full((B,), -7, dtype=torch.int32, device=DEV)
    assign = torch.full((B, N), -7, dtype=torch.int32, device=DEV)
    _lib.call('fgnn_lsap_accuracy', _lib.ptr(cd), N * N, N, _lib.ptr(nvd), B, N, _lib.ptr(correct), _lib.ptr(assign),
              _lib.stream_ptr())
    return correct.cpu().numpy(), assign.cpu().numpy()


@pytest.mark.parametrize('N', [1, 2, 7, 50, 64, 65, 120, 200, 300, 700, 1100])
def test_matching_equals_scipy_random_costs(N):
    rng = np.random.default_rng(N)
    B = 5 if N <= 300 else 2
    cost = rng.standard_normal((B, N, N)).astype(np.float32)
    nv = np.array([N, max(1, N // 2), max(1, N - 1), 1, N])[:B]
    correct, assign = _solve(cost, nv)
    for b in range(B):
        n = int(nv[b])
        _, cols = linear_sum_assignment(cost[b, :n, :n])
        assert np.array_equal(assign[b, :n], cols)
        assert np.all(assign[b, n:] == -1)
        assert correct[b] == int(np.sum(cols == np.arange(n)))


@pytest.mark.parametrize('hi', [1, 2, 3, 5])
def test_matching_equals_scipy_on_ties(hi):
    """small-integer and constant costs: the optimum is far from unique, the tie rules decide"""
    rng = np.random.default_rng(100 + hi)
    B, N = 8, 37
    cost = rng.integers(0, hi, (B, N, N)).astype(np.float32)
    cost[1] = cost[1] + cost[1].T
    correct, assign = _solve(cost)
    for b in range(B):
        _, cols = linear_sum_assignment(cost[b])
        assert np.array_equal(assign[b], cols), b
        assert np.array_equal(assign[b], lo.linear_sum_assignment_rows(cost[b]))
    if hi == 1:
        assert np.all(correct == N)                         # constant matrix -> identity (SciPy issue 11602)


def test_infeasible_matrix_counts_zero():
    cost = np.zeros((2, 4, 4), dtype=np.float32)
    cost[0, 1, :] = np.inf
    correct, assign = _solve(cost)
    assert correct[0] == 0 and np.all(assign[0] == -1)
    assert correct[1] == 4


def test_metric_equals_the_reference_formula():
    """accuracy_linear_assignment on raw scores (dense and MaskedTensor) against the reference's loop: log_softmax, SciPy per
    graph, preds == arange (toolbox/metrics.py:92-116)."""
    g = torch.Generator().manual_seed(5)
    B, N = 6, 50
    s = torch.randn(B, N, N, generator=g) * 3
    s[2] = torch.round(s[2])
    sd = s.to(DEV)
    acc, total = accuracy_linear_assignment(sd)
    per = accuracy_linear_assignment(sd, aggregate_score=False)
    w = torch.log_softmax(sd, -1)                       # the reference takes the device log-softmax to the host
    want = []
    for b in range(B):
        _, cols = linear_sum_assignment(-w[b].cpu().numpy())
        want.append(int(np.sum(cols == np.arange(N))))
    assert (acc, total) == (sum(want), B * N)
    assert per == [h / N for h in want]
    # ragged: the valid corner only, padding columns outside the row softmax
    nv = torch.tensor([50, 20, 33, 1, 49, 50], dtype=torch.int32)
    sm = s.clone()
    for b in range(B):
        sm[b, nv[b]:, :] = 0
        sm[b, :, nv[b]:] = 0
    mt = MaskedTensor(sm.to(DEV), nv.to(DEV), (1, 2), 'N')
    acc, total = accuracy_linear_assignment(mt)
    want = 0
    for b in range(B):
        n = int(nv[b])
        wb = torch.log_softmax(sm[b, :n, :n].to(DEV), -1)
        _, cols = linear_sum_assignment(-wb.cpu().numpy())
        want += int(np.sum(cols == np.arange(n)))
    assert (acc, total) == (want, int(nv.sum()))


def test_metric_small_cases():
    """identity-dominant scores are fully matched, a permuted optimum is counted against the identity, and the ragged form
    ignores the padding (toolbox/metrics.py:92-116)."""
    g = torch.Generator().manual_seed(0)
    s = torch.randn(3, 6, 6, generator=g)
    ref = 0
    for b in range(3):
        _, p = linear_sum_assignment(-torch.log_softmax(s[b], -1).numpy())
        ref += int((p == np.arange(6)).sum())
    assert accuracy_linear_assignment(s.to(DEV)) == (ref, 18)
    assert accuracy_linear_assignment((torch.eye(5)[None] * 10.0).to(DEV)) == (5, 5)
    assert accuracy_linear_assignment((torch.eye(4)[None].flip(-1) * 10.0).to(DEV)) == (0, 4)
    pad = torch.zeros(2, 5, 5)
    pad[0, :3, :3] = torch.eye(3) * 8
    pad[1] = torch.eye(5) * 8
    mt = MaskedTensor(pad.to(DEV), torch.tensor([3, 5]).to(DEV), (1, 2))
    assert accuracy_linear_assignment(mt) == (8, 8)
    assert accuracy_linear_assignment(mt, aggregate_score=False) == [1.0, 1.0]
