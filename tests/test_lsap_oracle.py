"""The assignment oracle (oracle/fgnn_lsap_oracle.py) against scipy.optimize.linear_sum_assignment -- the solver the reference's
accuracy_linear_assignment calls (toolbox/metrics.py:92-116): the same matching, row for row, ties included."""
import numpy as np
import pytest
from scipy.optimize import linear_sum_assignment

from oracle import fgnn_lsap_oracle as lo


def _cases():
    rng = np.random.default_rng(11)
    out = []
    for n in (1, 2, 3, 7, 20, 50, 64, 65):
        out.append(('float32 n=%d' % n, rng.standard_normal((n, n)).astype(np.float32)))
    for n in (4, 9, 30, 50):
        for hi in (2, 3, 5):
            out.append(('ties n=%d hi=%d' % (n, hi), rng.integers(0, hi, (n, n)).astype(np.float32)))
    for n in (5, 17):
        out.append(('constant n=%d' % n, np.full((n, n), 0.25, dtype=np.float32)))
    a = rng.integers(0, 3, (12, 12)).astype(np.float32)
    out.append(('symmetric ties', a + a.T))
    out.append(('diag dominant', (-5 * np.eye(15) + rng.integers(0, 2, (15, 15))).astype(np.float32)))
    return out


@pytest.mark.parametrize('name,cost', _cases(), ids=[c[0] for c in _cases()])
def test_oracle_matching_equals_scipy(name, cost):
    rows, cols = linear_sum_assignment(cost)
    assert np.array_equal(rows, np.arange(cost.shape[0]))
    got = lo.linear_sum_assignment_rows(cost)
    assert np.array_equal(got, cols)


def test_oracle_on_log_softmax_scores():
    """the metric's actual input: -log_softmax of score matrices (rows of near-equal entries included)"""
    torch = pytest.importorskip('torch')
    g = torch.Generator().manual_seed(3)
    s = torch.randn(6, 40, 40, generator=g)
    s[1] = 0.0                                              # constant scores: identity matching (SciPy issue 11602)
    s[2] = torch.round(s[2])                                # many exact ties
    cost = (-torch.log_softmax(s, -1)).numpy()
    hits, matches = lo.accuracy_linear_assignment(cost)
    for b in range(6):
        _, cols = linear_sum_assignment(cost[b])
        assert np.array_equal(matches[b], cols)
        assert hits[b] == int(np.sum(cols == np.arange(40)))
    assert hits[1] == 40


def test_oracle_infeasible_raises():
    c = np.array([[np.inf, np.inf], [1.0, 2.0]])
    with pytest.raises(ValueError):
        lo.linear_sum_assignment_rows(c)
    with pytest.raises(ValueError):
        linear_sum_assignment(c)
