"""CPU restatement of the assignment behind ``accuracy_linear_assignment``.  TEST INFRASTRUCTURE ONLY (only ``tests/`` import it).

The reference (toolbox/metrics.py:92-116) calls ``scipy.optimize.linear_sum_assignment(-log_softmax(scores))`` per graph and
counts ``preds == arange(n)``.  SciPy is a third-party dependency of the reference (requirements: scipy, no pin; 1.15.3 in this
image); its solver is the shortest-augmenting-path algorithm of D. F. Crouse, "On implementing 2D rectangular assignment
algorithms", IEEE Trans. Aerospace and Electronic Systems 52(4), 2016 (scipy/optimize/rectangular_lsap).  This file restates
that algorithm for a square fp64 cost matrix -- the `remaining` list filled in reverse and compacted by moving its last entry
into the freed slot, a strictly smaller path cost or an equal one on a still unassigned column wins a row scan -- in plain
Python loops, so that the ASSIGNMENT (not only its cost) can be compared with SciPy's, ties included.

Parity status: PINNED against scipy.optimize.linear_sum_assignment itself (tests/test_lsap_oracle.py runs both here and on the
GPU box: random fp32 costs, small-integer costs full of ties, constant matrices, model scores).
"""
import math

import numpy as np


def linear_sum_assignment_rows(cost):
    """cost: (n, n) array-like -> col4row (n,) int64: the column matched to each row; raises ValueError when infeasible."""
    c = np.asarray(cost, dtype=np.float64)
    n = c.shape[0]
    assert c.shape == (n, n)
    u = [0.0] * n
    v = [0.0] * n
    spc = [math.inf] * n
    path = [-1] * n
    col4row = [-1] * n
    row4col = [-1] * n
    for cur in range(n):
        min_val = 0.0
        remaining = [n - it - 1 for it in range(n)]
        num_remaining = n
        SR = [False] * n
        SC = [False] * n
        spc = [math.inf] * n
        sink = -1
        i = cur
        while sink == -1:
            index = -1
            lowest = math.inf
            SR[i] = True
            for it in range(num_remaining):
                j = remaining[it]
                r = ((min_val + c[i, j]) - u[i]) - v[j]
                if r < spc[j]:
                    path[j] = i
                    spc[j] = r
                if spc[j] < lowest or (spc[j] == lowest and row4col[j] == -1):
                    lowest = spc[j]
                    index = it
            min_val = lowest
            if min_val == math.inf:
                raise ValueError('cost matrix is infeasible')
            j = remaining[index]
            if row4col[j] == -1:
                sink = j
            else:
                i = row4col[j]
            SC[j] = True
            num_remaining -= 1
            remaining[index] = remaining[num_remaining]
        u[cur] += min_val
        for k in range(n):
            if SR[k] and k != cur:
                u[k] += min_val - spc[col4row[k]]
        for k in range(n):
            if SC[k]:
                v[k] -= min_val - spc[k]
        j = sink
        while True:
            k = path[j]
            row4col[j] = k
            col4row[k], j = j, col4row[k]
            if k == cur:
                break
    return np.asarray(col4row, dtype=np.int64)


def accuracy_linear_assignment(cost, sizes=None):
    """cost: (B, N, N) = -log_softmax(scores); sizes: valid vertices per graph.  Returns the per-graph hit counts and matchings."""
    cost = np.asarray(cost)
    B, N, _ = cost.shape
    hits, matches = [], []
    for b in range(B):
        n = N if sizes is None else int(sizes[b])
        m = linear_sum_assignment_rows(cost[b, :n, :n])
        hits.append(int(np.sum(m == np.arange(n))))
        matches.append(m)
    return hits, matches
