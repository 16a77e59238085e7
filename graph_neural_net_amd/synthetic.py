"""Synthetic graph-pair generator (host side, numpy only, explicit seeds).

Produces the input contract of the hot path: a pair ``(X, X_noise)`` of
``(2, n, n)`` fp32 tensors with channel 0 the symmetric 0/1 adjacency (zero
diagonal) and channel 1 ``diag(degree)`` -- the tensor representation built by
``adjacency_matrix_to_tensor_representation`` (reference
loaders/data_generator.py:118-125).  Graph families follow the reference's
generators (``:38-68``) and its Erdos-Renyi edge noise (``:79-87``), but are
written on ``numpy.random.default_rng`` so they need neither networkx nor the
network and are reproducible from a seed on the GPU box.
"""
import numpy as np
import torch


def erdos_renyi(rng, n, p):
    """Symmetric 0/1 adjacency, zero diagonal, upper-triangular Bernoulli(p)."""
    u = rng.random((n, n)) < p
    w = np.triu(u, 1)
    return (w | w.T).astype(np.float32)


def regular_degree(n, edge_density):
    """Degree used by the reference for 'Regular' graphs (data_generator.py:59-66)."""
    d = int(edge_density * n)
    if (n * d) % 2 == 1:
        d += 1
    return d


def random_regular(rng, n, d, swaps_per_edge=10):
    """Random d-regular simple graph: circulant seed, degree-preserving
    double-edge swaps, random relabelling."""
    if d >= n or (n * d) % 2:
        raise ValueError('no %d-regular graph on %d vertices' % (d, n))
    w = np.zeros((n, n), dtype=bool)
    idx = np.arange(n)
    for k in range(1, d // 2 + 1):
        w[idx, (idx + k) % n] = True
        w[(idx + k) % n, idx] = True
    if d % 2:
        w[idx, (idx + n // 2) % n] = True
        w[(idx + n // 2) % n, idx] = True
    edges = np.argwhere(np.triu(w, 1))
    m = len(edges)
    for _ in range(swaps_per_edge * m):
        a, b = rng.integers(0, m, size=2)
        if a == b:
            continue
        u, v = edges[a]
        s, t = edges[b]
        if rng.random() < 0.5:
            s, t = t, s
        # (u,v),(s,t) -> (u,t),(s,v)
        if u == t or s == v or u == s or v == t:
            continue
        if w[u, t] or w[s, v]:
            continue
        w[u, v] = w[v, u] = False
        w[s, t] = w[t, s] = False
        w[u, t] = w[t, u] = True
        w[s, v] = w[v, s] = True
        edges[a] = (min(u, t), max(u, t))
        edges[b] = (min(s, v), max(s, v))
    perm = rng.permutation(n)
    w = w[np.ix_(perm, perm)]
    return w.astype(np.float32)


def noise_erdos_renyi(rng, w, noise, edge_density):
    """W' = W(1-Z1) + (1-W)Z2 with Z1~ER(noise), Z2~ER(p*noise/(1-p))."""
    n = w.shape[0]
    z1 = erdos_renyi(rng, n, noise)
    z2 = erdos_renyi(rng, n, edge_density * noise / (1.0 - edge_density))
    return w * (1.0 - z1) + (1.0 - w) * z2


def tensor_representation(w):
    """(n,n) adjacency -> (2,n,n): ch0 = W, ch1 = diag(row sums)."""
    n = w.shape[0]
    x = np.zeros((2, n, n), dtype=np.float32)
    x[0] = w
    x[1][np.arange(n), np.arange(n)] = w.sum(1)
    return x


def make_pair(rng, n, family='Regular', edge_density=0.2, noise=0.1):
    if family == 'Regular':
        w = random_regular(rng, n, regular_degree(n, edge_density))
    elif family == 'ErdosRenyi':
        w = erdos_renyi(rng, n, edge_density)
    else:
        raise ValueError('unknown graph family %r' % (family,))
    wn = noise_erdos_renyi(rng, w, noise, edge_density)
    return tensor_representation(w), tensor_representation(wn)


def make_batch(seed, batch, n, family='Regular', edge_density=0.2, noise=0.1):
    """Constant-n batch: two (batch, 2, n, n) fp32 torch tensors."""
    rng = np.random.default_rng(seed)
    a = np.empty((batch, 2, n, n), dtype=np.float32)
    b = np.empty((batch, 2, n, n), dtype=np.float32)
    for i in range(batch):
        a[i], b[i] = make_pair(rng, n, family, edge_density, noise)
    return torch.from_numpy(a), torch.from_numpy(b)


def make_ragged_batch(seed, batch, n_lo, n_hi, family='ErdosRenyi', edge_density=0.2, noise=0.1):
    """Ragged batch: two lists of (2, n_i, n_i) tensors, n_i ~ U{n_lo..n_hi}, both
    graphs of a pair share n_i."""
    rng = np.random.default_rng(seed)
    xs, ys = [], []
    for _ in range(batch):
        n = int(rng.integers(n_lo, n_hi + 1))
        a, b = make_pair(rng, n, family, edge_density, noise)
        xs.append(torch.from_numpy(a))
        ys.append(torch.from_numpy(b))
    return xs, ys


def pack_adjacency(w_batch):
    """(G, n, n) 0/1 adjacency (numpy) -> (G, n, ceil(n/32)) uint32, bit j of word row i = W[i][j]
    (little-endian bit order): the compact wire format expanded on the device by
    ``inputs.expand_adjacency``."""
    w = np.asarray(w_batch) != 0
    g, n, _ = w.shape
    words = (n + 31) // 32
    padded = np.zeros((g, n, words * 32), dtype=bool)
    padded[:, :, :n] = w
    bits = np.packbits(padded.reshape(g, n, words, 32), axis=-1, bitorder='little')
    return np.ascontiguousarray(bits).view(np.uint32).reshape(g, n, words)
