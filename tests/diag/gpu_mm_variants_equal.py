"""Diagnostic (GPU): the wave-per-matrix forward matmul (N <= 64) against the workgroup-per-matrix kernel it replaced --
same k-step order, same normalisation expression, so the outputs must be bit-identical (dense, ragged, normalised).
usage: python tests/diag/gpu_mm_variants_equal.py"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from graph_neural_net_amd import _lib

dev = 'cuda:0'
lib = _lib.load()


def slab(t, nrm=None, beta=None):
    g, c, n, _ = t.shape
    return _lib.make_slab(t, c * n * n, n * n, c, nrm=nrm, beta=beta)


torch.manual_seed(0)
for N in (1, 2, 7, 8, 9, 20, 31, 32, 33, 40, 50, 63, 64):
    G, Cc = 3, 4
    a = torch.randn(G, Cc, N, N, device=dev)
    b = torch.randn(G, Cc, N, N, device=dev)
    nrm_a = torch.rand(G, Cc, 4, device=dev) + 0.5
    nrm_b = torch.rand(G, Cc, 4, device=dev) + 0.5
    beta = torch.randn(Cc, device=dev)
    nv = torch.tensor([N, max(1, N // 2), max(0, N - 1)], dtype=torch.int32, device=dev)
    res = []
    for variant in (1, 0):
        lib.fgnn_debug_matmul_variant(variant)
        out = torch.full((G, Cc, N, N), 3.0, device=dev)
        sa, sb = slab(a, nrm=nrm_a, beta=beta), slab(b, nrm=nrm_b, beta=beta)
        _lib.call('fgnn_chan_matmul_fwd', C.byref(sa), C.byref(sb), _lib.ptr(nv), G, N, _lib.ptr(out), Cc * N * N, N * N,
                  _lib.stream_ptr())
        torch.cuda.synchronize()
        res.append(out.cpu())
    lib.fgnn_debug_matmul_variant(1)
    print('N=%2d bit-identical: %s (max diff %.3g)' % (N, torch.equal(res[0], res[1]), (res[0] - res[1]).abs().max().item()))
    assert torch.equal(res[0], res[1])
