"""Isolated timing of the whole-matrix per-channel products (64 < N <= 128) on a ragged batch: plain launch order against the
largest-graph-first order of fgnn_ragged_tile_ranges_order, with and without one workgroup per product in the backward.
usage: python tools/gpu_mm_big_probe.py [pairs]     (default 8: the cfg5 batch of bench.py)"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graph_neural_net_amd import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
Cc, dev = 32, 'cuda:0'
g = torch.Generator().manual_seed(5)
n = torch.randint(30, 121, (B,), generator=g)
if B == 8:
    n = torch.tensor([70, 35, 67, 103, 95, 78, 56, 103])
nv = torch.cat([n, n]).to(torch.int32).to(dev)
G, N = 2 * B, int(n.max())
P = N * N


def timeit(fn, reps=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


K = 6       # operand sets cycled through (cache-cold, as inside a training step)
sets = []
for _ in range(K):
    a, b, dm = (torch.randn(G, Cc, N, N, device=dev) for _ in range(3))
    na, nb = torch.rand(G, Cc, 4, device=dev) + 0.5, torch.rand(G, Cc, 4, device=dev) + 0.5
    sets.append((_lib.make_slab(a, Cc * P, P, Cc, nrm=na), _lib.make_slab(b, Cc * P, P, Cc, nrm=nb), dm, torch.empty_like(a),
                 torch.empty_like(a), torch.empty_like(a), a, b, na, nb))
s12a, s12b = torch.empty(G * Cc * 2, device=dev), torch.empty(G * Cc * 2, device=dev)
ranges = torch.empty(_lib.FGNN_RANGE_WG + 1, dtype=torch.int32, device=dev)
order = torch.empty(G, dtype=torch.int32, device=dev)
st = _lib.stream_ptr()
_lib.call('fgnn_ragged_tile_ranges_order', _lib.ptr(nv), G, N, _lib.ptr(ranges), _lib.ptr(order), st)
flops = float((2 * nv.double() ** 3).sum() * Cc)
print('G = %d graphs, N = %d, n = %s: %.2f GFLOP per product, %.1f MB per slab' % (G, N, n.tolist(), flops / 1e9, G * Cc * P * 4 / 1e6))
it = [0]
lib = _lib.load()
for name, variant, o in (('plain order', 1, None), ('largest first, one workgroup per matrix', 3, order),
                         ('largest first, backward: one workgroup per product', 1, order)):
    lib.fgnn_debug_matmul_variant(variant)

    def fwd():
        s = sets[it[0] % K]; it[0] += 1
        _lib.call('fgnn_chan_matmul_fwd_ord', C.byref(s[0]), C.byref(s[1]), _lib.ptr(nv), G, N, _lib.ptr(s[3]), Cc * P, P, _lib.ptr(o), 0, st)

    def bwd():
        s = sets[it[0] % K]; it[0] += 1
        _lib.call('fgnn_chan_matmul_bwd_ord', C.byref(s[0]), C.byref(s[1]), _lib.ptr(s[2]), Cc * P, P, _lib.ptr(nv), G, N,
                  _lib.ptr(s[4]), _lib.ptr(s[5]), Cc * P, P, _lib.ptr(s12a), _lib.ptr(s12b), _lib.ptr(o), 0, st)

    tf, tb = timeit(fwd), timeit(bwd)
    print('%-52s fwd %6.1f us (%5.1f TFLOP/s)   bwd %6.1f us (%5.1f TFLOP/s)' % (name, tf, flops / tf / 1e6, tb, 2 * flops / tb / 1e6))
lib.fgnn_debug_matmul_variant(1)
