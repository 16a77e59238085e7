"""Isolated timing of the N <= 64 per-channel matmul entry points (gpu_mm_big_probe.py: the whole-matrix kernels on ragged batches) (forward / backward), eager launches, event-timed.
usage: python tools/gpu_mm_probe.py [G] ; FGNN_MM_WAVE=0 selects the workgroup-per-matrix kernels."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graph_neural_net_amd import _lib

G = int(sys.argv[1]) if len(sys.argv) > 1 else 64
Cc = 32
dev = 'cuda:0'


def slab(t, nrm=None, beta=None):
    g, c, n, _ = t.shape
    return _lib.make_slab(t, c * n * n, n * n, c, nrm=nrm, beta=beta)


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


for N in (32, 40, 48, 50, 56, 64):
    P = N * N
    a = torch.randn(G, Cc, N, N, device=dev)
    b = torch.randn(G, Cc, N, N, device=dev)
    dm = torch.randn(G, Cc, N, N, device=dev)
    nrm_a = torch.rand(G, Cc, 4, device=dev) + 0.5
    nrm_b = torch.rand(G, Cc, 4, device=dev) + 0.5
    out, da, db = torch.empty_like(a), torch.empty_like(a), torch.empty_like(a)
    s12a, s12b = torch.empty(G * Cc * 2, device=dev), torch.empty(G * Cc * 2, device=dev)
    sa, sb = slab(a, nrm=nrm_a), slab(b, nrm=nrm_b)
    st = _lib.stream_ptr()
    tf = timeit(lambda: _lib.call('fgnn_chan_matmul_fwd', C.byref(sa), C.byref(sb), None, G, N, _lib.ptr(out), Cc * P, P, st))
    tb = timeit(lambda: _lib.call('fgnn_chan_matmul_bwd', C.byref(sa), C.byref(sb), _lib.ptr(dm), Cc * P, P, None, G, N,
                                  _lib.ptr(da), _lib.ptr(db), Cc * P, P, _lib.ptr(s12a), _lib.ptr(s12b), st))
    tc = timeit(lambda: out.copy_(a))
    mb = G * Cc * P * 4 / 1e6
    print('N=%2d  fwd %6.1f us (%5.2f TB/s)   bwd %6.1f us (%5.2f TB/s)   copy of one slab %5.1f us (%5.2f TB/s)   slab %.1f MB'
          % (N, tf, 3 * mb / tf, tb, 5 * mb / tb, tc, 2 * mb / tc, mb))

# ---- the same forward at N = 50 cycling through K operand sets: K * 61 MB against the 256 MB Infinity Cache ----
N, P = 50, 2500
for K in (1, 2, 4, 8, 16):
    sets = []
    for _ in range(K):
        a = torch.randn(G, Cc, N, N, device=dev); b = torch.randn(G, Cc, N, N, device=dev)
        out = torch.empty_like(a); dm = torch.randn(G, Cc, N, N, device=dev); da = torch.empty_like(a); db = torch.empty_like(a)
        na = torch.rand(G, Cc, 4, device=dev) + 0.5; nb = torch.rand(G, Cc, 4, device=dev) + 0.5
        sets.append((slab(a, nrm=na), slab(b, nrm=nb), out, dm, da, db, a, b))
    st = _lib.stream_ptr()
    it = [0]
    def fwd():
        sa, sb, out = sets[it[0] % K][:3]; it[0] += 1
        _lib.call('fgnn_chan_matmul_fwd', C.byref(sa), C.byref(sb), None, G, N, _lib.ptr(out), Cc * P, P, st)
    def bwd():
        sa, sb, out, dm, da, db = sets[it[0] % K][:6]; it[0] += 1
        _lib.call('fgnn_chan_matmul_bwd', C.byref(sa), C.byref(sb), _lib.ptr(dm), Cc * P, P, None, G, N,
                  _lib.ptr(da), _lib.ptr(db), Cc * P, P, _lib.ptr(s12a), _lib.ptr(s12b), st)
    def cpy():
        s = sets[it[0] % K]; it[0] += 1
        s[2].copy_(s[6])
    print('K=%2d sets: fwd %5.1f us  bwd %5.1f us  copy %5.1f us' % (K, timeit(fwd, 64), timeit(bwd, 64), timeit(cpy, 64)))
    del sets
