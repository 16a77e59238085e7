"""Isolated forward per-channel product at N = 50 (64 graphs x 32 channels): the 8-byte-access wave kernel (variant 1, shipped) against the
four-byte one it replaces (variant 9) and the workgroup-per-matrix kernel (variant 0); 30 back-to-back launches in a replayed graph,
three operand sets (the protocol of tools/gpu_mm_ablate.py).  Variant 17 = two matrices per wave (chan_matmul_fwd_wp_kernel, N = 49 ... 56).   usage (GPU box): python tools/gpu_mm_wide_probe.py [N ...]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from graph_neural_net_amd import _lib
lib = _lib.load()
G, Cc = 64, 32
dev = 'cuda:0'
for N in [int(v) for v in sys.argv[1:]] or [50]:
    P = N * N
    K = 3
    sets = []
    for _ in range(K):
        a, b = torch.randn(G, Cc, N, N, device=dev), torch.randn(G, Cc, N, N, device=dev)
        na, nb = torch.rand(G, Cc, 4, device=dev) + 0.5, torch.rand(G, Cc, 4, device=dev) + 0.5
        sets.append((_lib.make_slab(a, Cc * P, P, Cc, nrm=na), _lib.make_slab(b, Cc * P, P, Cc, nrm=nb), torch.empty_like(a), a, b, na, nb))
    for variant in (1, 9, 17, 0, 1, 9, 17):
        lib.fgnn_debug_matmul_variant(variant)
        it = [0]

        def fwd():
            s = sets[it[0] % K]
            it[0] += 1
            _lib.call('fgnn_chan_matmul_fwd', C.byref(s[0]), C.byref(s[1]), None, G, N, _lib.ptr(s[2]), Cc * P, P, _lib.stream_ptr())
        for _ in range(10):
            fwd()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(30):
                fwd()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        print('N = %d variant %d: %.2f us per launch' % (N, variant, e0.elapsed_time(e1) / 300 * 1e3), flush=True)
    lib.fgnn_debug_matmul_variant(1)
