#!/usr/bin/env python3
"""The fused 64-wide MLP kernels (csrc/mlp64.hip) in isolation: 20 back-to-back launches in a replayed HIP graph, forward and backward,
for the input widths of a 64-feature model (2, 64, 66 = block 1's mlp3, 128) at G = 64, N = 50, and at G = 1, N = 4 (what a launch costs
before its first tile: image construction, workgroup reduction).   usage: python tools/gpu_mlp64_probe.py"""
import ctypes as C

import torch

from graph_neural_net_amd import _lib


def graph_time(fn, n=20, replays=10):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * replays) * 1e3


def main():
    dev = torch.device('cuda', 0)
    lib = _lib.load()
    for G, N in ((64, 50), (1, 4)):
        P = N * N
        for cin in (2, 64, 66, 128):
            x = torch.randn(G, cin, N, N, device=dev)
            ws = [torch.randn(64, k, device=dev) / k ** 0.5 for k in (cin, 64, 64)]
            bs = [torch.zeros(64, device=dev) for _ in range(3)]
            out, dz, dx = torch.empty(G, 64, N, N, device=dev), torch.randn(G, 64, N, N, device=dev), torch.empty(G, cin, N, N, device=dev)
            cnt = lib.fgnn_mlp64_param_count(cin)
            wpart = torch.empty(lib.fgnn_mlp64_num_workgroups() * cnt, device=dev)
            a = _lib.Mlp64Args()
            a.x, a.x_gstride, a.x_ld, a.cin = x.data_ptr(), cin * P, P, cin
            packed = torch.empty(lib.fgnn_mlp64_packed_floats(cin), device=dev)
            tp = graph_time(lambda: _lib.call('fgnn_mlp64_pack', *[_lib.ptr(t) for t in ws + bs], cin, _lib.ptr(packed), _lib.stream_ptr()))
            a.packed = packed.data_ptr()
            a.G, a.N = G, N
            a.out, a.o_gstride, a.o_ld = out.data_ptr(), 64 * P, P
            a.dz, a.dz_gstride, a.dz_ld = dz.data_ptr(), 64 * P, P
            a.wpart = wpart.data_ptr()
            tf = graph_time(lambda: _lib.call('fgnn_mlp64_fwd', C.byref(a), _lib.stream_ptr()))
            tb0 = graph_time(lambda: _lib.call('fgnn_mlp64_bwd', C.byref(a), _lib.stream_ptr()))
            a.dx, a.dx_gstride, a.dx_ld = dx.data_ptr(), cin * P, P
            tb1 = graph_time(lambda: _lib.call('fgnn_mlp64_bwd', C.byref(a), _lib.stream_ptr()))
            mf = 2.0 * G * P * 64 * (cin + 128)
            print('G %2d N %2d cin %3d: pack %4.1f us  fwd %6.1f us (%4.1f TF)   bwd %6.1f us   bwd + dx %6.1f us (%4.1f TF executed)'
                  % (G, N, cin, tp, tf, mf / tf * 1e-6, tb0, tb1, (mf * 3 - 2.0 * G * P * 64 * 64) / tb1 * 1e-6))


if __name__ == '__main__':
    main()
