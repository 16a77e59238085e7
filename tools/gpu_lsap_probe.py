"""accuracy_linear_assignment: the device kernel (csrc/lsap.hip) against the reference's host loop (log_softmax on the device,
copy, scipy.optimize.linear_sum_assignment per graph: toolbox/metrics.py:92-116).  usage: python tools/gpu_lsap_probe.py"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scipy.optimize import linear_sum_assignment
from graph_neural_net_amd.metrics import accuracy_linear_assignment

dev = 'cuda:0'
for B, N in ((32, 50), (8, 200), (64, 120)):
    s = torch.randn(B, N, N, device=dev) * 3

    def host():
        w = torch.log_softmax(s, -1)
        acc = 0
        for b in range(B):
            _, p = linear_sum_assignment(-w[b].cpu().numpy())
            acc += int(np.sum(p == np.arange(N)))
        return acc

    def device():
        return accuracy_linear_assignment(s)[0]

    assert host() == device()
    for name, fn in (('host loop (reference)', host), ('device kernel', device)):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        print('B = %2d  N = %3d  %-22s %8.3f ms per call' % (B, N, name, (time.perf_counter() - t0) * 100))

# the kernel alone (cost already formed), event-timed
from graph_neural_net_amd import _lib
for B, N in ((32, 50), (8, 200), (64, 120)):
    cost = (-torch.log_softmax(torch.randn(B, N, N, device=dev) * 3, -1)).contiguous()
    correct = torch.empty(B, dtype=torch.int32, device=dev)
    st = _lib.stream_ptr()
    f = lambda: _lib.call('fgnn_lsap_accuracy', _lib.ptr(cost), N * N, N, None, B, N, _lib.ptr(correct), None, st)
    f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record()
    torch.cuda.synchronize()
    print('B = %2d  N = %3d  fgnn_lsap_accuracy alone %8.3f ms' % (B, N, e0.elapsed_time(e1) / 10))
