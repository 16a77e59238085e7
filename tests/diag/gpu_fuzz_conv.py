#!/usr/bin/env python3
"""Diagnostic (not collected by pytest): random conv stacks through csrc/conv.hip's chain kernel (`_ConvChainFn`) against the per-layer
launches and the fp64 ATen definition with the MaskedTensor semantics -- output, input gradient, every parameter gradient, exact zeros in
the padding.  The widths are drawn around the boundaries the dispatch of `fgnn_conv_chain` branches on (1, 8, 16, 32, 64, 96, 128 and their
neighbours; first contraction 8 / 32 / 64 / 128 selects the all-full instantiation): the round-5 defect (a chain of 32-wide layers run by
the two-group all-full kernel) is a dispatch condition, and such a thing shows up here whatever the seed.
usage: python tests/diag/gpu_fuzz_conv.py [cases=300] [seed=0]   |   python tests/diag/gpu_fuzz_conv.py case cin w1,w2,.. G N gseed [n1,n2,..]"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from graph_neural_net_amd.layers import _ConvChainFn, _ConvFn, _chain_supported      # noqa: E402
from util import rel      # noqa: E402

DEV = 'cuda:0'
EDGE = [1, 2, 3, 7, 8, 9, 15, 16, 17, 24, 31, 32, 33, 40, 48, 63, 64, 65, 72, 96, 97, 100, 104, 127, 128]


def check(cin, widths, G, N, nv, gseed, chain=None):
    """-> list of failure messages for one conv stack (nv: int32 tensor of vertex counts or None)."""
    depth = len(widths)
    ragged = nv is not None
    chain = _chain_supported(cin, widths) if chain is None else chain
    mask = torch.zeros(G, 1, N, N, dtype=torch.float64)
    for g in range(G):
        n = N if nv is None else int(nv[g])
        mask[g, :, :n, :n] = 1
    gen = torch.Generator().manual_seed(gseed)
    x = (torch.randn(G, cin, N, N, generator=gen).double() * mask).float()
    ws, bs, k = [], [], cin
    for m in widths:
        ws.append(torch.randn(m, k, 1, 1, generator=gen) / k ** 0.5)
        bs.append(0.3 * torch.randn(m, generator=gen))
        k = m
    dy = (torch.randn(G, widths[-1], N, N, generator=gen).double() * mask).float()
    nvd = nv.to(DEV) if ragged else None

    def run(kind):
        xd = x.to(DEV).requires_grad_(True)
        wd = [w.to(DEV).requires_grad_(True) for w in ws]
        bd = [b.to(DEV).requires_grad_(True) for b in bs]
        if kind == 'chain':
            wb = []
            for w, b in zip(wd, bd):
                wb += [w, b]
            y = _ConvChainFn.apply(xd, nvd, *wb)
        else:
            y = xd
            for l, (w, b) in enumerate(zip(wd, bd)):
                y = _ConvFn.apply(y, nvd, w, b, l < len(wd) - 1)
        y.backward(dy.to(DEV))
        return [y.detach().cpu(), xd.grad.cpu()] + [w.grad.cpu() for w in wd] + [b.grad.cpu() for b in bd]

    ref = run('layers')
    got = run('chain') if chain else ref
    def aten(dt):
        xr = x.to(dt).requires_grad_(True)
        wr = [w.to(dt).requires_grad_(True) for w in ws]
        br = [b.to(dt).requires_grad_(True) for b in bs]
        yr = xr
        for l, (w, b) in enumerate(zip(wr, br)):
            yr = F.conv2d(yr, w, b)
            if l < len(wr) - 1:
                yr = F.relu(yr)
            yr = yr * mask.to(dt)
        yr.backward(dy.to(dt))
        return [yr.detach(), xr.grad] + [w.grad for w in wr] + [b.grad for b in br]

    truth, t32 = aten(torch.float64), aten(torch.float32)
    names = ['y', 'dx'] + ['dW%d' % l for l in range(depth)] + ['db%d' % l for l in range(depth)]
    msgs = []
    for nme, a, b_, t, c32 in zip(names, got, ref, truth, t32):
        # yard-stick: ATen's own fp32 evaluation against fp64 (a hidden pre-activation within rounding distance of zero is taken the
        # other way by any fp32 evaluation: seed 2 case 185 -- 871 k pre-activations -- is one, ATen fp32 6e-2 from fp64 as well)
        tol = max(2e-5, 4.0 * rel(c32, t))
        if not (rel(a, t) < tol):
            msgs.append('chain %s %.1e (ATen fp32 %.1e)' % (nme, rel(a, t), rel(c32, t)))
        if not (rel(b_, t) < tol):
            msgs.append('layers %s %.1e (ATen fp32 %.1e)' % (nme, rel(b_, t), rel(c32, t)))
    for kind, res in (('chain', got), ('layers', ref)):
        if (res[0].double() * (1 - mask)).abs().max() != 0 or (res[1].double() * (1 - mask)).abs().max() != 0:
            msgs.append(kind + ': padding not zero')
    return msgs


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'case':          # one stack: case cin w1,w2,.. G N gseed [n1,n2,..]
        cin, widths, G, N, gseed = int(sys.argv[2]), [int(v) for v in sys.argv[3].split(',')], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
        nv = torch.tensor([int(v) for v in sys.argv[7].split(',')], dtype=torch.int32) if len(sys.argv) > 7 else None
        msgs = check(cin, widths, G, N, nv, gseed)
        print('cin %d widths %s G %d N %d %s: %s' % (cin, widths, G, N, 'dense' if nv is None else nv.tolist(), '; '.join(msgs) or 'OK'))
        return 1 if msgs else 0
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    bad = skipped = 0
    for case in range(cases):
        depth = int(rng.integers(1, 4))
        cin = int(rng.choice(EDGE + [130, 160, 200, 256]))
        if rng.random() < 0.5:          # the MlpBlock_Real shape: every layer as wide as the last
            widths = [int(rng.choice(EDGE))] * depth
        else:
            widths = [int(rng.choice(EDGE)) for _ in range(depth)]
        chain = _chain_supported(cin, widths)
        skipped += not chain          # (then only the per-layer launches are checked)
        G = int(rng.integers(1, 5))
        N = int(rng.choice([3, 5, 8, 13, 23, 33]))
        ragged = bool(rng.integers(0, 2))
        nv = torch.tensor([int(rng.integers(1, N + 1)) for _ in range(G)], dtype=torch.int32) if ragged else None
        msgs = check(cin, widths, G, N, nv, case + 1000 * seed, chain)
        tag = 'case %3d: cin %3d widths %-16s G %d N %2d %s' % (case, cin, widths, G, N, ('nv=%s' % nv.tolist()) if ragged else 'dense')
        if msgs:
            print(tag, 'FAIL', '; '.join(msgs), flush=True)
        bad += bool(msgs)
    print('%d of %d cases failed (%d of them not chain shapes: per-layer launches only)' % (bad, cases, skipped))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
