#!/usr/bin/env python3
"""MlpBlock_Real (conv / ReLU chain + GraphNorm, models/layers.py:109-131) for a sweep of (input channels, width, depth) through the
module's HIP paths against fp64 torch autograd of the oracle's op sequence: output, input gradient and every parameter gradient.
usage (GPU box): python tests/diag/gpu_mlp_widths_sweep.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
from graph_neural_net_amd.layers import MlpBlock_Real      # noqa: E402
from oracle import fgnn_oracle as O                         # noqa: E402
from util import rel                                        # noqa: E402

DEV = 'cuda:0'
bad = 0
for cin, cout, depth in [(104, 32, 3), (75, 72, 3), (72, 32, 3), (80, 48, 2), (32, 48, 2), (64, 48, 2), (65, 32, 2), (65, 32, 3), (66, 16, 2), (96, 32, 1),
                         (128, 32, 2), (40, 32, 3), (72, 72, 2), (33, 32, 2), (19, 16, 2), (64, 32, 3), (100, 64, 3), (7, 40, 3)]:
    for N, B in ((9, 2), (33, 2)):
        torch.manual_seed(cin * 7 + cout + depth + N)
        mlp = MlpBlock_Real(cin, cout, depth).to(DEV)
        with torch.no_grad():
            for p in mlp.parameters():
                if p.dim() == 1:
                    p.add_(0.1 * torch.randn_like(p))
        x = torch.randn(B, cin, N, N, device=DEV, requires_grad=True)
        y = mlp(x)
        w = torch.randn_like(y)
        (y * w).sum().backward()
        sd = {k: v.detach().cpu().double() for k, v in mlp.state_dict().items()}
        ws = [sd['convs.%d.weight' % i] for i in range(depth)]
        bs = [sd['convs.%d.bias' % i] for i in range(depth)]
        leaves = [t.clone().requires_grad_(True) for t in ws + bs + [sd['gn.weight'], sd['gn.bias']]]
        x64 = x.detach().cpu().double().requires_grad_(True)
        y64 = O.mlp_block_real(x64, leaves[:depth], leaves[depth:2 * depth], leaves[-2], leaves[-1])
        (y64 * w.cpu().double()).sum().backward()
        names = ['convs.%d.weight' % i for i in range(depth)] + ['convs.%d.bias' % i for i in range(depth)] + ['gn.weight', 'gn.bias']
        got = dict(mlp.named_parameters())
        errs = {'y': rel(y.detach().cpu(), y64.detach()), 'dx': rel(x.grad.cpu(), x64.grad)}
        for n, l in zip(names, leaves):
            if n == 'convs.%d.bias' % (depth - 1):
                continue
            errs[n] = rel(got[n].grad.cpu(), l.grad.reshape(got[n].shape))
        worst = max(errs, key=errs.get)
        ok = errs[worst] < 1e-4
        bad += not ok
        print('cin %3d cout %3d depth %d N %2d: %s worst %s %.2e%s' % (cin, cout, depth, N, 'ok  ' if ok else 'FAIL', worst, errs[worst],
              '' if ok else '  ' + ' '.join('%s=%.1e' % kv for kv in errs.items())), flush=True)
print('%d failures' % bad)
