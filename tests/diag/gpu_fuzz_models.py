#!/usr/bin/env python3
"""Diagnostic (not collected by pytest): random MODEL configurations through the module surface (Siamese_Node_Exp) against the
oracle -- original_features_num in {1, 2, 3, 7, 32}, widths 4..72, depth 1..3, 1..3 blocks, constant-size and ragged batches:
exact engine, zero-padded engine and the per-layer conv.hip path.  Gates: scores within max(4x the fp32 oracle's distance to
fp64, 3e-5); flat gradient within 4x the fp32 oracle's distance to fp64 + 2 % (a ReLU tie, see gpu_fuzz_shapes.py).
`wide`: widths up to 128 and up to 64 input channels as well (the two- and four-group instantiations of the conv chain, split input-gradient chains).
usage: python tests/diag/gpu_fuzz_models.py [cases=60] [seed=0] [wide]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from graph_neural_net_amd.masked import from_list               # noqa: E402
from graph_neural_net_amd.siamese import Siamese_Node_Exp        # noqa: E402
from oracle import fgnn_oracle as O                              # noqa: E402
from util import is_zero_grad, rel                               # noqa: E402

DEV = 'cuda:0'


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    wide = len(sys.argv) > 3 and sys.argv[3] == 'wide'
    rng = np.random.default_rng(seed)
    torch.set_num_threads(16)
    bad = 0
    for case in range(cases):
        c0 = int(rng.choice([1, 2, 2, 3, 7, 32] + ([40, 64] if wide else [])))
        win = int(rng.choice([4, 8, 16, 24, 32, 32, 40, 72] + ([48, 64, 64, 96, 100, 128] if wide else [])))
        wout = int(rng.choice([4, 16, 32, 32, 48] + ([64, 64, 96, 128] if wide else [])))
        depth = int(rng.integers(1, 4))
        nblk = int(rng.integers(1, 4))
        ragged = bool(rng.integers(0, 2))
        B = int(rng.integers(1, 4))
        nmax = int(rng.choice([5, 12, 31, 33] + ([] if wide else [50, 70])))
        ns = [int(rng.integers(2, nmax + 1)) for _ in range(B)] if ragged else [nmax] * B
        torch.manual_seed(case)
        sd = O.init_state_dict(original_features_num=c0, num_blocks=nblk, in_features=win, out_features=wout, depth_of_mlp=depth)
        g = torch.Generator().manual_seed(case)
        for k in sd:                                     # non-trivial biases / affine parameters
            if k.endswith('.bias') and sd[k].dim() == 1:
                sd[k] = sd[k] + 0.1 * torch.randn(sd[k].shape, generator=g)
            elif k.endswith('gn.weight'):
                sd[k] = sd[k] * (1.0 + 0.2 * torch.randn(sd[k].shape, generator=g))
            elif k.endswith('gn.bias'):
                sd[k] = sd[k] + 0.05 * torch.randn(sd[k].shape, generator=g)
        xs = [torch.randn(c0, n, n, generator=g) for n in ns]
        ys = [torch.randn(c0, n, n, generator=g) for n in ns]
        s32, l32, g32 = O.step_fwd_bwd_ragged(xs, ys, sd)
        s64, l64, g64 = O.step_fwd_bwd_ragged([t.double() for t in xs], [t.double() for t in ys], {k: v.double() for k, v in sd.items()})
        ne = dict(type='node_embedding', block_init='block_emb', block_inside='block', num_blocks=nblk, in_features=win,
                  out_features=wout, depth_of_mlp=depth, constant_n_vertices=not ragged)
        model = Siamese_Node_Exp(c0, ne).to(DEV)
        model.load_state_dict({'node_embedder.' + k: v for k, v in sd.items()})
        net = model.node_embedder
        path = 'conv.hip modules' if net._standard_layout() is None else ('padded engine' if net._pad is not None else 'engine')
        if ragged:
            scores = model(from_list([t.to(DEV) for t in xs], dims=(1, 2), base_name='N'), from_list([t.to(DEV) for t in ys], dims=(1, 2), base_name='M'))
        else:
            scores = model(torch.stack(xs).to(DEV), torch.stack(ys).to(DEV))
        loss = model.loss(scores)
        loss.backward()
        torch.cuda.synchronize()
        msgs = []
        for i, (a, b32, b64) in enumerate(zip(list(scores), s32, s64)):
            if tuple(a.shape) != tuple(b64.shape) or rel(a.detach().cpu(), b64) > max(4 * rel(b32, b64), 3e-5):
                msgs.append('scores pair %d: %.2e (oracle32 %.2e)' % (i, rel(a.detach().cpu(), b64), rel(b32, b64)))
        if abs(loss.item() - l64.item()) > 2e-5 * abs(l64.item()) + 1e-6:
            msgs.append('loss %.7f vs %.7f' % (loss.item(), l64.item()))
        keys = [k for k in g64 if not is_zero_grad(k, depth)]
        grads = {n[len('node_embedder.'):]: p.grad.cpu() for n, p in model.named_parameters()}
        a = torch.cat([grads[k].reshape(-1).double() for k in keys])
        b = torch.cat([g32[k].reshape(-1).double() for k in keys])
        t = torch.cat([g64[k].reshape(-1) for k in keys])
        if not torch.isfinite(a).all() or (a - t).norm() > 4.0 * (b - t).norm() + 2e-2 * t.norm() + 1e-6:
            msgs.append('grads: ours-vs-fp64 %.2e, oracle32-vs-fp64 %.2e, |g| %.2e' % ((a - t).norm().item(), (b - t).norm().item(), t.norm().item()))
        print('case %3d: c0 %2d in %2d out %2d depth %d blocks %d %-22s [%s]' % (case, c0, win, wout, depth, nblk, ('ns=%s' % ns) if ragged else 'B=%d N=%d' % (B, nmax), path),
              'OK' if not msgs else 'FAIL ' + '; '.join(msgs), flush=True)
        bad += bool(msgs)
    print('%d of %d cases failed' % (bad, cases))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
