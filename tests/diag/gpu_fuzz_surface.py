#!/usr/bin/env python3
"""Random shapes through the module surface's structured path (Siamese_Node_Exp(input_form='tensor_representation').fused_step): dense and
MaskedTensor batches, fp32 and bf16, 1 - 3 blocks, N = 1 ... 140, directed graphs / self loops, garbage in the MaskedTensor padding -- each
against the bit-packed engine step on the same words (scores bit for bit, gradients bit for bit for constant-size batches, to 1e-6 for
ragged ones: device reciprocal against host division), captured and eager, and against the dense form of the same module (forward to
rounding).  One corrupted batch per case must be refused.
usage (GPU box): python tests/diag/gpu_fuzz_surface.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
from graph_neural_net_amd import synthetic                          # noqa: E402
from graph_neural_net_amd.engine import FgnnEngine, ParamLayout     # noqa: E402
from graph_neural_net_amd.masked import from_list                   # noqa: E402
from graph_neural_net_amd.siamese import Siamese_Node_Exp           # noqa: E402

DEV = 'cuda:0'
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def rep(w):
    x = np.zeros((2,) + w.shape, dtype=np.float32)
    x[0] = w
    x[1][np.arange(w.shape[0]), np.arange(w.shape[0])] = w.sum(-1)
    return torch.from_numpy(x)


def pk(x):
    return torch.from_numpy(synthetic.pack_adjacency(x[:, 0].cpu().numpy()).view(np.int32)).to(DEV)


fails = 0
for case in range(cases):
    blocks = int(rng.integers(1, 4))
    B = int(rng.integers(1, 6))
    ragged = bool(rng.integers(0, 2))
    bf16 = bool(rng.integers(0, 3) == 0)
    nmax = int(rng.integers(1, 141))
    sizes = [int(rng.integers(1, nmax + 1)) for _ in range(B)] if ragged else [nmax] * B
    if ragged:
        sizes[int(rng.integers(0, B))] = nmax
    dens = float(rng.uniform(0.05, 0.9))
    mk = lambda n: (rng.random((n, n)) < dens).astype(np.float32)           # directed, self loops allowed
    xs, ys = [rep(mk(n)) for n in sizes], [rep(mk(n)) for n in sizes]
    ne = dict(type='node_embedding', block_init='block_emb', block_inside='block', num_blocks=blocks, in_features=32, out_features=32, depth_of_mlp=3)
    if ragged:
        ne['constant_n_vertices'] = False
    torch.manual_seed(case)
    model = Siamese_Node_Exp(2, dict(ne), precision='bf16' if bf16 else 'fp32', input_form='tensor_representation').to(DEV)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith('.bias'):
                p.add_(0.1 * torch.randn_like(p))
    tag = 'case %d: blocks %d B %d %s %s sizes %s' % (case, blocks, B, 'ragged' if ragged else 'const', 'bf16' if bf16 else 'fp32', sizes)
    try:
        gran = model.RAGGED_GRANULE
        N = -(-nmax // gran) * gran if ragged else nmax
        if ragged:
            a, b = from_list([t.to(DEV) for t in xs], dims=(1, 2), base_name='N'), from_list([t.to(DEV) for t in ys], dims=(1, 2), base_name='M')
            for m in (a, b):
                for i, n in enumerate(sizes):
                    m.tensor.rename(None)[i, :, n:, :] = 9.0
                    m.tensor.rename(None)[i, :, :, n:] = -2.5
        else:
            a, b = torch.stack(xs).to(DEV), torch.stack(ys).to(DEV)
        pad = lambda lst: torch.stack([torch.nn.functional.pad(t, (0, N - t.shape[-1], 0, N - t.shape[-1])) for t in lst])
        lay = ParamLayout(2, blocks, 32, 32, 3)
        params = lay.flatten({k[len('node_embedder.'):]: v for k, v in model.state_dict().items()}, DEV)
        if bf16:
            from graph_neural_net_amd.engine16 import FgnnEngineBF16 as Eng
        else:
            Eng = FgnnEngine
        eng = Eng(lay, 2 * B, N, DEV, ragged=ragged, block1='structured')
        ge = torch.zeros_like(params)
        nv = torch.tensor(sizes * 2, dtype=torch.int32, device=DEV) if ragged else None
        se, le = eng.step(params, ge, None, nvalid=nv, bits=torch.cat([pk(pad(xs)), pk(pad(ys))]).contiguous())
        torch.cuda.synchronize()
        se, ge = se.clone(), ge.clone()
        for cap in (False, True, True):
            l, s = model.fused_step(a, b, capture=cap)
            st = s.tensor.rename(None) if ragged else s
            g = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
            assert torch.equal(st, se), 'scores (capture=%s)' % cap
            if ragged:
                assert ((g - ge).norm() <= 1e-6 * ge.norm() + 1e-12).item(), 'gradients'
            else:
                assert torch.equal(g, ge), 'gradients (capture=%s)' % cap
        model.check_input_form()
        # one corrupted entry inside a valid corner must be refused
        i = int(rng.integers(0, B))
        n = sizes[i]
        bad = (a.tensor.rename(None) if ragged else a)
        r, c = int(rng.integers(0, n)), int(rng.integers(0, n))
        bad[i, 0, r, c] = 0.5
        model.fused_step(a, b)
        try:
            model.check_input_form()
            raise AssertionError('a 0.5 entry was not refused')
        except RuntimeError:
            pass
    except Exception as exc:          # noqa: BLE001
        fails += 1
        print('FAIL', tag, '--', repr(exc)[:300], flush=True)
    if case % 10 == 9:
        print('%d cases, %d failures' % (case + 1, fails), flush=True)
print('%d cases, %d failures' % (cases, fails))
sys.exit(1 if fails else 0)
