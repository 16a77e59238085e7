"""CPU: oracle/fgnn_oracle_pinned.py (the op sequence of the reference with the ReLU / arg-max decisions as INPUTS) against the
committed vectors of tests/golden/pinned_decisions.npz -- the reference's own fp32 and fp64 runs with the decisions its forward
hooks saw (make_golden.py round5, where the comparison is torch.equal) -- and, where /root/reference exists, live."""
import os
import subprocess
import sys

import pytest
import torch

from oracle import fgnn_oracle as O
from oracle import fgnn_oracle_pinned as OP
from util import ROOT, load_golden, rel, sub

REF = '/root/reference'


def _case():
    d = load_golden('pinned_decisions.npz')
    return d, sub(d, 'sd/'), d['x1'], d['x2']


@pytest.mark.parametrize('tag,dtype,tol', [('f32', torch.float32, 1e-6), ('f64', torch.float64, 1e-13)])
def test_pinned_oracle_reproduces_the_reference_on_its_own_decisions(tag, dtype, tol):
    d, sd, x1, x2 = _case()
    masks, idx = OP.unpack_decisions({k: v.numpy() for k, v in d.items()}, prefix=tag + '/')
    s, l, g = OP.step_fwd_bwd_pinned(x1, x2, sd, masks, idx, dtype=dtype)
    assert rel(s, d[tag + '/scores']) <= tol and abs(l.item() - d[tag + '/loss'].item()) <= tol * abs(l.item())
    for k, v in sub(d, tag + '/grad/').items():
        assert rel(g[k], v) <= tol, (k, rel(g[k], v))
    # the decisions ARE those of the plain oracle in that precision (it is torch.equal to the reference, test_oracle_pinned.py)
    m2, i2 = OP.collect_decisions(torch.cat([x1, x2]).to(dtype), {k: v.to(dtype) for k, v in sd.items()})
    assert torch.equal(i2, idx) and all(torch.equal(m2[k], masks[k]) for k in masks)
    s0, l0, g0 = O.step_fwd_bwd(x1.to(dtype), x2.to(dtype), {k: v.to(dtype) for k, v in sd.items()})
    for k in g0:
        assert rel(g[k], g0[k]) <= tol


def test_fp64_arithmetic_on_the_fp32_branch():
    """What the GPU test computes: fp64 arithmetic, decisions of an fp32 evaluation.  On this fixture the fp32 and the fp64 run of
    the reference take the same decisions, so the cross evaluation equals the plain fp64 gradient; flipping ONE decision on a
    pixel that carries gradient moves the result by far more than rounding -- the pinned function follows its inputs."""
    d, sd, x1, x2 = _case()
    raw = {k: v.numpy() for k, v in d.items()}
    m32, i32 = OP.unpack_decisions(raw, prefix='f32/')
    s, l, g = OP.step_fwd_bwd_pinned(x1, x2, sd, m32, i32, dtype=torch.float64)
    for k, v in sub(d, 'x64on32/grad/').items():
        assert rel(g[k], v) <= 1e-13
        assert rel(g[k], d['f64/grad/' + k]) <= 1e-13
    # flip the decision of the largest hidden pre-activation... any live pixel: take one that is ON in the last block's mlp3
    key = (2, 3, 1)
    flipped = {k: v.clone() for k, v in m32.items()}
    on = flipped[key].nonzero()[0]
    flipped[key][tuple(on)] = False
    _, _, g2 = OP.step_fwd_bwd_pinned(x1, x2, sd, flipped, i32, dtype=torch.float64)
    moved = max(rel(g2[k], g[k]) for k in g if not k.endswith('convs.2.bias'))
    assert moved > 1e-7, moved
    # ... and another arg-max row moves it too
    i_f = i32.clone()
    i_f[0, 0, 0] = (i_f[0, 0, 0] + 1) % x1.shape[-1]
    _, _, g3 = OP.step_fwd_bwd_pinned(x1, x2, sd, m32, i_f, dtype=torch.float64)
    assert max(rel(g3[k], g[k]) for k in g if not k.endswith('convs.2.bias')) > 1e-7


def test_ragged_pinned_step_equals_the_per_graph_oracle():
    """step_fwd_bwd_pinned_ragged on a padded batch with the decisions of per-graph dense runs == oracle.step_fwd_bwd_ragged."""
    torch.manual_seed(3)
    sd = O.init_state_dict(num_blocks=2)
    g = torch.Generator().manual_seed(4)
    sd = {k: (v + 0.1 * torch.randn(v.shape, generator=g) if k.endswith('.bias') and v.dim() == 1 else v) for k, v in sd.items()}
    from graph_neural_net_amd import synthetic
    xs, ys = synthetic.make_ragged_batch(31, 3, 5, 11)
    sizes = [int(t.shape[-1]) for t in xs]
    nmax = max(sizes)
    pad = lambda lst: torch.stack([torch.nn.functional.pad(t, (0, nmax - t.shape[-1], 0, nmax - t.shape[-1])) for t in lst])
    x1, x2 = pad(xs), pad(ys)
    B = len(sizes)
    masks = {}
    idx = torch.zeros(2 * B, 32, nmax, dtype=torch.int64)
    for b, (a, c) in enumerate(zip(xs, ys)):
        for gi, t in ((b, a), (B + b, c)):
            m, i = OP.collect_decisions(t.unsqueeze(0), sd)
            n = sizes[b]
            for k, v in m.items():
                masks.setdefault(k, torch.zeros(2 * B, 32, nmax, nmax, dtype=torch.bool))[gi, :, :n, :n] = v[0]
            idx[gi, :, :n] = i[0]
    s, l, gr = OP.step_fwd_bwd_pinned_ragged(x1, x2, sizes, sd, masks, idx, dtype=torch.float32)
    s0, l0, g0 = O.step_fwd_bwd_ragged(xs, ys, sd)
    assert abs(l.item() - l0.item()) <= 1e-6 * abs(l0.item())
    for a, b in zip(s, s0):
        assert rel(a, b) <= 1e-6
    for k in g0:
        assert rel(gr[k], g0[k]) <= 1e-5, (k, rel(gr[k], g0[k]))


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree not present')
def test_pinned_oracle_bit_equal_to_reference_live():
    code = r'''
import sys, os
import torch
sys.dont_write_bytecode = True
sys.path.insert(0, %r)
sys.path.insert(0, os.path.join(%r, 'tests', 'golden'))
import make_golden as mg
mg.import_reference()
from graph_neural_net_amd import synthetic
model = mg.build_reference_model(3, seed=8)
mg.perturb_(model, 88)
x1, x2 = synthetic.make_batch(78, 2, 19, 'ErdosRenyi', 0.3, 0.1)
mg.check_pinned_oracle(model, x1, x2, 'live fp32')
mg.check_pinned_oracle(mg.f64(model), x1, x2, 'live fp64')
print('PINNED')
''' % (ROOT, ROOT)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0 and 'PINNED' in out.stdout, out.stderr[-2000:]
