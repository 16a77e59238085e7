"""GPU: ONE gradient gate for every engine variant on the reference-generated multi-seed fixture of
148 single-pair cases -- tests/gradgate.py states the classes and the gates, tests/test_gradgate_fixture.py shows on the CPU
that the reference's own two fp32 evaluations pass them and that other error classes do not.  Applied identically to
mfma='f32' (v_mfma_f32_32x32x2_f32) and mfma='x3' (bf16 matrix cores through the exact operand split).  Forward: scores
against the fp64 scores of the fixture, <= max(3e-5, 2 x the reference's own fp32 error) on every case."""
import numpy as np
import pytest
import torch

import gradgate as GG
from graph_neural_net_amd.engine import FgnnEngine, ParamLayout
from util import load_golden, sub, unpack_pairs

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
E2E_FWD_TOL = 3e-5


def run_cases(mode, groups=None, pair_bwd=None):
    """-> (err, terr, score_err) dicts per group for the engine variant `mode`: 'f32' / 'x3' = the contraction of the MLP kernels
    on dense inputs; a trailing 's' ('f32s', 'x3s') = bit-packed inputs with block 1 on its structured form
    (csrc/block1_struct.hip; N <= 256)."""
    struct = mode.endswith('s')
    mfma = mode[:-1] if struct else mode
    groups = GG.load_groups() if groups is None else groups
    sds = {'A': sub(load_golden('cfg2_reg_n50_b2_4blk.npz'), 'sd/'),
           'B': {k: torch.from_numpy(v) for k, v in groups['B']['sd'].items()}}
    err, terr, serr = {}, {}, {}
    for tag, g in groups.items():
        nblk = 4 if tag == 'A' else 2
        lay = ParamLayout(2, nblk, 32, 32, 3)
        params = lay.flatten(sds[tag], DEV)
        keep = torch.ones(lay.total, dtype=torch.bool)
        for name, off, shape in lay.entries:
            if name.endswith(GG.ZERO_GRAD_SUFFIX):
                keep[off:off + int(np.prod(shape))] = False
        engines = {}
        e_all, t_all, s_all = [], [], []
        for i in range(len(g['n'])):
            n = int(g['n'][i])
            w = (n + 31) // 32
            x1 = unpack_pairs(torch.from_numpy(g['bits'][i, 0:1, :n, :w].copy()), n)
            x2 = unpack_pairs(torch.from_numpy(g['bits'][i, 1:2, :n, :w].copy()), n)
            if n not in engines:
                engines[n] = FgnnEngine(lay, 2, n, DEV, mfma=mfma, block1='structured' if struct else 'generic')
                if pair_bwd is not None:
                    engines[n].PAIR_BWD = pair_bwd
            eng = engines[n]
            grads = torch.zeros_like(params)
            if struct:
                packed = torch.from_numpy(np.ascontiguousarray(g['bits'][i, :, :n, :w]).view(np.int32)).to(DEV)
                scores, loss = eng.step(params, grads, None, bits=packed)
            else:
                scores, loss = eng.step(params, grads, torch.cat([x1, x2]).contiguous().to(DEV))
            torch.cuda.synchronize()
            got = grads.cpu().double()
            ref = torch.from_numpy(g['g64'][i]).double()
            assert torch.isfinite(got).all() and torch.isfinite(scores).all()
            assert got[~keep].abs().max() < GG.ZERO_GRAD_ABS                     # analytically zero last-conv bias gradients
            if g['gnorm64'][i] < 1e-10:
                e_all.append((got - ref)[keep].abs().max().item())
            else:
                e_all.append(((got - ref)[keep].norm() / ref[keep].norm()).item())
            trow = []
            for name, off, shape in lay.entries:
                cnt = int(np.prod(shape))
                a, b = got[off:off + cnt], ref[off:off + cnt]
                s = b.abs().max().item()
                trow.append(0.0 if name.endswith(GG.ZERO_GRAD_SUFFIX) else ((a - b).abs().max().item() / s if s > 0 else (a - b).abs().max().item()))
            t_all.append(trow)
            s64 = torch.from_numpy(g['scores64'][i, :n, :n]).double()
            sc = scores.cpu().double()[0]
            s_all.append(((sc - s64).abs().max() / s64.abs().max()).item() if s64.abs().max() > 0 else (sc - s64).abs().max().item())
            assert abs(loss.item() - g['loss64'][i]) <= 1e-5 * abs(g['loss64'][i]) + 1e-7, (tag, i, loss.item(), g['loss64'][i])
        err[tag], terr[tag], serr[tag] = np.array(e_all), np.array(t_all), np.array(s_all)
    return err, terr, serr


@pytest.mark.parametrize('mode', ['f32', 'x3', 'f32s', 'x3s'])
def test_gradient_gate(mode):
    groups = GG.load_groups()
    err, terr, serr = run_cases(mode, groups)
    for tag, g in groups.items():             # forward: every case
        yard = np.maximum(g['score_err8'], g['score_err1'])
        bad = np.nonzero(serr[tag] > np.maximum(E2E_FWD_TOL, 2.0 * yard))[0]
        assert bad.size == 0, (mode, tag, [(int(g['n'][i]), serr[tag][i], yard[i]) for i in bad])
    out = GG.check(groups, err, terr, label=mode)
    print(mode, out)
