"""CPU: the multi-seed gradient fixture (tests/golden/gradgate_single_pairs.npz) and the gate built on it (tests/gradgate.py).
The gate must be satisfiable by the reference itself -- each of its two fp32 evaluations judged as if it were the engine -- and
must reject samples that are not of the reference's error class; the oracle's fp64 run reproduces the stored gradients."""
import numpy as np
import pytest
import torch

import gradgate as GG
from oracle import fgnn_oracle as O
from util import load_golden, sub, unpack_pairs


def test_fixture_shape_and_classes():
    groups = GG.load_groups()
    total = sum(len(g['n']) for g in groups.values())
    assert total >= 128 and len(groups['A']['n']) == 32 and set(groups['A']['n']) == {50}
    assert {1, 2, 3, 31, 33, 64, 65, 97} <= set(int(v) for v in groups['B']['n'])      # the shapes of test_degenerate_and_boundary_shapes
    deg, safe, near = GG.classes(groups['A'])
    assert near.all()                                  # N = 50, 4 blocks: every pair of the benchmarked batch is a near-tie case
    deg, safe, near = GG.classes(groups['B'])
    assert safe.sum() >= 24 and deg.sum() >= 4
    # in a safe case neither run of the reference flips a ReLU sign or a column-max index against fp64 ...
    g = groups['B']
    assert g['relu'][safe][:, :, 7:9].sum() == 0 and g['pool'][safe][:, :, 7:9].sum() == 0
    # ... and its two fp32 evaluations agree within 25 %
    r = g['err1'][safe] / g['err8'][safe]
    assert r.min() > 0.8 and r.max() < 1.25, (r.min(), r.max())
    # whereas on the benchmarked batch they differ by more than 3x on half of the pairs: the seed lottery, measured
    a = groups['A']
    assert ((a['err1'] > 3 * a['err8']) | (a['err8'] > 3 * a['err1'])).sum() >= 12


def test_the_reference_itself_passes_the_gate():
    groups = GG.load_groups()
    for pick, tpick in (('err1', 'terr1'), ('err8', 'terr8')):
        err = {t: np.where(g['gnorm64'] < 1e-10, 0.0, g[pick]) for t, g in groups.items()}
        out = GG.check(groups, err, {t: g[tpick] for t, g in groups.items()}, label='reference ' + pick)
        assert out['median_ratio_vs_8t'] <= 1.1 and out['median_ratio_vs_1t'] <= 1.1


def test_the_gate_rejects_another_error_class():
    groups = GG.load_groups()
    base = {t: np.where(g['gnorm64'] < 1e-10, 0.0, g['err8']) for t, g in groups.items()}
    # 3x worse on the safe cases only
    err = {t: np.where(GG.classes(g)[1], 3.0 * base[t], base[t]) for t, g in groups.items()}
    with pytest.raises(AssertionError, match='safe cases'):
        GG.check(groups, err)
    # a tf32-class evaluation: 1e-3 everywhere
    err = {t: np.where(g['gnorm64'] < 1e-10, 0.0, 1e-3) for t, g in groups.items()}
    with pytest.raises(AssertionError):
        GG.check(groups, err)
    # flips on most near-tie cases: 4x the reference on 60 % of them
    rng = np.random.default_rng(0)
    err = {t: np.where(GG.classes(g)[2] & (rng.random(len(g['n'])) < 0.6), 4.0 * np.maximum(g['err8'], g['err1']), base[t])
           for t, g in groups.items()}
    with pytest.raises(AssertionError, match='outliers|median'):
        GG.check(groups, err)
    # one broken pair
    err = {t: b.copy() for t, b in base.items()}
    err['A'][7] = 5e-2
    with pytest.raises(AssertionError, match='near-tie case beyond'):
        GG.check(groups, err)


def test_oracle_fp64_reproduces_the_stored_gradients():
    """The fixture stores the reference's fp64 gradient rounded to fp32; the pinned oracle evaluated here in fp64 must land on
    it (3e-8 L2 = the rounding), group A and group B weights."""
    groups = GG.load_groups()
    sdA = sub(load_golden('cfg2_reg_n50_b2_4blk.npz'), 'sd/')
    for tag, sd, picks in (('A', sdA, (0, 17)), ('B', {k: torch.from_numpy(v) for k, v in groups['B']['sd'].items()}, (30, 60, 100))):
        g = groups[tag]
        names = list(sd.keys())
        sd64 = {k: v.double() for k, v in sd.items()}
        for i in picks:
            n = int(g['n'][i])
            w = (n + 31) // 32
            x1 = unpack_pairs(torch.from_numpy(g['bits'][i, 0:1, :n, :w].copy()), n)
            x2 = unpack_pairs(torch.from_numpy(g['bits'][i, 1:2, :n, :w].copy()), n)
            _, _, g64 = O.step_fwd_bwd(x1.double(), x2.double(), sd64)
            flat = torch.cat([g64[k].reshape(-1) for k in names])
            ref = torch.from_numpy(g['g64'][i]).double()
            assert flat.numel() == ref.numel()
            assert ((flat - ref).norm() / ref.norm()).item() < 2e-7, (tag, i)
