"""CPU: the bf16 oracle (oracle/fgnn_oracle_bf16.py).

(1) Structure: with every rounding switched off its hand-written backward must agree with the fp32 oracle's autograd
    (which is pinned bit-for-bit to the reference, tests/test_oracle_pinned.py / golden_meta.json).
(2) Error level: on the N=200 fixture generated FROM THE REFERENCE (fp32, fp64 and an all-bf16 reference run,
    tests/golden/make_golden.py round2) the rounding scheme of the HIP kernels (bf16 storage / operands, fp32 accumulation
    and statistics) must be in the same error class as the reference's own bf16 run: L2-relative distance to the fp64
    truth <= BF16_CLASS x the reference-bf16 distance, for the scores and for the whole flat gradient.  (Four blocks of
    16-bit activations are chaotic: over a handful of inputs the ratio ours / reference-bf16 was measured between 0.4
    and 1.5 in both directions -- scores 4-9 % vs 5-11 %, gradients 50-130 % vs 50-120 % -- hence the factor 2.)
"""
import torch

from oracle import fgnn_oracle as O, fgnn_oracle_bf16 as OB
from graph_neural_net_amd import synthetic
from util import BF16_CLASS, flat_of, is_zero_grad, l2rel, load_golden, rel, sub, unpack_pairs


def _perturbed_sd(num_blocks, seed):
    torch.manual_seed(seed)
    sd = O.init_state_dict(num_blocks=num_blocks)
    g = torch.Generator().manual_seed(seed + 1)
    for k, v in sd.items():
        if k.endswith('.bias') and v.dim() == 1:
            v.add_(0.1 * torch.randn(v.shape, generator=g))
        elif k.endswith('gn.weight'):
            v.mul_(1 + 0.2 * torch.randn(v.shape, generator=g))
        elif k.endswith('gn.bias'):
            v.add_(0.05 * torch.randn(v.shape, generator=g))
    return sd


def test_unrounded_scheme_equals_fp32_oracle():
    sd = _perturbed_sd(2, 0)
    x1, x2 = synthetic.make_batch(3, 3, 20, 'ErdosRenyi', 0.3, 0.1)
    s0, l0, g0 = O.step_fwd_bwd(x1, x2, sd)
    s1, l1, g1 = OB.step_fwd_bwd(x1, x2, sd, rounding=False)
    assert rel(s1, s0) < 1e-5
    assert abs(l1 - l0).item() < 1e-5 * abs(l0.item())
    for k in g0:
        if is_zero_grad(k):
            assert g1[k].abs().max() < 1e-4
        else:
            assert rel(g1[k], g0[k]) < 1e-4, (k, rel(g1[k], g0[k]))


def test_rounding_is_bf16_rne():
    x = torch.tensor([1.0, 1.00390625, 1.005859375, -3.14159, 1e-40, 65504.0])
    assert torch.equal(OB.rbf(x), x.to(torch.bfloat16).float())
    assert OB.rbf(torch.tensor([1.00390625])).item() == 1.0          # tie -> even
    assert OB.rbf(torch.tensor([1.01171875])).item() == 1.015625     # tie -> even (upwards)


def test_bf16_scheme_not_worse_than_reference_bf16_run():
    d = load_golden('cfg4_er_n200_b1_4blk.npz')
    sd = sub(load_golden('cfg2_reg_n50_b2_4blk.npz'), 'sd/')
    n = int(d['n'])
    x1, x2 = unpack_pairs(d['bits1'], n), unpack_pairs(d['bits2'], n)
    s, l, g = OB.step_fwd_bwd(x1, x2, sd)
    keys = [k for k in sub(d, 'grad/') if not is_zero_grad(k)]
    g64 = flat_of(sub(d, 'grad64/'), keys)
    s64 = d['scores64_as_f32']
    ours_s, ref_s = l2rel(s, s64), l2rel(d['scores_refbf16'], s64)
    ours_g = l2rel(flat_of(g, keys), g64)
    ref_g = l2rel(flat_of(sub(d, 'grad_refbf16/'), keys), g64)
    assert ours_s <= BF16_CLASS * ref_s, (ours_s, ref_s)
    assert ours_g <= BF16_CLASS * ref_g, (ours_g, ref_g)
    assert abs(l.item() - d['loss64'].item()) <= BF16_CLASS * abs(d['loss_refbf16'].item() - d['loss64'].item()) + 1e-3


ONE_BLOCK_FIXTURES = ('bf16ref_reg_n50_b2_1blk.npz', 'bf16ref_er_n200_b1_1blk.npz')


def one_block_gates(scores, grads, d):
    """Gates against the reference-generated ONE-block fixtures (tests/golden/make_golden.py round3: the reference itself run
    in bf16 -- Network.half, models/utils.py:71-74 -- next to its fp32 and fp64 runs), where 16-bit arithmetic is not yet
    chaotic.  Distances are L2-relative to the fp64 truth.  A scheme with fp32 accumulation and statistics should be at
    least as close as the all-bf16 reference run; two bf16 evaluations with different rounding points are still two samples
    of the same noise, so single tensors scatter around the ratio 1:
      * the whole flat gradient:            ours <= 1.0 x reference-bf16   (measured: oracle scheme 0.84 / 0.50)
      * per gradient tensor:  median ratio  <= 1.0, every tensor <= 3.0 x  (measured: medians 0.84 / 0.56, worst 1.83 / 1.26 for
                                            the oracle scheme, 0.71 / 2.17 on the N = 50 fixture for the HIP engine: a single
                                            tensor's ratio is the quotient of two noise samples)
      * scores:                             ours <= 1.5 x reference-bf16   (measured: 1.37 / 0.80; the pooled maxima of one
                                                                            block are a handful of bf16-rounded values)
    Returns the measured ratios."""
    keys = [k for k in sub(d, 'grad/') if not is_zero_grad(k)]
    g64, g16 = sub(d, 'grad64/'), sub(d, 'grad_refbf16/')
    s64 = d['scores64_as_f32']
    r_scores = l2rel(scores, s64) / l2rel(d['scores_refbf16'], s64)
    r_flat = l2rel(flat_of(grads, keys), flat_of(g64, keys)) / l2rel(flat_of(g16, keys), flat_of(g64, keys))
    per = sorted(l2rel(grads[k], g64[k]) / l2rel(g16[k], g64[k]) for k in keys)
    median = per[len(per) // 2]
    assert r_flat <= 1.0, r_flat
    assert median <= 1.0 and per[-1] <= 3.0, (median, per[-3:])
    assert r_scores <= 1.5, r_scores
    return r_scores, r_flat, median, per[-1]


def test_one_block_scheme_is_at_least_as_close_as_the_reference_bf16_run():
    for name in ONE_BLOCK_FIXTURES:
        d = load_golden(name)
        n = int(d['n'])
        x1, x2 = unpack_pairs(d['bits1'], n), unpack_pairs(d['bits2'], n)
        s, l, g = OB.step_fwd_bwd(x1, x2, sub(d, 'sd/'))
        one_block_gates(s, g, d)
        assert abs(l.item() - d['loss64'].item()) <= 1.0 * abs(d['loss_refbf16'].item() - d['loss64'].item()) + 2e-4
