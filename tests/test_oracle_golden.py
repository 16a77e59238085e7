"""CPU: the oracle reproduces the golden vectors the reference itself produced."""
import torch

from oracle import fgnn_oracle as O
from util import is_zero_grad, load_golden, rel, sub

FWD_TOL = 5e-6    # same op sequence; only CPU ISA / thread-count reassociation can differ
GRAD_TOL = 2e-4


def _check_case(name):
    d = load_golden(name)
    sd = sub(d, 'sd/')
    scores, loss, grads = O.step_fwd_bwd(d['x1'], d['x2'], sd)
    assert rel(scores, d['scores']) < FWD_TOL
    assert abs(loss.item() - d['loss'].item()) < 1e-6 * abs(d['loss'].item()) + 1e-7
    keep = {}
    O.node_embedding(d['x1'], sd, keep)
    for k, v in sub(d, 'inter/').items():
        assert rel(keep[k][:v.shape[0]], v) < FWD_TOL, k
    for k, v in sub(d, 'grad/').items():
        if is_zero_grad(k):
            assert grads[k].abs().max() < 1e-4
        else:
            assert rel(grads[k], v) < GRAD_TOL, k


def test_cfg1_golden():
    _check_case('cfg1_er_n20_b4_1blk.npz')


def test_cfg2_golden():
    _check_case('cfg2_reg_n50_b2_4blk.npz')


def test_widths_golden():
    """channel widths other than 32/32/2 (3 -> 16 -> 48, depth 2): constant-size and ragged reference runs"""
    d = load_golden('widths_c3_16_48_d2_2blk.npz')
    sd = sub(d, 'sd/')
    scores, loss, grads = O.step_fwd_bwd(d['x1'], d['x2'], sd)
    assert rel(scores, d['scores']) < FWD_TOL
    assert abs(loss.item() - d['loss'].item()) < 1e-6 * abs(d['loss'].item()) + 1e-7
    for k, v in sub(d, 'grad/').items():
        if is_zero_grad(k, 2):
            assert grads[k].abs().max() < 1e-4
        else:
            assert rel(grads[k], v) < GRAD_TOL, k
    n = len(d['ragged/ns'])
    xs = [d['ragged/x1/%d' % i] for i in range(n)]
    ys = [d['ragged/x2/%d' % i] for i in range(n)]
    scores, loss, grads = O.step_fwd_bwd_ragged(xs, ys, sd)
    for i in range(n):
        assert rel(scores[i], d['ragged/scores/%d' % i]) < FWD_TOL
    assert abs(loss.item() - d['ragged/loss'].item()) < 1e-6
    for k, v in sub(d, 'ragged/grad/').items():
        if not is_zero_grad(k, 2):
            assert rel(grads[k], v) < GRAD_TOL, k


def test_wide64_golden():
    """the 64-feature model (2 -> 64 -> 64, depth 3, 4 blocks: the widths of csrc/mlp64.hip): constant-size regular pairs at N = 50 and a
    ragged batch, reference fp32 runs"""
    d = load_golden('wide64_c2_64_64_d3_4blk.npz')
    sd = sub(d, 'sd/')
    scores, loss, grads = O.step_fwd_bwd(d['x1'], d['x2'], sd)
    assert rel(scores, d['scores']) < FWD_TOL
    assert abs(loss.item() - d['loss'].item()) < 1e-6 * abs(d['loss'].item()) + 1e-7
    for k, v in sub(d, 'grad/').items():
        if is_zero_grad(k, 3):
            assert grads[k].abs().max() < 1e-4
        else:
            assert rel(grads[k], v) < GRAD_TOL, k
    n = len(d['ragged/ns'])
    xs = [d['ragged/x1/%d' % i] for i in range(n)]
    ys = [d['ragged/x2/%d' % i] for i in range(n)]
    scores, loss, grads = O.step_fwd_bwd_ragged(xs, ys, sd)
    for i in range(n):
        assert rel(scores[i], d['ragged/scores/%d' % i]) < FWD_TOL
    assert abs(loss.item() - d['ragged/loss'].item()) < 1e-6
    for k, v in sub(d, 'ragged/grad/').items():
        if not is_zero_grad(k, 3):
            assert rel(grads[k], v) < GRAD_TOL, k


def test_narrow_widths_golden():
    """3 -> 16 -> 24, depth 2, 3 blocks (the widths the fused engine runs zero-padded)"""
    d = load_golden('widths_c3_16_24_d2_3blk.npz')
    sd = sub(d, 'sd/')
    scores, loss, grads = O.step_fwd_bwd(d['x1'], d['x2'], sd)
    assert rel(scores, d['scores']) < FWD_TOL
    assert abs(loss.item() - d['loss'].item()) < 1e-6 * abs(d['loss'].item()) + 1e-7
    for k, v in sub(d, 'grad/').items():
        if is_zero_grad(k, 2):
            assert grads[k].abs().max() < 1e-4
        else:
            assert rel(grads[k], v) < GRAD_TOL, k


def test_ragged_golden():
    d = load_golden('ragged_er_b4_2blk.npz')
    sd = sub(d, 'sd/')
    n = len(d['ns'])
    xs = [d['x1/%d' % i] for i in range(n)]
    ys = [d['x2/%d' % i] for i in range(n)]
    scores, loss, grads = O.step_fwd_bwd_ragged(xs, ys, sd)
    for i in range(n):
        assert rel(scores[i], d['scores/%d' % i]) < FWD_TOL
        # the reference's own MaskedTensor branch agrees with per-graph dense runs
        assert torch.allclose(d['e1_masked/%d' % i], d['e1/%d' % i], atol=1e-5)
    assert abs(loss.item() - d['loss'].item()) < 1e-6
    for k, v in sub(d, 'grad/').items():
        if not is_zero_grad(k):
            assert rel(grads[k], v) < GRAD_TOL, k


def test_layers_golden():
    d = load_golden('layers_16to32_depth2.npz')
    msd, gsd = sub(d, 'mlp_sd/'), sub(d, 'gn_sd/')
    for i in range(3):
        x = d['x/%d' % i].unsqueeze(0)
        y = O.mlp_block_real(x, [msd['convs.0.weight'], msd['convs.1.weight']],
                             [msd['convs.0.bias'], msd['convs.1.bias']], msd['gn.weight'], msd['gn.bias'])
        assert rel(y.squeeze(0), d['mlp/%d' % i]) < FWD_TOL
        assert rel(O.graph_norm(x, gsd['weight'], gsd['bias']).squeeze(0), d['gn/%d' % i]) < FWD_TOL
        assert rel(O.normalize(x).squeeze(0), d['normalize/%d' % i]) < FWD_TOL


def test_pad_graph_list_bit_exact():
    xs = [torch.randn(2, n, n) for n in (3, 5, 4)]
    data, ns = O.pad_graph_list(xs)
    assert data.shape == (3, 2, 5, 5) and ns.tolist() == [3, 5, 4]
    for i, x in enumerate(xs):
        n = x.shape[-1]
        assert torch.equal(data[i, :, :n, :n], x)
        assert data[i, :, n:, :].abs().sum() == 0 and data[i, :, :, n:].abs().sum() == 0


def test_flop_model():
    # 1.3349 GFLOP per pair fwd+bwd at N=50
    assert abs(O.algorithmic_flops_per_pair(50) / 1e9 - 1.3349) < 1e-3


def test_input_contract_pinned_to_reference():
    """synthetic.tensor_representation / noise_erdos_renyi arithmetic == the reference's
    adjacency_matrix_to_tensor_representation / noise_erdos_renyi (loaders/data_generator.py:118-125, 79-87) on the
    fixture generated by calling the reference functions (tests/golden/make_golden.py round2)."""
    import numpy as np
    from graph_neural_net_amd import synthetic
    d = load_golden('input_contract.npz')
    idx = sorted({int(k.split('/')[1]) for k in d if k.startswith('w/')})
    assert len(idx) >= 3
    for i in idx:
        w = d['w/%d' % i].numpy()
        assert torch.equal(torch.from_numpy(synthetic.tensor_representation(w)), d['repr/%d' % i])
        z1, z2 = d['z1/%d' % i].numpy(), d['z2/%d' % i].numpy()
        assert torch.equal(torch.from_numpy((w * (1.0 - z1) + (1.0 - w) * z2).astype(np.float32)), d['w_noise/%d' % i])
        pe1, pe2, p = [float(v) for v in d['pe/%d' % i]]
        assert pe1 == 0.1 and abs(pe2 - p * 0.1 / (1 - p)) < 1e-15          # the probabilities the reference asks for

        class _Rng:                                                       # replay the fixture's noise graphs
            def __init__(self):
                self.q = [z1, z2]

        # synthetic.noise_erdos_renyi draws z1 then z2 with exactly these probabilities
        calls = []
        orig = synthetic.erdos_renyi
        synthetic.erdos_renyi = lambda rng, n, pp: (calls.append(pp), [z1, z2][len(calls) - 1])[1]
        try:
            wn = synthetic.noise_erdos_renyi(None, w, 0.1, p)
        finally:
            synthetic.erdos_renyi = orig
        assert calls[0] == 0.1 and abs(calls[1] - pe2) < 1e-15
        assert torch.equal(torch.from_numpy(wn.astype(np.float32)), d['w_noise/%d' % i])


def test_loss_reductions_against_reference_values():
    """triplet_loss 'mean' and 'mean_of_mean' of the reference on the cfg1 scores (toolbox/losses.py:8-34)."""
    d = load_golden('cfg1_er_n20_b4_1blk.npz')
    v = load_golden('losses_cfg1.npz')
    assert torch.equal(O.triplet_loss_mean(d['scores']), v['mean'])
    assert torch.equal(O.triplet_loss_mean_of_mean(d['scores']), v['mean_of_mean'])
