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
