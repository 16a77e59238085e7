"""The MaskedTensor protocol (maskedtensors/maskedtensor.py:98-112,189-384): an op that receives a MaskedTensor returns a
MaskedTensor with the same masks and exact zeros in the padding; "masked batch result == list of per-graph dense
results" (the reference's own test pattern, maskedtensors/test_maskedtensor.py:22-27,87-165, atol 1e-5, restated here).
The generic __torch_function__ path is pure torch and runs on the CPU as well; the same parametrisations run on the GPU
(-m gpu), where the fused HIP modules are compared too (tests/test_gpu_modules.py)."""
import pytest
import torch
import torch.nn as nn

from graph_neural_net_amd.masked import MaskedTensor, from_list, implements, SPECIAL_FUNCTIONS

N_FEATURES = 16
ATOL = 1e-5


def _tensor_list(dev, seed=0, vec=False):
    g = torch.Generator().manual_seed(seed)
    shape = (lambda n: (N_FEATURES, n)) if vec else (lambda n: (N_FEATURES, n, n))
    return [torch.empty(shape(n)).normal_(generator=g).to(dev) for n in range(40, 50)]


def _per_graph(lst, func):
    return [func(t.unsqueeze(0)).squeeze(0) for t in lst]


def _funcs(dev):
    torch.manual_seed(1)
    ln = nn.LayerNorm(N_FEATURES).to(dev)
    conv = nn.Conv2d(N_FEATURES, 2 * N_FEATURES, 1).to(dev)
    inorm = nn.InstanceNorm2d(N_FEATURES, affine=False, track_running_stats=False).to(dev)
    inorm_a = nn.InstanceNorm2d(N_FEATURES, affine=True, track_running_stats=False).to(dev)
    with torch.no_grad():
        inorm_a.weight.mul_(1.5)
        inorm_a.bias.add_(0.2)
    return [
        (lambda t: torch.add(t, 1), 'torch.add'),
        (lambda t: torch.mul(t, 2), 'torch.mul'),
        (lambda t: torch.sum(t, 2), 'torch.sum'),
        (lambda t: torch.max(t, 2)[0], 'torch.max(dim=2)'),
        (lambda t: torch.mean(t, dim=(-2, -1)), 'torch.mean'),
        (lambda t: torch.var(t, unbiased=False, dim=(-2, -1)), 'torch.var'),
        (lambda t: t.permute(0, 3, 2, 1), 'permute'),
        (conv, 'nn.Conv2d'),
        (lambda t: ln(t.permute(0, 3, 2, 1)), 'nn.LayerNorm'),
        (inorm, 'InstanceNorm2d'),
        (inorm_a, 'InstanceNorm2d_affine'),
        (lambda t: torch.diag_embed(t, dim1=-2, dim2=-1), 'torch.diag_embed'),
    ]


def _check_unary(dev):
    lst = _tensor_list(dev)
    for func, name in _funcs(dev):
        mt = from_list(lst, dims=(1, 2))
        res = func(mt)
        if isinstance(res, MaskedTensor):
            # padding is exactly zero after every op (maskedtensor.py:87-112)
            for i, n in enumerate(range(40, 50)):
                full = res.tensor[i]
                item = res[i]
                assert full.abs().sum() == item.abs().sum() or torch.allclose(full.abs().sum(), item.abs().sum()), name
            res = list(res)
        else:
            res = list(res)
        want = _per_graph(lst, func)
        for a, b in zip(res, want):
            assert a.size() == b.size(), (name, a.size(), b.size())
            assert torch.allclose(a, b, atol=ATOL), (name, (a - b).abs().max().item())


def _check_binary(dev):
    lst, other = _tensor_list(dev, 0), _tensor_list(dev, 1)
    cases = [
        (lambda a, b: torch.cat((a, b), dim=1), 'torch.cat', True),
        (lambda a, b: torch.stack((a, b), dim=1), 'torch.stack', True),
        (torch.matmul, 'torch.matmul', True),
        (torch.matmul, 'torch.matmul', False),
        (torch.add, 'Add', True),        # models/layers.py Add: torch.add(x1, x2)
    ]
    for func, name, same in cases:
        a = from_list(lst, dims=(1, 2))
        b = from_list(other, dims=(1, 2), base_name='N' if same else 'M')
        res = list(func(a, b))
        want = [func(x.unsqueeze(0), y.unsqueeze(0)).squeeze(0) for x, y in zip(lst, other)]
        for r, w in zip(res, want):
            assert r.size() == w.size(), (name, r.size(), w.size())
            assert torch.allclose(r, w, atol=ATOL), (name, (r - w).abs().max().item())


def _check_misc(dev):
    lst = _tensor_list(dev)
    mt = from_list(lst, dims=(1, 2))
    assert mt.names == ('B', None, 'N', 'N_') and mt.named_tensor.names == ('B', None, 'N', 'N_')
    assert set(mt.mask_dict) == {'N', 'N_'} and mt.mask_dict['N'].names == ('B', 'N')
    assert torch.equal(mt.mask_dict['N'].rename(None).sum(1).long().cpu(), torch.arange(40, 50))
    # global max (maskedtensor.py:213-217) and vector tensors (dims=(1,))
    assert torch.allclose(torch.max(mt), torch.stack([t.max() for t in lst]).max())
    vec = _tensor_list(dev, 2, vec=True)
    mv = from_list(vec, dims=(1,))
    for a, b in zip(list(torch.max(mv, dim=1)[0]), [t.max(0)[0] for t in vec]):
        assert torch.allclose(a, b, atol=ATOL)
    # a user-registered override takes precedence (the hook the fused kernels use in the reference: :189-200)
    calls = []

    @implements(torch.nn.functional.softplus)
    def _sp(x, *a, **k):
        calls.append(1)
        return x
    try:
        assert torch.nn.functional.softplus(mt) is mt and calls
    finally:
        del SPECIAL_FUNCTIONS[torch.nn.functional.softplus]
    # cross entropy / nll on padded score tensors go to the raw tensor (maskedtensor.py:378-384)
    sc = from_list([torch.randn(5, 5).to(dev), torch.randn(5, 5).to(dev)], dims=(0, 1))
    tgt = torch.arange(5, device=dev)
    assert torch.allclose(torch.nn.functional.cross_entropy(sc[0], tgt), torch.nn.functional.cross_entropy(sc.tensor[0], tgt))


def test_protocol_cpu():
    dev = torch.device('cpu')
    _check_unary(dev)
    _check_binary(dev)
    _check_misc(dev)


@pytest.mark.gpu
def test_protocol_gpu():
    dev = torch.device('cuda:0')
    _check_unary(dev)
    _check_binary(dev)
    _check_misc(dev)
