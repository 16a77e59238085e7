"""GPU: channel widths outside the fused 32-wide kernels (models/layers.py:113-123 takes any in/out features):
the generic 1x1-conv kernels of csrc/conv.hip through the C ABI, and the module surface built on them against
reference-generated goldens (tests/golden/make_golden.py widths) with the reference's own fp32-vs-fp64 error as
the yard-stick."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from graph_neural_net_amd import _lib
from graph_neural_net_amd.layers import MlpBlock_Real
from graph_neural_net_amd.masked import from_list
from graph_neural_net_amd.siamese import Siamese_Node_Exp
from oracle import fgnn_oracle as O
from util import is_zero_grad, load_golden, rel, sub

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
NE = dict(type='node_embedding', block_init='block_emb', block_inside='block', num_blocks=2, in_features=16,
          out_features=48, depth_of_mlp=2)


def _valid_mask(G, N, nv):
    m = torch.zeros(G, 1, N, N, dtype=torch.float64)
    for g in range(G):
        n = N if nv is None else int(nv[g])
        m[g, :, :n, :n] = 1
    return m


@pytest.mark.parametrize('K,M', [(3, 16), (19, 16), (64, 48), (33, 70), (1, 1), (256, 40)])
@pytest.mark.parametrize('N,ragged', [(7, False), (14, True), (45, True)])
def test_conv1x1_entry_points(K, M, N, ragged):
    """fgnn_conv1x1 (forward and input gradient) and fgnn_conv1x1_dw + fgnn_reduce_partials against an fp64 ATen
    conv + ReLU and its autograd gradients; padding of ragged graphs is exact zero."""
    G = 3
    g = torch.Generator().manual_seed(K * 1000 + M * 10 + N)
    nv = torch.tensor([N, max(1, N // 2), N - 1], dtype=torch.int32) if ragged else None
    mask = _valid_mask(G, N, nv)
    x = (torch.randn(G, K, N, N, generator=g).double() * mask).float()
    w = torch.randn(M, K, generator=g) / K ** 0.5
    b = 0.3 * torch.randn(M, generator=g)
    dy = (torch.randn(G, M, N, N, generator=g).double() * mask).float()
    # fp64 reference with the MaskedTensor semantics: conv, activation, re-mask
    xr = x.double().requires_grad_(True)
    wr, br = w.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = F.relu(F.conv2d(xr, wr[:, :, None, None], br)) * mask
    yr.backward(dy.double())
    # the same in fp32 on the CPU: the yard-stick for the fp32 rounding level
    y32 = F.relu(F.conv2d(x, w[:, :, None, None], b)) * mask.float()

    P = N * N
    xd, wd, bd, dyd = x.to(DEV), w.to(DEV), b.to(DEV), dy.to(DEV)
    nvd = nv.to(DEV) if ragged else None
    nvp = _lib.ptr(nvd) if ragged else None
    st = _lib.stream_ptr()
    y = torch.full((G, M, N, N), float('nan'), device=DEV)
    _lib.call('fgnn_conv1x1', _lib.ptr(xd), K * P, P, None, _lib.ptr(wd), K, 1, _lib.ptr(bd), 1, nvp, G, N, M, K,
              _lib.ptr(y), M * P, P, st)
    assert torch.isfinite(y).all()
    assert rel(y.cpu(), yr.detach()) < 4 * rel(y32, yr.detach()) + 1e-6
    assert (y.cpu().double() * (1 - mask)).abs().max() == 0
    dx = torch.full((G, K, N, N), float('nan'), device=DEV)
    _lib.call('fgnn_conv1x1', _lib.ptr(dyd), M * P, P, _lib.ptr(y), _lib.ptr(wd), 1, K, None, 0, nvp, G, N, K, M,
              _lib.ptr(dx), K * P, P, st)
    assert rel(dx.cpu(), xr.grad) < 1e-5
    assert (dx.cpu().double() * (1 - mask)).abs().max() == 0
    chunks = _lib.load().fgnn_conv1x1_dw_chunks(G, N)
    cnt = M * K + M
    wpart = torch.full((chunks * cnt,), float('nan'), device=DEV)
    flat = torch.empty(cnt, device=DEV)
    _lib.call('fgnn_conv1x1_dw', _lib.ptr(dyd), M * P, P, _lib.ptr(y), _lib.ptr(xd), K * P, P, nvp, G, N, M, K,
              _lib.ptr(wpart), st)
    _lib.call('fgnn_reduce_partials', _lib.ptr(wpart), chunks, cnt, _lib.ptr(flat), st)
    assert rel(flat[:M * K].view(M, K).cpu(), wr.grad) < 1e-5
    assert rel(flat[M * K:].cpu(), br.grad) < 1e-5
    # bit-reproducible
    flat2 = torch.empty(cnt, device=DEV)
    _lib.call('fgnn_conv1x1_dw', _lib.ptr(dyd), M * P, P, _lib.ptr(y), _lib.ptr(xd), K * P, P, nvp, G, N, M, K,
              _lib.ptr(wpart), st)
    _lib.call('fgnn_reduce_partials', _lib.ptr(wpart), chunks, cnt, _lib.ptr(flat2), st)
    assert torch.equal(flat, flat2)


def test_conv1x1_rejects_bad_arguments():
    x = torch.zeros(1, 4, 5, 5, device=DEV)
    w = torch.zeros(300, 4, device=DEV)
    y = torch.zeros(1, 300, 5, 5, device=DEV)
    with pytest.raises(RuntimeError, match='at most'):
        _lib.call('fgnn_conv1x1', _lib.ptr(x), 100, 25, None, _lib.ptr(w), 4, 1, None, 0, None, 1, 5, 300, 4,
                  _lib.ptr(y), 300 * 25, 25, _lib.stream_ptr())
    with pytest.raises(RuntimeError, match='strides'):
        _lib.call('fgnn_conv1x1', _lib.ptr(x), 100, 24, None, _lib.ptr(w), 4, 1, None, 0, None, 1, 5, 8, 4,
                  _lib.ptr(y), 8 * 25, 25, _lib.stream_ptr())


@pytest.mark.parametrize('cin,cout,depth', [(3, 16, 2), (19, 16, 2), (64, 48, 1), (48, 48, 3), (32, 64, 3), (5, 32, 3),
                                            # chains of ONE-group (32-wide) or THREE-group (96-wide) layers whose first K is 8 / 32 / 64 /
                                            # 128: the shapes the conv-chain's all-full instantiation was wrongly chosen for until round 5
                                            # (forward 128 -> 32 and 8 -> 32; the split input-gradient chain of every cin > 64 -> 32)
                                            (104, 32, 3), (72, 32, 3), (65, 32, 2), (128, 32, 2), (128, 32, 3), (8, 32, 2), (8, 32, 3),
                                            (64, 96, 2), (128, 96, 3), (100, 96, 3), (100, 64, 3), (128, 128, 2), (130, 128, 3)])
def test_mlp_block_any_width_matches_oracle(cin, cout, depth):
    """MlpBlock_Real on widths the fused kernels are not built for: forward + every gradient against the oracle run
    in fp64, with the fp32 oracle's own error as the yard-stick; masked batch == per-graph dense results."""
    torch.manual_seed(cin * 100 + cout)
    mlp = MlpBlock_Real(cin, cout, depth).to(DEV)
    assert not mlp.fused()
    with torch.no_grad():
        mlp.gn.weight.mul_(1.2)
        mlp.gn.bias.add_(0.1)
        for c in mlp.convs:
            c.bias.add_(0.1 * torch.randn_like(c.bias))
    sizes = (9, 12, 10)
    lst = [torch.randn(cin, n, n) for n in sizes]
    douts = [torch.randn(cout, n, n) for n in sizes]
    ws = [c.weight.detach().cpu() for c in mlp.convs]
    bs = [c.bias.detach().cpu() for c in mlp.convs]
    gw, gb = mlp.gn.weight.detach().cpu(), mlp.gn.bias.detach().cpu()

    def run(dt):
        ps = [t.to(dt).requires_grad_(True) for t in ws + bs + [gw, gb]]
        xs = [t.to(dt).requires_grad_(True) for t in lst]
        ys = [O.mlp_block_real(x.unsqueeze(0), ps[:depth], ps[depth:2 * depth], ps[-2], ps[-1]).squeeze(0) for x in xs]
        sum((y * d.to(dt)).sum() for y, d in zip(ys, douts)).backward()
        return ys, xs, ps

    y64, x64, p64 = run(torch.float64)
    y32, x32, p32 = run(torch.float32)
    xm = from_list([t.to(DEV) for t in lst], dims=(1, 2), base_name='N')
    xt = xm.tensor.detach().requires_grad_(True)
    from graph_neural_net_amd.masked import MaskedTensor
    out = mlp(MaskedTensor(xt, xm.nvalid, xm.masked_dims, xm.base_name))
    dpad = torch.zeros_like(out.tensor)
    for i, (d, n) in enumerate(zip(douts, sizes)):
        dpad[i, :, :n, :n] = d.to(DEV)
    (out.tensor * dpad).sum().backward()
    for i, n in enumerate(sizes):
        assert rel(out.tensor[i, :, :n, :n].cpu(), y64[i]) < 4 * rel(y32[i], y64[i]) + 1e-6
        assert rel(xt.grad[i, :, :n, :n].cpu(), x64[i].grad) < 4 * rel(x32[i].grad, x64[i].grad) + 1e-5
        pad = out.tensor[i].clone()
        pad[:, :n, :n] = 0
        assert pad.abs().max() == 0
    mine = [c.weight.grad for c in mlp.convs] + [c.bias.grad for c in mlp.convs] + [mlp.gn.weight.grad, mlp.gn.bias.grad]
    for j, (a, b64, b32) in enumerate(zip(mine, p64, p32)):
        if j == 2 * depth - 1:          # last conv bias: analytically zero gradient (GraphNorm removes the mean)
            assert a.abs().max() < 1e-4
            continue
        b64g, b32g = b64.grad.reshape(a.shape), b32.grad.reshape(a.shape)
        assert rel(a.cpu(), b64g) < 4 * rel(b32g, b64g) + 1e-5, j


def _load_model(d, **kw):
    model = Siamese_Node_Exp(3, dict(NE, **kw)).to(DEV)
    model.load_state_dict({'node_embedder.' + k: v for k, v in sub(d, 'sd/').items()})
    return model


def test_siamese_any_width_against_reference_golden():
    """original_features_num 3, in_features 16, out_features 48, depth 2 (convs 3->16, 19->16, 16->48, 64->48): scores,
    loss, intermediates and every gradient against the reference's own fp32 / fp64 runs."""
    d = load_golden('widths_c3_16_48_d2_2blk.npz')
    model = _load_model(d)
    assert model.node_embedder._standard_layout() is None          # not the fused engine: the per-layer modules
    out = model.node_embedder({'input': d['x1'].to(DEV)})
    for k, v in sub(d, 'inter/').items():
        assert rel(out[k].detach().cpu(), v) < 2e-5, k
    scores = model({'input': d['x1'].to(DEV)}, {'input': d['x2'].to(DEV)})
    loss = model.loss(scores)
    loss.backward()
    yard = rel(d['scores'], d['scores64'])
    assert rel(scores.detach().cpu(), d['scores64']) < max(2 * yard, 1e-5)
    assert abs(loss.item() - d['loss64'].item()) < 1e-5 * d['loss64'].item()
    for n, p in model.named_parameters():
        k = n[len('node_embedder.'):]
        if is_zero_grad(k, 2):
            assert p.grad.abs().max() < 1e-4
        else:
            yard = rel(d['grad/' + k], d['grad64/' + k])
            assert rel(p.grad.cpu(), d['grad64/' + k]) < 4 * yard + 1e-5, k
    # a second step accumulates into .grad like any autograd module
    model.loss(model({'input': d['x1'].to(DEV)}, {'input': d['x2'].to(DEV)})).backward()
    k = 'ne_bm_block2_mlp3.convs.0.weight'
    assert rel(getattr(model.node_embedder, 'ne_bm_block2_mlp3').convs[0].weight.grad.cpu(), 2 * d['grad64/' + k]) < 1e-4


def test_siamese_any_width_ragged_against_reference_golden():
    d = load_golden('widths_c3_16_48_d2_2blk.npz')
    model = _load_model(d, constant_n_vertices=False)
    n = len(d['ragged/ns'])
    m1 = from_list([d['ragged/x1/%d' % i].to(DEV) for i in range(n)], dims=(1, 2), base_name='N')
    m2 = from_list([d['ragged/x2/%d' % i].to(DEV) for i in range(n)], dims=(1, 2), base_name='M')
    scores = model(m1, m2)
    loss = model.loss(scores)
    loss.backward()
    for i, a in enumerate(list(scores)):
        ref64 = d['ragged/scores64/%d' % i]
        assert a.shape == ref64.shape
        assert rel(a.detach().cpu(), ref64) < max(2 * rel(d['ragged/scores/%d' % i], ref64), 1e-5)
    assert abs(loss.item() - d['ragged/loss64'].item()) < 1e-5 * d['ragged/loss64'].item()
    for name, p in model.named_parameters():
        k = name[len('node_embedder.'):]
        if not is_zero_grad(k, 2):
            yard = rel(d['ragged/grad/' + k], d['ragged/grad64/' + k])
            assert rel(p.grad.cpu(), d['ragged/grad64/' + k]) < 4 * yard + 1e-5, k


def test_any_width_trains_and_bf16_is_refused():
    d = load_golden('widths_c3_16_48_d2_2blk.npz')
    model = _load_model(d)
    batch = ({'input': d['x1'].to(DEV)}, {'input': d['x2'].to(DEV)})
    opt = model.configure_optimizers()['optimizer']
    first = None
    for _ in range(40):
        opt.zero_grad()
        loss = model.training_step(batch, 0)
        first = first if first is not None else loss.item()
        loss.backward()
        opt.step()
    assert loss.item() < 0.8 * first
    with pytest.raises(RuntimeError, match='bf16'):
        Siamese_Node_Exp(3, NE, precision='bf16').to(DEV)({'input': d['x1'].to(DEV)}, {'input': d['x2'].to(DEV)})


def test_narrow_widths_run_zero_padded_on_the_fused_engine():
    """original_features_num 3, in_features 16, out_features 24, depth 2, 3 blocks: every width <= 32, so Network embeds the
    model in the 32-wide fused engine (zero-padded parameters / input).  Scores, loss and every gradient against the
    reference's own fp32 / fp64 runs; .grad accumulation; agreement with the per-layer module path."""
    d = load_golden('widths_c3_16_24_d2_3blk.npz')
    ne = dict(NE, num_blocks=3, in_features=16, out_features=24, depth_of_mlp=2)
    model = Siamese_Node_Exp(3, ne).to(DEV)
    model.load_state_dict({'node_embedder.' + k: v for k, v in sub(d, 'sd/').items()})
    net = model.node_embedder
    assert net._standard_layout() is not None and net._pad is not None and net._pad['c0p'] == 32
    b1, b2 = {'input': d['x1'].to(DEV)}, {'input': d['x2'].to(DEV)}
    scores = model(b1, b2)
    assert scores.shape == d['scores'].shape
    loss = model.loss(scores)
    loss.backward()
    assert rel(scores.detach().cpu(), d['scores64']) < max(2 * rel(d['scores'], d['scores64']), 1e-5)
    assert abs(loss.item() - d['loss64'].item()) < 1e-5 * d['loss64'].item()
    fused = {}
    for n, p in model.named_parameters():
        k = n[len('node_embedder.'):]
        assert p.grad is not None and p.grad.shape == p.shape
        fused[k] = p.grad.clone()
        if is_zero_grad(k, 2):
            assert p.grad.abs().max() < 1e-4
        else:
            yard = rel(d['grad/' + k], d['grad64/' + k])
            assert rel(p.grad.cpu(), d['grad64/' + k]) < 4 * yard + 1e-5, k
    model.loss(model(b1, b2)).backward()                       # accumulates
    k = 'ne_bm_block3_mlp3.convs.0.weight'
    assert rel(net.ne_bm_block3_mlp3.convs[0].weight.grad, 2 * fused[k]) < 1e-6
    with torch.no_grad():                                       # evaluation forward
        assert torch.equal(model(b1, b2), scores.detach())
    # the per-layer module path (conv.hip) on the same weights
    ref = Siamese_Node_Exp(3, ne).to(DEV)
    ref.load_state_dict(model.state_dict())
    ref.node_embedder._layout = False
    s2 = ref(b1, b2)
    ref.loss(s2).backward()
    assert rel(s2.detach(), scores.detach()) < 2e-5
    for (n, p), (_, q) in zip(ref.named_parameters(), model.named_parameters()):
        if not is_zero_grad(n, 2):
            assert rel(p.grad, fused[n[len('node_embedder.'):]]) < 2e-3, n


@pytest.mark.parametrize('c0,width,depth,ragged', [(1, 8, 3, False), (2, 16, 3, True), (5, 32, 1, True), (32, 20, 2, False),
                                                   # the un-padded engine at depths 1 and 2 (the 2 / 32+2 channel kernels)
                                                   (2, 32, 2, False), (2, 32, 1, True), (32, 32, 2, True)])
def test_padded_engine_matches_oracle(c0, width, depth, ragged):
    """more narrow configurations (1- / 2-channel inputs keep the 2-channel first block, others pad the input to 32 channels),
    constant-size and ragged (MaskedTensor) batches, against the oracle in fp64 with its fp32 run as the yard-stick"""
    torch.manual_seed(c0 * 100 + width)
    sd = O.init_state_dict(original_features_num=c0, num_blocks=2, in_features=width, out_features=width, depth_of_mlp=depth)
    g = torch.Generator().manual_seed(c0)
    sizes = (9, 14, 11) if ragged else (12, 12, 12)
    xs = [torch.randn(c0, n, n, generator=g) for n in sizes]
    ys = [torch.randn(c0, n, n, generator=g) for n in sizes]
    s32, l32, g32 = O.step_fwd_bwd_ragged(xs, ys, sd)
    s64, l64, g64 = O.step_fwd_bwd_ragged([t.double() for t in xs], [t.double() for t in ys], {k: v.double() for k, v in sd.items()})
    ne = dict(NE, num_blocks=2, in_features=width, out_features=width, depth_of_mlp=depth, constant_n_vertices=not ragged)
    model = Siamese_Node_Exp(c0, ne).to(DEV)
    model.load_state_dict({'node_embedder.' + k: v for k, v in sd.items()})
    assert model.node_embedder._standard_layout() is not None
    assert (model.node_embedder._pad is None) == (c0 in (2, 32) and width == 32)
    if ragged:
        scores = model(from_list([t.to(DEV) for t in xs], dims=(1, 2), base_name='N'),
                       from_list([t.to(DEV) for t in ys], dims=(1, 2), base_name='M'))
        got = list(scores)
    else:
        scores = model(torch.stack(xs).to(DEV), torch.stack(ys).to(DEV))
        got = list(scores)
    loss = model.loss(scores)
    loss.backward()
    for a, b32, b64 in zip(got, s32, s64):
        assert a.shape == b64.shape and rel(a.detach().cpu(), b64) < max(4 * rel(b32, b64), 2e-5)
    assert abs(loss.item() - l64.item()) < 1e-5 * abs(l64.item())
    keys = [k for k in g64 if not is_zero_grad(k, depth)]
    grads = {n[len('node_embedder.'):]: p.grad.cpu() for n, p in model.named_parameters()}
    a = torch.cat([grads[k].reshape(-1).double() for k in keys])
    b = torch.cat([g32[k].reshape(-1).double() for k in keys])
    t = torch.cat([g64[k].reshape(-1) for k in keys])
    assert (a - t).norm() <= 4.0 * (b - t).norm() + 1e-5 * t.norm()


def test_narrow_model_in_bf16_runs_on_the_padded_bf16_engine():
    """precision='bf16' with in_features = out_features = 16 (2-channel input, depth 3): the bf16 engine on the zero-padded
    layout; scores / gradient direction follow the fp32 run of the same model."""
    torch.manual_seed(3)
    ne = dict(NE, num_blocks=2, in_features=16, out_features=16, depth_of_mlp=3)
    m32 = Siamese_Node_Exp(2, ne).to(DEV)
    m16 = Siamese_Node_Exp(2, ne, precision='bf16').to(DEV)
    m16.load_state_dict(m32.state_dict())
    from graph_neural_net_amd import synthetic
    x1, x2 = synthetic.make_batch(42, 4, 24, 'ErdosRenyi', 0.3, 0.1)
    x1, x2 = x1.to(DEV), x2.to(DEV)
    out = {}
    for name, m in (('fp32', m32), ('bf16', m16)):
        s = m(x1, x2)
        m.loss(s).backward()
        out[name] = (s.detach(), torch.cat([p.grad.reshape(-1) for p in m.parameters()]))
    assert m16.node_embedder._pad is not None
    from graph_neural_net_amd.engine16 import FgnnEngineBF16
    assert all(isinstance(e, FgnnEngineBF16) for e in m16.node_embedder._engines.values())
    s32, g32 = out['fp32']
    s16, g16 = out['bf16']
    assert torch.isfinite(s16).all() and torch.isfinite(g16).all()
    assert ((s16 - s32).norm() / s32.norm()).item() < 5e-2
    assert (torch.dot(g16, g32) / (g16.norm() * g32.norm())).item() > 0.95


@pytest.mark.parametrize('c0,win,wout,depth,nblk', [(2, 32, 16, 1, 3), (32, 32, 4, 3, 2), (2, 16, 32, 2, 2)])
def test_mixed_in_out_widths_are_not_mistaken_for_the_standard_layout(c0, win, wout, depth, nblk):
    """in_features = 32 with a narrower out_features has the standard parameter NAMES but not its shapes (it used to be taken
    for the 32/32 layout: garbage scores, out-of-bounds reads); it must run zero-padded and agree with the oracle."""
    torch.manual_seed(win + wout)
    sd = O.init_state_dict(original_features_num=c0, num_blocks=nblk, in_features=win, out_features=wout, depth_of_mlp=depth)
    g = torch.Generator().manual_seed(7)
    x1, x2 = torch.randn(2, c0, 17, 17, generator=g), torch.randn(2, c0, 17, 17, generator=g)
    s32, l32, g32 = O.step_fwd_bwd(x1, x2, sd)
    s64, l64, g64 = O.step_fwd_bwd(x1.double(), x2.double(), {k: v.double() for k, v in sd.items()})
    ne = dict(NE, num_blocks=nblk, in_features=win, out_features=wout, depth_of_mlp=depth)
    model = Siamese_Node_Exp(c0, ne).to(DEV)
    model.load_state_dict({'node_embedder.' + k: v for k, v in sd.items()})
    assert model.node_embedder._standard_layout() is not None and model.node_embedder._pad is not None
    scores = model(x1.to(DEV), x2.to(DEV))
    model.loss(scores).backward()
    assert rel(scores.detach().cpu(), s64) < max(4 * rel(s32, s64), 2e-5)
    keys = [k for k in g64 if not is_zero_grad(k, depth)]
    grads = {n[len('node_embedder.'):]: p.grad.cpu() for n, p in model.named_parameters()}
    a = torch.cat([grads[k].reshape(-1).double() for k in keys])
    b = torch.cat([g32[k].reshape(-1).double() for k in keys])
    t = torch.cat([g64[k].reshape(-1) for k in keys])
    assert (a - t).norm() <= 4.0 * (b - t).norm() + 1e-5 * t.norm()


def test_conv1x1_with_padded_strides():
    """fgnn_conv1x1 / fgnn_conv1x1_dw on tensors whose channel / graph strides are larger than the data (ld > N*N, gstride >
    C*ld): the values between the planes must be neither read into the result nor overwritten."""
    G, K, M, N = 2, 5, 37, 9
    P = N * N
    ldx, ldy = P + 7, P + 3
    gsx, gsy = K * ldx + 11, M * ldy + 5
    g = torch.Generator().manual_seed(0)
    xbuf = torch.full((G * gsx,), 1e30)
    x = torch.randn(G, K, P, generator=g)
    for gi in range(G):
        for k in range(K):
            xbuf[gi * gsx + k * ldx: gi * gsx + k * ldx + P] = x[gi, k]
    w = torch.randn(M, K, generator=g)
    b = torch.randn(M, generator=g)
    ref = torch.relu(torch.einsum('mk,gkp->gmp', w.double(), x.double()) + b.double()[None, :, None])
    xd, wd, bd = xbuf.to(DEV), w.to(DEV), b.to(DEV)
    ybuf = torch.full((G * gsy,), -7.0, device=DEV)
    _lib.call('fgnn_conv1x1', _lib.ptr(xd), gsx, ldx, None, _lib.ptr(wd), K, 1, _lib.ptr(bd), 1, None, G, N, M, K,
              _lib.ptr(ybuf), gsy, ldy, _lib.stream_ptr())
    yb = ybuf.cpu()
    seen = torch.zeros(G * gsy, dtype=torch.bool)
    for gi in range(G):
        for m in range(M):
            o = gi * gsy + m * ldy
            assert rel(yb[o:o + P], ref[gi, m]) < 1e-5
            seen[o:o + P] = True
    assert (yb[~seen] == -7.0).all()                     # nothing written between the planes
    # parameter gradients from the same strided tensors
    dy = torch.randn(G, M, P, generator=g)
    dbuf = torch.full((G * gsy,), 1e30)
    for gi in range(G):
        for m in range(M):
            dbuf[gi * gsy + m * ldy: gi * gsy + m * ldy + P] = dy[gi, m]
    chunks = _lib.load().fgnn_conv1x1_dw_chunks(G, N)
    cnt = M * K + M
    wpart = torch.empty(chunks * cnt, device=DEV)
    flat = torch.empty(cnt, device=DEV)
    _lib.call('fgnn_conv1x1_dw', _lib.ptr(dbuf.to(DEV)), gsy, ldy, None, _lib.ptr(xd), gsx, ldx, None, G, N, M, K,
              _lib.ptr(wpart), _lib.stream_ptr())
    _lib.call('fgnn_reduce_partials', _lib.ptr(wpart), chunks, cnt, _lib.ptr(flat), _lib.stream_ptr())
    dw_ref = torch.einsum('gmp,gkp->mk', dy.double(), x.double())
    assert rel(flat[:M * K].view(M, K).cpu(), dw_ref) < 1e-5
    assert rel(flat[M * K:].cpu(), dy.double().sum((0, 2))) < 1e-5


def test_fused_step_on_a_non_standard_width_is_the_captured_module_path():
    """Siamese_Node_Exp.fused_step on a model the fused engine is not built for (in = out = 64): the eager module path's launch
    sequence captured in one HIP graph -- loss, scores and every gradient equal the eager run bit for bit, replay after replay,
    and an optimizer step between replays is seen by the next one."""
    from graph_neural_net_amd.siamese import Siamese_Node_Exp
    ne = dict(type='node_embedding', block_init='block_emb', block_inside='block', num_blocks=2, in_features=64,
              out_features=64, depth_of_mlp=3)
    torch.manual_seed(3)
    model = Siamese_Node_Exp(2, ne, metric='max').to(DEV)
    g = torch.Generator().manual_seed(4)
    B, N = 3, 20
    x1 = torch.randn(B, 2, N, N, generator=g).to(DEV)
    x2 = torch.randn(B, 2, N, N, generator=g).to(DEV)
    scores = model(x1, x2)
    loss = model.loss(scores)
    loss.backward()
    eager = {n: p.grad.clone() for n, p in model.named_parameters()}
    for rep in range(3):
        for p in model.parameters():
            p.grad = None
        l2, s2 = model.fused_step(x1, x2)
        assert torch.equal(s2, scores.detach()) and torch.equal(l2, loss.detach().reshape(()))
        for n, p in model.named_parameters():
            assert torch.equal(p.grad, eager[n]), (rep, n)
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    opt.step()
    l3, _ = model.fused_step(x1, x2)
    with torch.no_grad():
        l_eager = model.loss(model(x1, x2))
    assert torch.equal(l3, l_eager.reshape(())) and l3.item() != loss.item()
    # another batch shape gets its own graph; the first one still replays
    y1, y2 = x1[:2].contiguous(), x2[:2].contiguous()
    l4, s4 = model.fused_step(y1, y2)
    with torch.no_grad():
        assert torch.equal(s4, model(y1, y2))
    l5, _ = model.fused_step(x1, x2)
    assert torch.equal(l5, l3)
    # the captured steps are kept least-recently-used first out (Siamese_Node_Exp.MODULE_STEPS_MAX): with room for two, a third
    # shape drops the one that was not just replayed -- the 3-pair batch (used last above) stays, the 2-pair one goes
    model.MODULE_STEPS_MAX = 2
    z1, z2 = x1[:1].contiguous(), x2[:1].contiguous()
    model.fused_step(z1, z2)
    kept = sorted(k[0][0] for k in model._module_steps)
    assert kept == [1, 3], kept
    l6, _ = model.fused_step(x1, x2)
    assert torch.equal(l6, l3)


def test_fused_step_on_models_that_run_zero_padded_on_the_engine():
    """Widths below 32 / 3 input channels: fused_step runs the 32-wide engine on zero-padded parameters inside ONE replayed graph
    (parameter scatter, step, gradient gather) and leaves the same loss, scores and p.grad as the eager module path."""
    from graph_neural_net_amd.siamese import Siamese_Node_Exp
    for c0, win, wout in ((2, 16, 16), (3, 16, 24)):
        torch.manual_seed(5)
        ne = dict(type='node_embedding', block_init='block_emb', block_inside='block', num_blocks=2, in_features=win,
                  out_features=wout, depth_of_mlp=3)
        model = Siamese_Node_Exp(c0, ne, metric='max').to(DEV)
        g = torch.Generator().manual_seed(6)
        x1, x2 = torch.randn(3, c0, 12, 12, generator=g).to(DEV), torch.randn(3, c0, 12, 12, generator=g).to(DEV)
        scores = model(x1, x2)
        loss = model.loss(scores)
        loss.backward()
        eager = {n: p.grad.clone() for n, p in model.named_parameters()}
        for cap in (False, True, True):
            for p in model.parameters():
                p.grad = None
            l2, s2 = model.fused_step(x1, x2, capture=cap)
            assert torch.equal(s2, scores.detach()) and abs(l2.item() - loss.item()) <= 1e-6 * abs(loss.item())
            for n, p in model.named_parameters():
                assert torch.equal(p.grad, eager[n]), (c0, win, wout, cap, n)


@pytest.mark.parametrize('N,ragged', [(7, False), (50, False), (64, True), (33, True)])
def test_graphnorm_plane_kernels_equal_the_two_pass_kernels(N, ragged):
    """fgnn_gn_plane_fwd / _bwd (one workgroup per (g, c) plane, one pass) against the statistics / apply kernels they replace
    and against the fp64 definition (models/layers.py:47-80 and its autograd)."""
    G, Cc = 3, 5
    g = torch.Generator().manual_seed(N)
    nv = torch.tensor([N, max(1, N // 2), N - 1], dtype=torch.int32) if ragged else None
    mask = _valid_mask(G, N, nv)
    x = (torch.randn(G, Cc, N, N, generator=g).double() * mask).float() * 2 + 0.5 * mask.float()
    dy = (torch.randn(G, Cc, N, N, generator=g).double() * mask).float()
    gw, gb = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g)
    xd, dyd, gwd, gbd = x.to(DEV), dy.to(DEV), gw.to(DEV), gb.to(DEV)
    nvd = nv.to(DEV) if ragged else None
    nvp = _lib.ptr(nvd) if ragged else None
    P, st = N * N, _lib.stream_ptr()
    f32 = dict(dtype=torch.float32, device=DEV)
    y1, y2 = torch.full_like(xd, float('nan')), torch.full_like(xd, float('nan'))
    n1, n2 = torch.empty(G * Cc * 4, **f32), torch.empty(G * Cc * 4, **f32)
    _lib.call('fgnn_gn_plane_fwd', _lib.ptr(xd), Cc * P, P, _lib.ptr(gwd), _lib.ptr(gbd), nvp, G, Cc, N, 1e-5, _lib.ptr(y1),
              Cc * P, P, _lib.ptr(n1), st)
    _lib.call('fgnn_gn_stats', _lib.ptr(xd), Cc * P, P, _lib.ptr(gwd), nvp, G, Cc, N, 1e-5, _lib.ptr(n2), st)
    _lib.call('fgnn_gn_apply', _lib.ptr(xd), Cc * P, P, _lib.ptr(n2), _lib.ptr(gbd), nvp, G, Cc, N, _lib.ptr(y2), Cc * P, P, st)
    assert rel(n1.cpu(), n2.cpu()) < 1e-5 and rel(y1.cpu(), y2.cpu()) < 1e-5
    assert (y1.cpu().double() * (1 - mask)).abs().max() == 0
    # fp64 definition
    xr = x.double().requires_grad_(True)
    gwr, gbr = gw.double().requires_grad_(True), gb.double().requires_grad_(True)
    ys = []
    for b in range(G):
        n = N if nv is None else int(nv[b])
        xb = xr[b, :, :n, :n]
        mean = xb.mean((-1, -2), keepdim=True)
        var = ((xb - mean) ** 2).mean((-1, -2), keepdim=True)
        yb = gwr[:, None, None] * (xb - mean) / (2 * torch.sqrt(n * (var + 1e-5))) + gbr[:, None, None]
        ys.append(torch.nn.functional.pad(yb, (0, N - n, 0, N - n)))
    yr = torch.stack(ys)
    yr.backward(dy.double())
    assert rel(y1.cpu(), yr.detach()) < 1e-5
    dz1, dz2 = torch.full_like(xd, float('nan')), torch.full_like(xd, float('nan'))
    s1, s2, coef = torch.empty(G * Cc * 2, **f32), torch.empty(G * Cc * 2, **f32), torch.empty(G * Cc * 4, **f32)
    dw1, db1, dw2, db2 = (torch.empty(Cc, **f32) for _ in range(4))
    _lib.call('fgnn_gn_plane_bwd', _lib.ptr(dyd), Cc * P, P, _lib.ptr(xd), Cc * P, P, _lib.ptr(n1), nvp, G, Cc, N, _lib.ptr(dz1),
              Cc * P, P, _lib.ptr(s1), _lib.ptr(dw1), _lib.ptr(db1), st)
    _lib.call('fgnn_gn_bwd_stats', _lib.ptr(dyd), Cc * P, P, _lib.ptr(xd), Cc * P, P, _lib.ptr(n2), nvp, G, Cc, N, _lib.ptr(s2), st)
    _lib.call('fgnn_gn_bwd_coef', _lib.ptr(s2), _lib.ptr(n2), nvp, G, Cc, N, _lib.ptr(coef), _lib.ptr(dw2), _lib.ptr(db2), st)
    _lib.call('fgnn_gn_bwd_apply', _lib.ptr(dyd), Cc * P, P, _lib.ptr(xd), Cc * P, P, _lib.ptr(coef), nvp, G, Cc, N, _lib.ptr(dz2),
              Cc * P, P, st)
    assert rel(dz1.cpu(), dz2.cpu()) < 2e-5 and rel(dw1.cpu(), dw2.cpu()) < 2e-5 and rel(db1.cpu(), db2.cpu()) < 2e-5
    assert rel(dz1.cpu(), xr.grad) < 2e-5 and rel(dw1.cpu(), gwr.grad) < 2e-5 and rel(db1.cpu(), gbr.grad) < 2e-5
    assert (dz1.cpu().double() * (1 - mask)).abs().max() == 0
    # bit-reproducible
    dz3, s3 = torch.empty_like(xd), torch.empty_like(s1)
    _lib.call('fgnn_gn_plane_bwd', _lib.ptr(dyd), Cc * P, P, _lib.ptr(xd), Cc * P, P, _lib.ptr(n1), nvp, G, Cc, N, _lib.ptr(dz3),
              Cc * P, P, _lib.ptr(s3), None, None, st)
    assert torch.equal(dz3, dz1) and torch.equal(s3, s1)


@pytest.mark.parametrize('cin,widths', [(64, (64, 64, 64)), (128, (64, 64, 64)), (2, (64, 64, 64)), (19, (48, 24)), (33, (40, 40, 70)),
                                        (7, (16,)), (64, (33, 64, 128))])
@pytest.mark.parametrize('ragged', [False, True])
def test_conv_chain_equals_the_per_layer_kernels(cin, widths, ragged):
    """_ConvChainFn (the whole conv stack of an MlpBlock_Real in one launch per direction, csrc/conv.hip conv_chain_kernel)
    against the per-layer _ConvFn launches it replaces and the fp64 ATen definition: output, input gradient and every
    parameter gradient; exact zeros in the padding of ragged graphs."""
    from graph_neural_net_amd.layers import _ConvChainFn, _ConvFn, _chain_supported
    assert _chain_supported(cin, list(widths))
    G, N = 3, 13
    g = torch.Generator().manual_seed(cin * 7 + len(widths))
    nv = torch.tensor([N, 6, N - 1], dtype=torch.int32) if ragged else None
    mask = _valid_mask(G, N, nv)
    x = (torch.randn(G, cin, N, N, generator=g).double() * mask).float()
    ws, bs, k = [], [], cin
    for m in widths:
        ws.append(torch.randn(m, k, 1, 1, generator=g) / k ** 0.5)
        bs.append(0.3 * torch.randn(m, generator=g))
        k = m
    dy = (torch.randn(G, widths[-1], N, N, generator=g).double() * mask).float()
    nvd = nv.to(DEV) if ragged else None

    def run(fn_kind):
        xd = x.to(DEV).requires_grad_(True)
        wd = [w.to(DEV).requires_grad_(True) for w in ws]
        bd = [b.to(DEV).requires_grad_(True) for b in bs]
        if fn_kind == 'chain':
            wb = []
            for w, b in zip(wd, bd):
                wb += [w, b]
            y = _ConvChainFn.apply(xd, nvd, *wb)
        else:
            y = xd
            for l, (w, b) in enumerate(zip(wd, bd)):
                y = _ConvFn.apply(y, nvd, w, b, l < len(wd) - 1)
        y.backward(dy.to(DEV))
        return [y.detach().cpu(), xd.grad.cpu()] + [w.grad.cpu() for w in wd] + [b.grad.cpu() for b in bd]

    got, ref = run('chain'), run('layers')
    # fp64 definition with the MaskedTensor semantics (re-mask after every op)
    xr = x.double().requires_grad_(True)
    wr = [w.double().requires_grad_(True) for w in ws]
    br = [b.double().requires_grad_(True) for b in bs]
    yr = xr
    for l, (w, b) in enumerate(zip(wr, br)):
        yr = F.conv2d(yr, w, b)
        if l < len(wr) - 1:
            yr = F.relu(yr)
        yr = yr * mask
    yr.backward(dy.double())
    truth = [yr.detach(), xr.grad] + [w.grad for w in wr] + [b.grad for b in br]
    for a, b_, t in zip(got, ref, truth):
        assert rel(a, t) < 2e-5, (rel(a, t), rel(b_, t))
        assert rel(a, b_) < 2e-5
    assert (got[0].double() * (1 - mask)).abs().max() == 0 and (got[1].double() * (1 - mask)).abs().max() == 0


@pytest.mark.parametrize('cin', [2, 19, 32, 64, 66, 100, 128])
@pytest.mark.parametrize('shape', [(3, 13, False), (3, 13, True), (5, 50, False), (4, 37, True)])
@pytest.mark.parametrize('need_dx', [True, False])
def test_mlp64_equals_the_conv_chain(cin, shape, need_dx):
    """_Mlp64Fn (csrc/mlp64.hip: the conv stack of a 64-wide MlpBlock_Real fused, one launch per direction, hidden activations
    recomputed in the backward, weight gradients accumulated in the waves' registers) against the conv chain it replaces and the
    fp64 ATen definition with the MaskedTensor semantics (models/layers.py:125-131): output, input gradient, every parameter
    gradient; exact zeros in the padding of ragged graphs."""
    from graph_neural_net_amd.layers import _ConvChainFn, _Mlp64Fn, _mlp64_supported
    assert _mlp64_supported(cin, [64, 64, 64])
    G, N, ragged = shape
    g = torch.Generator().manual_seed(cin * 11 + N)
    nv = torch.tensor([N, max(N // 2, 1), N - 1, 3, N][:G], dtype=torch.int32) if ragged else None
    mask = _valid_mask(G, N, nv)
    x = (torch.randn(G, cin, N, N, generator=g).double() * mask).float()
    ws, bs, k = [], [], cin
    for m in (64, 64, 64):
        ws.append(torch.randn(m, k, 1, 1, generator=g) / k ** 0.5)
        bs.append(0.3 * torch.randn(m, generator=g))
        k = m
    dy = (torch.randn(G, 64, N, N, generator=g).double() * mask).float()
    nvd = nv.to(DEV) if ragged else None

    def run(fn):
        xd = x.to(DEV).requires_grad_(need_dx)
        wd = [w.to(DEV).requires_grad_(True) for w in ws]
        bd = [b.to(DEV).requires_grad_(True) for b in bs]
        wb = []
        for w, b in zip(wd, bd):
            wb += [w, b]
        y = fn.apply(xd, None, nvd, None, None, None, *wb) if fn is _Mlp64Fn else fn.apply(xd, nvd, *wb)
        y.backward(dy.to(DEV))
        return [y.detach().cpu()] + ([xd.grad.cpu()] if need_dx else []) + [w.grad.cpu() for w in wd] + [b.grad.cpu() for b in bd]

    got, ref = run(_Mlp64Fn), run(_ConvChainFn)
    xr = x.double().requires_grad_(True)
    wr = [w.double().requires_grad_(True) for w in ws]
    br = [b.double().requires_grad_(True) for b in bs]
    yr = xr
    for l, (w, b) in enumerate(zip(wr, br)):
        yr = F.conv2d(yr, w, b)
        if l < 2:
            yr = F.relu(yr)
        yr = yr * mask
    yr.backward(dy.double())
    truth = [yr.detach()] + ([xr.grad] if need_dx else []) + [w.grad for w in wr] + [b.grad for b in br]
    assert len(got) == len(truth)
    for a, b_, t in zip(got, ref, truth):
        assert a.shape == t.shape
        assert rel(a, t) < 2e-5, (rel(a, t), rel(b_, t))
        assert rel(a, b_) < 2e-5
    assert (got[0].double() * (1 - mask)).abs().max() == 0
    if need_dx:
        assert (got[1].double() * (1 - mask)).abs().max() == 0


NE64 = dict(type='node_embedding', block_init='block_emb', block_inside='block', num_blocks=4, in_features=64, out_features=64,
            depth_of_mlp=3)


def _load_model64(d, **kw):
    model = Siamese_Node_Exp(2, dict(NE64, **kw)).to(DEV)
    model.load_state_dict({'node_embedder.' + k: v for k, v in sub(d, 'sd/').items()})
    return model


def _count_mlp64(fn):
    """run fn() with the C entry points counted: {name: calls}"""
    seen = {}
    orig = _lib.call

    def counting(name, *a, **k):
        seen[name] = seen.get(name, 0) + 1
        return orig(name, *a, **k)
    _lib.call = counting
    try:
        out = fn()
    finally:
        _lib.call = orig
    return out, seen


def test_siamese_64_wide_against_reference_golden():
    """original_features_num 2, in_features = out_features = 64, depth 3, 4 blocks (convs 2->64, 66->64, 64->64, 128->64) on regular-graph
    pairs at N = 50: every conv stack runs on the fused 64-wide kernels (csrc/mlp64.hip: 12 forward + 12 backward launches per side pair),
    scores, loss, intermediates and every gradient against the reference's own fp32 / fp64 runs (tests/golden/make_golden.py wide64)."""
    d = load_golden('wide64_c2_64_64_d3_4blk.npz')
    model = _load_model64(d)
    assert model.node_embedder._standard_layout() is None          # not the 32-wide engine: the per-layer modules
    out = model.node_embedder({'input': d['x1'].to(DEV)})
    for k, v in sub(d, 'inter/').items():              # yard-stick: the reference's own fp32 run against its fp64 run at the same node
        v64 = d['inter64/' + k]
        assert rel(out[k].detach().cpu()[:1], v64) < max(2 * rel(v, v64), 1e-5), k

    def step():
        scores = model({'input': d['x1'].to(DEV)}, {'input': d['x2'].to(DEV)})
        loss = model.loss(scores)
        loss.backward()
        return scores, loss
    (scores, loss), seen = _count_mlp64(step)
    assert seen.get('fgnn_mlp64_fwd') == 12 and seen.get('fgnn_mlp64_bwd') == 12 and 'fgnn_conv_chain' not in seen and 'fgnn_conv1x1' not in seen
    yard = rel(d['scores'], d['scores64'])
    assert rel(scores.detach().cpu(), d['scores64']) < max(2 * yard, 1e-5)
    assert abs(loss.item() - d['loss64'].item()) < 1e-5 * d['loss64'].item()
    worst = 0.0
    for n, p in model.named_parameters():
        k = n[len('node_embedder.'):]
        if is_zero_grad(k, 3):
            assert p.grad.abs().max() < 1e-4
        else:
            yard = rel(d['grad/' + k], d['grad64/' + k])
            e = rel(p.grad.cpu(), d['grad64/' + k])
            worst = max(worst, e / (yard + 1e-12))
            assert e < 4 * yard + 1e-5, k
    print('64-wide golden: worst gradient error / the reference\'s own fp32-vs-fp64 error = %.2f' % worst)


def test_siamese_64_wide_ragged_against_reference_golden():
    d = load_golden('wide64_c2_64_64_d3_4blk.npz')
    model = _load_model64(d, constant_n_vertices=False)
    n = len(d['ragged/ns'])
    m1 = from_list([d['ragged/x1/%d' % i].to(DEV) for i in range(n)], dims=(1, 2), base_name='N')
    m2 = from_list([d['ragged/x2/%d' % i].to(DEV) for i in range(n)], dims=(1, 2), base_name='M')

    def step():
        scores = model(m1, m2)
        loss = model.loss(scores)
        loss.backward()
        return scores, loss
    (scores, loss), seen = _count_mlp64(step)
    assert seen.get('fgnn_mlp64_fwd') == 12 and seen.get('fgnn_mlp64_bwd') == 12
    for i, a in enumerate(list(scores)):
        ref64 = d['ragged/scores64/%d' % i]
        assert a.shape == ref64.shape
        assert rel(a.detach().cpu(), ref64) < max(2 * rel(d['ragged/scores/%d' % i], ref64), 1e-5)
    assert abs(loss.item() - d['ragged/loss64'].item()) < 1e-5 * d['ragged/loss64'].item()
    for name, p in model.named_parameters():
        k = name[len('node_embedder.'):]
        if not is_zero_grad(k, 3):
            yard = rel(d['ragged/grad/' + k], d['ragged/grad64/' + k])
            assert rel(p.grad.cpu(), d['ragged/grad64/' + k]) < 4 * yard + 1e-5, k


def test_mlp64_rejects_bad_arguments():
    a = _lib.Mlp64Args()
    x = torch.zeros(1, 64, 5, 5, device=DEV)
    a.x, a.x_gstride, a.x_ld, a.cin, a.G, a.N = x.data_ptr(), 64 * 25, 25, 64, 1, 5
    with pytest.raises(RuntimeError, match='packed'):
        _lib.call('fgnn_mlp64_fwd', C.byref(a), _lib.stream_ptr())
    a.cin = 200
    with pytest.raises(RuntimeError, match='input channels'):
        _lib.call('fgnn_mlp64_fwd', C.byref(a), _lib.stream_ptr())
    assert not _lib.load().fgnn_mlp64_supported(64, 2, 64) and not _lib.load().fgnn_mlp64_supported(64, 3, 48)
    assert _lib.load().fgnn_mlp64_supported(66, 3, 64)


@pytest.mark.parametrize('cb', [2, 30, 64])
@pytest.mark.parametrize('ragged', [False, True])
def test_mlp64_two_slabs_equal_the_concatenated_input(cb, ragged):
    """_Mlp64Fn on the two parts of a Concat ([mult ; in] of a block, models/blocks_emb.py:33-36) against the same function on the
    concatenated tensor: output, both input gradients, every parameter gradient -- bit for bit (the same kernel arithmetic, only the
    addresses differ)."""
    from graph_neural_net_amd.layers import _Mlp64Fn
    G, N = 4, 21
    g = torch.Generator().manual_seed(900 + cb)
    nv = torch.tensor([N, 9, N - 2, 1], dtype=torch.int32) if ragged else None
    mask = _valid_mask(G, N, nv)
    xa = (torch.randn(G, 64, N, N, generator=g).double() * mask).float()
    xb = (torch.randn(G, cb, N, N, generator=g).double() * mask).float()
    ws, bs, k = [], [], 64 + cb
    for m in (64, 64, 64):
        ws.append(torch.randn(m, k, 1, 1, generator=g) / k ** 0.5)
        bs.append(0.3 * torch.randn(m, generator=g))
        k = m
    dy = (torch.randn(G, 64, N, N, generator=g).double() * mask).float()
    nvd = nv.to(DEV) if ragged else None

    def run(two):
        a, b = xa.to(DEV).requires_grad_(True), xb.to(DEV).requires_grad_(True)
        wd = [w.to(DEV).requires_grad_(True) for w in ws]
        bd = [b_.to(DEV).requires_grad_(True) for b_ in bs]
        wb = []
        for w, b_ in zip(wd, bd):
            wb += [w, b_]
        y = _Mlp64Fn.apply(a, b, nvd, None, None, None, *wb) if two else _Mlp64Fn.apply(torch.cat([a, b], 1), None, nvd, None, None, None, *wb)
        y.backward(dy.to(DEV))
        return [y.detach(), a.grad, b.grad] + [w.grad for w in wd] + [b_.grad for b_ in bd]

    for u, v in zip(run(True), run(False)):
        assert torch.equal(u, v)


def test_network_keeps_concat_lazy_for_the_fused_64_wide_mlp():
    """Network.forward: a Concat node read only by MlpBlock_Real stays a LazyCat (no concatenated copy) and still reads as the tensor
    through the returned dict (models/utils.py:60-69 returns every node output)."""
    from graph_neural_net_amd.layers import LazyCat
    d = load_golden('wide64_c2_64_64_d3_4blk.npz')
    model = _load_model64(d)
    out = model.node_embedder({'input': d['x1'].to(DEV)})
    for b in (1, 2, 3, 4):
        raw = out.raw('ne/bm/block%d/cat' % b)
        assert isinstance(raw, LazyCat) and raw._cat is None
    cat = out['ne/bm/block2/cat']
    assert torch.is_tensor(cat) and cat.shape[1] == 128
    assert torch.equal(cat[:, :64], out['ne/bm/block2/mult']) and torch.equal(cat[:, 64:], out['ne/bm/block1/mlp3'])
    assert dict(out.items())['ne/bm/block3/cat'].shape[1] == 128
    assert torch.is_tensor(dict(out)['ne/bm/block4/cat']) and torch.is_tensor({**out}['ne/bm/block4/cat']) and torch.is_tensor(out.copy()['ne/bm/block1/cat'])
    # the operand records prepack64 hands the MLPs are good for one pass only
    assert all(getattr(m, '_packed64', None) is None for m in model.node_embedder.modules())


def test_fan_out_sums_the_input_gradients_inside_the_mlp_kernels():
    """layers.fan_out: three fused MLPs reading one tensor (a block's mlp1, mlp2 and -- as the second slab -- mlp3) add their input
    gradients to ONE buffer inside their backward kernels; an ordinary reader's gradient joins through autograd.  Same result (to one
    rounding of the three-term sum) as autograd's own accumulation, and no `add` kernel is needed for the fused readers."""
    from graph_neural_net_amd.layers import _Mlp64Fn, fan_out
    G, N = 3, 17
    g = torch.Generator().manual_seed(77)
    x0 = torch.randn(G, 64, N, N, generator=g)
    m0 = torch.randn(G, 64, N, N, generator=g)

    def params(k):
        out = []
        for kk in (k, 64, 64):
            out += [(torch.randn(64, kk, 1, 1, generator=g) / kk ** 0.5).to(DEV).requires_grad_(True), (0.1 * torch.randn(64, generator=g)).to(DEV).requires_grad_(True)]
        return out
    p1, p2, p3 = params(64), params(64), params(128)
    dys = [torch.randn(G, 64, N, N, generator=g).to(DEV) for _ in range(3)]

    def run(fanned):
        leaf = x0.to(DEV).requires_grad_(True)
        x = leaf * 1.0                                   # a non-leaf, like a block's output
        if fanned:
            x = fan_out(x)
        sink = getattr(x, '_fgnn_sink', None)
        assert (sink is not None) == fanned
        mult = m0.to(DEV)
        y1 = _Mlp64Fn.apply(x, None, None, None, sink, None, *p1)
        y2 = _Mlp64Fn.apply(x, None, None, None, sink, None, *p2)
        y3 = _Mlp64Fn.apply(mult, x, None, None, None, sink, *p3)
        extra = (x * x).sum()                            # an ordinary reader
        for p in p1 + p2 + p3:
            p.grad = None
        torch.autograd.backward([y1, y2, y3, extra], dys + [torch.ones((), device=DEV)])
        return [leaf.grad] + [p.grad.clone() for p in p1 + p2 + p3]

    a, b = run(True), run(False)
    assert rel(a[0].cpu(), b[0].cpu()) < 1e-6
    for u, v in zip(a[1:], b[1:]):
        assert torch.equal(u, v)


def test_64_wide_model_trains_eager_and_captured():
    """The 64-feature model on the fused kernels learns (Adam, the reference's configure_optimizers), step by step through autograd and as
    the replayed fused_step -- which must track the eager steps (same kernels, same order)."""
    d = load_golden('wide64_c2_64_64_d3_4blk.npz')
    batch = ({'input': d['x1'].to(DEV)}, {'input': d['x2'].to(DEV)})
    model = _load_model64(d)
    opt = model.configure_optimizers()['optimizer']
    first = None
    for _ in range(30):
        opt.zero_grad()
        loss = model.training_step(batch, 0)
        first = first if first is not None else loss.item()
        loss.backward()
        opt.step()
    assert loss.item() < 0.95 * first                     # (ln 50 = 3.91 at the start; 3.95 -> 3.69 after 25 steps at lr 1e-3)
    # one captured step against one eager step from the same state
    m1, m2 = _load_model64(d), _load_model64(d)
    l1 = m1.loss(m1(*batch))
    l1.backward()
    for _ in range(2):                                     # (the first calls capture)
        out = m2.fused_step(*batch)
    l2 = out['loss'] if isinstance(out, dict) else out
    l2 = l2[0] if isinstance(l2, (tuple, list)) else l2
    assert abs(float(l1) - float(l2)) < 1e-6 * abs(float(l1)) + 1e-7
    for (n, p), (_, q) in zip(m1.named_parameters(), m2.named_parameters()):
        assert q.grad is not None and rel(q.grad.cpu(), p.grad.cpu()) < 1e-6, n
