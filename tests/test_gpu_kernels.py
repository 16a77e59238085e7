"""GPU: individual entry points of the C ABI against plain PyTorch fp32/fp64 references."""
import ctypes as C

import numpy as np
import pytest
import torch

from graph_neural_net_amd import _lib
from util import rel

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _slab(t, nrm=None, beta=None):
    G, Cc, N, _ = t.shape
    return _lib.make_slab(t, Cc * N * N, N * N, Cc, nrm=nrm, beta=beta)


@pytest.mark.parametrize('G,Cc,N', [(3, 4, 50), (2, 2, 7), (1, 3, 64), (2, 2, 65), (1, 2, 130), (1, 2, 256), (1, 1, 260)])
def test_chan_matmul_fwd_bwd(G, Cc, N):
    g = torch.Generator().manual_seed(N)
    a = torch.randn(G, Cc, N, N, generator=g)
    b = torch.randn(G, Cc, N, N, generator=g)     # asymmetric operands: catches transposed fragments
    dm = torch.randn(G, Cc, N, N, generator=g)
    ad, bd, dmd = a.to(DEV), b.to(DEV), dm.to(DEV)
    out = torch.empty_like(ad)
    P = N * N
    sa, sb = _slab(ad), _slab(bd)
    _lib.call('fgnn_chan_matmul_fwd', C.byref(sa), C.byref(sb), None, G, N, _lib.ptr(out), Cc * P, P, _lib.stream_ptr())
    ref = torch.matmul(a.double(), b.double())
    assert rel(out.cpu(), ref) < 2e-6
    da, db = torch.empty_like(ad), torch.empty_like(ad)
    _lib.call('fgnn_chan_matmul_bwd', C.byref(sa), C.byref(sb), _lib.ptr(dmd), Cc * P, P, None, G, N,
              _lib.ptr(da), _lib.ptr(db), Cc * P, P, None, None, _lib.stream_ptr())
    assert rel(da.cpu(), torch.matmul(dm.double(), b.double().transpose(-1, -2))) < 2e-6
    assert rel(db.cpu(), torch.matmul(a.double().transpose(-1, -2), dm.double())) < 2e-6


@pytest.mark.parametrize('seed', range(8))
def test_chan_matmul_random_ragged_shapes(seed):
    """Random (G, C, N, nvalid) across the three matmul paths (single tile, whole-matrix, generic), normalised
    operands, with the S1/S2 by-products of the backward checked against their definition."""
    rng = torch.Generator().manual_seed(1000 + seed)
    N = int(torch.randint(2, 150, (1,), generator=rng))
    G = int(torch.randint(1, 4, (1,), generator=rng))
    Cc = int(torch.randint(1, 5, (1,), generator=rng))
    nv = torch.randint(1, N + 1, (G,), generator=rng).to(torch.int32)
    nv[0] = N
    P = N * N
    a = torch.randn(G, Cc, N, N, generator=rng)
    b = torch.randn(G, Cc, N, N, generator=rng)
    dm = torch.randn(G, Cc, N, N, generator=rng)
    nrm_a = torch.rand(G, Cc, 4, generator=rng) + 0.5          # {mean, a, q, r2}
    nrm_b = torch.rand(G, Cc, 4, generator=rng) + 0.5
    beta_a, beta_b = torch.randn(Cc, generator=rng), torch.randn(Cc, generator=rng)
    ad, bd, dmd = a.to(DEV), b.to(DEV), dm.to(DEV)
    na, nb, ba, bb = nrm_a.to(DEV).contiguous(), nrm_b.to(DEV).contiguous(), beta_a.to(DEV), beta_b.to(DEV)
    sa, sb = _slab(ad, nrm=na, beta=ba), _slab(bd, nrm=nb, beta=bb)
    nvd = nv.to(DEV)
    out = torch.full((G, Cc, N, N), 3.0, device=DEV)
    _lib.call('fgnn_chan_matmul_fwd', C.byref(sa), C.byref(sb), _lib.ptr(nvd), G, N, _lib.ptr(out), Cc * P, P, _lib.stream_ptr())
    da = torch.full((G, Cc, N, N), 3.0, device=DEV)
    db = torch.full((G, Cc, N, N), 3.0, device=DEV)
    s12a = torch.empty(G * Cc * 2, device=DEV)
    s12b = torch.empty(G * Cc * 2, device=DEV)
    _lib.call('fgnn_chan_matmul_bwd', C.byref(sa), C.byref(sb), _lib.ptr(dmd), Cc * P, P, _lib.ptr(nvd), G, N,
              _lib.ptr(da), _lib.ptr(db), Cc * P, P, _lib.ptr(s12a), _lib.ptr(s12b), _lib.stream_ptr())
    out, da, db = out.cpu(), da.cpu(), db.cpu()
    s12a, s12b = s12a.cpu().view(G, Cc, 2), s12b.cpu().view(G, Cc, 2)
    for g, n in enumerate(nv.tolist()):
        ya = ((a[g, :, :n, :n].double() - nrm_a[g, :, 0, None, None]) * nrm_a[g, :, 1, None, None] + beta_a[:, None, None])
        yb = ((b[g, :, :n, :n].double() - nrm_b[g, :, 0, None, None]) * nrm_b[g, :, 1, None, None] + beta_b[:, None, None])
        d = dm[g, :, :n, :n].double()
        assert rel(out[g, :, :n, :n], ya @ yb) < 5e-6
        ra, rb = d @ yb.transpose(-1, -2), ya.transpose(-1, -2) @ d
        assert rel(da[g, :, :n, :n], ra) < 5e-6 and rel(db[g, :, :n, :n], rb) < 5e-6
        for t in (out, da, db):
            assert t[g, :, n:, :].abs().sum() == 0 and t[g, :, :, n:].abs().sum() == 0
        ua = a[g, :, :n, :n].double() - nrm_a[g, :, 0, None, None]
        ub = b[g, :, :n, :n].double() - nrm_b[g, :, 0, None, None]
        ref_a = torch.stack([ra.sum((-1, -2)), (ra * ua).sum((-1, -2))], -1)
        ref_b = torch.stack([rb.sum((-1, -2)), (rb * ub).sum((-1, -2))], -1)
        scale = max(ref_a.abs().max().item(), ref_b.abs().max().item(), 1.0)
        assert (s12a[g].double() - ref_a).abs().max() < 2e-5 * scale
        assert (s12b[g].double() - ref_b).abs().max() < 2e-5 * scale


@pytest.mark.parametrize('N,G,Cc', [(103, 6, 3), (128, 4, 2), (70, 5, 4), (200, 3, 2)])
def test_chan_matmul_longest_first_order_is_bit_identical(N, G, Cc):
    """fgnn_ragged_tile_ranges_order sorts the graphs by size (largest first, ties by index); the whole-matrix products
    launched in that order -- the backward with one workgroup per product -- and the k-steps skipped in a matrix' last
    chunk change nothing in the results: same bits as the plain entry points, S1 / S2 by-products included."""
    rng = torch.Generator().manual_seed(7 * N + G)
    nv = torch.randint(1, N + 1, (G,), generator=rng).to(torch.int32)
    nv[G // 2] = N
    if G > 2:
        nv[0] = nv[G - 1]                                   # a tie
    nvd = nv.to(DEV)
    ranges = torch.empty(_lib.FGNN_RANGE_WG + 1, dtype=torch.int32, device=DEV)
    order = torch.full((G,), -1, dtype=torch.int32, device=DEV)
    _lib.call('fgnn_ragged_tile_ranges_order', _lib.ptr(nvd), G, N, _lib.ptr(ranges), _lib.ptr(order), _lib.stream_ptr())
    want = sorted(range(G), key=lambda g: (-int(nv[g]), g))
    assert order.cpu().tolist() == want
    ranges2 = torch.empty_like(ranges)
    _lib.call('fgnn_ragged_tile_ranges', _lib.ptr(nvd), G, N, _lib.ptr(ranges2), _lib.stream_ptr())
    assert torch.equal(ranges, ranges2)
    P = N * N
    a = torch.randn(G, Cc, N, N, generator=rng).to(DEV)
    b = torch.randn(G, Cc, N, N, generator=rng).to(DEV)
    dm = torch.randn(G, Cc, N, N, generator=rng).to(DEV)
    na = (torch.rand(G, Cc, 4, generator=rng) + 0.5).to(DEV).contiguous()
    nb = (torch.rand(G, Cc, 4, generator=rng) + 0.5).to(DEV).contiguous()
    ba, bb = torch.randn(Cc, generator=rng).to(DEV), torch.randn(Cc, generator=rng).to(DEV)
    sa, sb = _slab(a, nrm=na, beta=ba), _slab(b, nrm=nb, beta=bb)
    res = []
    for o in (None, order):
        out = torch.full((G, Cc, N, N), 3.0, device=DEV)
        da, db = torch.full_like(out, 3.0), torch.full_like(out, 3.0)
        s12a, s12b = torch.full((G * Cc * 2,), 3.0, device=DEV), torch.full((G * Cc * 2,), 3.0, device=DEV)
        _lib.call('fgnn_chan_matmul_fwd_ord', C.byref(sa), C.byref(sb), _lib.ptr(nvd), G, N, _lib.ptr(out), Cc * P, P,
                  _lib.ptr(o), 0, _lib.stream_ptr())
        _lib.call('fgnn_chan_matmul_bwd_ord', C.byref(sa), C.byref(sb), _lib.ptr(dm), Cc * P, P, _lib.ptr(nvd), G, N,
                  _lib.ptr(da), _lib.ptr(db), Cc * P, P, _lib.ptr(s12a), _lib.ptr(s12b), _lib.ptr(o), 0, _lib.stream_ptr())
        res.append((out, da, db, s12a, s12b))
    for x, y in zip(*res):
        assert torch.equal(x, y)
    # and against the definition (the skipped k-steps): fp64 reference on the valid corner of the largest graph
    g = G // 2
    ya = (a[g].double() - na[g, :, 0, None, None]) * na[g, :, 1, None, None] + ba[:, None, None]
    yb = (b[g].double() - nb[g, :, 0, None, None]) * nb[g, :, 1, None, None] + bb[:, None, None]
    assert rel(res[1][0][g].cpu(), (ya @ yb).cpu()) < 5e-6
    assert rel(res[1][1][g].cpu(), (dm[g].double() @ yb.transpose(-1, -2)).cpu()) < 5e-6
    assert rel(res[1][2][g].cpu(), (ya.transpose(-1, -2) @ dm[g].double()).cpu()) < 5e-6


def test_chan_matmul_ragged_padding_is_zero():
    G, Cc, N = 3, 2, 20
    nv = torch.tensor([20, 13, 5], dtype=torch.int32)
    a = torch.randn(G, Cc, N, N)
    b = torch.randn(G, Cc, N, N)                  # garbage in the padding on purpose
    out = torch.full((G, Cc, N, N), 7.0, device=DEV)
    ad, bd = a.to(DEV), b.to(DEV)
    sa, sb = _slab(ad), _slab(bd)
    P = N * N
    _lib.call('fgnn_chan_matmul_fwd', C.byref(sa), C.byref(sb), _lib.ptr(nv.to(DEV)), G, N, _lib.ptr(out), Cc * P, P,
              _lib.stream_ptr())
    out = out.cpu()
    for g, n in enumerate(nv.tolist()):
        ref = a[g, :, :n, :n].double() @ b[g, :, :n, :n].double()
        assert rel(out[g, :, :n, :n], ref) < 2e-6
        assert out[g, :, n:, :].abs().sum() == 0 and out[g, :, :, n:].abs().sum() == 0


@pytest.mark.parametrize('N,nvs', [(120, [120, 75, 96, 64, 31, 1, 0]), (200, [200, 129, 160, 97]), (70, [70, 33, 64])])
def test_chan_matmul_ragged_minimal_fill_covers_every_live_tile(N, nvs):
    """fill = 1 (the tile-skipping engines): outputs poisoned with NaN beforehand; every pixel of every LIVE tile (a tile = 32
    consecutive pixels of the plane, live = holds a pixel of the valid corner) must come out as the product (inside the corner) or
    exactly zero (outside) -- for the forward product and both backward products; results equal fill = 0 bit for bit there."""
    G, Cc = len(nvs), 2
    rng = torch.Generator().manual_seed(N)
    nv = torch.tensor(nvs, dtype=torch.int32)
    nvd = nv.to(DEV)
    P = N * N
    a, b, dm = (torch.randn(G, Cc, N, N, generator=rng).to(DEV) for _ in range(3))
    na = (torch.rand(G, Cc, 4, generator=rng) + 0.5).to(DEV).contiguous()
    nb = (torch.rand(G, Cc, 4, generator=rng) + 0.5).to(DEV).contiguous()
    sa, sb = _slab(a, nrm=na), _slab(b, nrm=nb)
    order = torch.argsort(-nv, stable=True).to(torch.int32).to(DEV)
    res = []
    for fill in (0, 1):
        out, da, db = (torch.full((G, Cc, N, N), float('nan'), device=DEV) for _ in range(3))
        s12a, s12b = torch.empty(G * Cc * 2, device=DEV), torch.empty(G * Cc * 2, device=DEV)
        _lib.call('fgnn_chan_matmul_fwd_ord', C.byref(sa), C.byref(sb), _lib.ptr(nvd), G, N, _lib.ptr(out), Cc * P, P, _lib.ptr(order), fill,
                  _lib.stream_ptr())
        _lib.call('fgnn_chan_matmul_bwd_ord', C.byref(sa), C.byref(sb), _lib.ptr(dm), Cc * P, P, _lib.ptr(nvd), G, N, _lib.ptr(da), _lib.ptr(db),
                  Cc * P, P, _lib.ptr(s12a), _lib.ptr(s12b), _lib.ptr(order), fill, _lib.stream_ptr())
        torch.cuda.synchronize()
        res.append([t.cpu() for t in (out, da, db, s12a, s12b)])
    assert not any(torch.isnan(t).any() for t in res[0])                 # fill = 0 writes the whole frame
    for g, n in enumerate(nvs):
        live = torch.zeros(P, dtype=torch.bool)
        if n > 0:
            pix = (torch.arange(n)[:, None] * N + torch.arange(n)[None, :]).reshape(-1)        # the valid corner
            tiles = torch.unique(pix // 32)
            idx = (tiles[:, None] * 32 + torch.arange(32)[None, :]).reshape(-1)
            live[idx[idx < P]] = True
        for k in range(3):
            x0, x1 = res[0][k][g].reshape(Cc, P), res[1][k][g].reshape(Cc, P)
            assert torch.equal(x1[:, live], x0[:, live]), (g, n, k)
    assert torch.equal(res[0][3], res[1][3]) and torch.equal(res[0][4], res[1][4])


@pytest.mark.parametrize('N,nvs', [(70, [70, 33]), (120, [120, 75, 1]), (128, [128, 127]), (129, [129, 64]), (200, [200, 131]), (256, [256, 255, 17])])
def test_colmax_16_lanes_per_row_first_index_on_ties(N, nvs):
    """64 < N <= 256 (colmax_fwd_rows16_kernel): values with many exact ties, ragged graphs, normalised input: value and FIRST arg-max
    column per valid row bit-exact against torch, zeros in the padding rows."""
    G, Cc = len(nvs), 3
    rng = torch.Generator().manual_seed(N)
    x = torch.randint(-3, 4, (G, Cc, N, N), generator=rng).float()
    nrm = torch.zeros(G, Cc, 4)
    nrm[..., 0] = torch.randint(-2, 3, (G, Cc), generator=rng).float()       # mean
    nrm[..., 1] = 2.0 ** torch.randint(-1, 2, (G, Cc), generator=rng).float()  # a (power of two: exact arithmetic)
    beta = torch.randint(-1, 2, (Cc,), generator=rng).float()
    xd, nd, bd = x.to(DEV), nrm.to(DEV).contiguous(), beta.to(DEV)
    nv = torch.tensor(nvs, dtype=torch.int32, device=DEV)
    e = torch.full((G, Cc, N), 7.0, device=DEV)
    idx = torch.full((G, Cc, N), 7, dtype=torch.int32, device=DEV)
    s = _slab(xd, nrm=nd, beta=bd)
    _lib.call('fgnn_colmax_fwd', C.byref(s), _lib.ptr(nv), G, N, _lib.ptr(e), _lib.ptr(idx), _lib.stream_ptr())
    torch.cuda.synchronize()
    y = (x - nrm[..., 0, None, None]) * nrm[..., 1, None, None] + beta[None, :, None, None]
    for g, n in enumerate(nvs):
        val = y[g, :, :n, :n].max(-1)[0]
        first = (y[g, :, :n, :n] == val.unsqueeze(-1)).float().argmax(-1)
        assert torch.equal(e[g, :, :n].cpu(), val), (g, n)
        assert torch.equal(idx[g, :, :n].cpu().long(), first), (g, n)
        assert e[g, :, n:].abs().sum().item() == 0 and idx[g, :, n:].abs().sum().item() == 0


def test_colmax_first_index_on_ties_bit_exact():
    G, Cc, N = 2, 3, 11
    x = torch.randint(-3, 4, (G, Cc, N, N)).float()      # many exact ties
    xd = x.to(DEV)
    e = torch.empty(G, Cc, N, device=DEV)
    idx = torch.empty(G, Cc, N, dtype=torch.int32, device=DEV)
    s = _slab(xd)
    _lib.call('fgnn_colmax_fwd', C.byref(s), None, G, N, _lib.ptr(e), _lib.ptr(idx), _lib.stream_ptr())
    # first-max index (what torch.max on CPU returns for the reference) = N-1 - argmax of the reversed row
    val = x.max(-1)[0]
    first = (x == val.unsqueeze(-1)).float().argmax(-1)
    assert torch.equal(e.cpu(), val)
    assert torch.equal(idx.cpu().long(), first)
    de = torch.randn(G, Cc, N)
    dy = torch.empty(G, Cc, N, N, device=DEV)
    _lib.call('fgnn_colmax_bwd', _lib.ptr(de.to(DEV)), _lib.ptr(idx), None, G, Cc, N, _lib.ptr(dy), Cc * N * N, N * N,
              None, None, _lib.stream_ptr())
    ref = torch.zeros(G, Cc, N, N).scatter_(-1, first.unsqueeze(-1), de.unsqueeze(-1))
    assert torch.equal(dy.cpu(), ref)


@pytest.mark.parametrize('B,Cc,N,ragged', [(32, 32, 50, False), (3, 32, 64, True), (5, 8, 9, True), (1, 32, 1, False), (8, 32, 33, True)])
def test_score_ce_step_is_bit_identical_to_the_two_launches(B, Cc, N, ragged):
    """fgnn_score_ce_step (scoring + triplet loss + their backward in one launch, what FgnnEngine.step issues) against
    fgnn_score_ce_fwd_blocks + fgnn_score_ce_bwd: every output bit for bit, ragged batches with empty graphs included."""
    lib = _lib.load()
    assert lib.fgnn_score_ce_step_supported(B, Cc, N)
    g = torch.Generator().manual_seed(B * 100 + N)
    e1d, e2d = torch.randn(B, Cc, N, generator=g).to(DEV), torch.randn(B, Cc, N, generator=g).to(DEV)
    nv = None
    if ragged:
        nv = torch.randint(0, N + 1, (B,), generator=g).to(torch.int32)
        nv[0] = N
        nv = nv.to(DEV)
    blocks = lib.fgnn_score_row_blocks(B, N)
    gscale = torch.tensor([1.0 / 1234.0], device=DEV)
    outs = []
    for fused in (False, True):
        scores, lse = torch.full((B, N, N), 7.0, device=DEV), torch.full((B, N), 7.0, device=DEV)
        pl = torch.full((B * blocks,), 7.0, device=DEV)
        d1, d2 = torch.full_like(e1d, 7.0), torch.full_like(e2d, 7.0)
        if fused:
            _lib.call('fgnn_score_ce_step', _lib.ptr(e1d), _lib.ptr(e2d), _lib.ptr(nv), _lib.ptr(gscale), B, Cc, N, blocks, _lib.ptr(scores),
                      _lib.ptr(lse), _lib.ptr(pl), _lib.ptr(d1), _lib.ptr(d2), _lib.stream_ptr())
        else:
            _lib.call('fgnn_score_ce_fwd_blocks', _lib.ptr(e1d), _lib.ptr(e2d), _lib.ptr(nv), B, Cc, N, blocks, _lib.ptr(scores),
                      _lib.ptr(lse), _lib.ptr(pl), _lib.stream_ptr())
            _lib.call('fgnn_score_ce_bwd', _lib.ptr(e1d), _lib.ptr(e2d), _lib.ptr(scores), _lib.ptr(lse), _lib.ptr(nv), _lib.ptr(gscale),
                      B, Cc, N, _lib.ptr(d1), _lib.ptr(d2), _lib.stream_ptr())
        torch.cuda.synchronize()
        outs.append((scores, lse, pl, d1, d2))
    for a, b, name in zip(outs[0], outs[1], ('scores', 'lse', 'pair_loss', 'de1', 'de2')):
        assert torch.equal(a, b), name
    assert not lib.fgnn_score_ce_step_supported(64, 32, 50) and not lib.fgnn_score_ce_step_supported(8, 32, 65)


@pytest.mark.parametrize('B,Cc,N', [(3, 32, 50), (2, 8, 9)])
def test_score_and_ce(B, Cc, N):
    e1 = torch.randn(B, Cc, N)
    e2 = torch.randn(B, Cc, N)
    e1d, e2d = e1.to(DEV), e2.to(DEV)
    scores = torch.empty(B, N, N, device=DEV)
    lse = torch.empty(B, N, device=DEV)
    pl = torch.empty(B * _lib.FGNN_SCORE_SPLIT, device=DEV)
    _lib.call('fgnn_score_ce_fwd', _lib.ptr(e1d), _lib.ptr(e2d), None, B, Cc, N, _lib.ptr(scores), _lib.ptr(lse),
              _lib.ptr(pl), _lib.stream_ptr())
    a, b = e1.double().requires_grad_(True), e2.double().requires_grad_(True)
    sref = torch.matmul(a.transpose(1, 2), b)
    lref = sum(torch.nn.functional.cross_entropy(s, torch.arange(N), reduction='sum') for s in sref)
    assert rel(scores.cpu(), sref) < 2e-6
    assert rel(pl.sum().cpu(), lref) < 2e-6
    gscale = torch.tensor([0.37], device=DEV)
    d1, d2 = torch.empty_like(e1d), torch.empty_like(e2d)
    _lib.call('fgnn_score_ce_bwd', _lib.ptr(e1d), _lib.ptr(e2d), _lib.ptr(scores), _lib.ptr(lse), None,
              _lib.ptr(gscale), B, Cc, N, _lib.ptr(d1), _lib.ptr(d2), _lib.stream_ptr())
    (lref * 0.37).backward()
    assert rel(d1.cpu(), a.grad) < 5e-6 and rel(d2.cpu(), b.grad) < 5e-6


def test_gn_stats_two_pass_is_shift_robust():
    """Large mean / small variance: the tile-wise Chan combination must not cancel."""
    G, Cc, N = 2, 4, 50
    x = (1000.0 + 0.01 * torch.randn(G, Cc, N, N)).float()
    xd = x.to(DEV)
    nrm = torch.empty(G * Cc * 4, device=DEV)
    P = N * N
    _lib.call('fgnn_gn_stats', _lib.ptr(xd), Cc * P, P, None, None, G, Cc, N, 1e-5, _lib.ptr(nrm), _lib.stream_ptr())
    nrm = nrm.cpu().view(G, Cc, 4)
    mean = x.double().mean((-1, -2))
    var = x.double().var((-1, -2), unbiased=False)
    assert rel(nrm[..., 0], mean) < 1e-6
    assert rel(nrm[..., 3], 1.0 / (var + 1e-5)) < 1e-3


def test_unsupported_shapes_fail_loudly():
    args = _lib.MlpFwdArgs()
    args.G, args.N, args.depth, args.nmlp = 1, 4, 3, 1
    with pytest.raises(RuntimeError, match='fgnn_mlp_fwd'):
        _lib.call('fgnn_mlp_fwd', C.byref(args), _lib.stream_ptr())


def test_adam_step_matches_torch_adam():
    from graph_neural_net_amd.optim import FlatAdam
    g = torch.Generator().manual_seed(0)
    p0 = torch.randn(40000, generator=g)
    ref = p0.clone().requires_grad_(True)
    opt_ref = torch.optim.Adam([ref], lr=1e-3, amsgrad=False)
    pd = p0.clone().to(DEV)
    opt = FlatAdam(pd, lr=1e-3)
    for t in range(5):
        grad = torch.randn(40000, generator=g) * (10.0 ** (t - 2))
        ref.grad = grad.clone()
        opt_ref.step()
        opt.step(grad.to(DEV))
        assert rel(pd.cpu(), ref.detach()) < 1e-6, t
    assert rel(opt.exp_avg.cpu(), opt_ref.state[ref]['exp_avg']) < 1e-6
    assert rel(opt.exp_avg_sq.cpu(), opt_ref.state[ref]['exp_avg_sq']) < 1e-6


def test_accuracy_max_bit_exact():
    import numpy as np
    from graph_neural_net_amd.masked import from_list
    from graph_neural_net_amd.metrics import accuracy_max
    g = torch.Generator().manual_seed(1)
    w = torch.randint(-2, 3, (5, 50, 50), generator=g).float()       # ties on purpose
    w[0] = torch.eye(50) * 10
    acc, n = accuracy_max(w.to(DEV))
    ref = sum(int(np.sum(np.argmax(x.numpy(), 1) == np.arange(50))) for x in w)
    assert (acc, n) == (ref, 250)
    lst = [torch.randn(k, k, generator=g) for k in (7, 12, 9)]
    acc, n = accuracy_max(from_list([t.to(DEV) for t in lst], dims=(0, 1)))
    ref = sum(int(np.sum(np.argmax(x.numpy(), 1) == np.arange(len(x)))) for x in lst)
    assert (acc, n) == (ref, 28)


def test_expand_adjacency_bit_exact():
    import numpy as np
    from graph_neural_net_amd import synthetic
    from graph_neural_net_amd.inputs import expand_adjacency
    # the reference's own representation (fixture generated by calling its adjacency_matrix_to_tensor_representation)
    from util import load_golden
    d = load_golden('input_contract.npz')
    for i in sorted({int(kk.split('/')[1]) for kk in d if kk.startswith('w/')}):
        w = d['w/%d' % i].numpy()
        n = w.shape[0]
        bits = torch.from_numpy(synthetic.pack_adjacency(w[None]).view(np.int32)).to(DEV)
        assert torch.equal(expand_adjacency(bits, n).cpu()[0], d['repr/%d' % i])
    rng = np.random.default_rng(0)
    for n in (50, 33, 7):
        ws = np.stack([synthetic.erdos_renyi(rng, n, 0.3) for _ in range(5)])
        ref = np.stack([synthetic.tensor_representation(w) for w in ws])
        bits = torch.from_numpy(synthetic.pack_adjacency(ws).view(np.int32)).to(DEV)
        x = expand_adjacency(bits, n)
        assert torch.equal(x.cpu(), torch.from_numpy(ref))
    # ragged: graphs of 20 and 41 vertices padded to 41, garbage bits in the padding must be ignored
    n = 41
    small, big = synthetic.erdos_renyi(rng, 20, 0.5), synthetic.erdos_renyi(rng, n, 0.5)
    ws = np.ones((2, n, n), dtype=np.float32)
    ws[0, :20, :20] = small
    ws[1] = big
    ref = np.zeros((2, 2, n, n), dtype=np.float32)
    ref[0, :, :20, :20] = synthetic.tensor_representation(small)
    ref[1] = synthetic.tensor_representation(big)
    bits = torch.from_numpy(synthetic.pack_adjacency(ws).view(np.int32)).to(DEV)
    x = expand_adjacency(bits, n, nvalid=torch.tensor([20, n], dtype=torch.int32))
    assert torch.equal(x.cpu(), torch.from_numpy(ref))


@pytest.mark.parametrize('N', [1, 7, 32, 33, 34, 40, 49, 50, 58, 64])
def test_wave_per_matrix_forward_matmul_is_bit_identical_to_the_workgroup_kernel(N):
    """The shipped N <= 64 forward (one wave per matrix; for even N > 32 with 8-byte accesses and the even / odd column blocks of
    chan_matmul_fwd_w2_kernel) keeps the k-step order and the normalisation expression of the workgroup-per-matrix kernel it
    replaced: outputs must be equal bit for bit (dense and ragged with odd and even vertex counts, normalised operands; the
    untouched output is poisoned with NaN first).  Variants: 1 = shipped, 9 = the four-byte wave kernel, 0 = the workgroup kernel, 17 = two matrices per wave (a measurement build, N = 49 ... 56)."""
    lib = _lib.load()
    G, Cc = 3, 5
    g = torch.Generator().manual_seed(N)
    a = torch.randn(G, Cc, N, N, generator=g).to(DEV)
    b = torch.randn(G, Cc, N, N, generator=g).to(DEV)
    nrm_a = (torch.rand(G, Cc, 4, generator=g) + 0.5).to(DEV)
    nrm_b = (torch.rand(G, Cc, 4, generator=g) + 0.5).to(DEV)
    beta = torch.randn(Cc, generator=g).to(DEV)
    nv = torch.tensor([N, max(1, N // 2), max(0, N - 1)], dtype=torch.int32, device=DEV)
    # ragged planes: what lies outside a graph's valid corner is garbage the wave kernels must not look at (the workgroup kernel
    # they replaced gets the same operands with that garbage zeroed)
    ap, bp = a.clone(), b.clone()
    for gi, n in enumerate(nv.tolist()):
        for t, fill in ((ap, float('nan')), (bp, float('nan')), (a, 0.0), (b, 0.0)):
            t[gi, :, n:, :] = fill
            t[gi, :, :, n:] = fill
    res = []
    try:
        for variant in (1, 0, 9, 17):
            lib.fgnn_debug_matmul_variant(variant)
            out = torch.full((G, Cc, N, N), float('nan'), device=DEV)
            ua, ub = (a, b) if variant == 0 else (ap, bp)
            sa, sb = _slab(ua, nrm=nrm_a, beta=beta), _slab(ub, nrm=nrm_b, beta=beta)
            _lib.call('fgnn_chan_matmul_fwd', C.byref(sa), C.byref(sb), _lib.ptr(nv), G, N, _lib.ptr(out), Cc * N * N, N * N,
                      _lib.stream_ptr())
            torch.cuda.synchronize()
            res.append(out.cpu())
    finally:
        lib.fgnn_debug_matmul_variant(1)
    assert torch.equal(res[0], res[1]) and torch.equal(res[0], res[2]) and torch.equal(res[0], res[3])      # (17: two matrices per wave, N = 49 ... 56)
    # padding rows / columns of the ragged graphs are exact zeros
    for gi, n in enumerate(nv.tolist()):
        assert float(res[0][gi, :, n:, :].abs().max()) == 0.0 if n < N else True
        assert float(res[0][gi, :, :, n:].abs().max()) == 0.0 if n < N else True

@pytest.mark.parametrize('N', [1, 31, 32, 33, 50, 64, 65, 97, 200])
def test_pack_tensor_representation_round_trip_and_check(N):
    """inputs.pack_tensor_representation (dense loader batch -> bit-packed adjacency): the inverse of expand_adjacency bit for bit,
    ragged sizes included; a tensor that is not a tensor representation (a 0.5 in channel 0, a wrong degree, an off-diagonal entry
    in channel 1) is refused."""
    from graph_neural_net_amd import inputs, synthetic
    rng = np.random.default_rng(N)
    G = 4
    w = (rng.random((G, N, N)) < 0.4).astype(np.float32)            # directed, with self loops
    bits = torch.from_numpy(synthetic.pack_adjacency(w).view(np.int32)).to(DEV)
    x = inputs.expand_adjacency(bits, N)
    back = inputs.pack_tensor_representation(x)
    assert torch.equal(back, bits)
    nv = torch.tensor([N, max(N // 2, 1), 1, 0], dtype=torch.int32, device=DEV)
    xr = inputs.expand_adjacency(bits, N, nvalid=nv)
    br = inputs.pack_tensor_representation(xr, nvalid=nv)
    ref = synthetic.pack_adjacency(np.stack([np.pad(w[g, :n, :n], ((0, N - n), (0, N - n))) for g, n in enumerate(nv.tolist())]))
    assert torch.equal(br.cpu(), torch.from_numpy(ref.view(np.int32)))
    if N >= 2:
        for bad in ('half', 'degree', 'offdiag'):
            y = x.clone()
            if bad == 'half':
                y[1, 0, 0, N - 1] = 0.5
            elif bad == 'degree':
                y[2, 1, N - 1, N - 1] += 1.0
            else:
                y[0, 1, 0, 1] = 1.0
            with pytest.raises(RuntimeError):
                inputs.pack_tensor_representation(y)
            b2, flag = inputs.pack_tensor_representation(y, check='device')
            assert int(flag.item()) == 1
        _, flag = inputs.pack_tensor_representation(x, check='device')
        assert int(flag.item()) == 0


@pytest.mark.parametrize('Nin,N,ragged', [(50, 50, False), (33, 40, True), (97, 104, True), (1, 8, True), (64, 64, True)])
def test_pack_adjacency_pair_equals_the_two_launches_it_replaces(Nin, N, ragged):
    """fgnn_pack_adjacency_pair (both sides of a siamese batch in one launch, loaders/loaders.py:12-15) against two fgnn_pack_adjacency_ld
    launches + the copy of the vertex counts + fgnn_inv_node_count: words, counts, 1 / sum(n) and the verdict flag bit for bit; a bad entry
    on EITHER side raises the flag."""
    from graph_neural_net_amd import inputs, synthetic
    rng = np.random.default_rng(7 * Nin + N)
    B = 3
    nv1 = torch.tensor([Nin, max(Nin // 2, 1), 1], dtype=torch.int32, device=DEV) if ragged else None
    nv2 = torch.tensor([max(Nin - 1, 1), Nin, max(Nin // 3, 1)], dtype=torch.int32, device=DEV) if ragged else None
    xs = []
    for nv in (nv1, nv2):
        w = (rng.random((B, Nin, Nin)) < 0.4).astype(np.float32)
        bits = torch.from_numpy(synthetic.pack_adjacency(w).view(np.int32)).to(DEV)
        xs.append(inputs.expand_adjacency(bits, Nin, nvalid=nv).contiguous())
    words = (N + 31) // 32

    def two(x1, x2):
        bits = torch.full((2 * B, N, words), -1, dtype=torch.int32, device=DEV)
        flag = torch.zeros(1, dtype=torch.int32, device=DEV)
        for half, x, nv in ((bits[:B], x1, nv1), (bits[B:], x2, nv2)):
            _lib.call('fgnn_pack_adjacency_ld', _lib.ptr(x), _lib.ptr(nv) if ragged else None, B, Nin, N, _lib.ptr(half), _lib.ptr(flag),
                      _lib.stream_ptr())
        return bits, flag

    def one(x1, x2):
        bits = torch.full((2 * B, N, words), -1, dtype=torch.int32, device=DEV)
        flag = torch.zeros(1, dtype=torch.int32, device=DEV)
        nvo = torch.full((2 * B,), -7, dtype=torch.int32, device=DEV)
        inv = torch.full((1,), -7.0, device=DEV)
        _lib.call('fgnn_pack_adjacency_pair', _lib.ptr(x1), _lib.ptr(x2), _lib.ptr(nv1) if ragged else None, _lib.ptr(nv2) if ragged else None,
                  B, Nin, N, _lib.ptr(bits), _lib.ptr(nvo) if ragged else None, _lib.ptr(inv) if ragged else None, _lib.ptr(flag),
                  _lib.stream_ptr())
        return bits, flag, nvo, inv

    b2, f2 = two(*xs)
    b1, f1, nvo, inv = one(*xs)
    torch.cuda.synchronize()
    assert torch.equal(b1, b2) and int(f1.item()) == int(f2.item()) == 0
    if ragged:
        assert torch.equal(nvo, torch.cat([nv1, nv2]))
        ref = torch.empty(1, device=DEV)
        _lib.call('fgnn_inv_node_count', _lib.ptr(nv1), B, _lib.ptr(ref), _lib.stream_ptr())
        assert torch.equal(inv, ref)
    if Nin >= 2:
        for side in (0, 1):
            ys = [x.clone() for x in xs]
            ys[side][0, 0, 0, 1] = 0.5
            _, f, _, _ = one(*ys)
            assert int(f.item()) == 1
