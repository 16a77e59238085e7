"""GPU: the fused HIP path (through the C ABI) against the golden vectors of the reference
and against the oracle on the same seeded inputs.

Tolerances (max-norm relative error): the reference's own fp32
result differs from its fp64 result by ~1e-5 on scores and up to ~2e-3 on early-layer
gradients, so gradients are gated against the fp64 golden with the reference's own
fp32-vs-fp64 error as the yard-stick (per tensor we must be within 4x that + 1e-5,
and over the whole flat gradient no worse than 2x the reference's own error).
"""
import numpy as np
import pytest
import torch

from graph_neural_net_amd import synthetic
from graph_neural_net_amd.engine import FgnnEngine, ParamLayout
from oracle import fgnn_oracle as O
from util import is_zero_grad, load_golden, rel, sub, unpack_pairs

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
# Batch-level gradient gates of the large fixtures (see test_cfg2_full_batch_against_golden), derived from the multi-seed fixture
# (tests/golden/gradgate_single_pairs.npz, group A = the 32 pairs of the benchmarked batch; tests/gradgate.py): the error of a
# batch gradient is the rms of its pairs' errors, and between two equally valid fp32 evaluations of the reference (8 threads /
# 1 thread, and the two HIP engines judged the same way) the rms over 32 resampled pairs differs by 1.9 x at the 90th, 3.2 x at
# the 99th and 4.4 x at the 99.9th percentile (20 000 bootstrap batches) -> BIG_FLAT = 4; per tensor, the reference's own two
# runs differ by up to 10.3 x at batch level (rms over the pairs; median 1.4 x, 90 % 2.9 x) -> BIG_TENSOR = 10.
BIG_FLAT, BIG_TENSOR = 4.0, 10.0
OP_TOL = 1e-5          # per-block forward / scores at small depth
E2E_FWD_TOL = 3e-5     # block-4 activations / scores after 4 blocks (reference fp32-vs-fp64: 1.6e-5 / 1.4e-5)


def _run_engine(sd, x1, x2, nblk, nvalid=None):
    lay = ParamLayout(2, nblk, 32, 32, 3)
    params = lay.flatten(sd, DEV)
    grads = torch.zeros_like(params)
    B = x1.shape[0]
    eng = FgnnEngine(lay, 2 * B, x1.shape[-1], DEV, ragged=nvalid is not None)
    x = torch.cat([x1, x2]).contiguous().to(DEV)
    nv = None if nvalid is None else torch.cat([nvalid, nvalid]).to(DEV)
    scores, loss = eng.step(params, grads, x, nvalid=nv)
    torch.cuda.synchronize()
    return eng, params, lay, scores.cpu(), loss.item(), lay.unflatten(grads.cpu())


def _check_grads(got, d, flat_factor=2.0, tensor_factor=4.0, second_sample=None):
    """Gradients against the fp64 truth, in units of the reference's own fp32 error.  `second_sample`: another equally
    valid fp32 evaluation of the reference's op sequence (the pinned oracle run with one thread = another summation
    order).  After four blocks the fp32 error of a tensor is dominated by rare ReLU / arg-max flips, so it is a heavy-tailed
    random variable: at N=200 the 8-thread and the 1-thread oracle runs differ by 20x on individual tensors
    (tests/diag/gpu_graderr_table.py).  Where a second sample is given the yard-stick is the larger of the two."""
    keys = [k for k in sub(d, 'grad/') if not is_zero_grad(k)]
    flat = lambda pick: torch.cat([pick(k).reshape(-1).double() for k in keys])
    g64 = flat(lambda k: d['grad64/' + k])
    ours = (flat(lambda k: got[k]) - g64).norm() / g64.norm()
    theirs = (flat(lambda k: d['grad/' + k]) - g64).norm() / g64.norm()
    if second_sample is not None:
        theirs = max(theirs, (flat(lambda k: second_sample[k]) - g64).norm() / g64.norm())
    assert ours < flat_factor * theirs + 1e-6, (ours, theirs)      # whole-gradient L2 error vs the fp64 truth
    for k, ref in sub(d, 'grad/').items():
        if is_zero_grad(k):
            assert got[k].abs().max() < 1e-4, k
            continue
        ref64 = d['grad64/' + k]
        yard = rel(ref, ref64)                    # the reference's own fp32 error
        if second_sample is not None:
            yard = max(yard, rel(second_sample[k], ref64))
        assert rel(got[k], ref64) < tensor_factor * yard + 1e-5, (k, rel(got[k], ref64), yard)


def _one_thread_sample(x1, x2, sd, ragged=False):
    """The oracle's fp32 gradients evaluated with a single thread (different GEMM blocking / summation order);
    ragged: x1 / x2 are lists of per-graph tensors."""
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        return (O.step_fwd_bwd_ragged if ragged else O.step_fwd_bwd)(x1, x2, sd)[2]
    finally:
        torch.set_num_threads(n)


def test_cfg1_against_golden():
    d = load_golden('cfg1_er_n20_b4_1blk.npz')
    eng, params, lay, scores, loss, grads = _run_engine(sub(d, 'sd/'), d['x1'], d['x2'], 1)
    B = d['x1'].shape[0]
    for j in (1, 2, 3):
        y = eng.normalized(1, j, params).cpu()[:B]
        assert rel(y, d['inter/ne/bm/block1/mlp%d' % j]) < OP_TOL
    assert rel(eng.unpadded(eng.mult[1]).cpu()[:B], d['inter/ne/bm/block1/mult']) < OP_TOL
    assert rel(eng.E.cpu()[:B], d['inter/ne/suffix']) < OP_TOL
    assert rel(scores, d['scores']) < OP_TOL
    assert abs(loss - d['loss'].item()) < 1e-5 * abs(d['loss'].item())
    _check_grads(grads, d)
    for k, ref in sub(d, 'grad/').items():      # 1 block: also tight against the fp32 golden
        if not is_zero_grad(k):
            assert rel(grads[k], ref) < 2e-5, k


def test_cfg2_against_golden():
    d = load_golden('cfg2_reg_n50_b2_4blk.npz')
    eng, params, lay, scores, loss, grads = _run_engine(sub(d, 'sd/'), d['x1'], d['x2'], 4)
    assert rel(eng.normalized(1, 3, params).cpu()[:1], d['inter/ne/bm/block1/mlp3']) < OP_TOL
    assert rel(eng.normalized(4, 3, params).cpu()[:1], d['inter/ne/bm/block4/mlp3']) < E2E_FWD_TOL
    assert rel(eng.E.cpu()[:1], d['inter/ne/suffix']) < E2E_FWD_TOL
    assert rel(scores, d['scores']) < E2E_FWD_TOL
    assert rel(scores, d['scores64']) < E2E_FWD_TOL
    assert abs(loss - d['loss'].item()) < 1e-5 * abs(d['loss'].item())
    _check_grads(grads, d)


def test_ragged_against_golden():
    d = load_golden('ragged_er_b4_2blk.npz')
    ns = [int(v) for v in d['ns']]
    nmax = max(ns)
    xs = [d['x1/%d' % i] for i in range(len(ns))]
    ys = [d['x2/%d' % i] for i in range(len(ns))]
    x1, nv = O.pad_graph_list(xs)
    x2, _ = O.pad_graph_list(ys)
    eng, params, lay, scores, loss, grads = _run_engine(sub(d, 'sd/'), x1, x2, 2, nvalid=nv)
    E = eng.E.cpu()
    for i, n in enumerate(ns):
        assert rel(E[i, :, :n], d['e1/%d' % i]) < OP_TOL
        assert rel(E[len(ns) + i, :, :n], d['e2/%d' % i]) < OP_TOL
        assert rel(scores[i, :n, :n], d['scores/%d' % i]) < OP_TOL
        # padding is exactly zero (MaskedTensor invariant), bit-exact
        assert E[i, :, n:].abs().sum() == 0 and scores[i, n:, :].abs().sum() == 0 and scores[i, :, n:].abs().sum() == 0
    y = eng.normalized(2, 3, params).cpu()
    for i, n in enumerate(ns):
        assert y[i, :, n:, :].abs().sum() == 0 and y[i, :, :, n:].abs().sum() == 0
    assert abs(loss - d['loss'].item()) < 1e-5 * abs(d['loss'].item())
    for k, ref in sub(d, 'grad/').items():
        if is_zero_grad(k):
            assert grads[k].abs().max() < 1e-4
        else:
            assert rel(grads[k], ref) < 1e-4, (k, rel(grads[k], ref))


def test_seeded_random_against_oracle():
    """ER pairs, N=23 (odd N, N*N not a multiple of the tile), 2 blocks, perturbed affine."""
    torch.manual_seed(3)
    sd = O.init_state_dict(num_blocks=2)
    g = torch.Generator().manual_seed(9)
    for k in sd:
        if k.endswith('.bias') and sd[k].dim() == 1:
            sd[k] = 0.1 * torch.randn(sd[k].shape, generator=g)
        elif k.endswith('gn.weight'):
            sd[k] = sd[k] * (1 + 0.3 * torch.randn(sd[k].shape, generator=g))
        elif k.endswith('gn.bias'):
            sd[k] = 0.05 * torch.randn(sd[k].shape, generator=g)
    x1, x2 = synthetic.make_batch(5, 3, 23, 'ErdosRenyi', 0.3, 0.1)
    s_ref, l_ref, g_ref = O.step_fwd_bwd(x1, x2, sd)
    _, _, _, scores, loss, grads = _run_engine(sd, x1, x2, 2)
    assert rel(scores, s_ref) < OP_TOL
    assert abs(loss - l_ref.item()) < 1e-5 * abs(l_ref.item())
    for k, v in g_ref.items():
        if not is_zero_grad(k):
            assert rel(grads[k], v) < 1e-4, (k, rel(grads[k], v))


def test_full_size_properties():
    """BASELINE config 1 at full size (B=32, N=50, 4 blocks): size-independent properties."""
    lay = ParamLayout(2, 4, 32, 32, 3)
    params = lay.init_flat(0, DEV)
    x1, x2 = synthetic.make_batch(2000, 32, 50, 'Regular', 0.2, 0.1)
    x = torch.cat([x1, x2]).contiguous().to(DEV)
    eng = FgnnEngine(lay, 64, 50, DEV)
    g1 = torch.zeros_like(params)
    s1, l1 = eng.step(params, g1, x)
    s1, l1 = s1.clone(), l1.clone()
    E1 = eng.E.clone()
    # (1) run-to-run determinism: bit-exact (fixed-order reductions, no atomics)
    g2 = torch.zeros_like(params)
    s2, l2 = eng.step(params, g2, x)
    assert torch.equal(s1, s2) and torch.equal(l1, l2) and torch.equal(g1, g2)
    assert torch.isfinite(g1).all() and torch.isfinite(s1).all()
    # (2) graphs are independent: a sub-batch gives bit-identical embeddings
    eng8 = FgnnEngine(lay, 16, 50, DEV)
    xs = torch.cat([x1[:8], x2[:8]]).contiguous().to(DEV)
    eng8.embed(params, xs)
    assert torch.equal(eng8.E[:8], E1[:8]) and torch.equal(eng8.E[8:], E1[32:40])
    # (3) vertex-permutation equivariance of the embedder (per-graph relabelling permutes columns)
    perm = torch.randperm(50, generator=torch.Generator().manual_seed(1))
    xp = x1[:8][:, :, perm][:, :, :, perm]
    eng8.embed(params, torch.cat([xp, x2[:8]]).contiguous().to(DEV))
    assert rel(eng8.E[:8].cpu(), E1[:8].cpu()[:, :, perm]) < 1e-4
    # (4) data-parallel decomposition: sum of shard gradients (global normaliser) == full gradient
    acc = torch.zeros_like(params)
    for lo in range(0, 32, 8):
        xs = torch.cat([x1[lo:lo + 8], x2[lo:lo + 8]]).contiguous().to(DEV)
        gs = torch.zeros_like(params)
        eng8.step(params, gs, xs, total_nodes=32 * 50)
        acc += gs
    assert rel(acc.cpu(), g1.cpu()) < 1e-5
    # (5) oracle spot check on 2 pairs of the same batch
    sd = {k: v.clone() for k, v in lay.unflatten(params.cpu()).items()}
    s_ref = O.siamese_scores(x1[:2], x2[:2], sd)
    assert rel(s1[:2].cpu(), s_ref) < E2E_FWD_TOL


def _score_gate(ours, ref32, ref64):
    """scores: within max(3e-5, 2x the reference's own fp32-vs-fp64 error) of the fp64 truth"""
    yard = rel(ref32, ref64)
    assert rel(ours, ref64) < max(E2E_FWD_TOL, 2 * yard), (rel(ours, ref64), yard)


def test_cfg2_full_batch_against_golden():
    """BASELINE config 1 at the benchmarked size (B=32, N=50, 4 blocks): scores, loss and every gradient against the
    reference-generated fixture, gradients with the reference's own fp32-vs-fp64 error as the yard-stick."""
    d = load_golden('cfg2_reg_n50_b32_4blk.npz')
    sd = sub(load_golden('cfg2_reg_n50_b2_4blk.npz'), 'sd/')
    n = int(d['n'])
    x1, x2 = unpack_pairs(d['bits1'], n), unpack_pairs(d['bits2'], n)
    assert torch.equal(x1, synthetic.make_batch(2000, 32, 50, 'Regular', 0.2, 0.1)[0])      # the benchmarked batch
    eng, params, lay, scores, loss, grads = _run_engine(sd, x1, x2, 4)
    _score_gate(scores, d['scores'], d['scores64_as_f32'])
    assert rel(scores, d['scores']) < E2E_FWD_TOL
    assert abs(loss - d['loss'].item()) < 1e-5 * abs(d['loss'].item())
    # The fp32-vs-fp64 gradient error of a pair is bimodal (next test): ~1e-5 for most pairs, ~1e-3 (up to 9e-3) for the few
    # that hold a ReLU / arg-max decision within rounding distance of a tie, and the two fp32 evaluations hit different
    # pairs.  A batch figure -- of EITHER implementation -- is set by the two or three unlucky pairs it happens to hold, so
    # the batch-level gates against one realisation of the reference are looser (BIG_*) than the small-case gates (2x / 4x),
    # and the per-case distribution gate of tests/test_gpu_grad_gate.py is the sharp statement.
    _check_grads(grads, d, BIG_FLAT, BIG_TENSOR)


# (The per-pair statement -- the error distribution over single pairs matches the reference's, for ALL 32 pairs of the benchmarked
# batch and 116 more cases, with the reference's own flipped decisions and margins on record -- is tests/test_gpu_grad_gate.py.)


def test_cfg4_shape_n200_dense_er_fp32_golden():
    """BASELINE config 3 shape (N=200 dense ER, p=0.5) in fp32 against the reference-generated fixture (fp32 + fp64):
    exercises the whole-matrix matmul, the blocked score backward and the generic pooling kernels."""
    d = load_golden('cfg4_er_n200_b1_4blk.npz')
    sd = sub(load_golden('cfg2_reg_n50_b2_4blk.npz'), 'sd/')
    n = int(d['n'])
    x1, x2 = unpack_pairs(d['bits1'], n), unpack_pairs(d['bits2'], n)
    eng, params, lay, scores, loss, grads = _run_engine(sd, x1, x2, 4)
    _score_gate(scores, d['scores'], d['scores64_as_f32'])
    assert abs(loss - d['loss'].item()) < 1e-5 * abs(d['loss'].item())
    _check_grads(grads, d, BIG_FLAT, BIG_TENSOR, second_sample=_one_thread_sample(x1, x2, sd))


def test_cfg4_shape_n200_batch8_fp32():
    """The same shape at the config's batch (8 pairs) against the oracle run here in fp32 and fp64."""
    sd = sub(load_golden('cfg2_reg_n50_b2_4blk.npz'), 'sd/')
    x1, x2 = synthetic.make_batch(4000, 8, 200, 'ErdosRenyi', 0.5, 0.1)
    s_ref, l_ref, g_ref = O.step_fwd_bwd(x1, x2, sd)
    s64, l64, g64 = O.step_fwd_bwd(x1.double(), x2.double(), {k: v.double() for k, v in sd.items()})
    _, _, _, scores, loss, grads = _run_engine(sd, x1, x2, 4)
    _score_gate(scores, s_ref, s64)
    assert abs(loss - l_ref.item()) < 1e-5 * abs(l_ref.item())
    d = {'grad/' + k: v for k, v in g_ref.items()}
    d.update({'grad64/' + k: v for k, v in g64.items()})
    _check_grads(grads, d, BIG_FLAT, BIG_TENSOR, second_sample=_one_thread_sample(x1, x2, sd))


def test_cfg5_shape_ragged_30_120_golden():
    """BASELINE config 4 shape: variable-N batch, n in [30, 120], against the reference-generated per-graph dense runs
    (fp32 + fp64 yard-stick)."""
    d = load_golden('ragged_er_n30_120_b4_4blk.npz')
    sd = sub(load_golden('cfg2_reg_n50_b2_4blk.npz'), 'sd/')
    ns = [int(v) for v in d['ns']]
    xs = [unpack_pairs(d['bits1/%d' % i], n)[0] for i, n in enumerate(ns)]
    ys = [unpack_pairs(d['bits2/%d' % i], n)[0] for i, n in enumerate(ns)]
    x1, nv = O.pad_graph_list(xs)
    x2, _ = O.pad_graph_list(ys)
    eng, params, lay, scores, loss, grads = _run_engine(sd, x1, x2, 4, nvalid=nv)
    for i, n in enumerate(ns):
        _score_gate(scores[i, :n, :n], d['scores/%d' % i], d['scores64_as_f32/%d' % i])
        assert scores[i, n:, :].abs().sum() == 0 and scores[i, :, n:].abs().sum() == 0
    assert abs(loss - d['loss'].item()) < 1e-5 * abs(d['loss'].item())
    _check_grads(grads, d, BIG_FLAT, BIG_TENSOR, second_sample=_one_thread_sample(xs, ys, sd, ragged=True))


def test_cfg5_shape_ragged_30_120_batch8():
    """8 pairs with n in [30, 120] against per-graph dense oracle runs in fp32 and fp64."""
    sd = sub(load_golden('cfg2_reg_n50_b2_4blk.npz'), 'sd/')
    xs, ys = synthetic.make_ragged_batch(5000, 8, 30, 120)
    s_ref, l_ref, g_ref = O.step_fwd_bwd_ragged(xs, ys, sd)
    s64, l64, g64 = O.step_fwd_bwd_ragged([x.double() for x in xs], [y.double() for y in ys], {k: v.double() for k, v in sd.items()})
    x1, nv = O.pad_graph_list(xs)
    x2, _ = O.pad_graph_list(ys)
    eng, params, lay, scores, loss, grads = _run_engine(sd, x1, x2, 4, nvalid=nv)
    for i, n in enumerate(nv.tolist()):
        _score_gate(scores[i, :n, :n], s_ref[i], s64[i])
        assert scores[i, n:, :].abs().sum() == 0 and scores[i, :, n:].abs().sum() == 0
    assert abs(loss - l_ref.item()) < 1e-5 * abs(l_ref.item())
    d = {'grad/' + k: v for k, v in g_ref.items()}
    d.update({'grad64/' + k: v for k, v in g64.items()})
    _check_grads(grads, d, BIG_FLAT, BIG_TENSOR)


@pytest.mark.parametrize('B,N', [(1, 1), (1, 2), (2, 3), (1, 31), (3, 33), (1, 64), (1, 65), (1, 97)])
def test_degenerate_and_boundary_shapes(B, N):
    """Smallest graphs (n = 1: zero variance everywhere), a single pair, and sizes straddling the tile (32),
    single-tile matmul (64) and whole-matrix matmul (65+) boundaries: scores, loss and gradients vs the oracle."""
    torch.manual_seed(100 + N)
    sd = O.init_state_dict(num_blocks=2)
    x1, x2 = synthetic.make_batch(9000 + N, B, N, 'ErdosRenyi', 0.5, 0.1)
    s_ref, l_ref, g_ref = O.step_fwd_bwd(x1, x2, sd)
    s64, l64, g64 = O.step_fwd_bwd(x1.double(), x2.double(), {k: v.double() for k, v in sd.items()})
    eng, params, lay, scores, loss, grads = _run_engine(sd, x1, x2, 2)
    assert torch.isfinite(scores).all() and all(torch.isfinite(g).all() for g in grads.values())
    # fp64 yard-stick as for the golden cases: scores within max(3e-5, 2x the oracle's own fp32 error), the flat gradient
    # within 4x the oracle's own fp32-vs-fp64 distance (absolute floors for n <= 2, where scores / gradients are 0 by symmetry)
    assert (scores - s64.float()).abs().max() < 1e-6 or _score_gate(scores, s_ref, s64.float()) is None
    assert abs(loss - l_ref.item()) <= 1e-5 * abs(l_ref.item()) + 1e-7
    keys = [k for k in g_ref if not is_zero_grad(k)]
    flat = lambda g: torch.cat([g[k].reshape(-1).double() for k in keys])
    a, b, t = flat(grads), flat(g_ref), flat(g64)
    # (n = 2: the exact gradient is 0 by symmetry; ours is fp32 rounding noise of ~1e-7 per entry, 4e-5 in norm)
    assert (a - t).norm() <= 4.0 * (b - t).norm() + 1e-6 * t.norm() + 1e-4, ((a - t).norm().item(), (b - t).norm().item(), t.norm().item())


def test_ragged_with_single_vertex_and_full_size_graphs():
    """A ragged batch mixing n = 1, n = 2 and n = Nmax graphs."""
    torch.manual_seed(77)
    sd = O.init_state_dict(num_blocks=2)
    rng = np.random.default_rng(77)
    xs, ys = [], []
    for n in (1, 40, 2, 17, 40):
        a, b = synthetic.make_pair(rng, n, 'ErdosRenyi', 0.5, 0.1)
        xs.append(torch.from_numpy(a)); ys.append(torch.from_numpy(b))
    s_ref, l_ref, g_ref = O.step_fwd_bwd_ragged(xs, ys, sd)
    x1, nv = O.pad_graph_list(xs)
    x2, _ = O.pad_graph_list(ys)
    eng, params, lay, scores, loss, grads = _run_engine(sd, x1, x2, 2, nvalid=nv)
    for i, n in enumerate(nv.tolist()):
        assert rel(scores[i, :n, :n], s_ref[i]) < 1e-4 or (scores[i, :n, :n] - s_ref[i]).abs().max() < 1e-6
        assert scores[i, n:, :].abs().sum() == 0 and scores[i, :, n:].abs().sum() == 0
    assert abs(loss - l_ref.item()) < 1e-5 * abs(l_ref.item())
    # gradients against the fp64 truth with the oracle's own fp32 error as yard-stick (as for the constant-size cases)
    _, _, g64 = O.step_fwd_bwd_ragged([x.double() for x in xs], [y.double() for y in ys], {k: v.double() for k, v in sd.items()})
    keys = [k for k in g_ref if not is_zero_grad(k)]
    flat = lambda g: torch.cat([g[k].reshape(-1).double() for k in keys])
    a, b, t = flat(grads), flat(g_ref), flat(g64)
    # (n = 1 and n = 2 graphs: zero variance under the GraphNorm, the eps-regularised 1 / sqrt amplifies fp32 noise -- the
    # batch-level factor of the large cases, BIG_TENSOR = 10, instead of 4; measured 8.6 x; the old gate was 5e-3 |g|,
    # i.e. 160 x looser)
    assert (a - t).norm() <= BIG_TENSOR * (b - t).norm() + 1e-6 * t.norm() + 1e-4, ((a - t).norm().item(), (b - t).norm().item(), t.norm().item())


def test_ragged_bucketed_step_equals_padded_batch_and_oracle():
    """Size-bucketed ragged step (one engine pass per size class) == per-graph dense oracle."""
    from graph_neural_net_amd.trainer import FgnnTrainer
    torch.manual_seed(8)
    sd = O.init_state_dict(num_blocks=2)
    lay = ParamLayout(2, 2, 32, 32, 3)
    xs, ys = synthetic.make_ragged_batch(8000, 10, 9, 70)
    s_ref, l_ref, g_ref = O.step_fwd_bwd_ragged(xs, ys, sd)
    tr = FgnnTrainer(lay, lay.flatten(sd, DEV))
    sizes = [x.shape[-1] for x in xs]
    buckets = tr.bucket_by_size(sizes, 16)
    assert len(buckets) > 2 and sorted(i for _, b in buckets for i in b) == list(range(len(xs)))
    loss, scores = tr.model_step_ragged([x.to(DEV) for x in xs], [y.to(DEV) for y in ys], granule=16)
    torch.cuda.synchronize()
    for i in range(len(xs)):
        assert scores[i].shape == s_ref[i].shape and rel(scores[i].cpu(), s_ref[i]) < 1e-4
    assert abs(loss.item() - l_ref.item()) < 1e-5 * abs(l_ref.item())
    # gradient yard-stick: the oracle's own fp32-vs-fp64 error on this batch (graphs down to 9 vertices)
    _, _, g64 = O.step_fwd_bwd_ragged([x.double() for x in xs], [y.double() for y in ys], {k: v.double() for k, v in sd.items()})
    grads = lay.unflatten(tr.grads.cpu())
    keys = [k for k in g_ref if not is_zero_grad(k)]
    flat = lambda g: torch.cat([g[k].reshape(-1).double() for k in keys])
    ours = (flat(grads) - flat(g64)).norm() / flat(g64).norm()
    theirs = (flat(g_ref) - flat(g64)).norm() / flat(g64).norm()
    assert ours < 2 * theirs + 1e-6, (ours, theirs)


def test_captured_training_step_equals_eager():
    """HIP-graph-captured trainer step (model work + device-side Adam step count) == eager trainer step:
    identical loss trajectory and parameters (the kernels are the same and deterministic)."""
    from graph_neural_net_amd.trainer import FgnnTrainer
    lay = ParamLayout(2, 2, 32, 32, 3)
    p0 = lay.init_flat(3, DEV)
    batches = [synthetic.make_batch(7000 + i, 4, 18, 'ErdosRenyi', 0.3, 0.05) for i in range(3)]
    out = []
    for capture in (False, True):
        tr = FgnnTrainer(lay, p0.clone(), lr=2e-3, capture=capture)
        losses = []
        for s in range(7):
            x1, x2 = batches[s % 3]
            loss, scores = tr.train_step(x1.to(DEV), x2.to(DEV))
            losses.append(loss.item())
            if s == 3:
                tr.opt.lr = 1e-3                  # a scheduler step between replays must be picked up
        out.append((losses, tr.params.clone(), tr.opt.t))
    (l0, p_eager, t0), (l1, p_graph, t1) = out
    assert t0 == t1 == 7
    assert l0 == l1
    assert torch.equal(p_eager, p_graph)


def test_training_step_matches_oracle_plus_torch_adam():
    """engine step + fused Adam tracks oracle autograd + torch.optim.Adam on the CPU: the loss of steps 1..4
    (which depends on the previously updated parameters) agrees, and training makes progress.  (Parameters
    whose gradients are fp32 noise are moved by +-lr in a noise-determined direction by Adam in BOTH
    implementations, so parameters are not compared entry-wise here; the Adam kernel itself is pinned
    against torch.optim.Adam on identical gradients in test_gpu_kernels.py.)"""
    from graph_neural_net_amd.trainer import FgnnTrainer
    torch.manual_seed(6)
    sd = O.init_state_dict(num_blocks=2)
    lay = ParamLayout(2, 2, 32, 32, 3)
    x1, x2 = synthetic.make_batch(6000, 4, 16, 'ErdosRenyi', 0.3, 0.05)
    tr = FgnnTrainer(lay, lay.flatten(sd, DEV), lr=1e-3)
    ref = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    opt = torch.optim.Adam(list(ref.values()), lr=1e-3)
    losses = []
    for _ in range(4):
        loss, _ = tr.train_step(x1.to(DEV), x2.to(DEV))
        opt.zero_grad()
        l_ref = O.triplet_loss_mean(O.siamese_scores(x1, x2, ref))
        l_ref.backward()
        opt.step()
        assert abs(loss.item() - l_ref.item()) < 5e-4 * abs(l_ref.item()), (loss.item(), l_ref.item())
        losses.append(loss.item())
    for _ in range(20):
        loss, _ = tr.train_step(x1.to(DEV), x2.to(DEV))
    assert loss.item() < losses[0]


@pytest.mark.parametrize('N,B,ragged', [(50, 4, False), (33, 3, False), (41, 3, True)])
def test_bit_packed_adjacency_input_is_bit_identical(N, B, ragged):
    """block 1's kernels expand the bit-packed adjacency themselves (the (2, n, n) tensor of
    loaders/data_generator.py:118-125 never exists in HBM).  Scores, loss and every gradient must equal the dense-input
    step bit for bit; in the ragged case the padding bits are garbage (all ones)."""
    import numpy as np
    sd = sub(load_golden('cfg2_reg_n50_b2_4blk.npz'), 'sd/')
    lay = ParamLayout(2, 4, 32, 32, 3)
    params = lay.flatten(sd, DEV)
    rng = np.random.default_rng(5)
    G = 2 * B
    ns = [N] * G if not ragged else [int(v) for v in rng.integers(9, N + 1, size=B)] * 2
    ns[0] = N
    ns[B] = N
    ws = np.ones((G, N, N), dtype=np.float32) if ragged else np.zeros((G, N, N), dtype=np.float32)
    x = np.zeros((G, 2, N, N), dtype=np.float32)
    for g, n in enumerate(ns):
        w = synthetic.erdos_renyi(rng, n, 0.3)
        ws[g, :n, :n] = w
        x[g, :, :n, :n] = synthetic.tensor_representation(w)
    nv = torch.tensor(ns, dtype=torch.int32, device=DEV) if ragged else None
    bits = torch.from_numpy(synthetic.pack_adjacency(ws).view(np.int32)).to(DEV)
    out = []
    for kw in (dict(x=torch.from_numpy(x).to(DEV)), dict(x=None, bits=bits)):
        eng = FgnnEngine(lay, G, N, DEV, ragged=ragged)
        grads = torch.zeros_like(params)
        scores, loss = eng.step(params, grads, kw.pop('x'), nvalid=nv, **kw)
        torch.cuda.synchronize()
        out.append((scores.clone(), loss.clone(), grads))
    assert torch.equal(out[0][0], out[1][0])
    assert torch.equal(out[0][1], out[1][1])
    assert torch.equal(out[0][2], out[1][2])
    assert out[0][2].abs().sum() > 0


def test_engine_cache_is_bounded_and_mixed_eager_captured_adam_stays_in_step(monkeypatch):
    """(a) the per-shape engine cache of the trainer is an LRU with a byte budget; (b) a trainer that mixes
    captured (dense) and eager (ragged) steps keeps ONE Adam step count: it must end where an all-eager trainer ends."""
    from graph_neural_net_amd.trainer import FgnnTrainer
    lay = ParamLayout(2, 1, 32, 32, 3)
    p0 = lay.init_flat(4, DEV)
    dense = [synthetic.make_batch(7300 + i, 4, 16, 'ErdosRenyi', 0.3, 0.05) for i in range(2)]
    rng = np.random.default_rng(3)
    def ragged_batch(seed, sizes):
        xs, ys = [], []
        for i, n in enumerate(sizes):
            a, b = synthetic.make_batch(seed + i, 1, n, 'ErdosRenyi', 0.3, 0.05)
            xs.append(a[0].to(DEV)); ys.append(b[0].to(DEV))
        return xs, ys
    rag = [ragged_batch(7400, [9, 14, 12]), ragged_batch(7500, [11, 16, 10])]
    out = []
    for capture in (True, False):
        tr = FgnnTrainer(lay, p0.clone(), lr=1e-3, capture=capture)
        for s in range(6):
            if s % 2 == 0:
                x1, x2 = dense[(s // 2) % 2]
                tr.train_step(x1.to(DEV), x2.to(DEV))
            else:
                xs, ys = rag[(s // 2) % 2]
                tr.train_step_ragged(xs, ys)
        out.append((tr.params.clone(), tr.opt.t))
    assert out[0][1] == out[1][1] == 6
    assert torch.equal(out[0][0], out[1][0])
    # (a) a tiny budget keeps at most a couple of engines alive however many shapes go by
    monkeypatch.setattr(FgnnTrainer, 'ENGINE_CACHE_BYTES', 1 << 20)
    tr = FgnnTrainer(lay, p0.clone(), lr=1e-3)
    for i, sizes in enumerate(([9, 30], [17, 40], [25, 52], [33, 61], [12, 70])):
        xs, ys = ragged_batch(7600 + 10 * i, sizes)
        tr.train_step_ragged(xs, ys)
    assert len(tr._engines) <= 2, list(tr._engines)


def test_checkpoint_resume_continues_the_optimizer(tmp_path):
    """save_checkpoint(..., optimizer=) + restore_optimizer: a run resumed from the file takes the same steps as the
    uninterrupted run (parameters bit-identical)."""
    from graph_neural_net_amd import checkpoint
    from graph_neural_net_amd.trainer import FgnnTrainer
    lay = ParamLayout(2, 1, 32, 32, 3)
    p0 = lay.init_flat(6, DEV)
    batches = [synthetic.make_batch(7700 + i, 4, 16, 'ErdosRenyi', 0.3, 0.05) for i in range(3)]
    step = lambda tr, s: tr.train_step(batches[s % 3][0].to(DEV), batches[s % 3][1].to(DEV))
    ref = FgnnTrainer(lay, p0.clone(), lr=2e-3)
    for s in range(6):
        step(ref, s)
    a = FgnnTrainer(lay, p0.clone(), lr=2e-3)
    for s in range(3):
        step(a, s)
    f = tmp_path / 'mid.ckpt'
    checkpoint.save_checkpoint(str(f), lay, a.params, epoch=0, global_step=3, optimizer=a.opt)
    obj = torch.load(str(f), weights_only=True)
    layout, flat = checkpoint.load_checkpoint(obj, DEV)
    b = FgnnTrainer(layout, flat, lr=1.0)                  # lr comes back from the file
    assert checkpoint.restore_optimizer(obj, b.opt) and b.opt.t == 3 and b.opt.lr == 2e-3
    for s in range(3, 6):
        step(b, s)
    assert torch.equal(b.params, ref.params)


def _tile_live_np(N, nv):
    """liveness of the ceil(N*N/32) tiles of one graph: does a tile hold a pixel of the valid nv x nv corner?"""
    P = N * N
    p = np.arange(-(-P // 32) * 32)
    ok = (p < P) & (p // N < nv) & (p % N < nv)
    return ok.reshape(-1, 32).any(1)


@pytest.mark.parametrize('N,ns', [(120, [70, 35, 67, 103, 95, 78, 56, 103] * 4), (50, [50, 1, 17, 33, 50, 2]), (7, [3, 7, 0, 5]),
                                  (200, [200, 30])])
def test_ragged_tile_ranges_balance_the_work(N, ns):
    """fgnn_ragged_tile_ranges: 257 monotone bounds covering every tile, pieces of equal cost (a padding-only tile costs
    1/16 of a live one), against a numpy restatement of the tile liveness rule."""
    from graph_neural_net_amd import _lib
    nv = torch.tensor(ns, dtype=torch.int32, device=DEV)
    G = len(ns)
    tpg = -(-N * N // 32)
    ranges = torch.full((257,), -1, dtype=torch.int32, device=DEV)
    _lib.call('fgnn_ragged_tile_ranges', _lib.ptr(nv), G, N, _lib.ptr(ranges), _lib.stream_ptr())
    r = ranges.cpu().numpy().astype(np.int64)
    assert r[0] == 0 and r[-1] == G * tpg and (np.diff(r) >= 0).all()
    cost = np.concatenate([np.where(_tile_live_np(N, n), 16, 1) for n in ns])
    csum = np.concatenate([[0], np.cumsum(cost)])
    piece = csum[r[1:]] - csum[r[:-1]]
    assert piece.sum() == cost.sum()
    assert piece.max() <= cost.sum() / 256 + 16 and piece.min() >= cost.sum() / 256 - 16


def test_ragged_padding_tiles_are_skipped_not_trusted():
    """The ragged engine steps over padding-only tiles (work-balanced ranges, zero-fill).  (i) Forward results are
    bit-identical to the engine that computes every tile, gradients agree to fp32 summation-order noise; (ii) workspaces
    full of NaN / a previous batch with larger graphs do not leak into the result (bit-identical to a fresh engine)."""
    sd = sub(load_golden('cfg2_reg_n50_b2_4blk.npz'), 'sd/')
    lay = ParamLayout(2, 4, 32, 32, 3)
    params = lay.flatten(sd, DEV)
    xs, ys = synthetic.make_ragged_batch(5100, 6, 20, 90)
    x1, nv = O.pad_graph_list(xs)
    x2, _ = O.pad_graph_list(ys)
    x = torch.cat([x1, x2]).contiguous().to(DEV)
    nvd = torch.cat([nv, nv]).to(DEV)
    N, G = x.shape[-1], x.shape[0]

    def run(eng):
        g = torch.zeros_like(params)
        s, l = eng.step(params, g, x, nvalid=nvd)
        torch.cuda.synchronize()
        return s.clone(), l.clone(), g

    skip = FgnnEngine(lay, G, N, DEV, ragged=True)
    assert skip.ranges is not None
    s1, l1, g1 = run(skip)
    FgnnEngine.SKIP_PADDING_TILES = False
    try:
        full = FgnnEngine(lay, G, N, DEV, ragged=True)
    finally:
        FgnnEngine.SKIP_PADDING_TILES = True
    assert full.ranges is None
    s0, l0, g0 = run(full)
    assert torch.equal(s1, s0)
    assert abs(l1.item() - l0.item()) <= 1e-6 * abs(l0.item())
    assert ((g1 - g0).norm() / g0.norm()).item() < 2e-6
    # (ii) poison every workspace tensor, then a batch of full-size graphs, then the ragged batch again
    def poison(eng):
        W = eng._bwd
        ts = (list(eng.z.values()) + list(eng.mult.values()) + list(eng.nrm.values()) + eng.part + [eng.cnt, eng.E, eng.scores, eng.lse]
              + W['dy'] + [W['dmult'], W['dy1'], W['dy2'], W['s12part'], W['dE']] + list(W['s12'].values())
              + list(W['wpart'].values()) + W['coef'])
        for t in ts:
            t.fill_(float('nan'))
    poison(skip)
    s2, l2, g2 = run(skip)
    assert torch.equal(s2, s1) and torch.equal(l2, l1) and torch.equal(g2, g1)
    xf1, xf2 = synthetic.make_batch(5200, G // 2, N, 'ErdosRenyi', 0.3, 0.1)
    gtmp = torch.zeros_like(params)
    skip.step(params, gtmp, torch.cat([xf1, xf2]).contiguous().to(DEV), nvalid=torch.full((G,), N, dtype=torch.int32, device=DEV))
    s3, l3, g3 = run(skip)
    assert torch.equal(s3, s1) and torch.equal(l3, l1) and torch.equal(g3, g1)


@pytest.mark.parametrize('ns', [[3, 7, 1, 5], [33, 2, 17, 31], [64, 20, 64, 9], [65, 40, 1, 65], [130, 12], [1, 1]])
def test_ragged_skipping_engine_equals_full_engine_on_boundary_shapes(ns):
    """Padding-tile skipping against the engine that computes every tile, on shapes where a tile spans several rows
    (N < 32), straddles the 32 / 64 boundaries, or where whole workgroups get empty ranges (few tiles): scores and loss
    bit-identical, gradients equal up to the summation order of the per-workgroup partials."""
    torch.manual_seed(sum(ns))
    sd = O.init_state_dict(num_blocks=2)
    lay = ParamLayout(2, 2, 32, 32, 3)
    params = lay.flatten(sd, DEV)
    rng = np.random.default_rng(sum(ns))
    xs, ys = [], []
    for n in ns:
        a, b = synthetic.make_pair(rng, n, 'ErdosRenyi', 0.4, 0.1)
        xs.append(torch.from_numpy(a)); ys.append(torch.from_numpy(b))
    x1, nv = O.pad_graph_list(xs)
    x2, _ = O.pad_graph_list(ys)
    x = torch.cat([x1, x2]).contiguous().to(DEV)
    nvd = torch.cat([nv, nv]).to(DEV)
    out = []
    for skip in (True, False):
        FgnnEngine.SKIP_PADDING_TILES = skip
        try:
            eng = FgnnEngine(lay, x.shape[0], x.shape[-1], DEV, ragged=True)
        finally:
            FgnnEngine.SKIP_PADDING_TILES = True
        assert (eng.ranges is not None) == skip
        g = torch.zeros_like(params)
        sc, l = eng.step(params, g, x, nvalid=nvd)
        torch.cuda.synchronize()
        out.append((sc.clone(), l.clone(), g))
    (s1, l1, g1), (s0, l0, g0) = out
    assert torch.isfinite(s1).all() and torch.isfinite(g1).all()
    assert torch.equal(s1, s0)
    assert abs(l1.item() - l0.item()) <= 1e-6 * abs(l0.item())
    assert (g1 - g0).norm().item() <= 5e-6 * g0.norm().item() + 1e-7


@pytest.mark.parametrize('t16', ['0', 'pair'])
@pytest.mark.parametrize('B,N,bits', [(3, 33, False), (4, 50, True), (4, 50, False), (3, 33, True), (2, 64, False), (16, 50, False), (16, 50, True)])
def test_pair_backward_equals_the_two_launches_it_replaces(B, N, bits, t16, monkeypatch):
    """fgnn_mlp_bwd_pair (mlp1 + mlp2 of a block in one launch, the input gradient summed inside the wave pair): the block-input
    gradient slabs are bit-identical to those of the two accumulating fgnn_mlp_bwd launches -- (d_in3 + dx1) + dx2 in both --
    and every parameter gradient agrees up to the association of the per-wave partial sums; dense and bit-packed input.
    t16 = 'pair': fgnn_mlp_bwd_pair_t16 (16-pixel tiles; d_in as ONE fma chain over the three contributions) against the same two
    launches: equal to fp32 rounding."""
    monkeypatch.setattr(FgnnEngine, 'T16', t16)
    sd = sub(load_golden('cfg2_reg_n50_b2_4blk.npz'), 'sd/')
    lay = ParamLayout(2, 4, 32, 32, 3)
    params = lay.flatten(sd, DEV)
    x1, x2 = synthetic.make_batch(600 + N, B, N, 'ErdosRenyi', 0.3, 0.1)
    x = torch.cat([x1, x2]).contiguous().to(DEV)
    packed = torch.from_numpy(synthetic.pack_adjacency(torch.cat([x1, x2])[:, 0].numpy()).astype('int32')).to(DEV) if bits else None
    out = []
    for pair in (False, True):
        old = FgnnEngine.PAIR_BWD
        FgnnEngine.PAIR_BWD = pair
        try:
            eng = FgnnEngine(lay, 2 * B, N, DEV, mfma='f32')
            g = torch.zeros_like(params)
            sc, loss = eng.step(params, g, None if bits else x, bits=packed)
            torch.cuda.synchronize()
            out.append((sc.clone(), loss.clone(), g.clone(), eng.unpadded(eng._bwd['dy'][0]), eng.unpadded(eng._bwd['dy'][1])))   # (the row padding of a slab is never written)
            g2 = torch.zeros_like(params)
            eng.step(params, g2, None if bits else x, bits=packed)
            torch.cuda.synchronize()
            assert torch.equal(g, g2)                                    # run-to-run bit-reproducible
        finally:
            FgnnEngine.PAIR_BWD = old
    a, b = out
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    if t16 == '0':
        assert torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])
    else:
        assert ((a[3] - b[3]).norm() / a[3].norm()).item() < 2e-6 and ((a[4] - b[4]).norm() / a[4].norm()).item() < 2e-6
    assert ((a[2] - b[2]).norm() / a[2].norm()).item() < (1e-6 if t16 == '0' else 2e-6)     # (another pixel order in the weight-gradient sums)


@pytest.mark.parametrize('t16', ['0', 'pair'])
def test_pair_backward_on_ragged_batches(t16, monkeypatch):
    """The pair backward with per-graph vertex counts, with and without the padding-tile skipping: block-input gradients
    (valid corners) bit-identical to the two launches (16-pixel-tile kernel: equal to fp32 rounding), parameter gradients up to the
    association of partial sums."""
    monkeypatch.setattr(FgnnEngine, 'T16', t16)
    sd = sub(load_golden('cfg2_reg_n50_b2_4blk.npz'), 'sd/')
    lay = ParamLayout(2, 4, 32, 32, 3)
    params = lay.flatten(sd, DEV)
    xs, ys = synthetic.make_ragged_batch(31, 6, 9, 70)
    x1, nv = O.pad_graph_list(xs)
    x2, _ = O.pad_graph_list(ys)
    x = torch.cat([x1, x2]).contiguous().to(DEV)
    nvd = torch.cat([nv, nv]).to(DEV)
    for skip in (True, False):
        out = []
        for pair in (False, True):
            old = (FgnnEngine.PAIR_BWD, FgnnEngine.SKIP_PADDING_TILES)
            FgnnEngine.PAIR_BWD, FgnnEngine.SKIP_PADDING_TILES = pair, skip
            try:
                eng = FgnnEngine(lay, x.shape[0], x.shape[-1], DEV, ragged=True, mfma='f32')
            finally:
                FgnnEngine.PAIR_BWD, FgnnEngine.SKIP_PADDING_TILES = old
            eng.PAIR_BWD = pair
            g = torch.zeros_like(params)
            for buf in eng._alloc_bwd()['dy']:
                buf.zero_()                      # (skipped padding tiles are never written)
            sc, loss = eng.step(params, g, x, nvalid=nvd)
            torch.cuda.synchronize()
            din = [eng.unpadded(b) for b in eng._bwd['dy']]
            n = nvd.tolist()
            corners = [torch.cat([d[i, :, :n[i], :n[i]].reshape(-1) for i in range(len(n))]) for d in din]
            out.append((sc.clone(), loss.clone(), g.clone(), corners))
        a, b = out
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        if t16 == '0':
            assert all(torch.equal(u, v) for u, v in zip(a[3], b[3])), skip
        else:
            assert all(((u - v).norm() / u.norm()).item() < 2e-6 for u, v in zip(a[3], b[3])), skip
        assert ((a[2] - b[2]).norm() / a[2].norm()).item() < (1e-6 if t16 == '0' else 2e-6)
