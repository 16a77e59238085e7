"""GPU: the two opt-in execution modes of the fp32 engine.

* `mfma='x3'` (csrc/fgnn_x3.h, mlp_fwd_x3.hip, mlp_bwd_x3.hip): mlp1 / mlp2 with every contraction on the bf16 matrix cores
  through the exact three-way operand split.  It is a DIFFERENT fp32-class evaluation of the same function: its values
  differ from the fp32-MFMA kernels' by the reassociation noise of an fp32 sum (~1e-7 per GEMM), so the discrete decisions
  of the model (ReLU masks, pooling arg-max) fall differently on inputs that sit within rounding distance of a tie (a flip
  moves a pair's gradient error from ~1e-5 to ~1e-3).  Its gradients are gated by the SAME distribution gate as the fp32-MFMA
  engine's (tests/test_gpu_grad_gate.py, 148 reference-generated single-pair cases; profiles/archive/r04_gradgate_table.txt shows that
  the old single-seed gates fail 14 / 12 / 8 of 142 cases for f32 / x3 / the reference's own 1-thread run).  Here: forward
  tensors against fp64, determinism, recompute consistency, and the pair backward against the launches it replaces.
* `FgnnEngineDual`: two half-batch chains on two streams.  Same kernels, same per-pair arithmetic: scores bit-identical to the
  single engine, gradients equal up to the association of the per-workgroup partial sums.
"""
import pytest
import torch

from graph_neural_net_amd import synthetic
from graph_neural_net_amd.engine import FgnnEngine, ParamLayout
from graph_neural_net_amd.engine_dual import FgnnEngineDual
from oracle import fgnn_oracle as O
from util import is_zero_grad, load_golden, rel, sub, unpack_pairs

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _step(eng_cls, sd, x1, x2, nblk, **kw):
    lay = ParamLayout(2, nblk, 32, 32, 3)
    params = lay.flatten(sd, DEV)
    grads = torch.zeros_like(params)
    eng = eng_cls(lay, 2 * x1.shape[0], x1.shape[-1], DEV, **kw)
    scores, loss = eng.step(params, grads, torch.cat([x1, x2]).contiguous().to(DEV))
    torch.cuda.synchronize()
    return eng, params, scores.cpu().clone(), loss.item(), lay.unflatten(grads.cpu().clone())


def test_x3_forward_tensors_against_fp64():
    """Block outputs and scores of the x3 engine against the fp64 oracle: as close as the fp32-MFMA engine and the fp32
    oracle are (one block: 1e-6 level; four blocks: the 3e-5 gate of test_gpu_parity.py)."""
    d = load_golden('cfg2_reg_n50_b2_4blk.npz')
    sd = sub(d, 'sd/')
    x1, x2 = d['x1'], d['x2']
    s64, _, _ = O.step_fwd_bwd(x1.double(), x2.double(), {k: v.double() for k, v in sd.items()})
    keep = {}
    O.node_embedding(x1.double(), {k: v.double() for k, v in sd.items()}, keep)
    res = {}
    for mode in ('f32', 'x3'):
        eng, params, scores, loss, grads = _step(FgnnEngine, sd, x1, x2, 4, mfma=mode)
        assert eng.x3 == (mode == 'x3')
        G = x1.shape[0]
        y1 = eng.normalized(1, 1, params)[:G].cpu()
        res[mode] = (rel(y1, keep['ne/bm/block1/mlp1']), rel(scores, s64))
    assert res['x3'][0] < 2e-6 and res['x3'][0] < 4 * res['f32'][0] + 1e-7, res
    assert res['x3'][1] < 3e-5, res


def test_x3_is_deterministic_and_its_backward_recomputes_its_forward():
    """Bit-reproducible run to run; and the backward's recompute of the hidden activations uses the forward's arithmetic (same
    split, same eight partial products, same order): two steps on the same workspace give identical gradients, and the
    gradient is the same whether the forward ran once or twice before it."""
    sd = sub(load_golden('cfg2_reg_n50_b2_4blk.npz'), 'sd/')
    x1, x2 = synthetic.make_batch(77, 3, 33, 'ErdosRenyi', 0.4, 0.1)
    a = _step(FgnnEngine, sd, x1, x2, 4, mfma='x3')
    b = _step(FgnnEngine, sd, x1, x2, 4, mfma='x3')
    assert torch.equal(a[2], b[2]) and a[3] == b[3] and all(torch.equal(a[4][k], b[4][k]) for k in a[4])


@pytest.mark.parametrize('mode', ['f32', 'x3'])
def test_dual_chains_equal_the_single_engine(mode):
    """Two half-batch chains on two streams: scores bit-identical, loss and gradients equal up to the association of the
    partial sums, bit-reproducible run to run, eager == captured; odd pair counts split 3 + 2."""
    sd = sub(load_golden('cfg2_reg_n50_b2_4blk.npz'), 'sd/')
    for B, N in ((5, 24), (8, 50)):
        x1, x2 = synthetic.make_batch(40 + B, B, N, 'ErdosRenyi', 0.3, 0.1)
        _, _, s1, l1, g1 = _step(FgnnEngine, sd, x1, x2, 4, mfma=mode)
        eng, params, s2, l2, g2 = _step(FgnnEngineDual, sd, x1, x2, 4, mfma=mode)
        assert torch.equal(s1, s2)
        assert abs(l1 - l2) <= 1e-6 * abs(l1)
        for k in g1:
            if not is_zero_grad(k):
                assert rel(g2[k], g1[k]) < 1e-5, k
        _, _, s3, l3, g3 = _step(FgnnEngineDual, sd, x1, x2, 4, mfma=mode)
        assert torch.equal(s2, s3) and l2 == l3 and all(torch.equal(g2[k], g3[k]) for k in g2)
    # captured (fork / join become graph edges) == eager
    lay = eng.layout
    grads = torch.zeros_like(params)
    eng.stage_inputs(torch.cat([x1, x2]).contiguous().to(DEV))
    eng.step(params, grads, None)
    torch.cuda.synchronize()
    eager = grads.clone()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        eng.step(params, grads, None)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        eng.step(params, grads, None)
    grads.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(grads, eager)


def test_dual_chains_on_a_ragged_batch_equal_the_single_engine():
    """with tile ranges the MLP kernels ignore cu_share and run the full grid, so a ragged chain writes 256 rows
    of weight-gradient partials -- the dual engine must size and reduce them as such.  Scores bit-identical to the single
    ragged engine, gradients equal up to the association of the partial sums, reproducible run to run."""
    sd = sub(load_golden('cfg2_reg_n50_b2_4blk.npz'), 'sd/')
    xs, ys = synthetic.make_ragged_batch(5100, 6, 20, 70)
    x1, nv = O.pad_graph_list(xs)
    x2, _ = O.pad_graph_list(ys)
    lay = ParamLayout(2, 4, 32, 32, 3)
    params = lay.flatten(sd, DEV)
    x = torch.cat([x1, x2]).contiguous().to(DEV)
    nvd = torch.cat([nv, nv]).to(DEV)
    out = []
    for cls in (FgnnEngine, FgnnEngineDual, FgnnEngineDual):
        eng = cls(lay, x.shape[0], x.shape[-1], DEV, ragged=True)
        grads = torch.zeros_like(params)
        scores, loss = eng.step(params, grads, x, nvalid=nvd)
        torch.cuda.synchronize()
        out.append((scores.cpu().clone(), loss.item(), lay.unflatten(grads.cpu().clone())))
    (s1, l1, g1), (s2, l2, g2), (s3, l3, g3) = out
    assert torch.equal(s1, s2) and torch.equal(s2, s3)
    assert abs(l1 - l2) <= 1e-6 * abs(l1) and l2 == l3
    for k in g1:
        assert torch.isfinite(g2[k]).all(), k
        assert torch.equal(g2[k], g3[k]), k
        if not is_zero_grad(k):
            assert rel(g2[k], g1[k]) < 1e-5, (k, rel(g2[k], g1[k]))


@pytest.mark.parametrize('B,N,bits', [(3, 33, False), (4, 50, True), (4, 50, False), (2, 64, False), (16, 50, False)])
def test_x3_pair_backward_equals_the_two_x3_launches_it_replaces(B, N, bits):
    """fgnn_mlp_bwd_pair_x3 (mlp1 + mlp2 of a block in one launch on the bf16 matrix cores, weight-gradient operands transposed
    on the matrix pipe): the block-input gradient slabs are bit-identical to those of the two accumulating fgnn_mlp_bwd_x3
    launches -- same split, same partial products, (d_in3 + dx1) + dx2 in both -- and every parameter gradient agrees up to the
    association of the per-wave partial sums (the transposed operands are the same bf16 parts either way: exact)."""
    import numpy as np
    sd = sub(load_golden('cfg2_reg_n50_b2_4blk.npz'), 'sd/')
    lay = ParamLayout(2, 4, 32, 32, 3)
    params = lay.flatten(sd, DEV)
    x1, x2 = synthetic.make_batch(600 + N, B, N, 'ErdosRenyi', 0.3, 0.1)
    x = torch.cat([x1, x2]).contiguous().to(DEV)
    packed = torch.from_numpy(synthetic.pack_adjacency(torch.cat([x1, x2])[:, 0].numpy()).astype('int32')).to(DEV) if bits else None
    out = []
    for pair in (False, True):
        eng = FgnnEngine(lay, 2 * B, N, DEV, mfma='x3')
        eng.PAIR_BWD = pair
        g = torch.zeros_like(params)
        sc, loss = eng.step(params, g, None if bits else x, bits=packed)
        torch.cuda.synchronize()
        out.append((sc.clone(), loss.clone(), g.clone(), eng.unpadded(eng._bwd['dy'][0]), eng.unpadded(eng._bwd['dy'][1])))
        g2 = torch.zeros_like(params)
        eng.step(params, g2, None if bits else x, bits=packed)
        torch.cuda.synchronize()
        assert torch.equal(g, g2)                                    # run-to-run bit-reproducible
    a, b = out
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])
    assert ((a[2] - b[2]).norm() / a[2].norm()).item() < 1e-6, ((a[2] - b[2]).norm() / a[2].norm()).item()
