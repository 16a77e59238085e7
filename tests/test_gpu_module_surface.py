"""GPU: the structured block 1 behind the reference's module surface (models/trainers.py:60-76, loaders/loaders.py:5-15,
loaders/data_generator.py:118-125).  `Siamese_Node_Exp(..., input_form='tensor_representation')` takes the dense loader batch,
bit-packs + verifies it on the device and runs the same launch sequence as `FgnnEngine.step(bits=...)` /
`FgnnTrainer.train_step_bits`: bit for bit; a batch that is not a tensor representation is refused."""
import numpy as np
import pytest
import torch

from graph_neural_net_amd import synthetic
from graph_neural_net_amd.engine import FgnnEngine, ParamLayout
from graph_neural_net_amd.masked import from_list
from graph_neural_net_amd.siamese import Siamese_Node_Exp
from graph_neural_net_amd.trainer import FgnnTrainer
from util import is_zero_grad, rel

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _ne(blocks, ragged=False):
    ne = dict(type='node_embedding', block_init='block_emb', block_inside='block', num_blocks=blocks, in_features=32,
              out_features=32, depth_of_mlp=3)
    if ragged:
        ne['constant_n_vertices'] = False
    return ne


def _pk(x):
    return torch.from_numpy(synthetic.pack_adjacency(x[:, 0].cpu().numpy()).view(np.int32)).to(DEV)


def _perturb_biases(model, seed):
    """(with the reference's zero conv biases many pre-activations are exactly 0 and a ReLU mask is anybody's choice)"""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith('.bias'):
                p.add_((torch.rand(p.shape, generator=g) * 0.2 - 0.1).to(p.device).view(p.shape))


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
@pytest.mark.parametrize('B,N,blocks', [(2, 50, 4), (4, 24, 2), (3, 70, 1)])
def test_fused_step_on_tensor_representations_is_the_bit_packed_engine_step(precision, B, N, blocks):
    torch.manual_seed(11)
    model = Siamese_Node_Exp(2, _ne(blocks), precision=precision, input_form='tensor_representation').to(DEV)
    _perturb_biases(model, 1)
    x1, x2 = [t.to(DEV) for t in synthetic.make_batch(9100 + N, B, N, 'ErdosRenyi', 0.3, 0.1)]
    lay = ParamLayout(2, blocks, 32, 32, 3)
    sd = {k[len('node_embedder.'):]: v for k, v in model.state_dict().items()}
    params = lay.flatten(sd, DEV)
    # the engine line of bench.py: bit-packed input, block 1 structured
    if precision == 'bf16':
        from graph_neural_net_amd.engine16 import FgnnEngineBF16
        eng = FgnnEngineBF16(lay, 2 * B, N, DEV, block1='structured')
    else:
        eng = FgnnEngine(lay, 2 * B, N, DEV, block1='structured')
    assert eng.struct1
    grads = torch.zeros_like(params)
    scores_e, loss_e = eng.step(params, grads, None, bits=torch.cat([_pk(x1), _pk(x2)]).contiguous())
    torch.cuda.synchronize()
    scores_e, loss_e, grads = scores_e.clone(), loss_e.clone(), grads.clone()
    for cap in (False, True, True):
        for p in model.parameters():
            p.grad = None
        loss, scores = model.fused_step({'input': x1}, {'input': x2}, capture=cap)
        assert torch.equal(scores, scores_e) and torch.equal(loss.reshape(1), loss_e.reshape(1)), cap
        flat = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
        assert torch.equal(flat, grads), cap
    assert model.check_input_form()
    # ... and against the dense form of the same module (generic block 1): the same function to rounding
    dense = Siamese_Node_Exp(2, _ne(blocks), precision=precision).to(DEV)
    dense.load_state_dict(model.state_dict())
    ld, sdn = dense.fused_step(x1, x2)
    tol = 5e-2 if precision == 'bf16' else 2e-5
    assert rel(scores, sdn) < tol and abs(ld.item() - loss.item()) < tol * abs(ld.item())


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_trainer_from_module_packs_dense_batches_and_equals_train_step_bits(precision):
    """FgnnTrainer.from_module on an input_form='tensor_representation' module: train_step on the loader's dense batch == the same
    trainer fed bit-packed words (train_step_bits), parameters and losses bit for bit over changing batches, captured and eager."""
    lay = ParamLayout(2, 2, 32, 32, 3)
    batches = [synthetic.make_batch(8300 + s, 4, 24, 'ErdosRenyi', 0.3, 0.05) for s in range(3)]
    for capture in (True, False):
        torch.manual_seed(5)
        model = Siamese_Node_Exp(2, _ne(2), lr=2e-3, precision=precision, input_form='tensor_representation').to(DEV)
        p0 = lay.flatten({k[len('node_embedder.'):]: v for k, v in model.state_dict().items()}, DEV)
        tr = FgnnTrainer.from_module(model, capture=capture)
        assert tr.input_form == 'tensor_representation' and tr.block1 == 'structured'
        ref = FgnnTrainer(lay, p0.clone(), lr=2e-3, capture=capture, precision=precision, block1='structured')
        for x1, x2 in batches:
            l1, s1 = tr.train_step(x1.to(DEV), x2.to(DEV))
            l2, s2 = ref.train_step_bits(_pk(x1), _pk(x2))
            assert l1.item() == l2.item() and torch.equal(s1, s2)
        torch.cuda.synchronize()
        assert torch.equal(tr.params, ref.params) and tr.opt.t == ref.opt.t == 3
        # the module sees the trained weights (the trainer works in place on its flat buffer)
        assert torch.equal(model.node_embedder._flat, ref.params)


def test_a_batch_that_is_not_a_tensor_representation_is_refused():
    torch.manual_seed(2)
    model = Siamese_Node_Exp(2, _ne(1), input_form='tensor_representation').to(DEV)
    x1, x2 = [t.to(DEV) for t in synthetic.make_batch(9300, 3, 20, 'ErdosRenyi', 0.3, 0.1)]
    bad_entry = x1.clone()
    bad_entry[1, 0, 3, 7] = 0.5                                  # an adjacency entry outside {0, 1}
    bad_diag = x2.clone()
    bad_diag[2, 1, 4, 4] += 1.0                                  # channel 1 is not diag(row sums)
    off_diag = x2.clone()
    off_diag[0, 1, 2, 5] = 1.0                                   # channel 1 off the diagonal
    for a, b in ((bad_entry, x2), (x1, bad_diag), (x1, off_diag)):
        fresh = Siamese_Node_Exp(2, _ne(1), input_form='tensor_representation').to(DEV)
        with pytest.raises(RuntimeError, match='tensor representation'):          # first step of a shape: checked at once
            fresh.fused_step(a, b)
        fresh.fused_step(x1, x2)                                                  # the flag was reset: a good batch runs
    # later steps: the verdict stays on the device (no synchronisation per step) until somebody asks
    model.INPUT_CHECK_EVERY = 0
    model.fused_step(x1, x2)
    model.fused_step(bad_entry, x2)
    model.fused_step(x1, x2)
    with pytest.raises(RuntimeError, match='tensor representation'):
        model.check_input_form()
    assert model.check_input_form()                                               # ... and is cleared by the report
    model.INPUT_CHECK_EVERY = 2
    with pytest.raises(RuntimeError, match='tensor representation'):
        model.fused_step(x1, bad_diag)                                            # the 4th call of this shape: read back
    model.INPUT_CHECK_EVERY = 128
    # the dense form of the same module takes any tensor
    dense = Siamese_Node_Exp(2, _ne(1)).to(DEV)
    dense.fused_step(bad_entry, bad_diag)
    # the trainer built from the module refuses too
    tr = FgnnTrainer.from_module(Siamese_Node_Exp(2, _ne(1), input_form='tensor_representation').to(DEV), capture=True)
    with pytest.raises(RuntimeError, match='tensor representation'):
        tr.train_step(bad_entry, x2)
    tr.train_step(x1, x2)


def test_periodic_check_raises_without_being_asked():
    torch.manual_seed(2)
    model = Siamese_Node_Exp(2, _ne(1), input_form='tensor_representation').to(DEV)
    model.INPUT_CHECK_EVERY = 3
    x1, x2 = [t.to(DEV) for t in synthetic.make_batch(9301, 2, 12, 'ErdosRenyi', 0.3, 0.1)]
    bad = x1.clone()
    bad[0, 0, 1, 2] = 2.0
    model.fused_step(x1, x2)            # 1: first of the shape, checked
    model.fused_step(bad, x2)           # 2: not read back
    with pytest.raises(RuntimeError, match='tensor representation'):
        model.fused_step(x1, x2)        # 3: the sticky verdict of step 2 surfaces


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_fused_step_on_masked_tensor_representations(precision):
    """MaskedTensor batches (loaders/loaders.py:5-10) with the opt-in: the padded batch is packed to the engine's padded size,
    garbage in the padding is not looked at, results equal the dense form of the same module to rounding and the bit-packed
    engine step bit for bit; a bad entry INSIDE a valid corner is refused; vertex counts that differ between the sides are refused."""
    torch.manual_seed(7)
    model = Siamese_Node_Exp(2, _ne(2, ragged=True), precision=precision, input_form='tensor_representation', metric='max').to(DEV)
    _perturb_biases(model, 3)
    dense = Siamese_Node_Exp(2, _ne(2, ragged=True), precision=precision, metric='max').to(DEV)
    dense.load_state_dict(model.state_dict())
    lay = ParamLayout(2, 2, 32, 32, 3)
    graphs = set()
    for seed, sizes in ((1, (16, 9, 12)), (2, (13, 7, 15)), (3, (16, 16, 10))):            # all pad to N = 16
        rng = np.random.default_rng(seed)
        xs, ys = [], []
        for n in sizes:
            a, b = synthetic.make_pair(rng, n, 'ErdosRenyi', 0.4, 0.1)
            xs.append(torch.from_numpy(a).to(DEV))
            ys.append(torch.from_numpy(b).to(DEV))
        m1, m2 = from_list(xs, dims=(1, 2), base_name='N'), from_list(ys, dims=(1, 2), base_name='M')
        # garbage in the padding of the loader's tensors must not matter
        nmax = max(sizes)
        for m in (m1, m2):
            for i, n in enumerate(sizes):
                m.tensor.rename(None)[i, :, n:, :] = 7.5
                m.tensor.rename(None)[i, :, :, n:] = -3.0
        l1, s1, (acc1, tot1) = model.fused_step(m1, m2, metric=True)
        g1 = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone()
        # the engine on the same words
        B, N = len(sizes), 16
        pad = lambda lst: torch.stack([torch.nn.functional.pad(t, (0, N - t.shape[-1], 0, N - t.shape[-1])) for t in lst])
        bits = torch.cat([_pk(pad(xs)), _pk(pad(ys))]).contiguous()
        nv = torch.tensor(list(sizes) * 2, dtype=torch.int32, device=DEV)
        params = lay.flatten({k[len('node_embedder.'):]: v for k, v in model.state_dict().items()}, DEV)
        if precision == 'bf16':
            from graph_neural_net_amd.engine16 import FgnnEngineBF16
            eng = FgnnEngineBF16(lay, 2 * B, N, DEV, ragged=True, block1='structured')
        else:
            eng = FgnnEngine(lay, 2 * B, N, DEV, ragged=True, block1='structured')
        ge = torch.zeros_like(params)
        se, le = eng.step(params, ge, None, nvalid=nv, bits=bits)
        torch.cuda.synchronize()
        assert torch.equal(s1.tensor.rename(None), se)
        assert rel(g1, ge) < 1e-6                  # (fused_step scales by a device reciprocal of sum(n), the engine by a host division)
        assert int(tot1.item()) == sum(sizes)
        # the dense form of the module on clean tensors
        c1, c2 = from_list(xs, dims=(1, 2), base_name='N'), from_list(ys, dims=(1, 2), base_name='M')
        l2, s2 = dense.fused_step(c1, c2)
        tol = 5e-2 if precision == 'bf16' else 2e-5
        for i, n in enumerate(sizes):
            assert rel(s1.tensor.rename(None)[i, :n, :n], s2.tensor.rename(None)[i, :n, :n]) < tol
        assert abs(l1.item() - l2.item()) < tol * abs(l2.item())
        if precision == 'fp32':
            for (name, p), q in zip(model.named_parameters(), dense.parameters()):
                if not is_zero_grad(name):
                    assert rel(p.grad, q.grad) < 2e-4, (name, rel(p.grad, q.grad))
        eng_s = next(e for e in model.node_embedder._engines.values() if getattr(e, '_step_state', None) is not None)
        graphs.add(id(eng_s._step_state['graph'][True]))
    assert len(graphs) == 1                         # one capture served the three batches
    # a bad entry inside a valid corner
    m1.tensor.rename(None)[0, 0, 1, 2] = 0.25
    model.fused_step(m1, m2)
    with pytest.raises(RuntimeError, match='tensor representation'):
        model.check_input_form()
    # the sides must share the vertex counts
    other = from_list([y[:, :n - 1, :n - 1].contiguous() if i == 1 else y for i, (y, n) in enumerate(zip(ys, sizes))], dims=(1, 2), base_name='M')
    fresh = Siamese_Node_Exp(2, _ne(2, ragged=True), precision=precision).to(DEV)
    c1 = from_list(xs, dims=(1, 2), base_name='N')
    with pytest.raises(RuntimeError, match='vertex counts'):
        fresh.fused_step(c1, other)


def test_bf16_fused_step_on_masked_tensor_batches_equals_the_eager_module_path():
    """(review of round 4: a bf16 model given a MaskedTensor batch used to die with a TypeError in fused_step)"""
    torch.manual_seed(9)
    model = Siamese_Node_Exp(2, _ne(2, ragged=True), precision='bf16').to(DEV)
    rng = np.random.default_rng(4)
    sizes = (16, 11, 14)                      # nmax = 16 = the padded size of fused_step: same geometry as the eager path
    xs, ys = [], []
    for n in sizes:
        a, b = synthetic.make_pair(rng, n, 'ErdosRenyi', 0.4, 0.1)
        xs.append(torch.from_numpy(a).to(DEV))
        ys.append(torch.from_numpy(b).to(DEV))
    m1, m2 = from_list(xs, dims=(1, 2), base_name='N'), from_list(ys, dims=(1, 2), base_name='M')
    scores = model(m1, m2)
    loss = model.loss(scores)
    loss.backward()
    eager = {n: p.grad.clone() for n, p in model.named_parameters()}
    for cap in (False, True):
        for p in model.parameters():
            p.grad = None
        l2, s2 = model.fused_step(m1, m2, capture=cap)
        assert abs(l2.item() - loss.item()) <= 1e-5 * abs(loss.item())
        for i, n in enumerate(sizes):
            assert rel(s2.tensor.rename(None)[i, :n, :n], scores.tensor.rename(None)[i, :n, :n].detach()) < 1e-5
        for name, p in model.named_parameters():
            if not is_zero_grad(name):
                assert rel(p.grad, eager[name]) < 1e-4, (name, rel(p.grad, eager[name]))


def test_tensor_representation_opt_in_on_a_zero_padded_model_and_on_shapes_it_is_not_built_for():
    """A model narrower than the engine (16 / 24 features: run zero-padded, Network._padded_layout) takes the structured block 1 through
    the opt-in as well -- same function as its dense form to rounding; one input channel or N > 256 fall back to the dense path of the
    same module without complaint (nothing to verify there: no structure is assumed)."""
    torch.manual_seed(13)
    ne = dict(type='node_embedding', block_init='block_emb', block_inside='block', num_blocks=2, in_features=16, out_features=24, depth_of_mlp=3)
    model = Siamese_Node_Exp(2, dict(ne), input_form='tensor_representation').to(DEV)
    _perturb_biases(model, 5)
    dense = Siamese_Node_Exp(2, dict(ne)).to(DEV)
    dense.load_state_dict(model.state_dict())
    x1, x2 = [t.to(DEV) for t in synthetic.make_batch(9400, 3, 22, 'ErdosRenyi', 0.3, 0.1)]
    l1, s1 = model.fused_step(x1, x2)
    l2, s2 = dense.fused_step(x1, x2)
    eng = next(e for e in model.node_embedder._engines.values() if getattr(e, '_step_state', None) is not None)
    assert eng.struct1 and 'bits' in eng._step_state
    assert rel(s1, s2) < 2e-5 and abs(l1.item() - l2.item()) < 2e-5 * abs(l2.item())
    for (name, p), q in zip(model.named_parameters(), dense.parameters()):
        if not is_zero_grad(name):
            assert rel(p.grad, q.grad) < 2e-4, (name, rel(p.grad, q.grad))
    # one input channel: the dense path of the same module
    one = Siamese_Node_Exp(1, dict(ne, in_features=32, out_features=32), input_form='tensor_representation').to(DEV)
    a = torch.rand(2, 1, 12, 12, device=DEV)
    one.fused_step(a, a.clone())
    assert not one.check_input_form()                      # no flag was ever created: nothing was packed


def test_trainer_checkpoint_reads_the_pending_input_verdicts(tmp_path):
    """FgnnTrainer.save_checkpoint: a bad batch since the last check (the verdict is read on every INPUT_CHECK_EVERY-th step only) is
    reported BEFORE the parameters it touched are written; after the report the trainer saves."""
    from graph_neural_net_amd.engine import ParamLayout
    lay = ParamLayout(2, 1, 32, 32, 3)
    p0 = lay.init_flat(0, torch.device(DEV))
    tr = FgnnTrainer(lay, p0.clone(), lr=1e-3, input_form='tensor_representation')
    tr.INPUT_CHECK_EVERY = 0
    x1, x2 = [t.to(DEV) for t in synthetic.make_batch(9400, 3, 20, 'ErdosRenyi', 0.3, 0.1)]
    tr.train_step(x1, x2)
    bad = x1.clone()
    bad[0, 0, 2, 9] = 0.25
    tr.train_step(bad, x2)
    path = str(tmp_path / 'ck.pt')
    with pytest.raises(RuntimeError, match='tensor representation'):
        tr.save_checkpoint(path)
    import os
    assert not os.path.exists(path)
    tr.save_checkpoint(path)                                    # the report cleared the flag
    assert os.path.exists(path)
