#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ FROM THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference, read-only).  The
reference's Python files never travel: this script imports them in-process
(with two in-memory shims), feeds them seeded synthetic
inputs from ``graph_neural_net_amd.synthetic`` and stores inputs + expected
outputs as small ``.npz`` files.  It also asserts that ``oracle/fgnn_oracle.py``
is ``torch.equal`` to the reference on every forward tensor, the loss and all
gradients -- that is what pins the oracle.

Usage:  python tests/golden/make_golden.py     (from the repo root)
"""
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = '/root/reference'
OUT = os.path.join(ROOT, 'tests', 'golden')
sys.dont_write_bytecode = True


def import_reference():
    """Put the reference on sys.path behind two shims (nothing is written under REF)."""
    stubs = tempfile.mkdtemp(prefix='refstubs_')
    os.makedirs(os.path.join(stubs, 'pytorch_lightning'))
    with open(os.path.join(stubs, 'pytorch_lightning', '__init__.py'), 'w') as f:
        f.write('import torch.nn as nn\n'
                'class LightningModule(nn.Module):\n'
                '    def log(self, *a, **k): pass\n'
                'def seed_everything(*a, **k): pass\n')
    m = types.ModuleType('numpy.lib.arraysetops')
    m.isin = np.isin
    sys.modules['numpy.lib.arraysetops'] = m
    sys.path[:0] = [stubs, REF]


NODE_EMB = dict(type='node_embedding', block_init='block_emb', block_inside='block',
                in_features=32, out_features=32, depth_of_mlp=3)


def build_reference_model(num_blocks, seed, constant_n_vertices=True):
    from models.trainers import Siamese_Node_Exp
    torch.manual_seed(seed)
    ne = dict(NODE_EMB, num_blocks=num_blocks)
    if not constant_n_vertices:
        ne['constant_n_vertices'] = False
    return Siamese_Node_Exp(2, ne)


def perturb_(model, seed):
    """Make biases / gn affine non-trivial so the fixtures exercise them."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith('.bias') and p.dim() == 1:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
            elif name.endswith('gn.weight'):
                p.mul_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            elif name.endswith('gn.bias'):
                p.add_(0.05 * torch.randn(p.shape, generator=g))


def ref_step(model, x1, x2):
    model.zero_grad()
    scores = model({'input': x1}, {'input': x2})
    loss = model.loss(scores)
    loss.backward()
    grads = {n[len('node_embedder.'):]: p.grad.detach().clone() for n, p in model.named_parameters()}
    return scores.detach(), loss.detach(), grads


def check_oracle_bit_equal(model, x1, x2, tag):
    sys.path.insert(0, ROOT)
    from oracle import fgnn_oracle as O
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    s_ref, l_ref, g_ref = ref_step(model, x1, x2)
    s_or, l_or, g_or = O.step_fwd_bwd(x1, x2, sd)
    assert torch.equal(s_ref, s_or), tag + ': scores differ'
    assert torch.equal(l_ref, l_or), tag + ': loss differs'
    worst = 0.0
    for k in g_ref:
        if not torch.equal(g_ref[k], g_or[k]):
            worst = max(worst, O.max_rel_err(g_or[k], g_ref[k]))
    # intermediates
    inter_ref = model.node_embedder({'input': x1})
    keep = {}
    O.node_embedding(x1, sd, keep)
    for k, v in keep.items():
        assert torch.equal(inter_ref[k], v), tag + ': intermediate %s differs' % k
    return worst


def f64(model):
    import copy
    return copy.deepcopy(model).double()


def main():
    import_reference()
    sys.path.insert(0, ROOT)
    from graph_neural_net_amd import synthetic
    from oracle import fgnn_oracle as O
    torch.set_num_threads(8)
    meta = {'torch': torch.__version__, 'reference': 'mlelarge/graph_neural_net @ v1', 'cases': {}}

    # ---------------- cfg1: N=20 ER p=.2, B=4, 1 block ----------------
    model = build_reference_model(1, seed=0)
    perturb_(model, 100)
    x1, x2 = synthetic.make_batch(1000, 4, 20, 'ErdosRenyi', 0.2, 0.1)
    worst = check_oracle_bit_equal(model, x1, x2, 'cfg1')
    s, l, g = ref_step(model, x1, x2)
    m64 = f64(model)
    s64, l64, g64 = ref_step(m64, x1.double(), x2.double())
    inter = model.node_embedder({'input': x1})
    d = {'x1': x1.numpy(), 'x2': x2.numpy(), 'scores': s.numpy(), 'loss': l.numpy(),
         'scores64': s64.numpy(), 'loss64': l64.numpy()}
    for k, v in model.state_dict().items():
        d['sd/' + k[len('node_embedder.'):]] = v.numpy()
    for k, v in g.items():
        d['grad/' + k] = v.numpy()
    for k, v in g64.items():
        d['grad64/' + k] = v.numpy()
    for k in ('ne/bm/block1/mlp1', 'ne/bm/block1/mlp2', 'ne/bm/block1/mult', 'ne/bm/block1/mlp3', 'ne/suffix'):
        d['inter/' + k] = inter[k].detach().numpy()
    np.savez_compressed(os.path.join(OUT, 'cfg1_er_n20_b4_1blk.npz'), **d)
    meta['cases']['cfg1_er_n20_b4_1blk'] = {'oracle_bit_equal_forward': True, 'oracle_worst_grad_relerr': worst}

    # ---------------- cfg2 (small batch): N=50 Regular d=10, B=2, 4 blocks ----------------
    model = build_reference_model(4, seed=0)
    perturb_(model, 200)
    x1, x2 = synthetic.make_batch(2000, 2, 50, 'Regular', 0.2, 0.1)
    worst = check_oracle_bit_equal(model, x1, x2, 'cfg2')
    s, l, g = ref_step(model, x1, x2)
    m64 = f64(model)
    s64, l64, g64 = ref_step(m64, x1.double(), x2.double())
    inter = model.node_embedder({'input': x1})
    d = {'x1': x1.numpy(), 'x2': x2.numpy(), 'scores': s.numpy(), 'loss': l.numpy(),
         'scores64': s64.numpy(), 'loss64': l64.numpy()}
    for k, v in model.state_dict().items():
        d['sd/' + k[len('node_embedder.'):]] = v.numpy()
    for k, v in g.items():
        d['grad/' + k] = v.numpy()
    for k, v in g64.items():
        d['grad64/' + k] = v.numpy()
    for k in ('ne/bm/block1/mlp3', 'ne/bm/block4/mlp3', 'ne/suffix'):
        d['inter/' + k] = inter[k].detach().numpy()[:1]  # first graph only (size)
    np.savez_compressed(os.path.join(OUT, 'cfg2_reg_n50_b2_4blk.npz'), **d)
    meta['cases']['cfg2_reg_n50_b2_4blk'] = {'oracle_bit_equal_forward': True, 'oracle_worst_grad_relerr': worst}

    # ---------------- ragged: n in {5,7,6,12}, 2 blocks ----------------
    from maskedtensors import maskedtensor
    model_d = build_reference_model(2, seed=1)
    perturb_(model_d, 300)
    model_m = build_reference_model(2, seed=1, constant_n_vertices=False)
    model_m.load_state_dict(model_d.state_dict())
    rng = np.random.default_rng(3000)
    xs, ys = [], []
    for n in (5, 7, 6, 12):
        a, b = synthetic.make_pair(rng, n, 'ErdosRenyi', 0.4, 0.1)
        xs.append(torch.from_numpy(a))
        ys.append(torch.from_numpy(b))
    # (i) reference dense model per graph (the reference tests' definition of masked correctness)
    model_d.zero_grad()
    e1 = [model_d.node_embedder({'input': x.unsqueeze(0)})['ne/suffix'].squeeze(0) for x in xs]
    e2 = [model_d.node_embedder({'input': y.unsqueeze(0)})['ne/suffix'].squeeze(0) for y in ys]
    scores = [a.t() @ b for a, b in zip(e1, e2)]
    loss = 0
    tot = 0
    for sc in scores:   # triplet_loss('mean') on a list (toolbox/losses.py:27-34; get_device() needs a tensor)
        loss = loss + torch.nn.functional.cross_entropy(sc, torch.arange(sc.shape[0]), reduction='sum')
        tot += sc.shape[0]
    loss = loss / tot
    loss.backward()
    g = {n[len('node_embedder.'):]: p.grad.detach().clone() for n, p in model_d.named_parameters()}
    # (ii) reference MaskedTensor path for the node-embedder branch
    mt = maskedtensor.from_list(xs, dims=(1, 2), base_name='N')
    em = model_m.node_embedder({'input': mt})['ne/suffix']
    em_list = list(em)
    for a, b in zip(em_list, e1):
        assert torch.allclose(a, b, atol=1e-5), 'reference masked path disagrees with per-graph dense'
    # oracle agrees bit-for-bit with (i)
    sd = {k: v.detach() for k, v in model_d.state_dict().items()}
    s_or, l_or, g_or = O.step_fwd_bwd_ragged(xs, ys, sd)
    for a, b in zip(s_or, scores):
        assert torch.equal(a, b.detach()), 'ragged: oracle scores differ'
    assert torch.equal(l_or, loss.detach())
    d = {'loss': loss.detach().numpy(), 'ns': np.array([x.shape[-1] for x in xs])}
    for i, (x, y, sc, a, b) in enumerate(zip(xs, ys, scores, e1, e2)):
        d['x1/%d' % i] = x.numpy()
        d['x2/%d' % i] = y.numpy()
        d['scores/%d' % i] = sc.detach().numpy()
        d['e1/%d' % i] = a.detach().numpy()
        d['e2/%d' % i] = b.detach().numpy()
        d['e1_masked/%d' % i] = em_list[i].detach().numpy()
    pad = em.tensor.rename(None).detach()
    d['e1_masked_padded'] = pad.numpy()
    for k, v in model_d.state_dict().items():
        d['sd/' + k[len('node_embedder.'):]] = v.numpy()
    for k, v in g.items():
        d['grad/' + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, 'ragged_er_b4_2blk.npz'), **d)
    meta['cases']['ragged_er_b4_2blk'] = {'oracle_bit_equal_forward': True}

    # ---------------- layer-level: MlpBlock_Real(16->32, depth 2) + GraphNorm(16) (reference test_layers) ----
    from models.layers import MlpBlock_Real, GraphNorm, normalize
    torch.manual_seed(7)
    mlp = MlpBlock_Real(16, 32, 2)
    gn = GraphNorm(16)
    with torch.no_grad():
        gn.weight.mul_(1.3)
        gn.bias.add_(0.1)
        for c in mlp.convs:
            c.bias.add_(0.05 * torch.randn(c.bias.shape))
    lst = [torch.empty((16, n, n)).normal_() for n in (9, 12, 10)]
    d = {}
    for i, t in enumerate(lst):
        d['x/%d' % i] = t.numpy()
        d['mlp/%d' % i] = mlp(t.unsqueeze(0)).squeeze(0).detach().numpy()
        d['gn/%d' % i] = gn(t.unsqueeze(0)).squeeze(0).detach().numpy()
        d['normalize/%d' % i] = normalize(t.unsqueeze(0)).squeeze(0).numpy()
    for k, v in mlp.state_dict().items():
        d['mlp_sd/' + k] = v.numpy()
    for k, v in gn.state_dict().items():
        d['gn_sd/' + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, 'layers_16to32_depth2.npz'), **d)
    meta['cases']['layers_16to32_depth2'] = {}

    with open(os.path.join(OUT, 'golden_meta.json'), 'w') as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(json.dumps(meta, indent=1, sort_keys=True))


def _pack_pairs(synthetic, x1, x2):
    """bit-packed adjacencies of both sides (channel 0 is the 0/1 adjacency, channel 1 = its row sums)"""
    return synthetic.pack_adjacency(x1[:, 0].numpy()), synthetic.pack_adjacency(x2[:, 0].numpy())


def ref_step_bf16(model, x1, x2):
    """The reference run the way it is trained: parameters and activations in a 16-bit float (the Network.half
    recipe of models/utils.py:71-74 / precision=16 of commander_explore.py:120-122, here bf16 because the CPU has no
    fp16 conv).  Everything -- conv, normalize, matmul, max, scoring, loss -- runs in bf16 on ATen."""
    import copy
    m = copy.deepcopy(model).to(torch.bfloat16)
    m.zero_grad()
    scores = m({'input': x1.to(torch.bfloat16)}, {'input': x2.to(torch.bfloat16)})
    loss = m.loss(scores)
    loss.backward()
    grads = {n[len('node_embedder.'):]: p.grad.detach().float() for n, p in m.named_parameters()}
    return scores.detach().float(), loss.detach().float(), grads


def main_round2():
    """Round-2 fixtures: input contract, cfg2 at the benchmarked batch, the N=200 regime (fp32 / fp64 / bf16 reference
    runs) and a ragged n in [30, 120] batch.  The 4-block weights are those of cfg2_reg_n50_b2_4blk.npz (same seeds)."""
    import_reference()
    sys.path.insert(0, ROOT)
    from graph_neural_net_amd import synthetic
    from oracle import fgnn_oracle as O
    torch.set_num_threads(8)
    with open(os.path.join(OUT, 'golden_meta.json')) as f:
        meta = json.load(f)

    # ---------------- input contract (loaders/data_generator.py:79-87, 118-125) ----------------
    import loaders.data_generator as DG
    rng = np.random.default_rng(7000)
    d = {}
    cases = [('ErdosRenyi', 13, 0.3), ('Regular', 50, 0.2), ('ErdosRenyi', 7, 0.5), ('ErdosRenyi', 200, 0.5)]
    for i, (fam, n, p) in enumerate(cases):
        w = synthetic.random_regular(rng, n, synthetic.regular_degree(n, p)) if fam == 'Regular' else synthetic.erdos_renyi(rng, n, p)
        ref = DG.adjacency_matrix_to_tensor_representation(torch.from_numpy(w))
        mine = torch.from_numpy(synthetic.tensor_representation(w))
        assert ref.dtype == torch.float32 and torch.equal(ref, mine), 'tensor representation differs from the reference'
        # the reference's noise formula on GIVEN noise graphs (its generator is networkx's global RNG): patch the ER
        # generator it calls, record the two probabilities it asks for
        z1 = synthetic.erdos_renyi(rng, n, 0.1)
        z2 = synthetic.erdos_renyi(rng, n, p * 0.1 / (1 - p))
        asked, queue = [], [z1, z2]
        orig = DG.generate_erdos_renyi_netx
        DG.generate_erdos_renyi_netx = lambda pe, nn_: (asked.append((pe, nn_)), (None, torch.from_numpy(queue.pop(0))))[1]
        try:
            wn_ref = DG.noise_erdos_renyi(None, torch.from_numpy(w), 0.1, p)
        finally:
            DG.generate_erdos_renyi_netx = orig
        assert asked[0] == (0.1, n) and abs(asked[1][0] - p * 0.1 / (1 - p)) < 1e-15 and asked[1][1] == n
        wn_mine = w * (1.0 - z1) + (1.0 - w) * z2                      # synthetic.noise_erdos_renyi's arithmetic
        assert torch.equal(wn_ref, torch.from_numpy(wn_mine.astype(np.float32)))
        if n <= 50:
            d['w/%d' % i] = w
            d['repr/%d' % i] = ref.numpy()
            d['z1/%d' % i], d['z2/%d' % i], d['w_noise/%d' % i] = z1, z2, wn_ref.numpy()
            d['pe/%d' % i] = np.array([asked[0][0], asked[1][0], p])
    np.savez_compressed(os.path.join(OUT, 'input_contract.npz'), **d)
    meta['cases']['input_contract'] = {'tensor_representation_equal': True, 'noise_formula_equal': True}

    # ---------------- cfg2 at the benchmarked batch: N=50 Regular, B=32, 4 blocks ----------------
    model = build_reference_model(4, seed=0)
    perturb_(model, 200)
    old = np.load(os.path.join(OUT, 'cfg2_reg_n50_b2_4blk.npz'))
    for k, v in model.state_dict().items():
        assert np.array_equal(old['sd/' + k[len('node_embedder.'):]], v.numpy()), 'weights differ from the cfg2 B=2 fixture'
    x1, x2 = synthetic.make_batch(2000, 32, 50, 'Regular', 0.2, 0.1)
    worst = check_oracle_bit_equal(model, x1, x2, 'cfg2_b32')
    s, l, g = ref_step(model, x1, x2)
    s64, l64, g64 = ref_step(f64(model), x1.double(), x2.double())
    b1, b2 = _pack_pairs(synthetic, x1, x2)
    d = {'bits1': b1, 'bits2': b2, 'n': np.array(50), 'scores': s.numpy(), 'scores64_as_f32': s64.float().numpy(),
         'loss': l.numpy(), 'loss64': l64.numpy()}
    for k, v in g.items():
        d['grad/' + k] = v.numpy()
    for k, v in g64.items():
        d['grad64/' + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, 'cfg2_reg_n50_b32_4blk.npz'), **d)
    meta['cases']['cfg2_reg_n50_b32_4blk'] = {'oracle_bit_equal_forward': True, 'oracle_worst_grad_relerr': worst,
                                              'weights': 'cfg2_reg_n50_b2_4blk.npz sd/*'}

    # ---------------- N=200 regime: dense ER p=.5, B=1, 4 blocks; fp32, fp64 and bf16 reference runs ----------------
    x1, x2 = synthetic.make_batch(4000, 1, 200, 'ErdosRenyi', 0.5, 0.1)
    worst = check_oracle_bit_equal(model, x1, x2, 'cfg4_b1')
    s, l, g = ref_step(model, x1, x2)
    s64, l64, g64 = ref_step(f64(model), x1.double(), x2.double())
    s16, l16, g16 = ref_step_bf16(model, x1, x2)
    b1, b2 = _pack_pairs(synthetic, x1, x2)
    d = {'bits1': b1, 'bits2': b2, 'n': np.array(200), 'scores': s.numpy(), 'scores64_as_f32': s64.float().numpy(),
         'scores_refbf16': s16.numpy(), 'loss': l.numpy(), 'loss64': l64.numpy(), 'loss_refbf16': l16.numpy()}
    for k, v in g.items():
        d['grad/' + k] = v.numpy()
    for k, v in g64.items():
        d['grad64/' + k] = v.numpy()
    for k, v in g16.items():
        d['grad_refbf16/' + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, 'cfg4_er_n200_b1_4blk.npz'), **d)
    names = list(g.keys())
    flat = lambda gg: torch.cat([gg[k].reshape(-1).double() for k in names])
    meta['cases']['cfg4_er_n200_b1_4blk'] = {
        'oracle_bit_equal_forward': True, 'oracle_worst_grad_relerr': worst, 'weights': 'cfg2_reg_n50_b2_4blk.npz sd/*',
        'ref_fp32_vs_fp64_scores': O.max_rel_err(s, s64), 'ref_bf16_vs_fp64_scores': O.max_rel_err(s16, s64),
        'ref_fp32_vs_fp64_flatgrad_l2': float((flat(g) - flat(g64)).norm() / flat(g64).norm()),
        'ref_bf16_vs_fp64_flatgrad_l2': float((flat(g16) - flat(g64)).norm() / flat(g64).norm())}

    # ---------------- ragged n in [30, 120], 4 pairs, 4 blocks: per-graph dense runs, fp32 + fp64 ----------------
    xs, ys = synthetic.make_ragged_batch(5000, 4, 30, 120, 'ErdosRenyi', 0.2, 0.1)
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    s_or, l_or, g_or = O.step_fwd_bwd_ragged(xs, ys, sd)

    def ref_ragged(m, cast):
        m.zero_grad()
        e1 = [m.node_embedder({'input': cast(x).unsqueeze(0)})['ne/suffix'].squeeze(0) for x in xs]
        e2 = [m.node_embedder({'input': cast(y).unsqueeze(0)})['ne/suffix'].squeeze(0) for y in ys]
        scores = [a.t() @ b for a, b in zip(e1, e2)]
        loss, tot = 0, 0
        for sc in scores:   # triplet_loss('mean') on a list (toolbox/losses.py:27-34)
            loss = loss + torch.nn.functional.cross_entropy(sc, torch.arange(sc.shape[0]), reduction='sum')
            tot += sc.shape[0]
        loss = loss / tot
        loss.backward()
        return [sc.detach() for sc in scores], loss.detach(), {n[len('node_embedder.'):]: p.grad.detach().clone() for n, p in m.named_parameters()}

    sr, lr_, gr = ref_ragged(model, lambda t: t)
    for a, b in zip(s_or, sr):
        assert torch.equal(a, b), 'ragged big: oracle scores differ'
    assert torch.equal(l_or, lr_)
    sr64, lr64, gr64 = ref_ragged(f64(model), lambda t: t.double())
    d = {'ns': np.array([x.shape[-1] for x in xs]), 'loss': lr_.numpy(), 'loss64': lr64.numpy()}
    for i, (x, y) in enumerate(zip(xs, ys)):
        d['bits1/%d' % i] = synthetic.pack_adjacency(x[None, 0].numpy())
        d['bits2/%d' % i] = synthetic.pack_adjacency(y[None, 0].numpy())
        d['scores/%d' % i] = sr[i].numpy()
        d['scores64_as_f32/%d' % i] = sr64[i].float().numpy()
    for k, v in gr.items():
        d['grad/' + k] = v.numpy()
    for k, v in gr64.items():
        d['grad64/' + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, 'ragged_er_n30_120_b4_4blk.npz'), **d)
    meta['cases']['ragged_er_n30_120_b4_4blk'] = {'oracle_bit_equal_forward': True, 'weights': 'cfg2_reg_n50_b2_4blk.npz sd/*'}

    # ---------------- triplet_loss('mean_of_mean') of the reference on the cfg1 scores (toolbox/losses.py:14-15) ------
    from toolbox.losses import triplet_loss
    c1 = np.load(os.path.join(OUT, 'cfg1_er_n20_b4_1blk.npz'))
    sc = torch.from_numpy(c1['scores'])
    np.savez_compressed(os.path.join(OUT, 'losses_cfg1.npz'), mean=triplet_loss('mean')(sc).numpy(),
                        mean_of_mean=triplet_loss('mean_of_mean')(sc).numpy())
    meta['cases']['losses_cfg1'] = {}

    with open(os.path.join(OUT, 'golden_meta.json'), 'w') as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(json.dumps(meta, indent=1, sort_keys=True))


def main_widths():
    """Channel widths other than the default 32/32/2 (models/layers.py:113-123 takes any): original_features_num = 3,
    in_features = 16, out_features = 48, depth_of_mlp = 2, 2 blocks -- so the convs are 3->16, 16->16, 19->16, 16->48,
    48->48, 64->48 (odd and non-multiple-of-32 widths).  Constant-size batch (fp32 + fp64 reference runs) and a ragged
    batch run graph by graph through the dense reference model."""
    import_reference()
    sys.path.insert(0, ROOT)
    from oracle import fgnn_oracle as O
    from models.trainers import Siamese_Node_Exp
    torch.set_num_threads(8)
    with open(os.path.join(OUT, 'golden_meta.json')) as f:
        meta = json.load(f)
    torch.manual_seed(11)
    ne = dict(NODE_EMB, num_blocks=2, in_features=16, out_features=48, depth_of_mlp=2)
    model = Siamese_Node_Exp(3, ne)
    perturb_(model, 400)
    g = torch.Generator().manual_seed(401)
    x1 = torch.randn(3, 3, 14, 14, generator=g)
    x2 = torch.randn(3, 3, 14, 14, generator=g)
    worst = check_oracle_bit_equal(model, x1, x2, 'widths')
    s, l, gr = ref_step(model, x1, x2)
    s64, l64, gr64 = ref_step(f64(model), x1.double(), x2.double())
    inter = model.node_embedder({'input': x1})
    d = {'x1': x1.numpy(), 'x2': x2.numpy(), 'scores': s.numpy(), 'loss': l.numpy(), 'scores64': s64.numpy(),
         'loss64': l64.numpy(), 'config': np.array([3, 2, 16, 48, 2])}
    for k, v in model.state_dict().items():
        d['sd/' + k[len('node_embedder.'):]] = v.numpy()
    for k, v in gr.items():
        d['grad/' + k] = v.numpy()
    for k, v in gr64.items():
        d['grad64/' + k] = v.numpy()
    for k in ('ne/bm/block1/mlp1', 'ne/bm/block1/mlp3', 'ne/bm/block2/mlp3', 'ne/suffix'):
        d['inter/' + k] = inter[k].detach().numpy()
    # ragged: graph by graph through the dense model (the reference tests' definition of masked correctness)
    xs = [torch.randn(3, n, n, generator=g) for n in (6, 11, 9)]
    ys = [torch.randn(3, n, n, generator=g) for n in (6, 11, 9)]

    def ref_ragged(m, cast):
        m.zero_grad()
        e1 = [m.node_embedder({'input': cast(x).unsqueeze(0)})['ne/suffix'].squeeze(0) for x in xs]
        e2 = [m.node_embedder({'input': cast(y).unsqueeze(0)})['ne/suffix'].squeeze(0) for y in ys]
        scores = [a.t() @ b for a, b in zip(e1, e2)]
        loss, tot = 0, 0
        for sc in scores:   # triplet_loss('mean') on a list (toolbox/losses.py:27-34)
            loss = loss + torch.nn.functional.cross_entropy(sc, torch.arange(sc.shape[0]), reduction='sum')
            tot += sc.shape[0]
        loss = loss / tot
        loss.backward()
        return ([sc.detach() for sc in scores], loss.detach(),
                {n[len('node_embedder.'):]: p.grad.detach().clone() for n, p in m.named_parameters()})

    sr, lr_, grr = ref_ragged(model, lambda t: t)
    sr64, lr64, grr64 = ref_ragged(f64(model), lambda t: t.double())
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    s_or, l_or, _ = O.step_fwd_bwd_ragged(xs, ys, sd)
    for a, b in zip(s_or, sr):
        assert torch.equal(a, b), 'widths ragged: oracle scores differ'
    assert torch.equal(l_or, lr_)
    d['ragged/ns'] = np.array([x.shape[-1] for x in xs])
    d['ragged/loss'], d['ragged/loss64'] = lr_.numpy(), lr64.numpy()
    for i, (x, y) in enumerate(zip(xs, ys)):
        d['ragged/x1/%d' % i], d['ragged/x2/%d' % i] = x.numpy(), y.numpy()
        d['ragged/scores/%d' % i], d['ragged/scores64/%d' % i] = sr[i].numpy(), sr64[i].numpy()
    for k, v in grr.items():
        d['ragged/grad/' + k] = v.numpy()
    for k, v in grr64.items():
        d['ragged/grad64/' + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, 'widths_c3_16_48_d2_2blk.npz'), **d)
    meta['cases']['widths_c3_16_48_d2_2blk'] = {'oracle_bit_equal_forward': True, 'oracle_worst_grad_relerr': worst}

    # ---- widths that fit inside the 32-wide engine (zero-padded there): 3 -> 16 -> 24, depth 2, 3 blocks ----
    torch.manual_seed(12)
    model = Siamese_Node_Exp(3, dict(NODE_EMB, num_blocks=3, in_features=16, out_features=24, depth_of_mlp=2))
    perturb_(model, 500)
    g = torch.Generator().manual_seed(501)
    x1 = torch.randn(4, 3, 21, 21, generator=g)
    x2 = torch.randn(4, 3, 21, 21, generator=g)
    worst = check_oracle_bit_equal(model, x1, x2, 'widths_narrow')
    s, l, gr = ref_step(model, x1, x2)
    s64, l64, gr64 = ref_step(f64(model), x1.double(), x2.double())
    d = {'x1': x1.numpy(), 'x2': x2.numpy(), 'scores': s.numpy(), 'loss': l.numpy(), 'scores64': s64.numpy(),
         'loss64': l64.numpy(), 'config': np.array([3, 3, 16, 24, 2])}
    for k, v in model.state_dict().items():
        d['sd/' + k[len('node_embedder.'):]] = v.numpy()
    for k, v in gr.items():
        d['grad/' + k] = v.numpy()
    for k, v in gr64.items():
        d['grad64/' + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, 'widths_c3_16_24_d2_3blk.npz'), **d)
    meta['cases']['widths_c3_16_24_d2_3blk'] = {'oracle_bit_equal_forward': True, 'oracle_worst_grad_relerr': worst}
    with open(os.path.join(OUT, 'golden_meta.json'), 'w') as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(json.dumps({k: v for k, v in meta['cases'].items() if k.startswith('widths')}, indent=1, sort_keys=True))


def main_round3():
    """Round-3 fixtures: ONE block, where 16-bit arithmetic is not yet chaotic -- the reference run in bf16 (ref_step_bf16:
    the Network.half recipe of models/utils.py:71-74), fp32 and fp64 on N = 50 regular pairs (B = 2) and one N = 200 dense
    ER pair, with a perturbed 1-block model.  tests/test_gpu_bf16.py gates the HIP bf16 engine against them:
    distance to the fp64 truth <= 1.0 x the reference-bf16 distance, scores and every gradient tensor."""
    import_reference()
    sys.path.insert(0, ROOT)
    from graph_neural_net_amd import synthetic
    from oracle import fgnn_oracle as O
    torch.set_num_threads(8)
    with open(os.path.join(OUT, 'golden_meta.json')) as f:
        meta = json.load(f)
    model = build_reference_model(1, seed=3)
    perturb_(model, 300)
    for tag, (x1, x2) in (('bf16ref_reg_n50_b2_1blk', synthetic.make_batch(2300, 2, 50, 'Regular', 0.2, 0.1)),
                          ('bf16ref_er_n200_b1_1blk', synthetic.make_batch(4300, 1, 200, 'ErdosRenyi', 0.5, 0.1))):
        worst = check_oracle_bit_equal(model, x1, x2, tag)
        s, l, g = ref_step(model, x1, x2)
        s64, l64, g64 = ref_step(f64(model), x1.double(), x2.double())
        s16, l16, g16 = ref_step_bf16(model, x1, x2)
        b1, b2 = _pack_pairs(synthetic, x1, x2)
        d = {'bits1': b1, 'bits2': b2, 'n': np.array(x1.shape[-1]), 'scores': s.numpy(), 'scores64_as_f32': s64.float().numpy(),
             'scores_refbf16': s16.numpy(), 'loss': l.numpy(), 'loss64': l64.numpy(), 'loss_refbf16': l16.numpy()}
        for k, v in model.state_dict().items():
            d['sd/' + k[len('node_embedder.'):]] = v.numpy()
        for k, v in g.items():
            d['grad/' + k] = v.numpy()
        for k, v in g64.items():
            d['grad64/' + k] = v.float().numpy()
        for k, v in g16.items():
            d['grad_refbf16/' + k] = v.numpy()
        np.savez_compressed(os.path.join(OUT, tag + '.npz'), **d)
        names = [k for k in g if not k.endswith('convs.2.bias')]
        flat = lambda gg: torch.cat([gg[k].reshape(-1).double() for k in names])
        l2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
        meta['cases'][tag] = {
            'oracle_bit_equal_forward': True, 'oracle_worst_grad_relerr': worst,
            'ref_fp32_vs_fp64_scores_l2': l2(s, s64), 'ref_bf16_vs_fp64_scores_l2': l2(s16, s64),
            'ref_fp32_vs_fp64_flatgrad_l2': l2(flat(g), flat(g64)), 'ref_bf16_vs_fp64_flatgrad_l2': l2(flat(g16), flat(g64)),
            'ref_bf16_vs_fp64_worst_tensor_l2': max(l2(g16[k], g64[k]) for k in names)}
        print(tag, json.dumps(meta['cases'][tag], indent=1, sort_keys=True))
    with open(os.path.join(OUT, 'golden_meta.json'), 'w') as f:
        json.dump(meta, f, indent=1, sort_keys=True)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'round3':
        main_round3()
    elif len(sys.argv) > 1 and sys.argv[1] == 'round2':
        main_round2()
    elif len(sys.argv) > 1 and sys.argv[1] == 'widths':
        main_widths()
    else:
        main()
