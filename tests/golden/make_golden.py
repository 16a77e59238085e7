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


def main_wide64():
    """The 64-feature model (original_features_num = 2, in_features = out_features = 64, depth_of_mlp = 3, 4 blocks: the widths of
    csrc/mlp64.hip -- convs 2->64, 66->64, 64->64, 128->64): a constant-size batch of regular-graph pairs at N = 50 (fp32 + fp64
    reference runs, intermediates of block 1 / block 4) and a ragged batch run graph by graph through the dense reference model."""
    import_reference()
    sys.path.insert(0, ROOT)
    from oracle import fgnn_oracle as O
    from graph_neural_net_amd import synthetic
    from models.trainers import Siamese_Node_Exp
    torch.set_num_threads(8)
    with open(os.path.join(OUT, 'golden_meta.json')) as f:
        meta = json.load(f)
    torch.manual_seed(21)
    ne = dict(NODE_EMB, num_blocks=4, in_features=64, out_features=64, depth_of_mlp=3)
    model = Siamese_Node_Exp(2, ne)
    perturb_(model, 600)
    x1, x2 = synthetic.make_batch(601, 2, 50, 'Regular', 0.2, 0.1)
    worst = check_oracle_bit_equal(model, x1, x2, 'wide64')
    s, l, gr = ref_step(model, x1, x2)
    s64, l64, gr64 = ref_step(f64(model), x1.double(), x2.double())
    inter = model.node_embedder({'input': x1})
    d = {'x1': x1.numpy(), 'x2': x2.numpy(), 'scores': s.numpy(), 'loss': l.numpy(), 'scores64': s64.numpy(),
         'loss64': l64.numpy(), 'config': np.array([2, 4, 64, 64, 3])}
    for k, v in model.state_dict().items():
        d['sd/' + k[len('node_embedder.'):]] = v.numpy()
    for k, v in gr.items():
        d['grad/' + k] = v.numpy()
    for k, v in gr64.items():
        d['grad64/' + k] = v.numpy().astype(np.float32)        # (the fp64 run rounded once: 2^-24 against a yard-stick of 1e-6)
    inter64 = f64(model).node_embedder({'input': x1.double()})
    for k in ('ne/bm/block1/mlp1', 'ne/bm/block1/mlp3', 'ne/suffix'):
        d['inter/' + k] = inter[k].detach().numpy()[:1]
        d['inter64/' + k] = inter64[k].detach().numpy()[:1].astype(np.float32)
    g = torch.Generator().manual_seed(602)
    xs = [torch.randn(2, n, n, generator=g) for n in (20, 33, 27)]
    ys = [torch.randn(2, n, n, generator=g) for n in (20, 33, 27)]

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
        assert torch.equal(a, b), 'wide64 ragged: oracle scores differ'
    assert torch.equal(l_or, lr_)
    d['ragged/ns'] = np.array([x.shape[-1] for x in xs])
    d['ragged/loss'], d['ragged/loss64'] = lr_.numpy(), lr64.numpy()
    for i, (x, y) in enumerate(zip(xs, ys)):
        d['ragged/x1/%d' % i], d['ragged/x2/%d' % i] = x.numpy(), y.numpy()
        d['ragged/scores/%d' % i], d['ragged/scores64/%d' % i] = sr[i].numpy(), sr64[i].numpy()
    for k, v in grr.items():
        d['ragged/grad/' + k] = v.numpy()
    for k, v in grr64.items():
        d['ragged/grad64/' + k] = v.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(OUT, 'wide64_c2_64_64_d3_4blk.npz'), **d)
    meta['cases']['wide64_c2_64_64_d3_4blk'] = {'oracle_bit_equal_forward': True, 'oracle_worst_grad_relerr': worst}
    with open(os.path.join(OUT, 'golden_meta.json'), 'w') as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(json.dumps({k: v for k, v in meta['cases'].items() if k.startswith('wide64')}, indent=1, sort_keys=True))


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


# ----------------------------------------------------------------------------------------------------------------------
# Round 4: the multi-seed single-pair gradient fixture (tests/test_gpu_grad_gate.py)
# ----------------------------------------------------------------------------------------------------------------------
ZERO_GRAD_SUFFIX = 'convs.2.bias'        # analytically zero gradient (GraphNorm removes the mean): excluded from every norm


class _Tap:
    """Forward hooks on the imported reference: every conv output that feeds a ReLU (convs[:-1] of every MlpBlock_Real,
    models/layers.py:128-130) and the input of ColumnMaxPooling (models/layers.py:202-203), in call order (the siamese forward
    runs the embedder once per side)."""

    def __init__(self, model):
        from models.layers import MlpBlock_Real, ColumnMaxPooling
        self.relu_in, self.pool_in, self.handles = [], [], []
        for name, mod in model.node_embedder.named_modules():
            if isinstance(mod, MlpBlock_Real):
                for i, conv in enumerate(list(mod.convs)[:-1]):
                    self.handles.append(conv.register_forward_hook(
                        lambda m, inp, out, tag='%s.convs.%d' % (name, i): self.relu_in.append((tag, out.detach().clone()))))
            elif isinstance(mod, ColumnMaxPooling):
                self.handles.append(mod.register_forward_hook(lambda m, inp, out: self.pool_in.append(inp[0].detach().clone())))

    def remove(self):
        for h in self.handles:
            h.remove()


def _tapped_step(model, x1, x2):
    tap = _Tap(model)
    try:
        s, l, g = ref_step(model, x1, x2)
    finally:
        tap.remove()
    return s, l, g, tap


def _margins(t64, t32a, t32b):
    """Per tapped tensor, from the fp64 run and the reference's two fp32 runs (8 threads, 1 thread):
    ReLU inputs  -> [rms, E8, E1, min |pre| over the non-zero entries, #(|pre| < E), #(|pre| < 4 E), #(|pre| < 16 E), #sign flips 8t, #sign flips 1t]
    pooling input-> [rms, E8, E1, min top-1/top-2 gap over the rows, #(gap < 2 E), #(gap < 8 E), #(gap < 32 E), #arg-max flips 8t, #arg-max flips 1t]
    with E = max(E8, E1), E8 / E1 = max |fp32 - fp64| over the tensor: the reference's OWN fp32 error at that point of the graph.
    An entry that is exactly 0 in all three runs (W x + b with x = 0, b = 0) is not a tie: every evaluation takes the same branch."""
    relu, pool = [], []
    for (tag, a64), (_, a8), (_, a1) in zip(t64.relu_in, t32a.relu_in, t32b.relu_in):
        e8 = (a8.double() - a64).abs().max().item()
        e1 = (a1.double() - a64).abs().max().item()
        E = max(e8, e1)
        live = ~((a64 == 0) & (a8 == 0) & (a1 == 0))
        mag = a64.abs()[live]
        relu.append([a64.pow(2).mean().sqrt().item(), e8, e1, mag.min().item() if mag.numel() else float('inf'),
                     int((mag < E).sum()), int((mag < 4 * E).sum()), int((mag < 16 * E).sum()),
                     int((((a8 > 0) != (a64 > 0)) & live).sum()), int((((a1 > 0) != (a64 > 0)) & live).sum())])
    for a64, a8, a1 in zip(t64.pool_in, t32a.pool_in, t32b.pool_in):
        e8 = (a8.double() - a64).abs().max().item()
        e1 = (a1.double() - a64).abs().max().item()
        E = max(e8, e1)
        if a64.shape[-1] > 1:
            top = a64.topk(2, dim=-1).values
            gap = (top[..., 0] - top[..., 1]).reshape(-1)
        else:
            gap = torch.full((1,), float('inf'), dtype=torch.float64)
        i64, i8, i1 = a64.argmax(-1), a8.argmax(-1), a1.argmax(-1)
        pool.append([a64.pow(2).mean().sqrt().item(), e8, e1, gap.min().item(), int((gap < 2 * E).sum()), int((gap < 8 * E).sum()),
                     int((gap < 32 * E).sum()), int((i8 != i64).sum()), int((i1 != i64).sum())])
    return np.array(relu, dtype=np.float64), np.array(pool, dtype=np.float64)


def _grad_case(model, m64, x1, x2, names):
    """One single-pair case: the reference in fp32 with 8 threads, in fp32 with 1 thread (another GEMM blocking = another
    summation order) and in fp64, all three with the decision taps.  Returns what the fixture stores."""
    keep = [k for k in names if not k.endswith(ZERO_GRAD_SUFFIX)]
    flat = lambda g, ks: torch.cat([g[k].reshape(-1).double() for k in ks])
    torch.set_num_threads(8)
    s8, l8, g8, t8 = _tapped_step(model, x1, x2)
    torch.set_num_threads(1)
    s1, l1, g1, t1 = _tapped_step(model, x1, x2)
    torch.set_num_threads(8)
    s64, l64, g64, t64 = _tapped_step(m64, x1.double(), x2.double())
    f64 = flat(g64, keep)
    nrm = f64.norm().item()
    den = nrm if nrm > 0 else 1.0
    err8 = ((flat(g8, keep) - f64).norm() / den).item()
    err1 = ((flat(g1, keep) - f64).norm() / den).item()
    trel = lambda a, b: ((a.double() - b).abs().max() / b.abs().max()).item() if b.abs().max() > 0 else (a.double() - b).abs().max().item()
    terr8 = [trel(g8[k], g64[k]) for k in names]
    terr1 = [trel(g1[k], g64[k]) for k in names]
    relu, pool = _margins(t64, t8, t1)
    srel = lambda a: ((a.double() - s64).abs().max() / s64.abs().max()).item() if s64.abs().max() > 0 else 0.0
    return {'g64': flat(g64, names).float().numpy(), 'gnorm64': nrm, 'err8': err8, 'err1': err1,
            'terr8': np.array(terr8), 'terr1': np.array(terr1), 'relu': relu, 'pool': pool,
            'scores64': s64.float().numpy()[0], 'loss64': l64.item(), 'score_err8': srel(s8), 'score_err1': srel(s1),
            'loss8': l8.item()}


def main_round4():
    """The multi-seed gradient fixture: >= 128 SINGLE-PAIR cases, each with the reference's fp64
    gradient (stored as fp32: its rounding, 3e-8 relative in L2, is far below every error it is compared with), the L2 /
    per-tensor errors of the reference's own fp32 gradient evaluated with 8 threads and with 1 thread, fp64 scores and loss,
    and the decision margins of the fp64 forward (the taps of _Tap) in units of the reference's own fp32 error at the same
    tensor.  Group A: the 32 pairs of the benchmarked batch, one at a time (cfg2 weights, 4 blocks).  Group B: 116 pairs of the
    shapes of test_degenerate_and_boundary_shapes and in between (N = 1 ... 97, several seeds each) on a perturbed 2-block
    model.  tests/test_gpu_grad_gate.py applies ONE gate function to every engine variant on these cases."""
    import_reference()
    sys.path.insert(0, ROOT)
    from graph_neural_net_amd import synthetic
    torch.set_num_threads(8)
    with open(os.path.join(OUT, 'golden_meta.json')) as f:
        meta = json.load(f)
    out = {}
    summary = {}

    def run_group(tag, model, cases):
        m64 = f64(model)
        names = [n[len('node_embedder.'):] for n, _ in model.named_parameters()]
        rows = []
        for (x1, x2, label) in cases:
            worst = check_oracle_bit_equal(model, x1, x2, '%s %s' % (tag, label))       # the oracle stays pinned on every case
            assert worst < 1e-5
            rows.append(_grad_case(model, m64, x1, x2, names))
            r = rows[-1]
            print('%s %-12s err8 %.2e err1 %.2e  relu flips %d/%d  argmax flips %d/%d  near(4E) relu %d pool(8E) %d' % (
                tag, label, r['err8'], r['err1'], r['relu'][:, 7].sum(), r['relu'][:, 8].sum(), r['pool'][:, 7].sum(),
                r['pool'][:, 8].sum(), r['relu'][:, 5].sum(), r['pool'][:, 5].sum()), flush=True)
        out[tag + '/names'] = np.array([len(names)])
        out[tag + '/n'] = np.array([c[0].shape[-1] for c in cases])
        out[tag + '/label'] = np.array([hash(c[2]) % (1 << 31) for c in cases])
        nmax = max(c[0].shape[-1] for c in cases)
        words = (nmax + 31) // 32
        bits = np.zeros((len(cases), 2, nmax, words), dtype=np.uint32)
        for i, (x1, x2, _) in enumerate(cases):
            n = x1.shape[-1]
            w = (n + 31) // 32
            bits[i, 0, :n, :w] = synthetic.pack_adjacency(x1[:, 0].numpy())[0]
            bits[i, 1, :n, :w] = synthetic.pack_adjacency(x2[:, 0].numpy())[0]
        out[tag + '/bits'] = bits
        out[tag + '/g64'] = np.stack([r['g64'] for r in rows])
        for k in ('gnorm64', 'err8', 'err1', 'loss64', 'loss8', 'score_err8', 'score_err1'):
            out[tag + '/' + k] = np.array([r[k] for r in rows], dtype=np.float64)
        out[tag + '/terr8'] = np.stack([r['terr8'] for r in rows])
        out[tag + '/terr1'] = np.stack([r['terr1'] for r in rows])
        out[tag + '/relu'] = np.stack([r['relu'] for r in rows])          # (case, tapped tensor, 9)
        out[tag + '/pool'] = np.stack([r['pool'] for r in rows])          # (case, side, 9)
        sc = np.zeros((len(cases), nmax, nmax), dtype=np.float32)
        for i, r in enumerate(rows):
            n = r['scores64'].shape[-1]
            sc[i, :n, :n] = r['scores64']
        out[tag + '/scores64'] = sc
        e8, e1 = out[tag + '/err8'], out[tag + '/err1']
        ok = (e8 > 0) & (e1 > 0)
        ratio = np.where(ok, e1 / np.where(ok, e8, 1.0), 1.0)
        summary[tag] = {'cases': len(cases), 'err8_median': float(np.median(e8)), 'err1_median': float(np.median(e1)),
                        'err8_max': float(e8.max()), 'err1_max': float(e1.max()),
                        'ratio_1t_over_8t_gt3': int((ratio > 3).sum()), 'ratio_1t_over_8t_lt_third': int((ratio < 1 / 3).sum())}

    # ---- group A: the benchmarked batch, pair by pair ----
    model = build_reference_model(4, seed=0)
    perturb_(model, 200)
    old = np.load(os.path.join(OUT, 'cfg2_reg_n50_b2_4blk.npz'))
    for k, v in model.state_dict().items():
        assert np.array_equal(old['sd/' + k[len('node_embedder.'):]], v.numpy()), 'weights differ from the cfg2 fixture'
    x1, x2 = synthetic.make_batch(2000, 32, 50, 'Regular', 0.2, 0.1)
    run_group('A', model, [(x1[b:b + 1], x2[b:b + 1], 'cfg2[%d]' % b) for b in range(32)])

    # ---- group B: boundary shapes and in between, perturbed 2-block model ----
    model = build_reference_model(2, seed=1)
    perturb_(model, 300)
    for k, v in model.state_dict().items():
        out['B/sd/' + k[len('node_embedder.'):]] = v.numpy()
    cases = []
    plan = [(1, 2), (2, 2), (3, 4), (5, 6), (7, 8), (8, 8), (10, 8), (12, 8), (14, 8), (16, 8), (18, 8), (20, 8), (26, 6), (31, 6),
            (33, 6), (40, 4), (50, 4), (64, 4), (65, 4), (97, 4)]
    for n, reps in plan:
        for r in range(reps):
            a, b = synthetic.make_batch(9000 + 100 * n + r, 1, n, 'ErdosRenyi', 0.5, 0.1)
            cases.append((a, b, 'n%d[%d]' % (n, r)))
    run_group('B', model, cases)
    out['B/seed'] = np.array([9000 + 100 * n + r for n, reps in plan for r in range(reps)])

    np.savez_compressed(os.path.join(OUT, 'gradgate_single_pairs.npz'), **out)
    meta['cases']['gradgate_single_pairs'] = dict(summary, weights={'A': 'cfg2_reg_n50_b2_4blk.npz sd/*', 'B': 'B/sd/*'},
                                                  oracle_bit_equal_forward=True)
    with open(os.path.join(OUT, 'golden_meta.json'), 'w') as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(json.dumps(summary, indent=1, sort_keys=True))


def tap_decisions(tap, B):
    """The taps of one siamese step (side 1, then side 2) as the decisions of the STACKED batch cat(x1, x2): ({(blk, mlp, layer): bool
    (2B, C, N, N)}, idx (2B, C, N))."""
    per_side = len(tap.relu_in) // 2
    masks = {}
    for (tag, a), (_, b) in zip(tap.relu_in[:per_side], tap.relu_in[per_side:]):
        # tag = 'ne_bm_block<k>_mlp<j>.convs.<l>'
        blk = int(tag.split('block')[1].split('_')[0])
        j = int(tag.split('_mlp')[1].split('.')[0])
        l = int(tag.rsplit('.', 1)[1])
        masks[(blk, j, l)] = torch.cat([a, b]) > 0
    idx = torch.cat([tap.pool_in[0], tap.pool_in[1]]).max(-1)[1]
    return masks, idx


def check_pinned_oracle(model, x1, x2, tag):
    """oracle/fgnn_oracle_pinned.py fed the reference's OWN decisions is torch.equal to the reference: scores, loss, every gradient,
    in the model's dtype.  Returns (decisions, (scores, loss, grads) of the reference)."""
    sys.path.insert(0, ROOT)
    from oracle import fgnn_oracle_pinned as OP
    dt = next(model.parameters()).dtype
    s, l, g, tap = _tapped_step(model, x1.to(dt), x2.to(dt))
    masks, idx = tap_decisions(tap, x1.shape[0])
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    s2, l2, g2 = OP.step_fwd_bwd_pinned(x1, x2, sd, masks, idx, dtype=dt)
    assert torch.equal(s, s2), tag + ': pinned scores differ'
    assert torch.equal(l, l2), tag + ': pinned loss differs'
    for k in g:
        assert torch.equal(g[k], g2[k]), tag + ': pinned gradient %s differs' % k
    # ... and the plain oracle's own decision collector sees what the hooks saw
    m3, i3 = OP.collect_decisions(torch.cat([x1, x2]).to(dt), {k: v for k, v in sd.items()})
    # (the stacked 2B batch goes through ATen's conv in one call: same values per sample)
    assert torch.equal(i3, idx) and all(torch.equal(m3[k], masks[k]) for k in masks), tag + ': collect_decisions differs'
    return (masks, idx), (s, l, g)


def main_round5():
    """tests/golden/pinned_decisions.npz: a small step (2 blocks, perturbed weights, 3 pairs, N = 14) with the reference's decisions
    and results in fp32 AND in fp64, each checked torch.equal against oracle/fgnn_oracle_pinned.py fed those decisions; plus the
    cross evaluation the GPU test relies on -- the fp64 arithmetic on the branch the fp32 run took."""
    import_reference()
    sys.path.insert(0, ROOT)
    from graph_neural_net_amd import synthetic
    from oracle import fgnn_oracle_pinned as OP
    model = build_reference_model(2, seed=21)
    perturb_(model, 210)
    x1, x2 = synthetic.make_batch(2100, 3, 14, 'ErdosRenyi', 0.35, 0.1)
    out = {'x1': x1.numpy(), 'x2': x2.numpy()}
    for k, v in model.state_dict().items():
        out['sd/' + k[len('node_embedder.'):]] = v.numpy()
    (m32, i32), (s32, l32, g32) = check_pinned_oracle(model, x1, x2, 'fp32')
    m64_model = f64(model)
    (m64, i64), (s64, l64, g64) = check_pinned_oracle(m64_model, x1, x2, 'fp64')
    for tag, (m, i, s, l, g) in (('f32', (m32, i32, s32, l32, g32)), ('f64', (m64, i64, s64, l64, g64))):
        for k, v in OP.pack_decisions(m, i).items():
            out['%s/%s' % (tag, k)] = v
        out[tag + '/scores'] = s.numpy()
        out[tag + '/loss'] = np.array(l.item())
        for k, v in g.items():
            out['%s/grad/%s' % (tag, k)] = v.numpy()
    # the fp64 arithmetic on the fp32 run's branch (what tests/test_gpu_grad_pinned.py computes from the ENGINE's decisions)
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    sx, lx, gx = OP.step_fwd_bwd_pinned(x1, x2, sd, m32, i32, dtype=torch.float64)
    out['x64on32/loss'] = np.array(lx.item())
    for k, v in gx.items():
        out['x64on32/grad/' + k] = v.numpy()
    flips = sum(int((m32[k] != m64[k]).sum()) for k in m32) + int((i32 != i64).sum())
    np.savez_compressed(os.path.join(OUT, 'pinned_decisions.npz'), **out)
    with open(os.path.join(OUT, 'golden_meta.json')) as f:
        meta = json.load(f)
    meta['cases']['pinned_decisions'] = {'pairs': 3, 'n': 14, 'blocks': 2, 'decisions_fp32_vs_fp64_differ': flips,
                                         'pinned_oracle_bit_equal': ['fp32', 'fp64']}
    with open(os.path.join(OUT, 'golden_meta.json'), 'w') as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print('pinned_decisions.npz written; decisions that differ between the fp32 and the fp64 run:', flips)


def main_round5b():
    """tests/golden/pinned_reference_errors.npz: the yard-stick of tests/test_gpu_grad_pinned.py.  For the three benchmarked shapes
    (cfg2: the full 32-pair batch; cfg4 shape: 8 dense ER pairs of N = 200; cfg5: 8 ragged pairs, n in [30, 120]; weights of
    cfg2_reg_n50_b2_4blk.npz) the per-tensor max-norm relative error of the REFERENCE's own fp32 gradients (8 threads and 1 thread)
    against the fp64 evaluation of the branch that fp32 run took (oracle/fgnn_oracle_pinned.py with the fp32 run's decisions) -- i.e.
    what pure fp32 rounding costs the reference once the decisions are out of the comparison.  The oracle stands in for the
    reference (torch.equal to it: check_oracle_bit_equal / check_pinned_oracle, re-checked here on a slice of every case)."""
    import_reference()
    sys.path.insert(0, ROOT)
    from graph_neural_net_amd import synthetic
    from oracle import fgnn_oracle as O, fgnn_oracle_pinned as OP
    d = np.load(os.path.join(OUT, 'cfg2_reg_n50_b2_4blk.npz'))
    sd = {k[3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith('sd/')}
    names = list(sd.keys())
    model = build_reference_model(4, seed=0)
    model.load_state_dict({'node_embedder.' + k: v for k, v in sd.items()})
    out = {'names': np.array(names)}
    trel = lambda a, b: ((a.double() - b).abs().max() / b.abs().max()).item() if b.abs().max() > 0 else (a.double() - b).abs().max().item()

    def case(tag, x1, x2, sizes=None):
        if sizes is None:
            check_pinned_oracle(model, x1[:2], x2[:2], tag)                       # the pinned oracle == the reference, on a slice
            run32 = lambda: O.step_fwd_bwd(x1, x2, sd)

            def decide():           # side by side, like the step itself (ATen's blocking -- and with it the rounding -- depends on the batch)
                (ma, ia), (mb, ib) = OP.collect_decisions(x1, sd), OP.collect_decisions(x2, sd)
                return {k: torch.cat([ma[k], mb[k]]) for k in ma}, torch.cat([ia, ib])
        else:
            xs = [x1[b, :, :n, :n] for b, n in enumerate(sizes)]
            ys = [x2[b, :, :n, :n] for b, n in enumerate(sizes)]
            run32 = lambda: O.step_fwd_bwd_ragged(xs, ys, sd)

            def decide():
                B, nmax = len(sizes), x1.shape[-1]
                masks, idx = {}, torch.zeros(2 * B, 32, nmax, dtype=torch.int64)
                for b, n in enumerate(sizes):
                    for gi, t in ((b, xs[b]), (B + b, ys[b])):
                        m, i = OP.collect_decisions(t.unsqueeze(0), sd)
                        for k, v in m.items():
                            masks.setdefault(k, torch.zeros(2 * B, 32, nmax, nmax, dtype=torch.bool))[gi, :, :n, :n] = v[0]
                        idx[gi, :, :n] = i[0]
                return masks, idx
        errs, lerr = {}, []
        for threads in (8, 1):
            # each fp32 run against the fp64 evaluation of ITS OWN branch (another GEMM blocking = another rounding of a
            # pre-activation near zero: the two runs do not take the same decisions at N > 64)
            torch.set_num_threads(threads)
            _, l32, g32 = run32()
            masks, idx = decide()
            torch.set_num_threads(8)
            if sizes is None:
                _, l64, g64 = OP.step_fwd_bwd_pinned(x1, x2, sd, masks, idx, dtype=torch.float64)
            else:
                _, l64, g64 = OP.step_fwd_bwd_pinned_ragged(x1, x2, sizes, sd, masks, idx, dtype=torch.float64)
            errs[threads] = np.array([trel(g32[k], g64[k]) for k in names])
            lerr.append(abs(l32.item() - l64.item()) / abs(l64.item()))
            del masks, idx, g64
        e8, e1 = errs[8], errs[1]
        out[tag + '/err8'], out[tag + '/err1'] = e8, e1
        out[tag + '/loss_err'] = np.array(lerr)
        live = np.array([not k.endswith(ZERO_GRAD_SUFFIX) for k in names])
        print(tag, 'reference fp32 vs fp64 on its own branch: worst tensor 8 threads %.2e (%s), 1 thread %.2e; median %.2e / %.2e'
              % (e8[live].max(), names[int(np.argmax(np.where(live, e8, 0)))], e1[live].max(), np.median(e8[live]), np.median(e1[live])), flush=True)

    x1, x2 = synthetic.make_batch(2000, 32, 50, 'Regular', 0.2, 0.1)
    case('cfg2', x1, x2)
    xs, ys = synthetic.make_ragged_batch(5000, 8, 30, 120, 'ErdosRenyi', 0.2, 0.1)
    sizes = [int(t.shape[-1]) for t in xs]
    N = max(sizes)
    pad = lambda lst: torch.stack([torch.nn.functional.pad(t, (0, N - t.shape[-1], 0, N - t.shape[-1])) for t in lst])
    case('cfg5', pad(xs), pad(ys), sizes)
    x1, x2 = synthetic.make_batch(4000, 8, 200, 'ErdosRenyi', 0.5, 0.1)
    case('cfg4', x1, x2)
    np.savez_compressed(os.path.join(OUT, 'pinned_reference_errors.npz'), **out)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'round5b':
        main_round5b()
    elif len(sys.argv) > 1 and sys.argv[1] == 'round5':
        main_round5()
    elif len(sys.argv) > 1 and sys.argv[1] == 'round4':
        main_round4()
    elif len(sys.argv) > 1 and sys.argv[1] == 'round3':
        main_round3()
    elif len(sys.argv) > 1 and sys.argv[1] == 'round2':
        main_round2()
    elif len(sys.argv) > 1 and sys.argv[1] == 'widths':
        main_widths()
    elif len(sys.argv) > 1 and sys.argv[1] == 'wide64':
        main_wide64()
    else:
        main()
