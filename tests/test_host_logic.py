"""CPU: host-side logic (no compute calls into the HIP library)."""
import os
import re
import sys

import numpy as np
import pytest
import torch

from graph_neural_net_amd import _lib, dp, synthetic
from graph_neural_net_amd.engine import ParamLayout
from graph_neural_net_amd.masked import MaskedTensor, from_list
from graph_neural_net_amd.network import build_graph
from util import ROOT, load_golden, sub

NODE_EMB = dict(type='node_embedding', block_init='block_emb', block_inside='block', num_blocks=4,
                in_features=32, out_features=32, depth_of_mlp=3)


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, 'include', 'fgnn_hip.h')).read()
    declared = set(re.findall(r'\b(fgnn_[a-z0-9_]+)\s*\(', hdr))
    declared -= {'fgnn_hip'}
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), 'libfgnn_hip.so does not export %s' % name
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert lib.fgnn_version() >= 1
    assert lib.fgnn_tiles_per_graph(50) == 79
    assert lib.fgnn_mlp_param_count(64, 3) == _lib.mlp_param_count(64, 3) == 4192


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libfgnn_hip.so')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        _lib.load()


def test_cpu_tensor_is_rejected():
    from graph_neural_net_amd.layers import MlpBlock_Real, normalize
    with pytest.raises(RuntimeError, match='GPU'):
        MlpBlock_Real(2, 32, 3)(torch.zeros(1, 2, 4, 4))
    with pytest.raises(RuntimeError, match='GPU'):
        normalize(torch.zeros(1, 2, 4, 4))


def test_param_layout_matches_reference_state_dict():
    d = load_golden('cfg2_reg_n50_b2_4blk.npz')
    sd = sub(d, 'sd/')
    lay = ParamLayout(2, 4, 32, 32, 3)
    assert [e[0] for e in lay.entries] == list(sd.keys())          # reference named_parameters order
    assert all(tuple(sd[n].shape) == tuple(s) for n, _, s in lay.entries)
    assert lay.total == 40000
    flat = lay.flatten(sd, 'cpu')
    back = lay.unflatten(flat)
    assert all(torch.equal(back[k], sd[k]) for k in sd)
    assert lay.mlp[(1, 3)]['cin'] == 34 and lay.mlp[(2, 3)]['cin'] == 64 and lay.mlp[(1, 1)]['cin'] == 2


def test_module_state_dict_keys_and_shapes():
    from graph_neural_net_amd.siamese import Siamese_Node_Exp
    d = load_golden('cfg2_reg_n50_b2_4blk.npz')
    sd = {'node_embedder.' + k: v for k, v in sub(d, 'sd/').items()}
    model = Siamese_Node_Exp(2, dict(NODE_EMB))
    assert list(model.state_dict().keys()) == list(sd.keys())
    model.load_state_dict(sd)     # shapes agree -> a reference checkpoint drops in
    with pytest.raises(NotImplementedError):
        Siamese_Node_Exp(2, dict(NODE_EMB, type='nope'))


def test_build_graph_wiring():
    from graph_neural_net_amd.blocks import node_embedding
    g = build_graph({'input': (None, []), 'ne': node_embedding(2, 2, 32, 32, 3)})
    assert g['ne/bm/block1/mlp1'][1] == ['ne/bm/block1/in']
    assert g['ne/bm/block1/mult'][1] == ['ne/bm/block1/mlp1', 'ne/bm/block1/mlp2']
    assert g['ne/bm/block1/cat'][1] == ['ne/bm/block1/mult', 'ne/bm/block1/in']
    assert g['ne/bm/block1/mlp3'][1] == ['ne/bm/block1/cat']
    assert g['ne/bm/block2/in'][1] == ['ne/bm/block1/mlp3']
    assert g['ne/suffix'][1] == ['ne/bm/block2/mlp3']
    assert g['ne/in'][1] == ['input']


def test_from_list_padding_and_masks_bit_exact():
    lst = [torch.randn(3, n, n) for n in (4, 6, 5)]
    mt = from_list(lst, dims=(1, 2))
    assert mt.tensor.shape == (3, 3, 6, 6) and mt.sizes() == [4, 6, 5]
    for i, (t, u) in enumerate(zip(mt, lst)):
        assert torch.equal(t, u)
    masks = mt.mask_dict
    assert set(masks) == {'N', 'N_'}
    assert torch.equal(masks['N'].rename(None)[0], torch.tensor([1., 1, 1, 1, 0, 0])) and masks['N'].names == ('B', 'N')
    pad = mt.tensor.clone()
    for i, n in enumerate(mt.sizes()):
        pad[i, :, :n, :n] = 0
    assert pad.abs().sum() == 0
    with pytest.raises(ValueError):
        from_list([torch.zeros(2, 3, 4)], dims=(1, 2))


def test_synthetic_generators():
    rng = np.random.default_rng(0)
    w = synthetic.random_regular(rng, 50, 10)
    assert (w == w.T).all() and (np.diag(w) == 0).all() and (w.sum(1) == 10).all()
    assert set(np.unique(w)) <= {0.0, 1.0}
    assert synthetic.regular_degree(50, 0.2) == 10 and synthetic.regular_degree(5, 0.2) == 2  # 5*1 odd -> 2
    x, y = synthetic.make_pair(rng, 20, 'ErdosRenyi', 0.2, 0.1)
    for t in (x, y):
        assert t.shape == (2, 20, 20) and (t[0] == t[0].T).all()
        assert (np.diag(t[1]) == t[0].sum(1)).all() and (t[1] - np.diag(np.diag(t[1])) == 0).all()
    a1, b1 = synthetic.make_batch(7, 3, 12)
    a2, b2 = synthetic.make_batch(7, 3, 12)
    assert torch.equal(a1, a2) and torch.equal(b1, b2) and a1.shape == (3, 2, 12, 12)
    xs, ys = synthetic.make_ragged_batch(3, 5, 6, 11)
    assert all(6 <= t.shape[-1] <= 11 and t.shape == u.shape for t, u in zip(xs, ys))


def test_shard_range_partitions():
    for n, w in ((256, 8), (64, 8), (10, 3), (2, 4)):
        cover = []
        for r in range(w):
            lo, hi = dp.shard_range(n, r, w)
            cover += list(range(lo, hi))
        assert cover == list(range(n))


def test_pack_adjacency_roundtrip():
    rng = np.random.default_rng(1)
    w = np.stack([synthetic.erdos_renyi(rng, 37, 0.4) for _ in range(3)])
    bits = synthetic.pack_adjacency(w)
    assert bits.shape == (3, 37, 2) and bits.dtype == np.uint32
    back = ((bits[:, :, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(3, 37, 64)[:, :, :37]
    assert (back == (w != 0)).all()


def test_checkpoint_roundtrip_reference_format(tmp_path):
    """A Lightning-style {'state_dict': ...} file written in the reference's key/shape convention loads into the flat layout
    (hyper-parameters inferred from the shapes) and round-trips bit-exactly."""
    from graph_neural_net_amd import checkpoint
    from graph_neural_net_amd.engine import ParamLayout
    torch.manual_seed(5)
    lay0 = ParamLayout(2, 3, 32, 32, 2)
    sd = {k: v.clone() for k, v in lay0.unflatten(lay0.init_flat(5, 'cpu')).items()}
    f = tmp_path / 'epoch.1-step.2.ckpt'
    torch.save({'epoch': 1, 'state_dict': {'node_embedder.' + k: v for k, v in sd.items()}}, f)
    layout, flat = checkpoint.load_checkpoint(str(f), 'cpu')
    assert (layout.num_blocks, layout.depth, layout.c0) == (3, 2, 2)
    ref_layout = ParamLayout(2, 3, 32, 32, 2)
    assert torch.equal(flat, ref_layout.flatten(sd, 'cpu'))
    g = tmp_path / 'out.ckpt'
    checkpoint.save_checkpoint(str(g), layout, flat)
    back = torch.load(str(g), weights_only=False)['state_dict']
    assert set(back) == {'node_embedder.' + k for k in sd}
    assert all(torch.equal(back['node_embedder.' + k], sd[k]) for k in sd)
    bad = dict(sd); bad.pop(next(iter(bad)))
    with pytest.raises(RuntimeError):
        checkpoint.load_checkpoint({'state_dict': bad}, 'cpu', layout=layout)


def test_bucket_by_size_partitions_and_orders():
    from graph_neural_net_amd.trainer import FgnnTrainer
    sizes = [30, 120, 47, 48, 49, 16, 17, 120]
    b = FgnnTrainer.bucket_by_size(sizes, 16)
    assert [n for n, _ in b] == [16, 32, 48, 64, 128]
    assert dict(b)[48] == [2, 3] and dict(b)[64] == [4] and dict(b)[128] == [1, 7]
    assert sorted(i for _, idx in b for i in idx) == list(range(len(sizes)))


def test_metrics_on_host_scores_take_the_reference_route():
    """Scores that live on the host (saved scores, evaluation scripts) are handled the reference's way -- per-graph SciPy
    assignment / arg-max (toolbox/metrics.py:92-141) -- dense and MaskedTensor; scores on the GPU never take this route
    (tests/test_gpu_lsap.py: the device kernel, and a missing library raises)."""
    from scipy.optimize import linear_sum_assignment
    from graph_neural_net_amd.masked import from_list
    from graph_neural_net_amd.metrics import accuracy_linear_assignment, accuracy_max
    g = torch.Generator().manual_seed(4)
    s = torch.randn(3, 7, 7, generator=g)
    want = 0
    for b in range(3):
        r, c = linear_sum_assignment((-torch.log_softmax(s[b], -1)).numpy())
        want += int((r == c).sum())
    assert accuracy_linear_assignment(s) == (want, 21)
    assert accuracy_max(s) == (int((s.argmax(-1) == torch.arange(7)).sum()), 21)
    lst = [torch.randn(n, n, generator=g) for n in (4, 6)]
    mt = from_list(lst, dims=(0, 1))
    per = accuracy_linear_assignment(mt, aggregate_score=False)
    for t, a in zip(lst, per):
        r, c = linear_sum_assignment((-torch.log_softmax(t, -1)).numpy())
        assert a == int((r == c).sum()) / t.shape[0]
    assert accuracy_max(mt) == (sum(int((t.argmax(-1) == torch.arange(t.shape[0])).sum()) for t in lst), 10)


def test_bench_algorithmic_model_matches_survey_figures():
    """bench.py's per-pair and per-launch algorithmic figures."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    fl, by = bench.algorithmic_per_pair(50, 4)
    assert abs(fl / 1e9 - 1.3349) < 1e-3 and abs(by / 1e6 - 61.16) < 1e-2
    fl20, by20 = bench.algorithmic_per_pair(20, 1)
    assert abs(fl20 / 1e9 - 0.0385) < 1e-3 and abs(by20 / 1e6 - 1.80) < 5e-2      # cfg1
    b, f = bench.kernel_model('mlp_bwd[cin=32,dx=32]', 64, 50)
    assert b == 4.0 * 64 * 2500 * 128 == 81.92e6 and f == 4.0 * 64 * 2500 * 3072
    b, f = bench.kernel_model('fgnn_chan_matmul_fwd', 64, 50)
    assert b == 4.0 * 64 * 32 * 2500 * 3 and f == 2.0 * 64 * 32 * 50 ** 3


def test_reduce_lr_on_plateau_matches_torch():
    """optim.ReduceLROnPlateau == torch.optim.lr_scheduler.ReduceLROnPlateau with the reference's settings
    (models/trainers.py:92-104: factor 0.5, patience 3, min_lr 1e-5, mode 'min', torch's relative threshold 1e-4)."""
    import torch
    from graph_neural_net_amd.optim import ReduceLROnPlateau

    class _Opt:
        def __init__(self, lr):
            self.lr = lr

    rng = __import__('random').Random(0)
    for trial in range(5):
        series, v = [], 1.0
        for _ in range(80):
            v = v * (1.0 - 0.02 * rng.random()) if rng.random() < 0.35 else v * (1.0 + 0.01 * rng.random())
            series.append(v)
        series[10] = series[9] * (1 - 5e-5)          # an improvement below the relative threshold does not count
        p = torch.nn.Parameter(torch.zeros(1))
        topt = torch.optim.Adam([p], lr=1e-3)
        tsch = torch.optim.lr_scheduler.ReduceLROnPlateau(topt, factor=0.5, patience=3, min_lr=1e-5)
        mine_opt = _Opt(1e-3)
        mine = ReduceLROnPlateau(mine_opt, factor=0.5, patience=3, min_lr=1e-5)
        for m in series:
            tsch.step(m)
            mine.step(m)
            assert abs(mine_opt.lr - topt.param_groups[0]['lr']) < 1e-15, (trial, mine_opt.lr, topt.param_groups[0]['lr'])
        assert mine_opt.lr < 1e-3        # the schedule did something


def test_checkpoint_loads_without_pickle_and_refuses_code_by_default(tmp_path):
    """save_checkpoint writes tensors and plain scalars only (loads with weights_only=True, carries the keys Lightning's
    migration looks at and, optionally, the fused Adam state); a file that needs the pickle loader is refused unless the
    caller opts in."""
    from graph_neural_net_amd import checkpoint
    from graph_neural_net_amd.engine import ParamLayout
    lay = ParamLayout(2, 1, 32, 32, 3)
    flat = lay.init_flat(1, 'cpu')
    f = tmp_path / 'a.ckpt'
    checkpoint.save_checkpoint(str(f), lay, flat, epoch=3, global_step=77)
    obj = torch.load(str(f), weights_only=True)
    assert obj['epoch'] == 3 and obj['global_step'] == 77 and 'pytorch-lightning_version' in obj
    _, back = checkpoint.load_checkpoint(str(f), 'cpu')
    assert torch.equal(back, flat)

    import argparse                     # an arbitrary pickled object, as Lightning's hyper-parameter containers are
    g = tmp_path / 'b.ckpt'
    torch.save({'state_dict': {'node_embedder.' + k: v for k, v in lay.unflatten(flat).items()},
                'hparams': argparse.Namespace(lr=1e-3)}, str(g))
    with pytest.raises(RuntimeError, match='allow_pickle'):
        checkpoint.load_checkpoint(str(g), 'cpu')
    _, back = checkpoint.load_checkpoint(str(g), 'cpu', allow_pickle=True)       # explicit opt-in for a trusted file
    assert torch.equal(back, flat)


def test_committed_bench_lines_follow_the_contract():
    """The bench.py lines committed under profiles/ (what the round's numbers are quoted from) carry every field of the
    driver contract, a consistent value / ms_per_step pair and the roofline / cpu_baseline objects."""
    import glob, json, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, 'profiles', 'archive', 'r02_*_bench*.json')) + glob.glob(os.path.join(root, 'profiles', 'r06_*bench*.json')))
    assert files
    for f in files:
        d = json.loads(open(f).read().strip().split('\n')[-1])
        for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                  'vs_baseline', 'dtype', 'data', 'config', 'roofline'):
            assert k in d, (f, k)
        assert d['unit'] == 'pairs/s' and d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None
        assert 'workload' in d['config'] and 'model' not in d['config']
        pairs = d['config']['global_batch']
        assert abs(d['value'] - pairs / (d['ms_per_step'] * 1e-3)) < 1e-6 * d['value']
        r = d['roofline']
        assert r['bound'] in ('hbm', 'mfma') and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9 and 0 < r['frac'] < 1
        assert d['dtype'] == ('bf16' if 'bf16' in d['metric'] else 'f32')
        if d.get('cpu_baseline'):
            c = d['cpu_baseline']
            assert c['kind'] == 'port' and c['cores'] >= 1 and c['value'] > 0 and 'sample' in c


def test_bench_self_launch_relays_the_worst_return_code():
    """`python bench.py --gpus 2` without RANK in the environment starts its own rank processes (before touching the GPU) and
    exits with their worst return code: without a GPU both ranks stop with "bench.py needs a GPU" (there is no CPU path)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    if torch.cuda.is_available():
        pytest.skip('a GPU is present: covered by tests/test_00_gpu_two_ranks.py::test_bench_launches_its_own_ranks')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--steps', '1', '--warmup', '0'],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert 'needs a GPU' in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith('{')]

def test_structured_block1_host_geometry():
    """The host-side geometry functions of csrc/block1_struct.hip (no GPU): what shapes it covers, the workspace size, and which
    partial rows its backward writes -- the engine's reduction reads exactly those (fgnn_grad_job.rows)."""
    lib = _lib.load()
    assert lib.fgnn_block1_struct_supported(256, 3, 2) == 1 and lib.fgnn_block1_struct_supported(1, 3, 2) == 1
    assert lib.fgnn_block1_struct_supported(257, 3, 2) == 0 and lib.fgnn_block1_struct_supported(50, 2, 2) == 0
    assert lib.fgnn_block1_struct_supported(50, 3, 3) == 0
    nwg = lib.fgnn_mlp_bwd_num_workgroups()
    for G, N in ((64, 50), (16, 200), (16, 120), (256, 50), (80, 100), (2, 1), (4, 256)):
        rows = lib.fgnn_block1_struct_rows(G, N)
        chunks = (N + 2 + 31) // 32                      # rounds of 32 class instances (2 off-diagonal classes + one per vertex)
        assert rows % G == 0 and G <= rows <= nwg and rows // G <= chunks, (G, N, rows)
        assert rows // G == min(chunks, max(1, nwg // G)), (G, N, rows)
        ws = lib.fgnn_block1_struct_ws_floats(G, N)
        cs = 2 + (64 if N <= 64 else 128 if N <= 128 else 256)
        cp = (N + 7) // 8 * 8
        assert ws % 4 == 0 and ws >= 2 * G * 32 * cs + 2 * G * 32 * 4 + 4 * G * N + G + G * N * cp // 2, (G, N, ws)
    for G in (257, 300, 512, 1024):                      # more graphs than rows: row b sums the graphs b, b + nwg, ...
        assert lib.fgnn_block1_struct_rows(G, 50) == nwg
    assert lib.fgnn_block1_struct_table_floats(50) == 2 * (2 + 2 * 51) * 4 * 32
