"""Per-tensor gradient error table (ours vs fp64, reference/oracle fp32 vs fp64) for the two large fp32 parity cases:
python tests/diag/gpu_graderr_table.py [n200|ragged]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from graph_neural_net_amd import synthetic
from oracle import fgnn_oracle as O
from util import load_golden, sub, rel, is_zero_grad, unpack_pairs
from test_gpu_parity import _run_engine
which = sys.argv[1] if len(sys.argv) > 1 else 'n200'
sd = sub(load_golden('cfg2_reg_n50_b2_4blk.npz'), 'sd/')
if which == 'n200':
    x1, x2 = synthetic.make_batch(4000, 8, 200, 'ErdosRenyi', 0.5, 0.1)
    _, _, g32 = O.step_fwd_bwd(x1, x2, sd)
    _, _, g64 = O.step_fwd_bwd(x1.double(), x2.double(), {k: v.double() for k, v in sd.items()})
    # a second, equally valid fp32 evaluation: one thread (different GEMM blocking / summation order)
    torch.set_num_threads(1)
    _, _, g32b = O.step_fwd_bwd(x1, x2, sd)
    torch.set_num_threads(8)
    _, _, _, scores, loss, grads = _run_engine(sd, x1, x2, 4)
else:
    d = load_golden('ragged_er_n30_120_b4_4blk.npz')
    ns = [int(v) for v in d['ns']]
    xs = [unpack_pairs(d['bits1/%d' % i], n)[0] for i, n in enumerate(ns)]
    ys = [unpack_pairs(d['bits2/%d' % i], n)[0] for i, n in enumerate(ns)]
    x1, nv = O.pad_graph_list(xs); x2, _ = O.pad_graph_list(ys)
    g32 = sub(d, 'grad/'); g64 = sub(d, 'grad64/'); g32b = None
    _, _, _, scores, loss, grads = _run_engine(sd, x1, x2, 4, nvalid=nv)
keys = [k for k in g32 if not is_zero_grad(k)]
flat = lambda g: torch.cat([g[k].reshape(-1).double() for k in keys])
f64 = flat(g64)
print('flat: ours %.3e  ref32 %.3e' % ((flat(grads) - f64).norm() / f64.norm(), (flat(g32) - f64).norm() / f64.norm()),
      ('ref32(1 thread) %.3e' % ((flat(g32b) - f64).norm() / f64.norm())) if g32b else '')
for k in keys:
    print('%-44s ours %.2e  ref32 %.2e %s ratio %.1f' % (k, rel(grads[k], g64[k]), rel(g32[k], g64[k]),
          ('ref32b %.2e' % rel(g32b[k], g64[k])) if g32b else '', rel(grads[k], g64[k]) / rel(g32[k], g64[k])))
