"""Where does the bf16 channel matmul differ from the same-point oracle?  python tests/diag/gpu_mm16_check.py [N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from graph_neural_net_amd import synthetic
from test_gpu_bf16 import _sd, _run, OB
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
sd = _sd(1, 10 + N)
x1, x2 = synthetic.make_batch(N, 1, N, 'ErdosRenyi', 0.5, 0.1)
keep = {}
OB.step_fwd_bwd(x1, x2, sd, keep=keep)
eng, lay, scores, loss, grads = _run(sd, x1, x2, 1)
got = eng.dense(eng.mult[1]).float().cpu(); ref = keep[(1, 'mult')].float()
bad = (got - ref).abs() > 2.0 ** -6 * ref.abs() + 1e-30
print('bad', int(bad.sum()), 'of', bad.numel())
idx = bad.nonzero()
print('g', idx[:, 0].unique().tolist(), 'c', idx[:, 1].unique().tolist()[:40])
print('rows', idx[:, 2].unique().tolist()[:80])
print('cols', idx[:, 3].unique().tolist()[:80])
g, c = idx[0, 0].item(), idx[0, 1].item()
m = bad[g, c]
print('pattern rows x col-blocks (count of bad per 8 columns), channel', c)
for r in range(0, N):
    if m[r].any():
        print(r, ''.join('%d' % min(9, int(m[r, k:k + 8].sum())) for k in range(0, N, 8)), got[g, c, r, m[r]][:3].tolist(), ref[g, c, r, m[r]][:3].tolist())
