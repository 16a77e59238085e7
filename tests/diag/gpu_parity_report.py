"""Measured engine-vs-golden errors next to the gates of tests/test_gpu_parity.py (run on the GPU box; prints the table committed as
profiles/r06_parity.txt).  usage: python tests/diag/gpu_parity_report.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from graph_neural_net_amd.engine import FgnnEngine, ParamLayout
from util import is_zero_grad, load_golden, rel, sub, unpack_pairs
import test_gpu_parity as T

print('engine kernels: FGNN_T16 =', FgnnEngine.T16, ' (forward v_mfma_f32_32x32x2_f32; backward 16-pixel tiles on v_mfma_f32_16x16x4_f32)')
print('gates: per-op / one block OP_TOL = %.0e, scores and block-4 activations after four blocks E2E_FWD_TOL = %.0e (north_star: 1e-5;' % (T.OP_TOL, T.E2E_FWD_TOL))
print('       the reference\'s own fp32-vs-fp64 distance there is 1.4e-5 ... 1.6e-5, SURVEY section 0 row 5)')
rows = []
d = load_golden('cfg1_er_n20_b4_1blk.npz')
eng, params, lay, scores, loss, grads = T._run_engine(sub(d, 'sd/'), d['x1'], d['x2'], 1)
B = d['x1'].shape[0]
for j in (1, 2, 3):
    rows.append(('cfg1 (N=20, B=4, 1 block) block1/mlp%d normalised' % j, rel(eng.normalized(1, j, params).cpu()[:B], d['inter/ne/bm/block1/mlp%d' % j]), T.OP_TOL))
rows.append(('cfg1 block1/mult', rel(eng.unpadded(eng.mult[1]).cpu()[:B], d['inter/ne/bm/block1/mult']), T.OP_TOL))
rows.append(('cfg1 pooled embedding', rel(eng.E.cpu()[:B], d['inter/ne/suffix']), T.OP_TOL))
rows.append(('cfg1 scores vs fp32 golden', rel(scores, d['scores']), T.OP_TOL))
rows.append(('cfg1 worst gradient tensor vs fp32 golden', max(rel(grads[k], v) for k, v in sub(d, 'grad/').items() if not is_zero_grad(k)), 2e-5))
d = load_golden('cfg2_reg_n50_b2_4blk.npz')
eng, params, lay, scores, loss, grads = T._run_engine(sub(d, 'sd/'), d['x1'], d['x2'], 4)
rows.append(('cfg2 (N=50, B=2, 4 blocks) block1/mlp3 normalised', rel(eng.normalized(1, 3, params).cpu()[:1], d['inter/ne/bm/block1/mlp3']), T.OP_TOL))
rows.append(('cfg2 block4/mlp3 normalised', rel(eng.normalized(4, 3, params).cpu()[:1], d['inter/ne/bm/block4/mlp3']), T.E2E_FWD_TOL))
rows.append(('cfg2 pooled embedding', rel(eng.E.cpu()[:1], d['inter/ne/suffix']), T.E2E_FWD_TOL))
rows.append(('cfg2 scores vs fp32 golden', rel(scores, d['scores']), T.E2E_FWD_TOL))
rows.append(('cfg2 scores vs fp64 golden', rel(scores, d['scores64']), T.E2E_FWD_TOL))
rows.append(('   (the reference\'s own fp32 scores vs its fp64 scores)', rel(d['scores'], d['scores64']), float('nan')))
rows.append(('cfg2 loss, relative', abs(loss - d['loss'].item()) / abs(d['loss'].item()), 1e-5))
d = load_golden('cfg2_reg_n50_b32_4blk.npz')
sd = sub(load_golden('cfg2_reg_n50_b2_4blk.npz'), 'sd/')
n = int(d['n'])
x1, x2 = unpack_pairs(d['bits1'], n), unpack_pairs(d['bits2'], n)
eng, params, lay, scores, loss, grads = T._run_engine(sd, x1, x2, 4)
rows.append(('cfg2 FULL benchmarked batch (B=32) scores vs fp32 golden', rel(scores, d['scores']), T.E2E_FWD_TOL))
rows.append(('cfg2 FULL batch scores vs fp64 golden (as fp32)', rel(scores, d['scores64_as_f32']), T.E2E_FWD_TOL))
rows.append(('   (the reference\'s own fp32 scores vs fp64, full batch)', rel(d['scores'], d['scores64_as_f32']), float('nan')))
rows.append(('cfg2 FULL batch loss, relative', abs(loss - d['loss'].item()) / abs(d['loss'].item()), 1e-5))
print('%-62s %12s %10s' % ('quantity (max-norm relative error)', 'measured', 'gate'))
for name, v, gate in rows:
    print('%-62s %12.3e %10s' % (name, v, ('%.0e' % gate) if gate == gate else '-'))
