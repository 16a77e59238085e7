"""bf16 engine, four blocks, against the reference-generated N = 200 fixture: distances to fp64 in units of the reference's own
bf16 run (what BF16_CLASS gates)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
from test_gpu_bf16 import _run
from util import flat_of, is_zero_grad, l2rel, load_golden, sub, unpack_pairs
d = load_golden('cfg4_er_n200_b1_4blk.npz')
sd = sub(load_golden('cfg2_reg_n50_b2_4blk.npz'), 'sd/')
n = int(d['n'])
x1, x2 = unpack_pairs(d['bits1'], n), unpack_pairs(d['bits2'], n)
eng, lay, scores, loss, grads = _run(sd, x1, x2, 4)
keys = [k for k in sub(d, 'grad/') if not is_zero_grad(k)]
g64 = flat_of(sub(d, 'grad64/'), keys)
s64 = d['scores64_as_f32']
print('scores: ours %.4f ref-bf16 %.4f ratio %.3f' % (l2rel(scores, s64), l2rel(d['scores_refbf16'], s64), l2rel(scores, s64) / l2rel(d['scores_refbf16'], s64)))
a, b = l2rel(flat_of(grads, keys), g64), l2rel(flat_of(sub(d, 'grad_refbf16/'), keys), g64)
print('flat gradient: ours %.4f ref-bf16 %.4f ratio %.3f' % (a, b, a / b))
print('loss: ours %.3e ref-bf16 %.3e' % (abs(loss - d['loss64'].item()), abs(d['loss_refbf16'].item() - d['loss64'].item())))
