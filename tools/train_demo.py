"""End-to-end sanity run: train the default model on fresh synthetic regular-graph pairs with the fused HIP
step + fused Adam and print the loss / arg-max and Hungarian accuracy trend.  usage: python tools/train_demo.py [steps] [B] [N]
(FGNN_CAPTURE=0: eager launches; FGNN_PRECISION=bf16: the 16-bit kernel set)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graph_neural_net_amd import synthetic
from graph_neural_net_amd.engine import ParamLayout
from graph_neural_net_amd.metrics import accuracy_linear_assignment, accuracy_max
from graph_neural_net_amd.trainer import FgnnTrainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
N = int(sys.argv[3]) if len(sys.argv) > 3 else 50
dev = torch.device('cuda:0')
lay = ParamLayout(2, 4, 32, 32, 3)
capture = os.environ.get('FGNN_CAPTURE', '1') != '0'
precision = os.environ.get('FGNN_PRECISION', 'fp32')
tr = FgnnTrainer(lay, lay.init_flat(0, dev), lr=1e-3, capture=capture, precision=precision)
pool = [synthetic.make_batch(100 + i, B, N, 'Regular', 0.2, 0.05) for i in range(16)]    # host generation is slow
pool = [(a.to(dev), b.to(dev)) for a, b in pool]
t0 = time.time()
for s in range(steps):
    x1, x2 = pool[s % len(pool)]
    loss, scores = tr.train_step(x1, x2)
    if s % 25 == 0 or s == steps - 1:
        acc, n = accuracy_max(scores)
        hun, _ = accuracy_linear_assignment(scores)          # the reference's per-step metric (toolbox/metrics.py:92-116), on the device
        print('step %4d  loss %.4f  acc_max %.3f  acc_linear_assignment %.3f' % (s, loss.item(), acc / n, hun / n), flush=True)
torch.cuda.synchronize()
print('%.1f steps/s (%s, %s, incl. Adam)' % (steps / (time.time() - t0), precision, 'HIP graph replay' if capture else 'eager launches'))
