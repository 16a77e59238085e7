"""Debug: absolute per-wave timeline of one mlp_fwd variant (needs `make -C graph_neural_net_amd/csrc phases`).
usage: python tools/gpu_fwd_timeline.py [B] [ca] [cb] [nmlp]"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graph_neural_net_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'graph_neural_net_amd', '_dbg', 'libfgnn_hip_phases.so')
from graph_neural_net_amd import synthetic
from graph_neural_net_amd.engine import FgnnEngine, ParamLayout
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ca, cb, nmlp = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((2, 32), (3, 0), (4, 2)))
N = 50
lay = ParamLayout(2, 4, 32, 32, 3)
dev = torch.device('cuda:0')
params = lay.init_flat(0, dev); grads = torch.zeros_like(params)
x1, x2 = synthetic.make_batch(1, B, N, 'ErdosRenyi', 0.3, 0.1)
x = torch.cat([x1, x2]).contiguous().to(dev)
eng = FgnnEngine(lay, 2 * B, N, dev)
lib = _lib.load()
buf = torch.zeros(256 * 16 * 4, dtype=torch.int64, device=dev)
lib.fgnn_debug_fwd_stamps.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
for _ in range(3): eng.step(params, grads, x)
torch.cuda.synchronize()
assert lib.fgnn_debug_fwd_stamps(buf.data_ptr(), ca, cb, nmlp) == 0
eng.step(params, grads, x)
torch.cuda.synchronize()
st = buf.view(-1, 4).cpu()
st = st[st[:, 0] > 0].double()
t0 = st[:, 0].min()
start, pro, end, nt = st[:, 0] - t0, st[:, 1] - t0, st[:, 2] - t0, st[:, 3]
q = lambda v: ' '.join('%7.0f' % torch.quantile(v, p).item() for p in (0.0, 0.1, 0.5, 0.9, 1.0))
print('mlp_fwd<%d,%d,%d> B=%d: %d waves (min / p10 / median / p90 / max, cycles after the first wave started)' % (ca, cb, nmlp, B, len(st)))
print('  wave start      ', q(start))
print('  prologue done   ', q(pro))
print('  wave end        ', q(end))
for k in sorted(set(nt.tolist())):
    m = nt == k
    print('  waves with %d tiles: %4d, loop cycles median %.0f, end median %.0f' % (k, int(m.sum()), (end - pro)[m].median().item(), end[m].median().item()))
