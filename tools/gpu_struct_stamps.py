"""s_memtime stamps of sb_bwd_params_kernel (alt build with -DSB_STAMPS): python tools/gpu_struct_stamps.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from graph_neural_net_amd import _lib, synthetic
_lib.LIB_PATH = os.path.join(ROOT, 'graph_neural_net_amd', '_dbg', 'libfgnn_hip_ST.so')
from graph_neural_net_amd.engine import FgnnEngine, ParamLayout
B, N = 32, 50
dev = torch.device('cuda:0')
lay = ParamLayout(2, 4, 32, 32, 3)
params = lay.init_flat(0, dev)
x1, x2 = synthetic.make_batch(2000, B, N, 'Regular', 0.2, 0.1)
bits = torch.from_numpy(synthetic.pack_adjacency(torch.cat([x1, x2])[:, 0].numpy()).view(np.int32)).to(dev)
eng = FgnnEngine(lay, 2 * B, N, dev, block1='structured')
g = torch.zeros_like(params)
for _ in range(3): eng.step(params, g, None, bits=bits)
torch.cuda.synchronize()
buf = torch.zeros(128 * 16, dtype=torch.int64, device=dev)
lib = _lib.load()
lib.fgnn_debug_sb_stamps.argtypes = [C.c_void_p]
assert lib.fgnn_debug_sb_stamps(buf.data_ptr()) == 0
eng.step(params, g, None, bits=bits)
torch.cuda.synchronize()
st = buf.view(128, 16).cpu().double()
d = st[:, 1:12] - st[:, 0:11]
names = ['loads issued+landed in LDS', 'barrier', 'vertex classes + barrier', 'table loads -> LDS', 'barrier', 'S1/S2 + coef (+barrier)', 'dz (+barrier)', 'phase B', 'barrier + phase C (+barrier)', 'gradient sums', 'stores']
print('ticks per phase, mean over 128 workgroups; total %.0f' % (st[:, 11] - st[:, 0]).mean().item())
for n, v in zip(names, d.mean(0).tolist()):
    print('  %-32s %8.1f' % (n, v))
print('kernel span: first start -> last end %.0f ticks' % (st[:, 11].max() - st[:, 0].min()).item())
