"""Debug: per-phase cycle breakdown of the x3 pair backward (mlp_bwd_pair_x3.hip built with -DFGNN_PHASES into
graph_neural_net_amd/_dbg/libfgnn_hip_PH.so).  usage: python tools/gpu_phases_px3.py [B]"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graph_neural_net_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'graph_neural_net_amd', '_dbg', 'libfgnn_hip_PH.so')
from graph_neural_net_amd import synthetic
from graph_neural_net_amd.engine import FgnnEngine, ParamLayout

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = 50
lay = ParamLayout(2, 4, 32, 32, 3)
dev = torch.device('cuda:0')
params = lay.init_flat(0, dev); grads = torch.zeros_like(params)
x1, x2 = synthetic.make_batch(1, B, N, 'ErdosRenyi', 0.3, 0.1)
x = torch.cat([x1, x2]).contiguous().to(dev)
eng = FgnnEngine(lay, 2 * B, N, dev, mfma='x3')
lib = _lib.load()
buf = torch.zeros(256 * 8 * 16, dtype=torch.int64, device=dev)
lib.fgnn_debug_phase_buffer_px3.argtypes = [C.c_void_p]
for _ in range(3): eng.step(params, grads, x)
torch.cuda.synchronize()
assert lib.fgnn_debug_phase_buffer_px3(buf.data_ptr()) == 0
eng.step(params, grads, x)          # the LAST 32-channel pair launch of the step (block 2) leaves its stamps
torch.cuda.synchronize()
ph = buf.view(256, 8, 16).double().cpu()
tiles = 2 * B * ((N * N + 31) // 32)
names = ['records / consumed-wait / DMA', 'x wait+norm+split+L0 issue', 'h1, h2 (L0, L1 products)', 'dz (dy, z wait)', 'layer 2', 'layer 1',
         'split D0 + prefetch', 'dx + hand-over + store + emit', 'layer-0 wgrad']
for role in (0, 1):
    p = ph[:, 4 * role:4 * role + 4].reshape(-1, 16)
    loop = p[:, :9].sum(1)
    print('role %d (mlp%d): prologue %.0f, loop %.0f, end-barrier wait %.0f (+%.0f before), reduction %.0f ticks per wave' % (
        role, role + 1, p[:, 9].mean().item(), loop.mean().item(), p[:, 10].mean().item(), p[:, 12].mean().item(), p[:, 11].mean().item()))
    tot = p[:, :9].sum().item()
    print('   %.0f ticks per tile per wave' % (tot / tiles))
    for k, n in enumerate(names):
        print('   %-34s %8.0f  %5.1f%%' % (n, p[:, k].sum().item() / tiles, 100 * p[:, k].sum().item() / tot))
