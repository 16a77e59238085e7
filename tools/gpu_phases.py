"""Debug: per-phase cycle breakdown of one mlp_bwd variant (needs `make -C graph_neural_net_amd/csrc phases`).
usage: python tools/gpu_phases.py [B] [ca] [cb]"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graph_neural_net_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'graph_neural_net_amd', '_dbg', 'libfgnn_hip_phases.so')
from graph_neural_net_amd import synthetic
from graph_neural_net_amd.engine import FgnnEngine, ParamLayout

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ca = int(sys.argv[2]) if len(sys.argv) > 2 else 32
cb = int(sys.argv[3]) if len(sys.argv) > 3 else 0
N = 50
lay = ParamLayout(2, 4, 32, 32, 3)
dev = torch.device('cuda:0')
params = lay.init_flat(0, dev); grads = torch.zeros_like(params)
x1, x2 = synthetic.make_batch(1, B, N, 'ErdosRenyi', 0.3, 0.1)
x = torch.cat([x1, x2]).contiguous().to(dev)
eng = FgnnEngine(lay, 2 * B, N, dev)
lib = _lib.load()
buf = torch.zeros(256 * 8 * 16, dtype=torch.int64, device=dev)
lib.fgnn_debug_phase_buffer.argtypes = [C.c_void_p, C.c_int, C.c_int]
for _ in range(3): eng.step(params, grads, x)
torch.cuda.synchronize()
assert lib.fgnn_debug_phase_buffer(buf.data_ptr(), ca, cb) == 0
eng.step(params, grads, x)
torch.cuda.synchronize()
ph = buf.view(256 * 8, 16).double().cpu()
tiles = 2 * B * ((N * N + 31) // 32)
names = ['x+norm', 'L0 fwd issue', 'h1,h2 staged', 'dz (dy,z wait)', 'bwd L2', 'bwd L1', 'L0 stage+prefetch', 'L0 dgrad+wgrad', 'dx store+emit', 'PROLOGUE', 'END BARRIER WAIT', 'WG REDUCTION']
tot = ph.sum().item()
print('per wave: prologue %.0f, barrier wait %.0f, reduction %.0f cycles; loop %.0f' % (ph[:, 9].mean().item(), ph[:, 10].mean().item(), ph[:, 11].mean().item(), ph[:, :9].sum(1).mean().item()))
print('prologue split: args+first loads issued %.0f, image in LDS %.0f, records+barrier %.0f' % (ph[:, 12].mean().item(), ph[:, 13].mean().item(), ph[:, 9].mean().item()))
print('B=%d variant <%d,%d,3>: %d tiles, %.0f cycles per tile per wave (s_memtime ticks)' % (B, ca, cb, tiles, tot / tiles))
for k, n in enumerate(names):
    print('  %-20s %8.0f  %5.1f%%' % (n, ph[:, k].sum().item() / tiles, 100 * ph[:, k].sum().item() / tot))
