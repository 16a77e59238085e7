"""Debug: per-phase cycle breakdown of fgnn_mlp_bwd_pair_t16 (needs graph_neural_net_amd/_dbg/libfgnn_hip_ph16.so: tools/build_variant.sh ph16 -DFGNN_PHASES).
usage: python tools/gpu_phases_t16.py [B]"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graph_neural_net_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'graph_neural_net_amd', '_dbg', 'libfgnn_hip_ph16.so')
from graph_neural_net_amd import synthetic
from graph_neural_net_amd.engine import FgnnEngine, ParamLayout
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = 50
lay = ParamLayout(2, 4, 32, 32, 3)
dev = torch.device('cuda:0')
params = lay.init_flat(0, dev); grads = torch.zeros_like(params)
x1, x2 = synthetic.make_batch(1, B, N, 'ErdosRenyi', 0.3, 0.1)
x = torch.cat([x1, x2]).contiguous().to(dev)
eng = FgnnEngine(lay, 2 * B, N, dev)
lib = _lib.load()
buf = torch.zeros(256 * 8 * 16, dtype=torch.int64, device=dev)
lib.fgnn_debug_phase_buffer_t16.argtypes = [C.c_void_p]
for _ in range(3): eng.step(params, grads, x)
torch.cuda.synchronize()
assert lib.fgnn_debug_phase_buffer_t16(buf.data_ptr()) == 0
eng.step(params, grads, x)        # (three pair launches per step: the buffer holds the last one = block 2)
torch.cuda.synchronize()
ph = buf.view(256, 8, 16).double().cpu()
halves = 2 * B * ((N * N + 31) // 32) * 2
names = ['-', 'x + recompute', 'dz (dy,z wait)', 'layer 2', 'layer 1', 'layer 0 wgrad(+dgrad)', 'partner wait', 'handover/dgrad/store/emit', 'record + loop', 'PROLOGUE', 'END BARRIER WAIT', 'WG REDUCTION']
for role in (0, 1):
    r = ph[:, 4 * role:4 * role + 4, :]
    per_wave = r.reshape(-1, 16)
    loop = per_wave[:, :9].sum(1)
    print('role %d (mlp%d): per wave prologue %.0f, loop %.0f, barrier wait %.0f, reduction %.0f cycles; halves per wave %.2f'
          % (role, role + 1, per_wave[:, [9, 12, 13, 14, 15]].sum(1).mean(), loop.mean(), per_wave[:, 10].mean(), per_wave[:, 11].mean(), halves / 1024))
    print('  prologue split: kernel arguments %.0f, image loads issued %.0f, x + record loads issued, records landed %.0f, images in LDS %.0f, barrier + records %.0f' % (per_wave[:, 14].mean(), per_wave[:, 15].mean(), per_wave[:, 12].mean(), per_wave[:, 13].mean(), per_wave[:, 9].mean()))
    tot = per_wave[:, :9].sum().item()
    for k in range(1, 9):
        print('  %-28s %8.0f cycles per half  %5.1f%%' % (names[k], per_wave[:, k].sum().item() / halves, 100 * per_wave[:, k].sum().item() / tot))
    print('  total per half %.0f (MFMA issue floor 128 x 32 = 4096)' % (tot / halves))
