"""Debug: per-phase cycle breakdown of fgnn_mlp_bwd_pair_t16 on the cfg5 batch (8 ragged pairs, n in [30, 120]; the SKIP instantiation):
mean and worst wave.  Needs graph_neural_net_amd/_dbg/libfgnn_hip_ph16.so (tools/build_variant.sh ph16 mlp_bwd_pair_t16.hip -DFGNN_PHASES).
usage: python tools/gpu_phases_t16_ragged.py"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graph_neural_net_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'graph_neural_net_amd', '_dbg', 'libfgnn_hip_ph16.so')
from graph_neural_net_amd import synthetic
from graph_neural_net_amd.engine import FgnnEngine, ParamLayout
lay = ParamLayout(2, 4, 32, 32, 3)
dev = torch.device('cuda:0')
params = lay.init_flat(0, dev); grads = torch.zeros_like(params)
xs, ys = synthetic.make_ragged_batch(5000, 8, 30, 120, 'ErdosRenyi', 0.2, 0.1)
sizes = [int(t.shape[-1]) for t in xs]
N = max(sizes)
pad = lambda lst: torch.stack([torch.nn.functional.pad(t, (0, N - t.shape[-1], 0, N - t.shape[-1])) for t in lst])
x = torch.cat([pad(xs), pad(ys)]).contiguous().to(dev)
nvalid = torch.tensor(sizes * 2, dtype=torch.int32, device=dev)
eng = FgnnEngine(lay, 16, N, dev, ragged=True, block1='generic')
lib = _lib.load()
buf = torch.zeros(256 * 8 * 16, dtype=torch.int64, device=dev)
lib.fgnn_debug_phase_buffer_t16.argtypes = [C.c_void_p]
tn = float(sum(sizes) * 2)
for _ in range(3): eng.step(params, grads, x, nvalid=nvalid, total_nodes=tn)
torch.cuda.synchronize()
assert lib.fgnn_debug_phase_buffer_t16(buf.data_ptr()) == 0
eng.step(params, grads, x, nvalid=nvalid, total_nodes=tn)
torch.cuda.synchronize()
ph = buf.view(256, 8, 16).double().cpu()
print('sizes', sizes, 'sum n^2 x 2 =', 2 * sum(n * n for n in sizes), ' 16-pixel halves of valid pixels:', 2 * sum(n * n for n in sizes) / 16)
for role in (0, 1):
    pw = ph[:, 4 * role:4 * role + 4, :].reshape(-1, 16)
    loop = pw[:, :9].sum(1)
    pro = pw[:, [9, 12, 13, 14, 15]].sum(1)
    print('role %d: prologue mean %.0f max %.0f | loop mean %.0f max %.0f min %.0f | barrier wait mean %.0f | reduction %.0f  (cycles)'
          % (role, pro.mean(), pro.max(), loop.mean(), loop.max(), loop.min(), pw[:, 10].mean(), pw[:, 11].mean()))
    names = ['-', 'x + recompute', 'dz (dy,z wait)', 'layer 2', 'layer 1', 'layer 0 wgrad(+dgrad)', 'partner wait', 'handover/dgrad/store/emit', 'record + loop']
    tot = pw[:, :9].sum().item()
    for k in range(1, 9):
        print('  %-28s %5.1f%%  (%.0f cycles per wave)' % (names[k], 100 * pw[:, k].sum().item() / tot, pw[:, k].mean()))
