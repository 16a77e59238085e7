"""s_memtime stamps of the structured block-1 kernels (alt build with -DSB_STAMPS: bash /tmp/mkvariant2.sh ST block1_struct.hip -DSB_STAMPS):
    python tools/gpu_struct_stamps.py [cfg2|cfg4|cfg5]
Prints, per kernel, the mean ticks between consecutive stamps of a workgroup and the span first start -> last end (100 MHz ticks)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from graph_neural_net_amd import _lib, synthetic
_lib.LIB_PATH = os.path.join(ROOT, 'graph_neural_net_amd', '_dbg', 'libfgnn_hip_ST.so')
from graph_neural_net_amd.engine import FgnnEngine, ParamLayout
cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg4'
dev = torch.device('cuda:0')
lay = ParamLayout(2, 4, 32, 32, 3)
params = lay.init_flat(0, dev)
nv = None
if cfg == 'cfg4':
    from graph_neural_net_amd.engine16 import FgnnEngineBF16
    B, N = 8, 200
    x1, x2 = synthetic.make_batch(4000, B, N, 'ErdosRenyi', 0.5, 0.1)
    eng = FgnnEngineBF16(lay, 2 * B, N, dev, block1='structured')
elif cfg == 'cfg5':
    xs, ys = synthetic.make_ragged_batch(5000, 8, 30, 120, 'ErdosRenyi', 0.2, 0.1)
    sizes = [int(t.shape[-1]) for t in xs]
    N, B = max(sizes), 8
    pad = lambda lst: torch.stack([torch.nn.functional.pad(t, (0, N - t.shape[-1], 0, N - t.shape[-1])) for t in lst])
    x1, x2 = pad(xs), pad(ys)
    nv = torch.tensor(sizes * 2, dtype=torch.int32, device=dev)
    eng = FgnnEngine(lay, 2 * B, N, dev, ragged=True, block1='structured')
else:
    B, N = 32, 50
    x1, x2 = synthetic.make_batch(2000, B, N, 'Regular', 0.2, 0.1)
    eng = FgnnEngine(lay, 2 * B, N, dev, block1='structured')
bits = torch.from_numpy(synthetic.pack_adjacency(torch.cat([x1, x2])[:, 0].numpy()).view(np.int32)).to(dev)
g = torch.zeros_like(params)
for _ in range(3):
    eng.step(params, g, None, nvalid=nv, bits=bits)
torch.cuda.synchronize()
buf = torch.zeros(4 * 2048 * 8, dtype=torch.int64, device=dev)
lib = _lib.load()
lib.fgnn_debug_sb_stamps.argtypes = [C.c_void_p]
assert lib.fgnn_debug_sb_stamps(buf.data_ptr()) == 0
eng.step(params, g, None, nvalid=nv, bits=bits)
torch.cuda.synchronize()
st = buf.view(4, 2048, 8).cpu().double()
names = {0: ('sb_graph', ['bit rows (global) + barrier', 'columns + barrier', 'vertex records / y == 0', 'band: code (+ x16)']),
         1: ('sb_fwd', ['class values -> LDS', 'barrier + column terms', 'rows']),
         2: ('sb_bwd_reduce', ['rows (loads + sums)', 'barrier + vertex stage (+ barrier)', 'last lane'])}
for k, (name, ph) in names.items():
    s = st[k]
    live = s[:, 0] > 0
    s = s[live]
    n = len(ph)
    d = s[:, 1:n + 1] - s[:, 0:n]
    print('%s: %d workgroups, mean ticks per workgroup %.0f, span first start -> last end %.0f ticks (10 ns each)' % (
        name, int(live.sum()), (s[:, n] - s[:, 0]).mean().item(), (s[:, n].max() - s[:, 0].min()).item()))
    for nm, v, mx in zip(ph, d.mean(0).tolist(), d.max(0)[0].tolist()):
        print('    %-36s mean %7.0f   max %7.0f' % (nm, v, mx))
    print('    start spread: last workgroup starts %.0f ticks after the first' % (s[:, 0].max() - s[:, 0].min()).item())
