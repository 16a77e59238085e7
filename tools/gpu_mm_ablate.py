"""Isolated, event-timed forward per-channel product (wave per matrix, N = 50, 64 graphs x 32 channels) with ablated builds of
csrc/matmul.hip (-DFGNN_MM_NOMFMA: a VALU fma instead of every MFMA; -DFGNN_MM_NOSTORE: stores compiled in but never executed):
    python tools/gpu_mm_ablate.py [lib names in graph_neural_net_amd/_dbg ...]      ('main' = the shipped library)
One fresh process per build; K operand sets cycled so that the working set exceeds the L2 (4 MB per XCD)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] != '--child':
    for name in sys.argv[1:]:
        r = subprocess.run([sys.executable, __file__, '--child', name], capture_output=True, text=True)
        print('%-10s' % name, r.stdout.strip() or r.stderr[-300:], flush=True)
    raise SystemExit(0)
sys.path.insert(0, ROOT)
import torch
from graph_neural_net_amd import _lib
name = sys.argv[2]
if name != 'main':
    _lib.LIB_PATH = os.path.join(ROOT, 'graph_neural_net_amd', '_dbg', 'libfgnn_hip_%s.so' % name)
G, Cc, N = 64, 32, 50
P = N * N
dev = 'cuda:0'
K = 3
sets = []
for _ in range(K):
    a, b = torch.randn(G, Cc, N, N, device=dev), torch.randn(G, Cc, N, N, device=dev)
    na, nb = torch.rand(G, Cc, 4, device=dev) + 0.5, torch.rand(G, Cc, 4, device=dev) + 0.5
    sets.append((_lib.make_slab(a, Cc * P, P, Cc, nrm=na), _lib.make_slab(b, Cc * P, P, Cc, nrm=nb), torch.empty_like(a), a, b, na, nb))
it = [0]


def fwd():
    s = sets[it[0] % K]
    it[0] += 1
    _lib.call('fgnn_chan_matmul_fwd', C.byref(s[0]), C.byref(s[1]), None, G, N, _lib.ptr(s[2]), Cc * P, P, _lib.stream_ptr())


for _ in range(10):
    fwd()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(30):
        fwd()
g.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    g.replay()
e1.record()
torch.cuda.synchronize()
print('%.2f us per launch (30 back-to-back launches in a replayed graph)' % (e0.elapsed_time(e1) / 300 * 1e3))
