"""Phase time stamps of the bf16 forward matmul (debug build 5 of `make mm16_ablate`), N=200, G=16:
python tools/gpu_mm16_stamps.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from graph_neural_net_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'graph_neural_net_amd/_dbg/libfgnn_hip_mm5.so')
G, N = 16, 200
ldr, ldp = 200, 40000
dev = torch.device('cuda:0')
mk = lambda: (torch.randn(G * 32 * ldp, device=dev) * 0.1).to(torch.bfloat16)
za, zb, out = mk(), mk(), mk()
nrm = torch.rand(G * 32 * 4, device=dev) + 0.5
beta = torch.zeros(32, device=dev)
sa = _lib.make_slab16(za, 32 * ldp, ldp, 32, nrm=nrm, beta=beta)
sb = _lib.make_slab16(zb, 32 * ldp, ldp, 32, nrm=nrm, beta=beta)
st = _lib.stream_ptr()
for _ in range(3):
    _lib.call('fgnn_chan_matmul_fwd16', C.byref(sa), C.byref(sb), None, G, N, ldr, _lib.ptr(out), 32 * ldp, ldp, st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
_lib.call('fgnn_chan_matmul_fwd16', C.byref(sa), C.byref(sb), None, G, N, ldr, _lib.ptr(out), 32 * ldp, ldp, st)
e1.record(); torch.cuda.synchronize()
buf = np.zeros((1024, 16), dtype=np.uint64)
lib = C.CDLL(_lib.LIB_PATH)
assert lib.fgnn_debug_mm16_stamps(buf.ctypes.data_as(C.c_void_p)) == 0
ts = buf[:G * 32].astype(np.int64)
us = e0.elapsed_time(e1) * 1e3
names = ['start', 'loads issued', 'chunk0 staged', 'mfma0', 'staged1', 'mfma1', 'staged2', 'mfma2', 'staged3', 'mfma3', 'sync',
         'image written', 'copy-out issued', 'end']
# the counter is per XCD (workgroup b runs on XCD b % 8): align starts within an XCD, durations are per workgroup
tot = ts[:, 13] - ts[:, 0]
print('kernel %.1f us; workgroup life in ticks: median %d  min %d  max %d' % (us, np.median(tot), tot.min(), tot.max()))
xcd = np.arange(len(ts)) % 8
span = max((ts[xcd == x, 13].max() - ts[xcd == x, 0].min()) for x in range(8))
k = us / span                                   # us per tick, from the longest XCD span ~ kernel duration
print('ticks per us ~ %.1f' % (1 / k))
start = np.concatenate([ts[xcd == x, 0] - ts[xcd == x, 0].min() for x in range(8)])
order = np.concatenate([np.nonzero(xcd == x)[0] for x in range(8)])
st = np.empty(len(ts)); st[order] = start * k
for nm, sel in (('first round', st < us * 0.25), ('second round', st >= us * 0.25)):
    grp = ts[sel]
    print(nm, len(grp), 'workgroups; start at %.1f us (median)' % np.median(st[sel]))
    prev = grp[:, 0]
    for i in range(1, 14):
        if (grp[:, i] == 0).any():
            continue
        d = (grp[:, i] - prev) * k
        print('  %-18s +%5.2f us (median)  [%5.2f .. %5.2f]' % (names[i], np.median(d), d.min(), d.max()))
        prev = grp[:, i]
    print('  total %.2f us' % np.median((grp[:, 13] - grp[:, 0]) * k))
