"""Per-kernel cost inside a HIP graph: capture 20 back-to-back copies of one engine stage, replay, time."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
from graph_neural_net_amd import _lib, synthetic
from graph_neural_net_amd.engine import FgnnEngine, ParamLayout

lay = ParamLayout(2, 4, 32, 32, 3); dev = torch.device('cuda:0')
params = lay.init_flat(0, dev); grads = torch.zeros_like(params)
x1, x2 = synthetic.make_batch(1, 32, 50); x = torch.cat([x1, x2]).contiguous().to(dev)
eng = FgnnEngine(lay, 64, 50, dev)
eng.step(params, grads, x); torch.cuda.synchronize()
W = eng._bwd; gs = 32 * eng.ldp; st = None

def stage_fwd12():
    eng._mlp_fwd(params, 2, (1, 2), eng._slab_in(2, params), None)          # mlp12 + finalize2
def stage_fwd12_only():
    a = eng._slab_in(2, params)
    args = _lib.MlpFwdArgs(); L = lay
    args.G, args.N, args.depth, args.nmlp = eng.G, eng.N, L.depth, 2
    args.a = a
    for m, j in enumerate((1, 2)):
        rec = L.mlp[(2, j)]
        for l in range(L.depth):
            args.W[m][l] = eng._w(params, rec['w'][l]); args.bias[m][l] = eng._w(params, rec['b'][l])
        args.z[m] = eng.z[(2, j)].data_ptr(); args.part[m] = eng.part[m].data_ptr()
    args.ldz = eng.ldp; args.cnt = eng.cnt.data_ptr(); args.packed = eng._packs[('f', 2, 12)][4].data_ptr()
    _lib.call('fgnn_mlp_fwd', C.byref(args), _lib.stream_ptr())
def stage_matmul_fwd():
    ya, yb = eng._slab_z(2, 1, params), eng._slab_z(2, 2, params)
    _lib.call('fgnn_chan_matmul_fwd', C.byref(ya), C.byref(yb), None, eng.G, eng.N, _lib.ptr(eng.mult[2]), gs, eng.ldp, _lib.stream_ptr())
def stage_bwd1():
    eng._mlp_bwd(params, 2, 1, eng._slab_in(2, params), None, W['dy1'], W['coef'][0], W['dy'][1], None, True, False)
def stage_matmul_bwd():
    ya, yb = eng._slab_z(2, 1, params), eng._slab_z(2, 2, params)
    _lib.call('fgnn_chan_matmul_bwd', C.byref(ya), C.byref(yb), _lib.ptr(W['dmult']), gs, eng.ldp, None, eng.G, eng.N,
              _lib.ptr(W['dy1']), _lib.ptr(W['dy2']), gs, eng.ldp, _lib.ptr(W['s12'][(2, 1)]), _lib.ptr(W['s12'][(2, 2)]), _lib.stream_ptr())
def stage_coef2():
    _lib.call('fgnn_gn_bwd_coef2', _lib.ptr(W['s12'][(2, 1)]), _lib.ptr(W['s12'][(2, 2)]), _lib.ptr(eng.nrm[(2, 1)]), _lib.ptr(eng.nrm[(2, 2)]),
              None, eng.G, 32, eng.N, _lib.ptr(W['coef'][0]), _lib.ptr(W['coef'][1]), _lib.stream_ptr())

def timeit(name, fn, copies=20, reps=20):
    fn(); torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(copies): fn()
    g.replay(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): g.replay()
    torch.cuda.synchronize()
    print('%-18s %7.2f us per call (graph replay, %d copies)' % (name, (time.perf_counter() - t) / reps / copies * 1e6, copies))

timeit('mlp_fwd12+fin2', stage_fwd12)
timeit('mlp_fwd12 only', stage_fwd12_only)
timeit('matmul_fwd', stage_matmul_fwd)
timeit('mlp_bwd[32]', stage_bwd1)
timeit('matmul_bwd', stage_matmul_bwd)
timeit('coef2', stage_coef2)
def full(): eng.step(params, grads, x, total_nodes=1600.0)
timeit('full step', full, copies=1)
