"""Time the bf16 per-channel matmul kernels (N=200, G=16) with the shipped library and with the phase-ablated debug builds
(make -C graph_neural_net_amd/csrc mm16_ablate): python tools/gpu_mm16_ablate.py"""
import ctypes as C, os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    import torch
    from graph_neural_net_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, sys.argv[1])
    G, N = 16, 200
    ldr = 200; ldp = 40000
    dev = torch.device('cuda:0')
    mk = lambda: (torch.randn(G * 32 * ldp, device=dev) * 0.1).to(torch.bfloat16)
    za, zb, dm, out, da, db = mk(), mk(), mk(), mk(), mk(), mk()
    nrm = torch.rand(G * 32 * 4, device=dev) + 0.5
    beta = torch.zeros(32, device=dev)
    s12a, s12b = torch.empty(G * 64, device=dev), torch.empty(G * 64, device=dev)
    sa = _lib.make_slab16(za, 32 * ldp, ldp, 32, nrm=nrm, beta=beta)
    sb = _lib.make_slab16(zb, 32 * ldp, ldp, 32, nrm=nrm, beta=beta)
    st = _lib.stream_ptr()
    def fwd(): _lib.call('fgnn_chan_matmul_fwd16', C.byref(sa), C.byref(sb), None, G, N, ldr, _lib.ptr(out), 32 * ldp, ldp, st)
    def bwd(): _lib.call('fgnn_chan_matmul_bwd16', C.byref(sa), C.byref(sb), _lib.ptr(dm), 32 * ldp, ldp, None, G, N, ldr, _lib.ptr(da), _lib.ptr(db), 32 * ldp, ldp, _lib.ptr(s12a), _lib.ptr(s12b), st)
    res = []
    for f in (fwd, bwd):
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    print('%-44s fwd %7.1f us   bwd %7.1f us' % (sys.argv[1], res[0], res[1]))
else:
    libs = ['graph_neural_net_amd/libfgnn_hip.so'] + ['graph_neural_net_amd/_dbg/libfgnn_hip_mm%d.so' % k for k in (1, 2, 3, 4)]
    names = ['shipped', 'no stores', 'no MFMA', 'no LDS staging', 'no global loads']
    for lib, nm in zip(libs, names):
        print(nm.ljust(16), end=' ', flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), lib], check=False)
