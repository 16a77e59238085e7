"""profiles/<tag>_pmc_per_kernel.txt -> profiles/pmc_traffic.json (HBM bytes per launch of the main kernels,
read by bench.py for roofline.traffic).  usage: python profiles/make_pmc_traffic.py profiles/r01_f_pmc_per_kernel.txt"""
import json, os, re, sys

TAGS = {
    'mlp_bwd_kernel<32, 0, 3>': 'mlp_bwd[cin=32,dx=32]', 'mlp_bwd_kernel<32, 32, 3>': 'mlp_bwd[cin=64,dx=64]',
    'mlp_bwd_kernel<2, 0, 3>': 'mlp_bwd[cin=2,dx=0]', 'mlp_bwd_kernel<32, 2, 3>': 'mlp_bwd[cin=34,dx=32]',
    'mlp_fwd_kernel<32, 0, 2, 3>': 'mlp_fwd[cin=32,nmlp=2]', 'mlp_fwd_kernel<32, 32, 1, 3>': 'mlp_fwd[cin=64,nmlp=1]',
    'mlp_fwd_kernel<2, 0, 2, 3>': 'mlp_fwd[cin=2,nmlp=2]', 'mlp_fwd_kernel<32, 2, 1, 3>': 'mlp_fwd[cin=34,nmlp=1]',
    'chan_matmul_bwd1_kernel': 'fgnn_chan_matmul_bwd', 'chan_matmul_fwd1_kernel': 'fgnn_chan_matmul_fwd',
}
src = sys.argv[1]
out = {'_note': 'HBM traffic per launch from rocprofv3 PMC passes (separate FETCH_SIZE / WRITE_SIZE runs, '
                'profiles/pmc_passes.sh), bench workload B=32 N=50; bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB: FETCH_SIZE is '
                'doubled as prescribed for gfx950 in MI355X_MICROARCH.md (HBM section); WRITE_SIZE is uncalibrated there '
                'and used as reported.',
       '_source': os.path.relpath(src, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))}
cur = None
for line in open(src):
    m = re.match(r'^(\S.*?)\s+\(avg', line)
    if m:
        cur = TAGS.get(m.group(1).strip())
        if cur:
            out[cur] = {}
        continue
    if cur and line.strip():
        k, v = line.split()
        if k == 'FETCH_SIZE':
            out[cur]['fetch_kib'] = float(v)
        if k == 'WRITE_SIZE':
            out[cur]['write_kib'] = float(v)
for k, v in out.items():
    if isinstance(v, dict) and 'fetch_kib' in v and 'write_kib' in v:
        v['bytes'] = (2 * v['fetch_kib'] + v['write_kib']) * 1024
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'pmc_traffic.json'), 'w'), indent=1)
print(json.dumps({k: v.get('bytes') for k, v in out.items() if isinstance(v, dict)}, indent=1))
