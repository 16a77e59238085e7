"""profiles/<tag>_pmc_per_kernel.txt -> profiles/pmc_traffic.json (HBM bytes per launch of the main kernels,
read by bench.py for roofline.traffic).
usage: python profiles/make_pmc_traffic.py profiles/archive/r02_cfg2_x_pmc_per_kernel.txt [profiles/archive/r02_cfg4_x_pmc_per_kernel.txt
                                            [profiles/archive/r03_cfg5_x_pmc_per_kernel.txt]]
The second file is the bf16 N=200 workload (bench.py --config cfg4); its kernels are stored under 'cfg4:<tag>'.  The third is
the ragged fp32 workload (--config cfg5, 8 pairs, the batch of bench.py's seed); its kernels are stored under 'cfg5:<tag>'."""
import json, os, re, sys

TAGS = {
    'mlp_bwd_kernel<32, 0, 3>': 'mlp_bwd[cin=32,dx=32]', 'mlp_bwd_kernel<32, 32, 3>': 'mlp_bwd[cin=64,dx=64]',
    'mlp_bwd_kernel<2, 0, 3>': 'mlp_bwd[cin=2,dx=0]', 'mlp_bwd_kernel<32, 2, 3>': 'mlp_bwd[cin=34,dx=32]',
    'mlp_fwd_kernel<32, 0, 2, 3>': 'mlp_fwd[cin=32,nmlp=2]', 'mlp_fwd_kernel<32, 32, 1, 3>': 'mlp_fwd[cin=64,nmlp=1]',
    'mlp_fwd_kernel<2, 0, 2, 3>': 'mlp_fwd[cin=2,nmlp=2]', 'mlp_fwd_kernel<32, 2, 1, 3>': 'mlp_fwd[cin=34,nmlp=1]',
    'mlp_bwd_pair_kernel<32>': 'mlp_bwd_pair[cin=32,dx=32]', 'mlp_bwd_pair_kernel<2>': 'mlp_bwd_pair[cin=2,dx=0]',
    'mlp_bwd_pair_x3_kernel<32>': 'x3:mlp_bwd_pair[cin=32,dx=32]', 'mlp_bwd_pair_x3_kernel<2>': 'x3:mlp_bwd_pair[cin=2,dx=0]',
    'sb_fwd_kernel': 'fgnn_block1_struct_fwd', 'sb_bwd_reduce_kernel': 'fgnn_block1_struct_bwd',
    'sb_fwd_kernel<1>': 'fgnn_block1_struct_fwd', 'sb_bwd_reduce_kernel<1>': 'fgnn_block1_struct_bwd',
    'chan_matmul_bwd1_kernel': 'fgnn_chan_matmul_bwd', 'chan_matmul_fwd1_kernel': 'fgnn_chan_matmul_fwd', 'chan_matmul_fwd_w_kernel<7, true>': 'fgnn_chan_matmul_fwd',
}
# round 6: the 16-pixel-tile kernels (full template lists as rocprofv3 prints them)
TAGS.update({'mlp_bwd_pair_t16_kernel<false, true>': 'mlp_bwd_pair[cin=32,dx=32]', 'mlp_bwd_t16_kernel<32, false, false, true>': 'mlp_bwd[cin=64,dx=64]',
             'mlp_bwd_t16_kernel<2, true, false, false>': 'mlp_bwd[cin=34,dx=32]', 'mlp_bwd_t16_kernel<2, false, false, false>': 'mlp_bwd[cin=34,dx=32]'})
TAGS16 = {
    'mlp_bwd16_pair_kernel<32>': 'mlp_bwd16_pair[cin=32,dx=32]', 'mlp_bwd16_pair_kernel<2>': 'mlp_bwd16_pair[cin=2,dx=0]',
    'mlp_bwd16_kernel<32, 0, 3>': 'mlp_bwd16[cin=32,dx=32]', 'mlp_bwd16_kernel<32, 32, 3>': 'mlp_bwd16[cin=64,dx=64]',
    'mlp_bwd16_kernel<2, 0, 3>': 'mlp_bwd16[cin=2,dx=0]', 'mlp_bwd16_kernel<32, 2, 3>': 'mlp_bwd16[cin=34,dx=32]',
    'mlp_fwd16_kernel<32, 0, 2, 3>': 'mlp_fwd16[cin=32,nmlp=2]', 'mlp_fwd16_kernel<32, 32, 1, 3>': 'mlp_fwd16[cin=64,nmlp=1]',
    'mlp_fwd16_kernel<2, 0, 2, 3>': 'mlp_fwd16[cin=2,nmlp=2]', 'mlp_fwd16_kernel<32, 2, 1, 3>': 'mlp_fwd16[cin=34,nmlp=1]',
    'chan_matmul_bwd16_kernel<8, 7, 2>': 'fgnn_chan_matmul_bwd16', 'chan_matmul_fwd16_kernel<8, 7, true>': 'fgnn_chan_matmul_fwd16',
    'sb_fwd_kernel<4, true>': 'fgnn_block1_struct_fwd16', 'sb_bwd_reduce_kernel<4, true>': 'fgnn_block1_struct_bwd16',
}
# ragged batches run the SKIP = true instantiations (the ', false' of the packed-input flag is stripped below)
TAGS5 = {k.replace('>', ', true>') if k.startswith(('mlp_', )) else k: v for k, v in TAGS.items()}
TAGS5.update({'chan_matmul_bwd_big_kernel<4>': 'fgnn_chan_matmul_bwd', 'chan_matmul_fwd_big_kernel<4>': 'fgnn_chan_matmul_fwd', 'chan_matmul_fwd_big_kernel<4, true>': 'fgnn_chan_matmul_fwd',
              'sb_fwd_kernel<2>': 'fgnn_block1_struct_fwd', 'sb_bwd_reduce_kernel<2>': 'fgnn_block1_struct_bwd'})
TAGS5.update({'mlp_bwd_pair_t16_kernel<true, true>': 'mlp_bwd_pair[cin=32,dx=32]', 'mlp_bwd_t16_kernel<32, false, true, true>': 'mlp_bwd[cin=64,dx=64]',
              'mlp_bwd_t16_kernel<2, true, true, false>': 'mlp_bwd[cin=34,dx=32]'})
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rel = lambda f: os.path.relpath(f, root)
out = {'_note': 'HBM traffic per launch from rocprofv3 PMC passes (separate FETCH_SIZE / WRITE_SIZE runs, '
                'profiles/pmc_passes.sh); bench workloads cfg2 (B=32 N=50 fp32) and, under cfg4:<kernel> / cfg5:<kernel>, cfg4 (B=8 N=200 bf16) / cfg5 (8 ragged pairs, fp32); '
                'bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB: FETCH_SIZE is '
                'doubled as prescribed for gfx950 in MI355X_MICROARCH.md (HBM section); WRITE_SIZE is uncalibrated there '
                'and used as reported.',
       '_source': ' + '.join(rel(f) for f in sys.argv[1:4])}


def collect(src, tags, prefix):
    cur = None
    for line in open(src):
        m = re.match(r'^(\S.*?)\s+\(avg', line)
        if m:
            name = m.group(1).strip()
            cur = tags.get(name) or tags.get(name.replace(', false', ''))   # (PK / SKIP template flags of the dense kernels)
            if cur:
                cur = prefix + cur
                out[cur] = {}
            continue
        if cur and line.strip():
            k, v = line.split()
            if k == 'FETCH_SIZE':
                out[cur]['fetch_kib'] = float(v)
            if k == 'WRITE_SIZE':
                out[cur]['write_kib'] = float(v)


collect(sys.argv[1], TAGS, '')
if len(sys.argv) > 2:
    collect(sys.argv[2], TAGS16, 'cfg4:')
if len(sys.argv) > 3:
    collect(sys.argv[3], TAGS5, 'cfg5:')
for k, v in out.items():
    if isinstance(v, dict) and 'fetch_kib' in v and 'write_kib' in v:
        v['bytes'] = (2 * v['fetch_kib'] + v['write_kib']) * 1024
json.dump(out, open(os.path.join(root, 'profiles', 'pmc_traffic.json'), 'w'), indent=1)
print(json.dumps({k: v.get('bytes') for k, v in out.items() if isinstance(v, dict)}, indent=1))
