import os, sys, torch
sys.path.insert(0, 'tests')
from util import load_golden, rel, sub
from graph_neural_net_amd.siamese import Siamese_Node_Exp
DEV='cuda:0'
NE64 = dict(type='node_embedding', block_init='block_emb', block_inside='block', num_blocks=4, in_features=64, out_features=64, depth_of_mlp=3)
d = load_golden('wide64_c2_64_64_d3_4blk.npz')
model = Siamese_Node_Exp(2, NE64).to(DEV)
model.load_state_dict({'node_embedder.' + k: v for k, v in sub(d, 'sd/').items()})
out = model.node_embedder({'input': d['x1'].to(DEV)})
for k, v in sub(d, 'inter/').items():
    print(os.environ.get('FGNN_MLP64','1'), k, rel(out[k].detach().cpu()[:1], v))
scores = model({'input': d['x1'].to(DEV)}, {'input': d['x2'].to(DEV)})
print('scores vs fp64', rel(scores.detach().cpu(), d['scores64']), 'reference fp32 vs fp64', rel(d['scores'], d['scores64']), 'vs ref fp32', rel(scores.detach().cpu(), d['scores']))
