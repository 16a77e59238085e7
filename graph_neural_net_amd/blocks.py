"""FGNN block / model graph builders: the dict-of-nodes surface of the reference
(models/blocks_emb.py:9-43).  A node is either a callable (input = previous node) or a
``(callable, [input names])`` tuple; nested dicts are sub-graphs.  Keys and wiring are the
reference's: ``in, mlp1, mlp2, mult, cat, mlp3`` per block (mult first in the concat),
``block1..blockK`` in ``base_model``, ``in, bm, suffix`` in ``node_embedding``.
"""
from .layers import ColumnMaxPooling, Concat, Identity, Matmul, MlpBlock_Real


def block_emb(in_features, out_features, depth_of_mlp, constant_n_vertices=True):
    return {'in': Identity(),
            'mlp3': MlpBlock_Real(in_features, out_features, depth_of_mlp, constant_n_vertices=constant_n_vertices)}


def block(in_features, out_features, depth_of_mlp, constant_n_vertices=True):
    mk = lambda cin: MlpBlock_Real(cin, out_features, depth_of_mlp, constant_n_vertices=constant_n_vertices)
    nodes = {'in': Identity()}
    nodes['mlp1'] = (mk(in_features), ['in'])
    nodes['mlp2'] = (mk(in_features), ['in'])
    nodes['mult'] = (Matmul(), ['mlp1', 'mlp2'])
    nodes['cat'] = (Concat(), ['mult', 'in'])
    nodes['mlp3'] = mk(in_features + out_features)
    return nodes


def base_model(original_features_num, num_blocks, in_features, out_features, depth_of_mlp, block=block,
               constant_n_vertices=True):
    nodes = {'in': Identity()}
    width = original_features_num
    for k in range(1, num_blocks + 1):
        out = in_features if k < num_blocks else out_features
        nodes['block%d' % k] = block(width, out, depth_of_mlp, constant_n_vertices=constant_n_vertices)
        width = out
    return nodes


def node_embedding(original_features_num, num_blocks, in_features, out_features, depth_of_mlp, block=block,
                   constant_n_vertices=True, **kwargs):
    return {'in': Identity(),
            'bm': base_model(original_features_num, num_blocks, in_features, out_features, depth_of_mlp, block,
                             constant_n_vertices=constant_n_vertices),
            'suffix': ColumnMaxPooling()}
