"""Checkpoint compatibility.

The reference restores models with ``Siamese_Node_Exp.load_from_checkpoint`` (models/__init__.py:19-24):
a Lightning ``.ckpt`` is a ``torch.save``d dict whose ``'state_dict'`` entry holds the module's
parameters under the keys ``node_embedder.ne_bm_block{K}_mlp{J}.convs.{i}.weight`` ...  These helpers
read and write that format for the flat parameter buffer of the fused engine and for the module mirror,
without Lightning."""
import torch

from .engine import ParamLayout


def _state_dict_of(obj):
    sd = obj['state_dict'] if isinstance(obj, dict) and 'state_dict' in obj else obj
    if not isinstance(sd, dict) or not sd:
        raise RuntimeError('checkpoint holds no state_dict')
    return sd


def infer_layout(state_dict):
    """ParamLayout (model hyper-parameters) from the shapes in a reference state_dict."""
    sd = _state_dict_of(state_dict)
    strip = lambda k: k[len('node_embedder.'):] if k.startswith('node_embedder.') else k
    shapes = {strip(k): tuple(v.shape) for k, v in sd.items()}
    blocks = sorted({int(k.split('block')[1].split('_')[0]) for k in shapes if k.startswith('ne_bm_block')})
    if not blocks or blocks != list(range(1, len(blocks) + 1)):
        raise RuntimeError('not a node_embedding state_dict (no ne_bm_block{K} keys)')
    first = shapes['ne_bm_block1_mlp1.convs.0.weight']
    depth = len([k for k in shapes if k.startswith('ne_bm_block1_mlp1.convs.') and k.endswith('.weight')])
    last = shapes['ne_bm_block%d_mlp3.convs.%d.weight' % (blocks[-1], depth - 1)]
    return ParamLayout(original_features_num=first[1], num_blocks=len(blocks), in_features=first[0],
                       out_features=last[0], depth_of_mlp=depth)


def load_checkpoint(path_or_obj, device, layout=None):
    """-> (layout, flat fp32 parameter buffer on `device`) from a Lightning .ckpt / a plain state_dict file
    or an already loaded object.  Every tensor of the layout must be present with the reference's shape."""
    obj = torch.load(path_or_obj, map_location='cpu', weights_only=False) if isinstance(path_or_obj, (str, bytes)) or hasattr(path_or_obj, 'read') else path_or_obj
    sd = _state_dict_of(obj)
    layout = layout or infer_layout(sd)
    for name, _, shape in layout.entries:
        key = name if name in sd else 'node_embedder.' + name
        if key not in sd:
            raise RuntimeError('checkpoint is missing %s' % key)
        if tuple(sd[key].shape) != tuple(shape):
            raise RuntimeError('checkpoint tensor %s has shape %s, expected %s' % (key, tuple(sd[key].shape), tuple(shape)))
    return layout, layout.flatten(sd, device)


def save_checkpoint(path, layout, flat, extra=None):
    """Write the flat buffer in the reference's format ({'state_dict': {node_embedder.<name>: tensor}})."""
    sd = {'node_embedder.' + k: v.detach().cpu().clone() for k, v in layout.unflatten(flat).items()}
    obj = {'state_dict': sd}
    if extra:
        obj.update(extra)
    torch.save(obj, path)
    return obj
