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


def _load(path, allow_pickle):
    """torch.load restricted to tensors / plain containers; Lightning checkpoints that carry pickled hyper-parameter
    objects need allow_pickle=True (arbitrary code execution: only for files you trust)."""
    import pickle
    import warnings
    if hasattr(path, 'read') and not (hasattr(path, 'seekable') and path.seekable()):
        raise RuntimeError('load_checkpoint needs a path or a seekable stream')
    try:
        return torch.load(path, map_location='cpu', weights_only=True)
    except pickle.UnpicklingError as exc:       # what weights_only raises for a disallowed global; I/O errors propagate as they are
        if not allow_pickle:
            raise RuntimeError('checkpoint %r cannot be read with weights_only=True (%s); pass allow_pickle=True for a '
                               'trusted file' % (path, exc))
        warnings.warn('load_checkpoint: %r holds pickled objects; re-reading it with the UNSAFE loader (allow_pickle=True)' % (path,))
        if hasattr(path, 'seek'):
            path.seek(0)
        return torch.load(path, map_location='cpu', weights_only=False)


def load_checkpoint(path_or_obj, device, layout=None, allow_pickle=False):
    """-> (layout, flat fp32 parameter buffer on `device`) from a Lightning .ckpt / a plain state_dict file
    or an already loaded object.  Every tensor of the layout must be present with the reference's shape."""
    obj = _load(path_or_obj, allow_pickle) if isinstance(path_or_obj, (str, bytes)) or hasattr(path_or_obj, 'read') else path_or_obj
    sd = _state_dict_of(obj)
    layout = layout or infer_layout(sd)
    for name, _, shape in layout.entries:
        key = name if name in sd else 'node_embedder.' + name
        if key not in sd:
            raise RuntimeError('checkpoint is missing %s' % key)
        if tuple(sd[key].shape) != tuple(shape):
            raise RuntimeError('checkpoint tensor %s has shape %s, expected %s' % (key, tuple(sd[key].shape), tuple(shape)))
    return layout, layout.flatten(sd, device)


def save_checkpoint(path, layout, flat, extra=None, epoch=0, global_step=0, optimizer=None):
    """Write the flat buffer in the layout of a Lightning checkpoint: {'state_dict': {node_embedder.<name>: tensor},
    'epoch', 'global_step', 'pytorch-lightning_version'} (the keys Lightning's load_from_checkpoint migration looks at;
    models/__init__.py:19-24), plus -- optionally -- the fused Adam state so that a resumed run continues the optimizer
    ('fgnn_adam': exp_avg, exp_avg_sq, step, lr).  Only tensors and plain scalars are written, so the file loads with
    weights_only=True."""
    sd = {'node_embedder.' + k: v.detach().cpu().clone() for k, v in layout.unflatten(flat).items()}
    obj = {'state_dict': sd, 'epoch': int(epoch), 'global_step': int(global_step), 'pytorch-lightning_version': '1.9.0'}
    if optimizer is not None:
        obj['fgnn_adam'] = {'exp_avg': optimizer.exp_avg.detach().cpu().clone(), 'exp_avg_sq': optimizer.exp_avg_sq.detach().cpu().clone(),
                            'step': int(optimizer.t), 'lr': float(optimizer.lr)}
    if extra:
        obj.update(extra)
    torch.save(obj, path)
    return obj


def restore_optimizer(obj, optimizer):
    """Continue the fused Adam from a checkpoint written with save_checkpoint(..., optimizer=...)."""
    st = obj.get('fgnn_adam') if isinstance(obj, dict) else None
    if st is None:
        return False
    optimizer.exp_avg.copy_(st['exp_avg'])
    optimizer.exp_avg_sq.copy_(st['exp_avg_sq'])
    optimizer.t = int(st['step'])
    optimizer.lr = float(st['lr'])
    if getattr(optimizer, '_state', None) is not None:
        optimizer._state[0:1].fill_(optimizer.t)
    return True
