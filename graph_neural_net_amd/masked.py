"""Ragged batches: the host-side mirror of the reference's ``MaskedTensor``
(maskedtensors/maskedtensor.py).

The reference keeps a zero-padded batch plus one float 0/1 mask per *named* dimension and re-applies the masks
after every op through the ``__torch_function__`` protocol (``:87-112``), with hand-written overrides for the ops
whose masked semantics differ (``SPECIAL_FUNCTIONS`` / ``implements``, ``:189-384``).  The contract is the same
here -- an op that receives a MaskedTensor returns a MaskedTensor with the same masks and EXACT zeros in the
padding, ``list(mt)`` yields the un-padded per-graph tensors -- but a mask is stored as one int32 vertex count per
graph and name (``nvalid``), which is what the HIP kernels consume; the float masks of the reference
(``mask_dict``) are synthesised on demand.

Two levels:
  * the fused modules of ``layers.py`` (MlpBlock_Real, GraphNorm, normalize, Matmul, Concat, ColumnMaxPooling),
    the siamese scoring and ``triplet_loss`` take a MaskedTensor directly and pass its counts to the kernels
    (the fast paths);
  * every other ``torch.*`` function goes through ``__torch_function__`` below: unwrap to a named tensor, run the
    op, keep the masks whose names survive, re-mask.  Overrides restate the reference's special cases (``max``,
    ``conv2d``, ``linear``, ``cat``, ``stack``, ``flatten``, ``mean``, ``var``, ``instance_norm``, ``layer_norm``,
    ``diag_embed``, ``nll_loss``, ``cross_entropy``).

Deviation kept on purpose: ``.tensor`` is a plain (un-named) tensor -- raw device pointers are taken from it --
and ``.named_tensor`` / ``.names`` carry the reference's names ``('B', None, 'N', 'N_')``.
"""
import functools

import torch
import torch.nn.functional as F


def _default_names(ndim, masked_dims, base_name, batch_name='B'):
    names = [None] * ndim
    names[0] = batch_name
    for i, d in enumerate(masked_dims):
        names[d] = base_name + '_' * i
    return tuple(names)


class MaskedTensor:
    """Zero-padded batch ``tensor`` of shape (B, ..., Nmax[, Nmax]) + per-graph sizes ``nvalid`` (B,).

    ``masked_dims`` are the (batched) dimensions that are ragged, e.g. (2, 3) for
    (B, C, N, N) activations, (2,) for (B, C, N) embeddings, (1, 2) for (B, N, N) scores."""

    def __init__(self, tensor, nvalid, masked_dims, base_name='N', names=None, masks=None):
        if tensor.has_names():
            tensor = tensor.rename(None)
        self.tensor = tensor
        nvalid = nvalid.to(device=tensor.device, dtype=torch.int32)
        self.names = tuple(names) if names is not None else _default_names(tensor.dim(), tuple(masked_dims), base_name)
        # name -> int32 (B,) vertex counts; all names of a from_list tensor share one vector
        self.masks = dict(masks) if masks is not None else {self.names[d]: nvalid for d in masked_dims}
        self.nvalid = nvalid
        self.masked_dims = tuple(masked_dims)
        self.base_name = base_name
        self._sizes = None

    @classmethod
    def _from_named(cls, named, masks, apply_mask=True):
        """Result of a generic op: keep the masks whose names survived, optionally re-mask (maskedtensor.py:98-112)."""
        names = named.names
        t = named.rename(None)
        live = {nm: cnt for nm, cnt in masks.items() if nm in names}
        dims = tuple(d for d, nm in enumerate(names) if nm in live)
        if not dims:                                        # nothing ragged is left: a plain tensor
            return t
        first = names[dims[0]]
        out = cls(t, live[first], dims, first.rstrip('_') or first, names=names, masks=live)
        if apply_mask:
            out.mask_()
        return out

    # -- reference-compatible surface ------------------------------------------------
    @property
    def named_tensor(self):
        return self.tensor.refine_names(*self.names)

    def _bool_mask(self, dim):
        """(B, 1, .., size(dim), .., 1) bool mask of the valid entries along `dim`"""
        cnt = self.masks[self.names[dim]].to(self.tensor.device)
        size = self.tensor.size(dim)
        m = torch.arange(size, device=self.tensor.device)[None, :] < cnt[:, None]
        shape = [1] * self.tensor.dim()
        shape[0], shape[dim] = m.shape[0], size
        return m.view(shape)

    def mask_(self):
        """Zero the padding in place (maskedtensor.py:87-90)."""
        for d in self.masked_dims:
            self.tensor = self.tensor * self._bool_mask(d).to(self.tensor.dtype)
        return self

    def mask(self):
        return MaskedTensor(self.tensor.clone(), self.nvalid, self.masked_dims, self.base_name, self.names, self.masks).mask_()

    @property
    def mask_dict(self):
        """Float 0/1 masks keyed like the reference ('N', 'N_', ...), named ('B', name): maskedtensor.py:40-46."""
        out = {}
        for d in self.masked_dims:
            nm = self.names[d]
            cnt = self.masks[nm].to(self.tensor.device)
            ar = torch.arange(self.tensor.size(d), device=self.tensor.device)
            out[nm] = (ar[None, :] < cnt[:, None]).to(self.tensor.dtype).refine_names(self.names[0], nm)
        return out

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        """torch.* functions on MaskedTensors (maskedtensor.py:98-112): registered overrides first, else unwrap to
        named tensors, run the op, union the masks, drop the ones whose name vanished, re-mask the result."""
        if kwargs is None:
            kwargs = {}
        if func in SPECIAL_FUNCTIONS:
            return SPECIAL_FUNCTIONS[func](*args, **kwargs)
        new_args = [a.named_tensor if isinstance(a, MaskedTensor) else a for a in args]
        masks = {}
        for a in args:
            if isinstance(a, MaskedTensor):
                masks.update(a.masks)
        ret = func(*new_args, **kwargs)
        return cls._from_named(ret, masks, apply_mask=True)

    @property
    def shape(self):
        return self.tensor.size()

    @property
    def dtype(self):
        return self.tensor.dtype

    @property
    def device(self):
        return self.tensor.device

    @property
    def is_cuda(self):
        return self.tensor.is_cuda

    @property
    def get_device(self):
        return self.tensor.get_device()

    def size(self, *args):
        if args and isinstance(args[0], str):
            return self.tensor.size(self.names.index(args[0]))
        return self.tensor.size(*args)

    def dim(self):
        return self.tensor.dim()

    def contiguous(self, *args):
        self.tensor = self.tensor.contiguous(*args)
        return self

    def sizes(self):
        """Python list of the per-graph vertex counts (one host sync, cached)."""
        if self._sizes is None:
            self._sizes = [int(v) for v in self.nvalid.tolist()]
        return self._sizes

    def to(self, *args, **kwargs):
        return MaskedTensor(self.tensor.to(*args, **kwargs), self.nvalid, self.masked_dims, self.base_name, self.names,
                            self.masks)

    def permute(self, *dims):
        """permute with the names carried along (named tensors cannot permute; maskedtensor.py:165-176)"""
        if len(dims) != self.tensor.dim():
            raise ValueError
        names = tuple(self.names[d] for d in dims)
        return MaskedTensor._from_named(self.tensor.permute(*dims).refine_names(*names), self.masks, apply_mask=False)

    def view(self, *dims):
        """only reshapes trailing un-named dims (maskedtensor.py:140-151)"""
        names = [None] * len(dims)
        for i, nm in enumerate(self.names):
            if i < len(dims):
                names[i] = nm
        return MaskedTensor._from_named(self.tensor.view(*dims).refine_names(*names), self.masks, apply_mask=False)

    def __len__(self):
        return self.tensor.size(0)

    def __getitem__(self, index):
        item = self.tensor[index]
        if len(set(id(v) for v in self.masks.values())) == 1:
            n = self.sizes()[index]
            for dim in self.masked_dims:
                item = torch.narrow(item, dim - 1, 0, n)
            return item
        for dim in self.masked_dims:                       # names with different counts (e.g. a product of 'N' and 'M')
            item = torch.narrow(item, dim - 1, 0, int(self.masks[self.names[dim]][index]))
        return item

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def __repr__(self):
        return 'MaskedTensor(shape=%s, names=%s, nvalid=%s)' % (tuple(self.tensor.shape), self.names, self.sizes())


def from_list(tensor_list, dims, batch_name='B', base_name='N'):
    """Zero-pad a list of per-graph tensors into one batch (maskedtensor.py:8-48).

    ``dims`` are the ragged dimensions of the *un-batched* tensors, e.g. (1, 2) for
    (C, n, n).  All ragged dims of one graph must have the same length n_i."""
    dims = tuple(dims)
    n_dim = tensor_list[0].dim()
    sizes = []
    for t in tensor_list:
        n = t.size(dims[0])
        for d in dims:
            if t.size(d) != n:
                raise ValueError('from_list: ragged dims of one graph must agree (got %s)' % (tuple(t.shape),))
        sizes.append(n)
    shape = [len(tensor_list)] + [max(t.size(d) for t in tensor_list) for d in range(n_dim)]
    data = torch.zeros(shape, dtype=tensor_list[0].dtype, device=tensor_list[0].device)
    for i, t in enumerate(tensor_list):
        idx = (i,) + tuple(slice(0, t.size(d)) for d in range(n_dim))
        data[idx] = t
    nvalid = torch.tensor(sizes, dtype=torch.int32, device=data.device)
    return MaskedTensor(data, nvalid, tuple(d + 1 for d in dims), base_name)


# --------------------------------------------------------------------------------------------------------------
# torch function overrides (maskedtensor.py:186-384)
# --------------------------------------------------------------------------------------------------------------
SPECIAL_FUNCTIONS = {}


def implements(torch_function):
    """Register a torch function override for MaskedTensor (maskedtensor.py:189-200)."""
    @functools.wraps(torch_function)
    def decorator(func):
        SPECIAL_FUNCTIONS[torch_function] = func
        return func
    return decorator


def _like(mt, tensor, apply_mask):
    """same names / masks as `mt` around a new plain tensor"""
    out = MaskedTensor(tensor, mt.nvalid, mt.masked_dims, mt.base_name, mt.names, mt.masks)
    return out.mask_() if apply_mask else out


def _union_masks(items):
    masks = {}
    for a in items:
        if isinstance(a, MaskedTensor):
            masks.update(a.masks)
    return masks


@implements(torch.max)
def torch_max(masked_tensor, dim=None):
    """max over the whole batch, or along `dim` ignoring the padding (padding filled with the dtype's minimum)."""
    if dim is None:
        return torch.max(masked_tensor.tensor)
    t = masked_tensor.tensor
    lo = torch.finfo(t.dtype).min if t.dtype.is_floating_point else torch.iinfo(t.dtype).min
    for d in masked_tensor.masked_dims:
        t = torch.where(masked_tensor._bool_mask(d), t, torch.full((), lo, dtype=t.dtype, device=t.device))
    mx, idx = torch.max(t.refine_names(*masked_tensor.names), dim)
    return MaskedTensor._from_named(mx, masked_tensor.masks, apply_mask=True), idx.rename(None)


@implements(F.conv2d)
def torch_conv2d(inp, *args, **kwargs):
    """conv2d then re-mask: the bias must not leak into the padding (maskedtensor.py:230-238)"""
    return _like(inp, F.conv2d(inp.tensor, *args, **kwargs), True)


@implements(F.linear)
def torch_linear(inp, *args, **kwargs):
    return _like(inp, F.linear(inp.tensor, *args, **kwargs), True)


@implements(torch.cat)
def torch_cat(tensors, dim=0):
    """cat of the raw tensors, union of the masks, no re-mask (maskedtensor.py:250-263)"""
    first = next(a for a in tensors if isinstance(a, MaskedTensor))
    raw = [a.tensor if isinstance(a, MaskedTensor) else a for a in tensors]
    return MaskedTensor._from_named(torch.cat(raw, dim=dim).refine_names(*first.names), _union_masks(tensors), apply_mask=False)


@implements(torch.stack)
def torch_stack(tensors, dim=0):
    first = next(a for a in tensors if isinstance(a, MaskedTensor))
    raw = [a.tensor if isinstance(a, MaskedTensor) else a for a in tensors]
    names = first.names[:dim] + (None,) + first.names[dim:]
    return MaskedTensor._from_named(torch.stack(raw, dim=dim).refine_names(*names), _union_masks(tensors), apply_mask=False)


@implements(torch.flatten)
def torch_flatten(inp, start_dim=0, end_dim=-1):
    end = end_dim % inp.tensor.dim()
    names = inp.names[:start_dim] + (None,) + inp.names[end + 1:]
    return MaskedTensor._from_named(torch.flatten(inp.tensor, start_dim, end_dim).refine_names(*names), inp.masks, apply_mask=False)


def _valid_count(mt, keepdim):
    """number of valid entries over the masked dims per (batch, other dims): product of the per-name counts"""
    cnt = torch.ones((mt.tensor.size(0),), dtype=mt.tensor.dtype, device=mt.tensor.device)
    for d in mt.masked_dims:
        cnt = cnt * mt.masks[mt.names[d]].to(mt.tensor.dtype)
    shape = [mt.tensor.size(0)] + [1] * (mt.tensor.dim() - 1)
    cnt = cnt.view(shape)
    if not keepdim:
        for d in sorted(mt.masked_dims, reverse=True):
            cnt = cnt.squeeze(d)
    return cnt


@implements(torch.mean)
def torch_mean(masked_tensor, keepdim=False, *args, **kwargs):
    """mean over ALL masked dims (the `dim` argument is ignored, like the reference: maskedtensor.py:319-326);
    returns a plain tensor"""
    return torch.sum(masked_tensor.tensor, dim=masked_tensor.masked_dims, keepdim=keepdim) / _valid_count(masked_tensor, keepdim)


@implements(torch.var)
def torch_var(masked_tensor, keepdim=False, *args, **kwargs):
    """biased variance over all masked dims with the padding excluded (maskedtensor.py:328-335)"""
    means = torch_mean(masked_tensor, keepdim=True)
    sq = _like(masked_tensor, (masked_tensor.tensor - means) ** 2, True)
    return torch.sum(sq.tensor, dim=masked_tensor.masked_dims, keepdim=keepdim) / _valid_count(masked_tensor, keepdim)


@implements(F.instance_norm)
def torch_instance_norm(masked_tensor, eps=1e-05, weight=None, bias=None, *args, **kwargs):
    """InstanceNorm2d on (b, f, n, n) without running statistics (maskedtensor.py:337-351); accepts the positional
    signature of F.instance_norm(input, running_mean, running_var, weight, bias, use_input_stats, momentum, eps)"""
    if args or not isinstance(eps, float):          # called as F.instance_norm(x, rm, rv, weight, bias, use, mom, eps)
        full = (eps, weight, bias) + tuple(args)
        weight, bias = full[2] if len(full) > 2 else None, full[3] if len(full) > 3 else None
        eps = full[6] if len(full) > 6 else kwargs.get('eps', 1e-05)
    weight = kwargs.get('weight', weight)
    bias = kwargs.get('bias', bias)
    eps = kwargs.get('eps', eps)
    means = torch_mean(masked_tensor, keepdim=True)
    var_s = torch_var(masked_tensor, keepdim=True)
    res = (masked_tensor.tensor - means) / torch.sqrt(var_s + eps)
    if weight is not None and bias is not None:
        res = weight.reshape(1, -1, 1, 1) * res + bias.reshape(1, -1, 1, 1)
    return _like(masked_tensor, res, True)


@implements(F.layer_norm)
def torch_layer_norm(masked_tensor, *args, **kwargs):
    """layer_norm across un-masked (channel) dims, then re-mask (maskedtensor.py:353-364)"""
    return _like(masked_tensor, F.layer_norm(masked_tensor.tensor, *args, **kwargs), True)


@implements(torch.diag_embed)
def torch_diag_embed(inp, offset=0, dim1=-2, dim2=-1, *args, **kwargs):
    last = inp.names[-1]
    names = inp.names + (last + '_',)
    masks = dict(inp.masks)
    masks[last + '_'] = inp.masks[last]
    res = torch.diag_embed(inp.tensor, offset=offset, dim1=dim1, dim2=dim2)
    return MaskedTensor._from_named(res.refine_names(*names), masks, apply_mask=True)


@implements(F.nll_loss)
def torch_nll_loss(masked_tensor, target, *args, **kwargs):
    return F.nll_loss(masked_tensor.tensor, target, *args, **kwargs)


@implements(F.cross_entropy)
def torch_cross_entropy(masked_tensor, target, *args, **kwargs):
    return F.cross_entropy(masked_tensor.tensor, target, *args, **kwargs)
