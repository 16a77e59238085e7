"""Ragged batches: the host-side mirror of the reference's ``MaskedTensor``
(maskedtensors/maskedtensor.py).

The reference keeps a zero-padded batch plus one float 0/1 mask per named dimension and
re-multiplies the masks after every op (``:87-112``).  Here the padding contract is the
same (exact zeros outside the valid n_i x n_i region, ``list(mt)`` yields the un-padded
per-graph tensors), but the masks are represented by one int32 vertex count per graph
(``nvalid``), which is what the HIP kernels consume; ``mask_dict`` is synthesised on
demand for callers that still want the float masks.
"""
import torch


class MaskedTensor:
    """Zero-padded batch ``tensor`` of shape (B, ..., Nmax[, Nmax]) + per-graph sizes ``nvalid`` (B,).

    ``masked_dims`` are the (batched) dimensions that are ragged, e.g. (2, 3) for
    (B, C, N, N) activations, (2,) for (B, C, N) embeddings, (1, 2) for (B, N, N) scores."""

    def __init__(self, tensor, nvalid, masked_dims, base_name='N'):
        self.tensor = tensor
        self.nvalid = nvalid.to(device=tensor.device, dtype=torch.int32)
        self.masked_dims = tuple(masked_dims)
        self.base_name = base_name
        self._sizes = None

    # -- reference-compatible surface ------------------------------------------------
    @property
    def mask_dict(self):
        """Float 0/1 masks keyed like the reference ('N', 'N_', ...), maskedtensor.py:40-46."""
        out = {}
        for i, dim in enumerate(self.masked_dims):
            size = self.tensor.size(dim)
            ar = torch.arange(size, device=self.tensor.device)
            out[self.base_name + '_' * i] = (ar[None, :] < self.nvalid[:, None]).to(self.tensor.dtype)
        return out

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

    def size(self, *args):
        return self.tensor.size(*args)

    def dim(self):
        return self.tensor.dim()

    def sizes(self):
        """Python list of the per-graph vertex counts (one host sync, cached)."""
        if self._sizes is None:
            self._sizes = [int(v) for v in self.nvalid.tolist()]
        return self._sizes

    def to(self, *args, **kwargs):
        return MaskedTensor(self.tensor.to(*args, **kwargs), self.nvalid, self.masked_dims, self.base_name)

    def __len__(self):
        return self.tensor.size(0)

    def __getitem__(self, index):
        item = self.tensor[index]
        n = self.sizes()[index]
        for dim in self.masked_dims:
            item = torch.narrow(item, dim - 1, 0, n)
        return item

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def __repr__(self):
        return 'MaskedTensor(shape=%s, nvalid=%s)' % (tuple(self.tensor.shape), self.sizes())


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
