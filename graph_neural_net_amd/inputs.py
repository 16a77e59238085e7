"""Device-side input pipeline: bit-packed adjacency matrices are expanded
on the GPU into the (G, 2, n, n) tensor representation of the reference
(loaders/data_generator.py:118-125: channel 0 = W, channel 1 = diag(degree))."""
import torch

from . import _lib


def expand_adjacency(bits, n, nvalid=None):
    """bits: (G, n, ceil(n/32)) int32/uint32 device tensor (see synthetic.pack_adjacency) -> (G, 2, n, n) fp32."""
    if not bits.is_cuda:
        raise RuntimeError('expand_adjacency: bits are on %s; no CPU path' % (bits.device,))
    bits = bits.contiguous()
    G = bits.shape[0]
    if bits.shape[1] != n or bits.shape[2] != (n + 31) // 32 or bits.element_size() != 4:
        raise RuntimeError('expand_adjacency: expected (G, %d, %d) 32-bit words, got %s' % (n, (n + 31) // 32, tuple(bits.shape)))
    x = torch.empty(G, 2, n, n, dtype=torch.float32, device=bits.device)
    nv = nvalid.to(device=bits.device, dtype=torch.int32) if nvalid is not None else None
    _lib.call('fgnn_expand_adjacency', _lib.ptr(bits), _lib.ptr(nv) if nv is not None else None, G, n, _lib.ptr(x),
              _lib.stream_ptr())
    return x
