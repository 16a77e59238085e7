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



def pack_tensor_representation(x, nvalid=None, check=True):
    """The inverse, for loaders that produce the reference's dense batches: x (G, 2, n, n) fp32 device tensor (channel 0 = W,
    channel 1 = diag(row sums), loaders/data_generator.py:118-125) -> (G, n, ceil(n/32)) int32 bit-packed adjacency, the input form
    of `FgnnEngine.embed(bits=...)` / `FgnnTrainer.train_step_bits` and of the structured block 1.  check=True verifies on the
    device that x IS a tensor representation on the valid corners (entries of channel 0 in {0, 1}, channel 1 = diag(row sums)) and
    raises otherwise (one host sync); check='device' returns (bits, flag) with the verdict as a 1-element int32 device tensor
    instead; check=False skips it."""
    if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4 or x.shape[1] != 2 or x.shape[2] != x.shape[3]:
        raise RuntimeError('pack_tensor_representation: expected a (G, 2, n, n) fp32 device tensor, got %s %s on %s'
                           % (tuple(x.shape), x.dtype, x.device))
    x = x.contiguous()
    G, n = x.shape[0], x.shape[-1]
    bits = torch.empty(G, n, (n + 31) // 32, dtype=torch.int32, device=x.device)
    nv = nvalid.to(device=x.device, dtype=torch.int32) if nvalid is not None else None
    flag = torch.zeros(1, dtype=torch.int32, device=x.device) if check else None
    _lib.call('fgnn_pack_adjacency', _lib.ptr(x), _lib.ptr(nv) if nv is not None else None, G, n, _lib.ptr(bits),
              _lib.ptr(flag) if flag is not None else None, _lib.stream_ptr())
    if check == 'device':
        return bits, flag
    if check and int(flag.item()) != 0:
        raise RuntimeError('pack_tensor_representation: x is not the tensor representation of a 0/1 adjacency '
                           '(channel 0 in {0, 1}, channel 1 = diag(row sums)); run it through the dense path')
    return bits
