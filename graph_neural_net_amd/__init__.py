"""MI355X-native (gfx950) implementation of the 2-FGNN hot path of mlelarge/graph_neural_net.

Host side: the reference's nn.Module / MaskedTensor surface (``layers``, ``blocks``,
``network``, ``masked``, ``siamese``, ``losses``) and the fused ``engine``; device side:
hand-written HIP kernels in ``csrc/`` behind the C ABI of ``include/fgnn_hip.h``.
There is no CPU fallback: tensors must live on the GPU and ``libfgnn_hip.so`` must be built.
"""
__version__ = '0.1.0'
