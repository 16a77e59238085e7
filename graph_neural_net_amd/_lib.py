"""ctypes binding of ``libfgnn_hip.so`` (C ABI declared in ``include/fgnn_hip.h``).

The product path has no CPU fallback: if the library is missing, or an entry
point reports an error, a ``RuntimeError`` is raised.  PyTorch is used only for
device memory and the stream handle; tensors cross the boundary as raw device
pointers.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('FGNN_LIB') or os.path.join(_HERE, 'libfgnn_hip.so')      # FGNN_LIB: A/B runs of one-off debug builds

FGNN_H = 32
FGNN_TILE = 32
FGNN_MAX_DEPTH = 3
FGNN_RANGE_WG = 256          # include/fgnn_hip.h
FGNN_LSAP_MAX_N = 2048       # include/fgnn_hip.h: largest graph of fgnn_lsap_accuracy
FGNN_SCORE_SPLIT = 4

c_float_p = C.c_void_p   # device pointers travel as integers


class Slab(C.Structure):
    _fields_ = [('ptr', C.c_void_p), ('gstride', C.c_longlong), ('ldp', C.c_longlong), ('C', C.c_int),
                ('nrm', C.c_void_p), ('beta', C.c_void_p)]


class MlpFwdArgs(C.Structure):
    _fields_ = [('G', C.c_int), ('N', C.c_int), ('depth', C.c_int), ('nmlp', C.c_int),
                ('nvalid', C.c_void_p),
                ('a', Slab), ('b', Slab),
                ('W', (C.c_void_p * FGNN_MAX_DEPTH) * 2),
                ('bias', (C.c_void_p * FGNN_MAX_DEPTH) * 2),
                ('z', C.c_void_p * 2),
                ('ldz', C.c_longlong),
                ('part', C.c_void_p * 2),
                ('cnt', C.c_void_p), ('packed', C.c_void_p), ('xbits', C.c_void_p), ('xdeg', C.c_void_p),
                ('ranges', C.c_void_p), ('cu_share', C.c_int)]


class DwJob(C.Structure):
    _fields_ = [('dy', C.c_void_p), ('d_gstride', C.c_longlong), ('d_ld', C.c_longlong), ('relu_mask', C.c_void_p),
                ('x', C.c_void_p), ('x_gstride', C.c_longlong), ('x_ld', C.c_longlong), ('M', C.c_int), ('K', C.c_int)]


class ChainLayer(C.Structure):
    _fields_ = [('W', C.c_void_p), ('w_ostride', C.c_longlong), ('w_kstride', C.c_longlong), ('bias', C.c_void_p),
                ('M', C.c_int), ('K', C.c_int), ('relu', C.c_int), ('mask', C.c_void_p), ('out', C.c_void_p),
                ('o_gstride', C.c_longlong)]


class ChainArgs(C.Structure):
    _fields_ = [('x', C.c_void_p), ('x_gstride', C.c_longlong), ('x_ld', C.c_longlong), ('depth', C.c_int),
                ('layer', ChainLayer * 3), ('o_ld', C.c_longlong), ('nvalid', C.c_void_p), ('G', C.c_int), ('N', C.c_int)]


class Mlp64Args(C.Structure):          # fgnn_mlp64_args (csrc/mlp64.hip)
    _fields_ = [('x', C.c_void_p), ('x_gstride', C.c_longlong), ('x_ld', C.c_longlong), ('cin', C.c_int),
                ('xb', C.c_void_p), ('xb_gstride', C.c_longlong), ('xb_ld', C.c_longlong), ('cb', C.c_int),
                ('packed', C.c_void_p), ('nvalid', C.c_void_p), ('G', C.c_int), ('N', C.c_int),
                ('out', C.c_void_p), ('o_gstride', C.c_longlong), ('o_ld', C.c_longlong),
                ('dz', C.c_void_p), ('dz_gstride', C.c_longlong), ('dz_ld', C.c_longlong),
                ('dx', C.c_void_p), ('dx_gstride', C.c_longlong), ('dx_ld', C.c_longlong),
                ('dxb', C.c_void_p), ('dxb_gstride', C.c_longlong), ('dxb_ld', C.c_longlong),
                ('accumulate_dx', C.c_int), ('accumulate_dxb', C.c_int),
                ('wpart', C.c_void_p)]


class Mlp64PackJob(C.Structure):       # fgnn_mlp64_pack_job
    _fields_ = [('W', C.c_void_p * 3), ('bias', C.c_void_p * 3), ('cin', C.c_int), ('packed', C.c_void_p)]


class MlpBwdArgs(C.Structure):
    _fields_ = [('G', C.c_int), ('N', C.c_int), ('depth', C.c_int),
                ('nvalid', C.c_void_p),
                ('a', Slab), ('b', Slab),
                ('W', C.c_void_p * FGNN_MAX_DEPTH),
                ('bias', C.c_void_p * FGNN_MAX_DEPTH),
                ('dy', C.c_void_p), ('dgstride', C.c_longlong), ('ldd', C.c_longlong),
                ('z', C.c_void_p), ('zgstride', C.c_longlong), ('ldz', C.c_longlong),
                ('coef', C.c_void_p), ('s12', C.c_void_p), ('znrm', C.c_void_p),
                ('dxa', C.c_void_p), ('dxa_gstride', C.c_longlong), ('dxa_ld', C.c_longlong),
                ('dxb', C.c_void_p), ('dxb_gstride', C.c_longlong), ('dxb_ld', C.c_longlong),
                ('accumulate_a', C.c_int), ('accumulate_b', C.c_int),
                ('wpart', C.c_void_p), ('s12part', C.c_void_p), ('packed', C.c_void_p),
                ('s12tiles', C.c_void_p), ('s12_out', C.c_void_p), ('xbits', C.c_void_p), ('xdeg', C.c_void_p),
                ('ranges', C.c_void_p), ('cu_share', C.c_int)]


class Slab16(C.Structure):
    _fields_ = [('ptr', C.c_void_p), ('gstride', C.c_longlong), ('ldp', C.c_longlong), ('C', C.c_int),
                ('nrm', C.c_void_p), ('beta', C.c_void_p)]


class MlpFwd16Args(C.Structure):
    _fields_ = [('G', C.c_int), ('N', C.c_int), ('ldr', C.c_int), ('depth', C.c_int), ('nmlp', C.c_int),
                ('nvalid', C.c_void_p),
                ('a', Slab16), ('b', Slab16),
                ('z', C.c_void_p * 2), ('ldz', C.c_longlong),
                ('part', C.c_void_p * 2), ('cnt', C.c_void_p), ('packed', C.c_void_p), ('ranges', C.c_void_p)]


class MlpBwd16Args(C.Structure):
    _fields_ = [('G', C.c_int), ('N', C.c_int), ('ldr', C.c_int), ('depth', C.c_int),
                ('nvalid', C.c_void_p),
                ('a', Slab16), ('b', Slab16),
                ('dy', C.c_void_p), ('dgstride', C.c_longlong), ('ldd', C.c_longlong),
                ('z', C.c_void_p), ('zgstride', C.c_longlong), ('ldz', C.c_longlong),
                ('coef', C.c_void_p),
                ('dxa', C.c_void_p), ('dxa_gstride', C.c_longlong), ('dxa_ld', C.c_longlong),
                ('dxb', C.c_void_p), ('dxb_gstride', C.c_longlong), ('dxb_ld', C.c_longlong),
                ('accumulate_a', C.c_int), ('accumulate_b', C.c_int),
                ('wpart', C.c_void_p), ('s12part', C.c_void_p), ('packed', C.c_void_p), ('ranges', C.c_void_p)]


class GradJob(C.Structure):
    _fields_ = [('wpart', C.c_void_p), ('count', C.c_int), ('out', C.c_void_p), ('s12', C.c_void_p),
                ('nrm', C.c_void_p), ('dgn_w', C.c_void_p), ('dgn_b', C.c_void_p), ('rows', C.c_int), ('scale', C.c_float), ('scale_dev', C.c_void_p)]


class PackJob(C.Structure):
    _fields_ = [('kind', C.c_int), ('ca', C.c_int), ('cb', C.c_int), ('depth', C.c_int), ('nmlp', C.c_int),
                ('W', (C.c_void_p * FGNN_MAX_DEPTH) * 2), ('bias', (C.c_void_p * FGNN_MAX_DEPTH) * 2),
                ('out', C.c_void_p)]


MAX_GRAD_JOBS = 16
MAX_PACK_JOBS = 24
_VP, _LL, _I, _F = C.c_void_p, C.c_longlong, C.c_int, C.c_float

# name -> argtypes (restype is int unless listed in _RESTYPES)
_SIGNATURES = {
    'fgnn_last_error': [],
    'fgnn_version': [],
    'fgnn_tiles_per_graph': [_I],
    'fgnn_mlp_bwd_num_workgroups': [],
    'fgnn_pack_floats': [_I, _I, _I, _I, _I],
    'fgnn_pack_operands': [_VP, _I, _VP],
    'fgnn_mlp_fwd': [C.POINTER(MlpFwdArgs), _VP],
    'fgnn_mlp_fwd_t16': [C.POINTER(MlpFwdArgs), _VP],
    'fgnn_mlp_fwd_t16_supported': [C.POINTER(MlpFwdArgs)],
    'fgnn_mlp_fwd_t16_records': [_I],
    'fgnn_mlp_x3_supported': [_I, _I, _I, _I],
    'fgnn_pack_x3_floats': [_I, _I, _I, _I, _I],
    'fgnn_pack_x3_operands': [_VP, _I, _VP],
    'fgnn_mlp_fwd_x3': [C.POINTER(MlpFwdArgs), _VP],
    'fgnn_debug_mlp_fwd_masks': [C.POINTER(MlpFwdArgs), _VP, _VP, _VP],
    'fgnn_debug_mlp_fwd_x3_masks': [C.POINTER(MlpFwdArgs), _VP, _VP, _VP],
    'fgnn_debug_matmul_variant': [_I],
    'fgnn_gn_finalize': [_VP, _VP, _VP, _VP, _I, _I, _I, _F, _VP, _VP],
    'fgnn_gn_finalize2': [_VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _F, _VP, _VP, _VP],
    'fgnn_gn_finalize_r': [_VP, _VP, _VP, _VP, _I, _I, _I, _I, _F, _VP, _VP],
    'fgnn_gn_finalize2_r': [_VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _F, _VP, _VP, _VP],
    'fgnn_gn_bwd_coef2': [_VP, _VP, _VP, _VP, _VP, _I, _I, _I, _VP, _VP, _VP],
    'fgnn_gn_plane_supported': [_I],
    'fgnn_gn_plane_fwd': [_VP, _LL, _LL, _VP, _VP, _VP, _I, _I, _I, _F, _VP, _LL, _LL, _VP, _VP],
    'fgnn_gn_plane_bwd': [_VP, _LL, _LL, _VP, _LL, _LL, _VP, _VP, _I, _I, _I, _VP, _LL, _LL, _VP, _VP, _VP, _VP],
    'fgnn_gn_stats': [_VP, _LL, _LL, _VP, _VP, _I, _I, _I, _F, _VP, _VP],
    'fgnn_gn_apply': [_VP, _LL, _LL, _VP, _VP, _VP, _I, _I, _I, _VP, _LL, _LL, _VP],
    'fgnn_ragged_tile_ranges': [_VP, _I, _I, _VP, _VP],
    'fgnn_ragged_tile_ranges_order': [_VP, _I, _I, _VP, _VP, _VP],
    'fgnn_ragged_tile_ranges16': [_VP, _I, _I, _I, _VP, _VP],
    'fgnn_conv1x1_dw_multi': [C.POINTER(DwJob), _I, _VP, _I, _I, _VP, _VP],
    'fgnn_conv_chain_supported': [_I, _I, C.POINTER(C.c_int)],
    'fgnn_conv_chain': [C.POINTER(ChainArgs), _VP],
    'fgnn_mlp64_supported': [_I, _I, _I],
    'fgnn_mlp64_num_workgroups': [],
    'fgnn_mlp64_param_count': [_I],
    'fgnn_mlp64_packed_floats': [_I],
    'fgnn_mlp64_pack': [_VP, _VP, _VP, _VP, _VP, _VP, _I, _VP, _VP],
    'fgnn_mlp64_pack_multi': [C.POINTER(Mlp64PackJob), _I, _VP],
    'fgnn_mlp64_fwd': [C.POINTER(Mlp64Args), _VP],
    'fgnn_mlp64_bwd': [C.POINTER(Mlp64Args), _VP],
    'fgnn_conv1x1': [_VP, _LL, _LL, _VP, _VP, _LL, _LL, _VP, _I, _VP, _I, _I, _I, _I, _VP, _LL, _LL, _VP],
    'fgnn_conv1x1_dw_chunks': [_I, _I],
    'fgnn_conv1x1_dw': [_VP, _LL, _LL, _VP, _VP, _LL, _LL, _VP, _I, _I, _I, _I, _VP, _VP],
    'fgnn_chan_matmul_fwd': [C.POINTER(Slab), C.POINTER(Slab), _VP, _I, _I, _VP, _LL, _LL, _VP],
    'fgnn_chan_matmul_fwd_ord': [C.POINTER(Slab), C.POINTER(Slab), _VP, _I, _I, _VP, _LL, _LL, _VP, _I, _VP],
    'fgnn_chan_matmul_fwd_fin_supported': [_I],
    'fgnn_chan_matmul_fwd_fin': [C.POINTER(Slab), C.POINTER(Slab), _VP, _VP, _VP, _VP, _VP, _F, _VP, _I, _I, _VP, _LL, _LL, _VP],
    'fgnn_chan_matmul_fwd_fin_ord': [C.POINTER(Slab), C.POINTER(Slab), _VP, _VP, _VP, _VP, _VP, _F, _VP, _I, _I, _VP, _LL, _LL, _VP, _I, _VP],
    'fgnn_chan_matmul_fwd_fin_ord_r': [C.POINTER(Slab), C.POINTER(Slab), _VP, _VP, _VP, _VP, _VP, _F, _VP, _I, _I, _I, _VP, _LL, _LL, _VP, _I, _VP],
    'fgnn_colmax_fwd': [C.POINTER(Slab), _VP, _I, _I, _VP, _VP, _VP],
    'fgnn_colmax_fwd_fin_supported': [_I],
    'fgnn_colmax_fwd_fin': [C.POINTER(Slab), _VP, _VP, _VP, _F, _VP, _I, _I, _VP, _VP, _VP],
    'fgnn_colmax_fwd_fin_r': [C.POINTER(Slab), _VP, _VP, _VP, _F, _VP, _I, _I, _I, _VP, _VP, _VP],
    'fgnn_score_ce_fwd': [_VP, _VP, _VP, _I, _I, _I, _VP, _VP, _VP, _VP],
    'fgnn_score_row_blocks': [_I, _I],
    'fgnn_score_ce_fwd_blocks': [_VP, _VP, _VP, _I, _I, _I, _I, _VP, _VP, _VP, _VP],
    'fgnn_score_ce_bwd': [_VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _VP, _VP, _VP],
    'fgnn_score_ce_step_supported': [_I, _I, _I],
    'fgnn_inv_node_count': [_VP, _I, _VP, _VP],
    'fgnn_score_ce_step': [_VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP],
    'fgnn_score_bwd': [_VP, _VP, _VP, _VP, _I, _I, _I, _VP, _VP, _VP],
    'fgnn_ce_fwd': [_VP, _VP, _I, _I, _VP, _VP, _VP],
    'fgnn_ce_bwd': [_VP, _VP, _VP, _VP, _I, _I, _VP, _VP],
    'fgnn_colmax_bwd': [_VP, _VP, _VP, _I, _I, _I, _VP, _LL, _LL, C.POINTER(Slab), _VP, _VP],
    'fgnn_gn_bwd_stats': [_VP, _LL, _LL, _VP, _LL, _LL, _VP, _VP, _I, _I, _I, _VP, _VP],
    'fgnn_gn_bwd_coef': [_VP, _VP, _VP, _I, _I, _I, _VP, _VP, _VP, _VP],
    'fgnn_gn_bwd_coef_tiles': [_VP, _VP, _VP, _I, _I, _I, _VP, _VP, _VP],
    'fgnn_grad_finalize': [_VP, _I, _I, _I, _I, _VP],
    'fgnn_gn_bwd_apply': [_VP, _LL, _LL, _VP, _LL, _LL, _VP, _VP, _I, _I, _I, _VP, _LL, _LL, _VP],
    'fgnn_mlp_bwd': [C.POINTER(MlpBwdArgs), _VP],
    'fgnn_mlp_bwd_x3': [C.POINTER(MlpBwdArgs), _VP],
    'fgnn_mlp_bwd_pair': [C.POINTER(MlpBwdArgs), C.POINTER(MlpBwdArgs), _VP],
    'fgnn_mlp_bwd_pair_supported': [_I, _I],
    'fgnn_mlp_bwd_t16': [C.POINTER(MlpBwdArgs), _VP],
    'fgnn_mlp_bwd_t16_supported': [C.POINTER(MlpBwdArgs)],
    'fgnn_mlp_bwd_pair_t16': [C.POINTER(MlpBwdArgs), C.POINTER(MlpBwdArgs), _VP],
    'fgnn_mlp_bwd_pair_t16_supported': [_I, _I],
    'fgnn_mlp_bwd_pair_x3': [C.POINTER(MlpBwdArgs), C.POINTER(MlpBwdArgs), _VP],
    'fgnn_block1_struct_supported': [_I, _I, _I],
    'fgnn_block1_struct_table_floats': [_I],
    'fgnn_block1_struct_ws_floats': [_I, _I],
    'fgnn_block1_struct_rows': [_I, _I],
    'fgnn_block1_struct_tables': [_VP, _VP, _VP, _VP, _I, _I, _VP, _VP],
    'fgnn_block1_struct_fwd': [_VP, _VP, _I, _I, _VP, _VP, _VP, _VP, _VP, C.c_float, _VP, _VP, _VP, _LL, _LL, _VP, _VP, _VP, _VP, _VP, _VP, _VP],
    'fgnn_block1_struct_bwd': [_VP, _VP, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _LL, _LL, _VP, _VP, _VP, _VP, _VP, _VP],
    'fgnn_block1_struct_fwd_pack': [_VP, _VP, _I, _I, _VP, _VP, _VP, _VP, _VP, C.c_float, _VP, _VP, _VP, _LL, _LL, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _VP],
    'fgnn_block1_struct_fwd16_pack': [_VP, _VP, _I, _I, _I, _VP, _VP, _VP, _VP, _VP, C.c_float, _VP, _VP, _VP, _LL, _LL, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _VP],
    'fgnn_block1_struct_fwd16': [_VP, _VP, _I, _I, _I, _VP, _VP, _VP, _VP, _VP, C.c_float, _VP, _VP, _VP, _LL, _LL, _VP, _VP, _VP, _VP, _VP, _VP, _VP],
    'fgnn_block1_struct_bwd16': [_VP, _VP, _I, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _LL, _LL, _VP, _VP, _VP, _VP, _VP, _VP],
    'fgnn_mlp_bwd_coef_tiles_supported': [_I, _I],
    'fgnn_mlp_param_count': [_I, _I],
    'fgnn_reduce_partials': [_VP, _I, _I, _VP, _VP],
    'fgnn_chan_matmul_bwd': [C.POINTER(Slab), C.POINTER(Slab), _VP, _LL, _LL, _VP, _I, _I, _VP, _VP, _LL, _LL, _VP, _VP, _VP],
    'fgnn_chan_matmul_bwd_ord': [C.POINTER(Slab), C.POINTER(Slab), _VP, _LL, _LL, _VP, _I, _I, _VP, _VP, _LL, _LL, _VP, _VP, _VP, _I, _VP],
    'fgnn_sum_scale': [_VP, _I, _I, _F, _VP, _VP],
    'fgnn_adam_step': [_VP, _VP, _VP, _VP, _I, C.c_double, C.c_double, C.c_double, C.c_double, _I, C.c_double, _VP],
    'fgnn_adam_step_dev': [_VP, _VP, _VP, _VP, _I, _VP, _VP, _VP],
    'fgnn_expand_adjacency': [_VP, _VP, _I, _I, _VP, _VP],
    'fgnn_pack_adjacency': [_VP, _VP, _I, _I, _VP, _VP, _VP],
    'fgnn_pack_adjacency_ld': [_VP, _VP, _I, _I, _I, _VP, _VP, _VP],
    'fgnn_pack_adjacency_pair': [_VP, _VP, _VP, _VP, _I, _I, _I, _VP, _VP, _VP, _VP, _VP],
    'fgnn_adjacency_degree': [_VP, _VP, _I, _I, _VP, _VP],
    'fgnn_accuracy_max': [_VP, _VP, _I, _I, _VP, _VP],
    'fgnn_lsap_accuracy': [_VP, _LL, _I, _VP, _I, _I, _VP, _VP, _VP],
    # ---- bf16 path ----
    'fgnn_tiles_per_graph16': [_I, _I],
    'fgnn_to_bf16': [_VP, _VP, _I, _I, _I, _I, _VP, _LL, _LL, _VP],
    'fgnn_from_bf16': [_VP, _LL, _LL, _I, _I, _I, _I, _VP, _VP],
    'fgnn_pack16_floats': [_I, _I, _I, _I, _I],
    'fgnn_pack16_operands': [_VP, _I, _VP],
    'fgnn_mlp_fwd16': [C.POINTER(MlpFwd16Args), _VP],
    'fgnn_debug_mlp_fwd16_masks': [C.POINTER(MlpFwd16Args), _VP, _VP, _VP],
    'fgnn_gn_finalize_tpg': [_VP, _VP, _VP, _VP, _I, _I, _I, _I, _F, _VP, _VP],
    'fgnn_gn_finalize2_tpg': [_VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _F, _VP, _VP, _VP],
    'fgnn_gn_bwd_coef_tiles_tpg': [_VP, _VP, _VP, _I, _I, _I, _I, _VP, _VP, _VP],
    'fgnn_chan_matmul_fwd16': [C.POINTER(Slab16), C.POINTER(Slab16), _VP, _I, _I, _I, _VP, _LL, _LL, _VP],
    'fgnn_chan_matmul_fwd16_fin': [C.POINTER(Slab16), C.POINTER(Slab16), _VP, _VP, _VP, _VP, _VP, _F, _I, _VP, _I, _I, _I, _VP, _LL,
                                   _LL, _VP],
    'fgnn_chan_matmul_bwd16': [C.POINTER(Slab16), C.POINTER(Slab16), _VP, _LL, _LL, _VP, _I, _I, _I, _VP, _VP, _LL, _LL, _VP, _VP, _VP],
    'fgnn_chan_matmul_bwd16_t': [C.POINTER(Slab16), C.POINTER(Slab16), _VP, _LL, _LL, _VP, _I, _VP, _I, _I, _I, _VP, _VP, _LL, _LL,
                                 _VP, _VP, _VP],
    'fgnn_chan_matmul_bwd16_tc': [C.POINTER(Slab16), C.POINTER(Slab16), _VP, _LL, _LL, _VP, _I, _VP, _I, _I, _I, _VP, _VP, _LL, _LL,
                                  _VP, _VP, _VP, _VP, _VP],
    'fgnn_colmax_bwd16_coef': [_VP, _VP, _VP, _I, _I, _I, _I, _VP, _LL, _LL, C.POINTER(Slab16), _VP, _VP, _VP],
    'fgnn_colmax_fwd16': [C.POINTER(Slab16), _VP, _I, _I, _I, _VP, _VP, _VP],
    'fgnn_colmax_bwd16': [_VP, _VP, _VP, _I, _I, _I, _I, _VP, _LL, _LL, C.POINTER(Slab16), _VP, _VP],
    'fgnn_mlp_bwd16': [C.POINTER(MlpBwd16Args), _VP],
    'fgnn_mlp_bwd16_pair': [C.POINTER(MlpBwd16Args), C.POINTER(MlpBwd16Args), _VP],
}
_RESTYPES = {'fgnn_last_error': C.c_char_p, 'fgnn_block1_struct_ws_floats': C.c_longlong}
EXPORTS = tuple(_SIGNATURES)

_lib = None


def load():
    """Load the HIP library (once).  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            'graph_neural_net_amd: %s is missing. Build it with '
            '`python -c "import __graft_entry__ as g; g.build()"` or `make -C graph_neural_net_amd/csrc`. '
            'There is no CPU fallback.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    _lib = lib
    return lib


def last_error():
    return load().fgnn_last_error().decode('utf-8', 'replace')


# Optional per-launch timing (bench.py's roofline leg): when PROFILE is a list, every
# call is bracketed by two events on torch's current stream -- the stream the kernels are
# launched on -- and (tag, start, stop) is appended.
PROFILE = None


def call(name, *args, tag=None):
    """Call an int-returning entry point and raise on a non-zero status."""
    if PROFILE is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(load(), name)(*args)
        e1.record()
        PROFILE.append((tag or name, e0, e1, name, args))        # (name, args): enough to issue the same launch again, see relaunch()
    else:
        rc = getattr(load(), name)(*args)
    if rc != 0:
        raise RuntimeError('%s failed (rc=%d): %s' % (name, rc, last_error()))


def relaunch(name, args):
    """Issue a launch recorded in PROFILE once more, on the CURRENT stream (the stream is the last argument of every entry point)."""
    rc = getattr(load(), name)(*args[:-1], stream_ptr())
    if rc != 0:
        raise RuntimeError('%s failed (rc=%d): %s' % (name, rc, last_error()))


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('graph_neural_net_amd: expected a CUDA/HIP tensor, got device %s '
                           '(there is no CPU path)' % (t.device,))
    return C.c_void_p(t.data_ptr())


def make_slab(t, gstride, ldp, channels, nrm=None, beta=None):
    s = Slab()
    s.ptr = t.data_ptr() if t is not None else None
    s.gstride = gstride
    s.ldp = ldp
    s.C = channels
    s.nrm = nrm.data_ptr() if nrm is not None else None
    s.beta = beta.data_ptr() if beta is not None else None
    return s


def make_slab16(t, gstride, ldp, channels, nrm=None, beta=None):
    s = Slab16()
    s.ptr = t.data_ptr() if t is not None else None
    s.gstride = gstride
    s.ldp = ldp
    s.C = channels
    s.nrm = nrm.data_ptr() if nrm is not None else None
    s.beta = beta if isinstance(beta, int) else (beta.data_ptr() if beta is not None else None)
    return s


def tiles_per_graph(n):
    return (n * n + FGNN_TILE - 1) // FGNN_TILE


def mlp_param_count(cin, depth):
    return 32 * cin + 32 + (depth - 1) * (32 * 32 + 32)
