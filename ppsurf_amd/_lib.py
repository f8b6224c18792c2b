"""ctypes binding of the C ABI declared in include/ppsurf_amd.h.

The product path has NO fallback: if libppsurf_amd.so is missing or fails to load, importing an op raises.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libppsurf_amd.so')
if os.environ.get('PPS_LIB_VARIANT'):          # development aid: an ablation / tuning build made by `python -m ppsurf_amd.build --variant NAME`
    LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), 'libppsurf_amd_{}.so'.format(os.environ['PPS_LIB_VARIANT']))

_c = ctypes
_P, _I64, _I, _SZ = _c.c_void_p, _c.c_int64, _c.c_int, _c.c_size_t

# name -> (restype, argtypes); mirrors include/ppsurf_amd.h one to one
SIGNATURES = {
    'pps_abi_version': (_I, []),
    'pps_device_cu_count': (_I, []),
    'pps_knn_f32': (_I, [_P, _I64, _P, _I64, _I, _P, _P, _P]),
    'pps_knn_blocked_f32': (_I, [_P, _P, _P, _I64, _I64, _P, _I64, _P, _I64, _I, _P, _P, _P]),
    'pps_knn_blocked_groups_f32': (_I, [_P, _P, _P, _I64, _I64, _P, _I64, _P, _P, _I64, _I, _P, _P, _P]),
    'pps_knn_multi_f32': (_I, [_I, _P, _P, _P, _P, _P, _P, _P]),
    'pps_knn_blocked_batch_f32': (_I, [_I, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'pps_voxel_sample_max_points': (_I, []),
    'pps_voxel_sample_f32': (_I, [_P, _I64, _I64, _c.c_float, _P, _I, _c.c_uint32, _P, _P, _P, _P]),
    'pps_voxel_sample_large_ws_bytes': (_SZ, [_I64]),
    'pps_voxel_sample_large_f32': (_I, [_P, _I64, _I64, _c.c_float, _P, _I, _c.c_uint32, _P, _P, _P, _P, _SZ, _P]),
    'pps_voxel_sample_batch_f32': (_I, [_P, _I64, _I64, _I64, _P, _I, _c.c_uint32, _P, _P, _P, _P]),
    'pps_patch_normalize_f32': (_I, [_P, _P, _P, _I64, _I64, _I, _P, _P]),
    'pps_mc_cube_blocks': (_I64, [_I64, _I64, _I64]),
    'pps_mc_edge_blocks': (_I64, [_I64, _I64, _I64]),
    'pps_mc_count_f64': (_I, [_P, _I64, _I64, _I64, _c.c_double, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'pps_mc_emit_f64': (_I, [_P, _I64, _I64, _I64, _c.c_double, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _P, _P]),
    'pps_mesh_components_ws_bytes': (_SZ, [_I64]),
    'pps_mesh_small_components': (_I, [_P, _I64, _I64, _I, _P, _P, _P]),
    'pps_mesh_weld_ws_bytes': (_SZ, [_I64]),
    'pps_mesh_corner_weld': (_I, [_P, _I64, _I, _P, _P, _P, _P, _P]),
    'pps_mesh_face_filter_ws_bytes': (_SZ, [_I64]),
    'pps_mesh_face_filter': (_I, [_P, _I64, _P, _P, _P, _P]),
    'pps_dilate_box_u8': (_I, [_P, _P, _P, _I64, _I64, _I64, _I, _P]),
    'pps_grow_frontier_f64': (_I, [_P, _P, _P, _P, _P, _I64, _P]),
    'pps_grow_band_todo_f64': (_I, [_P, _P, _P, _I64, _P]),
    'pps_packed_dense_floats': (_SZ, [_I, _I]),
    'pps_pack_dense_f32': (_I, [_P, _I, _I, _P]),
    'pps_packed_dense_f16x3_halfs': (_SZ, [_I, _I]),
    'pps_pack_dense_f16x3': (_I, [_P, _I, _I, _P]),
    'pps_packed_xyz_floats': (_SZ, [_I]),
    'pps_pack_xyz_f32': (_I, [_P, _I, _P]),
    'pps_rows_dense256_f32': (_I, [_P, _I64, _I64, _I64, _P, _P, _P, _P]),
    'pps_interp_pool_f32': (_I, [_P, _P, _P, _P, _I64, _I, _P, _P, _P, _P]),
    'pps_interp_small_f32': (_I, [_P, _P, _P, _P, _I64, _I, _I, _P, _P, _P, _I, _P, _P]),
    'pps_interp_small_f16x3': (_I, [_P, _P, _P, _P, _I64, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P]),
    'pps_pointnet_stn_rows_f32': (_I, [_P, _I64, _I, _P, _P, _P, _P]),
    'pps_pointnet_stn_fc_f32': (_I, [_P, _I64, _P, _P, _P, _P]),
    'pps_pointnet_feat_rows_f32': (_I, [_P, _P, _I64, _I, _P, _P, _P, _P]),
    'pps_decode_tail_f32': (_I, [_P, _P, _I64, _P, _P, _P, _P, _P]),
    'pps_decode_ws_bytes': (_SZ, [_I64]),
    'pps_decode_fwd_f32': (_I, [_P, _P, _P, _P, _I64, _I, _P, _I, _P, _P, _P, _P, _P]),
    'pps_decode_fwd_events_f32': (_I, [_P, _P, _P, _P, _I64, _I, _P, _I, _P, _P, _P, _P, _P, _P]),
    'pps_decode_fwd_mixed_f32': (_I, [_P, _P, _P, _P, _I64, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P]),
    'pps_interp_pool_f16x3': (_I, [_P, _P, _P, _P, _I64, _I, _P, _P, _P, _P, _P, _P]),
    'pps_pointnet_f16x3': (_I, [_P, _I64, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    'pps_decode_tail_f16x3': (_I, [_P, _P, _I64, _P, _P, _P, _P, _P, _P]),
    'pps_fkaconv_geo_floats': (_SZ, []),
    'pps_fkaconv_ws_bytes': (_SZ, [_I64, _I]),
    'pps_fkaconv_fwd_f32': (_I, [_P, _P, _P, _P, _I64, _I64, _I, _I, _I, _P, _P, _P, _I, _P, _P, _P]),
    'pps_rows_linear_f32': (_I, [_P, _P, _I, _P, _P, _I, _P, _P, _P, _I, _I64, _I, _P, _P]),
    'pps_rows_gemm_f32': (_I, [_P, _P, _I, _P, _P, _I, _P, _P, _P, _I, _I64, _I, _P, _P]),
    'pps_gather_max_f32': (_I, [_P, _P, _I64, _I, _I, _P, _P]),
    'pps_csr_ws_bytes': (_SZ, [_I64, _I64]),
    'pps_csr_build': (_I, [_P, _I64, _I64, _I64, _I64, _I, _P, _P, _P, _P, _SZ, _P]),
    'pps_gather_rows_f32': (_I, [_P, _P, _I64, _I, _P, _P]),
    'pps_segment_sum_rows_f32': (_I, [_P, _P, _P, _I64, _I, _P, _P]),
    'pps_segment_sum_rows_16': (_I, [_P, _P, _P, _I64, _I, _I, _P, _P]),
    'pps_neighbour_contract_fwd_f32': (_I, [_P, _P, _P, _I64, _I, _I, _P, _P]),
    'pps_neighbour_contract_bwd_f32': (_I, [_P, _P, _P, _P, _I64, _I, _I, _P, _P, _P]),
    'pps_neighbour_contract_16_supported': (_I, [_I, _I]),
    'pps_neighbour_contract_fwd': (_I, [_P, _P, _P, _I64, _I, _I, _I, _P, _P]),
    'pps_neighbour_contract_bwd': (_I, [_P, _P, _P, _P, _I64, _I, _I, _I, _P, _P, _P]),
    'pps_gather_max_arg_f32': (_I, [_P, _P, _I64, _I, _I, _P, _P, _P]),
    'pps_gather_max_bwd_f32': (_I, [_P, _P, _P, _P, _I64, _I, _I, _P, _P]),
    'pps_gather_max_arg_16': (_I, [_P, _P, _I64, _I, _I, _I, _P, _P, _P]),
    'pps_gather_max_bwd_16': (_I, [_P, _P, _P, _P, _I64, _I, _I, _I, _P, _P]),
    'pps_fka_train_ws_bytes': (_SZ, [_I64, _I64, _I]),
    'pps_fka_geometry_fwd_f32': (_I, [_P, _P, _P, _I64, _I64, _I, _P, _c.c_float, _P, _P, _P, _P]),
    'pps_fka_geometry_bwd_f32': (_I, [_P, _P, _P, _I64, _I64, _I, _P, _P, _P, _P, _P, _P]),
    'pps_attn_pool_fwd': (_I, [_P, _P, _I64, _I, _I, _I, _I, _I, _P, _P]),
    'pps_attn_pool_bwd': (_I, [_P, _P, _P, _I64, _I, _I, _I, _I, _I, _P, _P, _P]),
    'pps_attn_pool_bwd_weights': (_I, [_P, _P, _P, _I64, _I, _I, _I, _I, _I, _P, _P, _P]),
    'pps_patch_attn_partials': (_I, [_I64]),
    'pps_patch_attn_fwd': (_I, [_P, _P, _I64, _I, _I, _I, _P, _P]),
    'pps_patch_attn_bwd': (_I, [_P, _P, _P, _I64, _I, _I, _I, _P, _P, _P]),
    'pps_patch_attn_bwd_weights': (_I, [_P, _P, _P, _I64, _I, _I, _I, _P, _P, _P, _P]),
    'pps_head_input_ws_bytes': (_SZ, [_I]),
    'pps_head_input_fwd': (_I, [_P, _P, _P, _P, _I64, _I, _I, _I, _P, _P, _P]),
    'pps_head_input_dwx': (_I, [_P, _P, _P, _P, _I64, _I, _I, _I, _P, _P, _P]),
    'pps_head_chain_ws_bytes': (_SZ, []),
    'pps_head_chain_fwd': (_I, [_P, _P, _P, _P, _I64, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'pps_rows3_ws_bytes': (_SZ, []),
    'pps_rows3_fwd': (_I, [_P, _I64, _P, _P, _I, _P, _P, _P, _P, _P, _c.c_float, _c.c_float, _P, _P, _P, _P]),
    'pps_rows3_bwd': (_I, [_P, _P, _P, _I64, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'pps_patch_transform_ws_bytes': (_SZ, []),
    'pps_patch_transform_fwd': (_I, [_P, _P, _P, _I, _P, _I, _I64, _I, _I, _P, _P]),
    'pps_patch_transform_bwd': (_I, [_P, _P, _P, _I, _P, _I, _P, _I64, _I, _I, _P, _P, _P, _P, _P]),
    'pps_rows_extrema_16': (_I, [_P, _I64, _I, _I, _I, _P, _P, _P, _P, _P]),
    'pps_rows_layer_supported': (_I, [_I, _I]),
    'pps_rows_layer_ws_bytes': (_SZ, [_I, _I]),
    'pps_rows_layer_fwd': (_I, [_P, _I64, _I, _I, _P, _P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _c.c_float, _c.c_float, _P, _P, _P, _P]),
    'pps_rows_layer_bwd': (_I, [_P, _P, _P, _I64, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'pps_rows_layer_bwd_attn': (_I, [_P, _P, _I64, _I, _I, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P]),
    'pps_rows_layer_pooled_supported': (_I, [_I, _I, _I]),
    'pps_rows_layer_bwd_pooled': (_I, [_P, _P, _P, _P, _I, _I64, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'pps_rows_layer_bwd_rank2': (_I, [_P, _P, _P, _P, _P, _P, _I, _I64, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'pps_bn_train_ws_bytes': (_SZ, [_I64, _I]),
    'pps_bn_train_fwd': (_I, [_P, _I64, _I, _I, _P, _P, _P, _P, _c.c_float, _c.c_float, _I, _P, _P, _P, _P]),
    'pps_bn_train_bwd': (_I, [_P, _P, _I64, _I, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P]),
    'pps_bn_add_relu_fwd': (_I, [_P, _P, _I64, _I, _I, _P, _P, _P, _P, _c.c_float, _c.c_float, _P, _P, _P, _P]),
    'pps_bn_add_relu_bwd': (_I, [_P, _P, _P, _I64, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'pps_col_sum': (_I, [_P, _I64, _I, _I, _P, _P, _P]),
    'pps_col_sum_strided': (_I, [_P, _I64, _I, _I64, _I, _P, _P, _P]),
    'pps_gemm_nt_16': (_I, [_P, _I64, _P, _I64, _P, _P, _I64, _I64, _I, _I, _I, _I, _P]),
    'pps_gemm_tn_ws_bytes': (_SZ, [_I64, _I, _I]),
    'pps_gemm_tn_16': (_I, [_P, _I64, _P, _I64, _I64, _I, _I, _I, _P, _P, _P]),
    'pps_transpose_entry_bytes': (_I, []),
    'pps_transpose_cast_pieces': (_I, [_P, _I, _I64, _I, _P]),
    'pps_adamw_piece_bytes': (_I, []),
    'pps_cast_piece_bytes': (_I, []),
    'pps_cast_pieces': (_I, [_P, _I, _I, _P]),
    'pps_adamw_step': (_I, [_P, _I, _P, _I, _P, _c.c_float, _c.c_float, _c.c_float, _c.c_float, _c.c_float, _P, _P, _P]),
}

_lib = None


class PpsError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise PpsError('{} not found: build it with `python -m ppsurf_amd.build` (hipcc, gfx950). '
                           'There is no CPU fallback.'.format(LIB_PATH))
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)          # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        raise PpsError('{} failed with status {} ({})'.format(what, rc, {1: 'bad argument', 2: 'launch failure'}.get(rc, '?')))
