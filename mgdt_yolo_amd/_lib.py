"""ctypes binding of libmgdt_hip.so (the C ABI declared in include/mgdt.h).

There is NO fallback: if the library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('MGDT_LIB') or os.path.join(_HERE, 'csrc', 'libmgdt_hip.so')      # MGDT_LIB: another build of the same ABI (tools/graph_bisect3.py)

F32, BF16 = 0, 1
ACT_NONE, ACT_SILU, ACT_RELU, ACT_GELU = 0, 1, 2, 3
SPR_SPLITS = 16
GRN_SPLITS = 8


class View(C.Structure):
    _fields_ = [('p', C.c_void_p), ('n', C.c_int32), ('h', C.c_int32), ('w', C.c_int32), ('c', C.c_int32),
                ('sn', C.c_int64), ('sh', C.c_int64), ('sw', C.c_int64), ('sc', C.c_int64)]


VP = C.POINTER(View)
_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t

# name -> (restype, argtypes); must list every symbol include/mgdt.h declares (tests check this)
class PackDesc(C.Structure):                  # == mgdt_pack_desc
    _fields_ = [('w', C.c_void_p), ('conv_bias', C.c_void_p), ('bn_gamma', C.c_void_p), ('bn_beta', C.c_void_p), ('bn_mean', C.c_void_p), ('bn_var', C.c_void_p),
                ('bn_eps', C.c_float), ('cin', C.c_int32), ('cout', C.c_int32), ('k', C.c_int32), ('dtype', C.c_int32), ('mode', C.c_int32),
                ('packed', C.c_void_p), ('bias_out', C.c_void_p)]


class WgradFinalDesc(C.Structure):            # == mgdt_wgrad_final_desc
    _fields_ = [('partial', C.c_void_p), ('dw', C.c_void_p), ('n', C.c_long), ('nsplit', C.c_int32), ('accumulate', C.c_int32)]


PROTOTYPES = {
    'mgdt_last_error': (C.c_char_p, []),
    'mgdt_version': (C.c_char_p, []),
    'mgdt_conv_packed_bytes': (_sz, [_i, _i, _i, _i]),
    'mgdt_conv_pack': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _i, _i, _i, _vp, _vp, _vp]),
    'mgdt_conv2d_fwd': (_i, [VP, VP, _vp, _vp, _vp, _vp, _i, _i, _i, VP, VP, VP, _i, _vp]),
    'mgdt_conv_packed_bytes_fp8': (_sz, [_i, _i, _i]),
    'mgdt_conv_pack_fp8': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _i, _i, _f, _vp, _vp, _vp, _vp]),
    'mgdt_conv2d_fp8_fwd': (_i, [VP, VP, _vp, _vp, _vp, _vp, _vp, _f, _i, _i, _i, VP, VP, VP, _vp]),
    'mgdt_conv_pack_batch': (_i, [_vp, _i, _vp]),
    'mgdt_conv2d_phase_fwd': (_i, [VP, _vp, _vp, _i, VP, VP, VP, _i, _vp]),
    'mgdt_conv_wgrad_splits': (_i, [_i, _i, _i]),
    'mgdt_wgrad_final_batch': (_i, [_vp, _i, _vp]),
    'mgdt_conv_pack_dgrad': (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    'mgdt_conv_pack_direct': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _i, _i, _vp, _vp, _vp]),
    'mgdt_conv2d_direct_fwd': (_i, [VP, _i, _vp, _vp, _i, _i, _i, _i, VP, _i, _vp]),
    'mgdt_spr_pool_fwd': (_i, [VP, _vp, _i, _vp]),
    'mgdt_spr_attn_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'mgdt_scale_channels_fwd': (_i, [VP, _vp, VP, _i, _vp]),
    'mgdt_sppf_pool_fwd': (_i, [VP, VP, VP, VP, _i, _vp]),
    'mgdt_adaptive_avgpool_fwd': (_i, [VP, VP, _i, _vp]),
    'mgdt_bilinear_fwd': (_i, [VP, VP, _i, _vp]),
    'mgdt_nearest_fwd': (_i, [VP, VP, _i, _vp]),
    'mgdt_copy_fwd': (_i, [VP, _i, VP, _i, _vp]),
    'mgdt_image_pad4_fwd': (_i, [VP, _i, VP, _i, _vp]),
    'mgdt_dwconv7_ln_fwd': (_i, [VP, _vp, _vp, _vp, _vp, _f, VP, _i, _vp]),
    'mgdt_cnx_mlp_packed_bytes': (_sz, [_i, _i]),
    'mgdt_cnx_mlp_pack': (_i, [_vp, _vp, _vp, _vp, _i, _vp, _i, _vp]),
    'mgdt_cnx_mlp_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'mgdt_cnx_mlp_fwd': (_i, [VP, VP, _vp, _vp, _vp, _vp, VP, _i, _vp]),
    'mgdt_cnx_block_supported': (_i, [_i, _i, _i, _i, _i]),
    'mgdt_cnx_block_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'mgdt_cnx_block_fwd': (_i, [VP, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _i, _vp, _sz, VP, _i, _vp]),
    'mgdt_pw_chain_packed_bytes': (_sz, [_i, _i]),
    'mgdt_pw_chain_pack': (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _i, _vp, _vp]),
    'mgdt_pw_chain3_fwd': (_i, [VP, _vp, _i, _i, VP, _i, _vp]),
    'mgdt_conv1x1_inject_supported': (_i, [_i, _i, _i, _i, _i, _i, _i]),
    'mgdt_conv1x1_inject_fwd': (_i, [VP, _vp, _vp, VP, VP, VP, _i, _vp]),
    'mgdt_conv1x1_inject_conv_supported': (_i, [_i, _i, _i, _i, _i, _i, _i, _i]),
    'mgdt_conv1x1_inject_conv_fwd': (_i, [VP, _vp, _vp, VP, VP, VP, _vp, _vp, _i, _vp, _vp, _i, VP, _i, _vp]),
    'mgdt_spr_attn_scale_fwd': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, VP, VP, VP, VP, _i, _vp]),
    'mgdt_stem2_packed_bytes': (_sz, []),
    'mgdt_stem2_pack': (_i, [_vp, _vp, _vp]),
    'mgdt_stem2_fwd': (_i, [VP, _i, _vp, _vp, _vp, _vp, VP, _vp]),
    'mgdt_detect_tail_supported': (_i, [_i, _i, _i, _i, _i]),
    'mgdt_detect_tail_fwd': (_i, [VP, VP, _vp, _vp, _vp, _vp, _i, _f, _i, _i, VP, _vp, _vp, _vp, _vp, _vp]),
    'mgdt_csp_block_supported': (_i, [_i, _i, _i, _i, _i, _i, _i, _i]),
    'mgdt_csp_block_tiles': (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'mgdt_csp_block_fwd': (_i, [_i, VP, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _i, VP, _vp, _i, _vp]),
    'mgdt_groupnorm_workspace_bytes': (_sz, [_i, _i]),
    'mgdt_groupnorm_fwd': (_i, [VP, _vp, _vp, _i, _f, _i, _vp, VP, _i, _vp]),
    'mgdt_tood_layer_attn_fwd': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    'mgdt_dcnv2_fwd': (_i, [VP, VP, _vp, _vp, VP, _i, _vp]),
    'mgdt_dcnv2_mfma_fwd': (_i, [VP, VP, _vp, VP, _i, _vp]),
    'mgdt_pixel_gate_fwd': (_i, [VP, VP, VP, _i, _vp]),
    'mgdt_gn_affine': (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _f, _vp, _vp, _vp, _vp, _vp]),
    'mgdt_nc_affine_act_bwd': (_i, [VP, VP, _vp, _vp, _i, VP, _i, _vp]),
    'mgdt_gn_bwd_workspace_bytes': (_sz, [_i, _i]),
    'mgdt_gn_bwd_coef': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    'mgdt_nc_axpby': (_i, [VP, _vp, VP, _vp, _vp, VP, _i, _vp]),
    'mgdt_pixel_gate_bwd': (_i, [VP, VP, VP, VP, VP, _i, _vp]),
    'mgdt_tood_layer_attn_bwd_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'mgdt_tood_layer_attn_bwd': (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    'mgdt_dcn_im2col': (_i, [VP, VP, VP, _i, _vp]),
    'mgdt_dcn_col2im_bwd': (_i, [VP, VP, VP, _vp, VP, _i, _vp]),
    'mgdt_val_match_fwd': (_i, [_vp, _vp, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _vp]),
    'mgdt_grn_stats_fwd': (_i, [VP, _vp, _vp, _vp, _i, _vp]),
    'mgdt_inject_fwd': (_i, [VP, VP, VP, VP, _i, _vp]),
    'mgdt_detect_decode_fwd': (_i, [VP, _i, _i, _f, _i, _i, _vp, _i, _vp]),
    'mgdt_detect_loss_workspace_bytes': (_sz, [_i, _i, _i]),
    'mgdt_detect_loss_fwd': (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _i, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _sz, _i, _vp]),
    'mgdt_detect_loss_fwd_dev': (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _sz, _i, _vp]),
    'mgdt_detect_loss_bwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _i, _f, _f, _f, _f, _vp, _vp, _sz, _i, _vp]),
    'mgdt_reduce_workspace_bytes': (_sz, [_i]),
    'mgdt_bn_stats_fwd': (_i, [VP, _f, _f, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    'mgdt_bn_act_fwd': (_i, [VP, _vp, _vp, _vp, _vp, _i, VP, VP, VP, _i, _vp]),
    'mgdt_bn_act_bwd': (_i, [VP, VP, _vp, _vp, _vp, _vp, _i, _vp, _vp, VP, _vp, _i, _vp]),
    'mgdt_conv_dgrad': (_i, [VP, _vp, _i, _i, VP, _i, _i, _vp]),
    'mgdt_gconv_dgrad': (_i, [VP, _vp, _i, _i, _i, VP, _i, _i, _vp]),
    'mgdt_gconv_wgrad_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'mgdt_gconv_wgrad': (_i, [VP, VP, _i, _i, _i, _vp, _i, _vp, _i, _vp]),
    'mgdt_conv_wgrad_workspace_bytes': (_sz, [_i, _i, _i]),
    'mgdt_conv_wgrad': (_i, [VP, VP, VP, _i, _i, _vp, _vp, _i, _vp, _i, _vp]),
    'mgdt_add_fwd': (_i, [VP, VP, VP, _i, _vp]),
    'mgdt_maxpool5_bwd': (_i, [VP, VP, _vp, _i, _vp]),
    'mgdt_nearest_bwd': (_i, [VP, VP, _i, _vp]),
    'mgdt_ew_binary': (_i, [VP, VP, VP, _i, _i, _vp]),
    'mgdt_channel_affine': (_i, [VP, _vp, _vp, VP, _i, _vp]),
    'mgdt_nc_reduce_workspace_bytes': (_sz, [_i, _i]),
    'mgdt_nc_reduce': (_i, [VP, VP, _vp, _vp, _i, _vp]),
    'mgdt_adaptive_avgpool_bwd': (_i, [VP, VP, _i, _i, _vp]),
    'mgdt_bilinear_bwd': (_i, [VP, VP, _i, _i, _vp]),
    'mgdt_spr_bwd_workspace_bytes': (_sz, [_i, _i, _i]),
    'mgdt_spr_bwd': (_i, [VP, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, VP, _vp, _vp, _i, _vp]),
    'mgdt_dwconv7_ln_train_fwd': (_i, [VP, _vp, _vp, _vp, _vp, _f, VP, VP, _i, _vp]),
    'mgdt_dwconv7_ln_bwd_workspace_bytes': (_sz, [_i]),
    'mgdt_dwconv7_ln_bwd': (_i, [VP, VP, VP, _vp, _vp, _f, VP, VP, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    'mgdt_grn_bwd': (_i, [VP, VP, _vp, _vp, _vp, _vp, VP, _vp, _vp, _vp, _i, _vp]),
    'mgdt_grad_norm_workspace_bytes': (_sz, []),
    'mgdt_grad_clip_coef': (_i, [_vp, C.c_long, _f, _vp, _vp, _vp]),
    'mgdt_sgd_step': (_i, [_vp, _vp, _vp, _vp, C.c_long, _f, _f, _f, _i, _i, _vp, _vp]),
    'mgdt_ema_update': (_i, [_vp, _vp, C.c_long, _f, _vp]),
    'mgdt_sgd_ema_step_dev': (_i, [_vp, _vp, _vp, _vp, C.c_long, _vp, C.c_long, _vp, _i, _i, _vp, _vp]),
    'mgdt_box_convert': (_i, [_vp, _vp, C.c_long, _i, _i, _vp]),
    'mgdt_box_iou': (_i, [_vp, _i, _vp, _i, _f, _vp, _vp]),
    'mgdt_bbox_iou': (_i, [_vp, _i, _vp, _i, C.c_long, _i, _i, _f, _vp, _vp]),
    'mgdt_scale_boxes': (_i, [_vp, C.c_long, _i, _f, _f, _f, _f, _f, _vp]),
    'mgdt_letterbox_fwd': (_i, [_vp, _i, _i, C.c_long, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'mgdt_ap_workspace_bytes': (_sz, [_i, _i]),
    'mgdt_ap_per_class': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, C.c_double, _vp, _vp, _vp, _vp, _vp]),
    'mgdt_nms_workspace_bytes': (_sz, [_i, _i, _i, _i, _i]),
    'mgdt_nms_fwd': (_i, [_vp, _i, _i, _i, _f, _f, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
}

_lib = None


def lib():
    """The loaded library (raises loudly when it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                               '(mgdt_yolo_amd has no CPU or PyTorch fallback)')
        import torch  # noqa: F401  torch's bundled libamdhip64 must be the HIP runtime of the process: loaded first, our DT_NEEDED
        l = C.CDLL(LIB_PATH)          # libamdhip64.so.7 resolves to it (a second runtime from /opt/rocm would see no device)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def check(status, what=''):
    if status != 0:
        raise RuntimeError(f'{what or "mgdt"} failed ({status}): {lib().mgdt_last_error().decode()}')
