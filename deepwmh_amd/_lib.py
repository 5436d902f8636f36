"""ctypes binding of libdeepwmh_amd.so (include/deepwmh_amd.h).  There is NO fallback: if the HIP
library is missing or a call fails, this module raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('WMH_LIB_PATH', os.path.join(_HERE, 'lib', 'libdeepwmh_amd.so'))   # override: diagnostic builds
# the same sources with -DWMH_DIAG (cycle stamps in the kernels, WMH_* variant switches from the environment): tools and a few tests only
DIAG_LIB_PATH = os.environ.get('WMH_DIAG_LIB_PATH', os.path.join(_HERE, 'lib', 'libdeepwmh_amd_diag.so'))

_vp, _i, _f, _sz, _d = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_double
_ip = C.POINTER(C.c_int)

# name -> (restype, argtypes); mirrors include/deepwmh_amd.h one to one
SIGNATURES = {
    'wmh_last_error': (C.c_char_p, []),
    'wmh_abi_version': (_i, []),
    'wmh_device_check': (_i, [_i]),
    'wmh_conv3d_k3_config': (_i, [_i, _i, _i, _i, _i, _i]),
    'wmh_conv3d_config': (_i, [_i, _i, _i, _i, _i, _i, _i, _i]),
    'wmh_conv3d_k3_packed_floats': (_sz, [_i, _i, _i]),
    'wmh_conv3d_k3_pack_weights': (_i, [_i, _vp, _i, _i, _vp]),
    'wmh_conv3d_k3_num_partials': (_i, [_i, _i, _i, _i, _i]),
    'wmh_conv3d_k3_workspace_floats': (_sz, [_i, _i, _i, _i, _i, _i, _i, _i]),
    'wmh_conv3d_k3_fwd': (_i, [_i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    'wmh_conv3d_k3_bf16_supported': (_i, [_i]),
    'wmh_conv3d_k3_bf16_config': (_i, [_i, _i, _i]),
    'wmh_conv3d_k3_f16x2_config': (_i, [_i, _i, _i]),
    'wmh_conv3d_k3_packed_halfs_bf16': (_sz, [_i, _i, _i]),
    'wmh_conv3d_k3_pack_weights_bf16': (_i, [_i, _vp, _i, _i, _vp]),
    'wmh_conv3d_k3_num_partials_bf16': (_i, [_i, _i, _i, _i, _i]),
    'wmh_conv3d_k3_num_partials_f16x2': (_i, [_i, _i, _i, _i, _i]),
    'wmh_conv3d_k3_pack_weights_bf16_device': (_i, [_i, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    'wmh_conv3d_k3_f16x2_supported': (_i, [_i]),
    'wmh_conv3d_k3_packed_halfs_f16x2': (_sz, [_i, _i, _i]),
    'wmh_conv3d_k3_pack_weights_f16x2': (_i, [_i, _vp, _i, _i, _vp]),
    'wmh_conv3d_k3_pack_weights_f16x2_device': (_i, [_i, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    'wmh_conv3d_k3_pack_job_bytes': (_sz, []),
    'wmh_conv3d_k3_pack_job_f16x2': (_i, [_i, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    'wmh_conv3d_k3_pack_weights_f16x2_batch': (_i, [_vp, _i, _vp]),
    'wmh_conv3d_k3_f16x2_workspace_floats': (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    'wmh_conv3d_k3_fwd_f16x2': (_i, [_i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    'wmh_conv3d_up_supported_f16x2': (_i, [_i, _i, _i, _i, _i, _i]),
    'wmh_conv3d_up_packed_bytes_f16x2': (_sz, [_i, _i]),
    'wmh_conv3d_up_num_partials_f16x2': (_i, [_i, _i, _i]),
    'wmh_conv3d_up_pack_weights_f16x2': (_i, [_vp, _i, _i, _vp, _i, _i, _vp]),
    'wmh_conv3d_up_fwd_f16x2': (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp]),
    'wmh_conv3d_k3_bf16_workspace_floats': (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    'wmh_conv3d_k3_fwd_bf16_splitk': (_i, [_i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    'wmh_conv3d_k3_fwd_bf16': (_i, [_i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    'wmh_conv3d_k3_fwd_bf16_st': (_i, [_i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _i, _vp]),
    'wmh_conv3d_k3_fwd_st': (_i, [_i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _i, _vp]),
    'wmh_convtranspose3d_mfma_fwd_st': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    'wmh_seghead_softmax_accumulate_st': (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _ip, _i, _ip, _f, _i, _vp]),
    'wmh_seghead_logits_st': (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _sz, _f, _i, _vp]),
    'wmh_instnorm_finalize': (_i, [_vp, _i, _vp, _vp, _f, _vp, _vp, _i, _i, _vp]),
    'wmh_instnorm_stats': (_i, [_vp, _vp, _vp, _f, _vp, _vp, _i, _i, _sz, _vp]),
    'wmh_instnorm_lrelu_apply': (_i, [_vp, _vp, _vp, _i, _i, _sz, _f, _vp]),
    'wmh_convtranspose3d_k2s2_fwd': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    'wmh_convtranspose3d_k2s2_packed_floats': (_sz, [_i, _i]),
    'wmh_convtranspose3d_k2s2_pack_weights': (_i, [_vp, _i, _i, _vp]),
    'wmh_convtranspose3d_k2s2_mfma_fwd': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    'wmh_convtranspose3d_packed_floats': (_sz, [_i, _i, _i]),
    'wmh_convtranspose3d_pack_weights': (_i, [_vp, _i, _i, _i, _vp]),
    'wmh_convtranspose3d_mfma_fwd': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _vp]),
    'wmh_extract_patches': (_i, [_vp, _vp, _i, _i, _i, _i, _ip, _i, _i, _i, _i, _ip, _ip, _vp]),
    'wmh_seghead_softmax_accumulate': (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _ip, _i, _ip, _f, _vp]),
    'wmh_seghead_logits': (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _sz, _f, _vp]),
    'wmh_accumulate_weight': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _ip, _vp]),
    'wmh_finalize_argmax': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _ip, _i, _i, _i, _vp]),
    'wmh_conv3d_k3_pack_weights_device': (_i, [_i, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    'wmh_conv3d_k3_wgrad_workspace_floats': (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    'wmh_conv3d_k3_wgrad': (_i, [_vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    'wmh_upsample_zero2': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'wmh_upsample_zero_plane': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'wmh_conv3d_wgrad_workspace_floats': (_sz, [_i, _i, _i, _i, _i, _i, _i, _i, _i]),
    'wmh_conv3d_wgrad': (_i, [_vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp]),
    'wmh_convtranspose3d_wgrad_workspace_floats': (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    'wmh_convtranspose3d_wgrad': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _vp]),
    'wmh_dice_ce_loss_workspace_doubles': (_sz, []),
    'wmh_dice_ce_loss_fwd_bwd': (_i, [_vp, _vp, _i, _i, _sz, _f, _f, _vp, _vp, _vp, _vp]),
    'wmh_online_eval_counts': (_i, [_vp, _vp, _i, _i, _sz, _vp, _vp]),
    'wmh_in_lrelu_bwd_scratch_doubles': (_sz, [_i, _i, _sz]),
    'wmh_in_lrelu_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _sz, _f, _vp]),
    'wmh_channel_sum': (_i, [_vp, _vp, _i, _i, _sz, _vp]),
    'wmh_convtranspose3d_k2s2_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    'wmh_conv3d_k3_wgrad_f16x2_supported': (_i, [_i, _i, _i, _i, _i, _i, _i]),
    'wmh_conv3d_k3_wgrad_f16x2_workspace_floats': (_sz, [_i, _i, _i, _i, _i, _i]),
    'wmh_conv3d_k3s2_wgrad_f16x2_workspace_floats': (_sz, [_i, _i, _i, _i, _i, _i]),
    'wmh_conv3d_k3_wgrad_f16x2': (_i, [_vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    'wmh_conv3d_k3s2_wgrad_f16x2': (_i, [_vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    'wmh_diag_build': (_i, []),
    'wmh_convtranspose3d_packed_dwords_f16x2': (_sz, [_i, _i, _i]),
    'wmh_convtranspose3d_pack_weights_f16x2_device': (_i, [_vp, _i, _i, _i, _vp, _vp]),
    'wmh_convtranspose3d_fwd_f16x2': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _vp]),
    'wmh_conv3d_k3s2_dgrad_packed_dwords_f16x2': (_sz, [_i, _i]),
    'wmh_conv3d_k3s2_dgrad_pack_weights_f16x2_device': (_i, [_vp, _i, _i, _vp, _vp]),
    'wmh_conv3d_k3s2_dgrad_f16x2': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'wmh_conv3d_k3s2_dgrad_packed_floats': (_sz, [_i, _i]),
    'wmh_conv3d_k3s2_dgrad_pack_weights_device': (_i, [_vp, _i, _i, _vp, _vp]),
    'wmh_conv3d_k3s2_dgrad': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'wmh_convtranspose3d_pack_weights_device': (_i, [_vp, _i, _i, _i, _vp, _vp]),
    'wmh_convtranspose3d_k2s2_wgrad_workspace_floats': (_sz, [_i, _i, _i, _i, _i, _i]),
    'wmh_convtranspose3d_k2s2_wgrad': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    'wmh_grad_clip_coef': (_i, [_vp, _sz, _f, _vp, _vp, _vp]),
    'wmh_sgd_nesterov_step': (_i, [_vp, _vp, _vp, _sz, _f, _f, _f, _vp, _i, _vp]),
    'wmh_cc_label': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'wmh_remove_small_components': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'wmh_fill_holes': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'wmh_nonzero_mask': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'wmh_mask_bbox': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'wmh_zscore_crop_workspace_doubles': (_sz, [_i]),
    'wmh_zscore_crop': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'wmh_uncrop_u8': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'wmh_resample_workspace_doubles': (_sz, [_i, _i, _i, _i]),
    'wmh_resample': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'wmh_resample_nearest_u8': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'wmh_softmax_mask': (_i, [_vp, _vp, _vp, _sz, _vp]),
    'wmh_ensemble_softmax': (_i, [_vp, _i, _vp, _vp, _sz, _vp]),
    'wmh_binary_overlap': (_i, [_vp, _vp, _vp, _sz, _vp]),
    'wmh_moments_workspace_doubles': (_sz, []),
    'wmh_masked_moments_f32': (_i, [_vp, _vp, _sz, _d, _vp, _vp, _vp]),
    'wmh_masked_moments_f64': (_i, [_vp, _vp, _sz, _d, _vp, _vp, _vp]),
    'wmh_zscore_f64': (_i, [_vp, _d, _d, _vp, _vp, _vp, _sz, _vp]),
    'wmh_mask_fill_f64': (_i, [_vp, _vp, _vp, _sz, _vp]),
    'wmh_vote_mask': (_i, [_vp, _i, _vp, _sz, _vp]),
    'wmh_label_vote': (_i, [_vp, _i, _i, _vp, _sz, _vp]),
    'wmh_histogram_f64': (_i, [_vp, _vp, _vp, _vp, _sz, _vp, _i, _vp, _vp]),
    'wmh_block_stats': (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    'wmh_zoom_linear_window': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    'wmh_sub_add_f64': (_i, [_vp, _vp, _vp, _sz, _vp]),
    'wmh_mul_mask_f64': (_i, [_vp, _vp, _sz, _vp]),
    'wmh_select_f64': (_i, [_vp, _vp, _vp, _sz, _vp]),
    'wmh_nll': (_i, [_vp, _vp, _i, _sz, _d, _i, _vp, _vp, _vp, _vp, _vp]),
    'wmh_nll_masked': (_i, [_vp, _vp, _vp, _i, _sz, _d, _i, _vp, _vp, _vp, _vp]),
    'wmh_median_filter_f64': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'wmh_component_filtering': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    'wmh_cl_from_ncdhw': (_i, [_vp, _vp, _i, _i, _sz, _vp]),
    'wmh_cl_to_ncdhw': (_i, [_vp, _vp, _i, _i, _sz, _vp]),
    'wmh_cl_conv3_first_packed_halfs': (_sz, [_i]),
    'wmh_cl_conv3_first_pack_weights': (_i, [_vp, _i, _vp]),
    'wmh_cl_conv3_first_num_partials': (_i, [_i, _i, _i]),
    'wmh_cl_conv3_first_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'wmh_cl_conv3_num_partials': (_i, [_i, _i, _i]),
    'wmh_cl_conv3_fwd': (_i, [_vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    'wmh_cl_conv3s2_num_partials': (_i, [_i, _i, _i]),
    'wmh_cl_conv3s2_fwd': (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    'wmh_cl_convt_packed_halfs': (_sz, [_i, _i]),
    'wmh_cl_convt_pack_weights': (_i, [_vp, _i, _i, _vp]),
    'wmh_cl_convt_fwd': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    'wmh_cl_seghead_softmax_accumulate': (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _ip, _i, _ip, _f, _vp]),
    'wmh_selftest_mfma_32x32x2': (_i, [_vp, _vp, _vp, _vp]),
}

_lib = None
_diag = None


class WMHError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "deepwmh_amd: %s not found. The HIP library is required (there is no CPU fallback); "
            "build it with `python -c \"import __graft_entry__ as g; g.build()\"` or "
            "`python deepwmh_amd/build.py`." % LIB_PATH)
    # torch first: its wheel bundles the HIP runtime it was built against, and the process must end up with ONE
    # runtime -- loaded after torch, this library binds to the copy torch already mapped (same SONAME)
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


# include/deepwmh_amd_diag.h: exported by libdeepwmh_amd_diag.so only
DIAG_SIGNATURES = {
    'wmh_debug_set_conv_stamps': (None, [_vp]),
    'wmh_conv3d_k3_wgrad_f16x2_stamp_count': (_sz, [_i, _i, _i, _i, _i, _i]),
    'wmh_conv3d_k3_wgrad_f16x2_stamped': (_i, [_vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp]),
}


def load_diag():
    """The diagnostic library as a SECOND handle beside the product library (bench.py's in-kernel clock leg, tools/): same ABI."""
    global _diag
    if _diag is None:
        if not os.path.exists(DIAG_LIB_PATH):
            raise ImportError('deepwmh_amd: %s not found (python deepwmh_amd/build.py builds it next to the product library)' % DIAG_LIB_PATH)
        import torch  # noqa: F401
        lib = C.CDLL(DIAG_LIB_PATH)
        for name, (res, args) in list(SIGNATURES.items()) + list(DIAG_SIGNATURES.items()):
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.wmh_diag_build() != 1:
            raise ImportError('%s was not compiled with -DWMH_DIAG' % DIAG_LIB_PATH)
        _diag = lib
    return _diag


def check(rc, what=''):
    if rc < 0:
        raise WMHError('%s failed (%d): %s' % (what, rc, load().wmh_last_error().decode()))
    return rc


def iarr(values):
    values = [int(v) for v in values]
    return (C.c_int * len(values))(*values)


def ptr(t):
    """Device/host pointer of a torch tensor (or None)."""
    if t is None:
        return None
    assert t.is_contiguous(), 'tensor must be contiguous'
    return C.c_void_p(t.data_ptr())


def current_stream():
    """torch's current HIP stream as the `void* stream` argument of the C ABI."""
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
