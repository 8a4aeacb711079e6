"""ctypes binding of libasrk.so (the C ABI declared in include/asrk.h).

The product path has NO CPU / eager-PyTorch fallback: if the HIP library is missing or a GPU op
is asked to run without a GPU this module raises.  (The CPU oracle under ``oracle/`` is test
infrastructure only and is never imported from here.)
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libasrk.so")

c_int, c_i64, c_f32, c_vp, c_sz = (ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p,
                                   ctypes.c_size_t)
c_f64 = ctypes.c_double

# name -> (restype, argtypes); every symbol include/asrk.h declares
SIGNATURES = {
    "asrk_version": (c_int, []),
    "asrk_strerror": (ctypes.c_char_p, [c_int]),
    "asrk_init": (c_int, [c_int]),
    "asrk_profile_enable": (None, [c_int]),
    "asrk_profile_families": (None, [ctypes.c_uint]),
    "asrk_profile_reset": (None, []),
    "asrk_profile_get": (c_int, [c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_i64)]),
    "asrk_profile_get_work": (c_int, [c_int, ctypes.POINTER(ctypes.c_double)]),
    "asrk_gemm_f32": (c_int, [c_int, c_int, c_int, c_int, c_int, c_f32, c_vp, c_int, c_vp, c_int,
                              c_f32, c_vp, c_int, c_vp, c_vp, c_int, c_int, c_vp, c_sz, c_vp]),
    "asrk_gemm_ws_bytes": (c_sz, [c_int, c_int, c_int, c_int]),
    "asrk_gemm_takes_split": (c_int, [c_int, c_int, c_int, c_int]),
    "asrk_split_panel_bytes": (ctypes.c_size_t, [c_int, c_int, c_int]),
    "asrk_split_panel_f32": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_vp, c_int, c_vp]),
    "asrk_gemm_panels_f32": (c_int, [c_int, c_int, c_int, c_f32, c_vp, c_int, c_int, c_int, c_int,
                                     c_vp, c_int, c_int, c_int, c_int, c_f32, c_vp, c_int, c_vp, c_vp, c_int, c_vp]),
    "asrk_copy3d_f32": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_i64, c_i64, c_i64, c_i64, c_int,
                                c_vp]),
    "asrk_colsum_f32": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_int, c_vp]),
    "asrk_tanh_fwd_f32": (c_int, [c_vp, c_vp, c_i64, c_vp]),
    "asrk_tanh_bwd_f32": (c_int, [c_vp, c_vp, c_vp, c_i64, c_vp]),
    "asrk_log_softmax_fwd_f32": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "asrk_log_softmax_bwd_f32": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "asrk_cross_entropy_fwd_f32": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_int, c_vp, c_vp, c_vp]),
    "asrk_cross_entropy_bwd_f32": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_int, c_vp, c_vp, c_vp,
                                           c_vp]),
    "asrk_ctc_prefix_score_f32": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int,
                                          c_int, c_int, c_int, c_f32, c_vp]),
    "asrk_ctc_prefix_score_multi_f32": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int,
                                                c_int, c_int, c_int, c_int, c_int, c_f32, c_vp]),
    "asrk_ctc_prefix_beam_ws_bytes": (c_sz, [c_int, c_int]),
    "asrk_ctc_prefix_beam_ws_offsets": (c_int, [c_int, c_int, c_int] + [ctypes.POINTER(c_i64)] * 6),
    "asrk_ctc_prefix_beam_f32": (c_int, [c_vp, c_int, c_int, c_vp, c_int, c_int, c_vp, c_f32, c_int, c_int, c_int,
                                         c_int, c_int, c_vp, c_sz, c_vp]),
    "asrk_fbank_frames_f32": (c_int, [c_vp, c_i64, c_vp, c_vp, c_int, c_int, c_int, c_int, c_f32, c_int,
                                      c_vp]),
    "asrk_power_spectrum_f32": (c_int, [c_vp, c_vp, c_i64, c_int, c_vp]),
    "asrk_log_floor_f32": (c_int, [c_vp, c_i64, c_f32, c_vp]),
    "asrk_delta_f32": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "asrk_cmvn_f32": (c_int, [c_vp, c_vp, c_int, c_int, c_f32, c_vp]),
    "asrk_transpose_f32": (c_int, [c_vp, c_vp, c_int, c_int, c_vp]),
    "asrk_fbank_frames_batch_f32": (c_int, [c_vp, c_int, c_i64, c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_int, c_int,
                                            c_int, c_f32, c_f32, c_int, c_vp]),
    "asrk_delta_cmvn_batch_f32": (c_int, [c_vp, c_vp, c_int, c_int, c_vp, c_int, c_int, c_int, c_f32, c_vp, c_int,
                                          c_vp]),
    "asrk_fbank_logmel_batch_f32": (c_int, [c_vp, c_int, c_i64, c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp,
                                            c_int, c_int, c_vp, c_int, c_int, c_int, c_f32, c_f32, c_int, c_f32, c_vp]),
    "asrk_lstm_ws_bytes": (c_sz, []),
    "asrk_lstm_xchg_bytes": (c_sz, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "asrk_lstm_plan_workgroups": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "asrk_lstm_rec_fwd_f32": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int,
                                      c_vp, c_int, c_vp, c_int, c_vp]),
    "asrk_lstm_rec_bwd_f32": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int,
                                      c_vp, c_int, c_vp, c_vp, c_int, c_vp]),
    "asrk_lstm_rec_fwd_pyr_f32": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int,
                                          c_vp, c_int, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "asrk_lstm_rec_fwd_len_f32": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int,
                                          c_vp, c_int, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "asrk_lstm_rec_bwd_pyr_f32": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int,
                                          c_vp, c_int, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "asrk_lstm_plan_is_bf": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "asrk_lstm_rec_fwd_pyr_panel_f32": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int,
                                                c_vp, c_int, c_vp, c_vp, c_int, c_int, c_vp, c_int, c_vp]),
    "asrk_lstm_rec_bwd_pyr_panel_f32": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int,
                                                c_vp, c_int, c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_int, c_vp]),
    "asrk_gru_rec_fwd_f32": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int,
                                     c_vp, c_int, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "asrk_gru_rec_bwd_f32": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int,
                                     c_vp, c_int, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "asrk_lstm_check_error": (c_int, [c_vp, c_vp]),
    "asrk_loc_conv_fwd_f32": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "asrk_loc_conv_bwd_f32": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int,
                                      c_vp]),
    "asrk_attn_energy_fwd_f32": (c_int, [c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                         c_int, c_int, c_int, c_int, c_int, c_f32, c_vp]),
    "asrk_attn_energy_bwd_f32": (c_int, [c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                         c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int,
                                         c_int, c_f32, c_vp]),
    "asrk_attn_context_fwd_f32": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_i64, c_vp]),
    "asrk_attn_context_bwd_f32": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_i64, c_vp]),
    "asrk_lstm_cell_fwd_f32": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_vp]),
    "asrk_lstm_cell_bwd_f32": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_vp]),
    "asrk_gru_cell_fwd_f32": (c_int, [c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_i64, c_int, c_int, c_vp]),
    "asrk_gru_cell_bwd_f32": (c_int, [c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64,
                                      c_int, c_int, c_vp]),
    "asrk_embedding_fwd_f32": (c_int, [c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_vp]),
    "asrk_embedding_bwd_f32": (c_int, [c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_vp]),
    "asrk_layer_norm_fwd_f32": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_f32, c_vp]),
    "asrk_layer_norm_bwd_f32": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int,
                                        c_vp]),
    "asrk_dropout_f32": (c_int, [c_vp, c_vp, c_i64, c_f32, ctypes.c_uint64, ctypes.c_uint64, c_vp]),
    "asrk_adadelta_step_f32": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_f64, c_f64, c_f64, c_vp, c_vp]),
    "asrk_adadelta_multi_f32": (c_int, [c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_f64, c_f64, c_f64, c_vp, c_vp]),
    "asrk_adam_multi_f32": (c_int, [c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_f64, c_f64, c_f64, c_f64, c_i64,
                                    c_vp, c_vp]),
    "asrk_adam_step_f32": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_f64, c_f64, c_f64, c_f64, c_i64, c_vp,
                                   c_vp]),
    "asrk_token_crop_i64": (c_int, [c_vp, c_i64, c_int, c_int, c_i64, c_i64, c_int, c_vp, c_i64, c_vp, c_vp]),
    "asrk_edit_distance_i64": (c_int, [c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_int, c_int, c_vp, c_vp]),
    "asrk_grad_norm_ws_bytes": (c_sz, [c_int, c_vp]),
    "asrk_grad_norm_multi_f32": (c_int, [c_int, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "asrk_fill_f32": (c_int, [c_vp, c_i64, c_f32, c_vp]),
    "asrk_topk_f32": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp]),
    "asrk_beam_select_f32": (c_int, [c_vp] * 4 + [c_int] * 6 + [c_vp] * 19 + [c_vp]),
    "asrk_gather_rows_multi_f32": (c_int, [c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp]),
    "asrk_conv_out_size": (c_int, [c_int, c_int, c_int, c_int]),
    "asrk_im2col_f32": (c_int, [c_vp, c_vp] + [c_int] * 10 + [c_i64] * 4 + [c_vp]),
    "asrk_im2col_ld_f32": (c_int, [c_vp, c_vp] + [c_int] * 11 + [c_i64] * 4 + [c_vp]),
    "asrk_col2im_f32": (c_int, [c_vp, c_vp] + [c_int] * 10 + [c_i64] * 4 + [c_vp]),
    "asrk_im2col_cl_f32": (c_int, [c_vp, c_vp] + [c_int] * 10 + [c_i64] * 4 + [c_vp]),
    "asrk_col2im_cl_f32": (c_int, [c_vp, c_vp] + [c_int] * 10 + [c_i64] * 4 + [c_vp]),
    "asrk_conv_weight_reorder_f32": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "asrk_conv3x3_supported": (c_int, [c_int] * 4),
    "asrk_conv3x3_weight_f32": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "asrk_conv3x3_f32": (c_int, [c_vp] * 5 + [c_int] * 6 + [c_vp]),
    "asrk_conv3x3_wgrad_ws_bytes": (c_sz, [c_int] * 5),
    "asrk_conv3x3_wgrad_f32": (c_int, [c_vp] * 5 + [c_int] * 5 + [c_vp, c_sz, c_vp]),
    "asrk_conv3x3_first_supported": (c_int, [c_int] * 4),
    "asrk_conv3x3_first_f32": (c_int, [c_vp] * 4 + [c_int] * 5 + [c_i64] * 4 + [c_int, c_vp]),
    "asrk_conv3x3_first_wgrad_ws_bytes": (c_sz, [c_int] * 5),
    "asrk_conv3x3_first_wgrad_f32": (c_int, [c_vp] * 5 + [c_int] * 5 + [c_i64] * 4 + [c_vp, c_sz, c_vp]),
    "asrk_relu_fwd_f32": (c_int, [c_vp, c_i64, c_vp]),
    "asrk_relu_bwd_f32": (c_int, [c_vp, c_vp, c_vp, c_i64, c_vp]),
    "asrk_maxpool2x2_fwd_f32": (c_int, [c_vp, c_vp, c_vp] + [c_int] * 4 + [c_i64] * 4 + [c_vp]),
    "asrk_maxpool2x2_bwd_f32": (c_int, [c_vp, c_vp, c_vp] + [c_int] * 4 + [c_i64] * 4 + [c_vp]),
    "asrk_speller_plan": (c_int, [c_vp, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "asrk_speller_fwd_f32": (c_int, [c_vp, c_vp]),
    "asrk_speller_bwd_f32": (c_int, [c_vp, c_vp, c_vp]),
    "asrk_speller_step_f32": (c_int, [c_vp, c_int, c_vp, c_i64, c_vp, c_vp]),
    "asrk_speller_dvalue_f32": (c_int, [c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_int, c_int, c_int,
                                        c_int, c_vp]),
    "asrk_joint_score_f32": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_f64, c_f64,
                                     c_f64, c_vp]),
    "asrk_lstm_cell_fused_f32": (c_int, [c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int,
                                         c_int, c_vp]),
    "asrk_transpose_ld_f32": (c_int, [c_vp, c_i64, c_vp, c_i64, c_int, c_int, c_vp]),
    "asrk_flac_info": (c_int, [ctypes.c_char_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int),
                               ctypes.POINTER(c_i64), c_vp]),
    "asrk_flac_decode_i32": (c_int, [ctypes.c_char_p, c_vp, c_i64, ctypes.POINTER(c_i64)]),
    "asrk_ctc_loss_fwd_f32": (c_int, [c_vp, c_i64, c_i64, c_int, c_int, c_int, c_vp, c_i64, c_int,
                                      c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "asrk_ctc_loss_bwd_f32": (c_int, [c_vp, c_i64, c_i64, c_int, c_int, c_int, c_vp, c_i64, c_int,
                                      c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64,
                                      c_i64, c_vp]),
}

_lib = None


class AsrkError(RuntimeError):
    pass


def load():
    """Load libasrk.so and bind every declared symbol. Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AsrkError(
            "libasrk.so not found at %s - build it with `python __graft_entry__.py` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path." % LIB_PATH)
    # PyTorch supplies the device memory the library works on, so ITS HIP runtime must be the one libasrk.so binds to:
    # torch is imported first (its bundled libamdhip64 is then already in the process and satisfies libasrk.so's
    # dependency).  Loaded the other way round, libasrk.so pulls /opt/rocm's runtime, torch later brings its own, and
    # launches from this library fail with hipErrorNoDevice (seen with build() followed by smoke() in one process).
    try:
        import torch  # noqa: F401
    except ImportError:        # a C-ABI-only consumer: the system HIP runtime alone
        pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def strerror(rc):
    return load().asrk_strerror(int(rc)).decode()


def check(rc, what=""):
    if rc != 0:
        raise AsrkError("%s failed: %s (rc=%d)" % (what or "asrk call", strerror(rc), rc))
