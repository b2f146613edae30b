"""ctypes binding of libtpgsr_hip.so (the C ABI declared in include/tpgsr_hip.h).

There is deliberately NO fallback: if the shared library is missing or a kernel call fails, the product path
raises.  The CPU oracle under oracle/ is test infrastructure and is never imported from here."""
import ctypes as C
import os

# torch bundles its own HIP runtime (torch/lib/libamdhip64.so, soname libamdhip64.so.7).  It MUST be the one already
# mapped when libtpgsr_hip.so (NEEDED libamdhip64.so.7) is dlopen'ed, otherwise the process ends up with two HIP
# runtimes and every call on a torch stream fails with hipErrorNoDevice.  Importing torch first guarantees that.
import torch  # noqa: F401  (load-order dependency, see above)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtpgsr_hip.so")

ACT_NONE, ACT_RELU, ACT_MISH, ACT_TANH, ACT_PRELU = 0, 1, 2, 3, 4
_ACT = {None: 0, "none": 0, "relu": 1, "mish": 2, "tanh": 3}

vp = C.c_void_p
ci = C.c_int
ll = C.c_longlong
cf = C.c_float


class ConvArgs(C.Structure):
    _fields_ = [("in_", vp), ("in2", vp), ("in_scale", vp), ("in_shift", vp), ("wt", vp), ("bias", vp), ("out", vp),
                ("bn_partial", vp),
                ("N", ci), ("H", ci), ("W", ci), ("Cin", ci),
                ("in_ld", ci), ("in_coff", ci), ("in2_ld", ci), ("in_act", ci), ("in_ps", ci),
                ("Cout", ci), ("KH", ci), ("KW", ci), ("pad_h", ci), ("pad_w", ci), ("OH", ci), ("OW", ci),
                ("out_ld", ci), ("out_coff", ci), ("out_act", ci), ("out_ps", ci),
                ("in_b", vp), ("cin_a", ci), ("in_b_ld", ci), ("in_dil_w", ci), ("wt_ld", ci), ("wt_coff", ci), ("stride_w", ci),
                ("terms", ci), ("kp", ci), ("wt_bf", vp), ("wt_bf_cin", ci), ("reserved0", ci),
                ("bnb_y", vp), ("bnb_mean", vp), ("bnb_rstd", vp), ("bnb_scale", vp), ("bnb_shift", vp), ("bnb_act", ci), ("bnb_store_dz", ci),
                ("fin_mode", ci), ("fin_accumulate", ci), ("fin_count", ll), ("fin_counter", vp), ("fin_gamma", vp), ("fin_beta", vp),
                ("fin_bias", vp), ("fin_scale", vp), ("fin_shift", vp), ("fin_mean", vp), ("fin_rstd", vp), ("fin_rm", vp), ("fin_rv", vp),
                ("fin_momentum", cf), ("fin_eps", cf), ("bn_row_tiles", ci), ("reserved1", ci), ("in2_scale", vp),
                ("sk_part", vp), ("sk_splits", ci), ("reserved2", ci)]


class WgradArgs(C.Structure):
    _fields_ = [("c", ConvArgs), ("dy", vp), ("dy_ld", ci), ("dy_coff", ci), ("dy_ps", ci), ("part", vp), ("dbpart", vp),
                ("zsplits", ci), ("reserved1", ci), ("dy_bf", vp)]


class GruWgradArgs(C.Structure):
    _fields_ = [("c", ConvArgs), ("dgi", vp), ("dghn", vp), ("h", vp), ("partC", vp), ("dbC", vp), ("partH", vp), ("dbH", vp),
                ("axis", ci), ("zsplits", ci)]


class BigruProjArgs(C.Structure):
    _fields_ = [("c", ConvArgs), ("w_hh", vp), ("b_hh", vp), ("h_out", vp), ("gates", vp), ("axis", ci), ("reserved", ci)]


class PackDesc(C.Structure):
    _fields_ = [("src", vp), ("dst_f", vp), ("dst_d", vp), ("Cout", ci), ("Cin", ci), ("KH", ci), ("KW", ci),
                ("kind", ci), ("f_ld", ci), ("f_coff", ci), ("wscale", cf), ("numel", ci), ("blk0", ci),
                ("src2", vp), ("src3", vp), ("d_ld", ci), ("cin_ld", ci)]


class ComposeBwdDesc(C.Structure):
    _fields_ = [("dWc", vp), ("dbc", vp), ("W1", vp), ("b1", vp), ("wih0", vp), ("wih1", vp), ("dW1", vp), ("db1", vp),
                ("dwih0", vp), ("dwih1", vp), ("dbih0", vp), ("dbih1", vp), ("Cin", ci), ("U", ci), ("G", ci), ("blk0", ci)]


class WgradReduceDesc(C.Structure):
    _fields_ = [("part", vp), ("dbpart", vp), ("dw", vp), ("db", vp), ("Z", ci), ("K", ci), ("Cin", ci), ("Cout", ci),
                ("KH", ci), ("KW", ci), ("layout", ci), ("accumulate", ci), ("gscale", cf), ("blk0", ci), ("cin_ld", ci),
                ("reserved", ci)]


class SplitDesc(C.Structure):
    _fields_ = [("src", vp), ("dst", vp), ("K", ci), ("N", ci), ("ld", ci), ("kp", ci), ("blk0", ci), ("cin", ci)]


class WgradBatchItem(C.Structure):
    _fields_ = [("w", WgradArgs), ("M", ci), ("K", ci), ("MB", ci), ("blk0", ci), ("nblk", ci), ("reserved0", ci), ("reserved1", ci), ("reserved2", ci)]


class ImageDesc(C.Structure):
    _fields_ = [("offset", ll), ("H", ci), ("W", ci), ("xb_off", ci), ("xk_off", ci), ("kx", ci), ("yb_off", ci), ("yk_off", ci), ("ky", ci)]


class BnDerive(C.Structure):
    """tpgsr_bn_derive: a BatchNorm finalized inside its first consumer's launch (csrc/bn_derive.h)"""
    _fields_ = [("rows", vp), ("nrows", ci), ("C", ci), ("count", ll), ("bias", vp), ("gamma", vp), ("beta", vp),
                ("running_mean", vp), ("running_var", vp), ("momentum", cf), ("eps", cf), ("scale", vp), ("shift", vp),
                ("save_mean", vp), ("save_rstd", vp), ("dgamma", vp), ("dbeta", vp), ("coef", vp), ("accumulate", ci), ("reserved", ci), ("flag", vp)]


class PlanArg(C.Union):
    """tpgsr_plan_arg: one launch argument of a native plan (pointer / integer / float)"""
    _fields_ = [("p", vp), ("i", ll), ("f", C.c_double)]


_SIGS = {
    "tpgsr_plan_create": (vp, []),
    "tpgsr_plan_destroy": (None, [vp]),
    "tpgsr_plan_size": (ci, [vp]),
    "tpgsr_plan_add_launch": (ci, [vp, C.c_char_p, C.POINTER(PlanArg), ci, ci]),
    "tpgsr_plan_add_fork": (ci, [vp]),
    "tpgsr_plan_add_join": (ci, [vp]),
    "tpgsr_plan_set_arg": (ci, [vp, ci, ci, C.POINTER(PlanArg)]),
    "tpgsr_plan_run": (ci, [vp, vp, vp]),
    "tpgsr_plan_add_edge": (ci, [vp, ci, ci]),
    "tpgsr_plan_run3": (ci, [vp, vp, vp, vp]),
    "tpgsr_stream_create": (vp, [C.POINTER(C.c_uint), ci]),
    "tpgsr_plan_set_mode": (None, [ci, ci, C.c_ulonglong, ci]),
    "tpgsr_plan_get_mode": (ci, []),
    "tpgsr_plan_fuzz_point": (ci, [vp]),
    "tpgsr_spin": (ci, [ci, ci, cf, ci, vp]),
    "tpgsr_plan_set_stamp": (ci, [ci]),
    "tpgsr_plan_stamp_epoch": (ci, [vp]),
    "tpgsr_plan_read_stamps": (ci, [vp, C.POINTER(cf), C.POINTER(C.c_longlong), ci]),
    "tpgsr_stream_destroy": (ci, [vp]),
    "tpgsr_pack_program": (ci, [vp, ci, ci, vp]),
    "tpgsr_mfma_probe": (ci, [vp, ci, ci, vp]),
    "tpgsr_copy": (ci, [vp, vp, ll, vp]),
    "tpgsr_zero": (ci, [vp, ll, vp]),
    "tpgsr_version": (ci, []),
    "tpgsr_sizeof": (ci, [ci]),
    "tpgsr_conv_fwd": (ci, [C.POINTER(ConvArgs), vp]),
    "tpgsr_bigru_proj_supported": (ci, [C.POINTER(BigruProjArgs)]),
    "tpgsr_bigru_proj_fwd": (ci, [C.POINTER(BigruProjArgs), vp]),
    "tpgsr_bigru_proj_set_enabled": (None, [ci]),
    "tpgsr_affine_act_bnd": (ci, [C.POINTER(BnDerive), vp, ll, ci, vp, vp]),
    "tpgsr_affine_act_pool_bnd": (ci, [C.POINTER(BnDerive), vp, ci, ci, ci, ci, ci, ci, vp, vp]),
    "tpgsr_bn_bwd_apply_bnd": (ci, [C.POINTER(BnDerive), vp, vp, vp, ll, vp, vp, ci, vp, vp]),
    "tpgsr_wgrad_splits": (ci, [ci, ci, ci]),
    "tpgsr_conv_wgrad": (ci, [C.POINTER(WgradArgs), vp]),
    "tpgsr_wgrad_halo_plan": (ci, [C.POINTER(ConvArgs), C.POINTER(ci), C.POINTER(C.c_longlong)]),
    "tpgsr_wgrad_reduce": (ci, [vp, vp, ci, ci, ci, ci, ci, ci, ci, vp, vp, ci, cf, vp]),
    "tpgsr_pack_blocks": (ci, [ci, ci, ci, ci, ci, ll]),
    "tpgsr_conv_wgrad_batch_prepare": (ci, [C.POINTER(WgradArgs), C.POINTER(WgradBatchItem)]),
    "tpgsr_conv_wgrad_batch": (ci, [vp, ci, ci, ci, ci, vp]),
    "tpgsr_wgrad_reduce_blocks2": (ci, [ci, ci, ci, ci, ci, ci, ci, ci]),
    "tpgsr_wgrad_reduce_program": (ci, [vp, ci, ci, vp]),
    "tpgsr_compose_bwd_blocks": (ci, [ci, ci, ci]),
    "tpgsr_compose_bwd_program": (ci, [vp, ci, ci, vp]),
    "tpgsr_pack_conv_weight": (ci, [vp, ci, ci, ci, ci, ci, cf, vp, vp, vp]),
    "tpgsr_pack_tail_weight": (ci, [vp, ci, ci, ci, vp, vp, vp]),
    "tpgsr_bn_finalize": (ci, [vp, ci, ci, ll, vp, vp, vp, vp, vp, cf, cf, ci, vp, vp, vp, vp, vp]),
    "tpgsr_bn_stats": (ci, [vp, ll, ci, ci, vp, ci, vp]),
    "tpgsr_bn_bwd_reduce": (ci, [vp, vp, vp, ll, ci, vp, vp, vp, vp, ci, vp, ci, vp]),
    "tpgsr_bn_bwd_finalize": (ci, [vp, ci, ci, ll, vp, vp, vp, vp, vp, ci, vp, vp]),
    "tpgsr_bn_bwd_apply": (ci, [vp, vp, vp, ll, ci, vp, vp, ci, vp, vp, vp]),
    "tpgsr_affine_act": (ci, [vp, ll, ci, vp, vp, ci, vp, vp]),
    "tpgsr_affine_act_pool": (ci, [vp, ci, ci, ci, ci, vp, vp, ci, ci, ci, vp, vp]),
    "tpgsr_affine_act_pool_bwd": (ci, [vp, vp, ci, ci, ci, ci, vp, vp, ci, ci, ci, vp, vp]),
    "tpgsr_prelu_fwd": (ci, [vp, vp, ll, vp, vp]),
    "tpgsr_prelu_bwd": (ci, [vp, vp, vp, vp, ll, vp, vp, ci, vp]),
    "tpgsr_add": (ci, [vp, vp, ll, vp, vp]),
    "tpgsr_act_bwd": (ci, [vp, vp, ll, ci, vp, vp]),
    "tpgsr_nchw_to_nhwc": (ci, [vp, ci, ci, ci, ci, vp, vp]),
    "tpgsr_nhwc_to_nchw": (ci, [vp, ci, ci, ci, ci, vp, vp]),
    "tpgsr_reduce_partials": (ci, [vp, ci, ci, vp, ci, vp]),
    "tpgsr_bigru_fwd": (ci, [vp, vp, vp, ci, ci, ci, ci, vp, vp, vp]),
    "tpgsr_bigru_bwd": (ci, [vp, vp, vp, vp, vp, ci, ci, ci, ci, vp, vp, vp]),
    "tpgsr_bigru_bwd2": (ci, [vp, vp, vp, vp, vp, ci, ci, ci, ci, vp, vp, vp]),
    "tpgsr_gru_gate_math_probe": (ci, [vp, vp, vp, ci, vp]),
    "tpgsr_gru_wgrad_splits": (ci, [ll]),
    "tpgsr_gru_wgrad": (ci, [C.POINTER(GruWgradArgs), vp]),
    "tpgsr_tps_grid_fwd": (ci, [vp, vp, vp, ci, ci, ci, vp, vp, vp]),
    "tpgsr_tps_grid_bwd": (ci, [vp, vp, vp, vp, ci, ci, ci, vp, vp]),
    "tpgsr_grid_sample_fwd": (ci, [vp, vp, ci, ci, ci, ci, ci, ci, ci, vp, vp]),
    "tpgsr_grid_sample_bwd": (ci, [vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, vp, vp, vp]),
    "tpgsr_strip_resample_fwd": (ci, [vp, vp, vp, ci, ci, ci, ci, ci, vp, vp]),
    "tpgsr_strip_resample_bwd": (ci, [vp, vp, vp, ci, vp, ci, ci, ci, ci, vp, vp]),
    "tpgsr_hsum": (ci, [vp, ci, ci, ci, ci, vp, ci, vp]),
    "tpgsr_bicubic_gray_fwd": (ci, [vp, ci, ci, ci, ci, ci, ci, vp, vp]),
    "tpgsr_bicubic_gray_bwd": (ci, [vp, ci, ci, ci, ci, ci, ci, vp, vp]),
    "tpgsr_pool2d_fwd": (ci, [vp, ci, ci, ci, ci, vp, vp, ci, ci, ci, ci, ci, ci, ci, vp, vp]),
    "tpgsr_pool2d_bwd": (ci, [vp, vp, ci, ci, ci, ci, vp, vp, ci, ci, ci, ci, ci, ci, ci, vp, vp]),
    "tpgsr_lstm_rec_gemm": (ci, [vp, vp, ll, vp, vp, ci, ci, ci, ci, vp, vp]),
    "tpgsr_lstm_step_fwd": (ci, [vp, vp, ci, vp, vp, vp, ci, ci, ci, ci, vp]),
    "tpgsr_lstm_seq_fwd": (ci, [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, vp]),
    "tpgsr_lstm_seq_hx_bytes": (C.c_longlong, []),
    "tpgsr_lstm_seq_probe": (ci, [vp, vp]),
    "tpgsr_lstm_seq_fwdg": (ci, [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, vp]),
    "tpgsr_lstm_seq_hg_bytes": (C.c_longlong, []),
    "tpgsr_lstm_seq_bwd": (ci, [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, vp]),
    "tpgsr_lstm_seq_px_bytes": (C.c_longlong, []),
    "tpgsr_lstm_seq_bwdg": (ci, [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, vp]),
    "tpgsr_lstm_seq_pg_bytes": (C.c_longlong, []),
    "tpgsr_lstm_wfrag_bytes": (C.c_longlong, []),
    "tpgsr_lstm_wfrag": (ci, [vp, vp, ci, vp]),
    "tpgsr_lstm_stepx_fwd": (ci, [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, vp]),
    "tpgsr_lstm_step_bwd": (ci, [vp, vp, vp, vp, ci, vp, ci, ci, ci, ci, vp]),
    "tpgsr_softmax_prior_fwd": (ci, [vp, vp, ci, ci, ci, ci, vp, vp, vp, ci, vp]),
    "tpgsr_semantic_loss_finalize": (ci, [vp, ci, ll, cf, vp, vp]),
    "tpgsr_softmax_prior_bwd": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, cf, vp, ci, vp]),
    "tpgsr_ctc_loss": (ci, [vp, ci, ci, vp, vp, vp, vp, ci, ci, ci, ci, cf, vp, vp, ci, ci, vp]),
    "tpgsr_tail_shiftsum_tanh": (ci, [vp, vp, ci, ci, ci, ci, ci, vp, vp]),
    "tpgsr_shiftsum_nhwc": (ci, [vp, ci, ci, ci, ci, ci, vp, vp]),
    "tpgsr_tail_bwd": (ci, [vp, vp, ci, ci, ci, ci, ci, vp, vp, ci, vp]),
    "tpgsr_tail_bwd_blocks": (ci, [ci, ci, ci, ci, ci]),
    "tpgsr_image_loss_fwd": (ci, [vp, vp, ci, ci, ci, ci, ci, vp, ci, vp]),
    "tpgsr_image_loss_finalize": (ci, [vp, ci, ll, ll, cf, cf, vp, vp]),
    "tpgsr_image_loss_bwd": (ci, [vp, vp, vp, ci, ci, ci, ci, ci, cf, cf, vp, vp]),
    "tpgsr_sumsq_partial": (ci, [vp, ll, vp, ci, vp]),
    "tpgsr_clip_coef": (ci, [vp, ci, cf, vp, vp, vp]),
    "tpgsr_adam_step": (ci, [vp, vp, vp, vp, ll, vp, cf, cf, cf, cf, vp, vp]),
    "tpgsr_step_inc": (ci, [vp, vp]),
    "tpgsr_clip_coef_steps": (ci, [vp, ci, cf, vp, vp, vp, ci, vp]),
    "tpgsr_scale_": (ci, [vp, ll, vp, vp]),
    "tpgsr_im2col3x3_c1": (ci, [vp, ci, ci, ci, vp, vp]),
    "tpgsr_col2im3x3_c1": (ci, [vp, ci, ci, ci, vp, vp]),
    "tpgsr_pad_channels": (ci, [vp, ll, ci, ci, vp, vp]),
    "tpgsr_semantic_loss_fwd": (ci, [vp, vp, ll, vp, ci, vp]),
    "tpgsr_semantic_loss_bwd": (ci, [vp, vp, vp, ll, vp, vp]),
    "tpgsr_copy_strided": (ci, [vp, ci, ci, vp, ci, ci, ll, ci, ci, vp]),
    "tpgsr_resize_nearest_fwd": (ci, [vp, ci, ci, ci, ci, ci, vp, vp]),
    "tpgsr_resize_nearest_bwd": (ci, [vp, ci, ci, ci, ci, ci, vp, vp]),
    "tpgsr_resize_bilinear_fwd": (ci, [vp, ci, ci, ci, ci, ci, ci, vp, vp]),
    "tpgsr_resize_bilinear_bwd": (ci, [vp, ci, ci, ci, ci, ci, ci, vp, vp]),
    "tpgsr_dilate2d": (ci, [vp, ci, ci, ci, ci, ci, ci, vp, vp]),
    "tpgsr_subsample2d": (ci, [vp, ci, ci, ci, ci, ci, ci, vp, vp]),
    "tpgsr_hreduce": (ci, [vp, ci, ci, ci, ci, cf, vp, vp]),
    "tpgsr_hbroadcast": (ci, [vp, ci, ci, ci, ci, cf, vp, vp]),
    "tpgsr_resample_ksize": (ci, [ci, ci]),
    "tpgsr_resample_coeffs": (ci, [ci, ci, vp, vp]),
    "tpgsr_resize_normalize": (ci, [vp, vp, vp, ci, ci, ci, ci, ci, vp, vp, vp, vp]),
    "tpgsr_ctc_greedy_decode": (ci, [vp, ci, ci, ci, vp, vp, vp]),
    "tpgsr_psnr": (ci, [vp, vp, ci, ci, ci, ci, vp, ci, vp, vp]),
    "tpgsr_ssim": (ci, [vp, vp, vp, ci, ci, ci, ci, ci, vp, ci, vp, vp]),
    "tpgsr_ssim_bwd": (ci, [vp, vp, vp, ci, ci, ci, ci, ci, vp, vp, cf, vp, ci, vp]),
    "tpgsr_split_bf_blocks": (ci, [ci, ci]),
    "tpgsr_split_bf_program": (ci, [vp, ci, ci, vp]),
    "tpgsr_tr_probe": (ci, [vp, vp]),
    "tpgsr_bicubic_resize": (ci, [vp, ci, ci, ci, ci, ci, ci, ci, cf, cf, vp, vp]),
    "tpgsr_aster_attention": (ci, [vp, vp, vp, vp, vp, ci, ci, ci, ci, vp, vp, vp]),
    "tpgsr_embed_concat": (ci, [vp, vp, ci, ci, vp, ci, ci, vp, vp]),
    "tpgsr_gru_cell": (ci, [vp, vp, vp, ci, ci, vp, vp]),
    "tpgsr_softmax_max": (ci, [vp, ci, ci, vp, vp, ci, ci, vp, vp]),
    "tpgsr_halo_trace": (ci, [vp]),
    "tpgsr_halo_capacity": (ci, [C.POINTER(ConvArgs)]),
    "tpgsr_conv_bn_row_tiles": (ci, [C.POINTER(ConvArgs)]),
    "tpgsr_conv_in2_scale_ok": (ci, [C.POINTER(ConvArgs)]),
    "tpgsr_conv_splitk_plan": (ci, [C.POINTER(ConvArgs), C.POINTER(C.c_longlong)]),
    "tpgsr_splitk_set_enabled": (None, [ci]),
    "tpgsr_halo_set_colmajor_min_bytes": (None, [C.c_longlong]),
    "tpgsr_halo_set_min_taps": (None, [ci]),
    "tpgsr_halo_set_ne9": (None, [ci]),
    "tpgsr_halo3_set_enabled": (None, [ci]),
    "tpgsr_wgrad3_set_enabled": (None, [ci]),
    "tpgsr_panel_set_enabled": (None, [ci]),
    "tpgsr_panel_set_min_m": (None, [C.c_longlong]),
    "tpgsr_panel_set_k192": (None, [ci]),
    "tpgsr_gru_set_prefetch": (None, [ci]),
    "tpgsr_mfma_bf16_probe": (ci, [vp, vp, vp, vp, ci, vp]),
}

EXPORTED_SYMBOLS = sorted(list(_SIGS.keys()) + ["tpgsr_last_error"])

_lib = None


class TpgsrKernelError(RuntimeError):
    pass


def load():
    """Load the shared library (building is __graft_entry__.build()'s / tpgsr_amd.build's job)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TpgsrKernelError(
            f"{LIB_PATH} is missing: build it with `python -m tpgsr_amd.build` (hipcc --offload-arch=gfx950). "
            "tpgsr_amd has no CPU / PyTorch fallback by design.")
    lib = C.CDLL(LIB_PATH)
    lib.tpgsr_last_error.restype = C.c_char_p
    lib.tpgsr_last_error.argtypes = []
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    for which, st in enumerate((ConvArgs, WgradArgs, PackDesc, WgradReduceDesc, ComposeBwdDesc, SplitDesc, ImageDesc, GruWgradArgs, WgradBatchItem, BnDerive, BigruProjArgs)):
        if lib.tpgsr_sizeof(which) != C.sizeof(st):
            raise TpgsrKernelError(f"ABI mismatch: {st.__name__} is {C.sizeof(st)} bytes in the binding, "
                                   f"{lib.tpgsr_sizeof(which)} in {LIB_PATH}: rebuild (python -m tpgsr_amd.build)")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().tpgsr_last_error().decode(errors="replace")
        raise TpgsrKernelError(f"{what or 'tpgsr kernel'} failed (rc={rc}): {msg}")


def act_code(name):
    return _ACT[name] if not isinstance(name, int) else name
