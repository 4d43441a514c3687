"""ctypes binding of librangeldm_hip.so (include/rangeldm_hip.h).  The product path has NO fallback: if the HIP
library is missing or a call fails, a RuntimeError carrying rldm_last_error() is raised."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RLDM_LIB selects another build of the same library (tools/: the -DRLDM_ABLATE timeline build); never a fallback
LIB_PATH = os.environ.get("RLDM_LIB") or os.path.join(_HERE, "librangeldm_hip.so")
RLDM_MAX_LEVELS = 8


class UNetConfigC(C.Structure):
    _fields_ = [("sample_w", C.c_int32), ("sample_h", C.c_int32), ("in_channels", C.c_int32),
                ("out_channels", C.c_int32), ("layers_per_block", C.c_int32), ("num_levels", C.c_int32),
                ("block_out_channels", C.c_int32 * RLDM_MAX_LEVELS), ("down_attn", C.c_int32 * RLDM_MAX_LEVELS),
                ("up_attn", C.c_int32 * RLDM_MAX_LEVELS), ("attention_head_dim", C.c_int32),
                ("norm_num_groups", C.c_int32), ("norm_eps", C.c_float), ("mid_attention", C.c_int32),
                ("flip_sin_to_cos", C.c_int32), ("freq_shift", C.c_int32)]


class VAEConfigC(C.Structure):
    _fields_ = [("in_channels", C.c_int32), ("out_channels", C.c_int32), ("ch", C.c_int32), ("num_levels", C.c_int32),
                ("ch_mult", C.c_int32 * RLDM_MAX_LEVELS), ("num_res_blocks", C.c_int32), ("z_channels", C.c_int32),
                ("double_z", C.c_int32), ("norm_num_groups", C.c_int32), ("norm_eps", C.c_float),
                ("scaling_factor", C.c_float)]


class SamplerConfigC(C.Structure):
    _fields_ = [("batch", C.c_int32), ("num_steps", C.c_int32), ("mode", C.c_int32), ("pos_encoding", C.c_int32),
                ("cond_channels", C.c_int32), ("coef", C.POINTER(C.c_float)), ("timesteps", C.POINTER(C.c_int64)),
                ("plan_flags", C.c_int32), ("prediction_type", C.c_int32)]


class ConvDescC(C.Structure):
    _fields_ = [("B", C.c_int32), ("Cin0", C.c_int32), ("Cin1", C.c_int32), ("Win", C.c_int32), ("Hin", C.c_int32),
                ("Cout", C.c_int32), ("ksize", C.c_int32), ("stride", C.c_int32), ("pad_mode", C.c_int32),
                ("upsample", C.c_int32), ("gn", C.c_int32), ("silu", C.c_int32), ("eps", C.c_float)]


class LidarConfigC(C.Structure):
    _fields_ = [("beams", C.c_int32), ("width", C.c_int32), ("mode", C.c_int32), ("mean", C.c_float), ("std", C.c_float),
                ("range_fill", C.c_float), ("intensity_fill", C.c_float), ("grid", C.c_int32 * 3),
                ("pc_range", C.c_float * 6), ("normalize_volume_densities", C.c_int32)]


class TrainConvDescC(C.Structure):
    _fields_ = [("B", C.c_int32), ("Win", C.c_int32), ("Hin", C.c_int32), ("Cin", C.c_int32), ("N", C.c_int32),
                ("taps", C.c_int32), ("stride", C.c_int32), ("mode", C.c_int32)]


class TrainFuseC(C.Structure):
    """rldm_train_fuse (include/rangeldm_hip.h): what a fused conv / weight-gradient launch folds in."""
    _fields_ = [("x1", C.c_void_p), ("C0", C.c_int32), ("cs0", C.c_void_p), ("cs1", C.c_void_p), ("gamma", C.c_void_p),
                ("beta", C.c_void_p), ("silu", C.c_int32), ("groups", C.c_int32), ("eps", C.c_float), ("cs_out", C.c_void_p),
                ("g0", C.c_void_p), ("g1", C.c_void_p), ("G0", C.c_int32), ("gcs0", C.c_void_p), ("gcs1", C.c_void_p),
                ("ggamma", C.c_void_p), ("gbeta", C.c_void_p), ("gsilu", C.c_int32), ("ggroups", C.c_int32), ("geps", C.c_float),
                ("gs_out", C.c_void_p)]


class PackDescC(C.Structure):
    _fields_ = [("first", C.c_int64), ("param_offset", C.c_int64), ("w_forward", C.c_void_p), ("w_transposed", C.c_void_p),
                ("N", C.c_int32), ("Cin", C.c_int32), ("taps", C.c_int32), ("pad_", C.c_int32)]


class HyperConfigC(C.Structure):
    _fields_ = [("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("ema_max_decay", C.c_float),
                ("ema_inv_gamma", C.c_float), ("ema_power", C.c_float), ("lr_warmup_steps", C.c_int64), ("total_steps", C.c_int64)]


class AdamWConfigC(C.Structure):
    _fields_ = [("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("weight_decay", C.c_float), ("max_grad_norm", C.c_float), ("ema_decay", C.c_float), ("step", C.c_int32)]


_P = C.c_void_p
# every symbol declared in include/rangeldm_hip.h: name -> (restype, argtypes)
PROTOTYPES = {
    "rldm_last_error": (C.c_char_p, []),
    "rldm_device_info": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]),
    "rldm_unet_create": (C.c_int, [C.POINTER(UNetConfigC), C.POINTER(_P)]),
    "rldm_unet_destroy": (None, [_P]),
    "rldm_unet_set_param": (C.c_int, [_P, C.c_char_p, _P, C.c_int64]),
    "rldm_unet_finalize": (C.c_int, [_P]),
    "rldm_unet_forward": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, _P]),
    "rldm_vae_create": (C.c_int, [C.POINTER(VAEConfigC), C.POINTER(_P)]),
    "rldm_vae_destroy": (None, [_P]),
    "rldm_vae_set_param": (C.c_int, [_P, C.c_char_p, _P, C.c_int64]),
    "rldm_vae_finalize": (C.c_int, [_P]),
    "rldm_vae_decode": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "rldm_vae_encode": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "rldm_diag_gaussian_sample": (C.c_int, [_P, _P, C.c_float, C.c_int, C.c_int, C.c_int, _P, _P]),
    "rldm_sched_ddim_step": (C.c_int, [C.POINTER(C.c_float), _P, _P, _P, _P, C.c_int64, _P]),
    "rldm_sched_ddpm_step": (C.c_int, [C.POINTER(C.c_float), _P, _P, _P, _P, C.c_int64, _P]),
    "rldm_sched_step": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_float), _P, _P, _P, _P, C.c_int64, _P]),
    "rldm_sched_add_noise": (C.c_int, [_P, _P, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int64, _P, _P]),
    "rldm_sampler_create": (C.c_int, [_P, _P, C.POINTER(SamplerConfigC), C.POINTER(_P)]),
    "rldm_sampler_destroy": (None, [_P]),
    "rldm_sample": (C.c_int, [_P, _P, _P, _P, _P, _P, _P]),
    "rldm_sampler_status": (C.c_int, [_P]),
    "rldm_unet_set_plan_flags": (C.c_int, [_P, C.c_int]),
    "rldm_debug_inject_trunk_error": (C.c_int, [_P, C.c_int]),
    "rldm_comm_bind": (C.c_int, []),
    "rldm_comm_unique_id": (C.c_int, [_P, C.c_size_t]),
    "rldm_comm_create": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(_P)]),
    "rldm_comm_destroy": (None, [_P]),
    "rldm_comm_info": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_size_t]),
    "rldm_allgather_images": (C.c_int, [_P, _P, _P, C.c_int64, _P]),
    "rldm_allreduce_grads": (C.c_int, [_P, _P, C.c_int64, C.c_int, _P]),
    "rldm_lidar_create": (C.c_int, [C.POINTER(LidarConfigC), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(_P)]),
    "rldm_lidar_destroy": (None, [_P]),
    "rldm_lidar_to_points": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "rldm_lidar_to_voxel": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "rldm_lidar_filter_points": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_float, _P, _P, _P]),
    "rldm_render_u8": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "rldm_lidar_project": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_float, _P, _P, _P, _P]),
    "rldm_bev_histogram": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_float, C.c_int, C.c_float, C.c_float, _P, _P]),
    "rldm_hist_jsd": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.POINTER(C.c_double), _P]),
    "rldm_hist_spectral_sq": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "rldm_hist_mmd": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_float, C.POINTER(C.c_double), _P]),
    "rldm_train_conv": (C.c_int, [C.POINTER(TrainConvDescC), _P, _P, _P, _P, C.c_int, _P, _P, C.c_int, _P]),
    "rldm_train_conv_splits": (C.c_int, [C.POINTER(TrainConvDescC), C.c_int]),
    "rldm_train_wgrad": (C.c_int, [C.POINTER(TrainConvDescC), _P, _P, _P, _P]),
    "rldm_train_wgrad_bias": (C.c_int, [C.POINTER(TrainConvDescC), _P, _P, _P, _P, C.c_int, C.c_int, _P, _P]),
    "rldm_train_colsum": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P, _P]),
    "rldm_train_gn_forward": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _P, _P, C.c_int, _P, _P, _P]),
    "rldm_train_gn_backward": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _P, _P, C.c_int,
                                         _P, _P, _P]),
    "rldm_train_conv_fused_ok": (C.c_int, [C.POINTER(TrainConvDescC), C.POINTER(TrainFuseC), C.c_int]),
    "rldm_train_conv_fused": (C.c_int, [C.POINTER(TrainConvDescC), C.POINTER(TrainFuseC), _P, _P, _P, _P, C.c_int, _P, _P, C.c_int, _P]),
    "rldm_train_wgrad_fused_ok": (C.c_int, [C.POINTER(TrainConvDescC), C.POINTER(TrainFuseC)]),
    "rldm_train_wgrad_fused": (C.c_int, [C.POINTER(TrainConvDescC), C.POINTER(TrainFuseC), _P, _P, _P, _P, C.c_int, C.c_int, _P, _P]),
    "rldm_train_defer_reduce": (C.c_int, [C.c_int]),
    "rldm_train_flush_reduce": (C.c_int, []),
    "rldm_calibrate": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), _P]),
    "rldm_calib_clock_stamp": (C.c_int, [_P, _P]),
    "rldm_train_wgrad_group": (C.c_int, [C.c_int]),
    "rldm_train_wgrad_group_flush": (C.c_int, []),
    "rldm_train_wgrad_group_pending": (C.c_int, []),
    "rldm_train_chan_stats": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "rldm_train_gn_backward_apply": (C.c_int, [_P, _P, _P, C.c_int, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _P, _P,
                                               _P, C.c_int, _P, C.c_int, _P, _P, _P]),
    "rldm_train_linear_rows": (C.c_int, [_P, C.c_int, _P, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "rldm_train_linear_rows_wgrad": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "rldm_train_attention_forward": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "rldm_train_attention_backward": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P]),
    "rldm_train_attention_qkv_forward": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "rldm_train_attention_qkv_backward": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "rldm_train_add": (C.c_int, [_P, _P, _P, C.c_int64, _P]),
    "rldm_train_copy_channels": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, _P]),
    "rldm_train_sum2x2": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "rldm_train_silu": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int, C.c_int, _P]),
    "rldm_train_timestep_embedding": (C.c_int, [_P, C.c_int, C.c_int, _P, _P]),
    "rldm_train_pack_input": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "rldm_train_unpack_output": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "rldm_train_mse": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "rldm_train_sqnorm": (C.c_int, [_P, C.c_int64, _P, _P]),
    "rldm_train_adamw": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int64, C.POINTER(AdamWConfigC), _P]),
    "rldm_train_adamw_dyn": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int64, C.POINTER(AdamWConfigC), _P, C.c_int, _P]),
    "rldm_train_hyper_step": (C.c_int, [_P, C.POINTER(HyperConfigC), _P, _P]),
    "rldm_train_pack_weights": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "rldm_train_pack_weights_all": (C.c_int, [_P, _P, C.c_int, C.c_int64, _P]),
    "rldm_train_pack_weights_tiled": (C.c_int, [_P, _P, C.c_int, C.c_int64, _P]),
    "rldm_unet_flops": (C.c_double, [_P, C.c_int]),
    "rldm_vae_decode_flops": (C.c_double, [_P, C.c_int, C.c_int, C.c_int]),
    "rldm_unet_num_launches": (C.c_int, [_P, C.c_int]),
    "rldm_unet_trunk_status": (C.c_int, [_P, C.c_int]),
    "rldm_sampler_profile": (C.c_int, [_P, _P, C.c_char_p, C.c_size_t]),
    "rldm_test_conv": (C.c_int, [C.POINTER(ConvDescC), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "rldm_bench_conv": (C.c_int, [C.POINTER(ConvDescC), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float),
                                  C.c_char_p, C.c_size_t, _P]),
    "rldm_debug_force_tile": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "rldm_test_conv_stats": (C.c_int, [C.POINTER(ConvDescC), _P, _P, _P, _P, _P]),
    "rldm_debug_set_flags": (C.c_int, [C.c_int]),
    "rldm_debug_set_flags2": (C.c_int, [C.c_int]),
    "rldm_debug_graph_trace": (C.c_int, [_P, C.c_int, C.c_char_p, C.c_size_t]),
    "rldm_debug_timestamps": (C.c_int, [_P]),
    "rldm_debug_block_times": (C.c_int, [_P, C.c_int]),
    "rldm_test_attention": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "rldm_test_attention_qkv": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _P, _P, _P, _P, _P, _P]),
    "rldm_bench_attention_qkv": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), _P]),
}

_lib = None


def lib():
    """Load (once) and return the bound library; raises if it has not been built (`__graft_entry__.build()`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                               f"(hipcc --offload-arch=gfx950).  rangeldm_amd has no CPU fallback.")
        # PyTorch first: its wheel carries its own libamdhip64, and the library's dependency on that soname must resolve to the copy
        # torch loads -- loaded the other way round the process holds two HIP runtimes, and this one then finds "no ROCm-capable
        # device" (seen with build() and smoke() in one process on a GPU box)
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)           # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().rldm_last_error()
        raise RuntimeError(f"librangeldm_hip: {what} failed: {msg.decode() if msg else 'unknown error'}")


def stream_ptr(device=None):
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("rangeldm_amd needs an MI355X (gfx950) visible to PyTorch-ROCm; there is no CPU fallback")
    name = C.create_string_buffer(64)
    cus = C.c_int(0)
    check(lib().rldm_device_info(name, 64, C.byref(cus)), "rldm_device_info")
    return name.value.decode(), cus.value
