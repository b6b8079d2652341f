"""ctypes binding of libresnet_mi.so (include/resnet_mi.h).

This is plumbing only: the product is the C-ABI library.  There is no CPU fallback -- if the library is
missing or a symbol cannot be resolved, importing/using this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RESNET_MI_LIB: load another build of the same library (kernel A/B experiments, tools/variant.sh)
LIB_PATH = os.environ.get("RESNET_MI_LIB") or os.path.join(_HERE, "libresnet_mi.so")

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)


class Dims(C.Structure):
    _fields_ = [("input", C.c_int), ("init_kernel_dim", C.c_int), ("init_conv_filters", C.c_int),
                ("init_conv_stride", C.c_int), ("init_maxpool_dim", C.c_int), ("init_maxpool_stride", C.c_int),
                ("n_conv_blocks", C.c_int), ("is_block_spatial_reduction", _ip), ("final_depth", C.c_int),
                ("output", C.c_int)]


class BatchNorm(C.Structure):
    _fields_ = [("spatial_dim", C.c_int), ("depth", C.c_int), ("gamma", _fp), ("beta", _fp)]


class ConvBlock(C.Structure):
    _fields_ = [("incoming_filters", C.c_int), ("incoming_spatial_dim", C.c_int), ("reduced_depth", C.c_int),
                ("expanded_depth", C.c_int), ("stride", C.c_int), ("depth_reduction", _fp),
                ("norm_depth_reduction", C.POINTER(BatchNorm)), ("spatial", _fp), ("norm_spatial", C.POINTER(BatchNorm)),
                ("depth_expansion", _fp), ("norm_expansion", C.POINTER(BatchNorm)), ("projection", _fp),
                ("norm_projection", C.POINTER(BatchNorm))]


class Params(C.Structure):
    _fields_ = [("init_conv_layer", _fp), ("norm_init_conv", C.POINTER(BatchNorm)),
                ("conv_blocks", C.POINTER(C.POINTER(ConvBlock))), ("fully_connected", _fp),
                ("locations", C.POINTER(_fp)), ("sizes", _ip), ("n_locations", C.c_int)]


class Cache_BatchNorm(C.Structure):
    _fields_ = [("input_size", C.c_int), ("feature_size", C.c_int), ("means", _fp), ("vars", _fp),
                ("normalized_temp", _fp), ("normalized", _fp)]


_cbp = C.POINTER(Cache_BatchNorm)


class Activation_ConvBlock(C.Structure):
    _fields_ = [("incoming_filters", C.c_int), ("incoming_spatial_dim", C.c_int), ("reduced_depth", C.c_int),
                ("expanded_depth", C.c_int), ("stride", C.c_int),
                ("post_reduced", _fp), ("norm_post_reduced", _cbp), ("post_reduced_activated", _fp),
                ("post_spatial", _fp), ("norm_post_spatial", _cbp), ("post_spatial_activated", _fp),
                ("post_expanded", _fp), ("norm_post_expanded", _cbp), ("post_expanded_norm_vals", _fp),
                ("transformed_residual", _fp), ("norm_post_projection", _cbp), ("post_projection_norm_vals", _fp),
                ("output", _fp), ("output_activated", _fp)]


class Activations(C.Structure):
    _fields_ = [("init_conv_applied", _fp), ("norm_init_conv", _cbp), ("init_conv_activated", _fp),
                ("max_inds", _ip), ("init_convblock_input", _fp),
                ("activation_conv_blocks", C.POINTER(C.POINTER(Activation_ConvBlock))), ("n_conv_blocks", C.c_int),
                ("final_conv_output_pooled", _fp), ("linear_output", _fp)]


class ResNet(C.Structure):
    _fields_ = [("dims", C.POINTER(Dims)), ("params", C.POINTER(Params))]


class Forward_Buffer(C.Structure):
    _fields_ = [("activations", C.POINTER(Activations)), ("pred", _fp), ("pred_cpu", _fp)]


class Backprop_Buffer(C.Structure):
    _fields_ = [("output_layer_deriv", _fp), ("param_derivs", C.POINTER(Params)), ("prev_means", C.POINTER(Params)),
                ("prev_vars", C.POINTER(Params)), ("activation_derivs", C.POINTER(Activations))]


class Batch(C.Structure):
    _fields_ = [("image_dim", C.c_int), ("image_size", C.c_int), ("n_images", C.c_int), ("cur_shard_id", C.c_int),
                ("cur_batch_in_shard", C.c_int), ("shard_n_images", C.c_int), ("full_shard_images", _fp),
                ("full_shard_correct_classes", _ip), ("images_float_cpu", _fp), ("images", _fp),
                ("correct_classes_cpu", _ip), ("correct_classes", _ip)]


class Train_ResNet(C.Structure):
    _fields_ = [("model", C.POINTER(ResNet)), ("cur_batch", C.POINTER(Batch)),
                ("forward_buffer", C.POINTER(Forward_Buffer)), ("backprop_buffer", C.POINTER(Backprop_Buffer)),
                ("learning_rate", C.c_float), ("weight_decay", C.c_float), ("base_mean_decay", C.c_float),
                ("base_var_decay", C.c_float), ("cur_mean_decay", C.c_float), ("cur_var_decay", C.c_float),
                ("eps", C.c_float), ("batch_size", C.c_int), ("n_epochs", C.c_int), ("cur_dump_id", C.c_int),
                ("cur_epoch", C.c_int), ("loss_per_epoch", _fp), ("accuracy_per_epoch", _fp),
                ("init_loaded", C.c_int), ("backend_ctx", C.c_void_p), ("dump_dir", C.c_char_p)]


MI_SRC_SHARDS, MI_SRC_BUFFER, MI_SRC_SYNTHETIC, MI_SRC_HOST = 0, 1, 2, 3
MI_LAYOUT_NHWC, MI_LAYOUT_NCHW = 0, 1
MI_DTYPE_F32, MI_DTYPE_BF16 = 0, 1
MI_STORE_FAST, MI_STORE_RECOMPUTE_BN, MI_STORE_FULL = 0, 1, 2

# every symbol include/resnet_mi.h declares: name -> (restype, argtypes)
_i, _f, _vp, _sz, _u64, _cp = C.c_int, C.c_float, C.c_void_p, C.c_size_t, C.c_uint64, C.c_char_p
_T = C.POINTER(Train_ResNet)
_B = C.POINTER(Batch)
PROTOTYPES = {
    "init_dimensions": (C.POINTER(Dims), [_i] * 7 + [_ip, _i, _i]),
    "mi_rng_create": (_vp, [_u64]),
    "mi_rng_destroy": (None, [_vp]),
    "init_resnet": (C.POINTER(ResNet), [C.POINTER(Dims), _vp]),
    "init_general_batch": (_B, [_i, _i, _i, _i]),
    "init_trainer": (_T, [C.POINTER(ResNet), _B, _i, _f, _f, _f, _f, _f, _i, _cp]),
    "init_trainer_cudnn_abi": (_T, [C.POINTER(ResNet), _B, _i, _f, _f, _f, _f, _f, _i, _vp, _cp]),
    "populate_class_info": (_vp, [_cp, _cp, _cp, _i]),
    "load_new_batch": (None, [_T, _vp, _B]),
    "forward_pass": (None, [_T]),
    "backwards_pass": (None, [_T]),
    "update_parameters": (None, [_T]),
    "dump_trainer": (None, [_i, _T, _cp]),
    "overwrite_trainer_hyperparams": (None, [_T, _i, _cp]),
    "overwrite_model_params": (None, [_T, _i, _cp]),
    "mi_device_count": (_i, []),
    "mi_set_device": (_i, [_i]),
    "mi_last_error": (_cp, []),
    "mi_device_synchronize": (None, []),
    "destroy_trainer": (None, [_T]),
    "mi_batch_source_shards": (None, [_B, _cp, _i]),
    "mi_batch_set_prefetch": (None, [_B, _i]),
    "mi_batch_source_buffer": (None, [_B, _cp, _cp, _i]),
    "mi_batch_source_synthetic": (None, [_B, _u64, _u64, _i, _i]),
    "mi_batch_source_host": (None, [_B, _i]),
    "mi_batch_last_status": (_i, [_B]),
    "mi_trainer_set_full_store": (None, [_T, _i]),
    "mi_trainer_set_dump_root": (None, [_T, _cp]),
    "mi_trainer_set_dump_every": (None, [_T, _i]),
    "mi_trainer_set_input_reset": (None, [_T, _i]),
    "mi_trainer_set_overlap": (None, [_T, _i]),
    "mi_host_loss": (_f, [_T, _ip]),
    "mi_copy_to_device": (None, [_vp, _vp, _sz]),
    "mi_copy_to_host": (None, [_vp, _vp, _sz]),
    "mi_dp_unique_id_bytes": (_i, []),
    "mi_dp_get_unique_id": (_i, [_vp, _i]),
    "mi_dp_init": (_i, [_T, _i, _i, _vp, _i]),
    "mi_dp_set_bucket_bytes": (None, [_T, _sz]),
    "mi_dp_world": (_i, [_T]),
    "mi_dp_enable_sync_bn": (_i, [_T, _vp, _i]),
    "mi_trainer_last_timings": (None, [_T, C.POINTER(C.c_float * 5)]),
    "mi_prof_enable": (None, [_i]),
    "mi_prof_reset": (None, []),
    "mi_prof_get": (None, [_i, C.POINTER(C.c_long), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "mi_malloc": (_vp, [_sz]),
    "mi_free": (None, [_vp]),
    "mi_op_conv_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i]),
    "mi_op_conv_dgrad": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i]),
    "mi_op_conv_wgrad": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i]),
    "mi_op_bn_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i]),
    "mi_op_bn_fwd_add_relu": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f]),
    "mi_op_bn_bwd": (_i, [_vp] * 10 + [_i, _i, _i, _f, _i]),
    "mi_op_bn_bwd_gate": (_i, [_vp] * 11 + [_i, _i, _i, _f]),
    "mi_op_maxpool_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i]),
    "mi_op_maxpool_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i]),
    "mi_op_avgpool_fwd": (_i, [_vp, _vp, _i, _i, _i]),
    "mi_op_avgpool_bwd": (_i, [_vp, _vp, _i, _i, _i]),
    "mi_op_relu_deriv": (_i, [_vp, _vp, _vp, _sz]),
    "mi_op_matmul": (_i, [_vp, _vp, _vp, _i, _i, _i]),
    "mi_op_matmul_lt": (_i, [_vp, _vp, _vp, _i, _i, _i]),
    "mi_op_matmul_rt": (_i, [_vp, _vp, _vp, _i, _i, _i]),
    "mi_op_softmax": (_i, [_vp, _vp, _i, _i]),
    "mi_op_ce_deriv": (_i, [_vp, _vp, _vp, _i, _i]),
    "mi_op_adam": (_i, [_vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _f, _f, _f, _vp]),
    "mi_op_nhwc_to_nchw": (_i, [_vp, _vp, _i, _i, _i, _i]),
    "mi_op_fill_uniform": (_i, [_vp, _sz, _u64, _f, _f]),
    "mi_debug_poison_lds": (_i, []),
    "mi_debug_conv_plan": (_i, [_i] * 7 + [_vp]),
    "mi_trainer_set_dtype": (_i, [_T, _i]),
    "mi_trainer_get_dtype": (_i, [_T]),
    "mi_trainer_set_store_policy": (_i, [_T, _i]),
    "mi_trainer_activation_bytes": (_sz, [_T]),
    "mi_trainer_device_bytes": (_sz, [_T]),
    "mi_clear_error": (None, []),
    "mi_trainer_check_errors": (_i, [_T]),
    "mi_trainer_end_epoch": (None, [_T, _f, _f, _f]),
    "mi_trainer_nan_location": (_i, [_T]),
    "mi_trainer_stem_dtype": (_i, [_T]),
    "mi_trainer_set_nan_exit": (None, [_T, _i]),
    "mi_debug_bn_merge": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp]),
    "mi_debug_dp_plan": (_i, [C.POINTER(Dims), _sz, _vp, _vp, _i]),
    "mi_debug_arena_floats": (_sz, [C.POINTER(Dims)]),
    "mi_debug_last_buckets": (_i, [_T, _vp, _vp, _i]),
    "mi_build_shard": (_i, [_cp, _cp, _cp, _i, _i, _i, _i]),
    "mi_batch_set_rank_slice": (None, [_B, _i, _i]),
    "mi_op_convert": (_i, [_vp, _i, _vp, _i, _sz]),
    "mi_bf16_conv_supported": (_i, [_i] * 7),
    "mi_bf16_pw_wgrad_supported": (_i, [_i] * 4),
    "mi_op_conv_fwd_bf16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i]),
    "mi_op_conv_dgrad_bf16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i]),
    "mi_op_conv_wgrad_bf16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i]),
    "mi_op_conv_dgrad_bn_bwd_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp]),
    "mi_op_conv_fwd_bf16_cl": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i]),
    "mi_op_conv_wgrad_bf16_cl": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i]),
    "mi_op_conv_wgrad_bf16_cl2": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i]),
    "mi_op_conv1x1_fwd_bf16_cl": (_i, [_vp, _vp, _vp, _i, _i, _i, _i]),
    "mi_op_bn_fwd_cl_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, C.c_float, _i]),
    "mi_op_conv_dgrad_bf16_cl": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i]),
    "mi_op_conv_dgrad_bn_bwd_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp]),
    "mi_op_stem_fwd_f32": (_i, [_vp, _vp, _vp, _i, _i]),
    "mi_op_stem_wgrad_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i]),
    "mi_op_stem_fwd_bf16": (_i, [_vp, _vp, _vp, _i, _i]),
    "mi_op_stem_wgrad_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i]),
    "mi_op_conv_bn_fwd_t": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i]),
    "mi_op_bn_fwd_t": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i]),
    "mi_op_bn_apply_t": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i]),
    "mi_op_bn_bwd_t": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _f, _i]),
    "mi_op_maxpool_fwd_t": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _i]),
    "mi_op_maxpool_bwd_t": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i]),
    "mi_op_avgpool_fwd_t": (_i, [_vp, _i, _vp, _i, _i, _i]),
    "mi_op_avgpool_bwd_t": (_i, [_vp, _vp, _i, _i, _i, _i]),
}

_lib = None


def load():
    """Load libresnet_mi.so and bind every declared entry point.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libresnet_mi.so is not built (run __graft_entry__.build() or make -C resnet_amd/csrc); "
                           "there is no CPU fallback")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
