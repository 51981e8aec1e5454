"""ctypes binding of ``libdctr_hip.so`` (C ABI declared in ``include/dctr.h``).

The HIP extension IS the product path: there is no CPU or eager-PyTorch fallback.  ``lib()``
raises ``DctrExtensionError`` when the shared library is missing or cannot be loaded, and every
op raises when asked to run without a HIP device.
"""
import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libdctr_hip.so")

c_i32, c_i64, c_f32 = ctypes.c_int32, ctypes.c_int64, ctypes.c_float
c_vp, c_sz = ctypes.c_void_p, ctypes.c_size_t

ABI_VERSION = 13

E_NULL = -1                      # DCTR_E_NULL: a required pointer (or required workspace) is missing
E_UNSUPPORTED = -5               # DCTR_E_UNSUPPORTED: valid request this build does not implement
POOL_SUM, POOL_MEAN, POOL_MAX = 0, 1, 2
CROSS_VECTOR, CROSS_MATRIX = 0, 1
OPT_CODES = {"adam": 0, "adagrad": 1, "rmsprop": 2, "sgd": 3}
ACT_LINEAR, ACT_RELU, ACT_SIGMOID, ACT_TANH, ACT_DICE = 0, 1, 2, 3, 4
STATUS_INDEX_OOR = 1
STATUS_TIMEOUT = 2

ACT_CODES = {None: ACT_LINEAR, "linear": ACT_LINEAR, "relu": ACT_RELU, "sigmoid": ACT_SIGMOID, "tanh": ACT_TANH,
             "dice": ACT_DICE, "Dice": ACT_DICE}
POOL_CODES = {"sum": POOL_SUM, "mean": POOL_MEAN, "max": POOL_MAX}


class DctrExtensionError(RuntimeError):
    pass


class DctrError(RuntimeError):
    """A dctr_* entry point returned non-zero (``rc``: its return code, e.g. E_UNSUPPORTED)."""
    rc = None


# ---------------------------------------------------------------------------------------------
# struct mirrors (field order and types must match include/dctr.h)
# ---------------------------------------------------------------------------------------------
class FieldDesc(ctypes.Structure):
    _fields_ = [("table", c_vp), ("lin_table", c_vp), ("vocab", c_i64), ("dim", c_i32), ("out_offset", c_i32),
                ("in_fm", c_i32), ("hash_mode", c_i32), ("identity", c_i32), ("row_pitch", c_i32)]


class GatherFmArgs(ctypes.Structure):
    _fields_ = [("fields", c_vp), ("ids", c_vp), ("ids_stride_f", c_i64), ("ids_stride_b", c_i64),
                ("ids_is_i64", c_i32), ("n_fields", c_i32), ("max_dim", c_i32), ("all_dim4", c_i32),
                ("any_hash", c_i32), ("n_dense", c_i32), ("dense", c_vp), ("dense_stride", c_i64),
                ("dense_lin_w", c_vp), ("dense_out_offset", c_i32), ("dense_copy_cols", c_i32), ("batch", c_i64),
                ("dnn_in", c_vp), ("out_stride", c_i64), ("fm_logit", c_vp), ("lin_logit", c_vp), ("status", c_vp),
                ("split_col", c_i32), ("split_field", c_i32), ("uniform_dim", c_i32), ("any_identity", c_i32),
                ("any_pitch", c_i32), ("n_pools", c_i32), ("pools", c_vp), ("pool_row0", c_i64), ("pool_pieces", c_i32), ("pool_flags", c_i32)]


class PoolSeq(ctypes.Structure):       # dctr_pool_seq_t: a sequence feature pooled inside dctr_embed_mlp_fwd (DEVICE array)
    _fields_ = [("idx", c_vp), ("length", c_vp), ("idx_stride", c_i64), ("maxlen", c_i32), ("combiner", c_i32)]


class PoolArgs(ctypes.Structure):
    _fields_ = [("idx", c_vp), ("table", c_vp), ("lin_table", c_vp), ("length", c_vp), ("weight", c_vp),
                ("vocab", c_i64), ("idx_stride", c_i64), ("batch", c_i64), ("idx_is_i64", c_i32), ("maxlen", c_i32),
                ("dim", c_i32), ("combiner", c_i32), ("weight_norm", c_i32), ("hash_mode", c_i32), ("out", c_vp),
                ("out_stride", c_i64), ("lin_out", c_vp), ("status", c_vp)]


class LookupArgs(ctypes.Structure):
    _fields_ = [("idx", c_vp), ("table", c_vp), ("vocab", c_i64), ("n", c_i64), ("idx_is_i64", c_i32), ("dim", c_i32),
                ("hash_mode", c_i32), ("pad_", c_i32), ("out", c_vp), ("out_stride", c_i64), ("mask", c_vp),
                ("status", c_vp)]


class CinArgs(ctypes.Structure):
    _fields_ = [("x", c_vp), ("batch", c_i64), ("x_stride", c_i64), ("fields", c_i32), ("dim", c_i32),
                ("n_layers", c_i32), ("split_half", c_i32), ("activation", c_i32), ("workspace_ready", c_i32), ("layer_size", c_vp),
                ("filters", c_vp), ("bias", c_vp), ("out", c_vp), ("workspace", c_vp), ("workspace_bytes", c_sz),
                ("save_y", c_vp)]


class MlpArgs(ctypes.Structure):
    _fields_ = [("x", c_vp), ("batch", c_i64), ("x_stride", c_i64), ("in_dim", c_i32), ("n_layers", c_i32),
                ("units", c_vp), ("kernels", c_vp), ("biases", c_vp), ("activation", c_i32), ("has_head", c_i32),
                ("dice_alpha", c_vp), ("dice_mean", c_vp), ("dice_var", c_vp), ("dice_eps", c_f32),
                ("sigmoid_out", c_i32), ("head_w", c_vp), ("add", c_vp * 4), ("global_bias", c_vp),
                ("y", c_vp), ("y_stride", c_i64), ("workspace", c_vp), ("workspace_bytes", c_sz),
                ("save_acts", c_vp), ("tile_rows", c_i32), ("precision", c_i32), ("probe", c_vp), ("bn_scale", c_vp), ("bn_shift", c_vp),
                ("cross_w", c_vp), ("cross_b", c_vp), ("cross_head", c_vp), ("cross_layers", c_i32), ("cross_const", c_vp)]


class FieldGrad(ctypes.Structure):
    _fields_ = [("g_table", c_vp), ("g_lin_table", c_vp), ("touched", c_vp)]


class GatherFmBwdArgs(ctypes.Structure):
    _fields_ = [("fwd", ctypes.POINTER(GatherFmArgs)), ("grads", c_vp), ("d_dnn_in", c_vp), ("d_stride", c_i64),
                ("d_fm", c_vp), ("d_lin", c_vp), ("g_dense_lin_w", c_vp), ("dense_lin_rows", c_vp)]


class PoolBwdArgs(ctypes.Structure):
    _fields_ = [("fwd", ctypes.POINTER(PoolArgs)), ("d_out", c_vp), ("d_stride", c_i64), ("d_lin_out", c_vp),
                ("g_table", c_vp), ("g_lin_table", c_vp), ("touched", c_vp)]


class DnnTrainLayer(ctypes.Structure):
    _fields_ = [("z", c_vp), ("z_stride", c_i64), ("rows", c_i64), ("n", c_i32), ("activation", c_i32), ("use_bn", c_i32),
                ("bn_eps", c_f32), ("bn_momentum", c_f32), ("dropout_rate", c_f32), ("dropout_seed", ctypes.c_uint64),
                ("bn_gamma", c_vp), ("bn_beta", c_vp), ("bn_moving_mean", c_vp), ("bn_moving_var", c_vp), ("bn_batch_mean", c_vp),
                ("bn_batch_var", c_vp), ("h", c_vp), ("h_stride", c_i64), ("dh", c_vp), ("dh_stride", c_i64), ("dz", c_vp),
                ("d_gamma", c_vp), ("d_beta", c_vp), ("workspace", c_vp)]


class MlpBwdArgs(ctypes.Structure):
    _fields_ = [("x", c_vp), ("batch", c_i64), ("x_stride", c_i64), ("in_dim", c_i32), ("n_layers", c_i32),
                ("units", c_vp), ("kernels", c_vp), ("acts", c_vp), ("activation", c_i32), ("pad_", c_i32),
                ("head_w", c_vp), ("dlogit", c_vp), ("d_kernels", c_vp), ("d_biases", c_vp), ("d_head_w", c_vp),
                ("dx", c_vp), ("dx_stride", c_i64), ("workspace", c_vp), ("workspace_bytes", c_sz),
                ("d_out", c_vp), ("d_out_stride", c_i64), ("biases", c_vp), ("dice_alpha", c_vp), ("dice_mean", c_vp),
                ("dice_var", c_vp), ("d_dice_alpha", c_vp), ("dice_eps", c_f32), ("pad2_", c_i32),
                ("dice_batch_mean", c_vp), ("dice_batch_var", c_vp), ("dw_stream", c_vp), ("saved_z", c_vp)]


class CrossMixBwdArgs(ctypes.Structure):
    _fields_ = [("x", c_vp), ("x_stride", c_i64), ("batch", c_i64), ("dim", c_i32), ("layers", c_i32), ("experts", c_i32),
                ("low_rank", c_i32), ("U", c_vp), ("V", c_vp), ("C", c_vp), ("gating", c_vp), ("bias", c_vp), ("dy", c_vp),
                ("dy_stride", c_i64), ("dU", c_vp), ("dV", c_vp), ("dC", c_vp), ("dgating", c_vp), ("dbias", c_vp), ("dx", c_vp),
                ("dx_stride", c_i64), ("dx_accumulate", c_i32), ("pad_", c_i32), ("workspace", c_vp), ("workspace_bytes", c_sz)]


class CinBwdArgs(ctypes.Structure):
    _fields_ = [("fwd", ctypes.POINTER(CinArgs)), ("d_out", c_vp), ("out_dim", c_i32), ("dx_accumulate", c_i32),
                ("d_filters", c_vp), ("d_bias", c_vp), ("dx", c_vp), ("dx_stride", c_i64), ("workspace", c_vp),
                ("workspace_bytes", c_sz), ("saved_y", c_vp)]


class CrossBwdArgs(ctypes.Structure):
    _fields_ = [("x", c_vp), ("x_stride", c_i64), ("batch", c_i64), ("dim", c_i32), ("layers", c_i32), ("mode", c_i32),
                ("dx_accumulate", c_i32), ("kernels", c_vp), ("bias", c_vp), ("dy", c_vp), ("dy_stride", c_i64),
                ("d_kernels", c_vp), ("d_bias", c_vp), ("dx", c_vp), ("dx_stride", c_i64), ("workspace", c_vp),
                ("workspace_bytes", c_sz), ("saved_u", c_vp), ("saved_x", c_vp)]


class AfmBwdArgs(ctypes.Structure):
    _fields_ = [("x", c_vp), ("batch", c_i64), ("x_stride", c_i64), ("fields", c_i32), ("dim", c_i32), ("att_factor", c_i32),
                ("dx_accumulate", c_i32), ("att_w", c_vp), ("att_b", c_vp), ("proj_h", c_vp), ("proj_p", c_vp), ("dy", c_vp),
                ("dx", c_vp), ("dx_stride", c_i64), ("d_att_w", c_vp), ("d_att_b", c_vp), ("d_proj_h", c_vp), ("d_proj_p", c_vp)]


class HostCol(ctypes.Structure):
    _fields_ = [("src", c_vp), ("stride_bytes", c_i64), ("kind", c_i32), ("reserved_", c_i32)]


HOST_KINDS = {"int32": 0, "int64": 1, "float32": 2, "float64": 3}


class CrossnetArgs(ctypes.Structure):
    _fields_ = [("x", c_vp), ("batch", c_i64), ("x_stride", c_i64), ("dim", c_i32), ("layers", c_i32), ("mode", c_i32),
                ("workspace_ready", c_i32), ("kernels", c_vp), ("bias", c_vp), ("y", c_vp), ("y_stride", c_i64), ("workspace", c_vp),
                ("workspace_bytes", c_sz), ("head_w", c_vp), ("logit", c_vp), ("save_u", c_vp), ("save_x", c_vp)]


class AdamSeg(ctypes.Structure):
    _fields_ = [("w", c_vp), ("m", c_vp), ("v", c_vp), ("g", c_vp), ("n", c_i64), ("l2", c_f32), ("pad_", c_i32), ("touched", c_vp)]


class DinAttnArgs(ctypes.Structure):
    _fields_ = [("query", c_vp), ("keys", c_vp), ("key_mask", c_vp), ("batch", c_i64), ("maxlen", c_i32),
                ("dim", c_i32), ("n_layers", c_i32), ("activation", c_i32), ("units", c_vp), ("kernels", c_vp),
                ("biases", c_vp), ("dice_alpha", c_vp), ("dice_mean", c_vp), ("dice_var", c_vp), ("dice_eps", c_f32),
                ("weight_normalization", c_i32), ("out_kernel", c_vp), ("out_bias", c_vp), ("out", c_vp),
                ("out_stride", c_i64), ("scores", c_vp), ("workspace", c_vp), ("workspace_bytes", c_sz)]


class DinGatherArgs(ctypes.Structure):
    _fields_ = [("n_feats", c_i32), ("ids_is_i64", c_i32), ("hist_ids", c_vp * 2), ("hist_stride", c_i64), ("query_ids", c_vp * 2),
                ("query_stride", c_i64), ("hist_table", c_vp * 2), ("query_table", c_vp * 2), ("hist_vocab", c_i64 * 2),
                ("query_vocab", c_i64 * 2), ("mask_zero", c_i32 * 2), ("status", c_vp)]


# every symbol include/dctr.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "dctr_abi_version": (ctypes.c_int, []),
    "dctr_last_error": (ctypes.c_char_p, []),
    "dctr_target_arch": (ctypes.c_char_p, []),
    "dctr_profile_next_launch": (ctypes.c_int, []),
    "dctr_profile_last_ms": (ctypes.c_float, []),
    "dctr_wall_clock_khz": (ctypes.c_int, []),
    "dctr_profile_arm": (ctypes.c_int, [c_i32]),
    "dctr_profile_collect": (ctypes.c_int, [c_vp, c_i32]),
    "dctr_hash_bucket_i32": (ctypes.c_int, [c_vp, c_i64, c_i64, ctypes.c_int, c_vp, c_vp]),
    "dctr_hash_bucket_i64": (ctypes.c_int, [c_vp, c_i64, c_i64, ctypes.c_int, c_vp, c_vp]),
    "dctr_hash_bucket_bytes": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i64, ctypes.c_int, c_vp, c_vp]),
    "dctr_embed_gather_fm": (ctypes.c_int, [ctypes.POINTER(GatherFmArgs), c_vp]),
    "dctr_sgemm": (ctypes.c_int, [c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_i64, c_vp, c_i32, c_i64, ctypes.c_float, c_vp, c_i32,
                                  c_i64, c_i32, c_vp]),
    "dctr_hash_fields": (ctypes.c_int, [c_vp, c_i32, c_vp, c_i64, c_i64, c_i32, c_i64, c_vp, c_i64, c_i32, c_vp]),
    "dctr_embed_pool": (ctypes.c_int, [ctypes.POINTER(PoolArgs), c_vp]),
    "dctr_embed_lookup": (ctypes.c_int, [ctypes.POINTER(LookupArgs), c_vp]),
    "dctr_embed_lookup_multi": (ctypes.c_int, [ctypes.POINTER(LookupArgs), c_i32, c_vp, c_vp, c_i32, c_vp]),
    "dctr_seq_weight_fwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "dctr_fm_fwd": (ctypes.c_int, [c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_vp]),
    "dctr_crossnet_workspace_bytes": (c_sz, [c_i32, c_i32, c_i32, c_vp]),
    "dctr_crossnet_fwd": (ctypes.c_int, [c_vp, c_i64, c_i32, c_i64, c_vp, c_vp, c_i32, c_i32, c_vp, c_i64, c_vp, c_sz, c_vp]),
    "dctr_crossnet_head_fwd": (ctypes.c_int, [ctypes.POINTER(CrossnetArgs), c_vp]),
    "dctr_crossnet_gather_head_fwd": (ctypes.c_int, [ctypes.POINTER(CrossnetArgs), ctypes.POINTER(GatherFmArgs), c_vp]),
    "dctr_crossnet_fwd_supported": (ctypes.c_int, [ctypes.POINTER(CrossnetArgs), ctypes.POINTER(GatherFmArgs)]),
    "dctr_crossnet_matrix_step": (ctypes.c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_i32, c_vp, c_i64, c_vp]),
    "dctr_cin_workspace_bytes": (c_sz, [ctypes.POINTER(CinArgs)]),
    "dctr_cin_fwd": (ctypes.c_int, [ctypes.POINTER(CinArgs), c_vp]),
    "dctr_cin_fwd_supported": (ctypes.c_int, [ctypes.POINTER(CinArgs), ctypes.POINTER(GatherFmArgs), ctypes.c_int32]),
    "dctr_cin_gather_fwd": (ctypes.c_int, [ctypes.POINTER(CinArgs), ctypes.POINTER(GatherFmArgs), c_vp, c_vp, c_vp]),
    "dctr_afm_fwd": (ctypes.c_int, [c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "dctr_host_pack_columns": (ctypes.c_int, [c_vp, c_i32, c_i64, c_i64, c_vp, c_i64, c_i32, c_i32]),
    "dctr_crossnet_mix_workspace_bytes": (ctypes.c_size_t, [c_i32, c_i32, c_i32, c_i32]),
    "dctr_crossnet_mix_fwd": (ctypes.c_int, [c_vp, c_i64, c_i32, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp,
                                             c_i64, c_vp, ctypes.c_size_t, c_vp]),
    "dctr_bi_interaction_fwd": (ctypes.c_int, [c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "dctr_inner_product_fwd": (ctypes.c_int, [c_vp, c_i64, c_i64, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "dctr_mlp_workspace_bytes": (c_sz, [ctypes.POINTER(MlpArgs)]),
    "dctr_mlp_fwd_supported": (ctypes.c_int, [ctypes.POINTER(GatherFmArgs), ctypes.POINTER(MlpArgs), ctypes.c_int32, ctypes.c_int32]),
    "dctr_crossnet_fold_consts": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp]),
    "dctr_mlp_fwd": (ctypes.c_int, [ctypes.POINTER(MlpArgs), c_vp]),
    "dctr_embed_mlp_fwd": (ctypes.c_int, [ctypes.POINTER(GatherFmArgs), ctypes.POINTER(MlpArgs), c_i32, c_i32, c_vp]),
    "dctr_embed_mlp_fwd_last_kernel": (ctypes.c_int, []),
    "dctr_embed_mlp_fwd_plan": (ctypes.c_int, [ctypes.POINTER(GatherFmArgs), ctypes.POINTER(MlpArgs), ctypes.POINTER(ctypes.c_int64),
                                               ctypes.POINTER(c_i32), ctypes.POINTER(c_i32), c_i32]),
    "dctr_bce_grad": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "dctr_bce_grad_w": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "dctr_embed_gather_fm_bwd": (ctypes.c_int, [ctypes.POINTER(GatherFmBwdArgs), c_vp]),
    "dctr_embed_pool_bwd": (ctypes.c_int, [ctypes.POINTER(PoolBwdArgs), c_vp]),
    "dctr_afm_bwd": (ctypes.c_int, [c_vp, c_vp]),
    "dctr_din_att_in_fwd": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp]),
    "dctr_din_wsum_fwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "dctr_din_wsum_bwd": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "dctr_din_att_in_bwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_i64, c_vp, c_vp]),
    "dctr_embed_lookup_bwd": (ctypes.c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "dctr_bi_interaction_bwd": (ctypes.c_int, [c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_i64, c_vp, c_i64, c_i32, c_vp]),
    "dctr_inner_product_bwd": (ctypes.c_int, [c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_i64, c_vp, c_i64, c_i32, c_vp]),
    "dctr_dense1_bwd": (ctypes.c_int, [c_vp, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp]),
    "dctr_crossnet_bwd_workspace_bytes": (c_sz, [ctypes.POINTER(CrossBwdArgs)]),
    "dctr_crossnet_bwd": (ctypes.c_int, [ctypes.POINTER(CrossBwdArgs), c_vp]),
    "dctr_cin_bwd_workspace_bytes": (c_sz, [ctypes.POINTER(CinBwdArgs)]),
    "dctr_cin_bwd": (ctypes.c_int, [ctypes.POINTER(CinBwdArgs), c_vp]),
    "dctr_mlp_bwd_workspace_bytes": (c_sz, [ctypes.POINTER(MlpBwdArgs)]),
    "dctr_mlp_bwd_join": (ctypes.c_int, [c_vp, c_vp]),
    "dctr_mlp_bwd": (ctypes.c_int, [ctypes.POINTER(MlpBwdArgs), c_vp]),
    "dctr_crossnet_mix_bwd_workspace_bytes": (c_sz, [ctypes.POINTER(CrossMixBwdArgs)]),
    "dctr_crossnet_mix_bwd": (ctypes.c_int, [ctypes.POINTER(CrossMixBwdArgs), c_vp]),
    "dctr_fm_bwd": (ctypes.c_int, [c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_vp, c_i64, c_i32, c_vp]),
    "dctr_din_softmax_fwd": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i32, c_vp, c_vp]),
    "dctr_din_softmax_bwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp]),
    "dctr_dnn_train_layer_fwd": (ctypes.c_int, [ctypes.POINTER(DnnTrainLayer), c_vp]),
    "dctr_dnn_train_layer_bwd": (ctypes.c_int, [ctypes.POINTER(DnnTrainLayer), c_vp]),
    "dctr_dice_train_fwd": (ctypes.c_int, [c_vp, c_i64, c_vp, c_i64, c_i32, c_vp, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "dctr_adam_step": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32, c_i32, c_vp]),
    "dctr_adam_multi": (ctypes.c_int, [c_vp, c_i32, c_i64, c_f32, c_f32, c_f32, c_f32, c_i32, c_vp]),
    "dctr_opt_multi": (ctypes.c_int, [c_i32, c_vp, c_i32, c_i64, c_f32, c_f32, c_f32, c_f32, c_i32, c_vp]),
    "dctr_opt_multi_l2": (ctypes.c_int, [ctypes.c_int32, c_vp, c_i32, c_i64, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                          c_i32, c_vp, ctypes.c_float, c_vp]),
    "dctr_din_attn_workspace_bytes": (c_sz, [ctypes.POINTER(DinAttnArgs)]),
    "dctr_din_attn_pool_fwd": (ctypes.c_int, [ctypes.POINTER(DinAttnArgs), c_vp]),
    "dctr_din_attn_gather_fwd": (ctypes.c_int, [ctypes.POINTER(DinAttnArgs), ctypes.POINTER(DinGatherArgs), c_vp]),
}

_LIB = None


def lib():
    """Load (once) and return the ctypes handle.  torch is imported FIRST so that the library's
    ``NEEDED libamdhip64.so.7`` binds to the HIP runtime PyTorch already loaded."""
    global _LIB
    if _LIB is not None:
        return _LIB
    import torch  # noqa: F401  (must precede CDLL: one HIP runtime per process)
    if not os.path.exists(LIB_PATH):
        raise DctrExtensionError(
            "HIP extension %s is not built. Run `python -m deepctr_amd.build` (needs hipcc; target gfx950). "
            "There is no CPU / PyTorch fallback for this path." % LIB_PATH)
    try:
        handle = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise DctrExtensionError("cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(handle, name)
        except AttributeError:
            raise DctrExtensionError("%s does not export %s (stale build? run python -m deepctr_amd.build --force)"
                                     % (LIB_PATH, name))
        fn.restype = res
        fn.argtypes = args
    ver = handle.dctr_abi_version()
    if ver != ABI_VERSION:
        raise DctrExtensionError("ABI mismatch: library %d, python %d" % (ver, ABI_VERSION))
    _LIB = handle
    return _LIB


def check(rc, what):
    if rc != 0:
        msg = lib().dctr_last_error()
        err = DctrError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else ""))
        err.rc = int(rc)
        raise err


def require_device():
    """Return the torch HIP device or raise: the path never runs on the CPU."""
    import torch
    if not torch.cuda.is_available():
        raise DctrExtensionError("no HIP device visible: deepctr_amd's forward path only runs on an AMD GPU "
                                 "(MI355X / gfx950); there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
