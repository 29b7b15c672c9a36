"""ctypes binding of libshowtell_hip.so (the C ABI declared in include/showtell_hip.h).

There is no CPU fallback: if the HIP library is missing, every op raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libshowtell_hip.so")

ST_F32, ST_BF16 = 0, 1
ST_CELL_GRU, ST_CELL_LSTM = 0, 1

c_p, c_i, c_f, c_l = C.c_void_p, C.c_int, C.c_float, C.c_long


class ConvDesc(C.Structure):
    _fields_ = [("x", c_p), ("w", c_p), ("y", c_p), ("bias", c_p), ("scale", c_p), ("shift", c_p),
                ("residual", c_p), ("stats", c_p), ("dtype", c_i), ("out_dtype", c_i),
                ("B", c_i), ("Hin", c_i), ("Win", c_i), ("Cin", c_i), ("Ho", c_i), ("Wo", c_i), ("N", c_i),
                ("KH", c_i), ("KW", c_i), ("stride", c_i), ("pad", c_i),
                ("ldx", c_i), ("ldw", c_i), ("ldy", c_i), ("relu", c_i), ("accumulate", c_i), ("Cin_logical", c_i), ("k_order", c_i), ("stats_replicas", c_i),
                ("in_stats", c_p), ("in_gamma", c_p), ("in_beta", c_p), ("in_count", c_f), ("in_eps", c_f), ("split_k", c_i)]


class Conv3x3ImgDesc(C.Structure):
    _fields_ = [("x", c_p), ("w_frag", c_p), ("y", c_p), ("stats", c_p), ("stats_replicas", c_i), ("scale", c_p), ("shift", c_p),
                ("relu", c_i), ("in_stats", c_p), ("in_gamma", c_p), ("in_beta", c_p), ("in_count", c_f), ("in_eps", c_f),
                ("B", c_i), ("H", c_i), ("W", c_i), ("C", c_i), ("N", c_i), ("in_stats_replicas", c_i)]


class Conv1x1WregDesc(C.Structure):
    _fields_ = [("x", c_p), ("w_frag", c_p), ("y", c_p), ("residual", c_p), ("stats", c_p), ("stats_replicas", c_i), ("scale", c_p),
                ("shift", c_p), ("relu", c_i), ("in_stats", c_p), ("in_gamma", c_p), ("in_beta", c_p), ("in_count", c_f), ("in_eps", c_f),
                ("in_stats_replicas", c_i), ("B", c_i), ("Hin", c_i), ("Win", c_i), ("C", c_i), ("N", c_i), ("stride", c_i)]


class Conv1x1KfuseDesc(C.Structure):
    _fields_ = [("raw", c_p), ("identity", c_p), ("x_out", c_p), ("w_frag", c_p), ("y", c_p), ("stats", c_p), ("stats_replicas", c_i),
                ("f_stats", c_p), ("f_gamma", c_p), ("f_beta", c_p), ("f_count", c_f), ("f_eps", c_f), ("f_stats_replicas", c_i),
                ("rows", c_l), ("C", c_i), ("N", c_i),
                ("id_stats", c_p), ("id_gamma", c_p), ("id_beta", c_p), ("id_stats_replicas", c_i)]


class StemConvPoolDesc(C.Structure):
    _fields_ = [("x_s2d", c_p), ("w_frag", c_p), ("y", c_p), ("stats", c_p), ("stats_replicas", c_i), ("gamma", c_p),
                ("scale", c_p), ("shift", c_p), ("B", c_i), ("H", c_i), ("W", c_i)]


class ConvB2bDesc(C.Structure):
    _fields_ = [("raw2", c_p), ("w3_frag", c_p), ("identity", c_p), ("x_out", c_p), ("w1_frag", c_p), ("y", c_p),
                ("stats", c_p), ("stats_replicas", c_i),
                ("bn2_stats", c_p), ("bn2_gamma", c_p), ("bn2_beta", c_p), ("bn2_replicas", c_i),
                ("bn3_stats", c_p), ("bn3_gamma", c_p), ("bn3_beta", c_p), ("bn3_replicas", c_i),
                ("id_stats", c_p), ("id_gamma", c_p), ("id_beta", c_p), ("id_replicas", c_i),
                ("count", c_f), ("eps", c_f), ("rows", c_l), ("C1", c_i), ("C2", c_i), ("N", c_i)]


class ConvC3c1Desc(C.Structure):
    _fields_ = [("x2", c_p), ("w3_frag", c_p), ("identity", c_p), ("x_out", c_p), ("w1_frag", c_p), ("y", c_p),
                ("stats", c_p), ("stats_replicas", c_i),
                ("bn2_stats", c_p), ("bn2_gamma", c_p), ("bn2_beta", c_p), ("bn2_replicas", c_i),
                ("bn3_stats", c_p), ("bn3_gamma", c_p), ("bn3_beta", c_p), ("bn3_replicas", c_i),
                ("count", c_f), ("eps", c_f),
                ("scale3", c_p), ("shift3", c_p), ("scale1", c_p), ("shift1", c_p), ("relu1", c_i),
                ("rows", c_l), ("C1", c_i), ("C2", c_i), ("N", c_i),
                ("id_stats", c_p), ("id_gamma", c_p), ("id_beta", c_p), ("id_replicas", c_i)]


class BnActDesc(C.Structure):
    _fields_ = [("x", c_p), ("y", c_p), ("res", c_p), ("stats", c_p), ("gamma", c_p), ("beta", c_p),
                ("running_mean", c_p), ("running_var", c_p), ("res_stats", c_p), ("res_gamma", c_p),
                ("res_beta", c_p), ("res_running_mean", c_p), ("res_running_var", c_p), ("res_bn", c_i),
                ("dtype", c_i), ("rows", c_l), ("C", c_i), ("count", c_f), ("eps", c_f), ("relu", c_i),
                ("stats_replicas", c_i), ("res_stats_replicas", c_i)]


ST_MAX_LAYERS = 8


class RnnParams(C.Structure):
    _fields_ = [("cell", c_i), ("dtype", c_i), ("L", c_i), ("in0", c_i), ("H", c_i), ("V", c_i), ("E", c_i),
                ("emb", c_p), ("w_ih", c_p * ST_MAX_LAYERS), ("w_hh", c_p * ST_MAX_LAYERS),
                ("b_ih", c_p * ST_MAX_LAYERS), ("b_hh", c_p * ST_MAX_LAYERS), ("w_lin", c_p), ("b_lin", c_p)]


class RnnGrads(C.Structure):
    _fields_ = [("emb", c_p), ("w_ih", c_p * ST_MAX_LAYERS), ("w_hh", c_p * ST_MAX_LAYERS),
                ("b_ih", c_p * ST_MAX_LAYERS), ("b_hh", c_p * ST_MAX_LAYERS), ("w_lin", c_p), ("b_lin", c_p)]


class AttnParams(C.Structure):
    _fields_ = [("rnn", RnnParams), ("F", c_i), ("A", c_i), ("P", c_i),
                ("w_enc", c_p), ("b_enc", c_p), ("w_dec", c_p), ("b_dec", c_p), ("w_full", c_p), ("b_full", c_p),
                ("w_init_h", c_p), ("b_init_h", c_p), ("w_init_c", c_p), ("b_init_c", c_p), ("w_embed", c_p), ("b_embed", c_p)]


class AttnGrads(C.Structure):
    _fields_ = [("rnn", RnnGrads)] + [(n, c_p) for n in ("w_enc", "b_enc", "w_dec", "b_dec", "w_full", "b_full", "w_init_h", "b_init_h",
                                                        "w_init_c", "b_init_c", "w_embed", "b_embed")]


class PackedSeq(C.Structure):
    _fields_ = [("B", c_i), ("T", c_i), ("ntok", c_i), ("Tcap", c_i), ("batch_sizes_host", C.POINTER(c_i)),
                ("rows_b", c_p), ("rows_t", c_p), ("prev_row", c_p), ("caption", c_p)]


class ImageBatchDesc(C.Structure):
    _fields_ = [("src", c_p), ("src_bytes", C.c_int64), ("offset", c_p), ("height", c_p), ("width", c_p), ("flip", c_p), ("batch", c_i),
                ("max_height", c_i), ("max_width", c_i), ("out_h", c_i), ("out_w", c_i), ("lut", c_p), ("tmp", c_p),
                ("out", c_p), ("out_u8", c_p)]


_SIGS = {
    "st_version": ([], c_i),
    "st_image_transform": ([C.POINTER(ImageBatchDesc), c_p], c_i),
    "st_conv": ([C.POINTER(ConvDesc), c_p], c_i),
    "st_conv_batch": ([C.POINTER(ConvDesc), c_i, c_p], c_i),
    "st_conv3x3_img_supported": ([c_i, c_i, c_i, c_i], c_i),
    "st_conv3x3_img": ([C.POINTER(Conv3x3ImgDesc), c_p], c_i),
    "st_conv3x3_s2_supported": ([c_i, c_i], c_i),
    "st_conv3x3_s2": ([C.POINTER(Conv3x3ImgDesc), c_p], c_i),
    "st_conv1x1_wreg_supported": ([c_i, c_i], c_i),
    "st_conv1x1_wreg": ([C.POINTER(Conv1x1WregDesc), c_p], c_i),
    "st_conv1x1_kfuse": ([C.POINTER(Conv1x1KfuseDesc), c_p], c_i),
    "st_conv1x1_kfuse_supported": ([c_i, c_i], c_i),
    "st_conv1x1_kfuse8": ([C.POINTER(Conv1x1KfuseDesc), c_p], c_i),
    "st_conv_b2b": ([C.POINTER(ConvB2bDesc), c_p], c_i),
    "st_conv_b2b_supported": ([c_i, c_i, c_i], c_i),
    "st_conv1x1_astat_supported": ([c_i, c_i], c_i),
    "st_conv1x1_astat": ([C.POINTER(Conv1x1WregDesc), c_p], c_i),
    "st_conv1x1_kstream_supported": ([c_i, c_i], c_i),
    "st_conv1x1_kstream": ([C.POINTER(Conv1x1WregDesc), c_p], c_i),
    "st_pack_conv_weight_frag": ([c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p], c_i),
    "st_tune": ([c_i, c_i, c_i], c_i),
    "st_prof_enable": ([c_i], c_i),
    "st_debug_stamps": ([c_p], c_i),
    "st_prof_collect": ([C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(c_l)], c_i),
    "st_bn_act": ([C.POINTER(BnActDesc), c_p], c_i),
    "st_bn_update_running": ([c_p, c_p, c_p, c_i, c_f, c_f, c_p], c_i),
    "st_nchw_to_nhwc": ([c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p], c_i),
    "st_nchw_to_s2d16": ([c_p, c_p, c_i, c_i, c_i, c_i, c_p], c_i),
    "st_stem_weight_s2d": ([c_p, c_p, c_i, c_i, c_p], c_i),
    "st_stem_weight_frag": ([c_p, c_p, c_p], c_i),
    "st_stem_weight_frag_packed": ([c_p, c_i, c_p, c_p], c_i),
    "st_stem_conv_pool": ([C.POINTER(StemConvPoolDesc), c_p], c_i),
    "st_nhwc_to_ncp_f32": ([c_p, c_p, c_i, c_i, c_i, c_i, c_p], c_i),
    "st_maxpool3x3s2": ([c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p], c_i),
    "st_maxpool3x3s2_bn": ([c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_f, c_f, c_p], c_i),
    "st_global_avgpool": ([c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p], c_i),
    "st_cast": ([c_p, c_p, c_i, c_i, c_l, c_p], c_i),
    "st_transpose": ([c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p], c_i),
    "st_transpose_colsum": ([c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p], c_i),
    "st_transpose_batch": ([c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p], c_i),
    "st_pack_conv_weight": ([c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p], c_i),
    "st_cast2d": ([c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p], c_i),
    "st_rnn_vocab_ld": ([c_i], c_i),
    "st_rnn_workspace_bytes": ([C.POINTER(RnnParams), C.POINTER(PackedSeq)], C.c_size_t),
    "st_rnn_forward": ([C.POINTER(RnnParams), C.POINTER(PackedSeq), c_p, c_p, c_p, C.c_size_t, c_p, c_i, c_i, c_p, c_i, c_p], c_i),
    "st_rnn_fused_loss_supported": ([C.POINTER(RnnParams)], c_i),
    "st_rnn_fused_loss_bytes": ([C.POINTER(RnnParams), C.POINTER(PackedSeq)], C.c_size_t),
    "st_rnn_fused_loss": ([C.POINTER(RnnParams), C.POINTER(PackedSeq), c_p, C.c_size_t, c_p, c_p, C.c_size_t, c_p, c_p], c_i),
    "st_rnn_fused_dlogits": ([C.POINTER(RnnParams), C.POINTER(PackedSeq), c_p, C.c_size_t, c_p, c_p, c_p, c_p, c_i, c_p], c_i),
    "st_rnn_backward": ([C.POINTER(RnnParams), C.POINTER(RnnGrads), C.POINTER(PackedSeq), c_p, c_p, c_i, c_p, c_p, C.c_size_t,
                         c_p, c_p, c_p], c_i),
    "st_rnn_greedy_workspace_bytes": ([C.POINTER(RnnParams), c_i], C.c_size_t),
    "st_rnn_greedy": ([C.POINTER(RnnParams), c_p, c_i, c_i, c_p, C.c_size_t, c_p, c_p, c_p], c_i),
    "st_rnn_step": ([C.POINTER(RnnParams), c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_p], c_i),
    "st_embedding_rows": ([c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p], c_i),
    "st_gather_state": ([c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p], c_i),
    "st_softmax_topk": ([c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_p], c_i),
    "st_beam_select": ([c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_l, c_p, c_p, c_p, c_p, c_p, c_p], c_i),
    "st_attn_workspace_bytes": ([c_p, c_p], C.c_size_t),
    "st_attn_forward": ([c_p, c_p, c_p, c_p, c_p, C.c_size_t, c_p, c_i, c_i, c_p, c_i, c_p], c_i),
    "st_attn_backward": ([c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_f, c_p, c_p, C.c_size_t, c_p], c_i),
    "st_attn_reg_loss": ([c_p, c_i, c_i, c_i, c_f, c_p, c_p], c_i),
    "st_attn_greedy_workspace_bytes": ([c_p, c_i], C.c_size_t),
    "st_attn_greedy": ([c_p, c_p, c_i, c_i, c_l, c_p, C.c_size_t, c_p, c_p], c_i),
    "st_cross_entropy": ([c_p, c_i, c_p, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_f, c_p, c_p], c_i),
    "st_head_workspace_bytes": ([c_i, c_i, c_i, c_i], C.c_size_t),
    "st_linear_bn1d_forward": ([c_p] * 7 + [c_i, c_i, c_i, c_i, c_i, c_f, c_f] + [c_p] * 6, c_i),
    "st_linear_bn1d_backward": ([c_p] * 6 + [c_i, c_i, c_i, c_i, c_i] + [c_p] * 5 + [C.c_size_t, c_p], c_i),
    "st_sgd_step": ([c_p, c_p, c_p, c_p, c_l, c_f, c_f, c_i, c_f, c_p], c_i),
    "st_adam_step": ([c_p, c_p, c_p, c_p, c_p, c_l, c_f, c_f, c_f, c_f, c_i, c_f, c_p], c_i),
    "st_resnet_create": ([c_i, c_i, C.POINTER(c_p)], c_i),
    "st_resnet_destroy": ([c_p], None),
    "st_resnet_num_convs": ([c_p], c_i),
    "st_resnet_feat_dim": ([c_p], c_i),
    "st_resnet_weight_elems": ([c_p], C.c_size_t),
    "st_resnet_bn_channels": ([c_p], C.c_size_t),
    "st_resnet_conv_info": ([c_p, c_i] + [C.POINTER(c_i)] * 6 + [C.POINTER(C.c_size_t)] * 2 + [C.POINTER(c_i), C.POINTER(C.c_size_t), C.POINTER(c_i)], c_i),
    "st_resnet_workspace_bytes": ([c_p, c_i, c_i, c_i], C.c_size_t),
    "st_resnet_set_taps": ([c_p, c_p, C.c_size_t], c_i),
    "st_resnet_forward": ([c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_f, c_f, c_p, C.c_size_t,
                           c_p, c_p, c_i, c_p, c_p], c_i),
    "st_resnet_update_running": ([c_p, c_p, c_p, c_p, c_f, c_p], c_i),
}

_lib = None


class ShowTellHipError(RuntimeError):
    pass


def declared_symbols():
    """Every entry point include/showtell_hip.h declares (kept in sync by tests/test_abi.py)."""
    return sorted([k for k in _SIGS if k not in _EXPERIMENTAL] + ["st_last_error"])


_EXPERIMENTAL = ("st_conv1x1_kfuse8",)     # measured, not routed: absent from the product build


def has_symbol(name):
    return hasattr(lib(), name)


def lib():
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise ShowTellHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback for the HIP path)")
        # torch must load ITS HIP runtime first: the library then binds to the same libamdhip64 instance
        # (loading /opt/rocm's copy before torch's leaves the process with two runtimes and no device).
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        L.st_last_error.restype = C.c_char_p
        L.st_last_error.argtypes = []
        for name, (args, res) in _SIGS.items():
            if name in _EXPERIMENTAL and not hasattr(L, name):
                continue                            # `make EXPERIMENTAL=1` builds only (csrc/Makefile)
            fn = getattr(L, name)
            fn.argtypes, fn.restype = args, res
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        raise ShowTellHipError(f"{what}: {lib().st_last_error().decode()}")
