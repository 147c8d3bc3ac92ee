"""ctypes binding of libladder_hip.so (the C ABI declared in include/ladder_hip.h).

There is NO CPU fallback: if the shared library is missing or a call returns an error code the
product path raises.  `python -m ladder_latent_data_distribution_modelling_amd.csrc.build`
(or `__graft_entry__.build()`) produces the library with hipcc for gfx950.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libladder_hip.so")

ERRORS = {-1: "LADDER_E_SHAPE", -2: "LADDER_E_ALIGN", -3: "LADDER_E_WORKSPACE", -4: "LADDER_E_LAUNCH"}
ACT = {None: 0, "none": 0, "leaky_relu": 1, "relu": 2, "tanh": 3}

# partial / scalar slot indices (mirror include/ladder_hip.h)
P_PIX_ABS, P_PIX_SQ, P_LOG_SDZ, P_MU2SD2_Z, P_CODE_ERR, P_CODE_SQRT, P_CODE_ABS, P_LOG_SDT, P_MU2SD2_T, P_LOGP = range(10)
P_FIXED = 16
S_NAMES = ["sigma", "mean_pixel_error", "entropy_z", "crossEntropy_prior_sg", "crossEntropy_prior",
           "l1_reconstruction_error", "l2_reconstruction_error", "reconstruction_likelihood", "sigma_regularisor",
           "elbo", "loss_ae", "inner_sigma", "mean_code_error", "code_reconstruction_likelihood",
           "code_l1_reconstruction_error", "representation_regularisor", "entropy_t",
           "crossEntropy_representation", "elbo_prior", "loss_prior",
           "_g_pix", "_g_sigma_var", "_g_code", "_g_inner_sigma_var", "_inv_B", "_inv_LB"]
S_INDEX = {n: i for i, n in enumerate(S_NAMES)}
S_COUNT = 32
ABSMAX_FLOATS = 512      # LADDER_ABSMAX_FLOATS: size of an absolute-maximum record
ABI_VERSION = 2          # LADDER_ABI_VERSION this binding was written against (buffer layouts behind the entry points: include/ladder_hip.h)


class LadderElboCfg(C.Structure):
    _fields_ = [("B_global", C.c_int), ("D", C.c_int), ("Z", C.c_int), ("R", C.c_int), ("L", C.c_int),
                ("sigma_uses_mpe", C.c_int), ("has_inner", C.c_int), ("use_sg", C.c_int),
                ("clamp_inner_sigma", C.c_int), ("inner_sigma_lb", C.c_float), ("inner_sigma_ub", C.c_float),
                ("hierarchical", C.c_int), ("prior_gmm", C.c_int)]


_p, _i, _f, _d, _z, _u64 = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t, C.c_uint64

# name -> (restype, argtypes); every int-returning export is error-checked by `call`.
PROTOTYPES = {
    "ladder_abi_version": (_i, []),
    "ladder_igemm_fwd_tile": (_i, [C.c_long, _i, _i]),
    "ladder_dense_fwd_is_persistent": (_i, [C.c_long, _i, _i]),
    "ladder_dense_fwd_nt": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "ladder_dense_bwd_data_nt": (_i, [_p, _p, _p, _i, _i, _i, _p, _i, _p]),
    "ladder_dense_bwd_weight_is_persistent": (_i, [C.c_long, _i, _i]),
    "ladder_igemm_fwd_splits": (_i, [C.c_long, _i, _i]),
    "ladder_conv2d_fwd_kernel_id": (_i, [_i] * 13),
    "ladder_conv2d_bwd_filter_kernel_id": (_i, [_i] * 12),
    "ladder_conv2d_fwd": (_i, [_p, _p, _p, _p] + [_i] * 13 + [_p, _z, _p]),
    "ladder_igemm_fwd_workspace_bytes": (_z, [C.c_long, _i, _i]),
    "ladder_filter_flip_transpose": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "ladder_conv2d_bwd_data": (_i, [_p, _p, _p] + [_i] * 12 + [_p, _i, _p, _z, _p]),
    "ladder_conv2d_bwd_filter_workspace_bytes": (_z, [_i] * 9),
    "ladder_conv2d_bwd_filter": (_i, [_p, _p, _p, _p] + [_i] * 12 + [_p, _z, _p]),
    "ladder_dense_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _z, _p]),
    "ladder_dense_bwd_data": (_i, [_p, _p, _p, _i, _i, _i, _p, _i, _p, _z, _p]),
    "ladder_dense_bwd_weight_workspace_bytes": (_z, [_i, _i, _i]),
    "ladder_dense_bwd_weight": (_i, [_p, _p, _p, _p, _i, _i, _i, _p, _z, _p]),
    "ladder_act_bwd": (_i, [_p, _p, _p, _z, _i, _p]),
    "ladder_bn_workspace_bytes": (_z, [_z, _i]),
    "ladder_bn_fwd_stats": (_i, [_p, _p, _z, _i, _p, _z, _p]),
    "ladder_bn_fwd_apply": (_i, [_p, _p, _d, _p, _p, _p, _p, _z, _i, _f, _i, _p]),
    "ladder_bn_bwd_stats": (_i, [_p, _p, _p, _p, _p, _p, _z, _i, _i, _p, _z, _p]),
    "ladder_bn_bwd_apply": (_i, [_p, _p, _p, _p, _p, _p, _d, _p, _p, _p, _z, _i, _i, _p]),
    "ladder_in_style_workspace_bytes": (_z, [_i, _i, _i]),
    "ladder_in_style_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _f, _i, _p, _z, _p]),
    "ladder_in_style_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _z, _p]),
    "ladder_resize_bilinear_fwd": (_i, [_p, _p] + [_i] * 6 + [_p]),
    "ladder_resize_bilinear_bwd": (_i, [_p, _p] + [_i] * 6 + [_p]),
    "ladder_resize_bilinear_bwd_gated": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p, _i, _p]),
    "ladder_depth_to_space": (_i, [_p, _p] + [_i] * 6 + [_p]),
    "ladder_pad_symmetric": (_i, [_p, _p] + [_i] * 5 + [_p]),
    "ladder_randn": (_i, [_p, _z, _u64, _u64, _p]),
    "ladder_gmm_packed_stride": (_i, [_i]),
    "ladder_gmm_prepare": (_i, [_p, _p, _p, _i, _i, _p, _p]),
    "ladder_gmm_workspace_bytes": (_z, [_i, _i]),
    "ladder_gmm_logprob_fwd_bwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _z, _p]),
    "ladder_gmm_logprob_rows": (_i, [_p, _p, _i, _i, _i, _p, _p]),
    "ladder_gmm_dense_logprob_rows": (_i, [_p, _p, _i, _i, _i, _p, _p, _z, _p]),
    "ladder_pixel_partials_workspace_bytes": (_z, [_z]),
    "ladder_pixel_partials": (_i, [_p, _p, _z, _p, _p, _z, _p]),
    "ladder_pixel_grad": (_i, [_p, _p, _p, _p, _z, _p]),
    "ladder_latent_fwd": (_i, [_p, _p, _p, _f, _p, _p, _p, _p, _p, _i, _i, _p]),
    "ladder_code_partials": (_i, [_p, _p, _p, _i, _p, _i, _i, _p]),
    "ladder_elbo_finalize": (_i, [_p, _p, _p, LadderElboCfg, _p, _p]),
    "ladder_code_grad": (_i, [_p, _p, _p, _i, _p, _p, _p, _i, _i, _p]),
    "ladder_latent_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _f, _p, _i, _p, _p, _i, _i, _p]),
    "ladder_adam_clip": (_i, [_p, _p, _p, _p, _z, _f, _f, _f, _f, _f, _p]),
    "ladder_adam_clip_dev": (_i, [_p, _p, _p, _p, _z, _p, _f, _f, _f, _f, _p]),
    "ladder_randn_dev": (_i, [_p, _z, _u64, _p, _u64, _p]),
    "ladder_u64_add": (_i, [_p, _u64, _p]),
    "ladder_crc32c_extend": (C.c_uint32, [C.c_uint32, _p, _z]),
    "ladder_gmm_dense_param_floats": (_z, [_i, _i]),
    "ladder_gmm_prepare_dense": (_i, [_p, _p, _p, _i, _i, _p, _p]),
    "ladder_gmm_dense_workspace_bytes": (_z, [_i, _i, _i, _i]),
    "ladder_gmm_dense_logprob_fwd_bwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _z, _p]),
    "ladder_diag_mixture_workspace_bytes": (_z, [_i, _i, _i]),
    "ladder_diag_mixture_fwd_bwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _z, _p]),
    "ladder_pad_symmetric_bwd": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "ladder_conv1x1_smallcout_eligible": (_i, [C.c_long, _i, _i]),
    "ladder_conv1x1_smallcout_bwd_workspace_bytes": (_z, [C.c_long, _i, _i]),
    "ladder_conv1x1_smallcout_bwd": (_i, [_p, _p, _p, _p, _p, _p, C.c_long, _i, _i, _i, _p, _z, _p]),
    "ladder_gather_rows": (_i, [_p, _i, _p, _p, _i, C.c_int64, _f, _p]),
    "ladder_vbgmm_state_doubles": (_z, [_i, _i]),
    "ladder_vbgmm_workspace_bytes": (_z, [_i, _i]),
    "ladder_vbgmm_fit": (_i, [_p, _i, _i, _i, _p, _p, _i, _d, _d, _d, _d, _i, _p, _p, _p, _p, _z, _p]),
    "ladder_vbgmm_shard_stats_doubles": (_z, [_i, _i]),
    "ladder_vbgmm_shard_workspace_bytes": (_z, [_i, _i, _i]),
    "ladder_vbgmm_shard_moments_doubles": (_z, [_i]),
    "ladder_vbgmm_shard_moments": (_i, [_p, _i, _i, _p, _p]),
    "ladder_vbgmm_shard_estep": (_i, [_p, _i, _i, _i, _p, _p, _i, _p, _p, _z, _p]),
    "ladder_vbgmm_shard_mstep": (_i, [_p, _p, _i, _i, _p, _i, _d, _d, _d, _d, _i, _i, _p, _p, _p, _p]),
    "ladder_axpy": (_i, [_p, _p, _z, _f, _i, _p]),
    "ladder_filter_pack_split_bytes": (_z, [_i, _i, _i, _i]),
    "ladder_filter_pack_split": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "ladder_filter_pack_job_blocks": (_i, [_i, _i, _i]),
    "ladder_filter_pack_split_multi_scratch_bytes": (_z, [_i]),
    "ladder_filter_pack_split_multi": (_i, [_p, _i, _i, _i, _p, _z, _p]),
    "ladder_conv2d_fwd_split_eligible": (_i, [_i] * 12),
    "ladder_conv2d_fwd_split_workspace_bytes": (_z, [_i] * 12),
    "ladder_conv2d_fwd_split": (_i, [_p, _p, _p, _p, _p] + [_i] * 14 + [_p, _z, _p]),
    "ladder_conv2d_fwd_split_bnstats_workspace_bytes": (_z, [_i] * 12),
    "ladder_conv2d_fwd_split_bnstats": (_i, [_p] * 5 + [_i] * 14 + [_p, _p, _z, _p]),
    "ladder_conv2d_fwd_bnstats_workspace_bytes": (_z, [_i] * 12),
    "ladder_conv2d_fwd_bnstats": (_i, [_p, _p, _p, _p] + [_i] * 13 + [_p, _p, _z, _p]),
    "ladder_conv2d_bwd_data_split_eligible": (_i, [_i] * 13),
    "ladder_conv2d_bwd_data_split_workspace_bytes": (_z, [_i] * 12),
    "ladder_conv2d_bwd_data_split": (_i, [_p, _p, _p, _p] + [_i] * 12 + [_p, _i, _i, _p, _z, _p]),
    "ladder_conv3x3_split_eligible": (_i, [_i] * 5),
    "ladder_conv3x3_split": (_i, [_p, _p, _p, _p, _p, _p] + [_i] * 7 + [_p]),
    "ladder_conv3x3_s2_bwd_data_split_eligible": (_i, [_i] * 12),
    "ladder_conv3x3_up2_split_eligible": (_i, [_i] * 6),
    "ladder_conv3x3_up2_split": (_i, [_p, _p, _p, _p, _p, _p] + [_i] * 8 + [_p]),
    "ladder_conv3x3_up2_split_proj": (_i, [_p] * 8 + [_i] * 9 + [_p]),
    "ladder_conv3x3_up2_bwd_data_split_eligible": (_i, [_i] * 6),
    "ladder_conv3x3_up2_bwd_data_split": (_i, [_p] * 5 + [_i] * 6 + [_p]),
    "ladder_conv3x3_up2_bwd_border": (_i, [_p, _p, _p] + [_i] * 6 + [_p]),
    "ladder_conv3x3_up2_bwd_borders_workspace_bytes": (_z, [_i] * 5),
    "ladder_conv3x3_up2_bwd_borders": (_i, [_p, _p, _p] + [_i] * 5 + [_p, _z, _p]),
    "ladder_conv3x3_up2_wgrad_eligible": (_i, [_i] * 5),
    "ladder_conv3x3_up2_wgrad_workspace_bytes": (_z, [_i] * 5),
    "ladder_conv3x3_up2_wgrad": (_i, [_p, _i, _p, _p, _p] + [_i] * 5 + [_p, _z, _p]),
    "ladder_conv3x3_up2_edges_workspace_bytes": (_z, [_i] * 5),
    "ladder_conv3x3_up2_edges": (_i, [_p] * 8 + [_i] * 8 + [_p, _z, _p]),
    "ladder_conv3x3_s2_bwd_data_split": (_i, [_p, _p, _p, _p, _p] + [_i] * 8 + [_p]),
    "ladder_conv3x3_f32_eligible": (_i, [_i] * 5),
    "ladder_conv3x3_up2_bwd_data_gated_f32_eligible": (_i, [_i] * 5),
    "ladder_conv3x3_up2_bwd_data_gated_f32": (_i, [_p, _p, _p, _p] + [_i] * 6 + [_p]),
    "ladder_conv3x3_up2_bwd_borders_gated": (_i, [_p, _p, _p, _p] + [_i] * 6 + [_p, _z, _p]),
    "ladder_conv3x3_s2_bwd_data_f32_eligible": (_i, [_i] * 7),
    "ladder_conv3x3_s2_fwd_f32_eligible": (_i, [_i] * 7),
    "ladder_conv3x3_s2_fwd_f32": (_i, [_p, _p, _p, _p] + [_i] * 8 + [_p]),
    "ladder_up2proj_eligible": (_i, [_i] * 5),
    "ladder_up2proj_fwd_combine": (_i, [_p] * 6 + [_i] * 6 + [_p]),
    "ladder_up2proj_bwd_combine": (_i, [_p, _p] + [_i] * 4 + [_p]),
    "ladder_up2proj_wgrad_unpack": (_i, [_p] * 4 + [_i, _i, _p]),
    "ladder_upfproj_eligible": (_i, [_i] * 6),
    "ladder_upfproj_fwd_combine": (_i, [_p] * 3 + [_i] * 6 + [_p]),
    "ladder_upfproj_bwd_combine": (_i, [_p, _p] + [_i] * 5 + [_p]),
    "ladder_up2proj_bwd_combine_walk": (_i, [_p, _p] + [_i] * 5 + [_p]),
    "ladder_up2proj_bwd_combine_proj_eligible": (_i, [_i] * 5),
    "ladder_up2proj_bwd_combine_proj_workspace_bytes": (_z, [_i] * 5),
    "ladder_up2proj_bwd_combine_proj": (_i, [_p] * 6 + [_i] * 6 + [_p, _z, _p]),
    "ladder_up2proj_fused_eligible": (_i, [_i] * 5),
    "ladder_up2proj_fused_preferred": (_i, [_i] * 5),
    "ladder_up2proj_fused_wide_tile": (_i, [_i] * 5),
    "ladder_up2proj_fused_workspace_bytes": (_z, [_i] * 5),
    "ladder_up2proj_fused_fwd": (_i, [_p] * 7 + [_i] * 7 + [_p, _z, _p]),
    "ladder_dense_small_eligible": (_i, [_i, _i, _i]),
    "ladder_dense_fwd_small": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "ladder_dense_bwd_data_small": (_i, [_p, _p, _p, _i, _i, _i, _p, _i, _p]),
    "ladder_dense_bwd_weight_small": (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    "ladder_dense_bwd_small": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _i, _p]),
    "ladder_dense_fwd_small_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "ladder_dense_bwd_data_small_f32": (_i, [_p, _p, _p, _i, _i, _i, _p, _i, _p]),
    "ladder_dense_bwd_weight_small_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    "ladder_dense_bwd_small_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _i, _p]),
    "ladder_presplit_bytes": (_z, [_z, _i]),
    "ladder_presplit": (_i, [_p, _p, _p, _z, _i, _i, _p]),
    "ladder_absmax_samples": (_i, [_p, _i, _z, _p, _p]),
    "ladder_conv3x3_split_proj": (_i, [_p, _p, _p, _p, _p, _p, _p, _p] + [_i] * 8 + [_p]),
    "ladder_conv_rgb_s2_eligible": (_i, [_i] * 10),
    "ladder_conv_rgb_s2_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "ladder_conv_rgb_s2_fwd_bnstats_workspace_bytes": (_z, [_i] * 4),
    "ladder_conv_rgb_s2_fwd_bnstats": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _z, _p]),
    "ladder_conv_rgb_s2_fwd_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "ladder_conv_rgb_s2_fwd_bnstats_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _z, _p]),
    "ladder_bn_stats_from_partials": (_i, [_p, _i, _p, _i, _p]),
    "ladder_bn_stats_minmax_from_partials": (_i, [_p, _i, _p, _i, _p]),
    "ladder_bn_fwd_stats_minmax": (_i, [_p, _p, _z, _i, _p, _z, _p]),
    "ladder_bn_fwd_apply_planes": (_i, [_p, _p, _d, _p, _p, _p, _p, _p, _z, _i, _f, _i, _p, _p]),
    "ladder_conv_rgb_s2_bwd_filter_workspace_bytes": (_z, [_i] * 4),
    "ladder_conv_rgb_s2_bwd_filter": (_i, [_p] * 6 + [_i] * 4 + [_p, _z, _p]),
    "ladder_conv_rgb_s2_bwd_filter_f32": (_i, [_p] * 4 + [_i] * 4 + [_p, _z, _p]),
    "ladder_bn_fwd_apply_absmax": (_i, [_p, _p, _d, _p, _p, _p, _p, _z, _i, _f, _i, _p, _p]),
    "ladder_bn_bwd_apply_absmax": (_i, [_p, _p, _p, _p, _p, _p, _d, _p, _p, _p, _z, _i, _i, _p, _p]),
    "ladder_in_style_fwd_absmax": (_i, [_p, _p, _p, _p, _i, _i, _i, _f, _i, _p, _z, _p, _p]),
    "ladder_in_style_fwd_resize2x": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _p, _z, _p, _p]),
    "ladder_in_style_fwd_resize2x_keep": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _p, _z, _p, _p]),
    "ladder_in_style_bwd_absmax": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _z, _p, _p]),
    "ladder_conv1x1_smallcout_bwd_absmax": (_i, [_p, _p, _p, _p, _p, _p, C.c_long, _i, _i, _i, _p, _z, _p, C.c_long, _p]),
    "ladder_absmax": (_i, [_p, _z, _p, _p]),
    "ladder_conv3x3_wgrad_split_eligible": (_i, [_i] * 6),
    "ladder_conv3x3_wgrad_split_workspace_bytes": (_z, [_i] * 5),
    "ladder_conv3x3_wgrad_split": (_i, [_p, _p, _p, _p, _p, _p] + [_i] * 6 + [_p, _z, _p]),
    "ladder_reduce_splits": (_i, [_p, _p, _i, _z, _p]),
    "ladder_conv2d_bwd_filter_split_eligible": (_i, [_i] * 12),
    "ladder_conv2d_bwd_filter_split_workspace_bytes": (_z, [_i] * 9),
    "ladder_conv2d_bwd_filter_split": (_i, [_p, _p, _p, _p, _p, _p] + [_i] * 13 + [_p, _z, _p]),
}

_lib = None


class LadderHipError(RuntimeError):
    pass


def load(path=None):
    """Load (once) and return the ctypes handle; raises LadderHipError if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or os.environ.get("LADDER_HIP_LIB", LIB_PATH)
    # PyTorch-ROCm is the device-memory / stream provider: its HIP runtime must be the one this library binds to,
    # so make sure it is loaded first (loading libamdhip64 twice, or ours first, breaks kernel launches).
    import torch  # noqa: F401
    if not os.path.exists(path):
        raise LadderHipError(
            "libladder_hip.so not found at %s: build it with `python -m "
            "ladder_latent_data_distribution_modelling_amd.csrc.build` (there is no CPU fallback)" % path)
    lib = C.CDLL(path)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)      # AttributeError here == header/library mismatch
        fn.restype = res
        fn.argtypes = args
    got = lib.ladder_abi_version()
    if got != ABI_VERSION:
        raise LadderHipError("%s reports ABI version %d, this binding needs %d (stale build? re-run `python -m "
                             "ladder_latent_data_distribution_modelling_amd.csrc.build`)" % (path, got, ABI_VERSION))
    _lib = lib
    return lib


def call(name, *args):
    """Invoke an int-returning export and raise on a non-zero status."""
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise LadderHipError("%s failed: %s (%d)" % (name, ERRORS.get(rc, "?"), rc))


def query(name, *args):
    return getattr(load(), name)(*args)
