"""ctypes binding of libsdn.so (C ABI: include/sdn.h).  Fails loudly -- there is no fallback path."""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class SdnUnavailable(RuntimeError):
    """libsdn.so is missing/unloadable, or there is no gfx950 device for a compute call."""


class SdnError(RuntimeError):
    """A libsdn entry point returned a negative status."""


_ERR = {-1: "SDN_E_INVALID (bad shape / null or misaligned pointer)", -2: "SDN_E_LAUNCH", -3: "SDN_E_WORKSPACE",
        -4: "SDN_E_ARCH"}


def lib_path() -> str:
    return os.environ.get("SDN_LIB", os.path.join(_HERE, "libsdn.so"))


class RepelParams(C.Structure):
    _fields_ = [("n_query", C.c_int32), ("n_ref", C.c_int32), ("channels", C.c_int32), ("hw", C.c_int32),
                ("weight_fn", C.c_int32), ("qnorm", C.c_int32), ("sigma", C.c_float), ("radius", C.c_float),
                ("scale", C.c_float), ("epsilon", C.c_float), ("gate", C.c_float)]


class GemmDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("M", "N", "K", "a_mode", "K1", "Hs", "Ws", "Cin", "Ho", "Wo", "stride",
                                         "upsample", "act", "out_kind", "rows_per_batch", "ld_rowbias", "ld_rowgate",
                                         "residual_bcast", "n_valid", "ldc", "asym_pad", "split_k", "res_pre", "x3_out")]


class UnetConfig(C.Structure):
    _fields_ = [("in_channels", C.c_int32), ("out_channels", C.c_int32), ("sample_size", C.c_int32),
                ("n_levels", C.c_int32), ("block_out_channels", C.c_int32 * 4), ("level_has_attn", C.c_int32 * 4),
                ("layers_per_block", C.c_int32), ("n_heads", C.c_int32), ("cross_dim", C.c_int32),
                ("text_len", C.c_int32), ("norm_groups", C.c_int32), ("dtype", C.c_int32), ("latent_repeat", C.c_int32)]


class ProfileRow(C.Structure):
    _fields_ = [("kernel", C.c_char * 24), ("launches", C.c_int32), ("ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double)]


class MmditConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("in_channels", "out_channels", "sample_size", "patch_size", "num_layers",
                                         "num_heads", "head_dim", "joint_dim", "pooled_dim", "text_len", "time_dim",
                                         "dtype")]


class VaeConfig(C.Structure):
    _fields_ = [("latent_channels", C.c_int32), ("out_channels", C.c_int32), ("sample_size", C.c_int32),
                ("n_levels", C.c_int32), ("block_out_channels", C.c_int32 * 4), ("layers_per_block", C.c_int32),
                ("norm_groups", C.c_int32), ("dtype", C.c_int32)]


class ClipConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("vocab_size", "hidden_size", "intermediate_size", "num_layers", "num_heads",
                                         "max_position_embeddings", "dtype")]


class AttnSegment2(C.Structure):
    _fields_ = [("q2", C.c_void_p), ("k2", C.c_void_p), ("v2", C.c_void_p), ("out2", C.c_void_p), ("n1", C.c_int32),
                ("ldq2", C.c_int32), ("ldk2", C.c_int32), ("ldv2", C.c_int32), ("ldo2", C.c_int32)]


class ParamInfo(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("kind", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32),
                ("rows_padded", C.c_int32), ("offset", C.c_int64)]


_vp, _i32, _i64, _f32, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t

# name -> (restype, argtypes).  Must list every symbol include/sdn.h declares (tests check this).
SIGNATURES = {
    "sdn_abi_version": (C.c_int, []),
    "sdn_device_arch_host": (C.c_char_p, []),
    "sdn_repel_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "sdn_repel_apply": (C.c_int, [C.POINTER(RepelParams), _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sdn_repel_calibrate": (C.c_int, [C.POINTER(RepelParams), _vp, _vp, _vp, _vp, _sz, _vp]),
    "sdn_cfg_combine": (C.c_int, [_vp, _i32, _i32, _i64, _f32, _vp, _vp]),
    "sdn_sld_guidance": (C.c_int, [_vp, _i32, _i64, _f32, _f32, _f32, _f32, _f32, _i32, _vp, _vp, _vp]),
    "sdn_cfg_combine_rows": (C.c_int, [_vp, _i32, _i32, _i64, _vp, _vp, _vp]),
    "sdn_sld_guidance_rows": (C.c_int, [_vp, _i32, _i64, _vp, _f32, _f32, _f32, _f32, _i32, _vp, _vp, _vp]),
    "sdn_pred_x0": (C.c_int, [_vp, _vp, _i64, _f32, _f32, _f32, _vp, _vp]),
    "sdn_sched_step": (C.c_int, [_vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _vp, _vp]),
    "sdn_add_noise": (C.c_int, [_vp, _vp, _i64, _f32, _f32, _vp, _vp]),
    "sdn_renoise_select": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i64, _f32, _f32, _vp]),
    "sdn_flow_euler_step": (C.c_int, [_vp, _vp, _i64, _f32, _f32, _vp, _vp]),
    "sdn_flow_endpoints": (C.c_int, [_vp, _vp, _i64, _f32, _vp, _vp, _vp]),
    "sdn_flow_renoise": (C.c_int, [_vp, _vp, _vp, _i64, _f32, _vp, _vp]),
    "sdn_gemm_bf16": (C.c_int, [C.POINTER(GemmDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sdn_gemm_f16": (C.c_int, [C.POINTER(GemmDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sdn_ln_fold": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp]),
    "sdn_gemm_ln_bf16": (C.c_int, [C.POINTER(GemmDesc), _vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp]),
    "sdn_gemm_ln_f16": (C.c_int, [C.POINTER(GemmDesc), _vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp]),
    "sdn_row_stats_bf16": (C.c_int, [_vp, _i64, _i32, _f32, _vp, _vp]),
    "sdn_row_stats_f16": (C.c_int, [_vp, _i64, _i32, _f32, _vp, _vp]),
    "sdn_ffn_geglu_fused": (C.c_int, [_i32, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sdn_gemm_stats_bf16": (C.c_int, [C.POINTER(GemmDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sdn_gemm_stats_f16": (C.c_int, [C.POINTER(GemmDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sdn_groupnorm_cols_bf16": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sdn_groupnorm_cols_f16": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sdn_gemm_splitk_bf16": (C.c_int, [C.POINTER(GemmDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sdn_gemm_splitk_f16": (C.c_int, [C.POINTER(GemmDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sdn_groupnorm_bf16": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "sdn_groupnorm_f16": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "sdn_layernorm_f16": (C.c_int, [_vp, _i64, _i32, _f32, _vp, _vp, _vp, _vp]),
    "sdn_conv_in_f16": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "sdn_timestep_embed_f16": (C.c_int, [_f32, _i32, _i32, _vp, _vp]),
    "sdn_layernorm_bf16": (C.c_int, [_vp, _i64, _i32, _f32, _vp, _vp, _vp, _vp]),
    "sdn_attention_bf16": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32,
                                     _vp]),
    "sdn_attention_f16": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32,
                                    _vp]),
    "sdn_conv_in_bf16": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "sdn_timestep_embed_bf16": (C.c_int, [_f32, _i32, _i32, _vp, _vp]),
    "sdn_gemm_f32": (C.c_int, [C.POINTER(GemmDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sdn_groupnorm_f32": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "sdn_layernorm_f32": (C.c_int, [_vp, _i64, _i32, _f32, _vp, _vp, _vp, _vp]),
    "sdn_attention_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp]),
    "sdn_randn_philox_plan": (C.c_int, [C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "sdn_randn_philox": (C.c_int, [_vp, _vp, _vp, _i32, C.c_int64, _vp, _vp]),
    "sdn_randn_philox_state": (C.c_int, [_vp, _vp, _vp, _i32, C.c_int64, _vp, _vp]),
    "sdn_gemm_x3": (C.c_int, [C.POINTER(GemmDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sdn_attention_x3": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp]),
    "sdn_conv_in_f32": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "sdn_timestep_embed_f32": (C.c_int, [_f32, _i32, _i32, _vp, _vp]),
    "sdn_unet_create": (C.c_int, [C.POINTER(UnetConfig), C.POINTER(_vp)]),
    "sdn_unet_destroy": (None, [_vp]),
    "sdn_unet_param_count": (C.c_int, [_vp]),
    "sdn_unet_param_info": (C.c_int, [_vp, _i32, C.POINTER(ParamInfo)]),
    "sdn_unet_weight_bytes": (_sz, [_vp]),
    "sdn_unet_workspace_bytes": (_sz, [_vp, _i32]),
    "sdn_unet_flops": (C.c_double, [_vp, _i32, C.POINTER(C.c_double)]),
    "sdn_unet_forward": (C.c_int, [_vp, _vp, _vp, _f32, _vp, _vp, _i32, _vp, _sz, _vp]),
    "sdn_mmdit_create": (C.c_int, [C.POINTER(MmditConfig), C.POINTER(_vp)]),
    "sdn_mmdit_forward": (C.c_int, [_vp, _vp, _vp, _f32, _vp, _vp, _vp, _i32, _vp, _sz, _vp]),
    "sdn_joint_attention": (C.c_int, [_i32, _vp, _vp, _vp, _vp, C.POINTER(AttnSegment2), _i32, _i32, _i32, _i32, _i32,
                                      _i32, _i32, _i32, _f32, _vp]),
    "sdn_layernorm_mod_bf16": (C.c_int, [_vp, _i64, _i32, _f32, _vp, _vp, _i32, _i32, _vp, _vp]),
    "sdn_layernorm_mod_f16": (C.c_int, [_vp, _i64, _i32, _f32, _vp, _vp, _i32, _i32, _vp, _vp]),
    "sdn_patchify_bf16": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "sdn_patchify_f16": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "sdn_unpatchify_f32": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "sdn_repeat": (C.c_int, [_vp, _sz, _i32, _vp, _vp]),
    "sdn_vae_decoder_create": (C.c_int, [C.POINTER(VaeConfig), C.POINTER(_vp)]),
    "sdn_vae_decode": (C.c_int, [_vp, _vp, _vp, _f32, _vp, _i32, _vp, _sz, _vp]),
    "sdn_vae_encoder_create": (C.c_int, [C.POINTER(VaeConfig), C.POINTER(_vp)]),
    "sdn_vae_encode": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _vp, _sz, _vp]),
    "sdn_gaussian_sample": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _f32, _vp, _vp]),
    "sdn_image_postprocess": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "sdn_latent_mix": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp, _vp]),
    "sdn_softmax_rows": (C.c_int, [_i32, _vp, _i64, _i64, _i32, _f32, _vp, _i64, _vp]),
    "sdn_transpose16": (C.c_int, [_vp, _i32, _i32, _i64, _vp, _i64, _vp]),
    "sdn_clip_create": (C.c_int, [C.POINTER(ClipConfig), C.POINTER(_vp)]),
    "sdn_clip_forward": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _vp, _sz, _vp]),
    "sdn_clip_embed": (C.c_int, [_i32, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp]),
    "sdn_masked_attention": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                                       _f32, _vp]),
    "sdn_split3": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp, _vp]),
    "sdn_expand3_weights": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _vp]),
    "sdn_groupnorm_f32_triple": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "sdn_layernorm_f32_triple": (C.c_int, [_vp, _i64, _i32, _f32, _vp, _vp, _vp, _vp]),
    "sdn_attention_x3_triple": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp]),
    "sdn_attention_x3_pairs": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _vp]),
    "sdn_clip_embed_f32": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp]),
    "sdn_masked_attention_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp]),
    "sdn_unet_prepare": (C.c_int, [_vp, _vp, _vp]),
    "sdn_unet_set_graph_mode": (None, [_vp, _i32]),
    "sdn_unet_set_text_version": (None, [_vp, C.c_uint64]),
    "sdn_unet_set_split_k": (None, [_vp, _i32]),
    "sdn_unet_profile_next": (None, [_vp]),
    "sdn_unet_profile_read": (C.c_int, [_vp, C.POINTER(ProfileRow), _i32]),
}


def lib():
    """The loaded library (cached).  Raises SdnUnavailable when it cannot be loaded."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise SdnUnavailable(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                                 f"(or `make -C safe_denoiser_amd/csrc`).  There is no CPU fallback.")
        try:
            handle = C.CDLL(path)
        except OSError as e:                                         # pragma: no cover
            raise SdnUnavailable(f"cannot load {path}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _LIB = handle
    return _LIB


def check(rc: int, what: str):
    if rc != 0:
        raise SdnError(f"{what} failed: {_ERR.get(rc, rc)}")


def require_gpu():
    if not torch.cuda.is_available():
        raise SdnUnavailable("no GPU visible: the safe-denoiser hot path only runs on an MI355X (gfx950); "
                             "there is no CPU fallback")


def dptr(t: torch.Tensor | None, dtype=None) -> int | None:
    """Device pointer of a contiguous CUDA(HIP) tensor, with loud checks."""
    if t is None:
        return None
    if not t.is_cuda:
        raise SdnUnavailable("tensor is not on the GPU: libsdn takes device pointers only (no CPU fallback)")
    if not t.is_contiguous():
        raise SdnError("tensor must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise SdnError(f"expected {dtype}, got {t.dtype}")
    return t.data_ptr()


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream
