"""ctypes binding of `libfmc_hip.so` (the C ABI declared in `include/fmc_hip.h`).

The library is the product: there is no CPU or PyTorch fallback.  If it is missing
or a kernel reports an error, the call raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FMC_HIP_LIB") or os.path.join(_HERE, "lib", "libfmc_hip.so")   # env: another BUILD of the same library (A/B timing)

FMC_BF16, FMC_F32 = 0, 1

_ERRORS = {-1: ValueError, -2: TypeError, -3: ValueError, -4: RuntimeError, -5: ValueError}

# name -> (restype, argtypes); mirrors include/fmc_hip.h one to one
SIGNATURES = {
    "fmc_version": (c_int, []),
    "fmc_last_error": (c_char_p, []),
    "fmc_groupnorm_workspace_bytes": (c_int64, [c_int, c_int, c_int]),
    "fmc_groupnorm_silu_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                       c_int, c_float, c_int, c_int, c_void_p, c_int, c_void_p]),
    "fmc_groupnorm_silu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                       c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "fmc_layernorm_add_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_int,
                                      c_int, c_int, c_void_p]),
    "fmc_layernorm_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_int,
                                  c_int, c_int, c_void_p]),
    "fmc_geglu_fwd": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "fmc_spatial_attn_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                     c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, c_int, c_float, c_int,
                                     c_void_p]),
    "fmc_temporal_attn_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                      c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, c_float, c_int, c_void_p]),
    "fmc_plucker_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "fmc_omc_rasterize_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                      c_void_p]),
    "fmc_gaussian_circle_mask_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "fmc_mask_modulate_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                      c_int, c_void_p]),
    "fmc_feature_add_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    "fmc_cfg_ddim_step": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_float, c_float, c_int,
                                  c_void_p]),
    "fmc_linear_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int64,
                                c_int64, c_int64, c_float, c_int, c_int, c_int, c_void_p, c_int64, c_void_p, c_int64, c_int,
                                c_void_p, c_void_p]),
    "fmc_linear_bf16_gn": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int64, c_int64, c_int64,
                                   c_float, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "fmc_linear_bf16_ln": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int64, c_int64, c_int64,
                                   c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
    "fmc_linear_bf16_lnc": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                    c_int, c_void_p]),
    "fmc_linear_bf16_ffblk": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int64, c_float, c_int, c_int, c_int,
                                      c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "fmc_groupnorm_fold_linear": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                          c_float, c_int, c_void_p]),
    "fmc_linear_bf16_imgw": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int64, c_int64, c_int, c_void_p, c_float, c_int,
                                     c_void_p]),
    "fmc_linear_bf16_fftail": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int64, c_int64,
                                       c_void_p, c_int, c_int, c_void_p]),
    "fmc_conv3x3_bf16_gn": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                    c_int64, c_int, c_int, c_void_p, c_int, c_void_p]),
    "fmc_groupnorm_partial_splits": (c_int, [c_int, c_int]),
    "fmc_groupnorm_partials": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "fmc_conv3x3_halo_supported": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "fmc_conv3x3_halo_packed_bytes": (c_int64, [c_int, c_int]),
    "fmc_conv3x3_halo_pack_weight": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "fmc_conv3x3_halo_tiles_per_image": (c_int, [c_int, c_int]),
    "fmc_conv3x3_halo_bf16": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                      c_int64, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "fmc_conv3x3_halo4_supported": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "fmc_conv3x3_halo4_pack_weight": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "fmc_conv3x3_halo4_row_blocks_per_image": (c_int, [c_int, c_int]),
    "fmc_conv3x3_halo4_tiles": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "fmc_conv3x3_halo4_bf16": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                       c_int64, c_int, c_int, c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p]),
    "fmc_linear4_supported": (c_int, [c_int64, c_int, c_int, c_int64]),
    "fmc_vendor_linear_candidates": (c_int, [c_int64, c_int, c_int, c_int64, c_int64, c_int64, c_int, c_int]),
    "fmc_vendor_linear_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int64, c_int64, c_int64, c_int, c_void_p,
                                       c_int64, c_void_p]),
    "fmc_geglu_pipe_supported": (c_int, [c_int64, c_int, c_int, c_int]),
    "fmc_geglu_pipe_ln_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    "fmc_attention_supported": (c_int, [c_int]),
    "fmc_attention_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int64, c_int64, c_int64, c_int64,
                                  c_int64, c_int64, c_float, c_int, c_int, c_void_p]),
    "fmc_vendor_init": (c_int, []),
    "fmc_vendor_destroy": (c_int, []),
    "fmc_vendor_version": (c_int, []),
    "fmc_vendor_workspace_bytes": (c_int64, []),
    "fmc_linear4_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int64, c_int64, c_int64, c_float, c_void_p]),
    "fmc_groupnorm_coef": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "fmc_groupnorm_apply_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                        c_float, c_int, c_int, c_void_p]),
    "fmc_split_bf16x3": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int64, c_int64, c_int, c_int, c_int, c_void_p]),
    "fmc_linear_x3_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int64,
                                  c_int64, c_int64, c_float, c_int, c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p]),
    "fmc_conv3x3_x3_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                   c_int, c_int, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_int64, c_void_p]),
    "fmc_layernorm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float,
                                  c_int, c_void_p]),
    "fmc_layernorm_bwd_add": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float,
                                      c_int, c_void_p]),
    "fmc_groupnorm_silu_bwd_add": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                           c_int, c_void_p, c_int, c_void_p]),
    "fmc_geglu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "fmc_spatial_attn_bwd": (c_int, [c_void_p] * 10 + [c_int] * 5 + [c_int64] * 10 + [c_int, c_float, c_int, c_void_p]),
    "fmc_temporal_attn_bwd": (c_int, [c_void_p] * 7 + [c_int] * 5 + [c_int64] * 9 + [c_float, c_int, c_void_p]),
    "fmc_fp8_scales_roll": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p]),
    "fmc_linear_fp8_qkv": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int64, c_void_p, c_void_p, c_void_p]),
    "fmc_temporal_attn_fp8_fwd": (c_int, [c_void_p] * 5 + [c_int] * 5 + [c_int64] * 6 + [c_float, c_void_p]),
    "fmc_temporal_attn_fp8_bwd": (c_int, [c_void_p] * 8 + [c_int] * 5 + [c_int64] * 9 + [c_float, c_void_p]),
    "fmc_nhwc_to_cmajor_padded": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int64, c_int, c_int, c_void_p]),
    "fmc_temporal_block_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_float, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "fmc_temporal_block_set_debug": (c_int, [c_void_p]),
    "fmc_xattn_block640_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                        c_float, c_void_p]),
    "fmc_xattn_pack_kv": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p]),
    "fmc_geglu320_ln_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "fmc_geglu640_ln_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "fmc_xattn_pack_kv40": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p]),
    "fmc_xattn_block320_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                        c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "fmc_conv3x3_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                 c_int, c_int, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_int64, c_void_p]),
}

_lib = None


class FmcLibraryMissing(ImportError):
    pass


def load() -> ctypes.CDLL:
    """Load (once) and type the shared library.  Raises `FmcLibraryMissing` when it has not been built
    (`python -c "import __graft_entry__ as g; g.build()"` or `make -C synfmc_amd/csrc`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise FmcLibraryMissing(
            f"{LIB_PATH} not found: build it with `make -C synfmc_amd/csrc` (hipcc --offload-arch=gfx950). "
            "There is no fallback path.")
    # PyTorch-ROCm bundles its own libamdhip64.so.7; it must be the HIP runtime this library binds to (streams and
    # device pointers are torch's), so make sure torch's copy is the one already mapped when we dlopen.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().fmc_last_error()
        raise _ERRORS.get(rc, RuntimeError)(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")
