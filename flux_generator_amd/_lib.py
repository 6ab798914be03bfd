"""ctypes binding of libfluxhip.so (C ABI declared in include/fluxhip.h).

There is no fallback: if the shared library is missing or was not built for gfx950 the import of
any op raises.  The product path never routes through PyTorch math or the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("FLUXHIP_LIB", _HERE / "lib" / "libfluxhip.so"))

c_void_p, c_int, c_int64, c_float = C.c_void_p, C.c_int, C.c_int64, C.c_float


class GemmGroup(C.Structure):
    _fields_ = [
        ("A", c_void_p), ("W", c_void_p), ("bias", c_void_p), ("C", c_void_p),
        ("res", c_void_p), ("gate", c_void_p),
        ("a_bstride", c_int64), ("c_bstride", c_int64), ("gate_bstride", c_int64), ("w_bstride", c_int64),
        ("M", C.c_int32), ("_pad", C.c_int32),
        ("add", c_void_p), ("add_bstride", c_int64),
    ]


class GemmDesc(C.Structure):
    _fields_ = [
        ("g", GemmGroup * 2),
        ("ngroups", C.c_int32), ("nbatch", C.c_int32),
        ("N", C.c_int32), ("K", C.c_int32),
        ("lda", C.c_int32), ("ldc", C.c_int32),
        ("epi", C.c_int32), ("row_bias", C.c_int32),
        ("n_split", C.c_int32), ("ldc2", C.c_int32),
        ("C2", c_void_p), ("c2_bstride", c_int64),
        ("c2_coloff", C.c_int32), ("tile_cfg", C.c_int32),
        ("alpha", c_float), ("out_f32", C.c_int32),
        ("ld_add", C.c_int32), ("_pad2", C.c_int32),
    ]


class GemmX3Desc(C.Structure):
    _fields_ = [
        ("A", c_void_p), ("W", c_void_p), ("bias", c_void_p), ("C", c_void_p), ("res", c_void_p),
        ("a_lo", c_int64), ("w_lo", c_int64), ("c_lo", c_int64), ("res_lo", c_int64),
        ("a_bstride", c_int64), ("c_bstride", c_int64), ("w_bstride", c_int64),
        ("M", C.c_int32), ("nbatch", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("lda", C.c_int32), ("ldc", C.c_int32),
        ("epi", C.c_int32), ("row_bias", C.c_int32), ("out_f32", C.c_int32), ("tile_cfg", C.c_int32),
        ("alpha", c_float), ("_pad", C.c_int32),
    ]


class Fp8Scales(C.Structure):
    _fields_ = [("a_scale", c_void_p * 2), ("w_scale", c_void_p * 2), ("a_scale_bstride", c_int64)]


class Fp8Mx(C.Structure):      # fluxhip_fp8_mx
    _fields_ = [("a_mx", c_void_p), ("a_mx_row0", C.c_int32 * 2), ("a_mx_bstride", c_int64), ("a_mx_kstride", c_int64),
                ("c8", c_void_p * 2), ("c8_bstride", c_int64), ("ldc8", C.c_int32), ("c8_coloff", C.c_int32),
                ("c_mx", c_void_p), ("c_mx_row0", C.c_int32 * 2), ("c_mx_bstride", c_int64), ("c_mx_kstride", c_int64)]


# name -> (restype, argtypes); every symbol include/fluxhip.h declares
SIGNATURES = {
    "fluxhip_abi_version": (c_int, []),
    "fluxhip_arch": (C.c_char_p, []),
    "fluxhip_gemm_bf16": (c_int, [C.POINTER(GemmDesc), c_void_p]),
    "fluxhip_gemm_tile_cfg": (c_int, [C.POINTER(GemmDesc)]),
    "fluxhip_gemm_tile_shape": (c_int, [c_int, C.POINTER(c_int), C.POINTER(c_int), C.POINTER(c_int)]),
    "fluxhip_gemm_set_trace": (c_int, [c_void_p]),
    "fluxhip_conv_set_x3_tile": (c_int, [c_int, c_int]),
    "fluxhip_conv_dxr_launches": (C.c_int64, []),
    "fluxhip_gemm_set_splitk_mode": (c_int, [c_int]),
    "fluxhip_attention_set_variant": (c_int, [c_int]),
    "fluxhip_gemm_rs_launches": (C.c_int64, []),
    "fluxhip_gemm_set_rs_timeout_us": (c_int, [c_int]),
    "fluxhip_gemm_set_lean": (c_int, [c_int]),
    "fluxhip_gemm_lean_launches": (C.c_int64, []),
    "fluxhip_set_workspace": (c_int, [c_void_p, c_int64]),
    "fluxhip_conv2d_bf16": (c_int, [c_void_p] * 6 + [c_int] * 10 + [c_void_p, c_void_p]),
    "fluxhip_conv2d_small": (c_int, [c_void_p] * 4 + [c_int] * 7 + [c_void_p]),
    "fluxhip_small_linear_bf16": (c_int, [c_void_p] * 4 + [c_int] * 5 + [c_void_p]),
    "fluxhip_ln_modulate_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int64, c_int64,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_void_p]),
    "fluxhip_qk_norm_rope_bf16": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                          c_int, c_float, c_void_p]),
    "fluxhip_attention_d128_bf16": (c_int, [c_void_p] * 4 + [c_int] * 5 + [c_float, c_void_p]),
    "fluxhip_silu_bf16": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "fluxhip_timestep_embedding_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_void_p]),
    "fluxhip_rope_table_bf16": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "fluxhip_euler_step_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_void_p]),
    "fluxhip_pack_latents_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "fluxhip_unpack_latents_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_void_p]),
    "fluxhip_groupnorm_silu_bf16": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_float, c_int, c_void_p, c_int64, c_void_p]),
    "fluxhip_attention_strided_bf16": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64,
                                               c_void_p, c_void_p] + [c_int] * 7 + [c_float, c_void_p]),
    "fluxhip_attention_strided_vt_bf16": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64,
                                                  c_void_p, c_int64, c_void_p] + [c_int] * 7 + [c_float, c_void_p]),
    "fluxhip_layernorm_affine_bf16": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_float, c_void_p]),
    "fluxhip_concat_channels_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "fluxhip_axpbypcz_bf16": (c_int, [c_void_p] * 4 + [c_int64, c_float, c_float, c_float, c_void_p]),
    "fluxhip_axpbypcz_dev_bf16": (c_int, [c_void_p] * 4 + [c_int64, c_void_p, c_void_p]),
    "fluxhip_pixel_linear_bf16": (c_int, [c_void_p] * 4 + [c_int64, c_int, c_int, c_int, c_float, c_void_p]),
    "fluxhip_sincos_embed_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "fluxhip_attention_masked_bf16": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64,
                                              c_void_p, c_void_p] + [c_int] * 6 + [c_float, c_void_p, c_int, c_void_p]),
    "fluxhip_rmsnorm_bf16": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_float, c_void_p]),
    "fluxhip_embedding_bf16": (c_int, [c_void_p] * 4 + [c_int64, c_int, c_int, c_int, c_void_p]),
    "fluxhip_softmax_rows_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_float, c_void_p]),
    # fp8 path
    "fluxhip_quantize_rows_fp8": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int64, c_void_p]),
    "fluxhip_quantize_rows_fp8_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int64, c_void_p]),
    "fluxhip_ln_modulate_fp8": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int64, c_int64,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_void_p]),
    "fluxhip_gemm_fp8": (c_int, [C.POINTER(GemmDesc), C.POINTER(Fp8Scales), c_void_p]),
    "fluxhip_gemm_fp8_tile_cfg": (c_int, [C.POINTER(GemmDesc)]),
    "fluxhip_debug_checksum": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "fluxhip_debug_checksum_lds": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_void_p]),
    "fluxhip_gemm_fp8_mx": (c_int, [C.POINTER(GemmDesc), C.POINTER(Fp8Scales), C.POINTER(Fp8Mx), c_void_p]),
    "fluxhip_attention_d128_mx": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int, c_int, c_int,
                                          c_int, c_float, c_void_p]),
    "fluxhip_quantize_mx_fp8": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int64, c_int64, c_int, c_int64, c_int64,
                                        c_void_p]),
    # fp32-faithful ("bf16x3") VAE path
    "fluxhip_split_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "fluxhip_join_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "fluxhip_gemm_x3": (c_int, [C.POINTER(GemmX3Desc), c_void_p]),
    "fluxhip_conv2d_x3": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64]
                          + [c_int] * 9 + [c_void_p, c_int64, C.POINTER(c_int), c_void_p, c_void_p]),
    "fluxhip_conv_up2x_x3": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64]
                             + [c_int] * 5 + [c_void_p, c_int64, C.POINTER(c_int), c_void_p, c_void_p]),
    "fluxhip_groupnorm_apply_x3": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64] + [c_int] * 4
                                   + [c_float, c_int, c_void_p, c_int64, c_int, c_void_p]),
    "fluxhip_groupnorm_silu_x3": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64] + [c_int] * 4
                                  + [c_float, c_int, c_void_p, c_int64, c_void_p]),
    "fluxhip_softmax_rows_x3": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_float, c_void_p]),
    "fluxhip_conv2d_small_x3": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "fluxhip_unpack_latents_x3": (c_int, [c_void_p, c_void_p, c_int64] + [c_int] * 5 + [c_float, c_float, c_void_p]),
    "fluxhip_pixel_linear_x3": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int,
                                        c_float, c_void_p]),
    # float32 arithmetic of the stable_diffusion/ UNet / CLIP (float16=False) on split tensors (ABI 9)
    "fluxhip_layernorm_x3": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_float, c_void_p]),
    "fluxhip_act_x3": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_void_p]),
    "fluxhip_addvec_x3": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int64, c_int, c_void_p]),
    "fluxhip_sincos_embed_x3": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "fluxhip_axpbypcz_f32": (c_int, [c_void_p] * 4 + [c_int64, c_float, c_float, c_float, c_void_p, c_void_p]),
    "fluxhip_softmax_rows_masked_x3": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_float, c_int, c_void_p]),
    "fluxhip_embedding_x3": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_void_p]),
    "fluxhip_pixel_linear_x3_f32in": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int,
                                              c_float, c_void_p]),
}
# float16-storage twins (the stable_diffusion/ models with float16=True): same signatures as the bf16 entry points
for _b, _f in (("fluxhip_gemm_bf16", "fluxhip_gemm_f16"), ("fluxhip_conv2d_bf16", "fluxhip_conv2d_f16"),
               ("fluxhip_small_linear_bf16", "fluxhip_small_linear_f16"), ("fluxhip_silu_bf16", "fluxhip_silu_f16"),
               ("fluxhip_groupnorm_silu_bf16", "fluxhip_groupnorm_silu_f16"),
               ("fluxhip_attention_strided_bf16", "fluxhip_attention_strided_f16"),
               ("fluxhip_attention_strided_vt_bf16", "fluxhip_attention_strided_vt_f16"),
               ("fluxhip_attention_masked_bf16", "fluxhip_attention_masked_f16"),
               ("fluxhip_layernorm_affine_bf16", "fluxhip_layernorm_affine_f16"),
               ("fluxhip_axpbypcz_bf16", "fluxhip_axpbypcz_f16"), ("fluxhip_axpbypcz_dev_bf16", "fluxhip_axpbypcz_dev_f16"),
               ("fluxhip_sincos_embed_f32", "fluxhip_sincos_embed_f32_f16"), ("fluxhip_embedding_bf16", "fluxhip_embedding_f16"),
               ("fluxhip_pixel_linear_x3", "fluxhip_pixel_linear_x3_f16in")):
    SIGNATURES[_f] = SIGNATURES[_b]

_lib = None


def load() -> C.CDLL:
    """Load libfluxhip.so and bind every declared symbol. Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(
            f"libfluxhip.so not found at {LIB_PATH}. Build it with "
            f"`python -c 'import __graft_entry__ as g; g.build()'` or flux_generator_amd/csrc/build.sh. "
            "There is no CPU / PyTorch fallback for the denoise path.")
    # torch first: its wheel bundles its own libamdhip64, and every launch of this library must go through the SAME HIP
    # runtime that owns torch's streams and allocations.  dlopen'ing libfluxhip.so before torch would bind it to the system
    # runtime under /opt/rocm instead, and every kernel launch then fails (two runtimes in one process).
    try:
        import torch  # noqa: F401
    except ImportError:       # pragma: no cover - the C ABI is usable without torch
        pass
    lib = C.CDLL(str(LIB_PATH))
    # FLUXHIP_LIB_AB=1 (tools/lib_ab.py: timing an OLDER build of the library against the current one through the current
    # Python host code) tolerates symbols that build does not have yet and its ABI version; never set it for product runs
    ab = bool(os.environ.get("FLUXHIP_LIB_AB")) and "FLUXHIP_LIB" in os.environ
    for name, (res, args) in SIGNATURES.items():
        if ab and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.fluxhip_abi_version() != 9 and not ab:
        raise RuntimeError("libfluxhip ABI version mismatch")
    if lib.fluxhip_arch() != b"gfx950":
        raise RuntimeError("libfluxhip was not built for gfx950")
    _lib = lib
    _attach_workspace(lib)
    return lib


_workspace = None          # keeps the split-K workspace tensor alive for the life of the process
WORKSPACE_BYTES = 96 << 20
_bound_device = None       # index of the GPU this process drives (bind_device)


def _attach_workspace(lib) -> None:
    """Give the library its split-K workspace (include/fluxhip.h: fluxhip_set_workspace): zero-filled device
    memory owned by this process, on the process's CURRENT device (bind_device makes that the model's device).
    One process drives one GPU, so one buffer is enough; on a host without a GPU (ABI tests)
    nothing is attached and split-K stays off."""
    global _workspace
    try:
        import torch
        if not torch.cuda.is_available():
            return
        dev = torch.device("cuda", torch.cuda.current_device())
        if _workspace is not None and _workspace.device == dev:
            return
        _workspace = torch.zeros(WORKSPACE_BYTES, dtype=torch.uint8, device=dev)
    except Exception:       # pragma: no cover - torch missing: the C ABI is still usable without split-K
        return
    if lib.fluxhip_set_workspace(_workspace.data_ptr(), WORKSPACE_BYTES) != 0:
        raise RuntimeError("fluxhip_set_workspace failed")


def bind_device(device):
    """Resolve the `device=` argument of a model class to the ONE GPU this process drives and make it the current
    device.  libfluxhip launches on the current device's stream and owns one process-wide split-K workspace
    (include/fluxhip.h: fluxhip_set_workspace), so the contract is one process per GPU (torchrun gives every rank its
    own); asking for a second GPU in the same process raises instead of launching kernels on the wrong device."""
    global _bound_device
    import torch
    d = torch.device(device)
    if d.type != "cuda":
        raise RuntimeError(f"libfluxhip needs a HIP device, got '{device}': there is no CPU fallback for this path")
    idx = d.index if d.index is not None else torch.cuda.current_device()
    if _bound_device is None:
        torch.cuda.set_device(idx)
        _bound_device = idx
    elif _bound_device != idx:
        raise RuntimeError(f"libfluxhip is bound to cuda:{_bound_device} in this process (one process per GPU); "
                           f"cuda:{idx} needs its own process (torchrun --nproc-per-node N)")
    elif torch.cuda.current_device() != idx:
        torch.cuda.set_device(idx)
    if _lib is not None:
        _attach_workspace(_lib)          # the library may have been loaded before the device was chosen
    return torch.device("cuda", idx)
