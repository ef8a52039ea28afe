"""ctypes binding of liblcv_hip.so (the C ABI declared in include/lcv_hip.h).

The library is the product path: when it is missing this module raises — there
is no CPU or eager-PyTorch fallback anywhere above it.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_void_p
from pathlib import Path

_LIB_PATH = Path(__file__).resolve().parent / "liblcv_hip.so"

P, I64, F32, I, F64 = c_void_p, c_int64, c_float, c_int, c_double

# name -> argtypes (every function returns int)
_SIGNATURES = {
    "lcv_device_check": [],
    "lcv_adaln_modulate_fwd": [P, P, P, I64, I64, I64, I64, I64, I64, I64, F32, P],
    "lcv_adaln_modulate_bwd": [P, P, P, P, P, I64, I64, I64, I64, I64, I64, I64, F32, P, P],
    "lcv_layernorm_affine_fwd": [P, P, P, P, I64, I64, F32, P],
    "lcv_layernorm_affine_bwd": [P, P, P, P, P, P, I64, I64, F32, P, P],
    "lcv_gate_residual_fwd": [P, P, P, P, I64, I64, I64, I64, I64, I64, P],
    "lcv_gate_residual_bwd": [P, P, P, P, P, I64, I64, I64, I64, I64, I64, P],
    "lcv_qknorm_rope_fwd": [P, P, P, P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, I64, I64, I64, I64, F32, F32, P],
    "lcv_qknorm_rope_bwd": [P, P, P, P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, I64, I64, I64, I64, I64, I64, F32, F32, P, P, I64, P],
    "lcv_attn_fwd": [P, P, P, P, P, I64, I64, I64, I64] + [I64] * 12 + [F32, P],
    "lcv_attn_bwd": [P, P, P, P, P, P, P, P, P, P, I, I64, I64, I64, I64] + [I64] * 21 + [F32, P],
    "lcv_gemm_nt": [P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, I64, I64, I64, I, I, P, P, I64, I64, I64, P],
    "lcv_gemm_set_workspace": [P, I64],
    "lcv_linear_f32_smallm": [P, P, P, P, I64, I64, I64, I, P],
    "lcv_lora_down": [P, P, P, I64, I64, I64, I64, I64, F32, P],
    "lcv_tn_skinny": [P, P, P, I64, I64, I64, I64, I64, F32, P, I64, P],
    "lcv_linear_f32_smallm_bwd": [P, P, P, P, I64, I64, I64, I, P],
    "lcv_timestep_embedding": [P, P, I64, I64, F32, P],
    "lcv_swiglu_fwd": [P, P, P, I64, I64, I64, P],
    "lcv_swiglu_bwd": [P, P, P, P, P, I64, I64, I64, P],
    "lcv_swiglu_bwd_interleaved": [P, P, P, I64, I64, P],
    "lcv_patchify": [P, P, I64, I64, I64, I64, I64, I64, P],
    "lcv_unpatchify": [P, P, I64, I64, I64, I64, I64, I, P],
    "lcv_unpatchify_bwd": [P, P, I64, I64, I64, I64, I64, P],
    "lcv_cfg_euler_step": [P, P, P, P, I64, I64, F32, F32, I, I, P],
    "lcv_euler_step": [P, P, I64, F32, I, P],
    "lcv_fm_noise": [P, P, P, P, I64, I64, P],
    "lcv_fm_mse": [P, P, P, P, P, P, I64, I64, I64, I64, I64, P],
    "lcv_fm_mse_samples": [P, P, P, P, P, I64, I64, I64, I64, I64, I64, I64, P],
    "lcv_grad_norm_clip": [P, I64, I64, I, F32, P, P, P],
    "lcv_adamw_step": [P, I64, I64, I, P, F64, F64, F64, F64, F64, I64, P],
    "lcv_sgd_step": [P, I64, I64, I, P, F64, F64, P],
    "lcv_transpose_pad": [P, P, I64, I64, I64, I64, P],
    "lcv_rowsum": [P, P, I64, I64, I, P],
    "lcv_linear_f32_smallm_wgrad": [P, P, P, P, I64, I64, I64, I, P],
    "lcv_gelu_tanh_fwd": [P, P, I64, P],
    "lcv_gelu_tanh_bwd": [P, P, P, I64, P],
    "lcv_causal_conv3d": [P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, I64, I, I, I, I, P],
    "lcv_conv3d_strided": [P, P, P, P, P, I64, I64, I64, I64, I64, I64, I64, I, I, I, I, I, I, I64, I64, I64, P],
    "lcv_vae_rmsnorm_silu": [P, P, P, I64, I64, I64, I, P],
    "lcv_softmax_rows": [P, P, I64, I64, I64, I64, F32, P],
    "lcv_gather_rows": [P, P, P, I64, I64, I64, P],
    "lcv_t5_rmsnorm": [P, P, P, I64, I64, F32, P],
    "lcv_geglu_tanh_fwd": [P, P, P, I64, I64, I64, P],
    "lcv_t5_attention": [P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, I64, P],
    "lcv_frame_metric_partials": [I64, I64, I64, I, P, P],
    "lcv_frame_sqerr": [P, P, I, P, I64, I64, P],
    "lcv_frame_ssim": [P, P, I, P, I64, I64, I64, I64, P, I, F32, I, F32, F32, P],
}

LCV_EPI_NONE, LCV_EPI_SWIGLU, LCV_EPI_GATE_RESIDUAL, LCV_EPI_GELU_TANH, LCV_EPI_SILU = 0, 1, 2, 3, 4


class LcvError(RuntimeError):
    """`code` is the library's return value when the error came from a C-ABI call (LCV_EINVAL -1: the caller's arguments;
    LCV_EDEVICE -2 / LCV_ELAUNCH -3: the device or a launch failed — the HIP context may be unusable afterwards)."""
    code = None

    @property
    def fatal(self) -> bool:
        return self.code in (-2, -3)


_lib = None


def lib_path() -> Path:
    return _LIB_PATH


def load():
    """Load the shared library once; raise loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.exists():
        raise LcvError(
            f"{_LIB_PATH} is missing: build it with `python __graft_entry__.py` (hipcc, gfx950). "
            "There is no fallback path.")
    # torch first: it carries its own libamdhip64 under the soname this library links to; if the system ROCm copy were
    # loaded before it, the process would hold two HIP runtimes and torch would then find no GPU
    import torch  # noqa: F401
    lib = ctypes.CDLL(str(_LIB_PATH), mode=os.RTLD_NOW | getattr(os, "RTLD_LOCAL", 0))
    lib.lcv_version.restype = c_int
    lib.lcv_version.argtypes = []
    lib.lcv_last_error.restype = c_char_p
    lib.lcv_last_error.argtypes = []
    lib.lcv_attn_fwd_last_kernel.restype = c_char_p
    lib.lcv_attn_fwd_last_kernel.argtypes = []
    lib.lcv_knobs_list.restype = c_char_p
    lib.lcv_knobs_list.argtypes = []
    lib.lcv_knobs_reload.restype = c_int           # a count, not a status
    lib.lcv_knobs_reload.argtypes = []
    lib.lcv_conv3d_last_kernel.restype = c_char_p
    lib.lcv_conv3d_last_kernel.argtypes = []
    lib.lcv_tn_skinny_ws_bytes.restype = c_int64     # a size, not a status
    lib.lcv_tn_skinny_ws_bytes.argtypes = [I64, I64, I64]
    lib.lcv_attn_bwd_ws_floats.restype = c_int64     # likewise
    lib.lcv_attn_bwd_ws_floats.argtypes = [I64, I64, I64, I64]
    for name, args in _SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            continue  # export coverage is asserted by tests/test_abi.py against include/lcv_hip.h
        fn.restype = c_int
        fn.argtypes = args
    _lib = lib
    return lib


def reload_knobs() -> int:
    """The library reads its LCV_* A/B knobs once (include/lcv_hip.h); call this after changing one in os.environ."""
    return load().lcv_knobs_reload()


def set_knob(name: str, value) -> None:
    """Set (or, with None, unset) an A/B knob in os.environ and make the library see it."""
    if value is None:
        os.environ.pop(name, None)
    else:
        os.environ[name] = str(value)
    if _lib is not None:
        _lib.lcv_knobs_reload()


def call(name: str, *args):
    lib = load()
    fn = getattr(lib, name)
    rc = fn(*args)
    if rc != 0:
        msg = lib.lcv_last_error()
        err = LcvError(f"{name} failed ({rc}): {msg.decode() if msg else ''}")
        err.code = rc
        raise err
