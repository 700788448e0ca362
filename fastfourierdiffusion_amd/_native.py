"""ctypes binding of libffd.so (the C ABI declared in include/ffd.h).

The shared library is built in-tree by ``fastfourierdiffusion_amd.build`` (hipcc,
gfx950 only) and lives next to this file.  There is no CPU or PyTorch fallback: if
the library is missing or no gfx950 device is usable, every compute entry point
raises ``FFDError`` loudly.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libffd.so")

FFD_MODEL_TRANSFORMER, FFD_MODEL_LSTM, FFD_MODEL_MLP = 0, 1, 2
FFD_SDE_VP, FFD_SDE_VE = 0, 1
# kernel classes (include/ffd.h FFD_K_*)
K_FFN, K_ATTN, K_OUTPROJ, K_LSTM_REC, K_LSTM_GATES, K_SDE, K_EMBED, K_UNEMBED = range(8)


class FFDError(RuntimeError):
    pass


class ModelDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("n_channels", C.c_int32), ("max_len", C.c_int32), ("d_model", C.c_int32),
        ("n_head", C.c_int32), ("num_layers", C.c_int32), ("dim_feedforward", C.c_int32), ("sde", C.c_int32),
        ("sde_a", C.c_double), ("sde_b", C.c_double), ("fourier_noise_scaling", C.c_int32), ("eps", C.c_double),
    ]


class SdeDesc(C.Structure):
    _fields_ = [("sde", C.c_int32), ("reserved", C.c_int32), ("a", C.c_double), ("b", C.c_double)]


class FrescaCfg(C.Structure):
    _fields_ = [("low_scale", C.c_float), ("high_scale", C.c_float), ("cutoff_ratio", C.c_double),
                ("strategy", C.c_int32), ("num_steps", C.c_int32)]


class CrfCaptureCfg(C.Structure):
    _fields_ = [("ring", C.c_void_p), ("n_slots", C.c_int32), ("every", C.c_int32), ("last", C.c_void_p),
                ("last_every", C.c_int32), ("reserved", C.c_int32)]


class CacheCfg(C.Structure):
    _fields_ = [("K", C.c_int32), ("R", C.c_int32)]


class CacheStats(C.Structure):
    _fields_ = [("recompute_count", C.c_int64), ("cache_hit_count", C.c_int64), ("current_step", C.c_int64),
                ("table_allocated", C.c_int32), ("reserved", C.c_int32)]


_P = C.c_void_p
_F = C.POINTER(C.c_float)

# name -> (restype, argtypes); must list every symbol include/ffd.h declares
SIGNATURES = {
    "ffd_freq_decompose": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_double, _P]),
    "ffd_hermite_predict": (C.c_int, [_P, C.POINTER(C.c_double), C.c_double, C.c_int, _P, C.c_int, C.c_size_t, _P]),
    "ffd_spectral_density": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "ffd_cache_crf_capture": (C.c_int, [_P, C.POINTER(CrfCaptureCfg)]),
    "ffd_create": (C.c_int, [C.POINTER(_P), C.POINTER(ModelDesc), C.c_int]),
    "ffd_destroy": (None, [_P]),
    "ffd_last_error": (C.c_char_p, [_P]),
    "ffd_version": (C.c_char_p, []),
    "ffd_load_weight": (C.c_int, [_P, C.c_char_p, _P, C.c_size_t]),
    "ffd_finalize_weights": (C.c_int, [_P]),
    "ffd_host_noise_scaling": (C.c_int, [C.c_int, C.c_int, _F]),
    "ffd_host_timesteps": (C.c_int, [C.c_int, C.c_double, _F, _F]),
    "ffd_host_gate": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "ffd_score_forward": (C.c_int, [_P, _P, C.c_float, _P, C.c_int, _P]),
    "ffd_score_forward_cached": (C.c_int, [_P, _P, C.c_float, _P, _P, C.c_int, C.c_int, _P]),
    "ffd_sde_step": (C.c_int, [C.POINTER(SdeDesc), _P, _P, _P, C.c_double, C.c_float, _P, C.c_uint64, C.c_uint64,
                               C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "ffd_prior": (C.c_int, [C.POINTER(SdeDesc), _P, _P, _P, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, _P]),
    "ffd_dft": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "ffd_idft": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "ffd_unstandardize_idft": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "ffd_dft_standardize": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "ffd_positional_encoding": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_float, _P]),
    "ffd_time_encoding": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "ffd_fresca": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_double, C.c_int, _P]),
    "ffd_lstm_trace": (C.c_int, [_P, C.POINTER(C.c_uint64), C.c_int, C.POINTER(C.c_int)]),
    "ffd_fresca2d": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_double, C.c_int, _P]),
    "ffd_fresca_enable": (C.c_int, [_P, C.POINTER(FrescaCfg)]),
    "ffd_fresca_disable": (C.c_int, [_P]),
    "ffd_cache_enable": (C.c_int, [_P, C.POINTER(CacheCfg)]),
    "ffd_cache_disable": (C.c_int, [_P]),
    "ffd_cache_reset": (C.c_int, [_P]),
    "ffd_cache_stats_get": (C.c_int, [_P, C.POINTER(CacheStats)]),
    "ffd_cache_tables_read": (C.c_int, [_P, _P, _P, _P]),
    "ffd_sample_batch": (C.c_int, [_P, _P, C.c_int, _F, C.c_int, C.c_float, C.c_int, C.c_int, C.c_uint64,
                                   C.c_uint64, _P, C.c_int, C.c_int, _P]),
    "ffd_flops_per_sample_step": (C.c_double, [_P, C.c_int]),
    "ffd_ffn_flops_per_launch": (C.c_double, [_P, C.c_int]),
    "ffd_kernel_timing_begin": (C.c_int, [_P, C.c_uint32, C.c_int]),
    "ffd_kernel_timing_end": (C.c_int, [_P]),
    "ffd_kernel_timing_get": (C.c_int, [_P, C.c_int, _F, C.POINTER(C.c_int)]),
    "ffd_kernel_work": (C.c_char_p, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "ffd_score_forward_ts": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, _P]),
    "ffd_cache_configure": (C.c_int, [_P, C.POINTER(CacheCfg)]),
    "ffd_tune": (C.c_int, [C.c_char_p, C.c_int]),
    "ffd_tune_get": (C.c_int, [C.c_char_p, C.POINTER(C.c_int)]),
    "ffd_bench_ffn": (C.c_int, [_P, C.c_int, C.c_int, _F, _P]),
    "ffd_probe_ffn_clock": (C.c_int, [_P, C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                      C.POINTER(C.c_uint64), C.c_int, C.POINTER(C.c_int), _P]),
    "ffd_async_status": (C.c_int, [_P]),
    "ffd_probe_attn": (C.c_int, [_P, C.c_int, C.c_int, C.c_double, C.c_int, _F, C.POINTER(C.c_uint64), C.c_int,
                                 C.POINTER(C.c_int), _P]),
}

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load libffd.so (once).  torch is imported first so that the HIP runtime already
    mapped by PyTorch-ROCm (same SONAME, libamdhip64.so.7) is the one libffd binds to --
    streams and device pointers are then shared with torch tensors."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FFDError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C fastfourierdiffusion_amd/csrc`). There is no CPU / PyTorch fallback.")
    try:
        import torch  # noqa: F401  (maps torch's bundled HIP runtime first)
    except Exception:  # pragma: no cover - torch is optional for the pure C ABI
        pass
    try:
        handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    except OSError as e:
        raise FFDError(f"failed to load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as e:
            raise FFDError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = handle
    return handle


def check(rc: int, ctx=None, what: str = "") -> None:
    if rc == 0:
        return
    msg = ""
    if ctx:
        raw = lib().ffd_last_error(ctx)
        msg = raw.decode() if raw else ""
    exc = {-1: AssertionError, -2: NotImplementedError}.get(rc, FFDError)
    raise exc(f"libffd {what} failed (status {rc}): {msg}")


def current_stream_ptr(device) -> int:
    import torch

    return int(torch.cuda.current_stream(device).cuda_stream)


def require_gpu_tensor(t, name: str = "tensor"):
    import torch

    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if t.device.type != "cuda":
        raise FFDError(
            f"{name} lives on {t.device}: fastfourierdiffusion_amd computes only on an MI355X (gfx950) device; "
            "move the model/tensors to cuda (there is no CPU fallback).")
    if t.dtype != torch.float32:
        raise AssertionError(f"{name} must be float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()
