"""ctypes loader for libohs_hip.so (the C ABI in include/ohs_hip.h).

There is no CPU fallback: if the shared library is missing this module raises,
and every create call fails with OHS_ERR_NO_DEVICE when no GPU is usable.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
EXP_LIB_PATH = os.path.join(HERE, "libohs_hip_exp.so")
# The product library reads no environment variable.  This LOADER does, for the lab: OHS_LIB=<path> (or OHS_LIB=exp for
# libohs_hip_exp.so) makes the whole package run on an experiments build (`python -m open_headstage_amd.build
# --experiments`, or an OHS_BUILD_TAG / OHS_EXTRA_DEFS variant of it) -- what tools/ and LABNOTES.md's A/B runs use.
_env_lib = os.environ.get("OHS_LIB")
LIB_PATH = EXP_LIB_PATH if _env_lib == "exp" else (_env_lib or os.path.join(HERE, "libohs_hip.so"))

OHS_OK = 0
OHS_ERR_INVALID_ARG = 1
OHS_ERR_NO_DEVICE = 2
OHS_ERR_HIP = 3
OHS_ERR_OUTSIDE_NYQUIST = 4
OHS_ERR_NEGATIVE_Q = 5
OHS_ERR_ALLOC = 6
OHS_ERR_IO = 7
OHS_ERR_PARSE = 8
OHS_ERR_RCCL = 9

class BandSettingC(C.Structure):
    """ohs_band_setting (include/ohs_hip.h) = BandSetting, autoeq_parser.rs:34-41"""
    _fields_ = [("enabled", C.c_int), ("filter_type", C.c_int), ("frequency", C.c_float),
                ("q", C.c_float), ("gain", C.c_float)]


fp = C.POINTER(C.c_float)
vp = C.c_void_p
vpp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes): the complete export list of include/ohs_hip.h
PROTOTYPES = {
    "ohs_version": (C.c_int, []),
    "ohs_status_string": (C.c_char_p, [C.c_int]),
    "ohs_last_error": (C.c_char_p, []),
    "ohs_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "ohs_device_pci_bus_id": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
    "ohs_debug_fft1024": (C.c_int, [C.c_int, fp, fp, C.c_int, C.c_size_t]),
    "ohs_engine_create": (C.c_int, [C.c_int, vpp]),
    "ohs_engine_clone": (C.c_int, [vp, vpp]),
    "ohs_engine_destroy": (None, [vp]),
    "ohs_engine_set_ir": (C.c_int, [vp, C.c_int, fp, C.c_size_t]),
    "ohs_engine_process_block": (C.c_int, [vp, vp, vp, vp, vp, C.c_size_t]),     # audio pointers as plain addresses (hot call)
    "ohs_engine_num_partitions": (C.c_int, [vp, C.c_int, C.POINTER(C.c_size_t)]),
    "ohs_engine_set_realtime": (C.c_int, [vp, C.c_int]),
    "ohs_rt_configure": (C.c_int, [C.c_int, C.c_uint64]),
    "ohs_engine_latency_samples": (C.c_int, [vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "ohs_engine_silent_frames": (C.c_int, [vp, C.POINTER(C.c_uint64)]),
    "ohs_engine_debug_rt_stamps": (C.c_int, [vp, C.POINTER(C.c_uint64)]),
    "ohs_biquad_coeffs": (C.c_int, [C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, fp]),
    "ohs_biquad_create": (C.c_int, [C.c_int, C.c_float, vpp]),
    "ohs_biquad_clone": (C.c_int, [vp, vpp]),
    "ohs_biquad_destroy": (None, [vp]),
    "ohs_biquad_reset_state": (C.c_int, [vp]),
    "ohs_biquad_set_enabled": (C.c_int, [vp, C.c_int]),
    "ohs_biquad_update_coeffs": (C.c_int, [vp, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]),
    "ohs_biquad_set_coeffs": (C.c_int, [vp, fp]),
    "ohs_biquad_process": (C.c_int, [vp, vp, C.c_size_t]),
    "ohs_eq_create": (C.c_int, [C.c_int, C.c_size_t, C.c_float, vpp]),
    "ohs_eq_destroy": (None, [vp]),
    "ohs_eq_update_band": (C.c_int, [vp, C.c_size_t, C.c_float, C.c_int, C.c_float, C.c_float,
                                     C.c_float, C.c_int]),
    "ohs_eq_set_band_coeffs": (C.c_int, [vp, C.c_size_t, fp, C.c_int]),
    "ohs_eq_get_band_coeffs": (C.c_int, [vp, C.c_size_t, fp, C.POINTER(C.c_int)]),
    "ohs_eq_process_block": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "ohs_eq_reset": (C.c_int, [vp]),
    "ohs_eq_set_exact_specials": (C.c_int, [vp, C.c_int]),
    "ohs_eq_set_flush_denormals": (C.c_int, [vp, C.c_int]),
    "ohs_engine_set_flush_denormals": (C.c_int, [vp, C.c_int]),
    "ohs_batch_set_flush_denormals": (C.c_int, [vp, C.c_int]),
    "ohs_batch_set_eq_exact_specials": (C.c_int, [vp, C.c_int]),
    "ohs_eq_frequency_response": (C.c_int, [vp, C.c_float, fp, C.c_size_t, fp]),
    "ohs_autoeq_parse_csv": (C.c_int, [C.c_char_p, C.POINTER(BandSettingC), C.c_size_t,
                                       C.POINTER(C.c_size_t)]),
    "ohs_autoeq_parse_csv_text": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(BandSettingC), C.c_size_t,
                                            C.POINTER(C.c_size_t)]),
    "ohs_sofa_open": (C.c_int, [C.c_char_p, vpp]),
    "ohs_sofa_open_ex": (C.c_int, [C.c_char_p, C.c_uint, C.c_float, vpp]),
    "ohs_sofa_loudness_factor": (C.c_int, [vp, fp]),
    "ohs_sofa_neighbors": (C.c_int, [vp, C.c_size_t, C.POINTER(C.c_int)]),
    "ohs_sofa_close": (None, [vp]),
    "ohs_sofa_info": (C.c_int, [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                                C.POINTER(C.c_float)]),
    "ohs_sofa_position": (C.c_int, [vp, C.c_size_t, fp]),
    "ohs_sofa_spherical_to_cartesian": (C.c_int, [fp, fp]),
    "ohs_sofa_cartesian_to_spherical": (C.c_int, [fp, fp]),
    "ohs_sofa_nearest": (C.c_int, [vp, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_size_t)]),
    "ohs_sofa_get_ir": (C.c_int, [vp, C.c_size_t, C.c_size_t, fp, C.c_size_t]),
    "ohs_sofa_get_hrtf_irs": (C.c_int, [vp, C.c_float, C.c_float, C.c_float, fp, fp, C.c_size_t, fp, fp]),
    "ohs_sofa_resample_ir": (C.c_int, [fp, C.c_size_t, C.c_float, C.c_float, fp, C.c_size_t,
                                       C.POINTER(C.c_size_t)]),
    "ohs_sofa_speaker_irs": (C.c_int, [vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                       C.POINTER(fp), C.c_size_t, C.POINTER(C.c_size_t)]),
    "ohs_engine_set_speakers": (C.c_int, [vp, vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_uint)]),
    "ohs_batch_set_speakers": (C.c_int, [vp, vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_uint)]),
    "ohs_node_batch_set_speakers": (C.c_int, [vp, vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_uint)]),
    "ohs_chain_process": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_float, vp, vp, C.c_size_t]),
    "ohs_batch_create": (C.c_int, [C.c_int, C.c_size_t, C.c_size_t, vpp]),
    "ohs_batch_destroy": (None, [vp]),
    "ohs_batch_set_ir": (C.c_int, [vp, C.c_int, fp, C.c_size_t]),
    "ohs_batch_set_eq_band_coeffs": (C.c_int, [vp, C.c_size_t, fp, C.c_int]),
    "ohs_batch_update_eq_band": (C.c_int, [vp, C.c_size_t, C.c_float, C.c_int, C.c_float, C.c_float,
                                           C.c_float, C.c_int]),
    "ohs_batch_set_stream_eq_band_coeffs": (C.c_int, [vp, C.c_size_t, C.c_size_t, fp, C.c_int]),
    "ohs_batch_update_stream_eq_band": (C.c_int, [vp, C.c_size_t, C.c_size_t, C.c_float, C.c_int, C.c_float, C.c_float,
                                                  C.c_float, C.c_int]),
    "ohs_batch_share_eq_table": (C.c_int, [vp]),
    "ohs_node_batch_set_stream_eq_band_coeffs": (C.c_int, [vp, C.c_size_t, C.c_size_t, fp, C.c_int]),
    "ohs_node_batch_share_eq_table": (C.c_int, [vp]),
    "ohs_batch_set_eq_enabled": (C.c_int, [vp, C.c_int]),
    "ohs_batch_set_gain": (C.c_int, [vp, C.c_float]),
    "ohs_batch_set_conv_plan": (C.c_int, [vp, C.c_int]),
    "ohs_batch_last_conv_plan": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ohs_batch_conv_plan_counts": (C.c_int, [vp, C.POINTER(C.c_uint64), C.c_int]),
    "ohs_batch_profile_eq_clock": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "ohs_node_batch_set_conv_plan": (C.c_int, [vp, C.c_int]),
    "ohs_batch_reset": (C.c_int, [vp]),
    "ohs_batch_process": (C.c_int, [vp, vp, vp, C.c_size_t, C.c_size_t, C.c_size_t, vp]),
    "ohs_batch_process_deferred": (C.c_int, [vp, vp, vp, C.c_size_t, C.c_size_t, C.c_size_t, vp]),
    "ohs_batch_join": (C.c_int, [vp, vp]),
    "ohs_batch_sync": (C.c_int, [vp, vp]),
    "ohs_batch_process_host": (C.c_int, [vp, vp, vp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t]),
    "ohs_batch_set_profiling": (C.c_int, [vp, C.c_int]),
    "ohs_batch_profile_read": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                         C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                         C.POINTER(C.c_uint64)]),
    "ohs_batch_algorithmic_bytes": (C.c_int, [vp, C.c_size_t, C.POINTER(C.c_uint64)]),
    "ohs_batch_kernel_bytes": (C.c_int, [vp, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "ohs_node_batch_create": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.c_size_t, C.c_size_t, vpp]),
    "ohs_node_batch_destroy": (None, [vp]),
    "ohs_node_batch_shard": (C.c_int, [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "ohs_node_batch_device_batch": (C.c_int, [vp, C.c_int, vpp]),
    "ohs_node_batch_set_tables": (C.c_int, [vp, C.POINTER(fp), C.POINTER(C.c_size_t), fp, C.POINTER(C.c_int)]),
    "ohs_node_batch_set_ir": (C.c_int, [vp, C.c_int, fp, C.c_size_t]),
    "ohs_node_batch_set_eq_band_coeffs": (C.c_int, [vp, C.c_size_t, fp, C.c_int]),
    "ohs_node_batch_set_eq_enabled": (C.c_int, [vp, C.c_int]),
    "ohs_node_batch_set_gain": (C.c_int, [vp, C.c_float]),
    "ohs_node_batch_reset": (C.c_int, [vp]),
    "ohs_node_batch_process_host": (C.c_int, [vp, vp, vp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t]),
    "ohs_node_batch_process": (C.c_int, [vp, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_size_t, C.c_size_t,
                                         C.c_size_t]),
    "ohs_node_batch_sync": (C.c_int, [vp]),
    "ohs_node_batch_stream": (C.c_int, [vp, C.c_int, vpp]),
    "ohs_node_batch_timer_begin": (C.c_int, [vp]),
    "ohs_node_batch_timer_end": (C.c_int, [vp, fp]),
    "ohs_node_batch_rccl_info": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
}


class OhsError(RuntimeError):
    def __init__(self, status: int, detail: str):
        self.status = status
        super().__init__(f"{lib().ohs_status_string(status).decode()} ({status}): {detail}")


# experiments build only (not in include/ohs_hip.h, not exported by the product library)
DEBUG_PROTOTYPES = {
    "ohs_debug_set_tuning": (C.c_int, [C.c_char_p, C.c_char_p]),
    "ohs_debug_inject_batch_failure": (C.c_int, [C.c_int]),
}

_lib = None
_exp = None


def _load(path: str, extra=None) -> C.CDLL:
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: build it with `python -m open_headstage_amd.build` "
            "(hipcc, gfx950). There is no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 and the
    # dynamic loader de-duplicates by SONAME only when torch's copy is loaded FIRST (torch's
    # libs ask for "libamdhip64.so", ours for "libamdhip64.so.7").  Two runtimes in one
    # process leave the second one without a device.  torch is plumbing here (device
    # tensors, streams, torch.distributed); without it the system ROCm runtime is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    for name, (res, args) in {**PROTOTYPES, **(extra or {})}.items():
        fn = getattr(L, name)     # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    return L


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = _load(LIB_PATH)
    return _lib


def experiments_lib() -> C.CDLL:
    """libohs_hip_exp.so, loaded BESIDE the product library: a second copy of the code with its own state, whose launch
    plans ohs_debug_set_tuning can edit.  Handles belong to the library that made them -- pass `library=` to
    BatchProcessor."""
    global _exp
    if _exp is None:
        _exp = _load(EXP_LIB_PATH, DEBUG_PROTOTYPES)
    return _exp


def set_tuning(key: str, value) -> None:
    """experiments build: edit one launch-plan constant (csrc/tuning.h) of libohs_hip_exp.so"""
    L = experiments_lib()
    rc = L.ohs_debug_set_tuning(key.encode(), str(value).encode())
    if rc != OHS_OK:
        raise OhsError(rc, L.ohs_last_error().decode(errors="replace"))


def check(status: int, library=None) -> None:
    if status != OHS_OK:
        raise OhsError(status, (library or lib()).ohs_last_error().decode(errors="replace"))
