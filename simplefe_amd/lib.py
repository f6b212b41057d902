"""ctypes binding of libsfe_dsp.so -- the C ABI declared in include/sfe_dsp.h.

The library is the product: there is no Python or CPU fallback.  If the in-tree .so is
missing or fails to load this module raises; if no GPU is usable the create calls fail
with SFE_ENODEV (and SfeError is raised by the wrappers in api.py).
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libsfe_dsp.so")

SFE_OK, SFE_EINVAL, SFE_ENOMEM, SFE_EHIP, SFE_ENODEV, SFE_ESTATE, SFE_ERANGE = 0, -1, -2, -3, -4, -5, -6
FIR_ALGO_AUTO, FIR_ALGO_DIRECT, FIR_ALGO_FFT = 0, 1, 2
FIR_VARIANT_AUTO, FIR_VARIANT_REGISTER_LOADS, FIR_VARIANT_LDS_DMA, FIR_VARIANT_WAVE_PRIVATE = -1, 0, 1, 2
FIR_VARIANT_NAMES = {-1: "auto", 0: "register loads", 1: "LDS-DMA", 2: "LDS-DMA, wave-private layout"}
RS_RESAMPLE, RS_DECIMATE = 0, 1
RS_ALGO_AUTO, RS_ALGO_DIRECT, RS_ALGO_FFT, RS_ALGO_MFMA = 0, 1, 2, 3
FMT_F32, FMT_U8, FMT_TX10 = 0, 1, 2


class TimeState(C.Structure):
    _fields_ = [("pos", C.c_int32), ("mu", C.c_float), ("leftover", C.c_int32)]


vp, sz, i32, f32 = C.c_void_p, C.c_size_t, C.c_int, C.c_float
fp = C.POINTER(C.c_float)

# name -> (restype, argtypes): every symbol include/sfe_dsp.h declares
SIGNATURES = {
    "sfe_dsp_version": (C.c_char_p, []),
    "sfe_dsp_last_error": (C.c_char_p, []),
    "sfe_dsp_device_count": (i32, [C.POINTER(i32)]),
    "sfe_dsp_set_device": (i32, [i32]),
    "sfe_dsp_get_device": (i32, [C.POINTER(i32)]),
    "sfe_dsp_sync": (i32, [vp]),
    "sfe_dsp_malloc": (i32, [C.POINTER(vp), sz]),
    "sfe_dsp_free": (i32, [vp]),
    "sfe_dsp_host_alloc": (i32, [C.POINTER(vp), sz]),
    "sfe_dsp_host_free": (i32, [vp]),
    "sfe_dsp_memcpy_h2d": (i32, [vp, vp, sz, vp]),
    "sfe_dsp_memcpy_d2h": (i32, [vp, vp, sz, vp]),
    "sfe_dsp_memset": (i32, [vp, i32, sz, vp]),
    "sfe_dsp_timer_create": (i32, [C.POINTER(vp)]),
    "sfe_dsp_timer_start": (i32, [vp, vp]),
    "sfe_dsp_timer_stop": (i32, [vp, vp]),
    "sfe_dsp_timer_elapsed_ms": (i32, [vp, C.POINTER(f32)]),
    "sfe_dsp_timer_destroy": (i32, [vp]),
    "sfe_dsp_synth_fill": (i32, [vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, vp]),
    "sfe_dsp_fir_create": (i32, [vp, i32, i32, i32, i32, i32, i32, C.POINTER(vp)]),
    "sfe_dsp_fir_create_per_channel": (i32, [vp, i32, i32, i32, i32, C.POINTER(vp)]),
    "sfe_dsp_fir_plan": (i32, [i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
    "sfe_dsp_fir_host_buffer": (i32, [vp, C.POINTER(fp), C.POINTER(i32)]),
    "sfe_dsp_fir_process_block": (i32, [vp]),
    "sfe_dsp_fir_process_stream": (i32, [vp, vp, vp, sz, sz, sz, vp]),
    "sfe_dsp_fir_process_host": (i32, [vp, vp, vp, sz]),
    "sfe_dsp_fir_set_algo": (i32, [vp, i32]),
    "sfe_dsp_fir_set_zero_copy_max": (i32, [vp, sz]),
    "sfe_dsp_fir_set_variant": (i32, [vp, i32]),
    "sfe_dsp_fir_get_variant": (i32, [vp, C.POINTER(i32), C.POINTER(i32), fp]),
    "sfe_dsp_fir_forget_calibrations": (i32, []),
    "sfe_dsp_fir_calibrate": (i32, [vp, vp, vp, sz, sz, sz, vp, C.POINTER(i32)]),
    "sfe_dsp_rs_set_algo": (i32, [vp, i32]),
    "sfe_dsp_fir_pipe_create": (i32, [vp, sz, C.POINTER(vp)]),
    "sfe_dsp_rs_pipe_create": (i32, [vp, sz, f32, C.POINTER(vp)]),
    "sfe_dsp_pipe_push": (i32, [vp, vp, sz, C.POINTER(sz)]),
    "sfe_dsp_pipe_pull": (i32, [vp, vp, sz, i32, C.POINTER(sz)]),
    "sfe_dsp_pipe_pending": (i32, [vp, C.POINTER(sz)]),
    "sfe_dsp_pipe_acquire": (i32, [vp, C.POINTER(vp), C.POINTER(sz)]),
    "sfe_dsp_pipe_commit": (i32, [vp, sz]),
    "sfe_dsp_pipe_peek": (i32, [vp, C.POINTER(vp), C.POINTER(sz), i32]),
    "sfe_dsp_pipe_release": (i32, [vp, sz]),
    "sfe_dsp_pipe_destroy": (i32, [vp]),
    "sfe_dsp_fir_load_history": (i32, [vp, vp, sz, sz, vp]),
    "sfe_dsp_rs_load_history": (i32, [vp, vp, sz, sz, vp]),
    "sfe_dsp_rs_seek": (i32, [vp, C.c_uint64, f32]),
    "sfe_dsp_rs_plan_seek": (i32, [C.POINTER(TimeState), i32, C.c_uint64, f32]),
    "sfe_dsp_rs_get_state": (i32, [vp, C.POINTER(TimeState)]),
    "sfe_dsp_rs_set_state": (i32, [vp, C.POINTER(TimeState)]),
    "sfe_dsp_fir_set_input_format": (i32, [vp, i32]),
    "sfe_dsp_fir_set_output_format": (i32, [vp, i32]),
    "sfe_dsp_rs_set_input_format": (i32, [vp, i32]),
    "sfe_dsp_fir_reset": (i32, [vp]),
    "sfe_dsp_fir_destroy": (i32, [vp]),
    "sfe_dsp_rs_create": (i32, [vp, i32, i32, i32, i32, i32, i32, i32, C.POINTER(vp)]),
    "sfe_dsp_rs_process": (i32, [vp, vp, i32, vp, i32, f32, C.POINTER(i32)]),
    "sfe_dsp_rs_process_stream": (i32, [vp, vp, sz, sz, vp, sz, sz, f32, C.POINTER(sz), vp]),
    "sfe_dsp_rs_set_exact": (i32, [vp, i32]),
    "sfe_dsp_rs_reset": (i32, [vp]),
    "sfe_dsp_rs_destroy": (i32, [vp]),
    "sfe_dsp_rs_plan": (i32, [C.POINTER(TimeState), i32, i32, i32, f32, vp, vp, i32, C.POINTER(i32)]),
    "sfe_dsp_fir_group_create": (i32, [vp, i32, i32, i32, i32, i32, C.POINTER(i32), i32, C.POINTER(vp)]),
    "sfe_dsp_fir_group_shards": (i32, [vp, C.POINTER(i32)]),
    "sfe_dsp_fir_group_shard": (i32, [vp, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(vp), C.POINTER(vp)]),
    "sfe_dsp_fir_group_process_stream": (i32, [vp, C.POINTER(vp), C.POINTER(vp), sz, sz, sz]),
    "sfe_dsp_fir_group_sync": (i32, [vp]),
    "sfe_dsp_fir_group_reset": (i32, [vp]),
    "sfe_dsp_fir_group_destroy": (i32, [vp]),
    "sfe_dsp_rs_group_create": (i32, [vp, i32, i32, i32, i32, i32, C.POINTER(i32), i32, i32, C.POINTER(vp)]),
    "sfe_dsp_rs_group_shards": (i32, [vp, C.POINTER(i32)]),
    "sfe_dsp_rs_group_shard": (i32, [vp, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(vp), C.POINTER(vp)]),
    "sfe_dsp_rs_group_process_stream": (i32, [vp, C.POINTER(vp), sz, sz, C.POINTER(vp), sz, sz, f32, C.POINTER(sz)]),
    "sfe_dsp_rs_group_sync": (i32, [vp]),
    "sfe_dsp_rs_group_reset": (i32, [vp]),
    "sfe_dsp_rs_group_destroy": (i32, [vp]),
    "sfe_dsp_rx_u8_to_f32": (i32, [vp, vp, sz, vp]),
    "sfe_dsp_tx_f32_to_10bit": (i32, [vp, vp, sz, vp]),
}

# the diagnostic library only (simplefe_amd/csrc/diag/sfe_dsp_diag.h; scripts/ load it by pointing LIB_PATH at it): bound when present
DIAG_SIGNATURES = {
    "sfe_dsp_probe_pair": (i32, [vp, sz, vp, sz, C.POINTER(C.c_float)]),
    "sfe_dsp_malloc_pair": (i32, [sz, sz, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "sfe_dsp_malloc_pair_screened": (i32, [sz, sz, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "sfe_dsp_mem_kind": (i32, [vp, C.POINTER(C.c_int)]),
}

_lib = None


def load():
    """Load the in-tree HIP extension.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m simplefe_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)          # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in DIAG_SIGNATURES.items():
        fn = getattr(L, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    _lib = L
    return L


class SfeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libsfe_dsp error {code}: {msg}")
        self.code = code


def check(rc):
    if rc != SFE_OK:
        raise SfeError(rc, load().sfe_dsp_last_error().decode(errors="replace"))
    return rc
