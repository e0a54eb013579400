"""ctypes binding of libcsic_hip.so (include/csic.h).  No fallback: if the HIP library is missing or
no device is visible, the failure is loud."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CSIC_LIB=<path> loads another build of the SAME sources instead -- the range-checking debug build (`make -C csrc debug` ->
# libcsic_hip_debug.so: every global access of the pixel kernels is checked against the frame extent and traps); nothing else.
LIB_PATH = os.environ.get("CSIC_LIB") or os.path.join(_HERE, "libcsic_hip.so")

# csic_status (include/csic.h)
OK = 0
EINVAL_NULL, EINVAL_DIMS, EINVAL_FACTOR, EINVAL_CHROMA_A, EINVAL_CHROMA_B = -1, -2, -3, -4, -5
EINVAL_BITS, EINVAL_OP_PERMUTATION, EINVAL_ROUNDING, EINVAL_FORMAT = -6, -7, -8, -9
EINVAL_NOT_DIVISIBLE, EINVAL_SAMPLING, EINVAL_STRIPE, EINVAL_SIZE = -10, -11, -12, -13
ENODEVICE, EHIP, ENOMEM, ECAPTURE = -20, -21, -22, -23
EIO, EFORMAT = -30, -31

OP_NOOP, OP_SPATIAL, OP_QUANT, OP_CHROMA = 0, 1, 2, 3
ROUND_FLOOR_HW, ROUND_TRUNC_SW = 0, 1
FMT_ARGB8888, FMT_YCBCR888X, FMT_PLANAR = 0, 1, 2
TUNE_VARIANT, TUNE_FORCE_GENERIC, TUNE_NONTEMPORAL, TUNE_NO_VECTOR, TUNE_BLOCK_THREADS = 1, 2, 3, 4, 5
FRAME_GRAPH_HIP, FRAME_GRAPH_DIRECT, FRAME_GRAPH_FUSED, FRAME_GRAPH_AUTO = 0, 1, 2, 3
FRAME_GRAPH_DEFAULT_BRANCHES, FRAME_GRAPH_DEFAULT_QUEUES = 4, 3
PIPELINE_STAGED, PIPELINE_ZERO_COPY = 0, 1


class IllegalArgumentException(ValueError):
    """What the reference's require()s throw at generator construction
    (e.g. ImageProcessor.scala:22-28; tested by SpatialDownsamplerSpec.scala:147-151)."""

    def __init__(self, status: int, message: str):
        super().__init__(message)
        self.status = status


class CsicRuntimeError(RuntimeError):
    """HIP / device failures (CSIC_ENODEVICE, CSIC_EHIP, CSIC_ENOMEM, CSIC_ECAPTURE)."""

    def __init__(self, status: int, message: str):
        super().__init__(message)
        self.status = status


class CsicIOError(OSError):
    """CSIC_EIO / CSIC_EFORMAT from the PNG codec."""

    def __init__(self, status: int, message: str):
        super().__init__(message)
        self.status = status


class CsicParams(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("chroma_a", C.c_int32), ("chroma_b", C.c_int32),
        ("y_bits", C.c_int32), ("cb_bits", C.c_int32), ("cr_bits", C.c_int32),
        ("factor", C.c_int32),
        ("op", C.c_int32 * 3),
        ("rounding", C.c_int32),
        ("sampling", C.c_int32),
        ("in_format", C.c_int32), ("out_format", C.c_int32),
        ("strict_divisible", C.c_int32),
    ]


class CsicPlanarLayout(C.Structure):
    _fields_ = [("y_width", C.c_int32), ("y_height", C.c_int32), ("chroma_width", C.c_int32), ("chroma_height", C.c_int32),
                ("module_width", C.c_int32), ("hold_h", C.c_int32), ("hold_v", C.c_int32), ("replay_last", C.c_int32),
                ("chroma_samples", C.c_int64), ("y_offset", C.c_int64), ("cb_offset", C.c_int64), ("cr_offset", C.c_int64),
                ("frame_bytes", C.c_int64), ("payload_bytes", C.c_int64)]


class CsicFilesStats(C.Structure):
    _fields_ = [("frames", C.c_int64), ("wall_s", C.c_double), ("decode_s", C.c_double), ("encode_s", C.c_double),
                ("gpu_wait_s", C.c_double), ("slot_wait_s", C.c_double), ("decode_threads", C.c_int32), ("encode_threads", C.c_int32),
                ("slots", C.c_int32), ("max_in_flight", C.c_int32), ("in_pixels", C.c_int64), ("out_pixels", C.c_int64)]


class CsicStreamIn(C.Structure):
    _fields_ = [("in_valid", C.c_int32), ("in_bits", C.c_uint32), ("out_ready", C.c_int32), ("sof", C.c_int32), ("eol", C.c_int32)]


class CsicStreamOut(C.Structure):
    _fields_ = [("in_ready", C.c_int32), ("out_valid", C.c_int32), ("out_bits", C.c_uint32)]


STREAM_TOP, STREAM_PROCESSOR, STREAM_RGB2YCBCR, STREAM_CHROMA, STREAM_SPATIAL, STREAM_QUANT = range(6)

# every symbol include/csic.h declares, with its prototype
PROTOTYPES = {
    "csic_abi_version": (C.c_int, []),
    "csic_params_default": (C.c_int, [C.POINTER(CsicParams), C.c_int32, C.c_int32]),
    "csic_validate": (C.c_int, [C.POINTER(CsicParams)]),
    "csic_out_dims": (C.c_int, [C.POINTER(CsicParams), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "csic_algorithmic_bytes": (C.c_int, [C.POINTER(CsicParams), C.POINTER(C.c_int64)]),
    "csic_stripe_rows": (C.c_int, [C.POINTER(CsicParams), C.c_int32, C.c_int32] + [C.POINTER(C.c_int32)] * 4),
    "csic_stripe_halo": (C.c_int, [C.POINTER(CsicParams), C.c_int32, C.c_int32, C.POINTER(C.c_int32)] + [C.POINTER(C.c_int32)] * 6),
    "csic_planar_layout_of": (C.c_int, [C.POINTER(CsicParams), C.POINTER(CsicPlanarLayout)]),
    "csic_reconstruct_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "csic_plan_preferred_pitch": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "csic_debug_build": (C.c_int, []),
    "csic_debug_probe_device": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "csic_strerror": (C.c_char_p, [C.c_int]),
    "csic_last_error": (C.c_char_p, []),
    "csic_device_count": (C.c_int, []),
    "csic_plan_create": (C.c_int, [C.POINTER(CsicParams), C.c_int, C.POINTER(C.c_void_p)]),
    "csic_plan_destroy": (C.c_int, [C.c_void_p]),
    "csic_plan_kernel_name": (C.c_char_p, [C.c_void_p]),
    "csic_plan_tune": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "csic_process_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "csic_process_batch_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "csic_process_pitched_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "csic_process_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    "csic_synth_frame_device": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_uint32, C.c_void_p]),
    "csic_copy_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "csic_checksum_device": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_uint64), C.c_void_p]),
    "csic_frame_graph_create": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32, C.c_int32,
                                          C.POINTER(C.c_void_p)]),
    "csic_frame_graph_create_ex": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32, C.c_int32,
                                             C.c_int32, C.POINTER(C.c_void_p)]),
    "csic_frame_graph_submit": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "csic_frame_graph_wait": (C.c_int, [C.c_void_p, C.c_int64]),
    "csic_frame_graph_backend": (C.c_int, [C.c_void_p]),
    "csic_frame_graph_launch_branches": (C.c_int, [C.c_void_p]),
    "csic_frame_graph_stream_ordered": (C.c_int, [C.c_void_p]),
    "csic_frame_graph_launch": (C.c_int, [C.c_void_p, C.c_void_p]),
    "csic_frame_graph_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "csic_frame_graph_destroy": (C.c_int, [C.c_void_p]),
    "csic_png_info": (C.c_int, [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "csic_png_read_argb": (C.c_int, [C.c_char_p, C.c_void_p, C.c_size_t]),
    "csic_png_write_argb": (C.c_int, [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "csic_pipeline_create": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]),
    "csic_pipeline_destroy": (C.c_int, [C.c_void_p]),
    "csic_pipeline_acquire_input": (C.c_int, [C.c_void_p, C.POINTER(C.POINTER(C.c_uint32))]),
    "csic_pipeline_submit": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "csic_pipeline_collect": (C.c_int, [C.c_void_p, C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_int64)]),
    "csic_pipeline_pending": (C.c_int, [C.c_void_p]),
    "csic_pipeline_set_mode": (C.c_int, [C.c_void_p, C.c_int32]),
    "csic_process_png_files": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                         C.c_int32, C.c_int32, C.POINTER(CsicFilesStats)]),
    "csic_stream_create": (C.c_int, [C.POINTER(CsicParams), C.c_int32, C.POINTER(C.c_void_p)]),
    "csic_stream_destroy": (C.c_int, [C.c_void_p]),
    "csic_stream_reset": (C.c_int, [C.c_void_p]),
    "csic_stream_eval": (C.c_int, [C.c_void_p, C.POINTER(CsicStreamIn), C.POINTER(CsicStreamOut)]),
    "csic_stream_step": (C.c_int, [C.c_void_p, C.POINTER(CsicStreamIn), C.POINTER(CsicStreamOut)]),
    "csic_stream_cycles": (C.c_int64, [C.c_void_p]),
    "csic_stream_depth": (C.c_int, [C.c_void_p]),
    "csic_stream_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int64, C.c_void_p, C.c_size_t,
                                  C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_int64)]),
    "csic_multi_create": (C.c_int, [C.POINTER(CsicParams), C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_void_p)]),
    "csic_multi_destroy": (C.c_int, [C.c_void_p]),
    "csic_multi_count": (C.c_int, [C.c_void_p]),
    "csic_multi_stripe": (C.c_int, [C.c_void_p, C.c_int32] + [C.POINTER(C.c_int32)] * 5),
    "csic_multi_process_device": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "csic_multi_synchronize": (C.c_int, [C.c_void_p]),
    "csic_multi_process_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
}

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C chroma-subsampling-image-compressor_amd/csrc`. There is no CPU fallback.")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 and loads it by
        # absolute path, so if libcsic_hip.so pulled in /opt/rocm's copy first the process would hold
        # two runtimes and the second one sees no GPU.  Importing torch first makes our DT_NEEDED
        # entry resolve (by soname) to the runtime torch already mapped.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (restype, argtypes) in PROTOTYPES.items():
            fn = getattr(L, name)            # AttributeError here == ABI drift; keep it loud
            fn.restype, fn.argtypes = restype, argtypes
        if L.csic_abi_version() != 1:
            raise ImportError("libcsic_hip.so ABI version mismatch")
        _lib = L
    return _lib


def check(status: int) -> int:
    """Maps negative csic_status to the exception the reference's host code would see."""
    if status >= 0:
        return status
    msg = lib().csic_last_error().decode() or lib().csic_strerror(status).decode()
    if EINVAL_SIZE <= status <= EINVAL_NULL:
        raise IllegalArgumentException(status, "requirement failed: " + msg)
    if status in (EIO, EFORMAT):
        raise CsicIOError(status, msg)
    raise CsicRuntimeError(status, msg)
