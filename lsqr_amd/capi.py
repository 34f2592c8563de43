"""ctypes binding of the C-ABI in include/lsqrhip.h (liblsqrhip.so).

This is the Python twin of the Fortran ISO_C_BINDING shim
(lsqr_amd/fortran/lsqr_module.f90): the same entry points, nothing else.  The
library is required -- there is no CPU fallback: if it is missing, or no gfx950
device is usable, every call raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "liblsqrhip.so")
if os.environ.get("LSQRHIP_LIB"):      # measurement scripts only (scripts/): another in-tree build of the same sources
    LIB_PATH = os.path.join(_HERE, "lib", os.path.basename(os.environ["LSQRHIP_LIB"]))

LOG_STRIDE = 14

# status codes (include/lsqrhip.h)
OK, ERR_SIZES, ERR_IROW, ERR_ICOL, ERR_NOT_INIT, ERR_MODE = 0, 1, 2, 3, 4, 5
ERR_NO_DEVICE, ERR_HIP, ERR_ALLOC, ERR_ARG, ERR_TOO_LARGE = 10, 11, 12, 13, 14

EXPORTS = [
    "lsqrhip_error_string", "lsqrhip_last_error", "lsqrhip_device_count", "lsqrhip_set_device",
    "lsqrhip_create", "lsqrhip_create_from_device_coo", "lsqrhip_destroy", "lsqrhip_retain", "lsqrhip_info",
    "lsqrhip_solve", "lsqrhip_solve_device", "lsqrhip_aprod", "lsqrhip_aprod_device",
    "lsqrhip_acheck", "lsqrhip_xcheck", "lsqrhip_log_count", "lsqrhip_log_fetch",
    "lsqrhip_log_extras", "lsqrhip_dnrm2", "lsqrhip_ddot", "lsqrhip_dscal", "lsqrhip_dcopy",
    "lsqrhip_last_timing", "lsqrhip_bench_kernel", "lsqrhip_set_option", "lsqrhip_get_option", "lsqrhip_set_stream", "lsqrhip_dev_alloc",
    "lsqrhip_dev_free", "lsqrhip_dev_upload", "lsqrhip_dev_download", "lsqrhip_dev_sync",
    "lsqrhip_shard_begin", "lsqrhip_shard_stage", "lsqrhip_shard_poll", "lsqrhip_shard_end", "lsqrhip_sum_chunks",
    "lsqrhip_create_f32", "lsqrhip_solve_f32", "lsqrhip_aprod_f32", "lsqrhip_solve_device_f32",
    "lsqrhip_aprod_device_f32",
    "lsqrhip_create_sharded", "lsqrhip_create_sharded_f32", "lsqrhip_rccl_unique_id", "lsqrhip_shard_comm_init", "lsqrhip_shard_solve",
    "lsqrhip_gen_count", "lsqrhip_gen_coo",
    "lsqrhip_create_operator", "lsqrhip_create_operator_f32", "lsqrhip_lstp_create", "lsqrhip_lstp_create_f32",
    "lsqrhip_lstp_vectors", "lsqrhip_acheck_f32", "lsqrhip_xcheck_f32",
]


# int aprod(void *user, int mode, int m, int n, double *d_x, double *d_y, void *hip_stream)
APROD_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)


class LsqrHipError(RuntimeError):
    def __init__(self, code: int, message: str, detail: str = ""):
        self.code = code
        self.message = message      # for codes 1..5: the reference's `error stop` string
        self.detail = detail
        super().__init__(f"[lsqrhip {code}] {message}" + (f" ({detail})" if detail and detail != message else ""))


class Timing(C.Structure):
    _fields_ = [("solve_ms", C.c_double), ("loop_ms", C.c_double), ("spmv1_ms", C.c_double),
                ("spmv2_ms", C.c_double), ("update_ms", C.c_double),
                ("spmv1_launches", C.c_int64), ("spmv2_launches", C.c_int64),
                ("update_launches", C.c_int64), ("spmv1_bytes", C.c_int64),
                ("spmv2_bytes", C.c_int64), ("vec_bytes", C.c_int64), ("itn", C.c_int)]


_lib = None


def lib() -> C.CDLL:
    """Load liblsqrhip.so (raises if it has not been built: run __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LsqrHipError(ERR_NO_DEVICE, "liblsqrhip.so is not built",
                           f"{LIB_PATH} missing; run `python -c 'import __graft_entry__ as g; g.build()'`")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, f64 = C.c_void_p, C.c_int, C.c_int64, C.c_double
    L.lsqrhip_error_string.restype = C.c_char_p
    L.lsqrhip_error_string.argtypes = [i32]
    L.lsqrhip_last_error.restype = C.c_char_p
    L.lsqrhip_device_count.restype = i32
    L.lsqrhip_set_device.argtypes = [i32]
    L.lsqrhip_create.argtypes = [i32, i32, i64, vp, vp, vp, C.POINTER(vp)]
    L.lsqrhip_create_from_device_coo.argtypes = [i32, i32, i64, vp, vp, vp, C.POINTER(vp)]
    L.lsqrhip_destroy.argtypes = [vp]
    L.lsqrhip_retain.argtypes = [vp]
    L.lsqrhip_info.argtypes = [vp, C.POINTER(i64)]
    solve_args = [vp, vp, f64, f64, f64, f64, i32, i32, i32, vp, vp] + [vp] * 7
    L.lsqrhip_solve.argtypes = solve_args
    L.lsqrhip_solve_device.argtypes = solve_args
    L.lsqrhip_aprod.argtypes = [vp, i32, vp, vp]
    L.lsqrhip_create_f32.argtypes = [i32, i32, i64, vp, vp, vp, C.POINTER(vp)]
    L.lsqrhip_solve_f32.argtypes = solve_args
    L.lsqrhip_aprod_f32.argtypes = [vp, i32, vp, vp]
    L.lsqrhip_solve_device_f32.argtypes = solve_args
    L.lsqrhip_aprod_device_f32.argtypes = [vp, i32, vp, vp]
    L.lsqrhip_aprod_device.argtypes = [vp, i32, vp, vp]
    L.lsqrhip_acheck.argtypes = [vp, f64, vp, vp]
    L.lsqrhip_xcheck.argtypes = [vp, f64, f64, f64, vp, vp, vp, vp, vp, vp, vp]
    L.lsqrhip_log_count.argtypes = [vp]
    L.lsqrhip_log_fetch.argtypes = [vp, i32, i32, vp]
    L.lsqrhip_log_extras.argtypes = [vp, vp]
    L.lsqrhip_dnrm2.argtypes = [vp, i64, vp, vp]
    L.lsqrhip_ddot.argtypes = [vp, i64, vp, vp, vp]
    L.lsqrhip_dscal.argtypes = [vp, i64, f64, vp]
    L.lsqrhip_dcopy.argtypes = [vp, i64, vp, vp]
    L.lsqrhip_last_timing.argtypes = [vp, C.POINTER(Timing)]
    L.lsqrhip_bench_kernel.argtypes = [vp, i32, i32, vp]
    L.lsqrhip_set_option.argtypes = [vp, C.c_char_p, i64]
    L.lsqrhip_get_option.argtypes = [vp, C.c_char_p, C.POINTER(i64)]
    L.lsqrhip_set_stream.argtypes = [vp, vp]
    L.lsqrhip_dev_alloc.argtypes = [C.POINTER(vp), i64]
    L.lsqrhip_dev_free.argtypes = [vp]
    L.lsqrhip_dev_upload.argtypes = [vp, vp, i64]
    L.lsqrhip_dev_download.argtypes = [vp, vp, i64]
    L.lsqrhip_shard_begin.argtypes = [vp, vp, i64, i32, i32, f64, f64, f64, f64, i32, i32, vp, vp, vp, vp]
    L.lsqrhip_create_sharded.argtypes = [i32, i32, i64, vp, vp, vp, i32, C.POINTER(vp)]
    L.lsqrhip_create_sharded_f32.argtypes = [i32, i32, i64, vp, vp, vp, i32, C.POINTER(vp)]
    L.lsqrhip_rccl_unique_id.argtypes = [vp]
    L.lsqrhip_shard_comm_init.argtypes = [vp, i32, i32, i64, i64, vp]
    L.lsqrhip_shard_solve.argtypes = [vp, vp, f64, f64, f64, f64, i32, i32, vp, vp] + [vp] * 7
    L.lsqrhip_shard_stage.argtypes = [vp, i32]
    L.lsqrhip_sum_chunks.argtypes = [vp, vp, i32, i64, vp]
    L.lsqrhip_shard_poll.argtypes = [vp, vp]
    L.lsqrhip_shard_end.argtypes = [vp, vp, vp] + [vp] * 7
    L.lsqrhip_gen_count.restype = i64
    L.lsqrhip_gen_count.argtypes = [i32, i64, i64, i64, i64, i64, i64]
    L.lsqrhip_gen_coo.argtypes = [i32, C.c_uint64, i64, i64, i64, i64, i64, i64, vp, vp, vp, vp, vp, vp]
    L.lsqrhip_create_operator.argtypes = [i32, i32, APROD_FN, vp, C.POINTER(vp)]
    L.lsqrhip_lstp_create.argtypes = [i32, i32, i32, i32, f64, C.POINTER(vp), C.POINTER(f64), C.POINTER(f64)]
    L.lsqrhip_lstp_vectors.argtypes = [vp, vp, vp, vp, vp, vp, C.POINTER(vp)]
    L.lsqrhip_acheck_f32.argtypes = [vp, f64, vp, vp]
    L.lsqrhip_xcheck_f32.argtypes = [vp, f64, f64, f64, vp, vp, vp, vp, vp, vp, vp]
    L.lsqrhip_create_operator_f32.argtypes = [i32, i32, APROD_FN, vp, C.POINTER(vp)]
    L.lsqrhip_lstp_create_f32.argtypes = [i32, i32, i32, i32, f64, C.POINTER(vp), C.POINTER(f64), C.POINTER(f64)]
    for name in EXPORTS:
        getattr(L, name)  # every declared symbol must be exported
    _lib = L
    return L


def check(rc: int) -> None:
    if rc != OK:
        L = lib()
        raise LsqrHipError(rc, L.lsqrhip_error_string(rc).decode(), L.lsqrhip_last_error().decode())


def device_count() -> int:
    return int(lib().lsqrhip_device_count())


def _ptr(a: np.ndarray | None):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class DeviceBuffer:
    """A raw HBM allocation (for hosts that do not bring their own allocator)."""

    def __init__(self, nbytes: int):
        self.ptr = C.c_void_p()
        self.nbytes = int(nbytes)
        check(lib().lsqrhip_dev_alloc(C.byref(self.ptr), self.nbytes))

    @classmethod
    def from_array(cls, a: np.ndarray) -> "DeviceBuffer":
        a = np.ascontiguousarray(a)
        buf = cls(a.nbytes)
        check(lib().lsqrhip_dev_upload(buf.ptr, _ptr(a), a.nbytes))
        return buf

    def copy_from(self, a: np.ndarray) -> None:
        a = np.ascontiguousarray(a)
        if a.nbytes > self.nbytes:
            raise ValueError("array larger than the device buffer")
        check(lib().lsqrhip_dev_upload(self.ptr, _ptr(a), a.nbytes))

    def to_array(self, dtype, count: int) -> np.ndarray:
        out = np.empty(count, dtype=dtype)
        check(lib().lsqrhip_dev_download(_ptr(out), self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            lib().lsqrhip_dev_free(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
