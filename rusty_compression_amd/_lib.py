"""ctypes binding of librusty_compression_amd.so (the C ABI in include/rusty_compression_amd.h).

PyTorch is used for device memory and streams only: every computation goes
through the C ABI.  There is NO fallback: if the HIP library is missing or no
GPU is present, the calls raise.
"""
from __future__ import annotations

import ctypes
import os
import re
import threading
from typing import Dict, Optional, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librusty_compression_amd.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "rusty_compression_amd.h")

# status codes (include/rusty_compression_amd.h) <-> RustyCompressionError (reference src/types.rs:11-21)
RC_OPT_TALL_SKINNY_FAST_PATH = 1
RC_OPT_WIDE_LAZY_QRCP = 2
RC_OPT_WIDE_COOP_QRCP = 3
RC_OPT_POWER_ITERATION_FIXED = 4
RC_OPT_FORK_BRANCHES = 5
RC_OPT_BLOCKED_QRCP = 6
RC_OPT_CONCURRENCY_HINT = 7
RC_OPT_COOP_PANEL = 8
RC_OK, RC_LINALG_ERROR, RC_COMPRESSION_ERROR, RC_LAYOUT_ERROR, RC_PIVOTED_QR_ERROR, RC_INVALID_ARGUMENT, RC_RUNTIME_ERROR = range(7)


class RustyCompressionError(Exception):
    """Base of the error enum of the reference (src/types.rs:11-21)."""


class LinalgError(RustyCompressionError):
    pass


class CompressionError(RustyCompressionError):
    """`Could not compress to desired tolerance` (src/types.rs:15-16)."""


class LayoutError(RustyCompressionError):
    pass


class PivotedQRError(RustyCompressionError):
    pass


class HipRuntimeError(RustyCompressionError):
    pass


_STATUS_EXC = {
    RC_LINALG_ERROR: LinalgError,
    RC_COMPRESSION_ERROR: CompressionError,
    RC_LAYOUT_ERROR: LayoutError,
    RC_PIVOTED_QR_ERROR: PivotedQRError,
    RC_INVALID_ARGUMENT: AssertionError,  # the reference panics (assert!) on these
    RC_RUNTIME_ERROR: HipRuntimeError,
}


class rc_matrix(ctypes.Structure):
    _fields_ = [
        ("data", ctypes.c_void_p),
        ("rows", ctypes.c_int64),
        ("cols", ctypes.c_int64),
        ("row_stride", ctypes.c_int64),
        ("col_stride", ctypes.c_int64),
    ]


class rc_rsvd_id_out(ctypes.Structure):
    _fields_ = [
        ("range_q", rc_matrix),
        ("u", rc_matrix),
        ("s", ctypes.c_void_p),
        ("vt", rc_matrix),
        ("qr_q", rc_matrix),
        ("qr_r", rc_matrix),
        ("qr_ind", ctypes.c_void_p),
        ("id_c", rc_matrix),
        ("id_z", rc_matrix),
    ]


def declared_symbols(header_path: str = HEADER_PATH):
    """Every function the header declares (used by the CPU test that checks the exports)."""
    text = open(header_path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rc_[a-z0-9_]+)\s*\(", text)))


_lib = None
_lib_lock = threading.Lock()


def lib() -> ctypes.CDLL:
    """Load the HIP library; fail loudly when it has not been built."""
    global _lib
    with _lib_lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    f"{LIB_PATH} is missing: build it with `python -c \"import __graft_entry__ as g; g.build()\"` "
                    "(hipcc --offload-arch=gfx950). There is no CPU fallback."
                )
            _lib = ctypes.CDLL(LIB_PATH)
            _lib.rc_last_error_message.restype = ctypes.c_char_p
            _lib.rc_last_error_message.argtypes = [ctypes.c_void_p]
            _lib.rc_abi_version.restype = ctypes.c_int32
    return _lib


def suffix(dtype) -> str:
    import torch

    if dtype == torch.float64:
        return "f64"
    if dtype == torch.float32:
        return "f32"
    if dtype == torch.complex128:
        return "c64"
    if dtype == torch.complex64:
        return "c32"
    raise TypeError(f"unsupported scalar type {dtype}: the engine is instantiated for f32, f64, c32 and c64")


def real_dtype(dtype):
    """The real type of a scalar type (`<A as Scalar>::Real` in the reference): singular values, norms."""
    import torch

    return {torch.complex128: torch.float64, torch.complex64: torch.float32}.get(dtype, dtype)


class rc_complex64(ctypes.Structure):
    _fields_ = [("re", ctypes.c_double), ("im", ctypes.c_double)]


class rc_complex32(ctypes.Structure):
    _fields_ = [("re", ctypes.c_float), ("im", ctypes.c_float)]


def scalar_arg(dtype, value: complex):
    """A scalar of `dtype` as the C ABI takes it by value (double / float / rc_complex64 / rc_complex32)."""
    import torch

    if dtype == torch.float64:
        return ctypes.c_double(float(value.real if isinstance(value, complex) else value))
    if dtype == torch.float32:
        return ctypes.c_float(float(value.real if isinstance(value, complex) else value))
    v = complex(value)
    return rc_complex64(v.real, v.imag) if dtype == torch.complex128 else rc_complex32(v.real, v.imag)


def real_out(dtype):
    """ctypes scalar receiving a real result (norms, relative differences) for `dtype`."""
    import torch

    return ctypes.c_double() if real_dtype(dtype) == torch.float64 else ctypes.c_float()


def mat(t) -> rc_matrix:
    """ndarray-style view descriptor of a 2-D (or 1-D, taken as n x 1) CUDA tensor."""
    if t is None:
        return rc_matrix(None, 0, 0, 0, 0)
    if not t.is_cuda:
        raise HipRuntimeError("expected a tensor in device memory")
    if t.dim() == 1:
        return rc_matrix(t.data_ptr(), t.shape[0], 1, t.stride(0), 1)
    if t.dim() != 2:
        raise AssertionError("expected a matrix")
    return rc_matrix(t.data_ptr(), t.shape[0], t.shape[1], t.stride(0), t.stride(1))


class Context:
    """One rc_context: a device, a HIP stream and a workspace arena."""

    def __init__(self, device: Optional[int] = None, stream_ptr: Optional[int] = None):
        import torch

        if not torch.cuda.is_available():
            raise HipRuntimeError("no MI355X / HIP device is visible: the engine has no CPU path")
        self.device = torch.cuda.current_device() if device is None else int(device)
        if stream_ptr is None:
            stream_ptr = torch.cuda.current_stream(self.device).cuda_stream
        self.stream_ptr = int(stream_ptr)
        self._h = ctypes.c_void_p()
        st = lib().rc_create(ctypes.byref(self._h), ctypes.c_int32(self.device), ctypes.c_void_p(self.stream_ptr))
        if st != RC_OK:
            raise HipRuntimeError(f"rc_create failed with status {st}")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().rc_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, status: int):
        if status != RC_OK:
            msg = lib().rc_last_error_message(self._h)
            msg = msg.decode() if msg else ""
            raise _STATUS_EXC.get(status, RustyCompressionError)(msg)

    def call(self, name: str, *args):
        fn = getattr(lib(), name)
        self.check(fn(self._h, *args))

    def synchronize(self):
        self.check(lib().rc_synchronize(self._h))

    def set_option(self, option: int, value: int):
        """rc_set_option; option ids: RC_OPT_TALL_SKINNY_FAST_PATH = 1."""
        self.check(lib().rc_set_option(self._h, ctypes.c_int32(option), ctypes.c_int64(value)))

    def get_health(self) -> int:
        """rc_get_health: failure bits of fast paths that could not fall back (graph capture); clears them."""
        w = ctypes.c_int32(0)
        self.check(lib().rc_get_health(self._h, ctypes.byref(w)))
        return int(w.value)

    def reserve_workspace(self, nbytes: int):
        self.check(lib().rc_reserve_workspace(self._h, ctypes.c_size_t(nbytes)))


_contexts: Dict[Tuple[int, int], Context] = {}


def default_context() -> Context:
    """Context bound to torch's current device and current stream."""
    import torch

    dev = torch.cuda.current_device() if torch.cuda.is_available() else 0
    sp = torch.cuda.current_stream(dev).cuda_stream if torch.cuda.is_available() else 0
    key = (dev, int(sp))
    ctx = _contexts.get(key)
    if ctx is None:
        ctx = Context(dev, sp)
        _contexts[key] = ctx
    return ctx


def i64p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(None)
