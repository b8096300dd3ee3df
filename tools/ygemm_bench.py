"""Developer tool: the two panel-end products of the blocked QRCP on the cfg5 shape (f32): Y = V^T A (32 x 4096 x 4096, both
operands K-contiguous) and the block update A^T -= F V^T (4096 x 4064 x 32), through rc.dot with the library's kernel timers."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rusty_compression_amd as rc
from rusty_compression_amd import _lib
n = 4096
a = rc.random_gaussian((n, n), rc.Rng(1), torch.float32)          # row-major (n x n): a.t() is the column-major matrix
w = a.t()                                                         # column-major view: w[r, c]
v = rc.random_gaussian((32, n), rc.Rng(2), torch.float32).t()     # column-major n x 32
ctx = _lib.default_context(); lib = _lib.lib()
for _ in range(2):
    y = rc.dot(v.t(), w)
lib.rc_profile_enable(ctx._h, 1); lib.rc_profile_reset(ctx._h)
for _ in range(6):
    y = rc.dot(v.t(), w)
cnt = ctypes.c_int32(0); lib.rc_profile_count(ctx._h, ctypes.byref(cnt))
tag = " ".join(f"{k}={os.environ[k]}" for k in sorted(os.environ) if k.startswith("RC_GEMM"))
for i in range(cnt.value):
    name = ctypes.create_string_buffer(192); ms = ctypes.c_double(0); calls = ctypes.c_int64(0)
    lib.rc_profile_get(ctx._h, i, name, 192, ctypes.byref(ms), ctypes.byref(calls))
    print(f"[{tag}] {name.value.decode():60s} {ms.value / max(calls.value,1) * 1e3:8.1f} us")
ref = v.t().double() @ w.double()
print("max err", float((y.double() - ref).abs().max()), "GB/s of A per call at the time above")
