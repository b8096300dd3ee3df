"""Developer probe: is the skinny f64 GEMM limited by HBM access pattern or by the CU side?
Times Y = A Omega for A of different heights (small A stays in the 256 MiB Infinity Cache)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rusty_compression_amd as rc
from rusty_compression_amd import _lib
ctx = _lib.default_context(); lib = _lib.lib()
n, l = 8192, 133
om = torch.empty((n, 134), dtype=torch.float64, device="cuda")[:, :l]
om.copy_(rc.random_gaussian((n, l), rc.Rng(2)))
for m in (1024, 2048, 4096, 8192, 16384):
    a = rc.random_gaussian((m, n), rc.Rng(1))
    for _ in range(3): y = rc.matmat(a, om)
    lib.rc_profile_enable(ctx._h, 1); lib.rc_profile_reset(ctx._h)
    for _ in range(10): y = rc.matmat(a, om)
    cnt = ctypes.c_int32(0); lib.rc_profile_count(ctx._h, ctypes.byref(cnt))
    for i in range(cnt.value):
        name = ctypes.create_string_buffer(192); ms = ctypes.c_double(0); calls = ctypes.c_int64(0)
        lib.rc_profile_get(ctx._h, i, name, 192, ctypes.byref(ms), ctypes.byref(calls))
        nm = name.value.decode(); avg = ms.value / max(calls.value, 1)
        if "k_gemm_mfma" in nm:
            fl = 2.0 * m * l * n
            print(f"m={m:6d} A={m*n*8/2**20:7.0f} MiB  {avg*1e3:8.1f} us  {fl/avg/1e9:6.2f} TF/s  A-stream {m*n*8/avg/1e6:8.1f} GB/s  [{os.environ.get('RC_GEMM_F64X4','1')}]")
    lib.rc_profile_enable(ctx._h, 0)
    del a
