"""Developer tool: rc_profile breakdown of a few calls (what the pieces of a call cost)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rusty_compression_amd as rc
from rusty_compression_amd import _lib
ctx = _lib.default_context(); lib = _lib.lib()
def prof(tag, fn, top=9):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); t = time.perf_counter() - t0
    lib.rc_profile_enable(ctx._h, 1); lib.rc_profile_reset(ctx._h)
    fn()
    cnt = ctypes.c_int32(0); lib.rc_profile_count(ctx._h, ctypes.byref(cnt))
    print(f"== {tag}: {t*1e3:.3f} ms")
    rows = []
    for i in range(cnt.value):
        name = ctypes.create_string_buffer(192); ms = ctypes.c_double(0); calls = ctypes.c_int64(0)
        lib.rc_profile_get(ctx._h, i, name, 192, ctypes.byref(ms), ctypes.byref(calls))
        rows.append((ms.value, calls.value, name.value.decode()))
    for ms, calls, nm in sorted(rows, reverse=True)[:top]: print(f"   {ms:9.3f} ms x{calls:<4d} {nm}")
    lib.rc_profile_enable(ctx._h, 0)
y32 = rc.random_gaussian((2048, 133), rc.Rng(3), torch.float32)
prof("f32 pivoted_qr 2048x133", lambda: rc.pivoted_qr(y32))
def cg(shape, dt, seed):
    real = torch.float64 if dt == torch.complex128 else torch.float32
    return torch.complex(rc.random_gaussian(shape, rc.Rng(seed), real), rc.random_gaussian(shape, rc.Rng(seed + 9), real))
b = cg((128, 2048), torch.complex128, 4)
prof("c64 compute_svd 128x2048", lambda: rc.compute_svd(b), top=14)
br = rc.random_gaussian((128, 2048), rc.Rng(4), torch.float64)
prof("f64 compute_svd 128x2048", lambda: rc.compute_svd(br), top=8)
