"""Developer tool: which path a tall-skinny f32 / f64 pivoted QR takes and what its pieces cost (rc_profile timers)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rusty_compression_amd as rc
from rusty_compression_amd import _lib
ctx = _lib.default_context(); lib = _lib.lib()
for dt in (torch.float32, torch.float64):
    for shape in ((2048, 133), (8192, 133), (2048, 64)):
        y = rc.random_gaussian(shape, rc.Rng(3), dt)
        rc.pivoted_qr(y); torch.cuda.synchronize()
        t0 = time.perf_counter(); rc.pivoted_qr(y); torch.cuda.synchronize(); t = time.perf_counter() - t0
        lib.rc_profile_enable(ctx._h, 1); lib.rc_profile_reset(ctx._h)
        rc.pivoted_qr(y)
        cnt = ctypes.c_int32(0); lib.rc_profile_count(ctx._h, ctypes.byref(cnt))
        print(f"== {dt} {shape}: {t*1e3:.3f} ms, health {ctx.get_health()}")
        rows = []
        for i in range(cnt.value):
            name = ctypes.create_string_buffer(192); ms = ctypes.c_double(0); calls = ctypes.c_int64(0)
            lib.rc_profile_get(ctx._h, i, name, 192, ctypes.byref(ms), ctypes.byref(calls))
            rows.append((ms.value, calls.value, name.value.decode()))
        for ms, calls, nm in sorted(rows, reverse=True)[:8]: print(f"   {ms:9.3f} ms x{calls:<4d} {nm}")
        lib.rc_profile_enable(ctx._h, 0)
