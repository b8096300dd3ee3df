import os, sys, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rusty_compression_amd as rc
from rusty_compression_amd import _lib, batch
a = rc.random_gaussian((4096, 4096), rc.Rng(500), torch.float32)
ctx = _lib.default_context(); lib = _lib.lib()
for _ in range(2): batch.column_id_rank(a, 64)
torch.cuda.synchronize()
t0=time.perf_counter(); batch.column_id_rank(a, 64); torch.cuda.synchronize(); print("one call ms", (time.perf_counter()-t0)*1e3)
lib.rc_profile_enable(ctx._h, 1); lib.rc_profile_reset(ctx._h)
batch.column_id_rank(a, 64)
cnt = ctypes.c_int32(0); lib.rc_profile_count(ctx._h, ctypes.byref(cnt))
for i in range(cnt.value):
    name = ctypes.create_string_buffer(192); ms = ctypes.c_double(0); calls = ctypes.c_int64(0)
    lib.rc_profile_get(ctx._h, i, name, 192, ctypes.byref(ms), ctypes.byref(calls))
    print(f"{name.value.decode():70s} {ms.value:9.3f} ms  x{calls.value}")
