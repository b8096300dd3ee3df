"""Developer tool: what the kernels of a cfg5 batch cost UNDER the batch's own concurrency, without a tracer: the library's event
timers (rc_profile_*) on every lane's context during one batch of 8, summed per kernel / op name, beside the same sums of one
lone rc_column_id_rank call.  (A rocprofv3 kernel trace serialises these lanes: mean kernels in flight 1.1.)
    python tools/batch_prof.py [lanes]"""
import ctypes, os, sys, collections
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rusty_compression_amd as rc
from rusty_compression_amd import _lib, batch
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 8
lib = _lib.lib()
mats = [rc.random_gaussian((4096, 4096), rc.Rng(500 + i), torch.float32) for i in range(8)]
def read(ctx):
    cnt = ctypes.c_int32(0); lib.rc_profile_count(ctx._h, ctypes.byref(cnt))
    out = {}
    for i in range(cnt.value):
        name = ctypes.create_string_buffer(192); ms = ctypes.c_double(0); calls = ctypes.c_int64(0)
        lib.rc_profile_get(ctx._h, i, name, 192, ctypes.byref(ms), ctypes.byref(calls))
        out[name.value.decode()] = (ms.value, calls.value)
    return out
for _ in range(3): batch.batch_column_id_packed(mats, 64, lanes=lanes)
torch.cuda.synchronize()
ctxs = batch._LanePool.get(torch.cuda.current_device(), lanes)
for c in ctxs: lib.rc_profile_enable(c._h, 1); lib.rc_profile_reset(c._h)
batch.batch_column_id_packed(mats, 64, lanes=lanes)
torch.cuda.synchronize()
tot = collections.defaultdict(lambda: [0.0, 0])
for c in ctxs:
    for k, (ms, n) in read(c).items(): tot[k][0] += ms; tot[k][1] += n
    lib.rc_profile_enable(c._h, 0)
ctx = _lib.default_context()
batch.column_id_rank(mats[0], 64); torch.cuda.synchronize()
lib.rc_profile_enable(ctx._h, 1); lib.rc_profile_reset(ctx._h)
batch.column_id_rank(mats[0], 64); torch.cuda.synchronize()
lone = read(ctx); lib.rc_profile_enable(ctx._h, 0)
print(f"{'timer':72s} {'in batch: ms per matrix':>24s} {'lone: ms':>10s}")
for k, (ms, n) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print(f"{k[:72]:72s} {ms / 8:24.4f} {lone.get(k, (0, 0))[0]:10.4f}")
