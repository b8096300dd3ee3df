"""Developer tool: time the two headline GEMM shapes through rc_gemm with the kernel-level
HIP-event timers.  Tile variants are selected with RC_GEMM_SKINNY_N / RC_GEMM_SKINNY_M /
RC_GEMM_TARGET_WGS / RC_GEMM_VEC (read once per process)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rusty_compression_amd as rc
from rusty_compression_amd import _lib

n, l, k = 8192, 133, 128
a = rc.random_gaussian((n, n), rc.Rng(1))
om = torch.empty((n, 134), dtype=torch.float64, device="cuda")[:, :l]
om.copy_(rc.random_gaussian((n, l), rc.Rng(2)))
q = torch.empty((n, k), dtype=torch.float64, device="cuda")
q.copy_(rc.random_gaussian((n, k), rc.Rng(3)))
ctx = _lib.default_context()
lib = _lib.lib()
for _ in range(2):
    y = rc.matmat(a, om); b = rc.conj_matmat(a, q).t()
lib.rc_profile_enable(ctx._h, 1); lib.rc_profile_reset(ctx._h)
for _ in range(int(os.environ.get("REPS", "5"))):
    y = rc.matmat(a, om)
    bt = rc.dot(q.t(), a)
cnt = ctypes.c_int32(0); lib.rc_profile_count(ctx._h, ctypes.byref(cnt))
tag = " ".join(f"{kk}={os.environ[kk]}" for kk in sorted(os.environ) if kk.startswith("RC_GEMM"))
for i in range(cnt.value):
    name = ctypes.create_string_buffer(192); ms = ctypes.c_double(0); calls = ctypes.c_int64(0)
    lib.rc_profile_get(ctx._h, i, name, 192, ctypes.byref(ms), ctypes.byref(calls))
    nm = name.value.decode(); avg = ms.value / max(calls.value, 1)
    if "k_gemm_mfma" in nm:
        dims = dict(x.split("=") for x in nm.split()[1:])
        fl = 2.0 * int(dims["M"]) * int(dims["N"]) * int(dims["K"])
        print(f"[{tag}] {nm}: {avg*1e3:.1f} us  {fl/avg/1e9:.2f} TF/s")
    else:
        print(f"[{tag}] {nm}: {avg*1e3:.1f} us")
# correctness spot check
ref = (a[:64].cpu().numpy() @ om.cpu().numpy())
import numpy as np
print("check", float(np.abs(y[:64].cpu().numpy() - ref).max()))
