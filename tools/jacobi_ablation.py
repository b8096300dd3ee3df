"""Time of the LDS Jacobi kernel (128 x 128 f64 core, 9 forced sweeps, non-fused and fused) with parts of its round removed.
Needs tools/jacobi_ablation.sh.  One process per variant (the library is chosen at import)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {32: "all of the round (rotate always)", 33: "- rotation records", 34: "- rcp/rsqrt of the rotation", 36: "- write-back to LDS",
         40: "- group reduction (DPP)", 48: "- LDS reads", 63: "loop + barrier only"}
if len(sys.argv) > 1:
    abl = int(sys.argv[1])
    sys.path.insert(0, ROOT)
    from rusty_compression_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "_dbg", f"librc_jac_abl_{abl}.so")
    import torch
    import rusty_compression_amd as rc
    import ctypes
    a = torch.randn(128, 128, dtype=torch.float64, device="cuda")
    ctx, lib = rc.default_context(), _lib.lib()
    for _ in range(3):
        rc.compute_svd(a)
    ctx.synchronize()
    lib.rc_profile_enable(ctx._h, 1); lib.rc_profile_reset(ctx._h)
    for _ in range(10):
        rc.compute_svd(a)
    cnt = ctypes.c_int32(0)
    ctx.check(lib.rc_profile_count(ctx._h, ctypes.byref(cnt)))
    for i in range(cnt.value):
        name = ctypes.create_string_buffer(192); ms = ctypes.c_double(0); calls = ctypes.c_int64(0)
        lib.rc_profile_get(ctx._h, i, name, 192, ctypes.byref(ms), ctypes.byref(calls))
        if name.value.decode().startswith("op:jacobi_svd"):
            us = ms.value / max(calls.value, 1) * 1000
            print(f"abl {abl:2d} fused={os.environ.get('RC_JACOBI_FUSED_V', '1')} {NAMES[abl]:36s} {us:8.1f} us  per round {us * 1000 / (9 * 127):7.1f} ns", flush=True)
else:
    for fused in ("0", "1"):
        for abl in NAMES:
            if fused == "1" and (abl & 1):
                continue  # no records: the consumer would spin to its bound
            env = dict(os.environ, RC_JACOBI_MAX_SWEEPS="9", RC_JACOBI_FUSED_V=fused)
            subprocess.run([sys.executable, __file__, str(abl)], env=env, check=False)
