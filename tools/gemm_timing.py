"""Per-phase s_memtime totals of the f64 GEMM main loop (wave 0 of workgroup (0,0)); needs the diagnostic build
tools/_dbg/librc_timing.so:
    cd rusty_compression_amd/csrc && make && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DRC_GEMM_TIMING -c kernels_gemm.hip -o /tmp/gemm_t.o \\
      && mkdir -p ../../tools/_dbg && hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_dbg/librc_timing.so /tmp/gemm_t.o $(ls _build/*.o | grep -v kernels_gemm)
Round-1 reading (committed kernel, cycles per 16-deep K tile of 11.8k): global-load issue 1.1k, fragment-read waits 3.6k,
MFMA issue 4.6k, vmcnt wait + ds_write 1.5k, barrier 1.0k (sketch) / 0.4k (projection)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rusty_compression_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "_dbg", "librc_timing.so")
import torch
import rusty_compression_amd as rc

n, l, k = 8192, 133, 128
a = rc.random_gaussian((n, n), rc.Rng(1))
om = torch.empty((n, 134), dtype=torch.float64, device="cuda")[:, :l]
om.copy_(rc.random_gaussian((n, l), rc.Rng(2)))
q = torch.empty((n, k), dtype=torch.float64, device="cuda")
q.copy_(rc.random_gaussian((n, k), rc.Rng(3)))
lib = _lib.lib()
names = ["global-load issue", "fragment reads (wait)", "MFMA issue", "vmcnt wait + ds_write", "barrier"]
for label, fn in (("sketch  (133 x 8192 x 8192, <1,1,136,256,16,2,4>)", lambda: rc.matmat(a, om)), ("project (128 x 8192 x 8192, <1,0,128,256,16,1,8>)", lambda: rc.dot(q.t(), a))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 8)()
    lib.rc_debug_gemm_timing(out)
    tot = sum(out[i] for i in range(5))
    print(label, "k-tiles", out[5], "total ticks", tot)
    for i in range(5):
        print(f"   {names[i]:26s} {out[i]:10d} ticks  {100.0 * out[i] / tot:5.1f} %   per k-tile {out[i] / max(out[5], 1):8.1f}")
