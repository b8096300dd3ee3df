"""Per-phase s_memtime totals of one round of k_jacobi_lds (wave 0 of the producer workgroup); needs the diagnostic build:
    make -C rusty_compression_amd/csrc && cd rusty_compression_amd/csrc && \\
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DRC_JAC_TIMING -c kernels_svd.hip -o /tmp/svd_t.o && mkdir -p ../../tools/_dbg && \\
    hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_dbg/librc_jac_timing.so /tmp/svd_t.o $(ls _build/*.o | grep -v kernels_svd)
Every stamp drains the wave's LDS queue and costs a few hundred cycles itself (the other 15 waves run unstamped and wait at the
barrier), so the sum is larger than the real round; the split is what counts."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rusty_compression_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "_dbg", "librc_jac_timing.so")
import torch
import rusty_compression_amd as rc

a = torch.randn(128, 128, dtype=torch.float64, device="cuda")
for _ in range(3):
    rc.compute_svd(a)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 8)()
_lib.lib().rc_debug_jacobi_timing(out)
names = ["LDS reads + dot products", "group reduction (DPP)", "rotation parameters", "apply + write-back (queue drained)", "record published", "barrier + loop control"]
rounds = max(int(out[6]), 1)
tot = sum(out[i] for i in range(6))
for i in range(6):
    print(f"{names[i]:40s} {out[i]:10d} ticks {100.0 * out[i] / tot:5.1f} %  per round {out[i] / rounds:8.1f}")
print("rounds", rounds, "total per round", tot / rounds)
