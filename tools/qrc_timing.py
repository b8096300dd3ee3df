"""Per-phase s_memtime totals of the cooperative blocked-QRCP panel k_qrb_coop (workgroup 0, last launch); needs the diagnostic
build tools/_dbg/librc_qrc_timing.so:
    cd rusty_compression_amd/csrc && make && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DRC_QRC_TIMING -c kernels_qrblk.hip -o /tmp/qrc_t.o \\
      && mkdir -p ../../tools/_dbg && hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_dbg/librc_qrc_timing.so /tmp/qrc_t.o $(ls _build/*.o | grep -v kernels_qrblk) -ldl"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rusty_compression_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "_dbg", "librc_qrc_timing.so")
import torch
import rusty_compression_amd as rc
from rusty_compression_amd import batch

a = rc.random_gaussian((4096, 4096), rc.Rng(500), torch.float32)
for _ in range(3):
    batch.column_id_rank(a, 64)
    torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 8)()
_lib.lib().rc_debug_qrc_timing(out)
names = ["A: local candidate + sync", "post column + sync (store complete)", "header post + poll + sync", "fetch pivot column + sync", "larfg + reflector + sync", "C: apply + norms"]
tot = sum(out[i] for i in range(6))
for i in range(6):
    print(f"{names[i]:40s} {out[i]:9d} ticks {100.0 * out[i] / max(tot, 1):5.1f} %  per step {out[i] / 32:7.1f}")
print("total per step", tot / 32)
