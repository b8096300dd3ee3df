"""Per-phase s_memtime totals of k_wq_coop (workgroup 0); needs the diagnostic build tools/_dbg/librc_timing.so:
    cd rusty_compression_amd/csrc && make && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DRC_COOP_TIMING -c kernels_wqcoop.hip -o /tmp/coop_t.o \\
      && mkdir -p ../../tools/_dbg && hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_dbg/librc_timing.so /tmp/coop_t.o $(ls _build/*.o | grep -v kernels_wqcoop)
Round-1 reading (128 x 8192 f64, cycles per Householder step, stamp cost ~470 each included): local candidate + barrier 2.9k,
post column + wait for the stores 2.9k, header post + poll (= grid barrier incl. skew) 5.3k, fetch pivot column 0.9k,
?larfg + reflector 2.9k, apply + norm down-date 2.8k.  Tried without gain: self-validating column words (no store wait),
rcp/rsq + Newton instead of IEEE division/sqrt in ?larfg and the down-date."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rusty_compression_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "_dbg", "librc_timing.so")
import torch
import rusty_compression_amd as rc

a = torch.randn(128, 8192, dtype=torch.float64, device="cuda")
for _ in range(3):
    q, r, ind = rc.pivoted_qr(a)
    torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 8)()
_lib.lib().rc_debug_coop_timing(out)
names = ["A: local candidate + sync", "post column + sync (store complete)", "header post + poll + sync", "fetch pivot column + sync", "larfg + reflector", "C: apply + norms"]
tot = sum(out[i] for i in range(6))
for i in range(6):
    print(f"{names[i]:40s} {out[i]:9d} ticks {100.0 * out[i] / tot:5.1f} %  per step {out[i] / 128:7.1f}")
print("total per step", tot / 128)
