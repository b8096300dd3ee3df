import os, sys, ctypes
os.environ["RC_COOP_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rusty_compression_amd as rc
a = torch.randn(128, 8192, dtype=torch.float64, device="cuda")
for _ in range(3):
    q, r, ind = rc.pivoted_qr(a)
    torch.cuda.synchronize()
