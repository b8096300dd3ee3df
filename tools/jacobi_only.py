"""Runs the LDS Jacobi SVD of a 128 x 128 f64 core a few times (target of tools/pmc_jacobi.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rusty_compression_amd as rc
a = torch.randn(128, 128, dtype=torch.float64, device="cuda")
for _ in range(6):
    rc.compute_svd(a)
torch.cuda.synchronize()
