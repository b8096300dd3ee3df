import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rusty_compression_amd as rc
g = torch.Generator(device="cuda").manual_seed(4)
u = torch.linalg.qr(torch.randn(16384, 4096, dtype=torch.float64, device="cuda", generator=g)).Q
v = torch.linalg.qr(torch.randn(4096, 4096, dtype=torch.float64, device="cuda", generator=g)).Q
sig = torch.logspace(0, -10, 4096, dtype=torch.float64, device="cuda")
mat = (u * sig) @ v.T
del u, v
q, hist = rc.sample_range_adaptive(mat, 1e-6, 64, rc.Rng(11))
b = rc.dot(q.t(), mat)
torch.cuda.synchronize()
t0 = time.perf_counter()
qq, r, ind = rc.pivoted_qr(b)
torch.cuda.synchronize()
print("pivoted_qr of B", tuple(b.shape), "s:", time.perf_counter() - t0, file=sys.stderr)
