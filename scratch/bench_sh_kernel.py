"""Pure kernel time of the standalone SH forward / backward (HIP events around the native launches)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-reconstruction_amd")]
import torch
from gsplat import _backend as B
from gsplat.cuda._wrapper import spherical_harmonics
dev = torch.device("cuda:0")
n, K = 3_000_000, 16
g = torch.Generator().manual_seed(0)
dirs = torch.randn(n, 3, generator=g).to(dev)
co = (torch.randn(n, K, 3, generator=g) * 0.1).to(dev).requires_grad_(True)
w = torch.randn(n, 3, generator=g).to(dev)
for _ in range(3):
    co.grad = None
    (spherical_harmonics(3, dirs, co) * w).sum().backward()
B.profiling_start()
for _ in range(10):
    co.grad = None
    (spherical_harmonics(3, dirs, co) * w).sum().backward()
res = B.profiling_stop()
out = {k: round(v[1], 4) for k, v in res.items()}
out["fwd_TBps"] = round(n * (12 * K + 24) / (res["gs_sh_fwd"][1] * 1e-3) / 1e12, 2)
out["bwd_TBps"] = round(n * (12 * K + 24) / (res["gs_sh_bwd"][1] * 1e-3) / 1e12, 2)
print(json.dumps(out))
