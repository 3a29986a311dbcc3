"""Standalone spherical_harmonics fwd/bwd (the reference's call pattern: vanilla.py:369 etc.), n rows, K=16, degree 3."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "3d-reconstruction_amd")):
    sys.path.insert(0, p)
import torch
from gsplat.cuda._wrapper import spherical_harmonics
dev = torch.device("cuda:0")
n, K = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000, 16
g = torch.Generator().manual_seed(0)
dirs = torch.randn(n, 3, generator=g).to(dev)
coeffs = (0.1 * torch.randn(n, K, 3, generator=g)).to(dev).requires_grad_(True)
w = torch.randn(n, 3, generator=g).to(dev)
def run(deg):
    for _ in range(3):
        out = spherical_harmonics(deg, dirs, coeffs); out.backward(w); coeffs.grad = None
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(10):
        ev[0].record(); out = spherical_harmonics(deg, dirs, coeffs); ev[1].record(); out.backward(w); ev[2].record()
        torch.cuda.synchronize(); tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2]); coeffs.grad = None
    return round(tf / 10, 4), round(tb / 10, 4)
res = {f"deg{d}": run(d) for d in (3, 1)}
fb = n * (12 + 12 * K + 12); bb = n * (12 + 12 + 12 * K)
print(json.dumps({"n": n, "ms_fwd_bwd": res, "fwd_GBps_deg3": round(fb / res["deg3"][0] / 1e6, 1), "bwd_GBps_deg3": round(bb / res["deg3"][1] / 1e6, 1)}))
