import os, sys, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-reconstruction_amd")]
import torch
import synth_scene
from gsplat.rendering import rasterization
dev = torch.device("cuda:0")
N, W, H = (int(os.environ.get("PH_N", 2000)), int(os.environ.get("PH_W", 160)), int(os.environ.get("PH_H", 96)))
g = {k: v.to(dev).requires_grad_(True) for k, v in synth_scene.make_gaussians(N, sh_degree=None, extent=10.0).items()}
vm, Ks = (t.to(dev) for t in synth_scene.make_cameras(1, W, H))
def fwd():
    for v in g.values():
        v.grad = None
    return rasterization(g["means"], g["quats"], g["scales"], g["opacities"], g["colors"], vm, Ks, W, H,
                                 near_plane=0.1, packed=False, absgrad=True, render_mode="RGB+ED")
def step():
    rc, ra, info = fwd()
    (rc.sum() + ra.sum()).backward()
for _ in range(20):
    step()
torch.cuda.synchronize()
import time
for name, fn in (("fwd only", fwd), ("fwd+bwd", step)):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200):
        fn()
    torch.cuda.synchronize(); print(name, (time.perf_counter() - t0) / 200 * 1e3, "ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(200):
    step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(32); print(s.getvalue()[:7000])
