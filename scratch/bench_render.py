"""Render-only throughput at BASELINE configs[4] shape: 6M Gaussians, 1920x1080 views, packed=True, no grad."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "3d-reconstruction_amd")):
    sys.path.insert(0, p)
import torch, synth_scene
from gsplat import _backend as B
from gsplat.rendering import rasterization
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6_000_000
C = int(sys.argv[2]) if len(sys.argv) > 2 else 1
packed = (sys.argv[3] != "dense") if len(sys.argv) > 3 else True
W, H = 1920, 1080
g = synth_scene.make_gaussians(N, sh_degree=3)
p = {k: v.to(dev) for k, v in g.items()}
def views(f0):
    vms, Kss = [], []
    for i in range(C):
        vm, Ks = synth_scene.make_cameras(1, W, H, frame=f0 + i)
        vms.append(vm); Kss.append(Ks)
    return torch.cat(vms).to(dev), torch.cat(Kss).to(dev)
cams = [views(f * C) for f in range(8)]
def step(i):
    vm, Ks = cams[i % len(cams)]
    with torch.no_grad():
        rc, ra, meta = rasterization(p["means"], p["quats"], p["scales"], p["opacities"], p["colors"], vm, Ks, W, H,
                                     near_plane=0.1, sh_degree=3, packed=packed, render_mode="RGB")
    return rc, meta
for i in range(3): rc, meta = step(i)
I = meta["flatten_ids"].numel(); nnz = meta["radii"].numel() if packed else int((meta["radii"] > 0).sum())
B.profiling_start()
for i in range(4): step(i)
k = B.profiling_stop()
torch.cuda.synchronize(); t0 = time.perf_counter()
steps = 16
for i in range(steps): step(i)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
print(json.dumps({"N": N, "C": C, "packed": packed, "ms_per_call": round(dt * 1e3, 3), "Mpix_s": round(C * W * H / dt / 1e6, 1),
                  "views_per_s": round(C / dt, 1), "I": I, "visible": nnz,
                  "kernels_ms": {a: round(b[1] * b[0] / 4, 4) for a, b in k.items()}}))
