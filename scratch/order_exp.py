"""Does the scatter's cost depend on the spatial coherence of the row order?  (same scene, rows sorted by azimuth)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "3d-reconstruction_amd")):
    sys.path.insert(0, p)
import torch, synth_scene
from gsplat import _backend as B
from gsplat.rendering import rasterization
dev = torch.device("cuda:0")
N, C, W, H = 1_000_000, 5, 1242, 375
g = synth_scene.make_gaussians(N, sh_degree=3)
vm, Ks = (t.to(dev) for t in synth_scene.make_cameras(C, W, H))
for name in ("random", "azimuth-sorted", "azimuth+elev-sorted"):
    if name == "random":
        perm = torch.arange(N)
    else:
        x, y, z = g["means"][:, 0], g["means"][:, 1], g["means"][:, 2] - 1.6
        az = torch.atan2(y, x)
        key = torch.floor((az + 3.1416) / 6.2832 * 512)
        if name.endswith("elev-sorted"):
            el = torch.atan2(z, torch.sqrt(x * x + y * y))
            key = key * 64 + torch.floor((el + 1.58) / 3.16 * 64)
        perm = torch.argsort(key, stable=True)
    p = {k: v[perm].to(dev) for k, v in g.items()}
    def step():
        with torch.no_grad():
            rasterization(p["means"], p["quats"], p["scales"], p["opacities"], p["colors"], vm, Ks, W, H, near_plane=0.1,
                          sh_degree=3, packed=True, render_mode="RGB+ED", rasterize_mode="antialiased")
    for _ in range(3): step()
    B.profiling_start()
    for _ in range(5): step()
    k = B.profiling_stop()
    print(name, json.dumps({a.replace("gs_", ""): round(b[1], 4) for a, b in k.items()}))
