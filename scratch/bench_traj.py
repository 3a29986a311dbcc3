"""BASELINE configs[4] on one rank: 6M Gaussians, this rank's 75 of 600 views at 1920x1080, packed, render-only."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "3d-reconstruction_amd")):
    sys.path.insert(0, p)
import torch, synth_scene
from gsplat.harness import render_trajectory
dev = torch.device("cuda:0")
N, W, H, V, WORLD = 6_000_000, 1920, 1080, 600, 8
g = {k: v.to(dev) for k, v in synth_scene.make_gaussians(N, sh_degree=3).items()}
cams = [synth_scene.make_cameras(1, W, H, frame=f % 120, origin_x=-30.0) for f in range(V)]
vm = torch.cat([c[0] for c in cams]).to(dev); Ks = torch.cat([c[1] for c in cams]).to(dev)
res = {}
for vpc in (1, 3, 5):
    out = None
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out, rng = render_trajectory(g["means"], g["quats"], g["scales"], g["opacities"], g["colors"], vm, Ks, W, H,
                                     sh_degree=3, views_per_call=vpc, rank=0, world_size=WORLD, near_plane=0.1, out=out)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    res[f"views_per_call={vpc}"] = {"views": len(rng), "seconds": round(dt, 4), "views_per_s": round(len(rng) / dt, 1),
                                    "Mpix_s_incl_D2H": round(len(rng) * W * H / dt / 1e6, 1)}
print(json.dumps(res))
