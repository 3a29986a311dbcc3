import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-reconstruction_amd")]
import torch, synth_scene
from gsplat.rendering import rasterization
dev = torch.device("cuda:0")
g = {k: v.to(dev) for k, v in synth_scene.make_gaussians(1_000_000, sh_degree=3, seed=42).items()}
vm, Ks = (t.to(dev) for t in synth_scene.make_cameras(5, 1242, 375))
with torch.no_grad():
    rc, ra, meta = rasterization(g["means"], g["quats"], g["scales"], g["opacities"], g["colors"], vm, Ks, 1242, 375,
                                 near_plane=0.1, sh_degree=3, packed=False, render_mode="RGB+ED", rasterize_mode="antialiased")
r = meta["radii"] > 0
print("visible per camera", r.float().mean(1).tolist(), "visible in any", float(r.any(0).float().mean()),
      "cameras per visible Gaussian", float(r.sum() / r.any(0).sum()))
w = r.any(0).reshape(-1)[: (1_000_000 // 64) * 64].reshape(-1, 64)
print("waves with no visible Gaussian", float((~w.any(1)).float().mean()))
