"""Host-side cost of one rasterization() fwd+bwd: a scene so small that the GPU work is negligible."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-reconstruction_amd")]
import torch
import synth_scene
from gsplat.rendering import rasterization
dev = torch.device("cuda:0")
for N, W, H in ((2000, 160, 96), (100_000, 310, 94), (1_000_000, 1242, 375)):
    g = {k: v.to(dev).requires_grad_(True) for k, v in synth_scene.make_gaussians(N, sh_degree=None, extent=10.0).items()}
    vm, Ks = (t.to(dev) for t in synth_scene.make_cameras(1, W, H))
    def step():
        for v in g.values():
            v.grad = None
        rc, ra, info = rasterization(g["means"], g["quats"], g["scales"], g["opacities"], g["colors"], vm, Ks, W, H,
                                     near_plane=0.1, packed=False, absgrad=True, render_mode="RGB+ED")
        (rc.sum() + ra.sum()).backward()
    for _ in range(5):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        step()
    torch.cuda.synchronize()
    print(f"N={N} {W}x{H}: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per fwd+bwd (wall)")
