"""Which record slots differ between the batched (C=5, rows prefetched) and per-camera (C=1, rows on demand) projection?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-reconstruction_amd")]
import torch
import synth_scene
from gsplat.cuda import _wrapper as Wm
from gsplat import _backend as B
dev = torch.device("cuda:0")
N, C, W, H = 200_000, 5, 1242, 375
g = {k: v.to(dev) for k, v in synth_scene.make_gaussians(N, sh_degree=3, seed=42).items()}
vm, Ks = (t.to(dev) for t in synth_scene.make_cameras(C, W, H))
def proj(vm, Ks):
    C = vm.shape[0]
    params = Wm._make_params(N, C, W, H, 16, 0.3, 0.1, 1e10, 0.0, True, B.GS_COLOR_SH, 3, 3, 16, True)
    with torch.no_grad():
        return Wm._ProjectSplats.apply(g["means"], g["quats"], g["scales"], g["opacities"], g["colors"], vm, Ks, None, params, None)
import inspect
print(inspect.signature(Wm._make_params))
sp5, m5, r5, t5 = proj(vm, Ks)
for c in range(C):
    sp1, m1, r1, t1 = proj(vm[c:c+1], Ks[c:c+1])
    vis = (r5[c] > 0) if r5.dim() == 2 else (r5[c] > 0).any(-1)
    a, b = sp5[c][vis.reshape(-1)], sp1[0][vis.reshape(-1)]
    diff = (a != b)
    print(f"camera {c}: visible {int(vis.sum())}; slots differing (count per slot): {diff.sum(0).tolist()}; radii equal {torch.equal(r5[c], r1[0])}")
