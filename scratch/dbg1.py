import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, '3d-reconstruction_amd')
import synth_scene
from gsplat.rendering import rasterization
dev = torch.device('cuda:0')
W,H,C,N = 120,80,3,3000
g = synth_scene.make_gaussians(N, sh_degree=None, extent=12.0)
vm, Ks = synth_scene.make_cameras(C, W, H)
gg = {k: v.to(dev) for k,v in g.items()}
for mode in ["RGB", "RGB+ED", "RGB+D"]:
    rc, ra, meta = rasterization(gg["means"], gg["quats"], gg["scales"], gg["opacities"], gg["colors"], vm.to(dev), Ks.to(dev), W, H, packed=False, render_mode=mode)
    vis = (meta["radii"]>0)
    c, n = torch.nonzero(vis)[5].tolist()
    print(mode, rc.shape, 'colors in', gg["colors"][n].tolist(), 'depth', meta["depths"][c,n].item())
    # raw record
    sp = rc.grad_fn
    print(' rc[0,40,60]=', rc[0,40,60].tolist(), 'nan?', torch.isnan(rc).any().item())
