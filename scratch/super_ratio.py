"""How many (tile, splat) pairs remain if horizontally / vertically adjacent tiles share one list?  (culled lists, headline shape)"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-reconstruction_amd")]
import torch
import synth_scene
from gsplat.cuda import _wrapper as Wm
from gsplat import _backend as B
dev = torch.device("cuda:0")
comp = len(sys.argv) > 1 and sys.argv[1] == "composite"
N, C, W, H = (3_000_000, 1, 1920, 1280) if comp else (1_000_000, 5, 1242, 375)
TW, TH = math.ceil(W / 16), math.ceil(H / 16)
T = TW * TH
g = {k: v.to(dev) for k, v in synth_scene.make_gaussians(N, sh_degree=None, seed=42).items()}
vm, Ks = synth_scene.make_cameras(C, W, H)
params = Wm._make_params(N, C, W, H, 16, 0.3, 0.1, 1e10, 0.0, True, B.GS_COLOR_N, 3, None, 0, True)
with torch.no_grad():
    splats, means2d, radii, tpg = Wm._ProjectSplats.apply(g["means"], g["quats"], g["scales"], g["opacities"], g["colors"],
                                                          vm.to(dev), Ks.to(dev), None, params, None)
ids, fids, offs, _, _ = Wm.isect_tiles_binned(splats, tpg, TW, TH, front=False, cull_tile_size=16)
I = fids.numel()
tn_bits = int(math.floor(math.log2(T))) + 1
tile = (ids >> 32) & ((1 << tn_bits) - 1)
cam = ids >> (32 + tn_bits)
tx, ty = tile % TW, tile // TW
row = fids.long()
lens = torch.bincount((cam * T + tile), minlength=C * T)
print(f"16x16: pairs {I}, tiles {C*T}, mean list {I/(C*T):.0f}, max {int(lens.max())}, lists > 1024: {int((lens > 1024).sum())}, > 2048: {int((lens > 2048).sum())}")
for sx, sy in ((2, 1), (1, 2), (2, 2), (4, 1), (4, 2)):
    TWs, THs = -(-TW // sx), -(-TH // sy)
    st = (cam * THs + ty // sy) * TWs + tx // sx
    key = st * (C * N) + row
    u = torch.unique(key)
    lens = torch.bincount(u // (C * N), minlength=C * TWs * THs)
    print(f"{16*sx}x{16*sy}: pairs {u.numel()} ({u.numel()/I:.3f}), tiles {C*TWs*THs}, mean list {u.numel()/(C*TWs*THs):.0f}, max {int(lens.max())}, lists > 1024: {int((lens > 1024).sum())}, > 2048: {int((lens > 2048).sum())}")
