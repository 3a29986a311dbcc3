import sys, torch, math
sys.path.insert(0, '.'); sys.path.insert(0, '3d-reconstruction_amd')
import synth_scene
from gsplat.rendering import rasterization
dev = torch.device('cuda:0')
N,C,W,H = 1_000_000,5,1242,375
g = synth_scene.make_gaussians(N, sh_degree=3)
vm, Ks = synth_scene.make_cameras(C, W, H)
p = {k: v.to(dev) for k,v in g.items()}
rc, ra, meta = rasterization(p["means"], p["quats"], p["scales"], p["opacities"], p["colors"], vm.to(dev), Ks.to(dev), W, H, near_plane=0.1, sh_degree=3, packed=False, render_mode="RGB+ED", absgrad=True, rasterize_mode="antialiased")
offs = meta["isect_offsets"].reshape(-1).long()
I = meta["flatten_ids"].numel()
ends = torch.cat([offs[1:], torch.tensor([I], device=dev)])
cnt = (ends - offs).float()
TH, TW = meta["tile_height"], meta["tile_width"]
last = meta["last_ids"].long()  # C,H,W
# per tile max last id
pad = torch.zeros(C, TH*16, TW*16, device=dev, dtype=torch.long)
pad[:, :H, :W] = last
tl = pad.reshape(C, TH, 16, TW, 16).permute(0,1,3,2,4).reshape(C*TH*TW, 256).max(dim=1).values
proc = torch.clamp(tl - offs + 1, min=0).float()
proc = torch.where(cnt > 0, proc, torch.zeros_like(proc))
print("tiles", cnt.numel(), "I", I, "mean cnt", cnt.mean().item(), "max cnt", cnt.max().item(), "p99", cnt.quantile(0.99).item())
print("processed(bwd) total", proc.sum().item(), "frac of I", proc.sum().item()/I, "max", proc.max().item(), "p99", proc.quantile(0.99).item(), "mean", proc.mean().item())
for c in range(C):
    cc = cnt.reshape(C,-1)[c]; pp = proc.reshape(C,-1)[c]
    print("cam", c, "isects", cc.sum().item(), "proc", pp.sum().item(), "max cnt", cc.max().item(), "max proc", pp.max().item(), "alpha mean", ra[c].mean().item())
tpg = meta["tiles_per_gauss"].float()
v = tpg[tpg>0]
print("tiles per visible gauss: mean", v.mean().item(), "max", v.max().item(), "p50", v.quantile(0.5).item(), "p99", v.quantile(0.99).item(), "frac>16", (v>16).float().mean().item())
r = meta["radii"][meta["radii"]>0].float()
print("radii mean", r.mean().item(), "p50", r.quantile(0.5).item(), "p99", r.quantile(0.99).item(), "max", r.max().item())
hist = torch.histc(proc, bins=10, min=0, max=proc.max().item())
print("proc hist", hist.tolist())

# ---- exact per-tile cull potential: min sigma over the tile's pixel-centre rectangle
ids = meta["isect_ids"]
fl = meta["flatten_ids"].long()
tile_glob = (ids >> 32)  # cam<<tile_bits | tile
tile_bits = int(math.floor(math.log2(TH*TW))) + 1
cam = tile_glob >> tile_bits
tid = tile_glob & ((1 << tile_bits) - 1)
ty = (tid // TW).float(); tx = (tid % TW).float()
m2 = meta["means2d"].detach().reshape(-1, 2)[fl]
con = meta["conics"].reshape(-1, 3)[fl]
op = meta["opacities"].reshape(-1)[fl]
x0 = tx*16 + 0.5; x1 = torch.minimum(tx*16 + 15.5, torch.tensor(W-0.5, device=dev))
y0 = ty*16 + 0.5; y1 = torch.minimum(ty*16 + 15.5, torch.tensor(H-0.5, device=dev))
mx, my = m2[:,0], m2[:,1]
a, b, c = con[:,0], con[:,1], con[:,2]
def q(dx, dy): return 0.5*(a*dx*dx + c*dy*dy) + b*dx*dy
inside = (mx >= x0) & (mx <= x1) & (my >= y0) & (my <= y1)
best = torch.full_like(mx, float("inf"))
for xe in (x0, x1):
    dx = mx - xe
    dy = torch.clamp(-b*dx/c, my - y1, my - y0)
    best = torch.minimum(best, q(dx, dy))
for ye in (y0, y1):
    dy = my - ye
    dx = torch.clamp(-b*dy/a, mx - x1, mx - x0)
    best = torch.minimum(best, q(dx, dy))
best = torch.where(inside, torch.zeros_like(best), best)
amax = op * torch.exp(-best)
keep = amax >= (1/255.)
print("exact-cull: keep frac of I", keep.float().mean().item())
# among those actually processed in bwd (k <= tile max last id)
k = torch.arange(I, device=dev)
tlin = cam * (TH*TW) + tid
procmask = k <= tl[tlin]
print("processed frac", procmask.float().mean().item(), "keep among processed", (keep & procmask).float().sum().item()/procmask.float().sum().item())
kept_per_tile = torch.zeros(C*TH*TW, device=dev).index_add_(0, tlin[procmask & keep], torch.ones(int((procmask&keep).sum()), device=dev))
print("kept processed per tile: mean", kept_per_tile.mean().item(), "max", kept_per_tile.max().item(), "p99", kept_per_tile.quantile(0.99).item())
