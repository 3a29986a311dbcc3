"""Distribution of tile list lengths and walked lengths on the bench scene."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "3d-reconstruction_amd")):
    sys.path.insert(0, p)
import torch, synth_scene
from gsplat.rendering import rasterization
dev = torch.device("cuda:0")
N, C, W, H = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (1_000_000, 5, 1242, 375)
g = synth_scene.make_gaussians(N, sh_degree=3)
p = {k: v.to(dev) for k, v in g.items()}
vm, Ks = synth_scene.make_cameras(C, W, H)
with torch.no_grad():
    rc, ra, meta = rasterization(p["means"], p["quats"], p["scales"], p["opacities"], p["colors"], vm.to(dev), Ks.to(dev),
                                 W, H, near_plane=0.1, sh_degree=3, packed=False, render_mode="RGB+ED",
                                 absgrad=True, rasterize_mode="antialiased")
off = meta["isect_offsets"].reshape(-1).long()
I = meta["flatten_ids"].numel()
end = torch.cat([off[1:], torch.tensor([I], device=dev)])
L = (end - off)
last = meta["last_ids"]  # [C,H,W]
TH, TW = meta["tile_height"], meta["tile_width"]
lp = torch.zeros(C, TH * 16, TW * 16, dtype=torch.int64, device=dev)
lp[:, :H, :W] = last
tmax = lp.reshape(C, TH, 16, TW, 16).amax(dim=(2, 4)).reshape(-1)
walked = torch.where(L > 0, (tmax - off + 1).clamp(min=0), torch.zeros_like(L))
walked = torch.minimum(walked, L)
for name, v in (("list length", L), ("walked", walked)):
    v = v.float()
    qs = torch.quantile(v, torch.tensor([0.5, 0.9, 0.99, 0.999, 1.0], device=dev))
    print(f"{name}: sum {int(v.sum())} mean {v.mean():.1f} p50 {qs[0]:.0f} p90 {qs[1]:.0f} p99 {qs[2]:.0f} p99.9 {qs[3]:.0f} max {qs[4]:.0f}")
srt = torch.sort(walked, descending=True).values
print("top walked:", srt[:20].tolist())
print("tiles with walked > 1000:", int((walked > 1000).sum()), " > 2000:", int((walked > 2000).sum()))
print("tiles with list > 2048:", int((L > 2048).sum()), "> 6144:", int((L > 6144).sum()), "of", L.numel(), " walked/listed =", float(walked.sum()) / float(L.sum()))
