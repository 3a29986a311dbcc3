"""A/B helper: bit hashes of the forward outputs and fp64 sums of the gradients on the bench scene (and a config-3-like
one), to compare two builds of the library."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "3d-reconstruction_amd")):
    sys.path.insert(0, p)
import torch
import synth_scene
from gsplat.rendering import rasterization

dev = torch.device("cuda:0")


def bits(t):
    v = t.detach().contiguous().view(torch.int32).to(torch.int64).reshape(-1)
    w = torch.arange(v.numel(), device=v.device, dtype=torch.int64) % 1000003 + 1
    return int((v * w).sum().item()) & 0xFFFFFFFFFFFF


def run(N, C, W, H, mode, sh):
    g = synth_scene.make_gaussians(N, sh_degree=3 if sh else None)
    p = {k: v.to(dev).requires_grad_(True) for k, v in g.items()}
    vm, Ks = synth_scene.make_cameras(C, W, H)
    rc, ra, meta = rasterization(p["means"], p["quats"], p["scales"], p["opacities"], p["colors"], vm.to(dev), Ks.to(dev),
                                 W, H, near_plane=0.1, sh_degree=3 if sh else None, packed=False, render_mode=mode,
                                 absgrad=True, rasterize_mode="antialiased")
    gen = torch.Generator().manual_seed(7)
    wc = torch.randn(rc.shape, generator=gen).to(dev); wa = torch.randn(ra.shape, generator=gen).to(dev)
    ((rc * wc).sum() + (ra * wa).sum()).backward()
    out = {"rc": bits(rc), "ra": bits(ra), "last": bits(meta["last_ids"]), "I": meta["flatten_ids"].numel()}
    for k, v in p.items():
        out["g_" + k] = [float(v.grad.double().sum()), float(v.grad.double().abs().sum())]
    out["absgrad"] = float(meta["means2d"].absgrad.double().sum())
    return out


res = {"cfg2": run(1_000_000, 5, 1242, 375, "RGB+ED", True), "rgb_1cam": run(300_000, 1, 1920, 1280, "RGB", False)}
print(json.dumps(res, indent=1))
