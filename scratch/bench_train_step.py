"""One training step around the rasterizer at the BASELINE configs[2] shape (3 M Gaussians, one 1920x1280 camera):
  fused:  collect_gaussians -> rasterization -> l1_ssim -> backward -> accumulate_stats
  torch:  the reference's torch-op chains for the same stages (vanilla.py:359-393, base.py:533-541, base.py:279-297 +
          vanilla.py:154-182) around the SAME rasterization call
so the difference is what the widening rows buy per step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-reconstruction_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "scratch")]
import torch
import synth_scene
from gsplat.cuda._wrapper import spherical_harmonics
from gsplat.densify import DensifyState, accumulate_stats
from gsplat.losses import l1_ssim
from gsplat.prepass import GaussianClass, collect_gaussians
from gsplat.rendering import rasterization
from bench_loss import torch_chain
from test_gpu_densify import _RefStats

dev = torch.device("cuda:0")
N, W, H, K = 3_000_000, 1920, 1280, 16
g = synth_scene.make_gaussians(N, sh_degree=3, seed=42)
raw = dict(means=g["means"], features_dc=g["colors"][:, 0, :].contiguous(), features_rest=g["colors"][:, 1:, :].contiguous(),
           opacities=torch.logit(g["opacities"].clamp(1e-4, 1 - 1e-4))[:, None], scales=torch.log(g["scales"]), quats=g["quats"])
raw = {k: v.to(dev).requires_grad_(True) for k, v in raw.items()}
vm, Ks = (t.to(dev) for t in synth_scene.make_cameras(1, W, H))
c2w = torch.linalg.inv(vm[0])
gt = torch.rand(H, W, 3, device=dev)
mask = torch.ones(H, W, device=dev)
KW = dict(viewmats=vm, Ks=Ks, width=W, height=H, packed=False, absgrad=True, sparse_grad=False, rasterize_mode="classic",
          near_plane=0.1, far_plane=1e10, render_mode="RGB+ED")


def zero():
    for v in raw.values():
        v.grad = None


state = DensifyState()
ref_stats = _RefStats(N)


def fused():
    zero()
    gs, _, flag = collect_gaussians([GaussianClass(**raw, sh_degree=3, sh_degree_to_use=3)], c2w)
    rc, ra, info = rasterization(means=gs["_means"], quats=gs["_quats"], scales=gs["_scales"],
                                 opacities=gs["_opacities"].squeeze(), colors=gs["_rgbs"], **KW)
    info["means2d"].retain_grad()
    l1, ss = l1_ssim(torch.clamp(rc[0, ..., :3], max=1.0), gt, mask)
    (0.8 * l1 + 0.2 * (1 - ss)).backward()
    accumulate_stats(info, state, absgrad=True)


def torch_way():
    zero()
    filter_mask = torch.ones_like(raw["means"][:, 0], dtype=torch.bool)
    colors = torch.cat((raw["features_dc"][:, None, :], raw["features_rest"]), dim=1)
    viewdirs = raw["means"].detach() - c2w.data[..., :3, 3]
    viewdirs = viewdirs / viewdirs.norm(dim=-1, keepdim=True)
    rgbs = torch.clamp(spherical_harmonics(3, viewdirs, colors) + 0.5, 0.0, 1.0)
    gs = dict(_means=raw["means"][filter_mask], _opacities=torch.sigmoid(raw["opacities"])[filter_mask],
              _rgbs=rgbs[filter_mask], _scales=torch.exp(raw["scales"])[filter_mask],
              _quats=(raw["quats"] / raw["quats"].norm(dim=-1, keepdim=True))[filter_mask])
    for k, v in gs.items():
        if torch.isnan(v).any():
            raise ValueError(k)
        if torch.isinf(v).any():
            raise ValueError(k)
    gs = {k: torch.cat([v], dim=0) for k, v in gs.items()}
    rc, ra, info = rasterization(means=gs["_means"], quats=gs["_quats"], scales=gs["_scales"],
                                 opacities=gs["_opacities"].squeeze(), colors=gs["_rgbs"], **KW)
    info["means2d"].retain_grad()
    l1, ss = torch_chain(torch.clamp(rc[0, ..., :3], max=1.0), gt, mask)
    (0.8 * l1 + 0.2 * (1 - ss)).backward()
    grads = info["means2d"].absgrad.clone()
    grads[..., 0] *= info["width"] / 2.0
    grads[..., 1] *= info["height"] / 2.0
    ref_stats.after_train(info["radii"][0], grads[0], max(info["width"], info["height"]))


def t(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


print(f"training step, 3 M Gaussians, 1920x1280: torch chains around the rasterizer {t(torch_way):.2f} ms   fused {t(fused):.2f} ms")
