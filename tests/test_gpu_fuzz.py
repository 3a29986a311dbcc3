"""A bounded, seeded subset of the randomised end-to-end sweep (scratch/fuzz.py is the open-ended version): 20 random
configurations of `rasterization()` -- N 1..12000, 1-3 cameras, every render mode, SH degree None/0..4, packed / dense,
backgrounds, radius clip, off-centre principal points, zero opacities, dense and faint scenes -- plus 4 forced
expected-depth cases on 1- and 7-Gaussian scenes, forward and every parameter gradient against the fp64 oracle.

Bars.  Images: at most 5e-3 of the elements outside 1e-5 + 1e-4 |ref| (decisions that flip at an ulp).  Radii /
intersection counts: at most 5e-3 mismatching.  Gradients: rel-L2 <= 2e-3 (the sweep's flag level; measured <= 5e-4),
EXCEPT the expected-depth modes ("ED", "RGB+ED") on scenes of 1 or 7 Gaussians: rel-L2 <= 3e-2 (measured up to 9e-3).
That is fp32 cancellation in gsplat's own formulation, not of this build: on a pixel covered by one faint Gaussian the
two alpha-gradient terms of D / alpha -- z v / alpha through the depth channel and -v (D / alpha) / alpha through the
alpha -- are each ~250x the gradient they leave; autograd through `render_colors[..., -1:] / alphas.clamp(1e-10)` does the
same subtraction in fp32.  The same scenes in "RGB" / "RGB+D" meet the general bar (asserted by the forced cases below).
PARITY UNPINNED against gsplat itself (oracle/__init__.py)."""
import random

import pytest
import torch

import oracle
import synth_scene

pytestmark = pytest.mark.gpu


def _recipe(case):
    rnd = random.Random(case)
    cfg = dict(
        N=rnd.choice([1, 7, 500, 3000, 12000]), C=rnd.choice([1, 1, 2, 3]), W=rnd.choice([17, 64, 130, 201]),
        H=rnd.choice([16, 33, 75, 120]), sh=rnd.choice([None, 0, 1, 2, 3, 4]),
        mode=rnd.choice(["RGB", "RGB+D", "RGB+ED", "D", "ED"]), rmode=rnd.choice(["classic", "antialiased"]),
        packed=rnd.choice([False, False, True]), bgs=rnd.choice([False, True]), rclip=rnd.choice([0.0, 0.0, 2.5]),
        extent=rnd.choice([3.0, 10.0, 25.0]), dense=rnd.random() < 0.3, faint=rnd.choice([1.0, 0.05]),
        zero_op=rnd.random() < 0.3, offcentre=rnd.random() < 0.3)
    return cfg


FORCED = {  # the cancellation cases and their controls
    "one_ED": dict(N=1, C=1, W=64, H=33, sh=None, mode="ED", rmode="classic", extent=3.0),
    "one_RGB+D": dict(N=1, C=1, W=64, H=33, sh=None, mode="RGB+D", rmode="classic", extent=3.0),
    "seven_RGB+ED": dict(N=7, C=2, W=130, H=75, sh=2, mode="RGB+ED", rmode="antialiased", extent=3.0),
    "seven_RGB": dict(N=7, C=2, W=130, H=75, sh=2, mode="RGB", rmode="antialiased", extent=3.0),
}


def _scene_seed(cfg, start, need=1):
    """Scenes of 1 or 7 Gaussians: the first seed from `start` on that puts at least `need` of them on screen."""
    if cfg["N"] > 7:
        return start
    for seed in range(start, start + 200):
        g = synth_scene.make_gaussians(cfg["N"], sh_degree=cfg["sh"], seed=seed, extent=cfg["extent"])
        if cfg.get("dense"):
            g["scales"] = g["scales"] * 3.0
        vm, Ks = synth_scene.make_cameras(cfg["C"], cfg["W"], cfg["H"], frame=seed % 5)
        r = oracle.gsplat_ref.fully_fused_projection(g["means"].double(), g["quats"].double(), g["scales"].double(),
                                                     vm.double(), Ks.double(), cfg["W"], cfg["H"], 0.3, 0.1, 1e10,
                                                     cfg.get("rclip", 0.0))[0]
        if int((r > 0).any(0).sum()) >= need:
            return seed
    pytest.fail("no seed puts the tiny scene on screen")


def _run(dev, cfg, seed):
    from gsplat.rendering import rasterization

    N, C, W, H, sh, mode = cfg["N"], cfg["C"], cfg["W"], cfg["H"], cfg["sh"], cfg["mode"]
    bgs = cfg.get("bgs", False) and mode not in ("D", "ED")
    g = synth_scene.make_gaussians(N, sh_degree=sh, seed=seed, extent=cfg["extent"])
    if cfg.get("dense"):
        g["scales"] = g["scales"] * 3.0
        g["opacities"] = g["opacities"] * cfg["faint"]
    if cfg.get("zero_op"):
        g["opacities"][: N // 5] = 0.0
    vm, Ks = synth_scene.make_cameras(C, W, H, frame=seed % 5)
    if cfg.get("offcentre"):
        Ks[:, 0, 2] += 3.3
        Ks[:, 1, 2] -= 2.1
    gen = torch.Generator().manual_seed(seed)
    bg = torch.rand(C, 3, generator=gen) if bgs else None
    kw = dict(near_plane=0.1, sh_degree=sh, render_mode=mode, rasterize_mode=cfg["rmode"],
              radius_clip=cfg.get("rclip", 0.0))
    packed = cfg.get("packed", False)
    p = {k: v.to(dev).requires_grad_(True) for k, v in g.items()}
    rc, ra, meta = rasterization(p["means"], p["quats"], p["scales"], p["opacities"], p["colors"], vm.to(dev),
                                 Ks.to(dev), W, H, packed=packed, absgrad=True,
                                 backgrounds=bg.to(dev) if bgs else None, **kw)
    wc, wa = torch.randn(rc.shape, generator=gen), torch.randn(ra.shape, generator=gen)
    ((rc * wc.to(dev)).sum() + (ra * wa.to(dev)).sum()).backward()
    q = {k: v.double().requires_grad_(True) for k, v in g.items()}
    rco, rao, mo = oracle.rasterization(q["means"], q["quats"], q["scales"], q["opacities"], q["colors"], vm.double(),
                                        Ks.double(), W, H, backgrounds=bg.double() if bgs else None, **kw)
    lo = (rco * wc.double()).sum() + (rao * wa.double()).sum()
    if lo.requires_grad:
        lo.backward()

    def bad(a, b):
        a, b = a.detach().double().cpu(), b.detach().double()
        return float(((a - b).abs() > 1e-5 + 1e-4 * b.abs()).double().mean()) if b.numel() else 0.0

    stats = {"rc": bad(rc, rco), "ra": bad(ra, rao)}
    n_visible = int((mo["radii"] > 0).sum())
    for k in q:
        assert p[k].grad is None or bool(torch.isfinite(p[k].grad).all()), k
        if q[k].grad is None:
            continue
        ref, got = q[k].grad, p[k].grad.double().cpu()
        stats["g_" + k] = float((got - ref).norm() / ref.norm()) if float(ref.norm()) > 1e-9 else float((got - ref).norm())
    if not packed:
        stats["radii_mismatch"] = float((meta["radii"].cpu() != mo["radii"]).double().mean())
        stats["isect_n"] = abs(meta["flatten_ids"].numel() - mo["flatten_ids"].numel()) / max(1, mo["flatten_ids"].numel())
    return stats, n_visible


def _check(name, cfg, stats, n_visible):
    ed_tiny = cfg["mode"] in ("ED", "RGB+ED") and cfg["N"] <= 7
    grad_bar = 3e-2 if ed_tiny else 2e-3
    print(f"[fuzz] {name}: {cfg} visible={n_visible} " + " ".join(f"{k}={v:.1e}" for k, v in stats.items()))
    fails = {k: v for k, v in stats.items() if v > (grad_bar if k.startswith("g_") else 5e-3)}
    assert not fails, (name, cfg, fails)


@pytest.mark.parametrize("case", list(range(20)))
def test_random_configuration_against_oracle(gpu_device, case):
    cfg = _recipe(case)
    stats, n_visible = _run(gpu_device, cfg, _scene_seed(cfg, case))
    assert n_visible >= 1
    _check(f"case{case}", cfg, stats, n_visible)


@pytest.mark.parametrize("name", list(FORCED))
def test_expected_depth_cancellation_cases(gpu_device, name):
    cfg = FORCED[name]
    stats, n_visible = _run(gpu_device, cfg, _scene_seed(cfg, 0, max(1, cfg["N"] // 2)))
    assert n_visible >= 1
    _check(name, cfg, stats, n_visible)
