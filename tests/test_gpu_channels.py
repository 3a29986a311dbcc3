"""More than 4 render channels (gsplat's channel_chunk loop): one projection, one
rasterizer pass per group of 4 channels.  Checked against the oracle and against per-group independent calls."""
import pytest
import torch

import oracle
import synth_scene
from util import assert_close

pytestmark = pytest.mark.gpu


def _setup(dev, N, C, W, H, D, seed=3):
    g = synth_scene.make_gaussians(N, sh_degree=None, seed=seed, extent=8.0)
    gen = torch.Generator().manual_seed(seed)
    g["colors"] = torch.rand(N, D, generator=gen)
    vm, Ks = synth_scene.make_cameras(C, W, H)
    return g, vm, Ks


def test_seven_channels_rgb_ed_against_oracle(gpu_device):
    from gsplat.rendering import rasterization

    dev = gpu_device
    N, C, W, H, D = 3000, 2, 96, 64, 6
    g, vm, Ks = _setup(dev, N, C, W, H, D)
    bg = torch.rand(C, D)
    gen = torch.Generator().manual_seed(1)
    wc, wa = torch.randn(C, H, W, D + 1, generator=gen), torch.randn(C, H, W, 1, generator=gen)
    p = {k: v.to(dev).requires_grad_(True) for k, v in g.items()}
    rc, ra, meta = rasterization(p["means"], p["quats"], p["scales"], p["opacities"], p["colors"], vm.to(dev), Ks.to(dev),
                                 W, H, near_plane=0.1, packed=False, render_mode="RGB+ED", absgrad=True,
                                 rasterize_mode="antialiased", backgrounds=bg.to(dev))
    assert rc.shape == (C, H, W, D + 1) and ra.shape == (C, H, W, 1)
    meta["means2d"].retain_grad()
    ((rc * wc.to(dev)).sum() + (ra * wa.to(dev)).sum()).backward()
    q = {k: v.double().requires_grad_(True) for k, v in g.items()}
    rco, rao, mo = oracle.rasterization(q["means"], q["quats"], q["scales"], q["opacities"], q["colors"], vm.double(),
                                        Ks.double(), W, H, near_plane=0.1, render_mode="RGB+ED",
                                        rasterize_mode="antialiased", backgrounds=bg.double())
    mo["means2d"].retain_grad()
    ((rco * wc.double()).sum() + (rao * wa.double()).sum()).backward()
    assert torch.equal(meta["radii"].cpu(), mo["radii"])
    assert_close("rc", rc, rco, max_bad_frac=2e-3)
    assert_close("ra", ra, rao, max_bad_frac=2e-3)
    for k in ("means", "quats", "scales", "opacities", "colors"):
        assert_close("v_" + k, p[k].grad, q[k].grad, max_bad_frac=5e-3, max_rel_l2=1e-4, scale_atol=True)
    assert_close("v_means2d", meta["means2d"].grad, mo["means2d"].grad, max_bad_frac=5e-3, max_rel_l2=1e-4,
                 scale_atol=True)
    assert meta["means2d"].absgrad.shape == (C, N, 2)


def test_nine_channels_equal_independent_groups(gpu_device):
    """RGB mode, 9 feature channels = groups 4 + 4 + 1: images equal three independent <=4-channel calls bit for bit,
    abs-gradients add up over the groups (the alpha gradient belongs to the first group)."""
    from gsplat.rendering import rasterization

    dev = gpu_device
    N, C, W, H, D = 4000, 1, 80, 48, 9
    g, vm, Ks = _setup(dev, N, C, W, H, D, seed=5)
    vm, Ks = vm.to(dev), Ks.to(dev)
    gen = torch.Generator().manual_seed(2)
    wc, wa = torch.randn(C, H, W, D, generator=gen).to(dev), torch.randn(C, H, W, 1, generator=gen).to(dev)

    def run(cols, wcs, use_wa):
        p = {k: v.to(dev).requires_grad_(True) for k, v in g.items() if k != "colors"}
        col = cols.clone().requires_grad_(True)
        rc, ra, meta = rasterization(p["means"], p["quats"], p["scales"], p["opacities"], col, vm, Ks, W, H,
                                     near_plane=0.1, packed=False, render_mode="RGB", absgrad=True)
        loss = (rc * wcs).sum() + ((ra * wa).sum() if use_wa else 0.0)
        loss.backward()
        return rc, ra, meta["means2d"].absgrad.clone(), p["means"].grad.clone(), col.grad.clone()

    cols = g["colors"].to(dev)
    rc, ra, ag, gm, gc = run(cols, wc, True)
    parts = [run(cols[:, a:b].contiguous(), wc[..., a:b], a == 0) for a, b in ((0, 4), (4, 8), (8, 9))]
    assert torch.equal(rc, torch.cat([q[0] for q in parts], -1)) and torch.equal(ra, parts[0][1])
    assert_close("absgrad", ag, sum(q[2] for q in parts), max_bad_frac=1e-3, max_rel_l2=1e-5, scale_atol=True)
    assert_close("v_means", gm, sum(q[3] for q in parts), max_bad_frac=1e-3, max_rel_l2=1e-5, scale_atol=True)
    assert_close("v_colors", gc, torch.cat([q[4] for q in parts], -1), max_bad_frac=1e-3, max_rel_l2=1e-5, scale_atol=True)
