"""Eval re-render cache (gsplat.rendering.EVAL_CACHE): the reference renders one image up to six times under no_grad with
only the opacities masked per class (/root/reference/3D-Reconstruction-main/models/trainers/scene_graph.py:282-301).
A cached re-render must be bit-identical to an independent call, and anything else that changes must miss."""
import math

import pytest
import torch

import synth_scene

pytestmark = pytest.mark.gpu


def _scene(dev, n=60_000, dense=False):
    g = synth_scene.make_gaussians(n, sh_degree=None, seed=5, extent=6.0 if dense else 20.0)
    if dense:
        gen = torch.Generator().manual_seed(9)
        g["means"][:, 2] = torch.rand(n, generator=gen) * 3.0 + 0.2
        g["scales"] = g["scales"] * 1.5
    return {k: v.to(dev) for k, v in g.items()}


def _call(p, opac, vm, Ks, W, H, **kw):
    from gsplat.rendering import rasterization

    with torch.no_grad():
        return rasterization(p["means"], p["quats"], p["scales"], opac, p["colors"], vm, Ks, W, H, near_plane=0.1,
                             packed=False, render_mode="RGB+ED", absgrad=True, rasterize_mode="antialiased", **kw)


def _independent(p, opac, vm, Ks, W, H, **kw):
    import gsplat.rendering as R

    old = R.EVAL_CACHE
    R.EVAL_CACHE = False
    try:
        return _call(p, opac, vm, Ks, W, H, **kw)
    finally:
        R.EVAL_CACHE = old


def _same(a, b):
    (rc1, ra1, m1), (rc0, ra0, m0) = a, b
    assert torch.equal(rc1, rc0) and torch.equal(ra1, ra0)
    # (last_ids index the lists the rasterizer ran on: gsplat's full lists for a cached call, the culled ones for an
    # independent call -- not comparable across the two; the images above are what they produce)
    assert torch.equal(m1["radii"], m0["radii"])
    vis = m0["radii"] > 0
    assert torch.equal(m1["opacities"][vis], m0["opacities"][vis])
    assert torch.equal(m1["flatten_ids"], m0["flatten_ids"])
    assert m1["width"] == m0["width"] and m1["height"] == m0["height"]


@pytest.mark.parametrize("dense", [False, True])
def test_masked_rerenders_hit_and_match(gpu_device, dense):
    import gsplat.rendering as R

    dev = gpu_device
    W, H = 200, 120
    p = _scene(dev, dense=dense)
    N = p["means"].shape[0]
    c2w = torch.linalg.inv(synth_scene.make_cameras(1, W, H)[0][0]).to(dev)
    Ks = synth_scene.make_cameras(1, W, H)[1].to(dev)
    labels = torch.randint(0, 4, (N,), device=dev)
    R.clear_cache()
    # the reference's pattern: viewmats = inv(camtoworlds)[None] is a NEW tensor on every call
    full = _call(p, p["opacities"], torch.linalg.inv(c2w)[None], Ks, W, H)
    ent = R._cache.entry
    assert ent is not None
    _same(full, _independent(p, p["opacities"], torch.linalg.inv(c2w)[None], Ks, W, H))
    for cls in range(4):
        opac = p["opacities"] * (labels == cls)
        if dense and cls == 3:
            opac = opac * 0.01  # nearly transparent: pixels outlive the sorted front -> redo pass on cached lists
        got = _call(p, opac, torch.linalg.inv(c2w)[None], Ks, W, H)
        assert R._cache.entry is ent, "a call that differs only in opacities must hit the cache"
        _same(got, _independent(p, opac, torch.linalg.inv(c2w)[None], Ks, W, H))
    # backgrounds are a rasterizer input: still a hit
    bg = torch.rand(1, 3, device=dev)
    got = _call(p, p["opacities"], torch.linalg.inv(c2w)[None], Ks, W, H, backgrounds=bg)
    assert R._cache.entry is ent
    _same(got, _independent(p, p["opacities"], torch.linalg.inv(c2w)[None], Ks, W, H, backgrounds=bg))


def test_cache_misses_when_inputs_change(gpu_device):
    import gsplat.rendering as R

    dev = gpu_device
    W, H = 160, 96
    p = _scene(dev, n=20_000)
    vm, Ks = (t.to(dev) for t in synth_scene.make_cameras(2, W, H))
    R.clear_cache()
    _call(p, p["opacities"], vm[:1], Ks[:1], W, H)
    e0 = R._cache.entry
    # another camera
    got = _call(p, p["opacities"], vm[1:], Ks[1:], W, H)
    assert R._cache.entry is not e0
    _same(got, _independent(p, p["opacities"], vm[1:], Ks[1:], W, H))
    e1 = R._cache.entry
    # in-place update of a parameter (an optimizer step): same storage, new version
    p["means"].add_(0.05)
    got = _call(p, p["opacities"], vm[1:], Ks[1:], W, H)
    assert R._cache.entry is not e1
    _same(got, _independent(p, p["opacities"], vm[1:], Ks[1:], W, H))
    e2 = R._cache.entry
    # another near plane
    from gsplat.rendering import rasterization
    with torch.no_grad():
        rasterization(p["means"], p["quats"], p["scales"], p["opacities"], p["colors"], vm[1:], Ks[1:], W, H,
                      near_plane=0.5, packed=False, render_mode="RGB+ED", rasterize_mode="antialiased")
    assert R._cache.entry is not e2
    # a call that needs gradients neither uses nor keeps a cache entry
    q = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    rc, ra, meta = rasterization(q["means"], q["quats"], q["scales"], q["opacities"], q["colors"], vm[1:], Ks[1:], W, H,
                                 near_plane=0.1, packed=False, render_mode="RGB+ED", rasterize_mode="antialiased")
    assert R._cache.entry is None
    (rc.sum() + ra.sum()).backward()
    assert q["means"].grad is not None and torch.isfinite(q["means"].grad).all()


def test_data_surgery_needs_clear_cache_and_inplace_ops_miss(gpu_device):
    """The key tracks `tensor._version`: an in-place op on a Gaussian tensor misses by itself; `.data` surgery does
    not bump the version, so the documented contract is `clear_cache()` after it."""
    import synth_scene
    from gsplat import rendering as R

    dev = gpu_device
    N, C, W, H = 3000, 1, 120, 80
    g = synth_scene.make_gaussians(N, sh_degree=None, extent=8.0)
    gg = {k: v.to(dev) for k, v in g.items()}
    vm, Ks = synth_scene.make_cameras(C, W, H)
    vm, Ks = vm.to(dev), Ks.to(dev)

    def render():
        with torch.no_grad():
            return R.rasterization(gg["means"], gg["quats"], gg["scales"], gg["opacities"], gg["colors"], vm, Ks, W, H,
                                   packed=False, render_mode="RGB+ED")

    R.clear_cache()
    a = render()
    e0 = R._cache.entry
    gg["means"].add_(0.05)  # in-place: version bump -> miss
    b = render()
    assert R._cache.entry is not e0
    assert not torch.equal(a[0], b[0])
    gg["means"].data.add_(0.05)  # .data surgery: the caller clears the cache
    R.clear_cache()
    c = render()
    with R.no_eval_cache():
        ref = render()
    assert torch.equal(c[0], ref[0]) and torch.equal(c[1], ref[1])
    R.clear_cache()


def test_forward_dropped_without_backward_leaves_other_tensors_alone(gpu_device):
    """The gradient record and the backward tile order are written on the side stream.  If the autograd graph is freed
    without a backward, the allocator must not hand their blocks to a main-stream tensor while those writes are still
    pending (record_stream in _RasterizeSplats.forward)."""
    import synth_scene
    from gsplat import rendering as R

    dev = gpu_device
    N, C, W, H = 200_000, 2, 400, 300
    g = synth_scene.make_gaussians(N, sh_degree=None, extent=20.0)
    gg = {k: v.to(dev).requires_grad_(True) for k, v in g.items()}
    vm, Ks = synth_scene.make_cameras(C, W, H)
    vm, Ks = vm.to(dev), Ks.to(dev)
    for _ in range(5):
        rc, ra, meta = R.rasterization(gg["means"], gg["quats"], gg["scales"], gg["opacities"], gg["colors"], vm, Ks,
                                       W, H, packed=False, render_mode="RGB+ED", absgrad=True)
        del rc, ra, meta  # no backward: the graph (and the zero-filled gradient record) is dropped here
        ones = torch.ones(C, N, 16, device=dev)  # the record's size: the block the allocator would reuse
        small = torch.ones(C * math.ceil(W / 16) * math.ceil(H / 16), device=dev, dtype=torch.int32)
        torch.cuda.synchronize()
        assert bool((ones == 1).all()) and bool((small == 1).all())
        del ones, small
