"""Full-size GPU checks at BASELINE.json's configurations (configs[1], [2], [4] shapes), where the fp64 oracle cannot
run the rasterizer end to end in seconds.  What is compared instead:

  * integer stages: bit-exact against the oracle's numpy restatement of A.3 / A.4 (it handles 10^7 intersections in
    seconds) fed the GPU's own fp32 means2d / radii / depths -- tile counts, sorted isect_ids, flatten_ids, offsets;
  * rasterize forward: a few hundred sampled pixels recomputed in fp64 from the GPU's own sorted lists by a vectorised
    restatement of A.5 (cumprod form), 1e-5 abs + 1e-4 rel, with a stated flip budget for the alpha < 1/255 and
    T <= 1e-4 decisions;
  * size-independent properties of the path: linearity of the render in the colours, camera-shard additivity of the
    gradients (rasterization(C > 1) == per-camera calls, gradients summed: the defined semantics of batching and what
    the data-parallel split relies on), run-to-run identity of every integer output and of the forward images,
    packed == unpacked renders, exact zeros for zero-opacity Gaussians.
"""
import math

import pytest
import torch

import oracle
import synth_scene
from util import assert_close, assert_rerun_close, close_report

pytestmark = pytest.mark.gpu

ALPHA_MIN, ALPHA_MAX, T_MIN = 1.0 / 255.0, 0.999, 1e-4


def _check_integer_stages(meta, C):
    TW, TH = meta["tile_width"], meta["tile_height"]
    tpg, ids, fids = oracle.isect_tiles(meta["means2d"].detach().cpu(), meta["radii"].cpu(), meta["depths"].cpu(), 16,
                                        TW, TH)
    assert torch.equal(meta["tiles_per_gauss"].cpu(), tpg), "tiles_per_gauss differs"
    assert meta["isect_ids"].numel() == ids.numel()
    assert torch.equal(meta["isect_ids"].cpu(), ids), "sorted intersection keys differ"
    assert torch.equal(meta["flatten_ids"].cpu(), fids), "sort order (flatten ids) differs"
    assert torch.equal(meta["isect_offsets"].cpu(), oracle.isect_offset_encode(ids, C, TW, TH)), "offsets differ"
    return ids.numel()


def _sampled_pixels_fp64(meta, colors_cn, rc, ra, n_samples, ed, seed=0, backgrounds=None):
    """Recomputes `n_samples` random pixels in fp64 from the GPU's sorted lists (Appendix A.5) and returns the number
    whose colours / alpha fall outside 1e-5 + 1e-4 |ref|.  colors_cn: [C*N, D] per-row colours incl. the depth channel."""
    dev = rc.device
    C, H, W, D = rc.shape
    TW, TH = meta["tile_width"], meta["tile_height"]
    g = torch.Generator().manual_seed(seed)
    cs = torch.randint(0, C, (n_samples,), generator=g).tolist()
    ys = torch.randint(0, H, (n_samples,), generator=g).tolist()
    xs = torch.randint(0, W, (n_samples,), generator=g).tolist()
    offs = meta["isect_offsets"].reshape(-1).cpu()
    I = meta["flatten_ids"].numel()
    m2 = meta["means2d"].detach().reshape(-1, 2)
    con = meta["conics"].reshape(-1, 3)
    opa = meta["opacities"].reshape(-1)
    bad = 0
    worst = 0.0
    for c, y, x in zip(cs, ys, xs):
        t = (c * TH + y // 16) * TW + x // 16
        s, e = int(offs[t]), int(offs[t + 1]) if t + 1 < offs.numel() else I
        out = torch.zeros(D, dtype=torch.float64, device=dev)
        T = torch.ones((), dtype=torch.float64, device=dev)
        if e > s:
            ids = meta["flatten_ids"][s:e].long()
            d = m2[ids].double() - torch.tensor([x + 0.5, y + 0.5], dtype=torch.float64, device=dev)
            k = con[ids].double()
            sigma = 0.5 * (k[:, 0] * d[:, 0] ** 2 + k[:, 2] * d[:, 1] ** 2) + k[:, 1] * d[:, 0] * d[:, 1]
            alpha = torch.clamp(opa[ids].double() * torch.exp(-sigma), max=ALPHA_MAX)
            a = torch.where((sigma >= 0) & (alpha >= ALPHA_MIN), alpha, torch.zeros_like(alpha))
            Tn = torch.cumprod(1.0 - a, 0)
            stop = (Tn <= T_MIN) & (a > 0)
            live = torch.cumsum(stop.to(torch.int32), 0) == 0  # entries strictly before the first stop
            a = a * live
            Tn = torch.cumprod(1.0 - a, 0)
            Tb = Tn / (1.0 - a)
            out = (colors_cn[ids].double() * (a * Tb)[:, None]).sum(0)
            T = Tn[-1]
        al = 1.0 - T
        if backgrounds is not None:
            out = out + T * backgrounds[c].double()
        if ed:
            out[-1] = out[-1] / torch.clamp(al, min=1e-10)
        got = torch.cat([rc[c, y, x].double(), ra[c, y, x].double()])
        ref = torch.cat([out, al.reshape(1)])
        err = (got - ref).abs()
        worst = max(worst, float(err.max()))
        if bool((err > 1e-5 + 1e-4 * ref.abs()).any()):
            bad += 1
    print(f"[parity] sampled pixels: {n_samples} checked, {bad} outside tolerance, worst abs err {worst:.3e}")
    return bad


def _sh_colors(means, sh, viewmats, degree):
    from gsplat.cuda._wrapper import _camera_centers, spherical_harmonics

    cp = _camera_centers(viewmats)
    dirs = means[None] - cp[:, None, :]
    C = viewmats.shape[0]
    return torch.clamp_min(spherical_harmonics(degree, dirs, sh[None].expand(C, -1, -1, -1)) + 0.5, 0.0)


# ------------------------------------------------------------------------------------------------- configs[1]
def test_config2_full_size(gpu_device):
    """1M Gaussians, 5 x 1242x375, SH3, RGB+ED, absgrad + antialiased (bench.py's workload)."""
    from gsplat.rendering import rasterization

    dev = gpu_device
    N, C, W, H = 1_000_000, 5, 1242, 375
    g = synth_scene.make_gaussians(N, sh_degree=3, seed=42)
    vm, Ks = synth_scene.make_cameras(C, W, H)
    p = {k: v.to(dev).requires_grad_(True) for k, v in g.items()}
    vm, Ks = vm.to(dev), Ks.to(dev)
    kw = dict(near_plane=0.1, far_plane=1e10, sh_degree=3, packed=False, render_mode="RGB+ED", absgrad=True,
              rasterize_mode="antialiased")
    rc, ra, meta = rasterization(p["means"], p["quats"], p["scales"], p["opacities"], p["colors"], vm, Ks, W, H, **kw)
    n_isects = _check_integer_stages(meta, C)
    assert n_isects > 5_000_000
    assert bool(torch.isfinite(rc).all()) and float(ra.min()) >= 0.0 and float(ra.max()) <= 1.0

    with torch.no_grad():
        cols = torch.cat([_sh_colors(p["means"], p["colors"], vm, 3), meta["depths"][..., None]], -1).reshape(-1, 4)
    bad = _sampled_pixels_fp64(meta, cols, rc.detach(), ra.detach(), 200, ed=True)
    assert bad <= 2, "more than 1 % of the sampled pixels differ from the fp64 restatement"  # flip budget

    gen = torch.Generator().manual_seed(1)
    wc = torch.randn(C, H, W, 4, generator=gen).to(dev)
    wa = torch.randn(C, H, W, 1, generator=gen).to(dev)
    ((rc * wc).sum() + (ra * wa).sum()).backward()
    grads = {k: v.grad.clone() for k, v in p.items()}
    absg = meta["means2d"].absgrad.clone()
    for k, v in grads.items():
        assert bool(torch.isfinite(v).all()), k
    # absgrad >= |grad| elementwise (sum of absolute values vs absolute value of the sum)
    assert float((absg < 0).sum()) == 0

    # run-to-run: integers and forward images bit-identical; gradients (float atomics) within tolerance
    for v in p.values():
        v.grad = None
    rc2, ra2, meta2 = rasterization(p["means"], p["quats"], p["scales"], p["opacities"], p["colors"], vm, Ks, W, H, **kw)
    for key in ("radii", "tiles_per_gauss", "isect_ids", "flatten_ids", "isect_offsets", "last_ids"):
        assert torch.equal(meta[key], meta2[key]), key
    assert torch.equal(rc, rc2) and torch.equal(ra, ra2)
    ((rc2 * wc).sum() + (ra2 * wa).sum()).backward()
    for k, v in p.items():
        assert_rerun_close(f"cfg2.rerun.v_{k}", v.grad, grads[k])

    # camera-shard additivity: 5 single-camera calls, gradients summed == the batched call
    for v in p.values():
        v.grad = None
    absg_c = []
    for c in range(C):
        rcc, rac, mc = rasterization(p["means"], p["quats"], p["scales"], p["opacities"], p["colors"], vm[c:c + 1],
                                     Ks[c:c + 1], W, H, **kw)
        assert torch.equal(rcc[0], rc[c]) and torch.equal(rac[0], ra[c]), "batched and per-camera renders differ"
        assert torch.equal(mc["radii"][0], meta["radii"][c])
        ((rcc * wc[c:c + 1]).sum() + (rac * wa[c:c + 1]).sum()).backward()
        absg_c.append(mc["means2d"].absgrad[0])
    for k, v in p.items():
        assert_rerun_close(f"cfg2.shard.v_{k}", v.grad, grads[k])
    assert_rerun_close("cfg2.shard.absgrad", torch.stack(absg_c), absg)


# ------------------------------------------------------------------------------------------------- configs[2]
def test_config3_composite_scene(gpu_device):
    """~3M Gaussians as four concatenated groups (background 2.6M + rigid 0.25M + deformable 0.1M + 7 x 6890 SMPL), ~5 %
    of the non-background opacities exactly zero (frame-invalid instances, rigid.py:493), one 1920x1280 camera, colours
    pre-evaluated by the standalone spherical_harmonics op (the reference's actual calling pattern, vanilla.py:369-370)."""
    from gsplat.cuda._wrapper import _camera_centers, spherical_harmonics
    from gsplat.rendering import rasterization

    dev = gpu_device
    W, H = 1920, 1280
    sizes = [2_600_000, 250_000, 100_000, 7 * 6890]
    N = sum(sizes)
    g = synth_scene.make_gaussians(N, sh_degree=3, seed=7)
    vm, Ks = synth_scene.make_cameras(1, W, H)
    p = {k: v.to(dev) for k, v in g.items()}
    vm, Ks = vm.to(dev), Ks.to(dev)
    gen = torch.Generator().manual_seed(3)
    zero = torch.rand(N, generator=gen) < 0.05
    zero[: sizes[0]] = False
    p["opacities"][zero.to(dev)] = 0.0
    for v in p.values():
        v.requires_grad_(True)
    dirs = (p["means"] - _camera_centers(vm)[0]).detach()
    rgb = torch.clamp(spherical_harmonics(3, dirs, p["colors"]) + 0.5, 0.0, 1.0)
    rc, ra, meta = rasterization(p["means"], p["quats"], p["scales"], p["opacities"], rgb, vm, Ks, W, H,
                                 near_plane=0.1, far_plane=1e10, packed=False, render_mode="RGB+ED", absgrad=True,
                                 rasterize_mode="classic")
    assert meta["tile_width"] * meta["tile_height"] == 80 * 120
    _check_integer_stages(meta, 1)
    cols = torch.cat([rgb.detach()[None], meta["depths"][..., None]], -1).reshape(-1, 4)
    bad = _sampled_pixels_fp64(meta, cols, rc.detach(), ra.detach(), 150, ed=True, seed=1)
    assert bad <= 2

    gen = torch.Generator().manual_seed(4)
    wc = torch.randn(1, H, W, 4, generator=gen).to(dev)
    wa = torch.randn(1, H, W, 1, generator=gen).to(dev)
    meta["means2d"].retain_grad()
    ((rc * wc).sum() + (ra * wa).sum()).backward()
    for k, v in p.items():
        assert bool(torch.isfinite(v.grad).all()), k
    zd = zero.to(dev)
    vis = meta["radii"][0] > 0
    # a zero-opacity Gaussian contributes nothing: colour / SH gradients exactly zero, never NaN (hard parts)
    assert float(p["colors"].grad[zd].abs().max()) == 0.0
    assert float(meta["means2d"].grad[0][zd].abs().max()) == 0.0
    # culled rows get exactly zero gradients
    assert float(p["scales"].grad[~vis].abs().max()) == 0.0
    assert float(p["opacities"].grad[~vis].abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------- configs[4]
def test_config5_packed_render_only(gpu_device):
    """6M Gaussians, 1920x1080 novel views, packed=True, no grad (two views of the 600-view trajectory)."""
    from gsplat.rendering import rasterization

    dev = gpu_device
    N, C, W, H = 6_000_000, 2, 1920, 1080
    g = synth_scene.make_gaussians(N, sh_degree=3, seed=11)
    vm, Ks = synth_scene.make_cameras(C, W, H, frame=3)
    p = {k: v.to(dev) for k, v in g.items()}
    vm, Ks = vm.to(dev), Ks.to(dev)
    with torch.no_grad():
        args = (p["means"], p["quats"], p["scales"], p["opacities"], p["colors"], vm, Ks, W, H)
        kw = dict(near_plane=0.1, sh_degree=3, render_mode="RGB", rasterize_mode="classic")
        rc, ra, meta = rasterization(*args, packed=True, **kw)
        rc_u, ra_u, meta_u = rasterization(*args, packed=False, **kw)
    assert torch.equal(rc, rc_u) and torch.equal(ra, ra_u), "packed and unpacked renders differ"
    vis = meta_u["radii"] > 0
    nnz = int(vis.sum())
    assert meta["radii"].shape == (nnz,) and meta["means2d"].shape == (nnz, 2) and meta["conics"].shape == (nnz, 3)
    cam, gid = meta["camera_ids"], meta["gaussian_ids"]
    flat = cam * N + gid
    assert bool((flat[1:] > flat[:-1]).all()), "packed rows must be in ascending (camera, gaussian) order"
    assert torch.equal(flat, vis.reshape(-1).nonzero()[:, 0])
    assert torch.equal(meta["radii"], meta_u["radii"][vis])
    assert torch.equal(meta["means2d"], meta_u["means2d"][vis])
    assert torch.equal(meta["depths"], meta_u["depths"][vis])
    assert torch.equal(meta["opacities"], meta_u["opacities"][vis])
    assert torch.equal(meta["tiles_per_gauss"], meta_u["tiles_per_gauss"][vis])
    # A.7: packed flatten_ids index the packed rows
    assert meta["flatten_ids"].dtype == torch.int32
    assert torch.equal(flat[meta["flatten_ids"].long()], meta_u["flatten_ids"].long())
    assert torch.equal(meta["isect_ids"], meta_u["isect_ids"])
    _check_integer_stages(meta_u, C)
    cols = _sh_colors(p["means"], p["colors"], vm, 3).reshape(-1, 3)
    bad = _sampled_pixels_fp64(meta_u, cols, rc, ra, 150, ed=False, seed=2)
    assert bad <= 2


# ------------------------------------------------------------------------------------------------- linearity
def test_render_is_linear_in_colours(gpu_device):
    """With the geometry fixed, rendered colours are a linear functional of the per-Gaussian colours (A.5: out = sum
    c_k alpha_k T_k): render(a c1 + b c2) == a render(c1) + b render(c2), alphas unchanged bit for bit."""
    from gsplat.rendering import rasterization

    dev = gpu_device
    N, C, W, H = 300_000, 2, 1242, 375
    g = synth_scene.make_gaussians(N, sh_degree=None, seed=5)
    vm, Ks = synth_scene.make_cameras(C, W, H)
    p = {k: v.to(dev) for k, v in g.items()}
    vm, Ks = vm.to(dev), Ks.to(dev)
    c1 = p["colors"]
    c2 = torch.rand(N, 3, generator=torch.Generator().manual_seed(6)).to(dev)

    def render(c):
        with torch.no_grad():
            return rasterization(p["means"], p["quats"], p["scales"], p["opacities"], c, vm, Ks, W, H, near_plane=0.1,
                                 packed=False, render_mode="RGB")[:2]

    r1, a1 = render(c1)
    r2, a2 = render(c2)
    r3, a3 = render(0.25 * c1 + 1.5 * c2)
    assert torch.equal(a1, a2) and torch.equal(a1, a3)
    assert_close("linearity", r3, 0.25 * r1.double() + 1.5 * r2.double(), atol=2e-5)
