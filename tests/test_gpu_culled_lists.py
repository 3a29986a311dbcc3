"""Culled tile lists (include/gsplat_mi355x.h, gs_isect_bin_count "Culled lists"): rasterization() with gradients (or
packed) runs on tile lists from which the pairs that cannot reach alpha 1/255 anywhere on the tile are left out.

  * every culled list is a SUBSEQUENCE of gsplat's list of the same tile (same order, nothing added);
  * every dropped pair is one gsplat's compositor skips at every pixel of the tile: its alpha, evaluated in fp64 over
    the tile's 256 pixel centres exactly as Appendix A.5 does, stays below 1/255;
  * the rendered images are bit-identical with and without the cull, the gradients agree to float-atomic order;
  * `meta` still hands out gsplat's exact isect_ids / flatten_ids / isect_offsets."""
import math

import pytest
import torch

import synth_scene
from util import close_report

pytestmark = pytest.mark.gpu


def _project(dev, N, C, W, H, seed=3, extent=15.0):
    from gsplat.cuda import _wrapper as Wm
    from gsplat import _backend as B

    g = synth_scene.make_gaussians(N, sh_degree=None, extent=extent, seed=seed)
    vm, Ks = synth_scene.make_cameras(C, W, H)
    g["opacities"][::7] = 0.0  # instances that are invalid in this frame arrive with opacity exactly 0
    g = {k: v.to(dev) for k, v in g.items()}
    params = Wm._make_params(N, C, W, H, 16, 0.3, 0.1, 1e10, 0.0, True, B.GS_COLOR_N, 3, None, 0, True)
    with torch.no_grad():
        splats, means2d, radii, tpg = Wm._ProjectSplats.apply(g["means"], g["quats"], g["scales"], g["opacities"],
                                                              g["colors"], vm.to(dev), Ks.to(dev), None, params, None)
    return splats, tpg


@pytest.mark.parametrize("shape", [(20000, 2, 320, 200), (3000, 1, 1242, 375), (60000, 3, 200, 120)])
def test_culled_lists_are_subsequences_and_drop_only_invisible_pairs(gpu_device, shape):
    from gsplat.cuda import _wrapper as Wm

    N, C, W, H = shape
    TW, TH = math.ceil(W / 16), math.ceil(H / 16)
    splats, tpg = _project(gpu_device, N, C, W, H, extent=8.0 if N == 60000 else 15.0)
    ids_e, fids_e, offs_e, _, _ = Wm.isect_tiles_binned(splats, tpg, TW, TH, front=False)
    ids_c, fids_c, offs_c, _, _ = Wm.isect_tiles_binned(splats, tpg, TW, TH, front=False, cull_tile_size=16)
    I_e, I_c = fids_e.numel(), fids_c.numel()
    print(f"[cull] {shape}: {I_e} pairs -> {I_c} ({I_c / max(I_e, 1):.3f})")
    assert 0 < I_c <= I_e
    ids_e, ids_c = ids_e.cpu(), ids_c.cpu()
    fe, fc = fids_e.cpu().long(), fids_c.cpu().long()
    # (tile, depth, row) identifies a pair; both arrays are sorted by it, so "subsequence" = sorted-set inclusion with
    # equal multiplicity one
    key_e = torch.stack([ids_e, fe], 1)
    key_c = torch.stack([ids_c, fc], 1)
    assert torch.equal(torch.unique(key_c, dim=0), key_c), "culled lists: not in gsplat's order or with repeats"
    both = torch.cat([key_e, key_c])
    uniq, cnt = torch.unique(both, dim=0, return_counts=True)
    assert uniq.shape[0] == I_e, "culled lists hold a pair gsplat's lists do not"
    dropped = uniq[cnt == 1]
    assert dropped.shape[0] == I_e - I_c
    # offsets of the culled lists
    tile_of = lambda ids: (ids >> 32)  # camera|tile  (depth bits below)
    T = TW * TH
    tn_bits = int(math.floor(math.log2(T))) + 1
    flat_tile = lambda ids: (tile_of(ids) >> tn_bits) * T + (tile_of(ids) & ((1 << tn_bits) - 1))
    cnt_c = torch.bincount(flat_tile(ids_c), minlength=C * T)
    assert torch.equal(torch.cumsum(cnt_c, 0) - cnt_c, offs_c.cpu().reshape(-1).long())
    # the dropped pairs: alpha over the tile's pixel centres, fp64, A.5's formula
    if dropped.shape[0]:
        rec = splats.reshape(-1, 16).cpu().double()
        row = dropped[:, 1]
        t = flat_tile(dropped[:, 0]) % T
        tx, ty = (t % TW).double(), (t // TW).double()
        px = torch.arange(16, dtype=torch.float64) + 0.5
        gx = (tx * 16)[:, None, None] + px[None, None, :]
        gy = (ty * 16)[:, None, None] + px[None, :, None]
        r = rec[row]
        dx, dy = gx - r[:, 0, None, None], gy - r[:, 1, None, None]
        sigma = 0.5 * (r[:, 2, None, None] * dx * dx + r[:, 4, None, None] * dy * dy) + r[:, 3, None, None] * dx * dy
        alpha = torch.clamp(r[:, 5, None, None] * torch.exp(-sigma), max=0.999)
        alpha = torch.where(sigma < 0, torch.zeros_like(alpha), alpha)
        worst = float(alpha.reshape(alpha.shape[0], -1).max(dim=1).values.max())
        print(f"[cull] dropped {dropped.shape[0]} pairs, their largest alpha on the tile {worst:.6f} (1/255 = {1 / 255:.6f})")
        assert worst < 1.0 / 255.0
    # zero-opacity splats leave no pair behind unless their box is too large to be culled (> 64 tiles)
    rec_c = splats.reshape(-1, 16)[fids_c.long()].cpu()
    zero_left = rec_c[:, 5] == 0
    if bool(zero_left.any()):
        boxes = tpg.reshape(-1)[fids_c.long()].cpu()[zero_left]
        assert int(boxes.min()) > 64, "a zero-opacity splat with a small box kept a pair"


@pytest.mark.parametrize("packed", [False, True])
def test_render_is_bit_identical_with_and_without_the_cull(gpu_device, packed):
    from gsplat.cuda import _wrapper as Wm
    from gsplat.rendering import rasterization

    dev = gpu_device
    N, C, W, H = 40000, 3, 330, 190
    g = synth_scene.make_gaussians(N, sh_degree=3, extent=12.0, seed=5)
    vm, Ks = synth_scene.make_cameras(C, W, H)
    gen = torch.Generator().manual_seed(2)
    wc = torch.randn(C, H, W, 4, generator=gen).to(dev)
    wa = torch.randn(C, H, W, 1, generator=gen).to(dev)
    out = {}
    for cull in (True, False):
        Wm.CULL_LISTS = cull
        try:
            p = {k: v.to(dev).requires_grad_(True) for k, v in g.items()}
            rc, ra, meta = rasterization(p["means"], p["quats"], p["scales"], p["opacities"], p["colors"], vm.to(dev),
                                         Ks.to(dev), W, H, sh_degree=3, packed=packed, render_mode="RGB+ED",
                                         absgrad=True, rasterize_mode="antialiased", near_plane=0.1)
            ((rc * wc).sum() + (ra * wa).sum()).backward()
            out[cull] = (rc.detach(), ra.detach(), {k: v.grad for k, v in p.items()}, meta["means2d"].absgrad,
                         {k: meta[k] for k in ("isect_ids", "flatten_ids", "isect_offsets", "tiles_per_gauss")})
        finally:
            Wm.CULL_LISTS = True
    (rc1, ra1, g1, ag1, m1), (rc0, ra0, g0, ag0, m0) = out[True], out[False]
    assert torch.equal(rc1, rc0) and torch.equal(ra1, ra0)
    for k in g0:
        _, rel_l2, _ = close_report(f"cull.v_{k}", g1[k], g0[k])
        assert rel_l2 <= 1e-5, (k, rel_l2)  # float-atomic order (the run-to-run bar of test_gpu_contract)
    _, rel_l2, _ = close_report("cull.absgrad", ag1, ag0)
    assert rel_l2 <= 1e-5
    # meta: gsplat's exact arrays either way
    for k in m0:
        assert torch.equal(m1[k], m0[k]), k


def test_eval_cache_calls_keep_full_lists(gpu_device):
    """no-grad, non-packed calls feed the opacity re-render cache: their lists must not depend on the opacities."""
    from gsplat.rendering import clear_cache, rasterization

    dev = gpu_device
    N, C, W, H = 20000, 1, 320, 200
    g = {k: v.to(dev) for k, v in synth_scene.make_gaussians(N, sh_degree=None, extent=12.0, seed=9).items()}
    vm, Ks = (t.to(dev) for t in synth_scene.make_cameras(C, W, H))
    clear_cache()
    with torch.no_grad():
        low = g["opacities"] * 0.01
        rasterization(g["means"], g["quats"], g["scales"], low, g["colors"], vm, Ks, W, H, near_plane=0.1)
        rc1, ra1, _ = rasterization(g["means"], g["quats"], g["scales"], g["opacities"], g["colors"], vm, Ks, W, H,
                                    near_plane=0.1)  # cache hit: re-render of the first call's lists
        clear_cache()
        rc0, ra0, _ = rasterization(g["means"], g["quats"], g["scales"], g["opacities"], g["colors"], vm, Ks, W, H,
                                    near_plane=0.1)
    assert torch.equal(rc1, rc0) and torch.equal(ra1, ra0)
