"""GPU parity tests: HIP kernels (through the C ABI / the gsplat-shaped host wrappers) vs the CPU oracle.

Bars (BASELINE.json north_star): bit-exact tile-intersection ids, sort order and offsets; rendered
RGB/depth/alpha and parameter gradients within 1e-5 absolute + 1e-4 relative (fp32).  PARITY UNPINNED
against gsplat itself (see oracle/__init__.py); the oracle runs in fp64 on the same fp32 inputs.

A handful of float outputs legitimately flip discrete decisions (alpha < 1/255 skip, T <= 1e-4 stop,
ceil() of the radius) when `exp`/fma rounding differs by an ulp; those tests state the flip budget they
allow next to the assertion.
"""
import math

import numpy as np
import pytest
import torch

import oracle
import synth_scene
from util import assert_close, close_report, to_dev

pytestmark = pytest.mark.gpu


def _scene(N, C, W, H, sh_degree=3, extent=12.0, seed=42):
    g = synth_scene.make_gaussians(N, sh_degree=sh_degree, extent=extent, seed=seed)
    vm, Ks = synth_scene.make_cameras(C, W, H)
    return g, vm, Ks


# --------------------------------------------------------------------------------------------- SH
@pytest.mark.parametrize("deg", [0, 1, 2, 3, 4])
def test_sh_fwd_bwd(gpu_device, deg):
    from gsplat.cuda._wrapper import spherical_harmonics

    torch.manual_seed(deg)
    n, K = 3000, 25 if deg == 4 else 16
    dirs = torch.randn(n, 3) * 3.0
    coeffs = torch.randn(n, K, 3)
    masks = torch.rand(n) > 0.2
    v = torch.randn(n, 3)
    d_g, c_g = dirs.to(gpu_device).requires_grad_(True), coeffs.to(gpu_device).requires_grad_(True)
    out = spherical_harmonics(deg, d_g, c_g, masks=masks.to(gpu_device))
    (out * v.to(gpu_device)).sum().backward()
    d_r, c_r = dirs.double().requires_grad_(True), coeffs.double().requires_grad_(True)
    ref = oracle.spherical_harmonics(deg, d_r, c_r, masks=masks)
    (ref * v.double()).sum().backward()
    assert_close(f"sh{deg}.out", out, ref)
    assert_close(f"sh{deg}.v_coeffs", c_g.grad, c_r.grad)
    ref_vd = d_r.grad if d_r.grad is not None else torch.zeros_like(d_r)  # degree 0 does not depend on dirs
    assert_close(f"sh{deg}.v_dirs", d_g.grad, ref_vd, atol=2e-5)
    # band 0 is C0 * c (anchor: /root/reference/3D-Reconstruction-main/models/gaussians/basics.py:80)
    if deg == 0:
        sel = masks
        assert torch.allclose(out.cpu()[sel], 0.28209479177387814 * coeffs[sel, 0, :], atol=1e-6)
    # no-mask path and zero rows
    out2 = spherical_harmonics(deg, dirs.to(gpu_device), coeffs.to(gpu_device))
    assert_close(f"sh{deg}.nomask", out2, oracle.spherical_harmonics(deg, dirs.double(), coeffs.double()))
    assert float(out.cpu()[~masks].abs().max()) == 0.0


def test_sh_empty(gpu_device):
    from gsplat.cuda._wrapper import spherical_harmonics

    out = spherical_harmonics(3, torch.zeros(0, 3, device=gpu_device), torch.zeros(0, 16, 3, device=gpu_device))
    assert out.shape == (0, 3)


# --------------------------------------------------------------------------------------------- integer stages
def test_scan_and_radix_sort(gpu_device):
    from gsplat.cuda._wrapper import exclusive_scan_i32, radix_sort_pairs

    rng = np.random.default_rng(0)
    for n in [1, 63, 2048, 2049, 100_003, 1_300_000]:
        x = rng.integers(0, 50, size=n).astype(np.int32)
        out, total = exclusive_scan_i32(torch.from_numpy(x).to(gpu_device))
        ref = np.cumsum(x.astype(np.int64)) - x
        assert np.array_equal(out.cpu().numpy(), ref.astype(np.int32)), n
        assert int(total.item()) == int(x.sum())
    for n, bits in [(1, 40), (5000, 46), (4096, 33), (70_001, 50), (1_000_003, 46), (300_000, 64)]:
        hi = (1 << min(bits + 3, 63)) - 1  # garbage above end_bit must be ignored
        keys = rng.integers(0, hi, size=n, dtype=np.int64)
        keys[: n // 3] &= (1 << 20) - 1  # many ties -> exercises stability
        vals = np.arange(n, dtype=np.int32)
        k, v = radix_sort_pairs(torch.from_numpy(keys).to(gpu_device), torch.from_numpy(vals).to(gpu_device), bits)
        mask = (1 << bits) - 1 if bits < 64 else -1
        order = np.argsort((keys & mask).astype(np.uint64), kind="stable")
        assert np.array_equal(v.cpu().numpy(), vals[order]), (n, bits)
        assert np.array_equal(k.cpu().numpy(), keys[order]), (n, bits)


def _synthetic_records(rng, C, N, TW, TH, hot_tile_rows, depth_levels, dev, box=4):
    """Splat records carrying only what the intersection stage reads: depth [10] and the tile box [13], [14]."""
    from gsplat import _backend as B

    x0 = rng.integers(0, TW, size=(C, N))
    y0 = rng.integers(0, TH, size=(C, N))
    w = rng.integers(1, box, size=(C, N))
    h = rng.integers(1, box, size=(C, N))
    big = rng.random((C, N)) < 0.01  # a few boxes above the cooperative threshold (> 16 tiles)
    w = np.where(big, rng.integers(5, max(6, TW), size=(C, N)), w)
    h = np.where(big, rng.integers(4, max(5, TH), size=(C, N)), h)
    x1 = np.minimum(x0 + w, TW)
    y1 = np.minimum(y0 + h, TH)
    # pile `hot_tile_rows` rows of camera 0 onto tile (1, 1): a segment longer than the LDS classes
    x0[0, :hot_tile_rows], y0[0, :hot_tile_rows] = 1, 1
    x1[0, :hot_tile_rows], y1[0, :hot_tile_rows] = 2, 2
    visible = rng.random((C, N)) < 0.6
    visible[0, :hot_tile_rows] = True
    tiles = np.where(visible, (x1 - x0) * (y1 - y0), 0).astype(np.int32)
    # few distinct depths -> many exact ties, which must come out in ascending flat row id
    depth = (0.5 + rng.integers(0, depth_levels, size=(C, N)) * 0.37).astype(np.float32)
    rec = np.zeros((C, N, B.GS_REC), dtype=np.float32)
    rec[..., 10] = depth
    rec[..., 13] = (x0 | (y0 << 16)).astype(np.int32).view(np.float32)
    rec[..., 14] = (x1 | (y1 << 16)).astype(np.int32).view(np.float32)
    ref_keys, ref_ids = [], []
    tile_bits = int(math.floor(math.log2(TW * TH))) + 1
    dbits = depth.view(np.int32).astype(np.int64)
    for c in range(C):
        for n in np.nonzero(tiles[c])[0]:
            ys, xs = np.meshgrid(np.arange(y0[c, n], y1[c, n]), np.arange(x0[c, n], x1[c, n]), indexing="ij")
            t = (ys * TW + xs).reshape(-1).astype(np.int64)
            ref_keys.append((np.int64(c) << (32 + tile_bits)) | (t << 32) | dbits[c, n])
            ref_ids.append(np.full(t.shape, c * N + n, dtype=np.int32))
    keys = np.concatenate(ref_keys) if ref_keys else np.zeros(0, np.int64)
    ids = np.concatenate(ref_ids) if ref_ids else np.zeros(0, np.int32)
    order = np.argsort(keys, kind="stable")  # emission order is ascending flat row id (Appendix A.3)
    keys, ids = keys[order], ids[order]
    lin = (keys >> (32 + tile_bits)) * (TW * TH) + ((keys >> 32) & ((1 << tile_bits) - 1))
    offsets = np.searchsorted(lin, np.arange(C * TW * TH), side="left").astype(np.int32)
    return (torch.from_numpy(rec).to(dev), torch.from_numpy(tiles).to(dev), keys, ids, offsets.reshape(C, TH, TW))


@pytest.mark.parametrize("C,N,TW,TH,hot,levels", [
    (2, 20_000, 13, 7, 6000, 5),      # global-scratch segment (> 4096), LDS classes, heavy ties
    (1, 3_000, 5, 4, 1500, 1 << 20),  # mid class (1024 < n <= 4096), nearly no ties
    (3, 500, 78, 24, 0, 3),           # wide tile grid, short segments, many empty tiles
    (1, 1, 2, 2, 1, 1),
])
def test_binned_isect_matches_stable_sort(gpu_device, C, N, TW, TH, hot, levels):
    """Both intersection pipelines (binned; emit + full-width radix sort + offset_encode) against numpy's stable sort of
    the (camera|tile|depth) keys: isect_ids, flatten_ids and isect_offsets bit for bit."""
    from gsplat.cuda._wrapper import isect_offset_encode, isect_tiles_binned, isect_tiles_from_splats

    rng = np.random.default_rng(C * 1000 + N)
    rec, tiles, keys, ids, offsets = _synthetic_records(rng, C, N, TW, TH, hot, levels, gpu_device)
    for _ in range(2):  # the scatter order inside a segment is arbitrary; the result must not be
        k, f, o, order, _ = isect_tiles_binned(rec, tiles, TW, TH)
        assert np.array_equal(k.cpu().numpy(), keys)
        assert np.array_equal(f.cpu().numpy(), ids)
        assert np.array_equal(o.cpu().numpy(), offsets)
        if order is not None:
            assert np.array_equal(np.sort(order.cpu().numpy()), np.arange(C * TW * TH))
    k2, f2 = isect_tiles_from_splats(rec, tiles, TW, TH)
    assert np.array_equal(k2.cpu().numpy(), keys)
    assert np.array_equal(f2.cpu().numpy(), ids)
    assert np.array_equal(isect_offset_encode(k2, C, TW, TH).cpu().numpy(), offsets)


@pytest.mark.parametrize("staged,workgroups,regions,box", [(False, 256, 0, 4), (True, 256, 0, 4), (True, 2, 0, 4),
                                                          (True, 3, 2, 4), (True, 256, 0, 9), (True, 2, 0, 9)])
def test_binned_isect_scatter_modes(gpu_device, staged, workgroups, regions, box):
    """The region scatter with direct 8-byte stores and staged through LDS (gs_isect_bin_scatter_regions, tile_counts):
    one slice per workgroup, groups of slices whose pairs exceed the staging area (the chunk is halved until it fits)
    and slices that exceed it alone (scattered directly) all give numpy's stable sort, bit for bit."""
    from gsplat.cuda import _wrapper as Wm

    rng = np.random.default_rng(77)
    C, N, TW, TH = 1, 40_000, 60, 34
    rec, tiles, keys, ids, offsets = _synthetic_records(rng, C, N, TW, TH, 0, 50, gpu_device, box=box)
    saved = (Wm.STAGED_SCATTER, Wm.SCATTER_WORKGROUPS, Wm.SCATTER_REGIONS)
    try:
        Wm.STAGED_SCATTER, Wm.SCATTER_WORKGROUPS, Wm.SCATTER_REGIONS = staged, workgroups, regions
        k, f, o, _, _ = Wm.isect_tiles_binned(rec, tiles, TW, TH)
    finally:
        Wm.STAGED_SCATTER, Wm.SCATTER_WORKGROUPS, Wm.SCATTER_REGIONS = saved
    n_slices = -(-N // 4096)
    print(f"[scatter] staged={staged} workgroups={workgroups} box<{box}: {keys.size} pairs over {n_slices} slices "
          f"({keys.size // n_slices} per slice)")
    assert np.array_equal(k.cpu().numpy(), keys)
    assert np.array_equal(f.cpu().numpy(), ids)
    assert np.array_equal(o.cpu().numpy(), offsets)


def test_binned_isect_empty(gpu_device):
    from gsplat import _backend as B
    from gsplat.cuda._wrapper import isect_tiles_binned

    rec = torch.zeros(2, 100, B.GS_REC, device=gpu_device)
    tiles = torch.zeros(2, 100, dtype=torch.int32, device=gpu_device)
    k, f, o, _, _ = isect_tiles_binned(rec, tiles, 4, 3)
    assert k.numel() == 0 and f.numel() == 0 and int(o.abs().sum()) == 0 and o.shape == (2, 3, 4)


# --------------------------------------------------------------------------------------------- staged pipeline
def _run_gpu_stages(g, vm, Ks, W, H, dev, mode="antialiased", sh_degree=3, render_mode="RGB+ED",
                    near=0.1, radius_clip=0.0, backgrounds=None):
    from gsplat.rendering import rasterization

    gg = to_dev(g, dev)
    for v in gg.values():
        v.requires_grad_(True)
    vmg = vm.to(dev).requires_grad_(True)
    rc, ra, meta = rasterization(
        gg["means"], gg["quats"], gg["scales"], gg["opacities"], gg["colors"], vmg, Ks.to(dev), W, H,
        near_plane=near, radius_clip=radius_clip, sh_degree=sh_degree, packed=False, render_mode=render_mode,
        absgrad=True, rasterize_mode=mode, backgrounds=backgrounds)
    return gg, vmg, rc, ra, meta


def _run_oracle(g, vm, Ks, W, H, mode="antialiased", sh_degree=3, render_mode="RGB+ED", near=0.1,
                radius_clip=0.0, backgrounds=None):
    go = {k: v.double().requires_grad_(True) for k, v in g.items()}
    vmo = vm.double().requires_grad_(True)
    rc, ra, meta = oracle.rasterization(
        go["means"], go["quats"], go["scales"], go["opacities"], go["colors"], vmo, Ks.double(), W, H,
        near_plane=near, radius_clip=radius_clip, sh_degree=sh_degree, render_mode=render_mode,
        rasterize_mode=mode, backgrounds=backgrounds)
    return go, vmo, rc, ra, meta


def test_projection_fwd(gpu_device):
    W, H, C, N = 160, 96, 2, 4096
    g, vm, Ks = _scene(N, C, W, H)
    # edge cases: behind camera, beyond far is untestable with 1e10; tiny + huge + degenerate scales
    g["scales"][:50] = 1e-3
    g["scales"][50:100, 2] = 0.0
    g["scales"][100:120] = 2.0
    gg, vmg, rc, ra, meta = _run_gpu_stages(g, vm, Ks, W, H, gpu_device)
    go, vmo, rco, rao, mo = _run_oracle(g, vm, Ks, W, H)
    r_g, r_o = meta["radii"].cpu(), mo["radii"]
    diff = (r_g != r_o)
    # radius = ceil(3 sqrt(lambda)) and the cull tests are float-derived: allow |delta| <= 1 on <= 0.1 % of rows
    print(f"[parity] radii mismatches: {int(diff.sum())} of {diff.numel()}")
    assert float(diff.double().mean()) <= 1e-3
    both = (r_g > 0) & (r_o > 0)
    assert int((r_g[both] - r_o[both]).abs().max()) <= 1
    assert int(both.sum()) > 500
    assert_close("means2d", meta["means2d"].cpu()[both], mo["means2d"][both], atol=1e-4)  # pixels, fp32 ulp at ~1e3
    assert_close("depths", meta["depths"].cpu()[both], mo["depths"][both])
    assert_close("conics", meta["conics"].cpu()[both], mo["conics"][both], rtol=2e-4)
    assert_close("opacities*comp", meta["opacities"].cpu()[both], mo["opacities"][both])


def test_projection_is_bitwise_the_same_alone_and_in_a_batch(gpu_device):
    """gs_projection_fwd runs one of two kernels (SH coefficient rows requested up front from four cameras on, on demand
    below): a camera's splat records must not depend on which one projected it."""
    from gsplat import _backend as B
    from gsplat.cuda import _wrapper as Wm

    N, C, W, H = 30_000, 5, 310, 94
    g = to_dev(synth_scene.make_gaussians(N, sh_degree=3, seed=4, extent=15.0), gpu_device)
    vm, Ks = (t.to(gpu_device) for t in synth_scene.make_cameras(C, W, H, frame=2))

    def project(vm, Ks):
        params = Wm._make_params(N, vm.shape[0], W, H, 16, 0.3, 0.1, 1e10, 0.0, True, B.GS_COLOR_SH, 3, 3, 16, True)
        with torch.no_grad():
            return Wm._ProjectSplats.apply(g["means"], g["quats"], g["scales"], g["opacities"], g["colors"], vm, Ks,
                                           None, params, None)

    sp5, _, r5, t5 = project(vm, Ks)
    seen = 0
    for c in range(C):
        sp1, _, r1, t1 = project(vm[c:c + 1], Ks[c:c + 1])
        assert torch.equal(r5[c], r1[0]) and torch.equal(t5[c], t1[0])
        vis = t5[c] > 0
        seen += int(vis.sum())
        assert torch.equal(sp5[c][vis], sp1[0][vis]), f"camera {c}: records differ between the batch and the single call"
    assert seen > 1000
    # ... nor on what the caller asks for in gs_proj_params.sh_rows (1 = up front, 2 = on demand), at any camera count
    for c_n in (1, 3, 5):
        got = []
        for rows in (1, 2):
            Wm._tls.sh_rows = rows
            try:
                got.append(project(vm[:c_n], Ks[:c_n]))
            finally:
                Wm._tls.sh_rows = 0
        (sa, _, ra, ta), (sb, _, rb, tb) = got
        assert torch.equal(ra, rb) and torch.equal(ta, tb)
        assert torch.equal(sa[ta > 0], sb[tb > 0]), f"{c_n} cameras: records differ between sh_rows = 1 and 2"


def test_fill_zero_large_and_small(gpu_device):
    """gs_fill_zero: buffers above 32 MiB go through the nontemporal fill kernel, the rest through hipMemsetAsync; either
    way exactly `bytes` bytes are cleared and nothing around them."""
    from gsplat import _backend as B

    guard = 1024  # floats on either side that must keep their ones
    for n_bytes in ((32 << 20) + 16, (32 << 20) + 4096 + 48, (48 << 20), (32 << 20), 4096, (32 << 20) + 4):
        n = n_bytes // 4
        buf = torch.ones(n + 2 * guard, device=gpu_device, dtype=torch.float32)
        inner = buf[guard:guard + n]
        assert inner.data_ptr() % 16 == 0
        B.check(B.lib().gs_fill_zero(inner.data_ptr(), n_bytes, B.stream_handle(gpu_device)), "gs_fill_zero")
        torch.cuda.synchronize()
        assert float(inner.abs().max()) == 0.0, n_bytes
        assert float(buf[:guard].min()) == 1.0 and float(buf[guard + n:].min()) == 1.0, n_bytes
    # a 4-byte aligned start (not 16): the memset path, whatever the size
    buf = torch.ones((40 << 20) // 4 + 8, device=gpu_device, dtype=torch.float32)
    B.check(B.lib().gs_fill_zero(buf.data_ptr() + 4, 40 << 20, B.stream_handle(gpu_device)), "gs_fill_zero")
    torch.cuda.synchronize()
    assert float(buf[1:1 + (40 << 20) // 4].abs().max()) == 0.0 and float(buf[0]) == 1.0 and float(buf[-1]) == 1.0


def test_isect_sort_offsets_bit_exact(gpu_device):
    """Integer stages are compared bit-for-bit: the oracle is fed the GPU's own fp32 means2d/radii/depths."""
    for (W, H, C, N) in [(160, 96, 2, 4096), (375, 250, 5, 20000), (33, 17, 1, 300)]:
        g, vm, Ks = _scene(N, C, W, H)
        gg, vmg, rc, ra, meta = _run_gpu_stages(g, vm, Ks, W, H, gpu_device)
        TW, TH = meta["tile_width"], meta["tile_height"]
        tpg, ids, fids = oracle.isect_tiles(meta["means2d"].cpu(), meta["radii"].cpu(), meta["depths"].cpu(), 16, TW, TH)
        assert torch.equal(meta["tiles_per_gauss"].cpu(), tpg)
        used = 32 + int(math.floor(math.log2(TW * TH))) + 1 + int(math.floor(math.log2(C))) + 1
        assert used <= 64
        assert torch.equal(meta["isect_ids"].cpu(), ids), "sorted intersection keys differ"
        assert torch.equal(meta["flatten_ids"].cpu(), fids), "sort order (flatten ids) differs"
        offs = oracle.isect_offset_encode(ids, C, TW, TH)
        assert torch.equal(meta["isect_offsets"].cpu(), offs)
        print(f"[parity] isect exact: W{W} H{H} C{C} N{N} I={ids.numel()}")


def _splat_fields(meta_splats):
    s = meta_splats
    return s[..., 0:2], s[..., 2:5], s[..., 5], s[..., 6:10]


@pytest.mark.parametrize("render_mode,bgflag", [("RGB+ED", False), ("RGB", True), ("ED", False), ("RGB+D", True)])
def test_raster_fwd_bwd_given_same_inputs(gpu_device, render_mode, bgflag):
    """rasterize_to_pixels fwd + bwd on identical inputs (the GPU's splat records and sorted lists)."""
    from gsplat import _backend as B
    from gsplat.cuda._wrapper import _RasterizeSplats

    W, H, C, N = 150, 70, 2, 3000
    g, vm, Ks = _scene(N, C, W, H)
    bg = torch.rand(C, 3) if bgflag else None
    gg, vmg, rc0, ra0, meta = _run_gpu_stages(g, vm, Ks, W, H, gpu_device, render_mode=render_mode,
                                              backgrounds=bg.to(gpu_device) if bgflag else None)
    ch = rc0.shape[-1]
    ed = render_mode in ("ED", "RGB+ED")
    # re-run the projection node alone to get the splat records as a leaf, then the raster node alone
    from gsplat.cuda import _wrapper as Wm
    campos = Wm._camera_centers(vmg.detach())
    want_depth = render_mode in ("RGB+D", "RGB+ED", "D", "ED")
    if render_mode in ("D", "ED"):
        params = Wm._make_params(N, C, W, H, 16, 0.3, 0.1, 1e10, 0.0, True, B.GS_COLOR_NONE, 0, None, 0, True)
        splats, m2d, radii, tiles = Wm._ProjectSplats.apply(
            gg["means"].detach(), gg["quats"].detach(), gg["scales"].detach(), gg["opacities"].detach(), None,
            vmg.detach(), Ks.to(gpu_device), None, params)
    else:
        params = Wm._make_params(N, C, W, H, 16, 0.3, 0.1, 1e10, 0.0, True, B.GS_COLOR_SH, 3, 3, 16, want_depth)
        splats, m2d, radii, tiles = Wm._ProjectSplats.apply(
            gg["means"].detach(), gg["quats"].detach(), gg["scales"].detach(), gg["opacities"].detach(),
            gg["colors"].detach(), vmg.detach(), Ks.to(gpu_device), campos, params)
    splats = splats.detach().clone().requires_grad_(True)
    bg_full = None
    if bgflag:
        bg_full = bg.to(gpu_device)
        if want_depth:
            bg_full = torch.cat([bg_full, torch.zeros(C, 1, device=gpu_device)], -1)
        bg_full = bg_full.clone().requires_grad_(True)
    m2d_hook = m2d.detach().clone().requires_grad_(True)
    rc, ra, isect_ids, flatten_ids, offsets, last_ids = _RasterizeSplats.apply(
        splats, m2d_hook, tiles, bg_full, W, H, 16, ch, ed, True, None)
    torch.manual_seed(3)
    wc, wa = torch.randn(C, H, W, ch), torch.randn(C, H, W, 1)
    ((rc * wc.to(gpu_device)).sum() + (ra * wa.to(gpu_device)).sum()).backward()

    # oracle on the same records / lists
    s64 = splats.detach().double().cpu()
    m2, cn, op, col = [t.clone().requires_grad_(True) for t in _splat_fields(s64)]
    colr = col[..., :ch]
    bgo = bg_full.detach().double().cpu().requires_grad_(True) if bgflag else None
    rco, rao, lasto = oracle.rasterize_to_pixels(m2, cn, colr, op, W, H, 16, offsets.cpu(), flatten_ids.cpu(), bgo)
    if ed:
        rco = torch.cat([rco[..., :-1], rco[..., -1:] / rao.clamp(min=1e-10)], -1)
    ((rco * wc.double()).sum() + (rao * wa.double()).sum()).backward()

    # forward.  exp()/fma rounding can flip "alpha < 1/255" or "T <= 1e-4" for a borderline pixel: budget 0.1 %
    assert_close(f"{render_mode}.alphas", ra, rao, max_bad_frac=1e-3)
    assert_close(f"{render_mode}.colors", rc, rco, max_bad_frac=1e-3)
    mism = float((last_ids.cpu() != lasto).double().mean())
    print(f"[parity] last_ids mismatch fraction {mism:.3e}")
    assert mism <= 1e-3
    # backward (atomics in fp32, run-to-run order varies): rtol 1e-4 + atol 1e-5 scaled by the loss weights
    vs = splats.grad.cpu()
    assert_close(f"{render_mode}.v_conics", vs[..., 2:5], cn.grad, max_bad_frac=2e-3, max_rel_l2=1e-4, scale_atol=True)
    assert_close(f"{render_mode}.v_opacity", vs[..., 5], op.grad, max_bad_frac=2e-3, max_rel_l2=1e-4, scale_atol=True)
    assert_close(f"{render_mode}.v_colors", vs[..., 6:6 + ch], col.grad[..., :ch], max_bad_frac=2e-3, max_rel_l2=1e-4, scale_atol=True)
    assert_close(f"{render_mode}.v_means2d", m2d_hook.grad.cpu(), m2.grad, max_bad_frac=2e-3, max_rel_l2=1e-4, scale_atol=True)
    assert float(vs[..., 12:].abs().max()) == 0.0
    if bgflag:
        assert_close(f"{render_mode}.v_bg", bg_full.grad, bgo.grad, rtol=2e-4)
    # absgrad attribute contract (R/models/trainers/base.py:282)
    assert hasattr(m2d_hook, "absgrad") and m2d_hook.absgrad.shape == (C, N, 2)
    assert bool((m2d_hook.absgrad >= m2d_hook.grad.abs() - 1e-4 * m2d_hook.absgrad.abs() - 1e-6).all())


def test_absgrad_small_image(gpu_device):
    """absgrad = sum over pixels of |per-pixel d/d means2d| -- checked against per-pixel autograd on a tiny image."""
    from gsplat import _backend as B
    from gsplat.cuda import _wrapper as Wm

    W, H, C, N = 20, 18, 1, 60
    g, vm, Ks = _scene(N, C, W, H, sh_degree=None, extent=4.0)
    g["scales"] *= 3.0
    dev = gpu_device
    gg = to_dev(g, dev)
    params = Wm._make_params(N, C, W, H, 16, 0.3, 0.1, 1e10, 0.0, False, B.GS_COLOR_N, 3, None, 0, False)
    splats, m2d, radii, tiles = Wm._ProjectSplats.apply(gg["means"], gg["quats"], gg["scales"], gg["opacities"],
                                                        gg["colors"], vm.to(dev), Ks.to(dev), None, params)
    splats = splats.detach().requires_grad_(True)
    hook = m2d.detach().requires_grad_(True)
    rc, ra, ids, fids, offs, last = Wm._RasterizeSplats.apply(splats, hook, tiles, None, W, H, 16, 3, False, True, None)
    torch.manual_seed(1)
    wc, wa = torch.randn(C, H, W, 3), torch.randn(C, H, W, 1)
    ((rc * wc.to(dev)).sum() + (ra * wa.to(dev)).sum()).backward()
    s64 = splats.detach().double().cpu()
    m2, cn, op, col = _splat_fields(s64)
    ref = oracle.gsplat_ref.means2d_absgrad_oracle(m2, cn, col[..., :3], op, W, H, 16, offs.cpu(), fids.cpu(),
                                                   wc.double(), wa.double())
    assert int((radii > 0).sum()) > 10
    assert_close("absgrad", hook.absgrad, ref, max_bad_frac=5e-3, max_rel_l2=1e-4)


# --------------------------------------------------------------------------------------------- end to end
@pytest.mark.parametrize("mode,sh", [("antialiased", 3), ("classic", None)])
def test_end_to_end_fwd_bwd(gpu_device, mode, sh):
    W, H, C, N = 200, 120, 3, 6000
    g, vm, Ks = _scene(N, C, W, H, sh_degree=sh)
    g["opacities"][:300] = 0.0  # frame-invalid instances are exactly 0 (R/models/nodes/rigid.py:493)
    gg, vmg, rc, ra, meta = _run_gpu_stages(g, vm, Ks, W, H, gpu_device, mode=mode, sh_degree=sh)
    go, vmo, rco, rao, mo = _run_oracle(g, vm, Ks, W, H, mode=mode, sh_degree=sh)
    torch.manual_seed(7)
    wc, wa = torch.randn(C, H, W, 4) / (H * W) * 100, torch.randn(C, H, W, 1) / (H * W) * 100
    meta["means2d"].retain_grad()
    ((rc * wc.to(gpu_device)).sum() + (ra * wa.to(gpu_device)).sum()).backward()
    ((rco * wc.double()).sum() + (rao * wa.double()).sum()).backward()
    # End to end the two sides do NOT share tile lists: a radius that differs by one ulp-induced ceil() step adds
    # or drops a tile for that Gaussian, changing a few pixels by <= ~1e-2.  Budget: 0.2 % of pixels.
    assert_close(f"e2e.{mode}.alphas", ra, rao, max_bad_frac=2e-3)
    assert_close(f"e2e.{mode}.colors", rc, rco, max_bad_frac=2e-3)
    # Gradients: fp32 sums of thousands of terms (and float atomics, so run-to-run order varies).  Measured
    # (profiles/r02_parity_report.txt): rel-L2 <= 1.5e-5 for every parameter, and at the north_star's strict elementwise
    # bar (1e-5 abs + 1e-4 rel) <= 5e-4 of the elements outside.  Asserted at ~3x the measured values, stated as budgets.
    for k in ["means", "quats", "scales", "opacities", "colors"]:
        assert torch.isfinite(gg[k].grad).all()
        frac, rel_l2, _ = close_report(f"e2e.{mode}.v_{k}", gg[k].grad, go[k].grad)
        assert rel_l2 <= 5e-5, (k, rel_l2)
        assert frac <= 1.5e-3, (k, frac)
    # v_viewmats: 12 numbers per camera, each a sum over every visible Gaussian with heavy cancellation, so the absolute
    # term of the elementwise bar is taken relative to the largest entry (scale_atol); rel-L2 measured <= 6e-6
    frac, rel_l2, _ = close_report(f"e2e.{mode}.v_viewmats", vmg.grad[:, :3, :], vmo.grad[:, :3, :], scale_atol=True)
    assert rel_l2 <= 5e-5, rel_l2
    assert frac <= 1.0 / 36.0 + 1e-9, frac
    assert meta["means2d"].grad is not None and meta["means2d"].absgrad.shape == (C, N, 2)
    assert torch.isfinite(gg["opacities"].grad[:300]).all()  # d/d opacity at exactly 0 is finite (may be non-zero)


def test_viewer_call_and_empty(gpu_device):
    """Second call site (R/models/trainers/base.py:772-787): render_mode RGB, radius_clip=4, defaults; and N=0."""
    from gsplat.rendering import rasterization

    W, H, C, N = 97, 53, 1, 2000
    g, vm, Ks = _scene(N, C, W, H, sh_degree=None)
    gg = to_dev(g, gpu_device)
    rc, ra, meta = rasterization(gg["means"], gg["quats"], gg["scales"], gg["opacities"], gg["colors"],
                                 vm.to(gpu_device), Ks.to(gpu_device), W, H, packed=False, radius_clip=4.0)
    go = {k: v.double() for k, v in g.items()}
    rco, rao, mo = oracle.rasterization(go["means"], go["quats"], go["scales"], go["opacities"], go["colors"],
                                        vm.double(), Ks.double(), W, H, radius_clip=4.0)
    assert rc.shape == (C, H, W, 3) and ra.shape == (C, H, W, 1)
    assert int(((meta["radii"].cpu() > 0) & (meta["radii"].cpu() <= 4)).sum()) == 0
    assert_close("viewer.colors", rc, rco, max_bad_frac=2e-3)
    assert_close("viewer.alphas", ra, rao, max_bad_frac=2e-3)
    # no Gaussians at all
    z = lambda *s: torch.zeros(*s, device=gpu_device)
    rc, ra, meta = rasterization(z(0, 3), z(0, 4), z(0, 3), z(0), z(0, 3), vm.to(gpu_device), Ks.to(gpu_device), W, H,
                                 packed=False)
    assert float(rc.abs().max()) == 0.0 and float(ra.abs().max()) == 0.0
    # everything behind the camera
    gg2 = dict(gg)
    gg2["means"] = gg["means"].clone()
    gg2["means"][:, 0] = -50.0 - gg2["means"][:, 0].abs()
    rc, ra, meta = rasterization(gg2["means"], gg2["quats"], gg2["scales"], gg2["opacities"], gg2["colors"],
                                 vm.to(gpu_device), Ks.to(gpu_device), W, H, packed=False)
    assert int((meta["radii"] > 0).sum()) == 0 and meta["flatten_ids"].numel() == 0
    assert float(ra.abs().max()) == 0.0


def test_packed_meta(gpu_device):
    from gsplat.rendering import rasterization

    W, H, C, N = 120, 80, 3, 3000
    g, vm, Ks = _scene(N, C, W, H, sh_degree=None)
    gg = to_dev(g, gpu_device)
    a = rasterization(gg["means"], gg["quats"], gg["scales"], gg["opacities"], gg["colors"], vm.to(gpu_device),
                      Ks.to(gpu_device), W, H, packed=False, render_mode="RGB+ED")
    b = rasterization(gg["means"], gg["quats"], gg["scales"], gg["opacities"], gg["colors"], vm.to(gpu_device),
                      Ks.to(gpu_device), W, H, packed=True, render_mode="RGB+ED")
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    ma, mb = a[2], b[2]
    vis = ma["radii"] > 0
    cid, gid = torch.nonzero(vis, as_tuple=True)
    assert torch.equal(mb["camera_ids"], cid) and torch.equal(mb["gaussian_ids"], gid)
    assert torch.equal(mb["radii"], ma["radii"][vis])
    assert torch.equal(mb["means2d"], ma["means2d"][vis])
    assert torch.equal(mb["depths"], ma["depths"][vis])
    assert torch.equal(mb["conics"], ma["conics"][vis])
    # packed under no_grad (the render-only harness): the dense means2d is not written at all, `meta` takes the means of
    # the visible rows from the splat records
    with torch.no_grad():
        c = rasterization(gg["means"], gg["quats"], gg["scales"], gg["opacities"], gg["colors"], vm.to(gpu_device),
                          Ks.to(gpu_device), W, H, packed=True, render_mode="RGB+ED")
    assert torch.equal(a[0], c[0]) and torch.equal(a[1], c[1])
    mc = c[2]
    for k in ("camera_ids", "gaussian_ids", "radii", "means2d", "depths", "conics"):
        assert torch.equal(mc[k], mb[k]), k
