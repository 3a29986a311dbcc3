"""gsplat.losses.l1_ssim (gs_l1_ssim_fwd / _bwd) against the oracle's restatement of base.py:533-541 +
pytorch_msssim 1.0.0's SSIM (oracle/ssim_ref.py; parity unpinned: the package is not available here).  fp64 oracle vs
fp32 kernels: the L1 mean to 2e-6, the SSIM mean to 2e-5 (variances are differences E[x^2] - mu^2 of fp32 moments, in
the package as here), the gradient image to 1e-4 of its largest entry."""
import pytest
import torch

from oracle import ssim_ref as O

pytestmark = pytest.mark.gpu


def _images(H, W, seed, smooth=True):
    g = torch.Generator().manual_seed(seed)
    gt = torch.rand(H, W, 3, generator=g)
    if smooth:  # natural-image-like: low-pass the noise so that local variances are small against the means
        k = torch.ones(3, 1, 7, 7) / 49.0
        gt = torch.nn.functional.conv2d(gt.permute(2, 0, 1)[None], k, padding=3, groups=3)[0].permute(1, 2, 0)
        gt = (gt - gt.min()) / (gt.max() - gt.min())
    pred = (gt + 0.15 * torch.randn(H, W, 3, generator=g)).clamp(0, 1)
    mask = (torch.rand(H, W, generator=g) > 0.1).float()
    mask[: H // 5, : W // 3] = 0.0  # an egocar-like masked block: flat zero region in both images
    return pred.contiguous(), gt.contiguous(), mask


@pytest.mark.parametrize("H,W,use_mask,smooth", [(64, 96, False, True), (75, 131, True, True), (11, 11, False, False),
                                                 (37, 12, True, False), (375, 1242, True, True)])
def test_l1_ssim_matches_oracle(gpu_device, H, W, use_mask, smooth):
    from gsplat.losses import l1_ssim

    dev = gpu_device
    pred, gt, mask = _images(H, W, H * 1000 + W, smooth)
    mask = mask if use_mask else None
    pr = pred.double().requires_grad_(True)
    l1_ref, ss_ref = O.rgb_loss_terms(pr, gt.double(), mask.double() if use_mask else None)
    (0.8 * l1_ref + 0.2 * (1 - ss_ref)).backward()

    pg = pred.to(dev).requires_grad_(True)
    l1, ss = l1_ssim(pg, gt.to(dev), mask.to(dev) if use_mask else None)
    assert abs(float(l1.detach()) - float(l1_ref.detach())) < 2e-6, (float(l1.detach()), float(l1_ref.detach()))
    assert abs(float(ss.detach()) - float(ss_ref.detach())) < 2e-5, (float(ss.detach()), float(ss_ref.detach()))
    (0.8 * l1 + 0.2 * (1 - ss)).backward()
    got, ref = pg.grad.cpu().double(), pr.grad
    err = float((got - ref).abs().max())
    assert err < 1e-4 * float(ref.abs().max()) + 1e-12, (err, float(ref.abs().max()))
    # separately weighted terms (the backward takes two upstream scalars)
    pg2 = pred.to(dev).requires_grad_(True)
    l1b, ssb = l1_ssim(pg2, gt.to(dev), mask.to(dev) if use_mask else None)
    ssb.backward()
    pr2 = pred.double().requires_grad_(True)
    O.rgb_loss_terms(pr2, gt.double(), mask.double() if use_mask else None)[1].backward()
    err = float((pg2.grad.cpu().double() - pr2.grad).abs().max())
    assert err < 1e-4 * float(pr2.grad.abs().max()) + 1e-12


def test_l1_ssim_properties(gpu_device):
    from gsplat.losses import l1_ssim

    dev = gpu_device
    pred, gt, mask = _images(120, 200, 5)
    x = gt.to(dev)
    l1, ss = l1_ssim(x, x)
    assert float(l1) == 0.0 and abs(float(ss) - 1.0) < 1e-6  # SSIM(x, x) = 1
    a = l1_ssim(pred.to(dev), x, mask.to(dev))
    b = l1_ssim(pred.to(dev), x, mask.to(dev))
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])  # fixed reduction order: bit-reproducible
    with torch.no_grad():  # no gradient needed: no derivative maps are kept
        c = l1_ssim(pred.to(dev).requires_grad_(True), x, mask.to(dev))
    assert torch.equal(c[1], a[1]) and not c[1].requires_grad
    with pytest.raises(AssertionError):
        l1_ssim(pred.to(dev)[:10], x[:10])


def _reference_pixel_losses(opacity, depth, sky_masks, valid, lidar, max_depth=80.0):
    """base.py:536-573 + models/losses.py (binary_cross_entropy "mean"; DepthLoss l1 / no normalisation / mean_on_hit)."""
    gt_occupied_mask = (1.0 - sky_masks) * valid
    pred_occupied_mask = opacity.squeeze() * valid
    bce = torch.nn.functional.binary_cross_entropy(pred_occupied_mask, gt_occupied_mask, reduction="none").mean()
    lidar_hit_mask = (lidar > 0).to(valid.dtype) * valid
    pd, gd = depth.squeeze() * lidar_hit_mask, lidar.squeeze() * lidar_hit_mask
    vm = (gd > 0.01) & (gd < max_depth) & (pd > 0.0001)
    dl = torch.nn.functional.l1_loss(pd[vm], gd[vm], reduction="none").mean()
    po = torch.clamp(opacity.squeeze(), 1e-6, 1 - 1e-6)
    ent = (-po * torch.log(po)).mean()
    return bce, dl, ent, int(vm.sum())


@pytest.mark.parametrize("H,W", [(37, 53), (375, 1242)])
def test_pixel_losses_match_reference_lines(gpu_device, H, W):
    from gsplat.losses import pixel_losses

    dev = gpu_device
    g = torch.Generator().manual_seed(H)
    opacity = torch.rand(H, W, 1, generator=g)
    opacity[0, :5] = 0.0       # exact 0 / 1: the logs clamp at -100
    opacity[1, :5] = 1.0
    depth = torch.rand(H, W, 1, generator=g) * 90.0
    depth[2, :7] = 0.0
    sky = (torch.rand(H, W, generator=g) > 0.7).float()
    valid = (torch.rand(H, W, generator=g) > 0.1).float()
    lidar = torch.rand(H, W, generator=g) * 100.0 * (torch.rand(H, W, generator=g) > 0.8).float()  # sparse hits, some > 80
    od, dd = opacity.double().requires_grad_(True), depth.double().requires_grad_(True)
    bce_r, dl_r, ent_r, cnt = _reference_pixel_losses(od, dd, sky.double(), valid.double(), lidar.double())
    (0.05 * bce_r + 0.01 * dl_r + 0.3 * ent_r).backward()
    og, dg = opacity.to(dev).requires_grad_(True), depth.to(dev).requires_grad_(True)
    res = pixel_losses(og, dg, sky.to(dev), valid.to(dev), lidar.to(dev))
    assert cnt > 10
    for name, ref in (("sky_bce", bce_r), ("depth_l1", dl_r), ("opacity_entropy", ent_r)):
        got = float(res[name].detach())
        assert abs(got - float(ref.detach())) < 1e-5 * max(1.0, abs(float(ref.detach()))), (name, got, float(ref.detach()))
    (0.05 * res["sky_bce"] + 0.01 * res["depth_l1"] + 0.3 * res["opacity_entropy"]).backward()
    for got, ref in ((og.grad, od.grad), (dg.grad, dd.grad)):
        err = float((got.cpu().double() - ref).abs().max())
        assert err < 1e-5 * float(ref.abs().max()) + 1e-12, (err, float(ref.abs().max()))
    # terms without inputs are absent; an empty selection gives NaN like torch's mean of nothing
    only = pixel_losses(opacity.to(dev))
    assert set(only) == {"opacity_entropy"}
    none_hit = pixel_losses(opacity.to(dev), depth.to(dev), None, None, torch.zeros(H, W, device=dev))
    assert torch.isnan(none_hit["depth_l1"])


def test_losses_accept_other_float_dtypes(gpu_device):
    """Inputs are computed in fp32 whatever they arrive as; gradients come back in the input's dtype."""
    from gsplat.envlight import cube_texture
    from gsplat.losses import l1_ssim, pixel_losses

    dev = gpu_device
    pred, gt, mask = _images(40, 50, 3)
    p64 = pred.double().to(dev).requires_grad_(True)
    l1, ss = l1_ssim(p64, gt.to(dev), mask.to(dev))
    (l1 + ss).backward()
    assert p64.grad.dtype == torch.float64 and torch.isfinite(p64.grad).all()
    op = torch.rand(40, 50, 1, dtype=torch.float64, device=dev).requires_grad_(True)
    res = pixel_losses(op, None, (torch.rand(40, 50, device=dev) > 0.5).float())
    (res["sky_bce"] + res["opacity_entropy"]).backward()
    assert op.grad.dtype == torch.float64 and op.grad.shape == op.shape
    tex = torch.rand(6, 8, 8, 3, dtype=torch.float64, device=dev).requires_grad_(True)
    cube_texture(tex, torch.randn(100, 3, device=dev)).sum().backward()
    assert tex.grad.dtype == torch.float64


# --------------------------------------------------------------------------------- every configurable variant
def _maps(H, W, seed):
    g = torch.Generator().manual_seed(seed)
    opacity = torch.rand(H, W, 1, generator=g).clamp(1e-4, 1 - 1e-4)
    opacity[0, :5] = 0.0   # exactly 0 / 1: the clipped ends of both BCE forms
    opacity[1, :5] = 1.0
    depth = torch.rand(H, W, 1, generator=g) * 90 + 0.05
    sky = (torch.rand(H, W, generator=g) > 0.7).float()
    valid = (torch.rand(H, W, generator=g) > 0.1).float()
    lidar = torch.where(torch.rand(H, W, generator=g) > 0.6, torch.rand(H, W, generator=g) * 100, torch.zeros(H, W))
    lidar[2, :7] = depth[2, :7, 0]  # exact hits: |d| = 0 rows of every loss type
    return opacity, depth, sky, valid, lidar


@pytest.mark.parametrize("kind", ["bce", "safe_bce"])
@pytest.mark.parametrize("loss_type,normalize,inverse,reduction,percentile", [
    ("l1", False, False, "mean_on_hit", None),
    ("l2", False, False, "mean_on_hw", None),
    ("smooth_l1", False, False, "sum", None),
    ("l1", True, False, "mean_on_hit", None),
    ("l2", True, True, "mean_on_hit", None),
    ("smooth_l1", True, True, "mean_on_hw", None),
    ("l1", False, False, "mean_on_hit", 0.95),
    ("l2", True, False, "sum", 0.5),
])
def test_pixel_loss_variants_match_reference_lines(gpu_device, kind, loss_type, normalize, inverse, reduction, percentile):
    """safe_bce and every DepthLoss configuration against oracle/loss_ref.py (the reference's lines, fp64 autograd)."""
    from gsplat.losses import pixel_losses
    from oracle import loss_ref as R

    dev = gpu_device
    H, W = 61, 83
    opacity, depth, sky, valid, lidar = _maps(H, W, 11)
    od, dd = opacity.double().requires_grad_(True), depth.double().requires_grad_(True)
    ref_m = R.mask_loss(od, sky.double(), valid.double(), kind)
    ref_d = R.depth_loss(dd, lidar.double(), valid.double(), loss_type, normalize, inverse, percentile, 80.0, reduction)
    (0.7 * ref_m + 1.3 * ref_d).backward()

    og, dg = opacity.to(dev).requires_grad_(True), depth.to(dev).requires_grad_(True)
    out = pixel_losses(og, dg, sky.to(dev), valid.to(dev), lidar.to(dev), 80.0, opacity_loss_type=kind,
                       depth_loss_type=loss_type, normalize=normalize, inverse_depth=inverse, reduction=reduction,
                       depth_error_percentile=percentile)
    (0.7 * out["sky_bce"] + 1.3 * out["depth_l1"] + 0.0 * out["opacity_entropy"]).backward()
    tol = lambda r: 2e-5 * max(1.0, abs(float(r)))
    assert abs(float(out["sky_bce"]) - float(ref_m)) <= tol(ref_m), (float(out["sky_bce"]), float(ref_m))
    assert abs(float(out["depth_l1"]) - float(ref_d)) <= 5 * tol(ref_d), (float(out["depth_l1"]), float(ref_d))
    for name, got, ref in (("v_opacity", og.grad, od.grad), ("v_depth", dg.grad, dd.grad)):
        err = float((got.cpu().double() - ref).abs().max())
        assert err <= 1e-4 * float(ref.abs().max()) + 1e-12, (name, err, float(ref.abs().max()))


def test_depth_error_map_is_reduction_none(gpu_device):
    from gsplat.losses import depth_error_map
    from oracle import loss_ref as R

    H, W = 40, 50
    opacity, depth, sky, valid, lidar = _maps(H, W, 3)
    m = depth_error_map(depth.to(gpu_device), lidar.to(gpu_device), valid.to(gpu_device), 80.0, "l2", True, False)
    ref = R.depth_loss(depth.double(), lidar.double(), valid.double(), "l2", True, False, None, 80.0, "none")
    got = m.cpu().reshape(-1)
    got = got[got >= 0].double()
    assert got.numel() == ref.numel() and torch.allclose(got, ref, rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("H,W", [(37, 52), (2, 2), (120, 200)])
def test_inverse_depth_smoothness(gpu_device, H, W):
    """kornia's formula (parity unpinned: oracle/loss_ref.py restates it) and its gradient w.r.t. the depth."""
    from gsplat.losses import inverse_depth_smoothness
    from oracle import loss_ref as R

    g = torch.Generator().manual_seed(H)
    depth = torch.rand(H, W, 1, generator=g) * 40 + 0.5
    image = torch.rand(H, W, 3, generator=g)
    dd = depth.double().requires_grad_(True)
    ref = R.inverse_depth_smoothness(dd, image.double())
    ref.backward()
    dg = depth.to(gpu_device).requires_grad_(True)
    out = inverse_depth_smoothness(dg, image.to(gpu_device))
    out.backward()
    assert abs(float(out) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))
    err = float((dg.grad.cpu().double() - dd.grad).abs().max())
    assert err <= 1e-4 * float(dd.grad.abs().max()) + 1e-12


def test_dynamic_region_and_affine_terms(gpu_device):
    from gsplat.losses import affine_regulariser, dynamic_region_l1
    from oracle import loss_ref as R

    dev = gpu_device
    g = torch.Generator().manual_seed(8)
    H, W = 45, 70
    rgb, pix = torch.rand(H, W, 3, generator=g), torch.rand(H, W, 3, generator=g)
    valid = (torch.rand(H, W, generator=g) > 0.2).float()
    dyn = torch.rand(H, W, 1, generator=g)
    rd = rgb.double().requires_grad_(True)
    ref = R.dynamic_region_loss(rd, pix.double(), valid.double(), dyn.double())
    ref.backward()
    rg = rgb.to(dev).requires_grad_(True)
    loss, cnt = dynamic_region_l1(rg, pix.to(dev), dyn.to(dev), valid.to(dev))
    loss.backward()
    assert int(cnt) == int(((dyn[..., 0] > 0.2) & valid.bool()).sum())
    assert abs(float(loss) - float(ref)) <= 2e-6
    assert float((rg.grad.cpu().double() - rd.grad).abs().max()) <= 1e-4 * float(rd.grad.abs().max())
    # empty region: an exact zero, zero gradient (the reference adds no term)
    rg2 = rgb.to(dev).requires_grad_(True)
    loss2, cnt2 = dynamic_region_l1(rg2, pix.to(dev), torch.zeros(H, W, 1, device=dev), valid.to(dev))
    loss2.backward()
    assert float(loss2) == 0.0 and int(cnt2) == 0 and float(rg2.grad.abs().max()) == 0.0

    for shape in [(3, 4), (1, 4, 4), (5, 3, 4)]:
        A = torch.randn(*shape, generator=g) * 0.3
        if shape[-2] >= 3:
            A[..., :3, :3] += torch.eye(3)
        A[..., 0, 0] = 1.0  # an exact match: sign(0) = 0
        ad = A.double().requires_grad_(True)
        ref = R.affine_loss(ad)
        ref.backward()
        ag = A.to(dev).requires_grad_(True)
        out = affine_regulariser(ag)
        out.backward()
        assert abs(float(out) - float(ref)) <= 2e-6, shape
        assert torch.allclose(ag.grad.cpu().double(), ad.grad, atol=1e-6), shape


def test_opacity_reset_in_place(gpu_device):
    """raw = logit(min(sigmoid(raw), reset)) and zeroed Adam moments (vanilla.py:268-280)."""
    from gsplat.densify import reset_opacities
    from oracle import loss_ref as R

    dev = gpu_device
    g = torch.Generator().manual_seed(1)
    raw = torch.randn(100_003, 1, generator=g) * 3
    ref = R.opacity_reset(raw.double(), 0.01)
    p = raw.to(dev)
    state = {"exp_avg": torch.randn(100_003, 1, device=dev), "exp_avg_sq": torch.rand(100_003, 1, device=dev),
             "step": torch.tensor(7.0)}
    reset_opacities(p, 0.01, state)
    assert torch.allclose(p.cpu().double(), ref, rtol=2e-6, atol=2e-6)
    assert float(state["exp_avg"].abs().max()) == 0.0 and float(state["exp_avg_sq"].abs().max()) == 0.0
    assert float(torch.sigmoid(p).max()) <= 0.01 * (1 + 1e-5)
    untouched = raw.to(dev)
    reset_opacities(untouched, 0.9999)  # above these opacities: unchanged up to the fp32 logit(sigmoid(.)) round trip
    mid = raw.abs() < 6
    assert torch.allclose(untouched.cpu()[mid], raw[mid], rtol=1e-3, atol=1e-3)


# ------------------------------------------------- against the REFERENCE's own models/losses.py (committed fixture)
def _ref_fixture(dev):
    import os

    import numpy as np

    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "losses_ref.npz"))
    f = lambda k: torch.from_numpy(z[k]).float().to(dev)
    return z, f


@pytest.mark.parametrize("name,op_key,kind", [("bce", "opacity", "bce"), ("safe_bce", "opacity", "safe_bce"),
                                              ("safe_bce_wild", "opacity_wild", "safe_bce")])
def test_mask_loss_matches_reference_fixture(gpu_device, name, op_key, kind):
    """The HIP mask term against values and gradients the reference's binary_cross_entropy / safe_binary_cross_entropy
    (models/losses.py:33-83) produced in fp64 on the same maps (tests/golden/make_golden.py::losses_golden)."""
    from gsplat.losses import pixel_losses

    z, f = _ref_fixture(gpu_device)
    og = f(op_key).requires_grad_(True)
    out = pixel_losses(og, None, f("sky"), f("valid"), None, opacity_loss_type=kind)
    out["sky_bce"].backward()
    ref, ref_g = float(z[f"mask_{name}"]), torch.from_numpy(z[f"mask_{name}_grad"])
    assert abs(float(out["sky_bce"]) - ref) <= 2e-5 * max(1.0, abs(ref)), (float(out["sky_bce"]), ref)
    err = float((og.grad.cpu().double() - ref_g).abs().max())
    assert err <= 1e-4 * float(ref_g.abs().max()) + 1e-12, (err, float(ref_g.abs().max()))


def test_depth_loss_matches_reference_fixture(gpu_device):
    """Every DepthLoss setting of the reference (models/losses.py:91-176: 3 types x normalize x inverse x 4 reductions,
    two percentiles, an empty hit set) against the reference's own fp64 outputs."""
    import numpy as np

    from gsplat.losses import depth_error_map, pixel_losses
    from util import DEPTH_CASES, PERCENTILE_CASES, depth_case_key

    dev = gpu_device
    z, f = _ref_fixture(dev)
    md = float(z["max_depth"])
    n_checked = 0
    for (ty, n, i, r, pc) in [c + (None,) for c in DEPTH_CASES] + PERCENTILE_CASES:
        k = depth_case_key(ty, n, i, r, pc)
        ref = np.asarray(z[k])
        if r == "none":
            m = depth_error_map(f("depth"), f("lidar"), f("valid"), md, ty, n, i).cpu().reshape(-1)
            got = m[m >= 0].double()
            assert got.numel() == ref.size, k
            scale = float(np.abs(ref).max())
            assert float((got - torch.from_numpy(ref)).abs().max()) <= 2e-5 * max(1.0, scale), k
            continue
        dg = f("depth").requires_grad_(True)
        out = pixel_losses(f("opacity"), dg, None, f("valid"), f("lidar"), md, depth_loss_type=ty, normalize=n,
                           inverse_depth=i, reduction=r, depth_error_percentile=pc)
        out["depth_l1"].backward()
        assert abs(float(out["depth_l1"]) - float(ref)) <= 1e-4 * max(1.0, abs(float(ref))), (k, float(out["depth_l1"]), float(ref))
        ref_g = torch.from_numpy(z[k + "_grad"])
        err = float((dg.grad.cpu().double() - ref_g).abs().max())
        assert err <= 1e-4 * float(ref_g.abs().max()) + 1e-12, (k, err, float(ref_g.abs().max()))
        n_checked += 1
    assert n_checked == 38
    for r in ("mean_on_hit", "mean_on_hw", "sum"):
        out = pixel_losses(f("opacity"), f("depth"), None, f("valid"), torch.zeros_like(f("lidar")), md, reduction=r)
        ref = float(z[f"depth_empty_{r}"])
        assert (np.isnan(ref) and bool(torch.isnan(out["depth_l1"]))) or float(out["depth_l1"]) == ref, r
