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
