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
