"""CPU restatement of the reference's RGB loss terms -- TEST INFRASTRUCTURE ONLY (imported by tests/ alone).

`BasicTrainer.compute_losses` (/root/reference/3D-Reconstruction-main/models/trainers/base.py:533-541) uses
`pytorch_msssim.SSIM(data_range=1.0, size_average=True, channel=3)` (base.py:114).  pytorch_msssim is pinned at 1.0.0
in the reference's requirements.txt:7 but is neither under /root/reference nor installed here, so its published
algorithm is restated below (`_fspecial_gauss_1d`, `gaussian_filter`, `_ssim`, `ssim` of that release): **parity
unpinned** -- anchored on the call site and on closed-form properties (SSIM(x, x) = 1, window sums to 1, symmetry),
not on vectors produced by the package itself.
"""
import torch
import torch.nn.functional as F


def fspecial_gauss_1d(size: int = 11, sigma: float = 1.5, dtype=torch.float64) -> torch.Tensor:
    coords = torch.arange(size, dtype=dtype)
    coords -= size // 2
    g = torch.exp(-(coords ** 2) / (2 * sigma ** 2))
    g /= g.sum()
    return g


def gaussian_filter(x: torch.Tensor, win: torch.Tensor) -> torch.Tensor:
    """x [B,C,H,W]; separable VALID (padding 0) depthwise filter along H then W."""
    C = x.shape[1]
    w = win.to(dtype=x.dtype, device=x.device).reshape(1, 1, 1, -1).repeat(C, 1, 1, 1)
    out = F.conv2d(x, w.transpose(2, 3), stride=1, padding=0, groups=C)
    out = F.conv2d(out, w, stride=1, padding=0, groups=C)
    return out


def ssim(X: torch.Tensor, Y: torch.Tensor, data_range: float = 1.0, K=(0.01, 0.03)) -> torch.Tensor:
    """pytorch_msssim.ssim(X, Y, data_range, size_average=True, win_size=11, win_sigma=1.5, nonnegative_ssim=False)."""
    win = fspecial_gauss_1d(11, 1.5, dtype=X.dtype)
    C1, C2 = (K[0] * data_range) ** 2, (K[1] * data_range) ** 2
    mu1, mu2 = gaussian_filter(X, win), gaussian_filter(Y, win)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = gaussian_filter(X * X, win) - mu1_sq
    sigma2_sq = gaussian_filter(Y * Y, win) - mu2_sq
    sigma12 = gaussian_filter(X * Y, win) - mu1_mu2
    cs_map = (2 * sigma12 + C2) / (sigma1_sq + sigma2_sq + C2)
    ssim_map = ((2 * mu1_mu2 + C1) / (mu1_sq + mu2_sq + C1)) * cs_map
    return torch.flatten(ssim_map, 2).mean(-1).mean()


def rgb_loss_terms(pred: torch.Tensor, gt: torch.Tensor, mask=None):
    """base.py:533-541 on [H,W,3] images: (Ll1, ssim) -- the reference's simloss is 1 - ssim."""
    if mask is not None:
        gt = gt * mask[..., None]
        pred = pred * mask[..., None]
    Ll1 = torch.abs(gt - pred).mean()
    s = ssim(gt.permute(2, 0, 1)[None, ...], pred.permute(2, 0, 1)[None, ...])
    return Ll1, s
