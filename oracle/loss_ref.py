"""TEST INFRASTRUCTURE -- not part of the product.  CPU (PyTorch autograd, fp64-capable) restatement of the loss terms
the reference computes on the rasterizer's output, in every configuration it can be given.  R =
/root/reference/3D-Reconstruction-main.  Only tests/ may import this.

PARITY: SafeBCE, DepthLoss, the affine / dynamic-region / opacity-reset lines follow the reference's own source line by
line (cited per function).  `inverse_depth_smoothness` restates kornia.losses.inverse_depth_smoothness_loss (kornia is
in R/requirements.txt but neither in /root/reference nor installed here): PARITY UNPINNED for that term."""
import math

import torch
import torch.nn.functional as F


class _SafeBCE(torch.autograd.Function):
    """R/models/losses.py:33-77 (SafeBCE)."""

    @staticmethod
    def forward(ctx, x, y, limit):
        ctx.ln_limit = math.log(limit)
        x = torch.clip(x, 0, 1)
        y = torch.clip(y, 0, 1)
        ctx.save_for_backward(x, y)
        return -torch.where(y == 0, torch.log(1 - x).clamp_min(ctx.ln_limit), torch.log(x).clamp_min(ctx.ln_limit))

    @staticmethod
    def backward(ctx, grad_output):
        x, y = ctx.saved_tensors
        limit = math.exp(ctx.ln_limit)
        x = torch.where(y == 0, torch.clip(x, 0, 1 - limit), torch.clip(x, limit, 1))
        grad_x = torch.where(y == 0, 1 / (1 - x), -1 / x) * grad_output * (~(x == y))
        return grad_x, None, None


def mask_loss(opacity, sky_mask, valid_mask, kind="bce"):
    """R/models/trainers/base.py:536-550 with the loss function chosen at base.py:231-239."""
    gt = (1.0 - sky_mask) * valid_mask
    pred = opacity.reshape(sky_mask.shape) * valid_mask
    if kind == "safe_bce":
        return _SafeBCE.apply(pred, gt, 0.1).mean()  # R/models/losses.py:79-81, reduce(..., "mean")
    return F.binary_cross_entropy(pred, gt, reduction="none").mean()  # losses.py:83-85


def depth_loss(pred_depth, gt_depth, valid_mask, loss_type="l1", normalize=False, inverse=False, percentile=None,
               max_depth=80.0, reduction="mean_on_hit"):
    """R/models/losses.py:91-199 (DepthLoss) as called at R/models/trainers/base.py:553-557."""
    hit = (gt_depth > 0).to(pred_depth.dtype) * valid_mask
    p = pred_depth.squeeze() * hit
    g = gt_depth.squeeze() * hit
    sel = (g > 0.01) & (g < max_depth) & (p > 0.0001)
    if normalize:
        p = torch.clamp(p[sel] / max_depth, 1e-06, 1.0)
        g = torch.clamp(g[sel] / max_depth, 1e-06, 1.0)
    else:
        p, g = p[sel], g[sel]
    if inverse:
        p, g = 1.0 / p, 1.0 / g
    if loss_type == "smooth_l1":
        err = F.smooth_l1_loss(p, g, reduction="none")
    elif loss_type == "l1":
        err = F.l1_loss(p, g, reduction="none")
    else:
        err = F.mse_loss(p, g, reduction="none")
    if percentile is not None:
        err = err.flatten()
        err = err[err.argsort(stable=True)[: int(len(err) * percentile)]]
    if reduction == "sum":
        return err.sum()
    if reduction == "none":
        return err
    if reduction == "mean_on_hit":
        return err.mean()
    return err.sum() / (gt_depth.shape[0] * gt_depth.shape[1])  # mean_on_hw


def inverse_depth_smoothness(depth, image):
    """R/models/trainers/base.py:575-585: kornia.losses.inverse_depth_smoothness_loss(idepth [1,3,H,W], image [1,3,H,W])
    = mean |d_x idepth| exp(-mean_c |d_x image|) + mean |d_y idepth| exp(-mean_c |d_y image|), d_x f = f[..., :-1] - f[..., 1:]."""
    idepth = (1 / (depth + 1e-5))[None].repeat(1, 1, 1, 3).permute(0, 3, 1, 2)
    img = image[None].permute(0, 3, 1, 2)
    gx = lambda t: t[:, :, :, :-1] - t[:, :, :, 1:]
    gy = lambda t: t[:, :, :-1, :] - t[:, :, 1:, :]
    wx = torch.exp(-torch.mean(torch.abs(gx(img)), dim=1, keepdim=True))
    wy = torch.exp(-torch.mean(torch.abs(gy(img)), dim=1, keepdim=True))
    return torch.mean(torch.abs(gx(idepth) * wx)) + torch.mean(torch.abs(gy(idepth) * wy))


def affine_loss(affine_trs):
    """R/models/trainers/base.py:587-596."""
    reg_mat = torch.eye(3, dtype=affine_trs.dtype)
    reg_shift = torch.zeros(3, dtype=affine_trs.dtype)
    return torch.abs(affine_trs[..., :3, :3] - reg_mat).mean() + torch.abs(affine_trs[..., :3, 3:] - reg_shift[:, None]).mean()


def dynamic_region_loss(rgb, pixels, valid_mask, dynamic_opacity):
    """R/models/trainers/base.py:533-534, :598-613.  Returns None when the region is empty (the reference adds no term)."""
    gt_rgb = pixels * valid_mask[..., None]
    predicted_rgb = rgb * valid_mask[..., None]
    m = (dynamic_opacity.detach() > 0.2).squeeze() & valid_mask.bool()
    if m.sum() > 0:
        return torch.abs(gt_rgb[m] - predicted_rgb[m]).mean()
    return None


def opacity_reset(raw_opacities, reset_alpha_value):
    """R/models/gaussians/vanilla.py:268-273."""
    reset_value = torch.min(torch.sigmoid(raw_opacities), torch.ones_like(raw_opacities) * reset_alpha_value)
    return torch.logit(reset_value)
