"""RGB loss terms on the rasterizer's output in two kernels.

`l1_ssim(pred, gt, mask)` returns what `BasicTrainer.compute_losses` computes at
/root/reference/3D-Reconstruction-main/models/trainers/base.py:533-541 with torch ops and
`pytorch_msssim.SSIM(data_range=1.0, size_average=True, channel=3)` (base.py:114): the mean absolute difference and the
SSIM of the masked images, differentiable w.r.t. `pred`:

    Ll1, ssim = l1_ssim(outputs["rgb"], image_infos["pixels"], valid_loss_mask)
    loss = w_rgb * Ll1 + w_ssim * (1 - ssim)
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _backend as B


class _L1Ssim(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt, mask, need):
        dev = B.require_gpu(pred, gt, mask)
        H, W = pred.shape[0], pred.shape[1]
        pred_c, gt_c = B.f32c(pred), B.f32c(gt)
        mask_c = B.f32c(mask) if mask is not None else None
        L = B.lib()
        nws = int(L.gs_l1_ssim_workspace_floats(H, W))
        ws = torch.empty(nws, device=dev, dtype=torch.float32)
        maps = torch.empty((H - 10) * (W - 10) * 9, device=dev, dtype=torch.float32) if need else None
        out = torch.empty(2, device=dev, dtype=torch.float32)
        with B.device_guard(dev), B.timed("gs_l1_ssim_fwd", dev):
            B.check(L.gs_l1_ssim_fwd(H, W, B.ptr(pred_c), B.ptr(gt_c), B.ptr(mask_c), B.ptr(maps), B.ptr(ws), nws,
                                     B.ptr(out), B.stream_handle(dev)), "gs_l1_ssim_fwd")
        ctx.has_mask = mask is not None
        ctx.in_dtype = pred.dtype
        if need:
            ctx.save_for_backward(pred_c, gt_c, mask_c if mask_c is not None else out, maps)
        return out

    @staticmethod
    def backward(ctx, g_out):
        pred, gt, mask, maps = ctx.saved_tensors
        mask = mask if ctx.has_mask else None
        dev = pred.device
        H, W = pred.shape[0], pred.shape[1]
        g = B.f32c(g_out)
        v = torch.empty_like(pred)
        with B.device_guard(dev), B.timed("gs_l1_ssim_bwd", dev):
            B.check(B.lib().gs_l1_ssim_bwd(H, W, B.ptr(pred), B.ptr(gt), B.ptr(mask), B.ptr(maps), B.ptr(g), B.ptr(v),
                                           B.stream_handle(dev)), "gs_l1_ssim_bwd")
        return v.to(ctx.in_dtype), None, None, None


def l1_ssim(pred: Tensor, gt: Tensor, mask: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """pred, gt: [H,W,3] in [0,1]; mask: [H,W] (multiplied into both images, base.py:533-534) or None.
    Returns (Ll1, ssim) as 0-d tensors; `1 - ssim` is the reference's `simloss`.  H, W must exceed 10."""
    assert pred.dim() == 3 and pred.shape[-1] == 3 and gt.shape == pred.shape, (pred.shape, gt.shape)
    assert mask is None or tuple(mask.shape) == tuple(pred.shape[:2]), "mask must be [H,W]"
    assert pred.shape[0] > 10 and pred.shape[1] > 10, "images must be larger than the 11-tap window"
    out = _L1Ssim.apply(pred, gt, mask, bool(pred.requires_grad and torch.is_grad_enabled()))
    return out[0], out[1]


class _PixelLosses(torch.autograd.Function):
    @staticmethod
    def forward(ctx, opacity, depth, sky_mask, valid_mask, lidar_depth, max_depth):
        dev = B.require_gpu(opacity, depth, sky_mask, valid_mask, lidar_depth)
        n = opacity.numel()
        c = [B.f32c(t).reshape(-1) if t is not None else None for t in (opacity, depth, sky_mask, valid_mask, lidar_depth)]
        L = B.lib()
        nws = int(L.gs_pixel_losses_workspace_floats())
        ws = torch.empty(nws, device=dev, dtype=torch.float32)
        out = torch.empty(4, device=dev, dtype=torch.float32)
        with B.device_guard(dev), B.timed("gs_pixel_losses_fwd", dev):
            B.check(L.gs_pixel_losses_fwd(n, B.ptr(c[0]), B.ptr(c[1]), B.ptr(c[2]), B.ptr(c[3]), B.ptr(c[4]),
                                          float(max_depth), B.ptr(ws), nws, B.ptr(out), B.stream_handle(dev)),
                    "gs_pixel_losses_fwd")
        ctx.present = [t is not None for t in c]
        ctx.max_depth = float(max_depth)
        ctx.shapes = (tuple(opacity.shape), tuple(depth.shape) if depth is not None else None)
        ctx.dtypes = (opacity.dtype, depth.dtype if depth is not None else None)
        ctx.save_for_backward(out, *[t for t in c if t is not None])
        return out

    @staticmethod
    def backward(ctx, g_out):
        out, *rest = ctx.saved_tensors
        it = iter(rest)
        c = [next(it) if pr else None for pr in ctx.present]
        dev = out.device
        n = c[0].numel()
        g = B.f32c(g_out)
        need_depth = c[1] is not None and ctx.needs_input_grad[1]
        v_op = torch.empty(n, device=dev, dtype=torch.float32)
        v_dp = torch.empty(n, device=dev, dtype=torch.float32) if need_depth else None
        with B.device_guard(dev), B.timed("gs_pixel_losses_bwd", dev):
            B.check(B.lib().gs_pixel_losses_bwd(n, B.ptr(c[0]), B.ptr(c[1]), B.ptr(c[2]), B.ptr(c[3]), B.ptr(c[4]),
                                                ctx.max_depth, B.ptr(out), B.ptr(g), B.ptr(v_op), B.ptr(v_dp),
                                                B.stream_handle(dev)), "gs_pixel_losses_bwd")
        return (v_op.reshape(ctx.shapes[0]).to(ctx.dtypes[0]),
                v_dp.reshape(ctx.shapes[1]).to(ctx.dtypes[1]) if need_depth else None, None, None, None, None)


def pixel_losses(opacity: Tensor, depth: Optional[Tensor] = None, sky_mask: Optional[Tensor] = None,
                 valid_mask: Optional[Tensor] = None, lidar_depth: Optional[Tensor] = None, max_depth: float = 80.0):
    """The per-pixel scalar terms of `compute_losses` in the reference's shipped configuration
    (configs/omnire.yaml:24-31), in one pass (base.py:536-573, models/losses.py):

        sky_bce          = binary_cross_entropy(opacity * valid, (1 - sky_mask) * valid, "mean")       (mask loss, "bce")
        depth_l1         = DepthLoss("l1", normalize=False, use_inverse_depth=False)(depth, lidar_depth,
                                                                                    (lidar_depth > 0) * valid)
        opacity_entropy  = (-p * log p).mean(), p = clamp(opacity, 1e-6, 1 - 1e-6)

    opacity [H,W] or [H,W,1]; depth [H,W,1]; masks / lidar_depth [H,W].  Returns a dict of 0-d tensors (terms whose
    inputs are None are absent), differentiable w.r.t. opacity and depth.  The other variants the reference can be
    configured with (safe_bce, l2 / smooth_l1, normalised or inverse depth, percentile clipping) are not built."""
    n = opacity.numel()
    for t in (depth, sky_mask, valid_mask, lidar_depth):
        assert t is None or t.numel() == n, "all maps must have H*W elements"
    assert lidar_depth is None or depth is not None
    out = _PixelLosses.apply(opacity, depth, sky_mask, valid_mask, lidar_depth, max_depth)
    res = {"opacity_entropy": out[3]}
    if sky_mask is not None:
        res["sky_bce"] = out[0]
    if lidar_depth is not None:
        res["depth_l1"] = out[1]
    return res
