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


_BCE = {"bce": 0, "safe_bce": 1}
_DEPTH_TYPE = {"l1": 0, "l2": 1, "smooth_l1": 2}
_DEPTH_RED = {"mean_on_hit": 0, "mean_on_hw": 1, "sum": 2}


def _opts(opacity_loss_type, depth_loss_type, normalize, inverse_depth, reduction, max_depth) -> B.PixelLossOpts:
    o = B.PixelLossOpts()
    o.bce_mode, o.bce_limit = _BCE[opacity_loss_type], 0.1  # limit: base.py:238
    o.depth_type, o.depth_normalize, o.depth_inverse = _DEPTH_TYPE[depth_loss_type], int(normalize), int(inverse_depth)
    o.depth_reduction, o.max_depth = _DEPTH_RED[reduction], float(max_depth)
    return o


class _PixelLosses(torch.autograd.Function):
    @staticmethod
    def forward(ctx, opacity, depth, sky_mask, valid_mask, lidar_depth, opts, select, want_err_map):
        dev = B.require_gpu(opacity, depth, sky_mask, valid_mask, lidar_depth)
        n = opacity.numel()
        c = [B.f32c(t).reshape(-1) if t is not None else None for t in (opacity, depth, sky_mask, valid_mask, lidar_depth)]
        sel = select.reshape(-1).to(torch.uint8).contiguous() if select is not None else None
        L = B.lib()
        nws = int(L.gs_pixel_losses_workspace_floats())
        ws = torch.empty(nws, device=dev, dtype=torch.float32)
        out = torch.empty(4, device=dev, dtype=torch.float32)
        err_map = torch.empty(n, device=dev, dtype=torch.float32) if want_err_map else None
        import ctypes

        with B.device_guard(dev), B.timed("gs_pixel_losses_fwd", dev):
            B.check(L.gs_pixel_losses_ex_fwd(n, ctypes.byref(opts), B.ptr(c[0]), B.ptr(c[1]), B.ptr(c[2]), B.ptr(c[3]),
                                             B.ptr(c[4]), B.ptr(sel), B.ptr(err_map), B.ptr(ws), nws, B.ptr(out),
                                             B.stream_handle(dev)), "gs_pixel_losses_ex_fwd")
        ctx.present = [t is not None for t in c]
        ctx.opts, ctx.sel = opts, sel
        ctx.shapes = (tuple(opacity.shape), tuple(depth.shape) if depth is not None else None)
        ctx.dtypes = (opacity.dtype, depth.dtype if depth is not None else None)
        ctx.save_for_backward(out, *[t for t in c if t is not None])
        if want_err_map:
            ctx.mark_non_differentiable(err_map)
            return out, err_map
        return out, None

    @staticmethod
    def backward(ctx, g_out, _g_err):
        import ctypes

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
            B.check(B.lib().gs_pixel_losses_ex_bwd(n, ctypes.byref(ctx.opts), B.ptr(c[0]), B.ptr(c[1]), B.ptr(c[2]),
                                                   B.ptr(c[3]), B.ptr(c[4]), B.ptr(ctx.sel), B.ptr(out), B.ptr(g),
                                                   B.ptr(v_op), B.ptr(v_dp), B.stream_handle(dev)),
                    "gs_pixel_losses_ex_bwd")
        return (v_op.reshape(ctx.shapes[0]).to(ctx.dtypes[0]),
                v_dp.reshape(ctx.shapes[1]).to(ctx.dtypes[1]) if need_depth else None, None, None, None, None, None,
                None)


def pixel_losses(opacity: Tensor, depth: Optional[Tensor] = None, sky_mask: Optional[Tensor] = None,
                 valid_mask: Optional[Tensor] = None, lidar_depth: Optional[Tensor] = None, max_depth: float = 80.0,
                 opacity_loss_type: str = "bce", depth_loss_type: str = "l1", normalize: bool = False,
                 inverse_depth: bool = False, reduction: str = "mean_on_hit",
                 depth_error_percentile: Optional[float] = None):
    """The per-pixel scalar terms of `compute_losses` (base.py:536-573, models/losses.py), in one pass:

        sky_bce          = binary_cross_entropy | safe_binary_cross_entropy(limit=0.1)
                           (opacity * valid, (1 - sky_mask) * valid, "mean")            (mask loss: base.py:231-239)
        depth_l1         = DepthLoss(depth_loss_type, normalize, inverse_depth, depth_error_percentile, max_depth,
                                     reduction)(depth, lidar_depth, (lidar_depth > 0) * valid)   (base.py:241-250)
        opacity_entropy  = (-p * log p).mean(), p = clamp(opacity, 1e-6, 1 - 1e-6)

    The defaults are the shipped configuration (configs/omnire.yaml:24-31: bce, l1, not normalised, mean_on_hit); the
    keyword names are the reference's config keys.  `depth_error_percentile` (losses.py:163-171, "not used for now" there)
    keeps the smallest `int(n_selected * p)` depth errors: one extra pass writes the error map, torch ranks it (stable,
    as the reference's argsort), and the keep-mask goes back into the kernel.
    opacity [H,W] or [H,W,1]; depth [H,W,1]; masks / lidar_depth [H,W].  Returns a dict of 0-d tensors (terms whose
    inputs are None are absent; the depth term keeps the key "depth_l1" whatever its type), differentiable w.r.t.
    opacity and depth."""
    n = opacity.numel()
    for t in (depth, sky_mask, valid_mask, lidar_depth):
        assert t is None or t.numel() == n, "all maps must have H*W elements"
    assert lidar_depth is None or depth is not None
    assert reduction in _DEPTH_RED, "reduction 'none' returns a map: use depth_error_map()"
    opts = _opts(opacity_loss_type, depth_loss_type, normalize, inverse_depth, reduction, max_depth)
    select = None
    if depth_error_percentile is not None and lidar_depth is not None:
        with torch.no_grad():
            err = depth_error_map(depth, lidar_depth, valid_mask, max_depth, depth_loss_type, normalize, inverse_depth)
            flat = err.reshape(-1)
            idx = torch.nonzero(flat >= 0).reshape(-1)  # the selected pixels, in the reference's (row-major) order
            keep = idx[torch.argsort(flat[idx], stable=True)[: int(idx.numel() * depth_error_percentile)]]
            select = torch.zeros(n, device=opacity.device, dtype=torch.uint8)
            select[keep] = 1
    out, _ = _PixelLosses.apply(opacity, depth, sky_mask, valid_mask, lidar_depth, opts, select, False)
    res = {"opacity_entropy": out[3]}
    if sky_mask is not None:
        res["sky_bce"] = out[0]
    if lidar_depth is not None:
        res["depth_l1"] = out[1]
    return res


def depth_error_map(depth: Tensor, lidar_depth: Tensor, valid_mask: Optional[Tensor] = None, max_depth: float = 80.0,
                    depth_loss_type: str = "l1", normalize: bool = False, inverse_depth: bool = False) -> Tensor:
    """DepthLoss(..., reduction="none") as a map: each pixel's depth error, -1 where the pixel is not selected (the
    reference returns the selected pixels' errors as a 1-D tensor: `m[m >= 0]`).  Not differentiable."""
    opts = _opts("bce", depth_loss_type, normalize, inverse_depth, "sum", max_depth)
    with torch.no_grad():
        op = torch.full_like(lidar_depth.reshape(-1), 0.5, dtype=torch.float32)
        _, err = _PixelLosses.apply(op, depth, None, valid_mask, lidar_depth, opts, None, True)
    return err.reshape(lidar_depth.shape)


class _InvDepthSmooth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, image):
        dev = B.require_gpu(depth, image)
        H, W = image.shape[0], image.shape[1]
        d, im = B.f32c(depth).reshape(-1), B.f32c(image)
        L = B.lib()
        nws = int(L.gs_pixel_losses_workspace_floats())
        ws = torch.empty(nws, device=dev, dtype=torch.float32)
        out = torch.empty(1, device=dev, dtype=torch.float32)
        with B.device_guard(dev), B.timed("gs_inv_depth_smooth_fwd", dev):
            B.check(L.gs_inv_depth_smooth_fwd(H, W, B.ptr(d), B.ptr(im), B.ptr(ws), nws, B.ptr(out),
                                              B.stream_handle(dev)), "gs_inv_depth_smooth_fwd")
        ctx.save_for_backward(d, im)
        ctx.shape, ctx.dtype = tuple(depth.shape), depth.dtype
        return out[0]

    @staticmethod
    def backward(ctx, g):
        d, im = ctx.saved_tensors
        dev = d.device
        H, W = im.shape[0], im.shape[1]
        v = torch.empty_like(d)
        with B.device_guard(dev), B.timed("gs_inv_depth_smooth_bwd", dev):
            B.check(B.lib().gs_inv_depth_smooth_bwd(H, W, B.ptr(d), B.ptr(im), B.ptr(B.f32c(g).reshape(1)), B.ptr(v),
                                                    B.stream_handle(dev)), "gs_inv_depth_smooth_bwd")
        return v.reshape(ctx.shape).to(ctx.dtype), None


def inverse_depth_smoothness(depth: Tensor, image: Tensor) -> Tensor:
    """`kornia.losses.inverse_depth_smoothness_loss(1 / (depth + 1e-5) repeated over 3 channels, image)` of
    base.py:575-585 (PVG's regulariser): edge-aware first-order smoothness of the inverse depth.  depth [H,W,1] or
    [H,W], image [H,W,3].  kornia is not available here: parity unpinned (oracle/loss_ref.py restates its formula)."""
    assert image.dim() == 3 and image.shape[-1] == 3 and depth.numel() == image.shape[0] * image.shape[1]
    return _InvDepthSmooth.apply(depth, image)


class _RegionL1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt, valid_mask, region, thresh):
        dev = B.require_gpu(pred, gt, valid_mask, region)
        n = region.numel()
        p, g, r = B.f32c(pred).reshape(-1, 3), B.f32c(gt).reshape(-1, 3), B.f32c(region).reshape(-1)
        vm = B.f32c(valid_mask).reshape(-1) if valid_mask is not None else None
        L = B.lib()
        nws = int(L.gs_pixel_losses_workspace_floats())
        ws = torch.empty(nws, device=dev, dtype=torch.float32)
        out = torch.empty(2, device=dev, dtype=torch.float32)
        with B.device_guard(dev), B.timed("gs_region_l1_fwd", dev):
            B.check(L.gs_region_l1_fwd(n, B.ptr(p), B.ptr(g), B.ptr(vm), B.ptr(r), float(thresh), B.ptr(ws), nws,
                                       B.ptr(out), B.stream_handle(dev)), "gs_region_l1_fwd")
        ctx.has_vm, ctx.thresh = vm is not None, float(thresh)
        ctx.shape, ctx.dtype = tuple(pred.shape), pred.dtype
        ctx.save_for_backward(p, g, vm if vm is not None else out, r, out)
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, g_loss, _g_out):
        p, g, vm, r, out = ctx.saved_tensors
        vm = vm if ctx.has_vm else None
        dev = p.device
        v = torch.empty_like(p)
        with B.device_guard(dev), B.timed("gs_region_l1_bwd", dev):
            B.check(B.lib().gs_region_l1_bwd(r.numel(), B.ptr(p), B.ptr(g), B.ptr(vm), B.ptr(r), ctx.thresh, B.ptr(out),
                                             B.ptr(B.f32c(g_loss).reshape(1)), B.ptr(v), B.stream_handle(dev)),
                    "gs_region_l1_bwd")
        return v.reshape(ctx.shape).to(ctx.dtype), None, None, None, None


def dynamic_region_l1(pred: Tensor, gt: Tensor, region_opacity: Tensor, valid_mask: Optional[Tensor] = None,
                      thresh: float = 0.2):
    """"vehicle_region_rgb_loss" of base.py:598-613: `|gt[m] - pred[m]|.mean()` with m = (Dynamic_opacity > 0.2) & valid,
    on the valid-masked images.  Returns (loss, n_pixels): the reference adds the term only `if m.sum() > 0` (a host
    synchronisation there); here the loss is an exact 0 with zero gradient in that case and the count stays on the device."""
    assert pred.shape[-1] == 3 and gt.shape == pred.shape and region_opacity.numel() * 3 == pred.numel()
    loss, out = _RegionL1.apply(pred, gt, valid_mask, region_opacity, thresh)
    return loss, out[1]


class _AffineReg(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A):
        dev = B.require_gpu(A)
        a = B.f32c(A)
        rows = a.shape[-2]
        m = a.numel() // (rows * 4)
        out = torch.empty(1, device=dev, dtype=torch.float32)
        with B.device_guard(dev):
            B.check(B.lib().gs_affine_reg(m, rows, B.ptr(a), None, B.ptr(out), None, B.stream_handle(dev)),
                    "gs_affine_reg")
        ctx.save_for_backward(a)
        ctx.dtype = A.dtype
        return out[0]

    @staticmethod
    def backward(ctx, g):
        (a,) = ctx.saved_tensors
        dev = a.device
        rows = a.shape[-2]
        m = a.numel() // (rows * 4)
        v = torch.zeros_like(a)
        with B.device_guard(dev):
            B.check(B.lib().gs_affine_reg(m, rows, B.ptr(a), B.ptr(B.f32c(g).reshape(1)), None, B.ptr(v),
                                          B.stream_handle(dev)), "gs_affine_reg")
        return v.to(ctx.dtype)


def affine_regulariser(affine_trs: Tensor) -> Tensor:
    """"affine_loss" of base.py:587-596: `|A[..., :3, :3] - I|.mean() + |A[..., :3, 3:]|.mean()` for the per-image
    affine colour transforms [..., R >= 3, 4]."""
    assert affine_trs.shape[-1] == 4 and affine_trs.shape[-2] >= 3, affine_trs.shape
    return _AffineReg.apply(affine_trs)
