"""Densification statistics on the device in one pass.

Drop-in for the bookkeeping the reference does after every backward: `BasicTrainer.postprocess_per_train_step`
(/root/reference/3D-Reconstruction-main/models/trainers/base.py:279-297) clones and rescales `means2d.absgrad`, then
per class `VanillaGaussians.after_train` (models/gaussians/vanilla.py:154-182) runs ~10 boolean-mask passes.
`accumulate_stats` reads `meta["radii"]` and the gradient record behind `meta["means2d"]` once and updates the three
accumulators in place; classes keep using them through their masks (`state.xys_grad_norm[class_mask]`).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import torch
from torch import Tensor

from . import _backend as B


@dataclass
class DensifyState:
    """The accumulators of vanilla.py:88-91, over ALL Gaussians of the scene."""
    xys_grad_norm: Optional[Tensor] = None  # [N] sum of |means2d grad| in pixel units over the steps a row was visible
    vis_counts: Optional[Tensor] = None     # [N] number of such steps (starts at 1, as the reference's does)
    max_2Dsize: Optional[Tensor] = None     # [N] max radii / max(W, H)


def accumulate_stats(meta: Dict, state: DensifyState, absgrad: bool = True, batch_size: int = 1) -> DensifyState:
    """Update `state` from the `meta` of the rasterization() call whose backward has just run (packed=False)."""
    radii = meta["radii"]
    m2d = meta["means2d"]
    grad = getattr(m2d, "absgrad", None) if absgrad else m2d.grad
    assert grad is not None, "run backward first (and means2d.retain_grad() when absgrad=False)"
    assert radii.dim() == 2, "accumulate_stats needs the dense [C,N] layout (packed=False)"
    C, N = radii.shape
    dev = B.require_gpu(radii, grad)
    assert grad.shape == (C, N, 2) and grad.dtype == torch.float32
    ok = N == 0 or (grad.stride(2) == 1 and grad.stride(0) == N * grad.stride(1) and grad.stride(1) >= 2)
    if not ok:
        grad = grad.contiguous()
    ld = grad.stride(1) if N > 0 else 2
    first = state.xys_grad_norm is None or state.xys_grad_norm.shape[0] != N
    if first:
        state.xys_grad_norm = torch.empty(N, device=dev, dtype=torch.float32)
        state.vis_counts = torch.empty(N, device=dev, dtype=torch.float32)
        state.max_2Dsize = torch.empty(N, device=dev, dtype=torch.float32)
    W, H = meta["width"], meta["height"]
    with torch.cuda.device(dev):
        B.check(B.lib().gs_densify_accumulate(C, N, B.ptr(radii), grad.data_ptr(), ld, W / 2.0 * batch_size,
                                              H / 2.0 * batch_size, 1.0 / float(max(W, H)), int(first),
                                              B.ptr(state.xys_grad_norm), B.ptr(state.vis_counts),
                                              B.ptr(state.max_2Dsize), B.stream_handle(dev)), "gs_densify_accumulate")
    return state
