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
    with B.device_guard(dev):
        B.check(B.lib().gs_densify_accumulate(C, N, B.ptr(radii), grad.data_ptr(), ld, W / 2.0 * batch_size,
                                              H / 2.0 * batch_size, 1.0 / float(max(W, H)), int(first),
                                              B.ptr(state.xys_grad_norm), B.ptr(state.vis_counts),
                                              B.ptr(state.max_2Dsize), B.stream_handle(dev)), "gs_densify_accumulate")
    return state


# ---------------------------------------------------------------------------------------------------------------------
# split / duplicate / cull of one class and its Adam moments (vanilla.py:197-283, basics.py:162-206)

_PARAM_KEYS = ("means", "features_dc", "features_rest", "opacities", "scales", "quats")
SCALE_3D, SCALE_BALL, SCALE_2D = 0, 1, 2


@dataclass
class RefineConfig:
    """The `ctrl_cfg` values `refinement_after` reads, with its step conditions already evaluated by the caller:
    pass None for a test the reference skips at this step."""
    densify_grad_thresh: float = 0.0002
    densify_size_thresh: float = 0.01
    scene_scale: float = 1.0
    n_split_samples: int = 2
    split_screen_size: Optional[float] = None   # None when step >= stop_screen_size_at (vanilla.py:217)
    cull_alpha_thresh: float = 0.005
    cull_scale_thresh: Optional[float] = None   # None when step <= reset_alpha_interval (vanilla.py:292)
    cull_screen_size: Optional[float] = None    # None when step >= stop_screen_size_at (vanilla.py:299)
    do_densify: bool = True                     # vanilla.py:204-207
    do_cull: bool = True                        # vanilla.py:262
    ball_gaussians: bool = False
    gaussian_2d: bool = False


def refine_gaussians(params: Dict[str, Tensor], state: Optional[DensifyState], cfg: RefineConfig,
                     adam: Optional[Dict[str, Dict[str, Tensor]]] = None, generator: Optional[torch.Generator] = None):
    """Split + duplicate + cull of one class in one composed pass (see csrc/refine.hip).

    params: the six raw tensors `means [n,3]`, `features_dc [n,3]`, `features_rest [n,K-1,3]`, `opacities [n,1]`,
    `scales [n,3]` (`[n,1]` for ball Gaussians), `quats [n,4]`; state: this class's accumulators (needed when
    cfg.do_densify or a screen-size test is on); adam: optional `{key: {"exp_avg": t, "exp_avg_sq": t}}` moments.
    Returns (new_params, new_adam, info) with rows in the reference's order (survivors, split children sample-major,
    duplicates; culled rows dropped); new rows get zero moments.  The `randn` draw for the split samples has the
    reference's shape `(n_split_samples * n_splits, 3)` (vanilla.py:321), so a shared seed gives the same children.
    One host synchronisation."""
    from .cuda._wrapper import exclusive_scan_i32

    p = {k: params[k] for k in _PARAM_KEYS}
    n = p["means"].shape[0]
    dev = B.require_gpu(*p.values())
    smode = SCALE_BALL if cfg.ball_gaussians else (SCALE_2D if cfg.gaussian_2d else SCALE_3D)
    assert p["scales"].shape == ((n, 1) if smode == SCALE_BALL else (n, 3)), p["scales"].shape
    assert p["opacities"].numel() == n and p["quats"].shape == (n, 4)
    c = {k: B.f32c(v.detach()) for k, v in p.items()}
    need_stats = cfg.do_densify or (cfg.do_cull and cfg.cull_scale_thresh is not None and cfg.cull_screen_size is not None)
    gn = vc = m2 = None
    if need_stats:
        assert state is not None and state.xys_grad_norm is not None and state.vis_counts is not None \
            and state.max_2Dsize is not None, "densification statistics missing (vanilla.py:210)"
        gn, vc, m2 = (B.f32c(t).reshape(-1) for t in (state.xys_grad_norm, state.vis_counts, state.max_2Dsize))
        assert gn.numel() == n and vc.numel() == n and m2.numel() == n
    neg = lambda v, scale=1.0: -1.0 if v is None else float(v) * scale
    flags = torch.empty(n, device=dev, dtype=torch.uint8)
    counts = torch.empty(4 * n + 1, device=dev, dtype=torch.int32)
    L, stream = B.lib(), B.stream_handle(dev)
    with B.device_guard(dev):
        B.check(L.gs_refine_decide(n, smode, int(cfg.do_densify), int(cfg.do_cull), cfg.densify_grad_thresh,
                                   cfg.densify_size_thresh * cfg.scene_scale, neg(cfg.split_screen_size),
                                   cfg.cull_alpha_thresh, neg(cfg.cull_scale_thresh, cfg.scene_scale),
                                   neg(cfg.cull_screen_size), B.ptr(c["scales"]), B.ptr(c["opacities"]), B.ptr(gn),
                                   B.ptr(vc), B.ptr(m2), B.ptr(flags), B.ptr(counts), stream), "gs_refine_decide")
    scan, _ = exclusive_scan_i32(counts)
    idx = torch.tensor([0, n, 2 * n, 3 * n, 4 * n], device=dev)
    b = scan[idx].tolist()  # the one host synchronisation
    n_splits, n_self, n_ksplit, n_kdup = b[1] - b[0], b[2] - b[1], b[3] - b[2], b[4] - b[3]
    ns = int(cfg.n_split_samples)
    n_out = n_self + ns * n_ksplit + n_kdup
    samples = torch.randn((ns * n_splits, 3), device=dev, generator=generator)
    src = torch.empty(n_out, device=dev, dtype=torch.int32)
    kinds = torch.empty(n_out, device=dev, dtype=torch.uint8)
    sidx = torch.empty(n_out, device=dev, dtype=torch.int32)
    new_p: Dict[str, Tensor] = {}
    new_adam: Optional[Dict[str, Dict[str, Tensor]]] = {} if adam is not None else None

    def gather(t: Tensor, mode: int) -> Tensor:
        t = B.f32c(t.detach())
        W = 1
        for d in t.shape[1:]:
            W *= int(d)
        out = torch.empty((n_out,) + tuple(t.shape[1:]), device=dev, dtype=torch.float32)
        B.check(L.gs_refine_gather_rows(n_out, W, mode, B.ptr(src), B.ptr(kinds), B.ptr(flags), B.ptr(t), B.ptr(out),
                                        stream), "gs_refine_gather_rows")
        return out

    with B.device_guard(dev):
        if n_out > 0:  # (everything culled: the outputs are empty and there is nothing to plan)
            B.check(L.gs_refine_plan(n, ns, B.ptr(flags), B.ptr(scan), B.ptr(src), B.ptr(kinds), B.ptr(sidx), stream),
                    "gs_refine_plan")
        new_means = torch.empty(n_out, 3, device=dev, dtype=torch.float32)
        B.check(L.gs_refine_split_means(n_out, smode, B.ptr(src), B.ptr(kinds), B.ptr(sidx), B.ptr(c["means"]),
                                        B.ptr(c["scales"]), B.ptr(c["quats"]), B.ptr(samples), B.ptr(new_means),
                                        stream), "gs_refine_split_means")
        new_p["means"] = new_means
        for k in _PARAM_KEYS[1:]:
            new_p[k] = gather(c[k], 2 if k == "scales" else 0)
        if adam is not None:
            for k, st in adam.items():
                new_adam[k] = {m: gather(v, 1) for m, v in st.items() if torch.is_tensor(v) and v.dim() > 0
                               and v.shape[0] == n}
    info = {"n_before": n, "n_after": n_out, "n_splits": n_splits, "n_kept_split_children": ns * n_ksplit,
            "n_kept_dups": n_kdup, "n_culled_old": n - n_self}
    return new_p, new_adam, info


def reset_opacities(raw_opacities: Tensor, reset_alpha_value: float, adam_state: Optional[dict] = None) -> None:
    """The opacity reset of `refinement_after` (/root/reference/3D-Reconstruction-main/models/gaussians/vanilla.py:268-280),
    in place and in one pass: `raw = logit(min(sigmoid(raw), reset_alpha_value))`, and the opacity parameter's Adam
    moments (`adam_state = optimizer.state[param]` with "exp_avg" / "exp_avg_sq") zeroed where the reference replaces
    them with `zeros_like`.  `raw_opacities` is the Parameter's `.data` (any shape, fp32, contiguous)."""
    dev = B.require_gpu(raw_opacities)
    assert raw_opacities.dtype == torch.float32 and raw_opacities.is_contiguous()
    ea = adam_state.get("exp_avg") if adam_state else None
    es = adam_state.get("exp_avg_sq") if adam_state else None
    for t in (ea, es):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous() and t.numel() == raw_opacities.numel())
    with B.device_guard(dev):
        B.check(B.lib().gs_opacity_reset(raw_opacities.numel(), float(reset_alpha_value), B.ptr(raw_opacities),
                                         B.ptr(ea), B.ptr(es), B.stream_handle(dev)), "gs_opacity_reset")
