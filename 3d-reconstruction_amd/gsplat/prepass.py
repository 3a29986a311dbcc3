"""Raw Parameters -> render inputs in one kernel per Gaussian class (training half).

Before every `rasterization()` call the reference activates each class's Parameters with ~25 torch ops
(/root/reference/3D-Reconstruction-main/models/gaussians/vanilla.py:359-393 `get_gaussians`: a `[N,K,3]` cat of the SH
coefficients, view directions, `spherical_harmonics`, `+0.5`/clamp, sigmoid, exp, quaternion normalisation, five
boolean-mask gathers, ten `isnan`/`isinf` reductions that each synchronise the host) and then concatenates the classes
(models/trainers/base.py:342-383 `collect_gaussians`).  `collect_gaussians` here does the same arithmetic with
`gs_activate_fwd` writing each class's slice of the scene-level arrays directly, and `gs_activate_bwd` producing the
gradients of the raw Parameters from the gradients of those arrays.  Same outputs, names and shapes as the reference's
`gs_dict` (`_means [N,3]`, `_opacities [N,1]`, `_rgbs [N,3]`, `_scales [N,3]`, `_quats [N,4]`) plus `class_labels`.

Classes whose means / quats go through an instance pose first (nodes/rigid.py:477-478) pass the world-frame tensors as
`means` / `quats` (any autograd tensor: the means pass through, the quaternions are normalised here, rigid.py:495) and
their per-point validity as `opacity_mask` (rigid.py:493).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import _backend as B

_KEYS = ("_means", "_opacities", "_rgbs", "_scales", "_quats")  # order of vanilla.py:380-386 and of the flag bits
SCALE_3D, SCALE_BALL, SCALE_2D = 0, 1, 2


@dataclass
class GaussianClass:
    """Raw Parameters of one class (vanilla.py:74-97) and the per-step choices `get_gaussians` makes for it."""
    means: Tensor            # [n,3]
    features_dc: Tensor      # [n,3]
    features_rest: Tensor    # [n,K-1,3]
    opacities: Tensor        # [n,1] (or [n]) logits
    scales: Tensor           # [n,3] log-scales ([n,1] with ball_gaussians)
    quats: Tensor            # [n,4] wxyz, any norm
    sh_degree: int           # the class's sh_degree; 0 -> colours are sigmoid(features_dc) (vanilla.py:371-372)
    sh_degree_to_use: int    # n = min(step // sh_degree_interval, sh_degree) (vanilla.py:368)
    label: int = 0           # class_labels value (base.py:361)
    opacity_mask: Optional[Tensor] = None  # [n] bool/float, multiplies the activated opacity (rigid.py:493)
    ball_gaussians: bool = False           # vanilla.py:115-121
    gaussian_2d: bool = False              # vanilla.py:116-126

    def scale_mode(self) -> int:
        if self.ball_gaussians:
            return SCALE_BALL  # (with gaussian_2d the reference asserts, vanilla.py:117-119, unless scales is [n,1])
        return SCALE_2D if self.gaussian_2d else SCALE_3D


class FiniteFlag:
    """Device word collecting the reference's NaN/Inf checks (vanilla.py:388-393).  `check()` reads it (one host
    synchronisation) and raises the reference's ValueError; call it whenever a synchronisation is affordable."""

    def __init__(self, device: torch.device):
        self.word = torch.zeros(1, dtype=torch.int32, device=device)

    def reset(self) -> None:
        """Clear the collected bits (stream-ordered: calls already queued still report into the old state)."""
        self.word.zero_()

    def check(self, step: Optional[int] = None) -> None:
        w = int(self.word.item()) & 0xFFFFFFFF
        if w == 0:
            return
        self.word.zero_()  # a reported problem is reported once
        for k, name in enumerate(_KEYS):  # the reference checks key by key, NaN before Inf
            for bit, what in ((2 * k, "NaN"), (2 * k + 1, "Inf")):
                if w >> bit & 1:
                    raise ValueError(f"{what} detected in gaussian {name} at step {step}")


class _Collect(torch.autograd.Function):
    @staticmethod
    def forward(ctx, spec, cam_center, flag_word, *tensors):
        # spec: per class (n, K_rest, degree, scale_mode, has_mask); tensors: 7 per class
        dev = cam_center.device
        total = sum(s[0] for s in spec)
        o_means = torch.empty(total, 3, device=dev, dtype=torch.float32)
        o_opac = torch.empty(total, 1, device=dev, dtype=torch.float32)
        o_rgbs = torch.empty(total, 3, device=dev, dtype=torch.float32)
        o_scales = torch.empty(total, 3, device=dev, dtype=torch.float32)
        o_quats = torch.empty(total, 4, device=dev, dtype=torch.float32)
        cmask = torch.empty(total, device=dev, dtype=torch.uint8)
        L = B.lib()
        stream = B.stream_handle(dev)
        saved = []
        row = 0
        with B.device_guard(dev):
            for ci, (n, kr, deg, smode, has_mask) in enumerate(spec):
                means, dc, rest, opac, scales, quats, mask = tensors[7 * ci:7 * ci + 7]
                means, dc, rest, opac, scales, quats = (B.f32c(t.detach()) for t in (means, dc, rest, opac, scales, quats))
                mask = B.f32c(mask) if has_mask else None
                with B.timed("gs_activate_fwd", dev):
                    B.check(L.gs_activate_fwd(n, kr, deg, B.ptr(means), B.ptr(cam_center), B.ptr(dc), B.ptr(rest),
                                              B.ptr(opac), B.ptr(mask), B.ptr(scales), smode, B.ptr(quats),
                                              B.ptr(o_means[row:]), B.ptr(o_opac[row:]), B.ptr(o_rgbs[row:]),
                                              B.ptr(o_scales[row:]), B.ptr(o_quats[row:]), B.ptr(cmask[row:]),
                                              B.ptr(flag_word), stream), "gs_activate_fwd")
                saved += [means, dc, opac, scales, quats] + ([mask] if has_mask else [])
                row += n
        ctx.spec = spec
        ctx.shapes = [tuple(t.shape) if t is not None else None for t in tensors]
        ctx.save_for_backward(cam_center, cmask, *saved)
        return o_means, o_opac, o_rgbs, o_scales, o_quats

    @staticmethod
    def backward(ctx, v_means, v_opac, v_rgbs, v_scales, v_quats):
        cam_center, cmask, *saved = ctx.saved_tensors
        dev = cam_center.device
        v_opac, v_rgbs, v_scales, v_quats = (B.f32c(t) for t in (v_opac, v_rgbs, v_scales, v_quats))
        L = B.lib()
        stream = B.stream_handle(dev)
        grads: List[Optional[Tensor]] = []
        row = 0
        si = 0
        with B.device_guard(dev):
            for ci, (n, kr, deg, smode, has_mask) in enumerate(ctx.spec):
                means, dc, opac, scales, quats = saved[si:si + 5]
                si += 5
                mask = None
                if has_mask:
                    mask = saved[si]
                    si += 1
                sh = ctx.shapes[7 * ci:7 * ci + 7]
                g_dc = torch.empty(sh[1], device=dev, dtype=torch.float32)
                g_rest = torch.empty(sh[2], device=dev, dtype=torch.float32)
                g_opac = torch.empty(sh[3], device=dev, dtype=torch.float32)
                g_scales = torch.empty(sh[4], device=dev, dtype=torch.float32)
                g_quats = torch.empty(sh[5], device=dev, dtype=torch.float32)
                with B.timed("gs_activate_bwd", dev):
                    B.check(L.gs_activate_bwd(n, kr, deg, B.ptr(means), B.ptr(cam_center), B.ptr(dc), B.ptr(opac),
                                              B.ptr(mask), B.ptr(scales), smode, B.ptr(quats), B.ptr(cmask[row:]),
                                              B.ptr(v_opac[row:]), B.ptr(v_rgbs[row:]), B.ptr(v_scales[row:]),
                                              B.ptr(v_quats[row:]), B.ptr(g_dc), B.ptr(g_rest) if kr > 0 else None,
                                              B.ptr(g_opac), B.ptr(g_scales), B.ptr(g_quats), stream),
                            "gs_activate_bwd")
                needs = ctx.needs_input_grad[3 + 7 * ci:3 + 7 * ci + 7]
                cls = [v_means[row:row + n], g_dc, g_rest, g_opac, g_scales, g_quats, None]
                grads += [g if nd else None for g, nd in zip(cls, needs)]
                row += n
        return (None, None, None, *grads)


def collect_gaussians(classes: Sequence[GaussianClass], camtoworld: Tensor, *, finite_flag: Optional[FiniteFlag] = None,
                      check_finite: bool = False, step: Optional[int] = None) -> Tuple[Dict[str, Tensor], Tensor, FiniteFlag]:
    """`get_gaussians` of every class + `collect_gaussians`' concatenation (base.py:342-383).

    camtoworld: the camera's [4,4] (or [..,4,4] with one camera) pose; only its translation is used (vanilla.py:366).
    Returns (gs_dict, class_labels, finite_flag).  `check_finite=True` reads the flag now (the reference's blocking
    behaviour); otherwise call `finite_flag.check()` wherever a host synchronisation is affordable."""
    assert len(classes) > 0
    dev = B.require_gpu(camtoworld, *(c.means for c in classes))
    cam_center = B.f32c(camtoworld.detach().reshape(-1, 4, 4)[0, :3, 3])
    flag = finite_flag if finite_flag is not None else FiniteFlag(dev)
    spec, tensors, labels = [], [], []
    for c in classes:
        n = c.means.shape[0]
        assert c.means.shape == (n, 3) and c.features_dc.shape == (n, 3) and c.quats.shape == (n, 4), "class shapes"
        assert c.features_rest.dim() == 3 and c.features_rest.shape[0] == n and c.features_rest.shape[2] == 3
        assert c.opacities.numel() == n
        kr = c.features_rest.shape[1]
        smode = c.scale_mode()
        assert c.scales.shape == ((n, 1) if smode == SCALE_BALL else (n, 3)), c.scales.shape
        if c.sh_degree > 0:
            deg = int(c.sh_degree_to_use)
            assert 0 <= deg <= 4 and (deg + 1) ** 2 - 1 <= kr, (deg, kr)
        else:
            deg = -1
        has_mask = c.opacity_mask is not None
        if has_mask:
            assert c.opacity_mask.numel() == n
        B.require_gpu(c.means, c.features_dc, c.features_rest, c.opacities, c.scales, c.quats, c.opacity_mask)
        spec.append((n, kr, deg, smode, has_mask))
        mask = c.opacity_mask.reshape(n).float() if has_mask else None
        tensors += [c.means, c.features_dc, c.features_rest, c.opacities, c.scales, c.quats, mask]
        labels.append(torch.full((n,), int(c.label), device=dev))
    outs = _Collect.apply(tuple(spec), cam_center, flag.word, *tensors)
    gs_dict = dict(zip(_KEYS, outs))
    if check_finite:
        flag.check(step)
    return gs_dict, torch.cat(labels, dim=0), flag


# ---------------------------------------------------------------------------------------------------------------------
# rigid nodes: instance-pose transform (nodes/rigid.py:403-470)

class _RigidTransform(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means, quats, point_ids, inst_quats, inst_trans):
        dev = B.require_gpu(means, quats, point_ids, inst_quats, inst_trans)
        n, n_inst = means.shape[0], inst_quats.shape[0]
        means, quats = B.f32c(means), B.f32c(quats)
        inst_quats, inst_trans = B.f32c(inst_quats), B.f32c(inst_trans)
        o_means = torch.empty(n, 3, device=dev, dtype=torch.float32)
        o_quats = torch.empty(n, 4, device=dev, dtype=torch.float32)
        with B.device_guard(dev), B.timed("gs_rigid_transform_fwd", dev):
            B.check(B.lib().gs_rigid_transform_fwd(n, n_inst, B.ptr(means), B.ptr(quats), B.ptr(point_ids),
                                                   B.ptr(inst_quats), B.ptr(inst_trans), B.ptr(o_means), B.ptr(o_quats),
                                                   B.stream_handle(dev)), "gs_rigid_transform_fwd")
        ctx.save_for_backward(means, quats, point_ids, inst_quats)
        ctx.set_materialize_grads(False)
        return o_means, o_quats

    @staticmethod
    def backward(ctx, v_means, v_quats):
        means, quats, point_ids, inst_quats = ctx.saved_tensors
        dev = means.device
        n, n_inst = means.shape[0], inst_quats.shape[0]
        need = ctx.needs_input_grad
        v_means = B.f32c(v_means) if v_means is not None else None
        v_quats = B.f32c(v_quats) if v_quats is not None else None
        g_means = torch.empty_like(means) if need[0] else None
        g_quats = torch.empty_like(quats) if need[1] else None
        g_iq = torch.zeros_like(inst_quats) if need[3] else None
        g_it = torch.zeros(n_inst, 3, device=dev, dtype=torch.float32) if need[4] else None
        with B.device_guard(dev), B.timed("gs_rigid_transform_bwd", dev):
            B.check(B.lib().gs_rigid_transform_bwd(n, n_inst, B.ptr(means), B.ptr(quats), B.ptr(point_ids),
                                                   B.ptr(inst_quats), B.ptr(v_means), B.ptr(v_quats), B.ptr(g_means),
                                                   B.ptr(g_quats), B.ptr(g_iq), B.ptr(g_it), B.stream_handle(dev)),
                    "gs_rigid_transform_bwd")
        return g_means, g_quats, None, g_iq, g_it


def rigid_transform(means: Tensor, quats: Tensor, point_ids: Tensor, inst_quats: Tensor, inst_trans: Tensor,
                    validate: bool = False) -> Tuple[Tensor, Tensor]:
    """RigidNodes.transform_means + transform_quats (nodes/rigid.py:403-470) for the current frame's pose rows
    `inst_quats [I,4]`, `inst_trans [I,3]` (after the caller's optional test-set interpolation, :411-450) and the
    per-point instance ids `point_ids [n]` or `[n,1]` (long).  Returns (world_means [n,3], world_quats [n,4]); the
    world quaternions are normalised later by `collect_gaussians` (rigid.py:495).  Differentiable w.r.t. means, quats
    and both pose tensors.  `validate=True` checks the ids on the host (one synchronisation)."""
    n = means.shape[0]
    assert means.shape == (n, 3) and quats.shape == (n, 4) and point_ids.numel() == n
    assert inst_quats.dim() == 2 and inst_quats.shape[1] == 4 and inst_trans.shape == (inst_quats.shape[0], 3)
    pid = point_ids.reshape(n).to(torch.int64).contiguous()
    if validate and n > 0:
        assert int(pid.min()) >= 0 and int(pid.max()) < inst_quats.shape[0], "point_ids out of range"
    return _RigidTransform.apply(means, quats, pid, inst_quats, inst_trans)
