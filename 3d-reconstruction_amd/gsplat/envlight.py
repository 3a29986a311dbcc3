"""Sky environment light lookup: the one nvdiffrast call of the reference's EnvLight
(/root/reference/3D-Reconstruction-main/models/modules.py:174-203),

    light = dr.texture(self.base[None, ...], l, filter_mode='linear', boundary_mode='cube')

as a HIP op.  nvdiffrast is CUDA-only, so on this part the reference's sky model cannot run without it.
`cube_texture(base, dirs, rot)` is differentiable w.r.t. the cube map `base [6,R,R,C]`; the view directions are data
(image_infos["viewdirs"]) and get no gradient.  `EnvLight` below mirrors the reference module (same Parameter name and
shape, same `to_opengl` rotation, same output shape) so its state_dict loads unchanged.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from . import _backend as B


class _CubeTexture(torch.autograd.Function):
    @staticmethod
    def forward(ctx, base, dirs, rot):
        dev = B.require_gpu(base, dirs, rot)
        R, C = base.shape[1], base.shape[3]
        base_c, dirs_c = B.f32c(base), B.f32c(dirs.detach()).reshape(-1, 3)
        rot_c = B.f32c(rot).reshape(9) if rot is not None else None
        n = dirs_c.shape[0]
        out = torch.empty(n, C, device=dev, dtype=torch.float32)
        with B.device_guard(dev), B.timed("gs_cube_texture_fwd", dev):
            B.check(B.lib().gs_cube_texture_fwd(n, R, C, B.ptr(base_c), B.ptr(dirs_c), B.ptr(rot_c), B.ptr(out),
                                                B.stream_handle(dev)), "gs_cube_texture_fwd")
        ctx.geom = (n, R, C, tuple(base.shape))
        ctx.in_dtype = base.dtype
        ctx.has_rot = rot is not None
        ctx.save_for_backward(dirs_c, rot_c if rot_c is not None else dirs_c)
        return out.reshape(tuple(dirs.shape[:-1]) + (C,))

    @staticmethod
    def backward(ctx, v_out):
        dirs_c, rot_c = ctx.saved_tensors
        rot_c = rot_c if ctx.has_rot else None
        n, R, C, shape = ctx.geom
        dev = dirs_c.device
        v = B.f32c(v_out).reshape(n, C)
        v_tex = torch.zeros(shape, device=dev, dtype=torch.float32)
        with B.device_guard(dev), B.timed("gs_cube_texture_bwd", dev):
            B.check(B.lib().gs_cube_texture_bwd(n, R, C, B.ptr(dirs_c), B.ptr(rot_c), B.ptr(v), B.ptr(v_tex),
                                                B.stream_handle(dev)), "gs_cube_texture_bwd")
        return v_tex.to(ctx.in_dtype), None, None


def cube_texture(base: Tensor, dirs: Tensor, rot: Optional[Tensor] = None) -> Tensor:
    """Bilinear, seamless cube-map lookup: base [6,R,R,C] (faces +x,-x,+y,-y,+z,-z, OpenGL orientation), dirs [...,3]
    (need not be unit), rot: optional [3,3], the directions are looked up as `dirs @ rot.T`.  Returns [...,C]."""
    assert base.dim() == 4 and base.shape[0] == 6 and base.shape[1] == base.shape[2], base.shape
    assert dirs.shape[-1] == 3 and (rot is None or tuple(rot.shape) == (3, 3))
    return _CubeTexture.apply(base, dirs, rot)


class EnvLight(torch.nn.Module):
    """modules.py:174-208 with the nvdiffrast call replaced; `base` and `to_opengl` as in the reference."""

    def __init__(self, class_name: str, resolution: int = 1024, device: torch.device = torch.device("cuda"), **kwargs):
        super().__init__()
        self.class_prefix = class_name + "#"
        self.device = device
        self.to_opengl = torch.tensor([[1, 0, 0], [0, 0, 1], [0, -1, 0]], dtype=torch.float32, device=device)
        self.base = torch.nn.Parameter(0.5 * torch.ones(6, resolution, resolution, 3, device=device))

    def forward(self, image_infos):
        l = image_infos["viewdirs"]
        return cube_texture(self.base, l, self.to_opengl)  # == (l @ to_opengl.T) looked up, shape l.shape[:-1] + (3,)

    def get_param_groups(self):
        return {self.class_prefix + "all": self.parameters()}
