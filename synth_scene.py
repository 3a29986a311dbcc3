"""Deterministic synthetic driving-scene generator ("Synthetic scene").

Shared by bench.py and tests/.  Produces *activated* Gaussian parameters exactly as the
reference's node classes hand them to `rasterization()`
(/root/reference/3D-Reconstruction-main/models/gaussians/vanilla.py:359-386) and a 5-camera
rig shaped like the reference's CARLA/KITTI rig (1242x375, horizontal FOV 90 deg,
/root/reference/3D-Reconstruction-main/datasets/kitti/kitti_5cams_preprocess.py:72-133).
"""
from __future__ import annotations

import math
from typing import Dict

import torch

C0 = 0.28209479177387814  # /root/reference/3D-Reconstruction-main/models/gaussians/basics.py:80


def make_gaussians(N: int, sh_degree: int | None = 3, seed: int = 42, extent: float = 40.0,
                   dtype=torch.float32) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    xy = (torch.rand(N, 2, generator=g) * 2 - 1) * extent
    z = torch.rand(N, 1, generator=g) * 10.0 - 2.0
    means = torch.cat([xy, z], dim=1)
    log_s = math.log(0.08) + 0.6 * torch.randn(N, 3, generator=g)
    scales = torch.exp(log_s).clamp(1e-3, 2.0)
    quats = torch.randn(N, 4, generator=g)
    quats = quats / quats.norm(dim=-1, keepdim=True)
    opacities = torch.sigmoid(1.5 * torch.randn(N, generator=g))
    out = dict(means=means, quats=quats, scales=scales, opacities=opacities)
    if sh_degree is None:
        out["colors"] = torch.rand(N, 3, generator=g)
    else:
        K = (sh_degree + 1) ** 2
        sh = 0.05 * torch.randn(N, K, 3, generator=g)
        sh[:, 0, :] = torch.randn(N, 3, generator=g) * 0.25 / C0
        out["colors"] = sh
    return {k: v.to(dtype) for k, v in out.items()}


def _look_at_c2w(eye, yaw_deg: float, pitch_deg: float) -> torch.Tensor:
    """OpenCV camera (x right, y down, z forward) in a z-up world."""
    yaw, pitch = math.radians(yaw_deg), math.radians(pitch_deg)
    fwd = torch.tensor([math.cos(yaw) * math.cos(pitch), math.sin(yaw) * math.cos(pitch), math.sin(pitch)],
                       dtype=torch.float64)
    up = torch.tensor([0.0, 0.0, 1.0], dtype=torch.float64)
    right = torch.linalg.cross(fwd, up)
    if float(right.norm()) < 1e-6:
        right = torch.tensor([0.0, -1.0, 0.0], dtype=torch.float64)
    right = right / right.norm()
    down = torch.linalg.cross(fwd, right)
    c2w = torch.eye(4, dtype=torch.float64)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2] = right, down, fwd
    c2w[:3, 3] = torch.as_tensor(eye, dtype=torch.float64)
    return c2w


# yaw, pitch of the five cameras (LEFT/RIGHT/FRONT/BACK/TOP naming of
# /root/reference/3D-Reconstruction-main/configs/datasets/kitti/5cams.yaml:4-8)
RIG = [(0.0, 0.0), (180.0, 0.0), (90.0, 0.0), (-60.0, 0.0), (30.0, 35.0)]


def make_cameras(C: int, width: int, height: int, frame: int = 0, dtype=torch.float32, origin_x: float = 0.0):
    """viewmats [C,4,4] (world->camera), Ks [C,3,3]; FOV 90 deg horizontally, pixel centres +0.5.  The ego sits at
    x = origin_x + 0.5 m * frame: with origin_x = 0 a 120-frame path ends at x = 59.5 m, 20 m outside
    the +-40 m slab of `make_gaussians`, and the forward camera shows nothing from frame 80 on; trajectory workloads
    start at origin_x = -30 so that every frame is inside the scene."""
    vms, Ks = [], []
    for c in range(C):
        yaw, pitch = RIG[c % len(RIG)]
        f = frame + c // len(RIG)
        c2w = _look_at_c2w((origin_x + 0.5 * f, 0.0, 1.6), yaw, pitch)
        vms.append(torch.linalg.inv(c2w))
        fx = width / 2.0
        Ks.append(torch.tensor([[fx, 0, width / 2.0], [0, fx, height / 2.0], [0, 0, 1.0]], dtype=torch.float64))
    return torch.stack(vms).to(dtype), torch.stack(Ks).to(dtype)
