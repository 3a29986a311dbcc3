"""CPU restatement (PyTorch, fp32 or fp64, differentiable by autograd) of the gsplat
1.x `rasterization()` path that the reference calls.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  PARITY UNPINNED (gsplat is an
un-vendored dependency; nothing in /root/reference pins its outputs).

Reference call sites this follows (R = /root/reference/3D-Reconstruction-main):
  * R/models/gaussians/basics.py:12-15     the four imported symbols
  * R/models/trainers/base.py:393-408      rasterization(...) training/eval call
  * R/models/trainers/base.py:772-787      rasterization(...) viewer call (RGB, radius_clip)
  * R/models/gaussians/vanilla.py:366-370  spherical_harmonics(n, dirs, coeffs) + 0.5, clamp
  * R/utils/geometry.py:116-152            quat_to_rotmat convention (wxyz, normalising)
The arithmetic itself restates the published gsplat algorithm as summarised in
SURVEY.md Appendix A (A.0 orchestration ... A.6 rasterize backward); each function
below names the appendix paragraph it restates.  Backward passes are obtained by
autograd over this forward, which is an independent derivation from the hand-written
HIP backward kernels it is used to check.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
from torch import Tensor

ALPHA_MIN = 1.0 / 255.0
ALPHA_MAX = 0.999
T_MIN = 1e-4


# --------------------------------------------------------------------------- helpers
def num_sh_bases(degree: int) -> int:
    """(degree+1)^2 for degree 0..4 -- R/models/gaussians/basics.py:13 (import),
    used at R/models/gaussians/vanilla.py:61,85."""
    assert 0 <= degree <= 4, "gsplat supports SH degree 0..4"
    return (degree + 1) ** 2


def quat_to_rotmat(quats: Tensor) -> Tensor:
    """wxyz quaternion -> rotation matrix, normalising first.
    Same convention as R/utils/geometry.py:116-152 (the in-tree twin)."""
    q = quats / quats.norm(dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    R = torch.stack(
        [
            1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
            2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
            2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y),
        ],
        dim=-1,
    )
    return R.reshape(quats.shape[:-1] + (3, 3))


def _sh_bases(deg: int, dirs: Tensor) -> Tensor:
    """Real SH basis values [..., (deg+1)^2] for (normalised) dirs.  Appendix A.2."""
    d = dirs / dirs.norm(dim=-1, keepdim=True)
    x, y, z = d.unbind(-1)
    out = [torch.full_like(x, 0.2820947917738781)]
    if deg >= 1:
        out += [-0.48860251190292 * y, 0.48860251190292 * z, -0.48860251190292 * x]
    if deg >= 2:
        z2 = z * z
        fTmp0B = -1.092548430592079 * z
        fC1 = x * x - y * y
        fS1 = 2.0 * x * y
        pSH6 = 0.9461746957575601 * z2 - 0.3153915652525201
        pSH7 = fTmp0B * x
        pSH5 = fTmp0B * y
        pSH8 = 0.5462742152960395 * fC1
        pSH4 = 0.5462742152960395 * fS1
        out += [pSH4, pSH5, pSH6, pSH7, pSH8]
    if deg >= 3:
        fTmp0C = -2.285228997322329 * z2 + 0.4570457994644658
        fTmp1B = 1.445305721320277 * z
        fC2 = x * fC1 - y * fS1
        fS2 = x * fS1 + y * fC1
        pSH12 = z * (1.865881662950577 * z2 - 1.119528997770346)
        pSH13 = fTmp0C * x
        pSH11 = fTmp0C * y
        pSH14 = fTmp1B * fC1
        pSH10 = fTmp1B * fS1
        pSH15 = -0.5900435899266435 * fC2
        pSH9 = -0.5900435899266435 * fS2
        out += [pSH9, pSH10, pSH11, pSH12, pSH13, pSH14, pSH15]
    if deg >= 4:
        fTmp0D = z * (-4.683325804901025 * z2 + 2.007139630671868)
        fTmp1C = 3.31161143515146 * z2 - 0.47308734787878
        fTmp2B = -1.770130769779931 * z
        fC3 = x * fC2 - y * fS2
        fS3 = x * fS2 + y * fC2
        pSH20 = 1.984313483298443 * z * pSH12 - 1.006230589874905 * pSH6
        pSH21 = fTmp0D * x
        pSH19 = fTmp0D * y
        pSH22 = fTmp1C * fC1
        pSH18 = fTmp1C * fS1
        pSH23 = fTmp2B * fC2
        pSH17 = fTmp2B * fS2
        pSH24 = 0.6258357354491763 * fC3
        pSH16 = 0.6258357354491763 * fS3
        out += [pSH16, pSH17, pSH18, pSH19, pSH20, pSH21, pSH22, pSH23, pSH24]
    return torch.stack(out, dim=-1)


def spherical_harmonics(
    degrees_to_use: int, dirs: Tensor, coeffs: Tensor, masks: Optional[Tensor] = None
) -> Tensor:
    """gsplat.cuda._wrapper.spherical_harmonics (R/models/gaussians/basics.py:15; called at
    R/models/gaussians/vanilla.py:369, R/models/nodes/rigid.py:486, smpl.py:475, ...).
    dirs [...,3] (normalised inside), coeffs [...,K,3] -> [...,3].  No +0.5 here (Appendix A.2).
    Rows with masks == False are zero."""
    assert (degrees_to_use + 1) ** 2 <= coeffs.shape[-2], coeffs.shape
    assert dirs.shape[:-1] == coeffs.shape[:-2], (dirs.shape, coeffs.shape)
    assert dirs.shape[-1] == 3 and coeffs.shape[-1] == 3
    nb = (degrees_to_use + 1) ** 2
    if masks is not None:
        assert masks.shape == dirs.shape[:-1]
        # keep masked rows finite for autograd
        dirs = torch.where(masks[..., None], dirs, torch.ones_like(dirs))
    basis = _sh_bases(degrees_to_use, dirs)  # [..., nb]
    out = (basis[..., :, None] * coeffs[..., :nb, :]).sum(dim=-2)
    if masks is not None:
        out = torch.where(masks[..., None], out, torch.zeros_like(out))
    return out


class _GsplatCompensation(torch.autograd.Function):
    """comp = sqrt(max(0, x)) with gsplat's add_blur_vjp derivative 0.5/(comp + 1e-6)
    (Appendix A.1 backward, `v_sqr = v_comp*0.5/(comp+1e-6)`)."""

    @staticmethod
    def forward(ctx, x):
        comp = torch.sqrt(torch.clamp_min(x, 0.0))
        ctx.save_for_backward(comp)
        return comp

    @staticmethod
    def backward(ctx, v):
        (comp,) = ctx.saved_tensors
        return v * 0.5 / (comp + 1e-6)


# --------------------------------------------------------------------------- A.1 projection
def fully_fused_projection(
    means: Tensor,  # [N,3]
    quats: Tensor,  # [N,4]
    scales: Tensor,  # [N,3]
    viewmats: Tensor,  # [C,4,4]
    Ks: Tensor,  # [C,3,3]
    width: int,
    height: int,
    eps2d: float = 0.3,
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    radius_clip: float = 0.0,
    calc_compensations: bool = False,
) -> Tuple[Tensor, Tensor, Tensor, Tensor, Optional[Tensor]]:
    """Appendix A.1.  Returns radii [C,N] int32, means2d [C,N,2], depths [C,N], conics [C,N,3],
    compensations [C,N] or None.  Culled rows (radii == 0) are zero in every output."""
    W, H = float(width), float(height)
    R = viewmats[:, :3, :3]
    t = viewmats[:, :3, 3]
    mean_c = torch.einsum("cij,nj->cni", R, means) + t[:, None, :]
    Rq = quat_to_rotmat(quats)
    M = Rq * scales[:, None, :]
    covar = M @ M.transpose(-1, -2)
    covar_c = torch.einsum("cij,njk,clk->cnil", R, covar, R)

    x, y, z = mean_c.unbind(-1)
    ok = (z >= near_plane) & (z <= far_plane)
    zs = torch.where(ok, z, torch.ones_like(z))
    fx, fy = Ks[:, 0, 0][:, None], Ks[:, 1, 1][:, None]
    cx, cy = Ks[:, 0, 2][:, None], Ks[:, 1, 2][:, None]
    tan_fovx = 0.5 * W / fx
    tan_fovy = 0.5 * H / fy
    lim_x_pos = (W - cx) / fx + 0.3 * tan_fovx
    lim_x_neg = cx / fx + 0.3 * tan_fovx
    lim_y_pos = (H - cy) / fy + 0.3 * tan_fovy
    lim_y_neg = cy / fy + 0.3 * tan_fovy
    rz = 1.0 / zs
    rz2 = rz * rz
    tx = zs * torch.minimum(lim_x_pos, torch.maximum(-lim_x_neg, x * rz))
    ty = zs * torch.minimum(lim_y_pos, torch.maximum(-lim_y_neg, y * rz))
    zero = torch.zeros_like(zs)
    J = torch.stack(
        [
            torch.stack([fx * rz, zero, -fx * tx * rz2], dim=-1),
            torch.stack([zero, fy * rz, -fy * ty * rz2], dim=-1),
        ],
        dim=-2,
    )  # [C,N,2,3]
    cov2d = J @ covar_c @ J.transpose(-1, -2)
    mean2d = torch.stack([fx * x * rz + cx, fy * y * rz + cy], dim=-1)

    c00, c01, c11 = cov2d[..., 0, 0], cov2d[..., 0, 1], cov2d[..., 1, 1]
    det_orig = c00 * c11 - c01 * c01
    c00 = c00 + eps2d
    c11 = c11 + eps2d
    det = c00 * c11 - c01 * c01
    ok = ok & (det > 0)
    dets = torch.where(ok, det, torch.ones_like(det))
    comp = _GsplatCompensation.apply(det_orig / dets)
    conic = torch.stack([c11 / dets, -c01 / dets, c00 / dets], dim=-1)

    b = 0.5 * (c00 + c11)
    v1 = b + torch.sqrt(torch.clamp_min(b * b - dets, 0.01))
    radius = torch.ceil(3.0 * torch.sqrt(v1)).detach()
    ok = ok & (radius > radius_clip)
    mx, my = mean2d.detach().unbind(-1)
    inside = ~((mx + radius <= 0) | (mx - radius >= W) | (my + radius <= 0) | (my - radius >= H))
    ok = ok & inside

    radii = torch.where(ok, radius, torch.zeros_like(radius)).to(torch.int32)
    okf = ok[..., None]
    means2d = torch.where(okf, mean2d, torch.zeros_like(mean2d))
    depths = torch.where(ok, z, torch.zeros_like(z))
    conics = torch.where(okf, conic, torch.zeros_like(conic))
    comps = torch.where(ok, comp, torch.zeros_like(comp)) if calc_compensations else None
    return radii, means2d, depths, conics, comps


# --------------------------------------------------------------------------- A.3 / A.4 integer stages
def isect_tiles(
    means2d: Tensor,  # [C,N,2]
    radii: Tensor,  # [C,N] int
    depths: Tensor,  # [C,N]
    tile_size: int,
    tile_width: int,
    tile_height: int,
) -> Tuple[Tensor, Tensor, Tensor]:
    """Appendix A.3.  fp32 arithmetic exactly as the device does it (inputs are cast to float32).
    Returns tiles_per_gauss [C,N] int32, isect_ids [I] int64 (sorted), flatten_ids [I] int32.
    The sort is a *stable* ascending sort on the low 32+tile_n_bits+cam_n_bits bits."""
    C, N = radii.shape
    m = means2d.detach().to(torch.float32).cpu().numpy().reshape(C * N, 2)
    r = radii.detach().cpu().numpy().reshape(C * N).astype(np.int64)
    d = depths.detach().to(torch.float32).cpu().numpy().reshape(C * N)
    ts = np.float32(tile_size)
    tr = r.astype(np.float32) / ts
    tx = m[:, 0] / ts
    ty = m[:, 1] / ts
    with np.errstate(invalid="ignore"):
        x0 = np.clip(np.floor(tx - tr), 0, tile_width).astype(np.int64)
        x1 = np.clip(np.ceil(tx + tr), 0, tile_width).astype(np.int64)
        y0 = np.clip(np.floor(ty - tr), 0, tile_height).astype(np.int64)
        y1 = np.clip(np.ceil(ty + tr), 0, tile_height).astype(np.int64)
    vis = r > 0
    tiles = np.where(vis, (y1 - y0) * (x1 - x0), 0).astype(np.int64)
    n_tiles = tile_width * tile_height
    tile_n_bits = int(math.floor(math.log2(n_tiles))) + 1
    cam_n_bits = int(math.floor(math.log2(C))) + 1

    idx = np.nonzero(tiles)[0]
    cnt = tiles[idx]
    total = int(cnt.sum())
    flat = np.repeat(idx, cnt)
    start = np.repeat(np.cumsum(cnt) - cnt, cnt)
    local = np.arange(total, dtype=np.int64) - start
    w = np.repeat(x1[idx] - x0[idx], cnt)
    ti = np.repeat(y0[idx], cnt) + local // np.maximum(w, 1)
    tj = np.repeat(x0[idx], cnt) + local % np.maximum(w, 1)
    tile_id = ti * tile_width + tj
    cid = flat // N
    depth_bits = d.view(np.int32).astype(np.int64)[flat] & 0xFFFFFFFF
    keys = (cid << (32 + tile_n_bits)) | (tile_id << 32) | depth_bits
    used_bits = 32 + tile_n_bits + cam_n_bits
    mask = (1 << used_bits) - 1
    order = np.argsort(keys & mask, kind="stable")
    isect_ids = torch.from_numpy(keys[order].astype(np.int64))
    flatten_ids = torch.from_numpy(flat[order].astype(np.int32))
    tiles_per_gauss = torch.from_numpy(tiles.astype(np.int32).reshape(C, N))
    return tiles_per_gauss, isect_ids, flatten_ids


def isect_offset_encode(isect_ids: Tensor, C: int, tile_width: int, tile_height: int) -> Tensor:
    """Appendix A.4: offsets[c,ty,tx] = lower bound of (cam,tile) in the sorted keys."""
    n_tiles = tile_width * tile_height
    tile_n_bits = int(math.floor(math.log2(n_tiles))) + 1
    k = isect_ids.cpu().numpy() >> 32
    cid = k >> tile_n_bits
    tid = k & ((1 << tile_n_bits) - 1)
    lin = cid * n_tiles + tid
    off = np.searchsorted(lin, np.arange(C * n_tiles), side="left").astype(np.int32)
    return torch.from_numpy(off.reshape(C, tile_height, tile_width))


# --------------------------------------------------------------------------- A.5 rasterize
def rasterize_to_pixels(
    means2d: Tensor,  # [C,N,2]
    conics: Tensor,  # [C,N,3]
    colors: Tensor,  # [C,N,D]
    opacities: Tensor,  # [C,N]
    width: int,
    height: int,
    tile_size: int,
    isect_offsets: Tensor,  # [C,TH,TW] int32
    flatten_ids: Tensor,  # [I] int32
    backgrounds: Optional[Tensor] = None,  # [C,D]
    chunk: int = 2048,
) -> Tuple[Tensor, Tensor, Tensor]:
    """Appendix A.5 (forward); autograd over it gives A.6.  Returns render_colors [C,H,W,D],
    render_alphas [C,H,W,1], last_ids [C,H,W] int32."""
    C, TH, TW = isect_offsets.shape
    D = colors.shape[-1]
    dt = means2d.dtype
    ts = tile_size
    n_isects = int(flatten_ids.shape[0])
    fids = flatten_ids.to(torch.int64)
    n_tiles = C * TH * TW
    offs = isect_offsets.reshape(-1).tolist() + [n_isects]
    sizes = [offs[t + 1] - offs[t] for t in range(n_tiles)]
    # gather every intersection's operands ONCE, then split per tile: keeps autograd's backward O(I)
    # (a per-tile index/slice would zero-fill an [N]- or [I]-sized gradient per tile)
    xy_t = torch.split(means2d.reshape(-1, 2)[fids], sizes)
    co_t = torch.split(conics.reshape(-1, 3)[fids], sizes)
    col_t = torch.split(colors.reshape(-1, D)[fids], sizes)
    op_t = torch.split(opacities.reshape(-1)[fids], sizes)

    jj, ii = torch.meshgrid(torch.arange(ts), torch.arange(ts), indexing="xy")  # [ts,ts] x,y
    jj = jj.reshape(-1)
    ii = ii.reshape(-1)
    P = ts * ts
    zero_col = torch.zeros(P, D, dtype=dt)
    one_T = torch.ones(P, dtype=dt)
    tiles_c, tiles_T, tiles_last = [], [], []
    for t_lin in range(n_tiles):
        s = offs[t_lin]
        e = offs[t_lin + 1]
        tyi = (t_lin % (TH * TW)) // TW
        txi = t_lin % TW
        px = (txi * ts + jj).to(dt) + 0.5
        py = (tyi * ts + ii).to(dt) + 0.5
        inside = ((txi * ts + jj) < width) & ((tyi * ts + ii) < height)
        out = zero_col
        T = one_T
        done = ~inside
        last = torch.zeros(P, dtype=torch.int64)
        for cs in range(s, e, chunk):
            ce = min(e, cs + chunk)
            lo, hi = cs - s, ce - s
            xy = xy_t[t_lin][lo:hi]
            co = co_t[t_lin][lo:hi]
            dx = xy[None, :, 0] - px[:, None]
            dy = xy[None, :, 1] - py[:, None]
            sigma = 0.5 * (co[None, :, 0] * dx * dx + co[None, :, 2] * dy * dy) + co[None, :, 1] * dx * dy
            alpha = torch.clamp(op_t[t_lin][lo:hi][None, :] * torch.exp(-sigma), max=ALPHA_MAX)
            valid = (sigma >= 0) & (alpha >= ALPHA_MIN)
            a_eff = torch.where(valid, alpha, torch.zeros_like(alpha))
            om = 1.0 - a_eff
            Tincl = torch.cumprod(om, dim=1) * T[:, None]
            Texcl = torch.cat([T[:, None], Tincl[:, :-1]], dim=1)
            stop = Tincl <= T_MIN
            include = valid & ~stop & ~done[:, None]
            w = torch.where(include, a_eff * Texcl, torch.zeros_like(a_eff))
            out = out + w @ col_t[t_lin][lo:hi]
            T = T * torch.where(include, om, torch.ones_like(om)).prod(dim=1)
            kidx = torch.arange(cs, ce)[None, :].expand_as(include)
            last = torch.maximum(last, torch.where(include, kidx, torch.zeros_like(kidx)).max(dim=1).values)
            done = done | stop[:, -1]
            if bool(done.all()):
                break
        tiles_c.append(out)
        tiles_T.append(T)
        tiles_last.append(last)

    def assemble(lst, tail):
        x = torch.stack(lst, dim=0).reshape((C, TH, TW, ts, ts) + tail)
        perm = (0, 1, 3, 2, 4) + tuple(range(5, 5 + len(tail)))
        x = x.permute(*perm).reshape((C, TH * ts, TW * ts) + tail)
        return x[:, :height, :width]

    rc = assemble(tiles_c, (D,))
    Tf = assemble(tiles_T, ())
    last_ids = assemble(tiles_last, ()).to(torch.int32)
    alphas = (1.0 - Tf)[..., None]
    if backgrounds is not None:
        rc = rc + Tf[..., None] * backgrounds[:, None, None, :]
    return rc, alphas, last_ids


# --------------------------------------------------------------------------- A.0 orchestration
def rasterization(
    means: Tensor,
    quats: Tensor,
    scales: Tensor,
    opacities: Tensor,
    colors: Tensor,
    viewmats: Tensor,
    Ks: Tensor,
    width: int,
    height: int,
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    radius_clip: float = 0.0,
    eps2d: float = 0.3,
    sh_degree: Optional[int] = None,
    packed: bool = False,
    tile_size: int = 16,
    backgrounds: Optional[Tensor] = None,
    render_mode: str = "RGB",
    sparse_grad: bool = False,
    absgrad: bool = False,
    rasterize_mode: str = "classic",
    channel_chunk: int = 32,
) -> Tuple[Tensor, Tensor, Dict]:
    """Appendix A.0 -- the signature the reference calls at R/models/trainers/base.py:393-408.
    `packed` only changes gsplat's internal layout, not the rendered result; this restatement
    always runs the dense [C,N] layout and returns dense meta.  `absgrad` is honoured by
    `meta["means2d"].absgrad` NOT being available here: tests obtain the abs-gradient oracle
    from `means2d_absgrad_oracle` below."""
    N = means.shape[0]
    C = viewmats.shape[0]
    width, height = int(width), int(height)
    assert means.shape == (N, 3) and quats.shape == (N, 4) and scales.shape == (N, 3)
    assert opacities.shape == (N,), opacities.shape
    assert viewmats.shape == (C, 4, 4) and Ks.shape == (C, 3, 3)
    assert render_mode in ("RGB", "D", "ED", "RGB+D", "RGB+ED"), render_mode
    assert rasterize_mode in ("classic", "antialiased")

    radii, means2d, depths, conics, comps = fully_fused_projection(
        means, quats, scales, viewmats, Ks, width, height, eps2d, near_plane, far_plane,
        radius_clip, calc_compensations=(rasterize_mode == "antialiased"),
    )
    opac = opacities[None, :].expand(C, N)
    if comps is not None:
        opac = opac * comps

    if sh_degree is None:
        assert colors.dim() in (2, 3)
        cols = colors[None].expand(C, -1, -1) if colors.dim() == 2 else colors
    else:
        assert (sh_degree + 1) ** 2 <= colors.shape[-2], colors.shape
        campos = torch.linalg.inv(viewmats)[:, :3, 3]
        dirs = means[None, :, :] - campos[:, None, :]
        shs = colors[None].expand(C, -1, -1, -1) if colors.dim() == 3 else colors
        cols = spherical_harmonics(sh_degree, dirs, shs, masks=radii > 0)
        cols = torch.clamp_min(cols + 0.5, 0.0)

    if render_mode in ("RGB+D", "RGB+ED"):
        cols = torch.cat([cols, depths[..., None]], dim=-1)
        if backgrounds is not None:
            backgrounds = torch.cat([backgrounds, torch.zeros_like(backgrounds[:, :1])], dim=-1)
    elif render_mode in ("D", "ED"):
        cols = depths[..., None]
        if backgrounds is not None:
            backgrounds = torch.zeros_like(backgrounds[:, :1])

    TW = math.ceil(width / tile_size)
    TH = math.ceil(height / tile_size)
    tiles_per_gauss, isect_ids, flatten_ids = isect_tiles(means2d, radii, depths, tile_size, TW, TH)
    isect_offsets = isect_offset_encode(isect_ids, C, TW, TH)
    rc, ra, last_ids = rasterize_to_pixels(
        means2d, conics, cols, opac, width, height, tile_size, isect_offsets, flatten_ids, backgrounds
    )
    if render_mode in ("ED", "RGB+ED"):
        rc = torch.cat([rc[..., :-1], rc[..., -1:] / ra.clamp(min=1e-10)], dim=-1)

    meta = {
        "radii": radii, "means2d": means2d, "depths": depths, "conics": conics,
        "opacities": opac, "colors": cols, "tile_width": TW, "tile_height": TH,
        "tiles_per_gauss": tiles_per_gauss, "isect_ids": isect_ids, "flatten_ids": flatten_ids,
        "isect_offsets": isect_offsets, "last_ids": last_ids, "width": width, "height": height,
        "tile_size": tile_size, "n_cameras": C,
    }
    return rc, ra, meta


def means2d_absgrad_oracle(
    means2d: Tensor, conics: Tensor, colors: Tensor, opacities: Tensor,
    width: int, height: int, tile_size: int, isect_offsets: Tensor, flatten_ids: Tensor,
    v_colors: Tensor, v_alphas: Tensor, backgrounds: Optional[Tensor] = None,
) -> Tensor:
    """Appendix A.6 `v_means2d_abs`: sum over pixels of |d loss_pixel / d means2d| (abs per pixel,
    before any reduction).  Computed by one autograd pass per pixel-slot of the tiles: pixels at
    the same intra-tile position belong to different tiles, but a Gaussian may cover several
    tiles, so the per-pixel gradient is isolated with one backward per (tile, pixel) -- only
    usable for SMALL images (tests use <= 32x32)."""
    C = isect_offsets.shape[0]
    m = means2d.detach().clone().requires_grad_(True)
    rc, ra, _ = rasterize_to_pixels(
        m, conics.detach(), colors.detach(), opacities.detach(), width, height, tile_size,
        isect_offsets, flatten_ids, backgrounds,
    )
    per_pixel = (rc * v_colors).sum(-1) + (ra * v_alphas).sum(-1)  # [C,H,W]
    acc = torch.zeros_like(m)
    for c in range(C):
        for i in range(height):
            for j in range(width):
                (g,) = torch.autograd.grad(per_pixel[c, i, j], m, retain_graph=True)
                acc += g.abs()
    return acc
