"""Host-side operators over the C ABI (include/gsplat_mi355x.h), mirroring `gsplat.cuda._wrapper`.

The reference imports exactly one symbol from this module path:
    from gsplat.cuda._wrapper import spherical_harmonics
(/root/reference/3D-Reconstruction-main/models/gaussians/basics.py:15; called at
models/gaussians/vanilla.py:369, models/nodes/rigid.py:486, smpl.py:475, deformable.py:83,
models/gaussians/deformgs.py:136, pvg.py:399).  The stage-level operators below (projection,
tile intersection, sort, offsets, rasterize) are what `gsplat.rendering.rasterization` composes and
what the parity tests call one by one.
"""
from __future__ import annotations

import ctypes
import math
import threading
from typing import Optional, Tuple

import torch
from torch import Tensor

from .. import _backend as B


# ----------------------------------------------------------------------------- spherical harmonics
class _SphericalHarmonics(torch.autograd.Function):
    @staticmethod
    def forward(ctx, degree: int, dirs: Tensor, coeffs: Tensor, masks: Optional[Tensor]):
        dev = B.require_gpu(dirs, coeffs, masks)
        K = coeffs.shape[-2]
        lead = dirs.shape[:-1]
        d = B.f32c(dirs).reshape(-1, 3)
        c = B.f32c(coeffs).reshape(-1, K, 3)
        m = masks.reshape(-1).to(torch.uint8).contiguous() if masks is not None else None
        n = d.shape[0]
        out = torch.empty(n, 3, device=dev, dtype=torch.float32)
        with B.device_guard(dev), B.timed("gs_sh_fwd", dev):
            B.check(B.lib().gs_sh_fwd(n, K, degree, B.ptr(d), B.ptr(c), B.ptr(m), B.ptr(out), B.stream_handle(dev)),
                    "gs_sh_fwd")
        ctx.save_for_backward(d, c, m if m is not None else torch.empty(0, device=dev))
        ctx.has_mask = m is not None
        ctx.degree, ctx.K, ctx.lead = degree, K, lead
        ctx.coeff_shape = coeffs.shape
        return out.reshape(lead + (3,))

    @staticmethod
    def backward(ctx, v_out: Tensor):
        d, c, m = ctx.saved_tensors
        m = m if ctx.has_mask else None
        dev = d.device
        n = d.shape[0]
        v = B.f32c(v_out).reshape(-1, 3)
        v_coeffs = torch.empty(n, ctx.K, 3, device=dev, dtype=torch.float32)
        need_dirs = ctx.needs_input_grad[1]
        v_dirs = torch.empty(n, 3, device=dev, dtype=torch.float32) if need_dirs else None
        with B.device_guard(dev), B.timed("gs_sh_bwd", dev):
            B.check(B.lib().gs_sh_bwd(n, ctx.K, ctx.degree, B.ptr(d), B.ptr(c), B.ptr(m), B.ptr(v), B.ptr(v_coeffs),
                                      B.ptr(v_dirs), B.stream_handle(dev)), "gs_sh_bwd")
        v_c = v_coeffs.reshape(ctx.coeff_shape) if ctx.needs_input_grad[2] else None
        v_d = v_dirs.reshape(ctx.lead + (3,)) if need_dirs else None
        return None, v_d, v_c, None


def spherical_harmonics(degrees_to_use: int, dirs: Tensor, coeffs: Tensor, masks: Optional[Tensor] = None) -> Tensor:
    """Real SH colours.  dirs [...,3] (normalised inside), coeffs [...,K,3], masks [...] -> [...,3].

    Same contract as gsplat's op: `(degrees_to_use+1)**2 <= K`; differentiable w.r.t. coeffs and dirs;
    masked-out rows are zero; no +0.5 (callers add it, vanilla.py:370)."""
    assert (degrees_to_use + 1) ** 2 <= coeffs.shape[-2], coeffs.shape
    assert dirs.shape[:-1] == coeffs.shape[:-2], (dirs.shape, coeffs.shape)
    assert dirs.shape[-1] == 3, dirs.shape
    assert coeffs.shape[-1] == 3, coeffs.shape
    if masks is not None:
        assert masks.shape == dirs.shape[:-1], masks.shape
    return _SphericalHarmonics.apply(int(degrees_to_use), dirs, coeffs, masks)


# ----------------------------------------------------------------------------- projection (+colour fusion)
def _camera_centers(viewmats: Tensor) -> Tensor:
    """Camera centres -inv(R) t of world->camera matrices, by the adjugate (tiny, differentiable)."""
    R, t = viewmats[:, :3, :3], viewmats[:, :3, 3]
    r0, r1, r2 = R[:, 0], R[:, 1], R[:, 2]
    c12 = torch.linalg.cross(r1, r2)
    c20 = torch.linalg.cross(r2, r0)
    c01 = torch.linalg.cross(r0, r1)
    det = (r0 * c12).sum(-1, keepdim=True)
    return -(t[:, 0:1] * c12 + t[:, 1:2] * c20 + t[:, 2:3] * c01) / det


def _make_params(N, C, width, height, tile_size, eps2d, near_plane, far_plane, radius_clip, calc_comp,
                 color_mode, color_dim, sh_degree, sh_K, depth_channel) -> B.ProjParams:
    p = B.ProjParams()
    p.N, p.C, p.width, p.height = N, C, width, height
    p.tile_size = tile_size
    p.tile_w = math.ceil(width / tile_size)
    p.tile_h = math.ceil(height / tile_size)
    p.eps2d, p.near_plane, p.far_plane, p.radius_clip = eps2d, near_plane, min(far_plane, 3.0e38), radius_clip
    p.calc_compensations = int(calc_comp)
    p.color_mode, p.color_dim = color_mode, color_dim
    p.sh_degree, p.sh_K = (sh_degree if sh_degree is not None else 0), sh_K
    p.depth_channel = int(depth_channel)
    # SH coefficient rows of the forward: 0 = by camera count; the trajectory harness asks for "on demand" (2): its
    # neighbouring views see the same quarter of the Gaussians however many it renders per call
    p.sh_rows = int(getattr(_tls, "sh_rows", 0))
    return p


# How the backward of in-rasterization SH colours (GS_COLOR_SH) is computed:
#   False  fused: gs_projection_bwd accumulates v_sh over the cameras in registers (one kernel)
#   True   split: gs_projection_bwd_split (geometry + the packed 12-B colour gradients of the visible pairs) followed by
#          gs_sh_bwd_packed (v_sh + the view-direction part of v_means from the packed message)
# A gradient exchange that is active on the calling thread (gsplat.distributed.GradientExchange, mode "colour") always
# takes the split route: the packed message is what crosses the ranks.
SPLIT_SH_BACKWARD = False

_tls = threading.local()


def current_exchange():
    """The gsplat.distributed.GradientExchange active on this thread (None outside its `with` block)."""
    return getattr(_tls, "exchange", None)


def colour_message_layout(C: int, N: int, cap: int):
    """Byte offsets {campos, words, prefix, vals, total} of the packed colour-gradient message (include/gsplat_mi355x.h)."""
    out = (ctypes.c_int64 * 5)()
    B.check(B.lib().gs_colour_message_layout(C, N, cap, out), "gs_colour_message_layout")
    return dict(zip(("campos", "words", "prefix", "vals", "total"), (int(v) for v in out)))


class _ProjectSplats(torch.autograd.Function):
    """means/quats/scales/opacities/colours/cameras -> splat records [C,N,16] + means2d [C,N,2]
    (+ radii, tiles_per_gauss, non-differentiable)."""

    @staticmethod
    def forward(ctx, means, quats, scales, opacities, colors, viewmats, Ks, campos, params: B.ProjParams,
                covars=None):
        dev = B.require_gpu(means, quats, scales, opacities, colors, viewmats, Ks, campos, covars)
        N, C = params.N, params.C
        means, opacities = B.f32c(means), B.f32c(opacities)
        if covars is not None:  # [N,6] upper triangle; quats / scales are not used
            covars, quats, scales = B.f32c(covars), None, None
        else:
            quats, scales = B.f32c(quats), B.f32c(scales)
        viewmats, Ks = B.f32c(viewmats), B.f32c(Ks)
        colors = B.f32c(colors) if colors is not None else None
        campos = B.f32c(campos) if campos is not None else None
        radii = torch.empty(C, N, device=dev, dtype=torch.int32)
        tiles = torch.empty(C, N, device=dev, dtype=torch.int32)
        # the dense [C,N,2] means2d is what gradients hang on and what a dense `meta` shows; a caller that wants neither
        # (packed, no gradients: the render-only harness) sets _tls.skip_means2d and the 8 bytes per (camera, Gaussian)
        # are not written (the records hold the means of the visible rows)
        # (inside Function.forward grad mode is always off, so the guard is the inputs' requires_grad, not grad mode)
        want_grad = any(t is not None and t.requires_grad for t in (means, quats, scales, opacities, colors, viewmats,
                                                                    covars))
        skip_m2d = getattr(_tls, "skip_means2d", False) and not want_grad
        means2d = None if skip_m2d else torch.empty(C, N, 2, device=dev, dtype=torch.float32)
        # rows with radii == 0 are never written nor read (as gsplat's torch.empty projection outputs)
        splats = torch.empty(C, N, B.GS_REC, device=dev, dtype=torch.float32)
        # split SH backward: the forward also leaves each camera's visibility words; their scanned populations give
        # every visible pair its row in the packed colour-gradient message of the backward
        exchange = current_exchange()
        split = (params.color_mode == B.GS_COLOR_SH and covars is None and N > 0 and want_grad
                 and (SPLIT_SH_BACKWARD or (exchange is not None and exchange.packs_colours)))
        vis_words = vis_counts = None
        if split:
            n_words = (N + 63) // 64
            vis_words = torch.empty(C, n_words, device=dev, dtype=torch.int64)
            vis_counts = torch.empty(C, n_words, device=dev, dtype=torch.int32)
        with B.device_guard(dev), B.timed("gs_projection_fwd", dev):
            B.check(B.lib().gs_projection_fwd(
                ctypes.byref(params), B.ptr(means), B.ptr(quats), B.ptr(scales), B.ptr(opacities), B.ptr(viewmats),
                B.ptr(Ks), B.ptr(campos), B.ptr(colors), B.ptr(radii), B.ptr(means2d), B.ptr(splats), B.ptr(tiles),
                B.ptr(covars), B.ptr(vis_words), B.ptr(vis_counts), B.stream_handle(dev)), "gs_projection_fwd")
        ctx.split = None
        if split:
            # the number of visible pairs sizes the message: to the host through pinned memory behind the scan, so
            # that reading it in backward costs no synchronisation of its own (the forward's n_isects read, later on
            # this stream, already covers it); with an exchange, the largest count over the ranks
            # ... and the largest camera count: every rank lays its message out alike
            vis_prefix = torch.empty_like(vis_counts)
            stat = torch.empty(2, device=dev, dtype=torch.int64)
            wsb = int(B.lib().gs_exclusive_scan_workspace_bytes(C * n_words))
            ws = torch.empty(max(wsb, 8), device=dev, dtype=torch.uint8)
            with B.device_guard(dev), B.timed("gs_visibility_prefix", dev):
                B.check(B.lib().gs_visibility_prefix(C, n_words, B.ptr(vis_counts), B.ptr(vis_prefix), B.ptr(stat),
                                                     B.ptr(ws), wsb, B.stream_handle(dev)), "gs_visibility_prefix")
            host = torch.empty(2, dtype=torch.int64, pin_memory=True)
            ready = torch.cuda.Event()
            if exchange is not None:
                # the MAX over the ranks and the copy to the host run on the side stream: nothing on the main stream
                # (the whole intersection stage comes next) waits for the collective's latency
                main, side = torch.cuda.current_stream(dev), B.side_stream(dev)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    stat_max = exchange.max_over_ranks(stat)
                    host.copy_(stat_max, non_blocking=True)
                    ready.record(side)
                stat.record_stream(side)
            else:
                host.copy_(stat, non_blocking=True)
                ready.record(torch.cuda.current_stream(dev))
            ctx.split = (vis_prefix, host, ready)
        ctx.exchange = exchange
        ctx.params = params
        ctx.has_colors = colors is not None
        ctx.has_campos = campos is not None
        ctx.has_covars = covars is not None
        empty = torch.empty(0, device=dev)
        ctx.save_for_backward(means, quats if quats is not None else empty, scales if scales is not None else empty,
                              opacities, colors if colors is not None else empty, viewmats, Ks,
                              campos if campos is not None else empty, radii, splats,
                              covars if covars is not None else empty)
        ctx.mark_non_differentiable(radii, tiles)
        ctx.set_materialize_grads(False)
        return splats, means2d, radii, tiles

    @staticmethod
    def backward(ctx, v_splats, v_means2d, _v_radii, _v_tiles):
        means, quats, scales, opacities, colors, viewmats, Ks, campos, radii, splats, covars = ctx.saved_tensors
        covars = covars if ctx.has_covars else None
        if covars is not None:
            quats = scales = None
        p = ctx.params
        dev = means.device
        N, C = p.N, p.C
        colors = colors if ctx.has_colors else None
        campos = campos if ctx.has_campos else None
        if v_splats is None:
            v_splats = torch.zeros(C, N, B.GS_REC, device=dev, dtype=torch.float32)
        v_splats = B.f32c(v_splats)
        if v_means2d is None:
            v_means2d = torch.zeros(C, N, 2, device=dev, dtype=torch.float32)
        # accept the strided view v_splats[..., 0:2] handed back by the rasterizer without a copy
        ok_layout = (v_means2d.dtype == torch.float32 and v_means2d.stride(2) == 1 and N > 0
                     and v_means2d.stride(0) == N * v_means2d.stride(1) and v_means2d.stride(1) >= 2)
        if not ok_layout:
            v_means2d = B.f32c(v_means2d)
        ld = v_means2d.stride(1) if N > 0 else 2
        ex = ctx.exchange
        ctx.flat_geometry = None
        if ex is not None and quats is not None and scales is not None and opacities.dim() == 1:
            # one buffer for the four geometry gradients (sections padded to 256 B): ONE all-reduce instead of four
            up = lambda n: (n + 63) // 64 * 64
            o_m, o_s, o_o = up(4 * N), up(4 * N) + up(3 * N), up(4 * N) + 2 * up(3 * N)
            flat = ctx.flat_geometry = torch.empty(o_o + up(N), device=dev, dtype=torch.float32)
            v_quats = flat[:4 * N].view(N, 4)
            v_means = flat[o_m:o_m + 3 * N].view(N, 3)
            v_scales = flat[o_s:o_s + 3 * N].view(N, 3)
            v_opac = flat[o_o:o_o + N]
        else:
            v_means = torch.empty_like(means)
            v_quats = torch.empty_like(quats) if quats is not None else None
            v_scales = torch.empty_like(scales) if scales is not None else None
            v_opac = torch.empty_like(opacities)
        v_covars = torch.empty_like(covars) if covars is not None else None
        v_colors = torch.empty_like(colors) if colors is not None else None
        need_vm = ctx.needs_input_grad[5] or ctx.needs_input_grad[7]
        v_viewmats = torch.zeros(C, 4, 4, device=dev, dtype=torch.float32) if need_vm else None
        v_campos = torch.zeros(C, 3, device=dev, dtype=torch.float32) if (need_vm and campos is not None) else None
        if ctx.split is not None:
            return _ProjectSplats._backward_split(ctx, means, quats, scales, opacities, colors, viewmats, Ks, campos,
                                                  radii, splats, v_splats, v_means2d, ld, v_means, v_quats, v_scales,
                                                  v_opac, v_colors, v_viewmats, v_campos)
        with B.device_guard(dev), B.timed("gs_projection_bwd", dev):
            B.check(B.lib().gs_projection_bwd(
                ctypes.byref(p), B.ptr(means), B.ptr(quats), B.ptr(scales), B.ptr(opacities), B.ptr(viewmats),
                B.ptr(Ks), B.ptr(campos), B.ptr(colors), B.ptr(radii), B.ptr(splats), v_means2d.data_ptr(), ld,
                B.ptr(v_splats), B.ptr(v_means), B.ptr(v_quats), B.ptr(v_scales), B.ptr(v_opac), B.ptr(v_colors),
                B.ptr(v_viewmats), B.ptr(v_campos), B.ptr(covars), B.ptr(v_covars), B.stream_handle(dev)),
                "gs_projection_bwd")
        if ex is not None:
            # dense exchange: SUM over the ranks of every Gaussian-parameter gradient.  Per-camera inputs (viewmats,
            # [C,N,D] colours) belong to this rank's cameras and stay local.
            shared_colours = v_colors if p.color_mode in (B.GS_COLOR_N, B.GS_COLOR_SH) else None
            geometry = ([ctx.flat_geometry] if ctx.flat_geometry is not None
                        else [v_means, v_quats, v_scales, v_covars, v_opac])
            ex.all_reduce([t for t in geometry + [shared_colours] if t is not None]).wait()
        g = ctx.needs_input_grad
        return (v_means if g[0] else None, v_quats if (g[1] and v_quats is not None) else None,
                v_scales if (g[2] and v_scales is not None) else None,
                v_opac if g[3] else None, v_colors if (g[4] and v_colors is not None) else None,
                v_viewmats if g[5] else None, None, v_campos if (g[7] and v_campos is not None) else None, None,
                v_covars if (len(g) > 9 and g[9] and v_covars is not None) else None)

    @staticmethod
    def _backward_split(ctx, means, quats, scales, opacities, colors, viewmats, Ks, campos, radii, splats, v_splats,
                        v_means2d, ld, v_means, v_quats, v_scales, v_opac, v_colors, v_viewmats, v_campos):
        """Geometry kernel + packed colour-gradient message -> (all-gather over the ranks) -> SH kernel."""
        p, ex, dev = ctx.params, ctx.exchange, means.device
        N, C = p.N, p.C
        vis_prefix, host, ready = ctx.split
        ready.synchronize()  # (long complete: recorded before the forward's n_isects read)
        cap, lay_cams = int(host[0]), int(host[1])
        assert lay_cams >= C and cap >= 0, (lay_cams, C, cap)
        lay = colour_message_layout(lay_cams, N, cap)
        # (tests set POISON_MESSAGES: the sections of absent cameras and the rows beyond the visible pairs are never
        # written by the geometry kernel and must never be read by the SH kernel)
        msg = (torch.full((lay["total"],), 0xEE, device=dev, dtype=torch.uint8) if POISON_MESSAGES
               else torch.empty(lay["total"], device=dev, dtype=torch.uint8))
        with B.device_guard(dev), B.timed("gs_projection_bwd", dev):
            B.check(B.lib().gs_projection_bwd_split(
                ctypes.byref(p), B.ptr(means), B.ptr(quats), B.ptr(scales), B.ptr(opacities), B.ptr(viewmats),
                B.ptr(Ks), B.ptr(campos), B.ptr(radii), B.ptr(splats), v_means2d.data_ptr(), ld, B.ptr(v_splats),
                B.ptr(vis_prefix), B.ptr(v_means), B.ptr(v_quats), B.ptr(v_scales), B.ptr(v_opac), B.ptr(v_viewmats),
                B.ptr(msg), cap, lay_cams, B.stream_handle(dev)), "gs_projection_bwd_split")
        n_msgs, local, msgs, work = 1, 0, msg, None
        if ex is not None:
            # ONE all-gather of the messages (12 B per visible pair) instead of a 12 K-byte-per-Gaussian all-reduce of
            # v_sh; the geometry gradients (44 B per Gaussian) are SUM all-reduced behind it while the SH kernel runs
            msgs, n_msgs, local = ex.all_gather_messages(msg)
            work = ex.all_reduce([ctx.flat_geometry] if ctx.flat_geometry is not None
                                 else [v_means, v_quats, v_scales, v_opac])
        v_means_sh = torch.empty_like(v_means) if ex is not None else v_means
        with B.device_guard(dev), B.timed("gs_sh_bwd_packed", dev):
            B.check(B.lib().gs_sh_bwd_packed(
                N, p.sh_K, p.sh_degree, lay_cams, n_msgs, lay["total"], cap, B.ptr(means), B.ptr(colors), B.ptr(msgs),
                B.ptr(v_colors), B.ptr(v_means_sh), int(ex is None), local, B.ptr(v_campos), B.stream_handle(dev)),
                "gs_sh_bwd_packed")
        if work is not None:
            work.wait()
            v_means.add_(v_means_sh)
        g = ctx.needs_input_grad
        return (v_means if g[0] else None, v_quats if g[1] else None, v_scales if g[2] else None,
                v_opac if g[3] else None, v_colors if g[4] else None, v_viewmats if g[5] else None, None,
                v_campos if (g[7] and v_campos is not None) else None, None, None)


# ----------------------------------------------------------------------------- integer stages
def exclusive_scan_i32(x: Tensor) -> Tuple[Tensor, Tensor]:
    """Exclusive prefix sum of an int32 tensor (flattened).  Returns (scan int32, total int64[1])."""
    dev = B.require_gpu(x)
    x = x.contiguous().reshape(-1)
    assert x.dtype == torch.int32
    n = x.numel()
    out = torch.empty_like(x)
    total = torch.empty(1, device=dev, dtype=torch.int64)
    wsb = int(B.lib().gs_exclusive_scan_workspace_bytes(n))
    ws = torch.empty(max(wsb, 8), device=dev, dtype=torch.uint8)
    with B.device_guard(dev), B.timed("gs_exclusive_scan_i32", dev):
        B.check(B.lib().gs_exclusive_scan_i32(B.ptr(x), B.ptr(out), n, B.ptr(total), B.ptr(ws), wsb,
                                              B.stream_handle(dev)), "gs_exclusive_scan_i32")
    return out, total


def radix_sort_pairs(keys: Tensor, vals: Tensor, end_bit: int, begin_bit: int = 0,
                     label: str = "gs_radix_sort_pairs_u64_u32", n_dev: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """Stable ascending sort of (int64 key, int32 value) pairs on key bits [begin_bit, end_bit).

    With `n_dev` (int64 device scalar) only the first min(n_dev, len) pairs are sorted (the rest of the returned
    buffers is undefined); the host does not need to know the count to enqueue the sort."""
    dev = B.require_gpu(keys, vals)
    assert keys.dtype == torch.int64 and vals.dtype == torch.int32 and keys.shape == vals.shape and keys.dim() == 1
    keys, vals = keys.contiguous(), vals.contiguous()
    n = keys.numel()
    if n == 0:
        return keys, vals
    kb, vb = torch.empty_like(keys), torch.empty_like(vals)
    wsb = int(B.lib().gs_radix_sort_workspace_bytes(n))
    ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
    in_b = ctypes.c_int32(0)
    with B.device_guard(dev), B.timed(label, dev):
        B.check(B.lib().gs_radix_sort_pairs_u64_u32_ex(n, B.ptr(n_dev), begin_bit, end_bit, B.ptr(keys), B.ptr(vals),
                                                       B.ptr(kb), B.ptr(vb), B.ptr(ws), wsb, ctypes.byref(in_b),
                                                       B.stream_handle(dev)), "gs_radix_sort_pairs_u64_u32_ex")
    return (kb, vb) if in_b.value else (keys, vals)


def _key_bits(n_cameras: int, tile_width: int, tile_height: int) -> Tuple[int, int]:
    tile_n_bits = int(math.floor(math.log2(tile_width * tile_height))) + 1
    cam_n_bits = int(math.floor(math.log2(n_cameras))) + 1
    return tile_n_bits, cam_n_bits


def isect_tiles_from_splats(splats: Tensor, tiles_per_gauss: Tensor, tile_width: int, tile_height: int,
                            sort: bool = True) -> Tuple[Tensor, Tensor]:
    """gsplat's own isect_tiles decomposition from splat records (tile boxes were computed by the projection kernel):
    cumsum -> emission in ascending flat row id -> stable radix sort on the (camera|tile|depth) key (Appendix A.3).
    Returns (isect_ids int64 [I], flatten_ids int32 [I]).  The rasterization path uses `isect_tiles_binned`, which
    produces the same arrays; this one is the generic route (any tile count) and its cross-check."""
    dev = B.require_gpu(splats, tiles_per_gauss)
    C, N = tiles_per_gauss.shape
    cum, total = exclusive_scan_i32(tiles_per_gauss)
    n_isects = int(total.item())  # host sync (gsplat has the same one)
    isect_ids = torch.empty(n_isects, device=dev, dtype=torch.int64)
    flatten_ids = torch.empty(n_isects, device=dev, dtype=torch.int32)
    if n_isects == 0:
        return isect_ids, flatten_ids
    with B.device_guard(dev), B.timed("gs_isect_emit", dev):
        B.check(B.lib().gs_isect_emit(C, N, tile_width, tile_height, B.ptr(splats), B.ptr(tiles_per_gauss),
                                      B.ptr(cum), B.ptr(isect_ids), B.ptr(flatten_ids), B.stream_handle(dev)),
                "gs_isect_emit")
    if sort:
        tile_n_bits, cam_n_bits = _key_bits(C, tile_width, tile_height)
        isect_ids, flatten_ids = radix_sort_pairs(isect_ids, flatten_ids, 32 + tile_n_bits + cam_n_bits)
    return isect_ids, flatten_ids


def set_absgrad_accumulate(on: bool) -> None:
    """While on, rasterizer passes created on this thread ADD their abs-gradient to `means2d.absgrad` in backward
    instead of replacing it (several passes sharing one means2d: the >4-channel path)."""
    _tls.absgrad_accumulate = bool(on)


def take_front_sort():
    """The FrontSort of this thread's last _RasterizeSplats.forward (None if its lists were sorted whole)."""
    fs = getattr(_tls, "front_sort", None)
    _tls.front_sort = None
    return fs


FRONT_SORT = True  # rasterization() sorts long tile lists front-first (False: every list whole, as gsplat does)
FRONT_K_TARGET = 1024  # leading entries of a long tile list that the front sort orders (gs_isect_tile_sort_front)


class FrontSort:
    """State of a front-sorted intersection list (gs_isect_tile_sort_front): what is needed to sort the tiles the
    rasterizer flags (`redo`) and to complete every tile when the caller reads the arrays (`finalize`)."""

    def __init__(self, geom, offsets, order, n_long, pairs, isect_ids, flatten_ids, sorted_counts, redo_flags=None):
        self.geom, self.offsets, self.order, self.n_long = geom, offsets, order, n_long
        self.pairs, self.isect_ids, self.flatten_ids, self.sorted_counts = pairs, isect_ids, flatten_ids, sorted_counts
        if redo_flags is None:  # (the composite prepare call zeroes the one it is given)
            redo_flags = torch.zeros(sorted_counts.numel() + 1, device=sorted_counts.device, dtype=torch.int32)
        self.redo_flags = redo_flags
        self.size_classes_host = None  # ctypes int32[3], when the host has read the size classes
        self.complete = False

    def redo(self) -> None:
        C, tw, th = self.geom
        dev = self.isect_ids.device
        with B.device_guard(dev), B.timed("gs_isect_tile_sort_redo", dev):
            B.check(B.lib().gs_isect_tile_sort_redo(C, tw, th, B.ptr(self.offsets), self.isect_ids.numel(),
                                                    B.ptr(self.order), B.ptr(self.n_long), B.ptr(self.redo_flags),
                                                    B.ptr(self.pairs), B.ptr(self.isect_ids), B.ptr(self.flatten_ids),
                                                    B.ptr(self.sorted_counts), B.stream_handle(dev)),
                    "gs_isect_tile_sort_redo")

    def finalize(self) -> None:
        """Sort every segment whole, in place (the already-sorted prefixes are rewritten with the same values)."""
        if self.complete:
            return
        self.complete = True
        C, tw, th = self.geom
        dev = self.isect_ids.device
        with B.device_guard(dev):
            B.check(B.lib().gs_isect_tile_sort(C, tw, th, B.ptr(self.offsets), self.isect_ids.numel(),
                                               B.ptr(self.order), B.ptr(self.n_long), B.ptr(self.pairs),
                                               B.ptr(self.isect_ids), B.ptr(self.flatten_ids), B.stream_handle(dev)),
                    "gs_isect_tile_sort")
        self.pairs = None  # (stream-ordered: the allocator reuses it only behind the sort)


_ISECT_LABELS = ("gs_isect_bin_count", "gs_isect_bin_offsets", "gs_raster_tile_order", "gs_isect_bin_scatter",
                 "gs_isect_tile_sort")
_RASTER_FWD_LABELS = ("gs_rasterize_fwd", "gs_rasterize_fwd_redo", "gs_isect_tile_sort_redo")
_host_scalars = threading.local()
_post_slots = threading.local()
POISON_MESSAGES = False  # tests only: fill the colour-gradient message with 0xEE before the kernel writes it
POSTED_RESULTS = True  # n_isects and the size classes are posted into a polled host slot by the kernels themselves


def _host_scalar() -> Tensor:
    """A pinned int64 the composite prepare call writes n_isects into (one per thread: the call synchronises before it
    returns, so the value is consumed before the next call can overwrite it)."""
    t = getattr(_host_scalars, "t", None)
    if t is None:
        t = _host_scalars.t = torch.zeros(3, dtype=torch.int64).pin_memory()  # n_isects | 3 x int32 size classes
    return t


class _PostSlot:
    """One host slot for posted results (include/gsplat_mi355x.h, "posted results"): int64 n_isects | int32[3] size
    classes, in portable pinned memory (one address for every device).  Owned by the thread that made it
    (threading.local) and freed with it; a wait on the slot is synchronous, so a thread needs exactly one."""

    def __init__(self):
        p = ctypes.c_void_p()
        B.check(B.lib().gs_host_post_alloc(ctypes.byref(p), 64), "gs_host_post_alloc")
        self.addr = p.value
        self.views = (p.value, (ctypes.c_int64 * 1).from_address(p.value), (ctypes.c_int32 * 3).from_address(p.value + 8))

    def __del__(self):
        try:
            if self.addr:
                B.lib().gs_host_post_free(ctypes.c_void_p(self.addr))
                self.addr = None
        except Exception:  # interpreter shutdown: the process's pinned memory goes with it
            pass


def _post_slot():
    """(address, int64 view, int32 view) of the calling thread's host slot."""
    slot = getattr(_post_slots, "slot", None)
    if slot is None:
        slot = _post_slots.slot = _PostSlot()
    return slot.views


SLICES_TOTAL = 512        # count-pass workgroups over all cameras (row slices x cameras)
SCATTER_REGIONS = 0       # bands of tile rows in the region scatter; 0: chosen per shape (see isect_tiles_binned)
SCATTER_WORKGROUPS = 256  # ... and about this many scatter workgroups (slices are grouped to get there)
FUSED_LAUNCHES = True  # the composite entry points (one FFI call per launch sequence); False: one call per kernel
_bin_max_tiles = []


def bin_max_tiles() -> int:
    """Tiles per camera the binned pipeline takes (gs_isect_bin_max_tiles; constant of the library)."""
    if not _bin_max_tiles:
        _bin_max_tiles.append(int(B.lib().gs_isect_bin_max_tiles()))
    return _bin_max_tiles[0]


# the front-first sort (+ redo pass) is used only when a tile list exceeds 2048 entries (size class 0; 1: 1024 entries)
FRONT_FROM_CLASS = 0
STAGED_SCATTER = True  # region scatter sorts a workgroup's pairs by tile in LDS and writes whole runs (False: 8-B stores)
REGION_SCATTER = True  # scatter from the count pass's visible lists by tile band (False: by row slice, from the records)
# rasterization() builds CULLED tile lists when it renders with gradients or packed (gs_isect_bin_count, "Culled
# lists"): pairs whose splat cannot reach alpha 1/255 anywhere on the tile are left out.  No rendered value or gradient
# changes; meta["isect_ids" / "flatten_ids" / "isect_offsets"] are then produced on first read, in gsplat's exact form.
CULL_LISTS = True
# gradient records up to this size are cleared on the main stream; larger ones on the side stream, behind the forward
SIDE_STREAM_MIN_BYTES = 32 << 20


# Speculative list sizes (include/gsplat_mi355x.h, "device-side n_isects"): rasterization() sizes the pair / id arrays
# from the LAST call of the same shape (x SPEC_MARGIN) and issues scatter, sort and the forward rasterizer without waiting
# for this call's count; the posted count is looked at once everything is queued, and a call that does not fit -- or
# that has a tile list the speculative launches could not sort -- is redone through the exact-size path.  Same bits
# either way.  Only taken when the previous call of the shape had no tile list above 2048 entries (those shapes are
# GPU-bound and want the front-first sort, whose launches depend on the host knowing the size classes).
# Every size class of the sort is launched on speculative lists -- a class that was not launched would leave its lists'
# ids unwritten and the rasterizer would gather through them (tried with a host hint of the expected classes: a GPU
# fault, profiles/r03_ab_runs.txt item 2) -- which costs two launches that find nothing to do; on GPU-bound shapes
# that is all speculation changes (configs[1]: +0.004 ms), so it is used up to SPEC_MAX_ROWS (camera, Gaussian) rows.
SPECULATE_SIZES = True
SPEC_MARGIN = 1.25
SPEC_MAX_ROWS = 3_500_000
_spec_tls = threading.local()


def _spec_table() -> dict:
    t = getattr(_spec_tls, "table", None)
    if t is None:
        t = _spec_tls.table = {}
    return t


class SpecLists:
    """Tile lists built with a speculative capacity (gs_isect_bin_build_dev).  `resolve()` -- call it after the launches
    that read the lists have been queued -- waits for the posted count (already there) and says whether the speculation
    held; if not, `redo()` builds the lists again through the exact-size path."""

    def __init__(self, capacity, total, isect_ids, flatten_ids, offsets, order, wait_posted, build_exact, key):
        self.capacity, self.total = capacity, total
        self.isect_ids, self.flatten_ids, self.offsets, self.order = isect_ids, flatten_ids, offsets, order
        self._wait, self._build, self._key = wait_posted, build_exact, key
        self.n_isects = self.classes = None

    def resolve(self) -> bool:
        self.n_isects, self.classes = self._wait()
        _spec_table()[self._key] = (self.n_isects, tuple(self.classes))
        ok = 0 < self.n_isects <= self.capacity and (self.classes[0] == 0 or self.isect_ids is not None)
        if ok:
            n = self.n_isects
            self.flatten_ids = self.flatten_ids[:n]
            self.isect_ids = self.isect_ids[:n] if self.isect_ids is not None else None
        return ok

    def abandon(self) -> None:
        """The caller failed before resolve(): the kernels that post into this thread's slot may still be in flight,
        and their values must not land after the NEXT call has armed the slot."""
        dev = self.offsets.device
        with B.device_guard(dev):
            B.lib().gs_stream_synchronize(B.stream_handle(dev))

    def redo(self):
        dev = self.flatten_ids.device
        if self.classes[0] > 0 and self.isect_ids is None:
            # a speculative sort launch may have met a list beyond its buffer and raised the contract flag: let it
            # finish, then clear the flag (this call's lists are rebuilt below; nobody else's launches are in flight
            # on this stream)
            with B.device_guard(dev):
                B.check(B.lib().gs_stream_synchronize(B.stream_handle(dev)), "gs_stream_synchronize")
                B.lib().gs_contract_status(1)
        return self._build(self.n_isects, self.classes)


def isect_tiles_binned(splats: Tensor, tiles_per_gauss: Tensor, tile_width: int, tile_height: int,
                       front: bool = False, cull_tile_size: int = 0, while_counting=None, want_ids: bool = True,
                       spec_key=None):
    """isect_tiles + sort + isect_offset_encode in one binned pipeline (include/gsplat_mi355x.h, "binned isect_tiles").
    Returns (isect_ids int64 [I] sorted, flatten_ids int32 [I], isect_offsets int32 [C,TH,TW], tile_order, front_sort),
    the first three bit-identical to `isect_tiles_from_splats(sort=True)` + `isect_offset_encode`; tile_order is the
    longest-list-first tile permutation (int32 [C*TH*TW], or None) that the per-tile kernels are launched in.
    front_sort is None unless front=True, when it is a `FrontSort`: only the leading entries of long tile lists are in
    order until its `finalize()` runs.
    cull_tile_size > 0 (the rasterizer's tile edge in pixels): the culled lists of include/gsplat_mi355x.h -- a subset
    of gsplat's, each tile's list still in gsplat's order; the offsets describe the culled lists.
    while_counting: a callable run after the count / offsets launches are queued and before the host waits for
    n_isects -- host work (allocations, independent launches) placed there hides behind the GPU's first stage.
    want_ids=False: a caller that only rasterizes; isect_ids is then None unless a list is longer than 2048 entries
    (the longest sort class borrows the array as scratch).
    spec_key (hashable, or None): the caller can take speculatively sized lists -- the return value is then a
    `SpecLists` instead of the 5-tuple whenever the last call with this key allows it (SPECULATE_SIZES above)."""
    dev = B.require_gpu(splats, tiles_per_gauss)
    assert not cull_tile_size or REGION_SCATTER, "culled lists need the region scatter"
    C, N = tiles_per_gauss.shape
    T = tile_width * tile_height
    lib, st = B.lib(), B.stream_handle(dev)
    # 1024-thread workgroups, two per CU over all cameras (fewer, longer slices keep the scatter's runs long)
    n_slices = max(1, min(max(SLICES_TOTAL // C, 1), -(-max(N, 1) // 4096)))
    table = torch.empty(C * n_slices * T, device=dev, dtype=torch.int32)
    counts = torch.empty(C * T, device=dev, dtype=torch.int32)
    offsets = torch.empty(C, tile_height, tile_width, device=dev, dtype=torch.int32)
    total = torch.empty(1, device=dev, dtype=torch.int64)
    # region scatter: bands of tile rows x groups of slices, about one 1024-thread workgroup per CU.  With direct 8-B
    # stores two bands measured best (of 1..32 bands x 1..32 slices per group at 5 x 1242x375 and 1 x 1920x1280:
    # DESIGN.md section 6);
    # staged through LDS (STAGED_SCATTER): bands of 2 000-5 000 tiles measured best -- one band per camera at 1242x375
    # (1 872 tiles: 42.8 us against 53.8 / 76.2 with two / four bands), two at 1920x1280 (9 600 tiles: 94 us against 122
    # with one band, 116 / 121 with three / four); profiles/r02_ab_runs.txt item 12
    if SCATTER_REGIONS:
        regions = min(SCATTER_REGIONS, tile_height)
    else:
        regions = min(max(1, T // 4000), tile_height) if STAGED_SCATTER else min(2, tile_height)
    # (a caller may ask for another workgroup count on its thread: the render-only harness, whose batches share the GPU)
    group = max(1, -(-(n_slices * C * regions) // (getattr(_tls, "scatter_workgroups", 0) or SCATTER_WORKGROUPS)))
    vis_list = vis_cnt = vis_masks = None
    if REGION_SCATTER:
        seg = int(lib.gs_isect_bin_slice_rows(N, n_slices))
        vis_list = torch.empty(C * n_slices * seg, 2, device=dev, dtype=torch.int64)
        vis_cnt = torch.empty(C * n_slices, device=dev, dtype=torch.int32)
        if cull_tile_size:
            vis_masks = torch.empty(C * n_slices * seg, device=dev, dtype=torch.int64)
    want_order = C * T <= 131072
    use_front = front and want_order
    fused = FUSED_LAUNCHES and REGION_SCATTER and not any(B.profiling_active(k) for k in _ISECT_LABELS)
    order = n_long = redo_flags = None
    if want_order:
        order = torch.empty(C * T, device=dev, dtype=torch.int32)
        n_long = torch.empty(3, device=dev, dtype=torch.int32)  # size classes of the order
    def _build_lists(n_isects, classes):
        """Everything behind the host's read of n_isects / the size classes: arrays of the exact size, scatter, sort."""
        uf, sc, rf = use_front, None, None
        if want_order:
            # host copy of the tile order's size classes: launches for classes that hold no tile are skipped, and
            # without a tile above 2048 entries there is no front to sort first
            sc = (ctypes.c_int32 * 3)(*classes)
            if sc[FRONT_FROM_CLASS] == 0:
                uf = False
        if uf and fused:  # (cleared here, once it is known that a front will be sorted, not by the prepare call)
            rf = torch.zeros(C * T + 1, device=dev, dtype=torch.int32)
        need_ids = want_ids or uf or sc is None or sc[0] > 0 or n_isects == 0
        isect_ids = torch.empty(n_isects, device=dev, dtype=torch.int64) if need_ids else None
        flatten_ids = torch.empty(n_isects, device=dev, dtype=torch.int32)
        if n_isects == 0:
            return isect_ids, flatten_ids, offsets, None, None
        pairs = torch.empty(n_isects, device=dev, dtype=torch.int64)
        if fused:
            with B.device_guard(dev):
                B.check(lib.gs_isect_bin_build(C, N, tile_width, tile_height, n_slices, group, regions, B.ptr(vis_list),
                                               B.ptr(vis_cnt), B.ptr(vis_masks), B.ptr(table), B.ptr(offsets), n_isects,
                                               B.ptr(order), B.ptr(n_long), FRONT_K_TARGET if uf else 0,
                                               B.ptr(pairs), B.ptr(isect_ids), B.ptr(flatten_ids),
                                               B.ptr(counts) if (STAGED_SCATTER or uf) else None,
                                               ctypes.addressof(sc) if sc is not None else None, st),
                        "gs_isect_bin_build")
            if uf:
                fs = FrontSort((C, tile_width, tile_height), offsets, order, n_long, pairs, isect_ids, flatten_ids, counts,
                               rf)
                fs.size_classes_host = sc
                return isect_ids, flatten_ids, offsets, order, fs
            return isect_ids, flatten_ids, offsets, order, None
        with B.device_guard(dev), B.timed("gs_isect_bin_scatter", dev):
            if vis_list is not None:
                B.check(lib.gs_isect_bin_scatter_regions(C, N, tile_width, tile_height, n_slices, group, regions,
                                                         B.ptr(vis_list), B.ptr(vis_cnt), B.ptr(vis_masks), B.ptr(table),
                                                         B.ptr(offsets), B.ptr(counts) if STAGED_SCATTER else None,
                                                         B.ptr(pairs), st), "gs_isect_bin_scatter_regions")
            else:
                B.check(lib.gs_isect_bin_scatter(C, N, tile_width, tile_height, B.ptr(splats), B.ptr(tiles_per_gauss),
                                                 n_slices, B.ptr(table), B.ptr(offsets), B.ptr(pairs), st),
                        "gs_isect_bin_scatter")
        sc_ptr = ctypes.addressof(sc) if sc is not None else None
        if uf:
            # `counts` (the list lengths) becomes sorted_counts: the front kernel overwrites the long tiles' entries
            with B.device_guard(dev), B.timed("gs_isect_tile_sort", dev):
                B.check(lib.gs_isect_tile_sort_front_ex(C, tile_width, tile_height, B.ptr(offsets), n_isects, B.ptr(order),
                                                        B.ptr(n_long), FRONT_K_TARGET, B.ptr(pairs), B.ptr(isect_ids),
                                                        B.ptr(flatten_ids), B.ptr(counts), sc_ptr, st),
                        "gs_isect_tile_sort_front")
            fs = FrontSort((C, tile_width, tile_height), offsets, order, n_long, pairs, isect_ids, flatten_ids, counts)
            fs.size_classes_host = sc
            return isect_ids, flatten_ids, offsets, order, fs
        with B.device_guard(dev), B.timed("gs_isect_tile_sort", dev):
            B.check(lib.gs_isect_tile_sort_ex(C, tile_width, tile_height, B.ptr(offsets), n_isects, B.ptr(order),
                                              B.ptr(n_long), B.ptr(pairs), B.ptr(isect_ids), B.ptr(flatten_ids), sc_ptr, st),
                    "gs_isect_tile_sort")
        return isect_ids, flatten_ids, offsets, order, None

    if fused:
        # one FFI call: count + offsets + tile order + the host read of n_isects (include/gsplat_mi355x.h, "composite
        # launches")
        if POSTED_RESULTS:
            slot_addr, slot_n, slot_sc = _post_slot()
        else:
            host_n = _host_scalar()
            slot_addr = host_n.data_ptr()
        with B.device_guard(dev):
            try:
                B.check(lib.gs_isect_bin_prepare(C, N, tile_width, tile_height, B.ptr(splats), B.ptr(tiles_per_gauss),
                                                 n_slices, B.ptr(table), B.ptr(vis_list), B.ptr(vis_cnt),
                                                 B.ptr(vis_masks), int(cull_tile_size), B.ptr(counts), B.ptr(offsets),
                                                 B.ptr(total), B.ptr(order), B.ptr(n_long), B.ptr(redo_flags), slot_addr,
                                                 slot_addr + 8 if want_order else None, 2 if POSTED_RESULTS else 0, st),
                        "gs_isect_bin_prepare")
                if while_counting is not None:
                    while_counting()
                    while_counting = None
            except BaseException:
                # kernels that post into this thread's slot may still be in flight: let them finish, or their values
                # would land in the slot after the NEXT call has armed it and be taken for that call's results
                lib.gs_stream_synchronize(st)
                raise

        def wait_posted():
            with B.device_guard(dev):
                try:
                    if POSTED_RESULTS:
                        # the kernels post n_isects / the size classes themselves: no copy kernels, no stream
                        # synchronisation, and the launches queued behind the producers (tile order) still run while
                        # the host goes on
                        B.check(lib.gs_host_post_wait(slot_addr, slot_addr + 8 if want_order else None, 10_000, st),
                                "gs_host_post_wait")
                        return int(slot_n[0]), (list(slot_sc) if want_order else None)
                    B.check(lib.gs_stream_synchronize(st), "gs_stream_synchronize")
                    return int(host_n[0]), (host_n.view(torch.int32)[2:5].tolist() if want_order else None)
                except BaseException:
                    lib.gs_stream_synchronize(st)  # (as above)
                    raise

        prev = _spec_table().get(spec_key) if (spec_key is not None and SPECULATE_SIZES and POSTED_RESULTS
                                               and want_order and 0 < C * N <= SPEC_MAX_ROWS) else None
        if prev is not None and prev[0] > 0 and prev[1][0] == 0 and prev[1][FRONT_FROM_CLASS] == 0:
            capacity = int(prev[0] * SPEC_MARGIN) + 1024
            try:
                s_ids = torch.empty(capacity, device=dev, dtype=torch.int64) if want_ids else None
                s_fids = torch.empty(capacity, device=dev, dtype=torch.int32)
                s_pairs = torch.empty(capacity, device=dev, dtype=torch.int64)
                with B.device_guard(dev):
                    B.check(lib.gs_isect_bin_build_dev(C, N, tile_width, tile_height, n_slices, group, regions,
                                                       B.ptr(vis_list), B.ptr(vis_cnt), B.ptr(vis_masks), B.ptr(table),
                                                       B.ptr(offsets), capacity, B.ptr(total), B.ptr(order),
                                                       B.ptr(n_long), B.ptr(s_pairs), B.ptr(s_ids), B.ptr(s_fids),
                                                       B.ptr(counts) if STAGED_SCATTER else None, st),
                            "gs_isect_bin_build_dev")
            except BaseException:
                # (as above: the slot is armed and nobody will wait on it -- let the posting kernels finish first)
                with B.device_guard(dev):
                    lib.gs_stream_synchronize(st)
                raise
            return SpecLists(capacity, total, s_ids, s_fids, offsets, order, wait_posted, _build_lists, spec_key)
        n_isects, classes = wait_posted()
        if spec_key is not None and classes is not None:
            _spec_table()[spec_key] = (n_isects, tuple(classes))
        return _build_lists(n_isects, classes)
    else:
        with B.device_guard(dev), B.timed("gs_isect_bin_count", dev):
            B.check(lib.gs_isect_bin_count(C, N, tile_width, tile_height, B.ptr(splats), B.ptr(tiles_per_gauss),
                                           n_slices, B.ptr(table), B.ptr(vis_list), B.ptr(vis_cnt), B.ptr(vis_masks),
                                           int(cull_tile_size), st), "gs_isect_bin_count")
        with B.device_guard(dev), B.timed("gs_isect_bin_offsets", dev):
            B.check(lib.gs_isect_bin_offsets(C, tile_width, tile_height, n_slices, B.ptr(table), B.ptr(counts),
                                             B.ptr(offsets), B.ptr(total), st), "gs_isect_bin_offsets")
        if want_order:
            with B.device_guard(dev), B.timed("gs_raster_tile_order", dev):
                B.check(lib.gs_raster_tile_order(C * T, None, 0, B.ptr(counts), B.ptr(order), B.ptr(n_long), st),
                        "gs_raster_tile_order")
        if while_counting is not None:
            while_counting()
        n_isects = int(total.item())  # the one host sync of the pipeline (gsplat has the same one)
        return _build_lists(n_isects, n_long.tolist() if want_order else None)


def isect_offset_encode(isect_ids: Tensor, n_cameras: int, tile_width: int, tile_height: int) -> Tensor:
    """Appendix A.4: offsets [C,tile_h,tile_w] int32 = lower bound of every (camera, tile) in the sorted keys."""
    dev = B.require_gpu(isect_ids)
    offsets = torch.empty(n_cameras, tile_height, tile_width, device=dev, dtype=torch.int32)
    with B.device_guard(dev), B.timed("gs_isect_offset_encode", dev):
        B.check(B.lib().gs_isect_offset_encode(isect_ids.numel(), B.ptr(isect_ids.contiguous()), n_cameras, tile_width,
                                               tile_height, B.ptr(offsets), B.stream_handle(dev)),
                "gs_isect_offset_encode")
    return offsets


# ----------------------------------------------------------------------------- rasterize
def launch_raster_fwd(splats, offsets, flatten_ids, order, fs, bg, geom, rc, ra, last_ids, work, opac_override,
                      final_T=None) -> None:
    """gs_rasterize_fwd (+ the redo pass of front-sorted lists) into preallocated outputs."""
    C, N, width, height, tile_size, channels, ed = geom
    dev = splats.device
    n_isects = flatten_ids.numel()
    if fs is not None and fs.complete:  # every list has been sorted whole since (meta["flatten_ids"] was read)
        fs = None

    def raster(redo_pass: int) -> None:
        with B.device_guard(dev), B.timed("gs_rasterize_fwd" if not redo_pass else "gs_rasterize_fwd_redo", dev):
            B.check(B.lib().gs_rasterize_fwd(C, width, height, tile_size, channels, B.ptr(splats), B.ptr(offsets),
                                             B.ptr(flatten_ids), n_isects, B.ptr(bg), int(ed), B.ptr(rc), B.ptr(ra),
                                             B.ptr(last_ids), B.ptr(order), B.ptr(work),
                                             B.ptr(fs.sorted_counts) if fs is not None else None,
                                             B.ptr(fs.redo_flags) if fs is not None else None, redo_pass,
                                             B.ptr(opac_override), N, B.ptr(final_T), B.stream_handle(dev)),
                    "gs_rasterize_fwd")

    if fs is not None and FUSED_LAUNCHES and not any(B.profiling_active(k) for k in _RASTER_FWD_LABELS):
        # pass 0 + redo sort + pass 1 in one FFI call
        with B.device_guard(dev):
            B.check(B.lib().gs_rasterize_fwd_front(C, width, height, tile_size, channels, B.ptr(splats), B.ptr(offsets),
                                                   B.ptr(flatten_ids), n_isects, B.ptr(bg), int(ed), B.ptr(rc),
                                                   B.ptr(ra), B.ptr(last_ids), B.ptr(order), B.ptr(work),
                                                   B.ptr(fs.sorted_counts), B.ptr(fs.redo_flags), B.ptr(opac_override),
                                                   N, B.ptr(final_T), B.ptr(fs.n_long), B.ptr(fs.pairs),
                                                   B.ptr(fs.isect_ids),
                                                   ctypes.addressof(fs.size_classes_host)
                                                   if fs.size_classes_host is not None else None,
                                                   B.stream_handle(dev)),
                    "gs_rasterize_fwd_front")
        return
    raster(0)
    if fs is not None:
        # tiles whose pixels were still live at the end of their sorted front: sort those lists whole and rasterize
        # them again (both launches visit flagged tiles only; usually there are none)
        fs.redo()
        raster(1)


def take_raster_state():
    """(splats, offsets, isect_ids, flatten_ids, tile_order, front_sort) of this thread's last _RasterizeSplats.forward:
    what a re-render with other opacities needs (gsplat.rendering's eval cache)."""
    st = getattr(_tls, "raster_state", None)
    _tls.raster_state = None
    return st


def rerender(state, opacities: Tensor, backgrounds, geom):
    """Rasterize cached records again with `opacities` [N] in place of the ones they were projected with (no grad)."""
    splats, offsets, isect_ids, flatten_ids, order, fs = state
    C, N, width, height, tile_size, channels, ed = geom
    dev = splats.device
    rc = torch.empty(C, height, width, channels, device=dev, dtype=torch.float32)
    ra = torch.empty(C, height, width, 1, device=dev, dtype=torch.float32)
    last_ids = torch.empty(C, height, width, device=dev, dtype=torch.int32)
    if fs is not None:
        fs.redo_flags.zero_()
    bg = B.f32c(backgrounds) if backgrounds is not None else None
    launch_raster_fwd(splats, offsets, flatten_ids, order if flatten_ids.numel() > 0 else None, fs, bg, geom, rc, ra,
                      last_ids, None, B.f32c(opacities))
    return rc, ra, last_ids


class _RasterizeSplats(torch.autograd.Function):
    """splat records -> (render_colors [C,H,W,CH], render_alphas [C,H,W,1]); tile intersection, sort and
    offsets (integer, non-differentiable) run inside forward."""

    @staticmethod
    def forward(ctx, splats, means2d, tiles_per_gauss, backgrounds, width, height, tile_size, channels, ed, absgrad,
                packed_index, cull=False):
        dev = B.require_gpu(splats)
        C, N = tiles_per_gauss.shape
        TW, TH = math.ceil(width / tile_size), math.ceil(height / tile_size)
        fs = None
        want_grad = ctx.needs_input_grad[0]
        out = {}

        def prepare_outputs():
            # host work that does not depend on n_isects: run while the GPU is still counting (isect_tiles_binned)
            out["rc"] = torch.empty(C, height, width, channels, device=dev, dtype=torch.float32)
            out["ra"] = torch.empty(C, height, width, 1, device=dev, dtype=torch.float32)
            out["last_ids"] = torch.empty(C, height, width, device=dev, dtype=torch.int32)
            # the backward starts from each pixel's final transmittance: kept exactly, not re-derived from 1 - alpha
            out["final_T"] = torch.empty(C, height, width, device=dev, dtype=torch.float32) if want_grad else None
            out["v_splats"] = None
            if want_grad:
                # the backward accumulates atomically into a zeroed gradient record
                out["v_splats"] = v = torch.empty_like(splats)
                if v.numel() * 4 <= SIDE_STREAM_MIN_BYTES:
                    # small record: the fill is a few microseconds of GPU time; the stream hand-over (two events, two
                    # waits) would cost the host more than that
                    with B.device_guard(dev):
                        B.check(B.lib().gs_fill_zero(B.ptr(v), v.numel() * 4, B.stream_handle(dev)), "gs_fill_zero")

        spec = None
        if TW * TH <= bin_max_tiles():
            culled = bool(cull and REGION_SCATTER)
            lists = isect_tiles_binned(
                splats, tiles_per_gauss, TW, TH, front=FRONT_SORT, cull_tile_size=tile_size if culled else 0,
                while_counting=prepare_outputs,
                want_ids=not culled,  # (culled lists are internal: meta builds gsplat's own on demand)
                # (N is not part of the key: densification changes it every refine interval, the count moves little)
                spec_key=(dev.index, C, width, height, channels, culled))
            if isinstance(lists, SpecLists):
                # speculatively sized lists: the forward rasterizer is queued on them before the host has looked at
                # this call's intersection count (include/gsplat_mi355x.h, "device-side n_isects")
                spec = lists
                isect_ids, flatten_ids, offsets, order = spec.isect_ids, spec.flatten_ids, spec.offsets, spec.order
            else:
                isect_ids, flatten_ids, offsets, order, fs = lists
        else:
            order = None
            isect_ids, flatten_ids = isect_tiles_from_splats(splats, tiles_per_gauss, TW, TH)
            offsets = isect_offset_encode(isect_ids, C, TW, TH)
            prepare_outputs()
        try:
            bg = B.f32c(backgrounds) if backgrounds is not None else None
            rc, ra, last_ids, final_T = out["rc"], out["ra"], out["last_ids"], out["final_T"]
            n_isects = flatten_ids.numel()  # (speculative lists: their capacity, > 0, until `spec.resolve()` below)
            ctx.v_splats, ctx.v_ready = out["v_splats"], None
            if ctx.v_splats is not None and ctx.v_splats.numel() * 4 > SIDE_STREAM_MIN_BYTES:
                # large record: zero it on a side stream, where the HBM-bound fill hides behind the issue-bound forward
                # rasterizer (issued here, behind the intersection stage: in front of it, it doubles the count pass)
                main, side = torch.cuda.current_stream(dev), B.side_stream(dev)
                side.wait_stream(main)
                with B.device_guard(dev):  # (raw stream handle: no torch stream-context switch on the host)
                    B.check(B.lib().gs_fill_zero(B.ptr(ctx.v_splats), ctx.v_splats.numel() * 4, side.cuda_stream),
                            "gs_fill_zero")
                ctx.v_ready = torch.cuda.Event()
                ctx.v_ready.record(side)
                # written on the side stream: if the graph is dropped without a backward, the allocator must not hand
                # the block to a main-stream tensor while the fill is still pending
                ctx.v_splats.record_stream(side)
            # longest-first tile schedule: by list length now, by the walked length (written by the forward) for backward
            n_tiles = C * TH * TW
            work = None
            if n_isects == 0:
                order = None
            if n_isects > 0 and n_tiles <= 131072:
                work = torch.empty(n_tiles, device=dev, dtype=torch.int32) if ctx.needs_input_grad[0] else None
            if n_isects > 0 and n_tiles <= 131072 and order is None:
                order = torch.empty(n_tiles, device=dev, dtype=torch.int32)
                with B.device_guard(dev), B.timed("gs_raster_tile_order", dev):
                    B.check(B.lib().gs_raster_tile_order(n_tiles, B.ptr(offsets), n_isects, None, B.ptr(order), None,
                                                         B.stream_handle(dev)), "gs_raster_tile_order")
            geom = (C, N, width, height, tile_size, channels, ed)
            if spec is not None:
                with B.device_guard(dev):
                    status = B.lib().gs_rasterize_fwd_dev(C, width, height, tile_size, channels, B.ptr(splats),
                                                          B.ptr(offsets), B.ptr(flatten_ids), spec.capacity,
                                                          B.ptr(spec.total), B.ptr(bg), int(ed), B.ptr(rc), B.ptr(ra),
                                                          B.ptr(last_ids), B.ptr(order), B.ptr(work), None, N,
                                                          B.ptr(final_T), B.stream_handle(dev))
                # GS_ERR_CONTRACT here: a speculative sort launch has already met a list beyond its buffer (a tile list
                # above 6144 entries without the key array as scratch) -- a failed speculation, which resolve() reports too
                # (size class 0 is not empty), not an error of this call
                if status != B.GS_ERR_CONTRACT:
                    B.check(status, "gs_rasterize_fwd_dev")
                if spec.resolve() and status == 0:
                    isect_ids, flatten_ids = spec.isect_ids, spec.flatten_ids
                else:
                    # the count did not fit the capacity (or a list was too long for the speculative sort): nothing was
                    # written; the same launches again on lists of the exact size
                    isect_ids, flatten_ids, offsets, order, fs = spec.redo()
                    if flatten_ids.numel() == 0:
                        order = work = None
                    elif order is None and n_tiles <= 131072:
                        order = torch.empty(n_tiles, device=dev, dtype=torch.int32)
                        with B.device_guard(dev):
                            B.check(B.lib().gs_raster_tile_order(n_tiles, B.ptr(offsets), flatten_ids.numel(), None,
                                                                 B.ptr(order), None, B.stream_handle(dev)),
                                    "gs_raster_tile_order")
                    launch_raster_fwd(splats, offsets, flatten_ids, order, fs, bg, geom, rc, ra, last_ids, work, None,
                                      final_T)
                n_isects = flatten_ids.numel()
            else:
                launch_raster_fwd(splats, offsets, flatten_ids, order, fs, bg, geom, rc, ra, last_ids, work, None, final_T)
        except BaseException:
            if spec is not None and spec.n_isects is None:  # (failed before the posted count was consumed)
                spec.abandon()
            raise
        _tls.front_sort = fs
        _tls.last_tile_work = work  # (diagnostics: bench.py sums it into the list entries the backward walks)
        _tls.raster_state = (splats, offsets, isect_ids, flatten_ids, order, fs)  # handed to rasterization() for the lazy completion of meta["isect_ids"/"flatten_ids"]
        order_bwd = None
        if work is not None:
            # (a tensor of its own: `order` and its size classes still describe the sort launches of `fs`)
            order_bwd = torch.empty_like(order)
            if ctx.v_ready is not None and not B.profiling_active("gs_raster_tile_order"):
                # only the backward reads it: on the side stream (raw handle), behind the fill and off the main stream,
                # where it overlaps whatever the caller queues next; one event covers the fill and the order
                side = B.side_stream(dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with B.device_guard(dev):
                    B.check(B.lib().gs_raster_tile_order(n_tiles, B.ptr(offsets), n_isects, B.ptr(work),
                                                         B.ptr(order_bwd), None, side.cuda_stream),
                            "gs_raster_tile_order")
                ctx.v_ready = torch.cuda.Event()
                ctx.v_ready.record(side)
                work.record_stream(side)
                order_bwd.record_stream(side)
            else:
                with B.device_guard(dev), B.timed("gs_raster_tile_order", dev):
                    B.check(B.lib().gs_raster_tile_order(n_tiles, B.ptr(offsets), n_isects, B.ptr(work),
                                                         B.ptr(order_bwd), None, B.stream_handle(dev)),
                            "gs_raster_tile_order")
        ctx.tile_order = order_bwd
        ctx.geom = (C, N, width, height, tile_size, channels, int(ed), bool(absgrad))
        ctx.absgrad_accumulate = getattr(_tls, "absgrad_accumulate", False)
        ctx.has_bg = bg is not None
        ctx.has_packed = packed_index is not None
        empty = torch.empty(0, device=dev)
        ctx.has_final_T = final_T is not None
        ctx.save_for_backward(splats, means2d, offsets, flatten_ids, bg if bg is not None else empty, rc, ra, last_ids,
                              packed_index if packed_index is not None else empty,
                              final_T if final_T is not None else empty)
        if isect_ids is None:  # (culled internal lists without keys: an empty placeholder for the output slot)
            isect_ids = torch.empty(0, device=dev, dtype=torch.int64)
        ctx.mark_non_differentiable(isect_ids, flatten_ids, offsets, last_ids)
        ctx.set_materialize_grads(False)  # no zero tensors for the integer outputs' (undefined) gradients
        return rc, ra, isect_ids, flatten_ids, offsets, last_ids

    @staticmethod
    def backward(ctx, v_rc, v_ra, *_):
        splats, means2d, offsets, flatten_ids, bg, rc, ra, last_ids, packed_index, final_T = ctx.saved_tensors
        final_T = final_T if ctx.has_final_T else None
        C, N, width, height, tile_size, channels, ed, absgrad = ctx.geom
        dev = splats.device
        bg = bg if ctx.has_bg else None
        v_rc = B.f32c(v_rc) if v_rc is not None else torch.zeros_like(rc)
        v_ra = B.f32c(v_ra) if v_ra is not None else torch.zeros_like(ra)
        if ctx.v_splats is not None:
            v_splats, ctx.v_splats = ctx.v_splats, None  # single use (a second backward allocates afresh)
            if ctx.v_ready is not None:
                torch.cuda.current_stream(dev).wait_event(ctx.v_ready)
        else:
            v_splats = torch.zeros_like(splats)
        with B.device_guard(dev), B.timed("gs_rasterize_bwd", dev):
            B.check(B.lib().gs_rasterize_bwd(C, width, height, tile_size, channels, B.ptr(splats), B.ptr(offsets),
                                             B.ptr(flatten_ids), flatten_ids.numel(), B.ptr(bg), ed, B.ptr(rc),
                                             B.ptr(ra), B.ptr(last_ids), B.ptr(v_rc), B.ptr(v_ra), int(absgrad),
                                             B.ptr(v_splats), B.ptr(ctx.tile_order), B.ptr(final_T),
                                             B.stream_handle(dev)), "gs_rasterize_bwd")
        if ctx.has_packed:
            flat = v_splats.reshape(-1, B.GS_REC)
            v_means2d = flat[packed_index, 0:2]
            if absgrad:
                means2d.absgrad = flat[packed_index, 10:12]
        else:
            v_means2d = v_splats[..., 0:2]
            if absgrad:
                # same attribute contract as gsplat: read by BasicTrainer.postprocess_per_train_step
                # (/root/reference/3D-Reconstruction-main/models/trainers/base.py:282)
                ag = v_splats[..., 10:12]
                if ctx.absgrad_accumulate and getattr(means2d, "absgrad", None) is not None:
                    ag = means2d.absgrad + ag
                means2d.absgrad = ag
        v_bg = None
        if ctx.has_bg and ctx.needs_input_grad[3]:
            vc = v_rc
            if ed:
                vc = vc.clone()
                vc[..., -1] = vc[..., -1] / ra[..., 0].clamp(min=1e-10)
            v_bg = (vc * (1.0 - ra)).sum(dim=(1, 2))
        return v_splats, v_means2d, None, v_bg, None, None, None, None, None, None, None, None


# ----------------------------------------------------------------------------- gsplat's stage-level functions
# `rasterization()` above runs the stages fused (one record per splat, lists built inside the rasterize op).  gsplat also
# exports the stages one by one (gsplat/cuda/_wrapper.py of gsplat 1.x: fully_fused_projection, isect_tiles,
# isect_offset_encode, rasterize_to_pixels); code that drives them itself -- gsplat's own tests do -- finds them here
# with the same signatures, on the same kernels.  They are the general route, not the fast one: each validates its
# integer inputs on the host before a kernel indexes with them.
def fully_fused_projection(means: Tensor, covars: Optional[Tensor], quats: Optional[Tensor], scales: Optional[Tensor],
                           viewmats: Tensor, Ks: Tensor, width: int, height: int, eps2d: float = 0.3,
                           near_plane: float = 0.01, far_plane: float = 1e10, radius_clip: float = 0.0,
                           packed: bool = False, sparse_grad: bool = False, calc_compensations: bool = False):
    """Appendix A.1: -> (radii [C,N] int32, means2d [C,N,2], depths [C,N], conics [C,N,3],
    compensations [C,N] or None).  Rows with radii == 0 are zero in every output.  Differentiable in means, covars or
    quats/scales, and viewmats.
    packed=True (A.7): -> (camera_ids [nnz] int64, gaussian_ids [nnz] int64, radii [nnz], means2d [nnz,2], depths [nnz],
    conics [nnz,3], compensations [nnz] or None), the visible rows in ascending (camera, Gaussian) order; a gather of the
    dense outputs (one host synchronisation for nnz), so gradients flow as in the dense form."""
    if sparse_grad:
        raise NotImplementedError("gsplat (MI355X build): sparse_grad is not supported (the reference never sets it)")
    if packed:
        radii, means2d, depths, conics, comps = fully_fused_projection(
            means, covars, quats, scales, viewmats, Ks, width, height, eps2d, near_plane, far_plane, radius_clip, False,
            False, calc_compensations)
        camera_ids, gaussian_ids = torch.nonzero(radii > 0, as_tuple=True)
        return (camera_ids, gaussian_ids, radii[camera_ids, gaussian_ids], means2d[camera_ids, gaussian_ids],
                depths[camera_ids, gaussian_ids], conics[camera_ids, gaussian_ids],
                comps[camera_ids, gaussian_ids] if comps is not None else None)
    assert (covars is not None) or (quats is not None and scales is not None), "covars or quats+scales"
    N, C = means.shape[0], viewmats.shape[0]
    if covars is not None and covars.dim() == 3:  # gsplat takes [N,3,3] too
        covars = torch.stack([covars[:, 0, 0], covars[:, 0, 1], covars[:, 0, 2], covars[:, 1, 1], covars[:, 1, 2],
                              covars[:, 2, 2]], dim=-1)
    # no colours, the depth in the first colour slot, opacity 1: record[5] is then the compensation itself
    params = _make_params(N, C, int(width), int(height), 16, eps2d, near_plane, far_plane, radius_clip,
                          calc_compensations, B.GS_COLOR_NONE, 0, None, 0, True)
    ones = torch.ones(N, device=means.device, dtype=torch.float32)
    splats, means2d, radii, _tiles = _ProjectSplats.apply(means, quats, scales, ones, None, viewmats, Ks, None, params,
                                                          covars)
    vis = radii > 0
    zero = torch.zeros((), device=means.device, dtype=torch.float32)
    depths = torch.where(vis, splats[..., 6], zero)
    conics = torch.where(vis[..., None], splats[..., 2:5], zero)
    comps = torch.where(vis, splats[..., 5], zero) if calc_compensations else None
    return radii, means2d, depths, conics, comps


def isect_tiles(means2d: Tensor, radii: Tensor, depths: Tensor, tile_size: int, tile_width: int, tile_height: int,
                sort: bool = True, packed: bool = False, n_cameras: Optional[int] = None,
                camera_ids: Optional[Tensor] = None, gaussian_ids: Optional[Tensor] = None):
    """Appendix A.3: -> (tiles_per_gauss [C,N] int32, isect_ids [I] int64, flatten_ids [I] int32).  Any tile_size.
    The tile boxes are computed here in fp32 exactly as the projection kernel does it, then emitted and sorted by the
    kernels of `isect_tiles_from_splats`.
    packed=True (A.7): means2d [nnz,2], radii [nnz], depths [nnz] with `n_cameras` and `camera_ids` [nnz] (gaussian_ids
    is not needed); tiles_per_gauss is [nnz] and flatten_ids index the packed rows.  Ties in (camera, tile, depth) come
    out in ascending packed row, as gsplat's stable sort of its emission order leaves them."""
    if packed:
        assert n_cameras is not None and camera_ids is not None, "packed=True needs n_cameras and camera_ids"
        dev = B.require_gpu(means2d, radii, depths, camera_ids)
        nnz, C = radii.shape[0], int(n_cameras)
        assert means2d.shape == (nnz, 2) and depths.shape == (nnz,) and camera_ids.shape == (nnz,)
        empty = lambda dt: torch.empty(0, device=dev, dtype=dt)
        if nnz == 0:
            return empty(torch.int32), empty(torch.int64), empty(torch.int32)
        cam = camera_ids.long()
        if bool(((cam < 0) | (cam >= C)).any()):
            raise ValueError("isect_tiles: camera_ids must lie in [0, n_cameras)")
        # packed row r of camera c sits at column r of camera c's row of a [C, nnz] layout (every other cell holds no
        # tiles): the kernels' flat id c * nnz + r gives the camera by division and the packed row by remainder, and
        # ascending flat id within a camera is ascending packed row
        rows = torch.arange(nnz, device=dev)
        m_d = torch.zeros(C, nnz, 2, device=dev, dtype=torch.float32)
        r_d = torch.zeros(C, nnz, device=dev, dtype=torch.int32)
        d_d = torch.zeros(C, nnz, device=dev, dtype=torch.float32)
        m_d[cam, rows] = means2d.detach().to(torch.float32)
        r_d[cam, rows] = radii.to(torch.int32)
        d_d[cam, rows] = depths.detach().to(torch.float32)
        tiles, isect_ids, fids = isect_tiles(m_d, r_d, d_d, tile_size, tile_width, tile_height, sort=sort)
        flatten_ids = (fids % nnz).to(torch.int32)
        if not sort:  # gsplat emits in packed-row order whatever the camera
            order = torch.sort(flatten_ids.long(), stable=True).indices
            isect_ids, flatten_ids = isect_ids[order], flatten_ids[order]
        return tiles[cam, rows].contiguous(), isect_ids, flatten_ids
    dev = B.require_gpu(means2d, radii, depths)
    C, N = radii.shape
    assert means2d.shape == (C, N, 2) and depths.shape == (C, N), (means2d.shape, depths.shape)
    assert 0 < tile_width <= 0xFFFF and 0 < tile_height <= 0xFFFF
    if C * N == 0:
        return (torch.zeros(C, N, device=dev, dtype=torch.int32), torch.empty(0, device=dev, dtype=torch.int64),
                torch.empty(0, device=dev, dtype=torch.int32))
    assert C * N < 2 ** 31, "flat row ids are int32"
    with torch.no_grad():
        ts = float(tile_size)
        m = means2d.detach().to(torch.float32) / ts
        tr = radii.to(torch.float32) / ts
        x0 = torch.floor(m[..., 0] - tr).clamp(0, tile_width)
        x1 = torch.ceil(m[..., 0] + tr).clamp(0, tile_width)
        y0 = torch.floor(m[..., 1] - tr).clamp(0, tile_height)
        y1 = torch.ceil(m[..., 1] + tr).clamp(0, tile_height)
        box = torch.stack([x0, x1, y0, y1], dim=-1)
        ok = (radii > 0) & torch.isfinite(box).all(dim=-1)  # (a NaN mean gives no tiles: the kernel's float->int is 0 too)
        x0, x1, y0, y1 = (torch.where(ok, v, torch.zeros_like(v)).to(torch.int32) for v in (x0, x1, y0, y1))
        tiles = ((y1 - y0) * (x1 - x0)).contiguous()
        rec = torch.zeros(C, N, B.GS_REC, device=dev, dtype=torch.float32)
        rec[..., 10] = depths.detach().to(torch.float32)
        bits = rec.view(torch.int32)
        bits[..., 13] = x0 | (y0 << 16)
        bits[..., 14] = x1 | (y1 << 16)
        isect_ids, flatten_ids = isect_tiles_from_splats(rec, tiles, tile_width, tile_height, sort=sort)
    return tiles, isect_ids, flatten_ids


class _RasterizeLists(torch.autograd.Function):
    """records [R,16] + caller-supplied tile lists -> (render_colors, render_alphas); `channels` <= 4."""

    @staticmethod
    def forward(ctx, records, means2d, offsets, flatten_ids, backgrounds, width, height, channels, absgrad):
        dev = B.require_gpu(records, offsets, flatten_ids)
        C, TH, TW = offsets.shape
        n_isects, n_tiles = flatten_ids.numel(), C * TH * TW
        records = B.f32c(records)
        bg = B.f32c(backgrounds) if backgrounds is not None else None
        want_grad = ctx.needs_input_grad[0]
        rc = torch.empty(C, height, width, channels, device=dev, dtype=torch.float32)
        ra = torch.empty(C, height, width, 1, device=dev, dtype=torch.float32)
        last_ids = torch.empty(C, height, width, device=dev, dtype=torch.int32)
        final_T = torch.empty(C, height, width, device=dev, dtype=torch.float32) if want_grad else None
        order = None
        if n_isects > 0 and n_tiles <= 131072:
            order = torch.empty(n_tiles, device=dev, dtype=torch.int32)
            with B.device_guard(dev):
                B.check(B.lib().gs_raster_tile_order(n_tiles, B.ptr(offsets), n_isects, None, B.ptr(order), None,
                                                     B.stream_handle(dev)), "gs_raster_tile_order")
        geom = (C, records.shape[0], width, height, 16, channels, False)
        launch_raster_fwd(records, offsets, flatten_ids, order, None, bg, geom, rc, ra, last_ids, None, None, final_T)
        ctx.geom = (C, width, height, channels, absgrad)
        ctx.has_bg, ctx.has_final_T = bg is not None, final_T is not None
        empty = torch.empty(0, device=dev)
        ctx.save_for_backward(records, means2d, offsets, flatten_ids, bg if bg is not None else empty, rc, ra, last_ids,
                              final_T if final_T is not None else empty, order if order is not None else empty)
        ctx.has_order = order is not None
        return rc, ra

    @staticmethod
    def backward(ctx, v_rc, v_ra):
        records, means2d, offsets, flatten_ids, bg, rc, ra, last_ids, final_T, order = ctx.saved_tensors
        C, width, height, channels, absgrad = ctx.geom
        dev = records.device
        v_rc = B.f32c(v_rc) if v_rc is not None else torch.zeros_like(rc)
        v_ra = B.f32c(v_ra) if v_ra is not None else torch.zeros_like(ra)
        v_records = torch.zeros_like(records)
        if flatten_ids.numel() > 0:
            with B.device_guard(dev), B.timed("gs_rasterize_bwd", dev):
                B.check(B.lib().gs_rasterize_bwd(C, width, height, 16, channels, B.ptr(records), B.ptr(offsets),
                                                 B.ptr(flatten_ids), flatten_ids.numel(),
                                                 B.ptr(bg) if ctx.has_bg else None, 0, B.ptr(rc), B.ptr(ra),
                                                 B.ptr(last_ids), B.ptr(v_rc), B.ptr(v_ra), int(absgrad is not None),
                                                 B.ptr(v_records), B.ptr(order) if ctx.has_order else None,
                                                 B.ptr(final_T) if ctx.has_final_T else None, B.stream_handle(dev)),
                        "gs_rasterize_bwd")
        if absgrad is not None:
            # (pass id, state shared by the passes of one rasterize_to_pixels call): above four channels the attribute
            # is the SUM of the passes' abs-gradients, as in rasterization()'s >4-channel path (set_absgrad_accumulate)
            pass_id, acc = absgrad
            if pass_id in acc["seen"]:  # a second backward through the same graph starts over
                acc["seen"], acc["sum"] = set(), None
            ag = v_records[:, 10:12].reshape(means2d.shape)
            acc["sum"] = ag if acc["sum"] is None else acc["sum"] + ag
            acc["seen"].add(pass_id)
            means2d.absgrad = acc["sum"]
        v_bg = (v_rc * (1.0 - ra)).sum(dim=(1, 2)) if (ctx.has_bg and ctx.needs_input_grad[4]) else None
        return v_records, None, None, None, v_bg, None, None, None, None


def rasterize_to_pixels(means2d: Tensor, conics: Tensor, colors: Tensor, opacities: Tensor, image_width: int,
                        image_height: int, tile_size: int, isect_offsets: Tensor, flatten_ids: Tensor,
                        backgrounds: Optional[Tensor] = None, masks: Optional[Tensor] = None, packed: bool = False,
                        absgrad: bool = False):
    """Appendix A.5 / A.6: -> (render_colors [C,H,W,D], render_alphas [C,H,W,1]).  Dense ([C,N,*]) or packed ([nnz,*])
    operands; any channel count (passes of four; `means2d.absgrad` is then the sum of the passes' abs-gradients, an upper
    bound of the one-kernel value); `masks` [C,tile_h,tile_w] (False: the tile shows the background, alpha 0).
    `flatten_ids` and `isect_offsets` are checked on the host (one synchronisation) before the kernels index with them."""
    if int(tile_size) != 16:
        raise NotImplementedError("gsplat (MI355X build): tile_size must be 16")
    dev = B.require_gpu(means2d, conics, colors, opacities, isect_offsets, flatten_ids)
    C, TH, TW = isect_offsets.shape
    width, height = int(image_width), int(image_height)
    assert TH == math.ceil(height / 16) and TW == math.ceil(width / 16), (isect_offsets.shape, width, height)
    if packed:
        R = means2d.shape[0]
        assert means2d.shape == (R, 2) and conics.shape == (R, 3) and opacities.shape == (R,) and colors.shape[0] == R
    else:
        assert means2d.dim() == 3 and means2d.shape[0] == C, means2d.shape
        R = C * means2d.shape[1]
        assert conics.shape == means2d.shape[:2] + (3,) and opacities.shape == means2d.shape[:2]
        assert colors.shape[:2] == means2d.shape[:2]
    D = colors.shape[-1]
    assert D >= 1 and (backgrounds is None or backgrounds.shape == (C, D))
    assert isect_offsets.dtype == torch.int32 and flatten_ids.dtype == torch.int32 and flatten_ids.dim() == 1
    isect_offsets, flatten_ids = isect_offsets.contiguous(), flatten_ids.contiguous()
    n_isects = flatten_ids.numel()
    # the kernels gather records through these: out-of-range values are a memory fault, not an exception
    off = isect_offsets.reshape(-1)
    bad = (off[0] != 0) | (off[1:] < off[:-1]).any() | (off[-1] > n_isects)
    if n_isects > 0:
        bad = bad | (flatten_ids.min() < 0) | (flatten_ids.max() >= R)
    if bool(bad):
        raise ValueError("rasterize_to_pixels: isect_offsets must start at 0, be non-decreasing and stay within "
                         "len(flatten_ids); flatten_ids must lie in [0, rows)")
    m2 = means2d.reshape(R, 2).to(torch.float32)
    head = torch.cat([m2, conics.reshape(R, 3).to(torch.float32), opacities.reshape(R, 1).to(torch.float32)], dim=-1)
    cols = colors.reshape(R, D).to(torch.float32)
    outs, alphas = [], None
    acc = {"seen": set(), "sum": None} if absgrad else None
    for c0 in range(0, D, 4):
        ch = min(4, D - c0)
        rec = torch.cat([head, cols[:, c0:c0 + ch], head.new_zeros(R, B.GS_REC - 6 - ch)], dim=-1)
        bg = backgrounds[:, c0:c0 + ch] if backgrounds is not None else None
        rc, ra = _RasterizeLists.apply(rec, means2d, isect_offsets, flatten_ids, bg, width, height, ch,
                                       (c0, acc) if absgrad else None)
        outs.append(rc)
        alphas = ra if alphas is None else alphas
    render_colors = outs[0] if len(outs) == 1 else torch.cat(outs, dim=-1)
    if masks is not None:
        assert masks.shape == (C, TH, TW)
        pix = masks.to(torch.bool).repeat_interleave(16, dim=1).repeat_interleave(16, dim=2)[:, :height, :width, None]
        fill = (backgrounds[:, None, None, :].expand_as(render_colors) if backgrounds is not None
                else torch.zeros_like(render_colors))
        render_colors = torch.where(pix, render_colors, fill)
        alphas = torch.where(pix, alphas, torch.zeros_like(alphas))
    return render_colors, alphas
