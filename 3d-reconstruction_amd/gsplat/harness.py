"""Render-only trajectory harness (BASELINE configs[4]).

The reference renders a novel-view trajectory one frame at a time and blocks on `outputs["rgb"].cpu().numpy()` twice
per frame before the next frame is even projected
(/root/reference/3D-Reconstruction-main/models/video_utils.py:647-706).  Here the views of a trajectory are sharded
contiguously over the ranks (no collective: every rank holds all Gaussians and writes its own frames), rendered
`views_per_call` at a time under no_grad, converted on the device exactly as the reference converts on the host
(`clip(1e-6, 1 - 1e-6) * 255 -> uint8`, or left as float32), and copied into one pinned host buffer on a side stream
while the next batch renders.  One host synchronisation at the end.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _backend as B
from .distributed import shard_views
from .rendering import no_eval_cache, rasterization


SH_ROWS = 2  # neighbouring views: coefficient rows on demand, whatever views_per_call is (gs_proj_params.sh_rows)
SCATTER_WORKGROUPS_SHARED = 128  # region-scatter workgroups when batches alternate over streams (render_trajectory)


def frames_to_uint8(rgb: Tensor) -> Tensor:
    """video_utils.py:694-700: `rgb.clip(min=1e-6, max=1 - 1e-6)`, `(rgb * 255).astype(np.uint8)` (truncation), in one
    pass on the device (gs_frames_to_uint8)."""
    dev = B.require_gpu(rgb)
    src = B.f32c(rgb)
    out = torch.empty(src.shape, device=dev, dtype=torch.uint8)
    with B.device_guard(dev):
        B.check(B.lib().gs_frames_to_uint8(src.numel(), B.ptr(src), B.ptr(out), B.stream_handle(dev)),
                "gs_frames_to_uint8")
    return out


def render_trajectory(means: Tensor, quats: Tensor, scales: Tensor, opacities: Tensor, colors: Tensor,
                      viewmats: Tensor, Ks: Tensor, width: int, height: int, *, sh_degree: Optional[int] = None,
                      views_per_call: int = 8, rank: int = 0, world_size: int = 1, as_uint8: bool = True,
                      out: Optional[Tensor] = None, streams: int = 2, **raster_kwargs) -> Tuple[Tensor, range]:
    """Render views `shard_views(V, rank, world_size)` of a [V,4,4] / [V,3,3] trajectory.

    Returns (frames, view_range): frames is a pinned host tensor [len(view_range), H, W, 3] (uint8, or float32 with
    as_uint8=False) holding the RGB channels; `out` may supply it.  Extra keyword arguments go to `rasterization`
    (near_plane, far_plane, packed, rasterize_mode, backgrounds per view [V,3], ...).  Device tensors only.
    streams > 1: successive batches go to that many HIP streams in turn (batches are independent: one's projection and
    count pass, which wait on memory, share the GPU with the previous one's sort and compositing, which do not); the
    frames are the same bits either way.  Defaults from the 6 M-Gaussian 1920x1080 trajectory (views/s, same box):
    4 views x 1 stream 1465, 8 x 1 1484, 4 x 2 1554, 8 x 2 1585, 16 x 2 1596, 8 x 3 1553, 2 x 4 1393."""
    dev = B.require_gpu(means, quats, scales, opacities, colors, viewmats, Ks)
    V = viewmats.shape[0]
    assert viewmats.shape == (V, 4, 4) and Ks.shape == (V, 3, 3), (viewmats.shape, Ks.shape)
    assert views_per_call >= 1
    mine = shard_views(V, rank, world_size)
    n_mine = len(mine)
    dtype = torch.uint8 if as_uint8 else torch.float32
    if out is None:
        out = torch.empty(n_mine, height, width, 3, dtype=dtype, pin_memory=True)
    assert out.shape == (n_mine, height, width, 3) and out.dtype == dtype and out.device.type == "cpu"
    backgrounds = raster_kwargs.pop("backgrounds", None)
    raster_kwargs.setdefault("packed", True)
    raster_kwargs.setdefault("render_mode", "RGB")
    assert streams >= 1
    main = torch.cuda.current_stream(dev)
    copy = B.side_stream(dev)
    lanes = [main] if streams == 1 else _lane_streams(dev, streams)
    from .cuda import _wrapper as W

    prev_rows, prev_wg = getattr(W._tls, "sh_rows", 0), getattr(W._tls, "scatter_workgroups", 0)
    W._tls.sh_rows = SH_ROWS
    if len(lanes) > 1 and not prev_wg:
        # with another batch running beside it the region scatter does best on half the CUs (slice groups twice as
        # long: longer runs per tile, and room for the other stream's kernels); 6 M Gaussians, 1920x1080, 8 views per
        # call x 2 streams, views/s by workgroups: 64 1447, 96 1554, 128 1679, 192 1643, 256 1587, 512 1574
        W._tls.scatter_workgroups = SCATTER_WORKGROUPS_SHARED
    try:
        for lane in lanes:
            if lane is not main:
                lane.wait_stream(main)  # (the Gaussians and cameras were produced on the caller's stream)
        _render_batches(means, quats, scales, opacities, colors, viewmats, Ks, width, height, sh_degree, views_per_call,
                        mine, n_mine, as_uint8, out, backgrounds, raster_kwargs, lanes, copy)
    finally:
        W._tls.sh_rows, W._tls.scatter_workgroups = prev_rows, prev_wg
        for lane in lanes:
            if lane is not main:
                main.wait_stream(lane)  # (the caller may free or overwrite the inputs behind this call)
    copy.synchronize()
    return out, mine


_lanes = {}


def _lane_streams(dev, n: int):
    """n compute streams per device for independent batches (made once: creating a stream is not free)."""
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    have = _lanes.setdefault(key, [])
    while len(have) < n:
        have.append(torch.cuda.Stream(device=dev))
    return have[:n]


def _render_batches(means, quats, scales, opacities, colors, viewmats, Ks, width, height, sh_degree, views_per_call,
                    mine, n_mine, as_uint8, out, backgrounds, raster_kwargs, lanes, copy) -> None:
    with torch.no_grad(), no_eval_cache():  # every batch is a new camera set: nothing to reuse
        for i, b0 in enumerate(range(0, n_mine, views_per_call)):
            lane = lanes[i % len(lanes)]
            idx = slice(mine.start + b0, mine.start + min(b0 + views_per_call, n_mine))
            # (everything a batch allocates comes from its lane's pool of the caching allocator and is released to it)
            with torch.cuda.stream(lane):
                bg = backgrounds[idx] if backgrounds is not None else None
                rc, _, _ = rasterization(means, quats, scales, opacities, colors, viewmats[idx], Ks[idx], width, height,
                                         sh_degree=sh_degree, backgrounds=bg, **raster_kwargs)
                frames = rc[..., :3]
                frames = frames_to_uint8(frames) if as_uint8 else frames.contiguous()
                del rc
            copy.wait_stream(lane)
            with torch.cuda.stream(copy):
                out[b0:b0 + frames.shape[0]].copy_(frames, non_blocking=True)
            frames.record_stream(copy)
