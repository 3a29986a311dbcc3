"""ctypes binding of libgsplat_mi355x.so (C ABI declared in include/gsplat_mi355x.h).

There is NO fallback: if the shared library is missing or a call fails this module raises.  PyTorch is
used only for device memory, streams and autograd plumbing; every arithmetic step of the hot path runs
in the hand-written gfx950 kernels behind these entry points.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int32, c_int64, c_void_p
from typing import Optional

import torch

GS_REC = 16
GS_COLOR_NONE, GS_COLOR_N, GS_COLOR_CN, GS_COLOR_SH = 0, 1, 2, 3
GS_ERR_CONTRACT = -5  # include/gsplat_mi355x.h

_LIB_NAME = "libgsplat_mi355x.so"
_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "lib", _LIB_NAME)


class PixelLossOpts(Structure):
    """Mirror of `gs_pixel_loss_opts` (include/gsplat_mi355x.h)."""

    _fields_ = [
        ("bce_mode", c_int32), ("bce_limit", c_float),
        ("depth_type", c_int32), ("depth_normalize", c_int32), ("depth_inverse", c_int32),
        ("depth_reduction", c_int32), ("max_depth", c_float),
    ]


class ProjParams(Structure):
    """Mirror of `gs_proj_params` (include/gsplat_mi355x.h)."""

    _fields_ = [
        ("N", c_int32), ("C", c_int32),
        ("width", c_int32), ("height", c_int32),
        ("tile_size", c_int32), ("tile_w", c_int32), ("tile_h", c_int32),
        ("eps2d", c_float), ("near_plane", c_float), ("far_plane", c_float), ("radius_clip", c_float),
        ("calc_compensations", c_int32),
        ("color_mode", c_int32), ("color_dim", c_int32),
        ("sh_degree", c_int32), ("sh_K", c_int32),
        ("depth_channel", c_int32),
        ("sh_rows", c_int32),
    ]


_P = c_void_p
_SIGNATURES = {
    "gs_version": (ctypes.c_int, []),
    "gs_error_string": (c_char_p, [ctypes.c_int]),
    "gs_fill_zero": (ctypes.c_int, [_P, c_int64, _P]),
    "gs_frames_to_uint8": (ctypes.c_int, [c_int64, _P, _P, _P]),
    "gs_projection_fwd": (ctypes.c_int, [POINTER(ProjParams)] + [_P] * 15 + [_P]),
    "gs_colour_message_layout": (ctypes.c_int, [c_int32, c_int32, c_int64, POINTER(c_int64)]),
    "gs_projection_bwd_split": (ctypes.c_int, [POINTER(ProjParams)] + [_P] * 10 + [c_int64] + [_P] * 8 + [c_int64, c_int32, _P]),
    "gs_sh_bwd_packed": (ctypes.c_int, [c_int32] * 5 + [c_int64, c_int64] + [_P] * 5 + [c_int32, c_int32, _P, _P]),
    "gs_projection_bwd": (ctypes.c_int, [POINTER(ProjParams)] + [_P] * 11 + [c_int64] + [_P] * 10 + [_P]),
    "gs_sh_fwd": (ctypes.c_int, [c_int64, c_int32, c_int32, _P, _P, _P, _P, _P]),
    "gs_sh_bwd": (ctypes.c_int, [c_int64, c_int32, c_int32, _P, _P, _P, _P, _P, _P, _P]),
    "gs_exclusive_scan_workspace_bytes": (c_int64, [c_int64]),
    "gs_exclusive_scan_i32": (ctypes.c_int, [_P, _P, c_int64, _P, _P, c_int64, _P]),
    "gs_isect_emit": (ctypes.c_int, [c_int32, c_int32, c_int32, c_int32, _P, _P, _P, _P, _P, _P]),
    "gs_isect_bin_max_tiles": (c_int32, []),
    "gs_isect_bin_slice_rows": (c_int32, [c_int32, c_int32]),
    "gs_isect_bin_count": (ctypes.c_int, [c_int32, c_int32, c_int32, c_int32, _P, _P, c_int32, _P, _P, _P, _P, c_int32,
                                          _P]),
    "gs_isect_bin_prepare": (ctypes.c_int, [c_int32, c_int32, c_int32, c_int32, _P, _P, c_int32, _P, _P, _P, _P, c_int32,
                                            _P, _P, _P, _P, _P, _P, _P, _P, c_int32, _P]),
    "gs_stream_synchronize": (ctypes.c_int, [_P]),
    "gs_host_post_alloc": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), c_int64]),
    "gs_host_post_free": (ctypes.c_int, [_P]),
    "gs_contract_status": (ctypes.c_int, [c_int32]),
    "gs_isect_bin_build_dev": (ctypes.c_int, [c_int32] * 7 + [_P] * 5 + [c_int64] + [_P] * 8),
    "gs_rasterize_fwd_dev": (ctypes.c_int, [c_int32] * 5 + [_P, _P, _P, c_int64, _P, _P, c_int32] + [_P] * 6 +
                             [c_int32, _P, _P]),
    "gs_host_post_wait": (ctypes.c_int, [_P, _P, c_int32, _P]),
    "gs_visibility_prefix": (ctypes.c_int, [c_int32, c_int32, _P, _P, _P, _P, c_int64, _P]),
    "gs_isect_bin_build": (ctypes.c_int, [c_int32] * 7 + [_P] * 5 + [c_int64, _P, _P, c_int32, _P, _P, _P, _P, _P, _P]),
    "gs_isect_tile_sort_ex": (ctypes.c_int, [c_int32, c_int32, c_int32, _P, c_int64, _P, _P, _P, _P, _P, _P, _P]),
    "gs_isect_tile_sort_front_ex": (ctypes.c_int, [c_int32, c_int32, c_int32, _P, c_int64, _P, _P, c_int32, _P, _P, _P,
                                                   _P, _P, _P]),
    "gs_isect_tile_sort_redo_ex": (ctypes.c_int, [c_int32, c_int32, c_int32, _P, c_int64, _P, _P, _P, _P, _P, _P, _P,
                                                  _P, _P]),
    "gs_rasterize_fwd_front": (ctypes.c_int, [c_int32] * 5 + [_P, _P, _P, c_int64, _P, c_int32] + [_P] * 8 +
                               [c_int32, _P, _P, _P, _P, _P, _P]),
    "gs_isect_bin_offsets": (ctypes.c_int, [c_int32, c_int32, c_int32, c_int32, _P, _P, _P, _P, _P]),
    "gs_isect_bin_offsets_post": (ctypes.c_int, [c_int32, c_int32, c_int32, c_int32, _P, _P, _P, _P, _P, _P, _P]),
    "gs_isect_bin_scatter": (ctypes.c_int, [c_int32, c_int32, c_int32, c_int32, _P, _P, c_int32, _P, _P, _P, _P]),
    "gs_isect_bin_scatter_regions": (ctypes.c_int, [c_int32] * 7 + [_P] * 8),
    "gs_isect_tile_sort": (ctypes.c_int, [c_int32, c_int32, c_int32, _P, c_int64, _P, _P, _P, _P, _P, _P]),
    "gs_radix_sort_workspace_bytes": (c_int64, [c_int64]),
    "gs_radix_sort_pairs_u64_u32_ex": (ctypes.c_int, [c_int64, _P, c_int32, c_int32, _P, _P, _P, _P, _P, c_int64, POINTER(c_int32), _P]),
    "gs_radix_sort_pairs_u64_u32_bits": (ctypes.c_int, [c_int64, c_int32, c_int32, _P, _P, _P, _P, _P, c_int64, POINTER(c_int32), _P]),
    "gs_radix_sort_pairs_u64_u32": (ctypes.c_int, [c_int64, c_int32, _P, _P, _P, _P, _P, c_int64, POINTER(c_int32), _P]),
    "gs_radix_sort_pairs_u64_u32_static": (ctypes.c_int, [c_int64, c_int32, c_int32, _P, _P, _P, _P, POINTER(c_int32), _P]),
    "gs_isect_offset_encode": (ctypes.c_int, [c_int64, _P, c_int32, c_int32, c_int32, _P, _P]),
    "gs_isect_tile_sort_front": (ctypes.c_int, [c_int32, c_int32, c_int32, _P, c_int64, _P, _P, c_int32, _P, _P, _P, _P, _P]),
    "gs_isect_tile_sort_redo": (ctypes.c_int, [c_int32, c_int32, c_int32, _P, c_int64, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gs_rasterize_fwd": (ctypes.c_int, [c_int32] * 5 + [_P, _P, _P, c_int64, _P, c_int32, _P, _P, _P, _P, _P, _P, _P, c_int32, _P, c_int32, _P, _P]),
    "gs_rasterize_bwd": (ctypes.c_int, [c_int32] * 5 + [_P, _P, _P, c_int64, _P, c_int32, _P, _P, _P, _P, _P, c_int32, _P, _P, _P, _P]),
    "gs_raster_tile_order": (ctypes.c_int, [c_int32, _P, c_int64, _P, _P, _P, _P]),
    "gs_raster_tile_order_post": (ctypes.c_int, [c_int32, _P, c_int64, _P, _P, _P, _P, _P]),
    "gs_densify_accumulate": (ctypes.c_int, [c_int32, c_int32, _P, _P, c_int64, c_float, c_float, c_float, c_int32,
                                             _P, _P, _P, _P]),
    "gs_rigid_transform_fwd": (ctypes.c_int, [c_int64, c_int32] + [_P] * 8),
    "gs_rigid_transform_bwd": (ctypes.c_int, [c_int64, c_int32] + [_P] * 11),
    "gs_activate_fwd": (ctypes.c_int, [c_int64, c_int32, c_int32] + [_P] * 7 + [c_int32] + [_P] * 9),
    "gs_activate_bwd": (ctypes.c_int, [c_int64, c_int32, c_int32] + [_P] * 6 + [c_int32] + [_P] * 12),
    "gs_refine_decide": (ctypes.c_int, [c_int64, c_int32, c_int32, c_int32] + [c_float] * 6 + [_P] * 8),
    "gs_refine_plan": (ctypes.c_int, [c_int64, c_int32, _P, _P, _P, _P, _P, _P]),
    "gs_refine_gather_rows": (ctypes.c_int, [c_int64, c_int32, c_int32, _P, _P, _P, _P, _P, _P]),
    "gs_refine_split_means": (ctypes.c_int, [c_int64, c_int32] + [_P] * 9),
    "gs_l1_ssim_workspace_floats": (c_int64, [c_int32, c_int32]),
    "gs_l1_ssim_fwd": (ctypes.c_int, [c_int32, c_int32, _P, _P, _P, _P, _P, c_int64, _P, _P]),
    "gs_l1_ssim_bwd": (ctypes.c_int, [c_int32, c_int32, _P, _P, _P, _P, _P, _P, _P]),
    "gs_pixel_losses_workspace_floats": (c_int64, []),
    "gs_pixel_losses_fwd": (ctypes.c_int, [c_int64, _P, _P, _P, _P, _P, c_float, _P, c_int64, _P, _P]),
    "gs_pixel_losses_bwd": (ctypes.c_int, [c_int64, _P, _P, _P, _P, _P, c_float, _P, _P, _P, _P, _P]),
    "gs_pixel_losses_ex_fwd": (ctypes.c_int, [c_int64, POINTER(PixelLossOpts)] + [_P] * 8 + [c_int64, _P, _P]),
    "gs_pixel_losses_ex_bwd": (ctypes.c_int, [c_int64, POINTER(PixelLossOpts)] + [_P] * 11),
    "gs_inv_depth_smooth_fwd": (ctypes.c_int, [c_int32, c_int32, _P, _P, _P, c_int64, _P, _P]),
    "gs_inv_depth_smooth_bwd": (ctypes.c_int, [c_int32, c_int32, _P, _P, _P, _P, _P]),
    "gs_region_l1_fwd": (ctypes.c_int, [c_int64, _P, _P, _P, _P, c_float, _P, c_int64, _P, _P]),
    "gs_region_l1_bwd": (ctypes.c_int, [c_int64, _P, _P, _P, _P, c_float, _P, _P, _P, _P]),
    "gs_affine_reg": (ctypes.c_int, [c_int32, c_int32, _P, _P, _P, _P, _P]),
    "gs_opacity_reset": (ctypes.c_int, [c_int64, c_float, _P, _P, _P, _P]),
    "gs_cube_texture_fwd": (ctypes.c_int, [c_int64, c_int32, c_int32, _P, _P, _P, _P, _P]),
    "gs_cube_texture_bwd": (ctypes.c_int, [c_int64, c_int32, c_int32, _P, _P, _P, _P, _P]),
    "gs_mark_visible": (ctypes.c_int, [c_int64, _P, _P, _P]),
    "gs_compact_visible": (ctypes.c_int, [c_int32, c_int32] + [_P] * 10 + [_P]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES.keys())

_lib: Optional[ctypes.CDLL] = None


def lib_path() -> str:
    """In-tree build output; GSPLAT_MI355X_LIB points at another build of the same ABI (kernel A/B experiments)."""
    return os.path.normpath(os.environ.get("GSPLAT_MI355X_LIB") or _LIB_PATH)


def lib() -> ctypes.CDLL:
    """Load (once) and return the native library; raise loudly if it is not built."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise RuntimeError(
                f"{_LIB_NAME} not found at {path}: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C 3d-reconstruction_amd/csrc). There is no CPU/PyTorch fallback for this path."
            )
        handle = ctypes.CDLL(path)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the ABI is incomplete
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = lib().gs_error_string(int(status))
        raise RuntimeError(f"{what} failed with status {status}: {msg.decode() if msg else '?'}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device pointer of a contiguous tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def require_gpu(*tensors: torch.Tensor) -> torch.device:
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "gsplat (MI355X build): tensors must live on a HIP device; this build has no CPU path "
                f"(got a tensor on {t.device})."
            )
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"gsplat (MI355X build): tensors on different devices: {dev} vs {t.device}")
    if dev is None:
        raise RuntimeError("gsplat (MI355X build): no tensor given")
    return dev


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_handle(device: torch.device) -> int:
    """hipStream_t of torch's current stream on `device` (the raw accessor skips building a Stream object: this is
    called once per native launch)."""
    if _raw_stream is not None and device.index is not None:
        return _raw_stream(device.index)
    return torch.cuda.current_stream(device).cuda_stream


class _NoGuard:
    __slots__ = ()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_no_guard = _NoGuard()


def device_guard(device: torch.device):
    """`with device_guard(dev):` makes `dev` the current HIP device for a native launch; free when it already is."""
    if device.index is None or device.index == torch.cuda.current_device():
        return _no_guard
    return torch.cuda.device(device)


_side_streams = {}


def side_stream(device: torch.device) -> "torch.cuda.Stream":
    """One auxiliary HIP stream per device for copies/fills that overlap a compute-bound kernel."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    st = _side_streams.get(key)
    if st is None:
        st = _side_streams[key] = torch.cuda.Stream(device=device)
    return st


# ---- optional per-entry-point timing with HIP events on the launch stream (used by bench.py) ----------------
_prof = None
_prof_only = None


def profiling_start(only=None) -> None:
    """Start bracketing native launches with HIP events.  `only`: iterable of entry-point labels to restrict to
    (every bracket costs a few microseconds of stream bubble, so a timed region brackets one kernel only)."""
    global _prof, _prof_only
    _prof = {}
    _prof_only = set(only) if only is not None else None


def profiling_active(label: Optional[str] = None) -> bool:
    """True while launches labelled `label` are being bracketed (such a launch must stay on the current stream so the
    brackets see it; every other launch keeps its production stream).  Without a label: any timing on at all."""
    if _prof is None:
        return False
    return label is None or _prof_only is None or label in _prof_only


def profiling_stop():
    """Returns {entry point: (calls, mean ms)}; synchronises the device once."""
    global _prof
    rec, _prof = _prof, None
    if not rec:
        return {}
    torch.cuda.synchronize()
    return {k: (len(v), sum(a.elapsed_time(b) for a, b in v) / len(v)) for k, v in rec.items()}


class timed:
    """`with timed(name, device):` brackets a native launch with two events on the current stream."""

    __slots__ = ("name", "dev", "a")

    def __init__(self, name: str, dev: torch.device):
        self.name, self.dev, self.a = name, dev, None

    def __enter__(self):
        if _prof is not None and (_prof_only is None or self.name in _prof_only):
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record(torch.cuda.current_stream(self.dev))
        return self

    def __exit__(self, *exc):
        if self.a is not None:
            b = torch.cuda.Event(enable_timing=True)
            b.record(torch.cuda.current_stream(self.dev))
            _prof.setdefault(self.name, []).append((self.a, b))
        return False


def f32c(t: torch.Tensor) -> torch.Tensor:
    """fp32 + contiguous (no copy when already so)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()
