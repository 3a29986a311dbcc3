"""MI355X (gfx950) native drop-in for the four `gsplat` symbols the reference imports
(/root/reference/3D-Reconstruction-main/models/gaussians/basics.py:12-15):

    from gsplat.rendering import rasterization
    from gsplat.cuda_legacy._wrapper import num_sh_bases
    from gsplat.cuda_legacy._torch_impl import quat_to_rotmat
    from gsplat.cuda._wrapper import spherical_harmonics

gsplat's stage-level functions (fully_fused_projection, isect_tiles, isect_offset_encode, rasterize_to_pixels) are
exported too, with gsplat 1.x's signatures, for code that drives the stages itself.

Put the directory that contains this package (`3d-reconstruction_amd/`) on PYTHONPATH ahead of any
other gsplat; see INTEGRATION.md.
"""
from .rendering import rasterization  # noqa: F401
from .cuda._wrapper import (  # noqa: F401
    spherical_harmonics,
    fully_fused_projection,
    isect_tiles,
    isect_offset_encode,
    rasterize_to_pixels,
)
from .cuda_legacy._wrapper import num_sh_bases  # noqa: F401
from .cuda_legacy._torch_impl import quat_to_rotmat  # noqa: F401

__version__ = "0.1.0+mi355x"
