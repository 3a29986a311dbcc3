"""CPU oracle for the gsplat `rasterization()` hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under `3d-reconstruction_amd/` may import this
package; only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
leg use it, and only as the checker / the reported CPU baseline.

PARITY UNPINNED: the arithmetic of this path lives in the third-party package
`gsplat` (nerfstudio-project/gsplat, 1.0 <= version < 1.5 implied by the call
sites, not vendored, not pinned, not installable here), and the reference ships
no tests or golden vectors for it.  This oracle restates gsplat's published
algorithm and is anchored on the reference's call sites
(/root/reference/3D-Reconstruction-main/models/trainers/base.py:393-408,
models/gaussians/vanilla.py:359-395) plus the two in-tree anchors that do exist:
utils/geometry.py:116-152 (`quat_to_rotmat`, golden vectors under tests/golden/)
and models/gaussians/basics.py:76-89 (SH band-0 constant C0).
"""
from .gsplat_ref import (  # noqa: F401
    num_sh_bases,
    quat_to_rotmat,
    spherical_harmonics,
    fully_fused_projection,
    isect_tiles,
    isect_offset_encode,
    rasterize_to_pixels,
    rasterization,
)
