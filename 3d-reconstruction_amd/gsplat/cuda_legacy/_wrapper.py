"""`gsplat.cuda_legacy._wrapper.num_sh_bases` (imported at
/root/reference/3D-Reconstruction-main/models/gaussians/basics.py:13; used at
models/gaussians/vanilla.py:61,85, models/nodes/rigid.py:93, smpl.py:182,623)."""


def num_sh_bases(degree: int) -> int:
    """Number of real SH basis functions up to `degree` (0..4): 1, 4, 9, 16, 25."""
    if degree == 0:
        return 1
    if degree == 1:
        return 4
    if degree == 2:
        return 9
    if degree == 3:
        return 16
    assert degree == 4, "SH degree must be in 0..4"
    return 25
