"""`gsplat.cuda_legacy._torch_impl.quat_to_rotmat` (imported at
/root/reference/3D-Reconstruction-main/models/gaussians/basics.py:14; used at
models/gaussians/vanilla.py:327 and pvg.py:307 inside split_gaussians, on [n,4] tensors every
refine_interval steps -- not a hot op, so it is plain differentiable torch code on whatever device the
input lives on; same wxyz / normalising convention as the in-tree twin utils/geometry.py:116-152)."""
import torch
from torch import Tensor


def quat_to_rotmat(quat: Tensor) -> Tensor:
    assert quat.shape[-1] == 4, quat.shape
    w, x, y, z = torch.unbind(torch.nn.functional.normalize(quat, dim=-1), dim=-1)
    mat = torch.stack(
        [
            1 - 2 * (y**2 + z**2), 2 * (x * y - w * z), 2 * (x * z + w * y),
            2 * (x * y + w * z), 1 - 2 * (x**2 + z**2), 2 * (y * z - w * x),
            2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x**2 + y**2),
        ],
        dim=-1,
    )
    return mat.reshape(quat.shape[:-1] + (3, 3))
