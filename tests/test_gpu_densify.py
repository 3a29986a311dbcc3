"""gsplat.densify.accumulate_stats against a torch replay of the reference's bookkeeping:
/root/reference/3D-Reconstruction-main/models/trainers/base.py:279-297 (gradient scaling) +
models/gaussians/vanilla.py:154-182 (`after_train`), three training steps, one camera (the reference's batch size)."""
import pytest
import torch

import synth_scene

pytestmark = pytest.mark.gpu


class _RefStats:
    """vanilla.py:154-182 with filter_mask = all points."""

    def __init__(self, n):
        self.num_points, self.xys_grad_norm, self.vis_counts, self.max_2Dsize = n, None, None, None

    def after_train(self, radii, xys_grad, last_size):
        visible_mask = (radii > 0).flatten()
        full_mask = visible_mask.clone()
        grads = xys_grad.norm(dim=-1)
        if self.xys_grad_norm is None:
            self.xys_grad_norm = grads.clone()
            self.vis_counts = torch.ones_like(self.xys_grad_norm)
        else:
            self.vis_counts[full_mask] = self.vis_counts[full_mask] + 1
            self.xys_grad_norm[full_mask] = grads[visible_mask] + self.xys_grad_norm[full_mask]
        if self.max_2Dsize is None:
            self.max_2Dsize = torch.zeros(self.num_points, device=radii.device, dtype=torch.float32)
        newradii = radii[visible_mask]
        self.max_2Dsize[full_mask] = torch.maximum(self.max_2Dsize[full_mask], newradii / float(last_size))


@pytest.mark.parametrize("absgrad", [True, False])
def test_accumulate_matches_reference_bookkeeping(gpu_device, absgrad):
    from gsplat.densify import DensifyState, accumulate_stats
    from gsplat.rendering import rasterization

    dev = gpu_device
    N, W, H = 20_000, 200, 120
    g = {k: v.to(dev) for k, v in synth_scene.make_gaussians(N, sh_degree=None, extent=15.0).items()}
    ref, state = _RefStats(N), DensifyState()
    for step in range(3):
        vm, Ks = (t.to(dev) for t in synth_scene.make_cameras(1, W, H, frame=3 * step))
        p = {k: v.clone().requires_grad_(True) for k, v in g.items()}
        rc, ra, info = rasterization(p["means"], p["quats"], p["scales"], p["opacities"], p["colors"], vm, Ks, W, H,
                                     near_plane=0.1, packed=False, absgrad=True, render_mode="RGB+ED")
        info["means2d"].retain_grad()
        (rc.square().sum() + ra.sum()).backward()
        # base.py:279-297
        grads = (info["means2d"].absgrad if absgrad else info["means2d"].grad).clone()
        grads[..., 0] *= info["width"] / 2.0 * 1
        grads[..., 1] *= info["height"] / 2.0 * 1
        ref.after_train(info["radii"][0], grads[0], max(info["width"], info["height"]))
        accumulate_stats(info, state, absgrad=absgrad, batch_size=1)
        assert torch.allclose(state.xys_grad_norm, ref.xys_grad_norm, rtol=1e-6, atol=1e-9), step
        assert torch.equal(state.vis_counts, ref.vis_counts), step
        assert torch.equal(state.max_2Dsize, ref.max_2Dsize), step
    assert float(state.vis_counts.max()) >= 3 and float(state.max_2Dsize.max()) > 0


def test_accumulate_multi_camera_is_sum_of_cameras(gpu_device):
    """C > 1 (camera-sharded ranks all-reduce exactly these): per-camera terms add, counts per visible camera."""
    from gsplat.densify import DensifyState, accumulate_stats
    from gsplat.rendering import rasterization

    dev = gpu_device
    N, W, H, C = 8_000, 160, 96, 3
    g = {k: v.to(dev) for k, v in synth_scene.make_gaussians(N, sh_degree=None, extent=12.0).items()}
    vm, Ks = (t.to(dev) for t in synth_scene.make_cameras(C, W, H))
    state = DensifyState()
    for step in range(2):
        p = {k: v.clone().requires_grad_(True) for k, v in g.items()}
        rc, ra, info = rasterization(p["means"], p["quats"], p["scales"], p["opacities"], p["colors"], vm, Ks, W, H,
                                     near_plane=0.1, packed=False, absgrad=True, render_mode="RGB")
        (rc.square().sum() + ra.sum()).backward()
        prev = None if step == 0 else (state.xys_grad_norm.clone(), state.vis_counts.clone(), state.max_2Dsize.clone())
        accumulate_stats(info, state)
        ag = info["means2d"].absgrad.clone()
        ag[..., 0] *= W / 2.0
        ag[..., 1] *= H / 2.0
        nrm, vis = ag.norm(dim=-1), info["radii"] > 0
        if prev is None:
            assert torch.allclose(state.xys_grad_norm, nrm.sum(0), rtol=1e-6)
            assert torch.equal(state.vis_counts, torch.ones(N, device=dev))
            assert torch.equal(state.max_2Dsize, (info["radii"] * vis).amax(0).float() / max(W, H))
        else:
            assert torch.allclose(state.xys_grad_norm, prev[0] + (nrm * vis).sum(0), rtol=1e-6)
            assert torch.equal(state.vis_counts, prev[1] + vis.sum(0))
            assert torch.equal(state.max_2Dsize, torch.maximum(prev[2], (info["radii"] * vis).amax(0).float() / max(W, H)))
