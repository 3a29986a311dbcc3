"""Render-only trajectory harness (gsplat.harness.render_trajectory; the reference's per-frame loop is
/root/reference/3D-Reconstruction-main/models/video_utils.py:647-706): frames equal one rasterization() call per view
followed by the reference's host-side uint8 conversion; rank shards tile the trajectory."""
import numpy as np
import pytest
import torch

import synth_scene

pytestmark = pytest.mark.gpu


def test_trajectory_matches_per_view_calls(gpu_device):
    from gsplat.harness import render_trajectory
    from gsplat.rendering import rasterization

    dev = gpu_device
    W, H, V = 176, 100, 7
    g = {k: v.to(dev) for k, v in synth_scene.make_gaussians(30_000, sh_degree=3, extent=15.0).items()}
    cams = [synth_scene.make_cameras(1, W, H, frame=f) for f in range(V)]
    vm = torch.cat([c[0] for c in cams]).to(dev)
    Ks = torch.cat([c[1] for c in cams]).to(dev)
    ref_u8, ref_f = [], []
    with torch.no_grad():
        for v in range(V):
            rc, _, _ = rasterization(g["means"], g["quats"], g["scales"], g["opacities"], g["colors"], vm[v:v + 1],
                                     Ks[v:v + 1], W, H, near_plane=0.1, sh_degree=3, packed=True, render_mode="RGB")
            rgb = rc[0].cpu().numpy()  # what the reference does per frame (video_utils.py:682, 694-700)
            ref_f.append(rgb)
            ref_u8.append((rgb.clip(min=1.e-6, max=1 - 1.e-6) * 255).astype(np.uint8))
    frames, rng = render_trajectory(g["means"], g["quats"], g["scales"], g["opacities"], g["colors"], vm, Ks, W, H,
                                    sh_degree=3, views_per_call=3, near_plane=0.1)
    assert rng == range(0, V) and frames.shape == (V, H, W, 3) and frames.dtype == torch.uint8 and frames.is_pinned()
    assert np.array_equal(frames.numpy(), np.stack(ref_u8))
    # batches alternating over two and three HIP streams: the same bits, twice in a row (the second pass reuses the
    # lanes' allocator pools), and the caller's stream may overwrite the inputs right behind the call
    for n_streams in (2, 3):
        for _ in range(2):
            vm2, Ks2 = vm.clone(), Ks.clone()
            fr, _ = render_trajectory(g["means"], g["quats"], g["scales"], g["opacities"], g["colors"], vm2, Ks2, W, H,
                                      sh_degree=3, views_per_call=2, near_plane=0.1, streams=n_streams)
            vm2.zero_(), Ks2.zero_()
            assert np.array_equal(fr.numpy(), np.stack(ref_u8)), n_streams
    # float output, dense (unpacked) calls, two ranks
    parts = []
    for r in range(2):
        fr, rg = render_trajectory(g["means"], g["quats"], g["scales"], g["opacities"], g["colors"], vm, Ks, W, H,
                                   sh_degree=3, views_per_call=2, rank=r, world_size=2, as_uint8=False, packed=False,
                                   near_plane=0.1)
        assert fr.shape[0] == len(rg)
        parts.append((rg, fr))
    assert parts[0][0] == range(0, 4) and parts[1][0] == range(4, 7)
    got = torch.cat([fr for _, fr in parts]).numpy()
    assert np.array_equal(got, np.stack(ref_f))


def test_config5_trajectory_at_full_size_views_spread_over_the_trajectory(gpu_device):
    """BASELINE configs[4] at its own size (6 M Gaussians, SH 3, 1920x1080, packed, no gradients): 8 views taken from all
    over the 600-view trajectory (rank 5 of 8 renders views 375..449; here frames 0, 37, ..., 119 of the 120-frame ego
    path and two ranks' worth of sharding), through the harness in batches of 4 against one dense rasterization() call per
    view + the reference's host-side conversion (video_utils.py:694-700): bit-equal uint8 frames."""
    from gsplat.harness import render_trajectory
    from gsplat.rendering import rasterization

    dev = gpu_device
    N, W, H = 6_000_000, 1920, 1080
    g = {k: v.to(dev) for k, v in synth_scene.make_gaussians(N, sh_degree=3, seed=42).items()}
    frames_of = [0, 17, 37, 59, 60, 83, 101, 119]
    cams = [synth_scene.make_cameras(1, W, H, frame=f, origin_x=-30.0) for f in frames_of]  # (inside the +-40 m slab)
    vm = torch.cat([c[0] for c in cams]).to(dev)
    Ks = torch.cat([c[1] for c in cams]).to(dev)
    V = len(frames_of)
    got = []
    for r in range(2):
        fr, rg = render_trajectory(g["means"], g["quats"], g["scales"], g["opacities"], g["colors"], vm, Ks, W, H,
                                   sh_degree=3, views_per_call=4, rank=r, world_size=2, near_plane=0.1)
        assert rg == range(4 * r, 4 * r + 4) and fr.dtype == torch.uint8
        got.append(fr.numpy().copy())
    got = np.concatenate(got)
    assert got.shape == (V, H, W, 3)
    with torch.no_grad():
        for v in range(V):
            rc, ra, _ = rasterization(g["means"], g["quats"], g["scales"], g["opacities"], g["colors"], vm[v:v + 1],
                                      Ks[v:v + 1], W, H, near_plane=0.1, sh_degree=3, packed=False, render_mode="RGB")
            ref = (rc[0].cpu().numpy().clip(min=1.e-6, max=1 - 1.e-6) * 255).astype(np.uint8)
            assert float(ra.mean()) > 0.5, "the view must show the scene"
            assert np.array_equal(got[v], ref), f"view {v} (frame {frames_of[v]})"
