"""The widened pieces composed into the reference's training step (models/trainers/base.py:forward / compute_losses /
postprocess_per_train_step, models/gaussians/vanilla.py:get_gaussians / after_train / refinement_after):
raw Parameters -> gsplat.prepass.collect_gaussians -> gsplat.rendering.rasterization -> gsplat.losses.l1_ssim ->
backward -> gsplat.densify.accumulate_stats -> Adam, with one gsplat.densify.refine_gaussians (+ Adam-moment surgery in
the reference's dup_in_optim / remove_from_optim style) in the middle.  A small scene is fitted to renders of a target
scene: the loss must fall, everything must stay finite, and the refinement must change the point count."""
import pytest
import torch
from torch.nn import Parameter

import synth_scene

pytestmark = pytest.mark.gpu

_KEYS = ("means", "features_dc", "features_rest", "opacities", "scales", "quats")


def _raw_scene(n, seed, dev):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    means = torch.stack([(torch.rand(n, generator=g) - 0.5) * 8, (torch.rand(n, generator=g) - 0.5) * 4,
                         torch.rand(n, generator=g) * 6 + 3], dim=-1)
    return dict(means=means, features_dc=r(n, 3) * 1.0, features_rest=r(n, 3, 3) * 0.1, opacities=r(n, 1) + 1.0,
                scales=r(n, 3) * 0.3 - 1.6, quats=r(n, 4))


def test_fit_small_scene_with_all_fused_pieces(gpu_device):
    from gsplat.densify import DensifyState, RefineConfig, accumulate_stats, refine_gaussians
    from gsplat.losses import l1_ssim
    from gsplat.prepass import GaussianClass, collect_gaussians
    from gsplat.rendering import rasterization

    dev = gpu_device
    W, H, n_cams = 160, 96, 3
    vm = torch.eye(4).repeat(n_cams, 1, 1)
    vm[1, 0, 3], vm[2, 0, 3] = 0.6, -0.6  # three cameras side by side looking down +z
    Ks = torch.tensor([[W / 2.0, 0, W / 2.0], [0, W / 2.0, H / 2.0], [0, 0, 1.0]]).repeat(n_cams, 1, 1)
    vm, Ks = vm.to(dev), Ks.to(dev)
    c2ws = torch.linalg.inv(vm)

    def render(raw, c, step_flag=None):
        cls = GaussianClass(**raw, sh_degree=1, sh_degree_to_use=1)
        gs, _, flag = collect_gaussians([cls], c2ws[c], finite_flag=step_flag)
        rc, ra, info = rasterization(means=gs["_means"], quats=gs["_quats"], scales=gs["_scales"],
                                     opacities=gs["_opacities"].squeeze(), colors=gs["_rgbs"], viewmats=vm[c][None],
                                     Ks=Ks[c][None], width=W, height=H, packed=False, absgrad=True, sparse_grad=False,
                                     rasterize_mode="classic", near_plane=0.1, far_plane=1e10, render_mode="RGB+ED")
        return torch.clamp(rc[0, ..., :3], max=1.0), info, flag

    target_raw = {k: v.to(dev) for k, v in _raw_scene(1500, 1, dev).items()}
    with torch.no_grad():
        targets = [render(target_raw, c)[0] for c in range(n_cams)]

    start = _raw_scene(1500, 1, dev)
    g = torch.Generator().manual_seed(7)
    start["means"] = start["means"] + 0.05 * torch.randn(1500, 3, generator=g)
    start["features_dc"] = start["features_dc"] * 0.2
    start["scales"] = start["scales"] + 0.2
    params = {k: Parameter(v.to(dev)) for k, v in start.items()}
    lrs = dict(means=2e-3, features_dc=2e-2, features_rest=1e-3, opacities=5e-2, scales=5e-3, quats=1e-3)
    opt = torch.optim.Adam([{"params": [params[k]], "lr": lrs[k], "name": k} for k in _KEYS], eps=1e-15)
    state = DensifyState()
    flag = None
    losses, n_points = [], []
    for step in range(60):
        c = step % n_cams
        rgb, info, flag = render(params, c, flag)
        info["means2d"].retain_grad()
        l1, ss = l1_ssim(rgb, targets[c])
        loss = 0.8 * l1 + 0.2 * (1.0 - ss)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        accumulate_stats(info, state, absgrad=True)
        opt.step()
        losses.append(float(loss.detach()))
        n_points.append(params["means"].shape[0])
        if step == 35:
            cfg = RefineConfig(densify_grad_thresh=float(torch.quantile(state.xys_grad_norm / state.vis_counts, 0.9)),
                               densify_size_thresh=0.2, scene_scale=1.0, cull_alpha_thresh=0.05, n_split_samples=2)
            adam = {grp["name"]: {m: opt.state[grp["params"][0]][m] for m in ("exp_avg", "exp_avg_sq")}
                    for grp in opt.param_groups}
            new_p, new_adam, rinfo = refine_gaussians({k: v.data for k, v in params.items()}, state, cfg, adam=adam)
            for grp in opt.param_groups:  # basics.py:162-206: move the state onto the new Parameter
                k = grp["name"]
                st = opt.state.pop(grp["params"][0])
                params[k] = Parameter(new_p[k])
                st["exp_avg"], st["exp_avg_sq"] = new_adam[k]["exp_avg"], new_adam[k]["exp_avg_sq"]
                grp["params"] = [params[k]]
                opt.state[params[k]] = st
            state = DensifyState()  # vanilla.py:281-283
            assert rinfo["n_after"] != rinfo["n_before"] and rinfo["n_splits"] + rinfo["n_kept_dups"] > 0, rinfo
    flag.check()  # no NaN / Inf in any step's activated Gaussians (one synchronisation for the whole run)
    first, last = sum(losses[:3]) / 3, sum(losses[-3:]) / 3
    print(f"[fit] loss {first:.4f} -> {last:.4f}, points {n_points[0]} -> {n_points[-1]}")
    assert all(l == l for l in losses)
    assert last < 0.75 * first, (first, last)
    assert n_points[-1] != n_points[0]
    for k in _KEYS:
        assert torch.isfinite(params[k]).all(), k
