"""gsplat.densify.refine_gaussians against a replay of the reference's own lines on the same device:
/root/reference/3D-Reconstruction-main/models/gaussians/vanilla.py:208-265 (`refinement_after`: split + duplicate, then
cull on the grown arrays), :285-312 `cull_gaussians`, :314-342 `split_gaussians`, :344-357 `dup_gaussians`, and the Adam
moment surgery models/gaussians/basics.py:162-206 (`dup_in_optim`, `remove_from_optim`).  Row structure (who survives, in
which order) must be identical; copied tensors bit-identical; the split children's means and the shrunk scales to 1e-6
(fp32, transcendental functions)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

_KEYS = ("means", "features_dc", "features_rest", "opacities", "scales", "quats")


def _quat_to_rotmat(q):
    w, x, y, z = torch.unbind(q / q.norm(dim=-1, keepdim=True), dim=-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=-1).reshape(-1, 3, 3)


class _RefClass:
    """The reference's lines with `self.` state; Adam moments carried as plain dicts keyed like the params."""

    def __init__(self, p, adam, stats, cfg):
        self.p = {k: v.clone() for k, v in p.items()}
        self.adam = {k: {m: t.clone() for m, t in st.items()} for k, st in adam.items()}
        self.xys_grad_norm, self.vis_counts, self.max_2Dsize = (t.clone() for t in stats)
        self.cfg = cfg

    def get_scaling(self):
        s = self.p["scales"]
        if self.cfg.ball_gaussians:
            return torch.exp(s).repeat(1, 3)
        if self.cfg.gaussian_2d:
            e = torch.exp(s)
            return torch.cat([e[..., :2], torch.zeros_like(e[..., :1])], dim=-1)
        return torch.exp(s)

    def split_gaussians(self, split_mask, samps):
        n_splits = int(split_mask.sum().item())
        centered = torch.randn((samps * n_splits, 3), device=split_mask.device)
        scaled = self.get_scaling()[split_mask].repeat(samps, 1) * centered
        quats = self.p["quats"][split_mask]
        quats = quats / quats.norm(dim=-1, keepdim=True)
        rots = _quat_to_rotmat(quats.repeat(samps, 1))
        rotated = torch.bmm(rots, scaled[..., None]).squeeze(-1)
        new_means = rotated + self.p["means"][split_mask].repeat(samps, 1)
        new_dc = self.p["features_dc"][split_mask].repeat(samps, 1)
        new_rest = self.p["features_rest"][split_mask].repeat(samps, 1, 1)
        new_opac = self.p["opacities"][split_mask].repeat(samps, 1)
        new_scales = torch.log(torch.exp(self.p["scales"][split_mask]) / 1.6).repeat(samps, 1)
        self.p["scales"][split_mask] = torch.log(torch.exp(self.p["scales"][split_mask]) / 1.6)
        new_quats = self.p["quats"][split_mask].repeat(samps, 1)
        return dict(means=new_means, features_dc=new_dc, features_rest=new_rest, opacities=new_opac,
                    scales=new_scales, quats=new_quats)

    def dup_in_optim(self, idcs, n):
        for k, st in self.adam.items():
            for m in st:
                rep = (n,) + tuple(1 for _ in range(st[m].dim() - 1))
                st[m] = torch.cat([st[m], torch.zeros_like(st[m][idcs]).repeat(*rep)], dim=0)

    def refinement_after(self):
        cfg = self.cfg
        if cfg.do_densify:
            avg = self.xys_grad_norm / self.vis_counts
            high = (avg > cfg.densify_grad_thresh).squeeze()
            splits = (self.get_scaling().max(dim=-1).values > cfg.densify_size_thresh * cfg.scene_scale).squeeze()
            if cfg.split_screen_size is not None:
                splits |= (self.max_2Dsize > cfg.split_screen_size).squeeze()
            splits &= high
            new_s = self.split_gaussians(splits, cfg.n_split_samples)
            dups = (self.get_scaling().max(dim=-1).values <= cfg.densify_size_thresh * cfg.scene_scale).squeeze()
            dups &= high
            new_d = {k: self.p[k][dups] for k in _KEYS}
            for k in _KEYS:
                self.p[k] = torch.cat([self.p[k], new_s[k], new_d[k]], dim=0)
            self.max_2Dsize = torch.cat([self.max_2Dsize, torch.zeros_like(new_s["scales"][:, 0]),
                                         torch.zeros_like(new_d["scales"][:, 0])], dim=0)
            self.dup_in_optim(torch.where(splits)[0], cfg.n_split_samples)
            self.dup_in_optim(torch.where(dups)[0], 1)
        if cfg.do_cull:
            culls = (torch.sigmoid(self.p["opacities"]) < cfg.cull_alpha_thresh).squeeze()
            if cfg.cull_scale_thresh is not None:
                toobigs = (torch.exp(self.p["scales"]).max(dim=-1).values > cfg.cull_scale_thresh * cfg.scene_scale).squeeze()
                culls = culls | toobigs
                if cfg.cull_screen_size is not None:
                    culls = culls | (self.max_2Dsize > cfg.cull_screen_size).squeeze()
            for k in _KEYS:
                self.p[k] = self.p[k][~culls]
            for st in self.adam.values():
                for m in st:
                    st[m] = st[m][~culls]


def _scene(dev, n, ball=False, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g, device=dev)
    p = dict(means=r(n, 3) * 10, features_dc=r(n, 3), features_rest=r(n, 15, 3) * 0.3, opacities=r(n, 1) * 3.0,
             scales=r(n, 1 if ball else 3) * 1.0 - 3.0, quats=r(n, 4))
    adam = {k: {"exp_avg": torch.randn_like(v), "exp_avg_sq": torch.rand_like(v)} for k, v in p.items()}
    vis = torch.randint(1, 40, (n,), generator=g, device=dev).float()
    gn = torch.rand(n, generator=g, device=dev) * 0.0006 * vis
    m2 = torch.rand(n, generator=g, device=dev) * 0.2
    return p, adam, (gn, vis, m2)


@pytest.mark.parametrize("name,kw", [
    ("early", dict()),                                                                   # step <= reset interval
    ("screen", dict(split_screen_size=0.05, cull_scale_thresh=0.5, cull_screen_size=0.15)),
    ("late", dict(cull_scale_thresh=0.5)),                                               # past stop_screen_size_at
    ("cull_only", dict(do_densify=False, cull_scale_thresh=0.5, cull_screen_size=0.15)),
    ("densify_only", dict(do_cull=False, split_screen_size=0.05, n_split_samples=3)),
    ("ball", dict(ball_gaussians=True, split_screen_size=0.05, cull_scale_thresh=0.5, cull_screen_size=0.15)),
    ("2d", dict(gaussian_2d=True, cull_scale_thresh=0.2, cull_screen_size=0.15)),
    ("nothing", dict(do_densify=False, do_cull=False)),
])
def test_refine_matches_reference_lines(gpu_device, name, kw):
    from gsplat.densify import DensifyState, RefineConfig, refine_gaussians

    dev = gpu_device
    cfg = RefineConfig(densify_grad_thresh=0.0003, densify_size_thresh=0.05, scene_scale=1.3, cull_alpha_thresh=0.05,
                       **kw)
    n = 20_011
    p, adam, stats = _scene(dev, n, ball=cfg.ball_gaussians, seed=len(name))
    ref = _RefClass(p, adam, stats, cfg)
    torch.manual_seed(1234)
    ref.refinement_after()
    torch.manual_seed(1234)
    state = DensifyState(*stats)
    new_p, new_adam, info = refine_gaussians(p, state, cfg, adam=adam)
    print(f"[refine {name}] {info}")
    assert info["n_after"] == ref.p["means"].shape[0], (info, ref.p["means"].shape)
    if cfg.do_densify:
        assert info["n_splits"] > 100 and info["n_kept_dups"] > 100
    if cfg.do_cull:
        assert info["n_culled_old"] > 100
    for k in _KEYS:
        assert new_p[k].shape == ref.p[k].shape, (k, new_p[k].shape, ref.p[k].shape)
        if k in ("means", "scales"):
            assert torch.allclose(new_p[k], ref.p[k], rtol=1e-6, atol=1e-6), (k, float((new_p[k] - ref.p[k]).abs().max()))
        else:
            assert torch.equal(new_p[k], ref.p[k]), k
        for m in ("exp_avg", "exp_avg_sq"):
            assert torch.equal(new_adam[k][m], ref.adam[k][m]), (k, m)
    # the inputs are untouched (the reference shrinks split parents in place; here they leave in the new tensor)
    p2, _, _ = _scene(dev, n, ball=cfg.ball_gaussians, seed=len(name))
    assert all(torch.equal(p[k], p2[k]) for k in _KEYS)


def test_refine_empty_and_all_culled(gpu_device):
    from gsplat.densify import DensifyState, RefineConfig, refine_gaussians

    dev = gpu_device
    p, adam, stats = _scene(dev, 0)
    new_p, new_adam, info = refine_gaussians(p, DensifyState(*stats), RefineConfig(), adam=adam)
    assert info["n_after"] == 0 and new_p["features_rest"].shape == (0, 15, 3)
    p, adam, stats = _scene(dev, 1000)
    p["opacities"].fill_(-20.0)
    new_p, new_adam, info = refine_gaussians(p, DensifyState(*stats), RefineConfig(cull_alpha_thresh=0.005), adam=adam)
    assert info["n_after"] == 0 and new_p["means"].shape == (0, 3) and new_adam["quats"]["exp_avg"].shape == (0, 4)
