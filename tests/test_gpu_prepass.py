"""gsplat.prepass.collect_gaussians (gs_activate_fwd / _bwd) against a replay of the reference's own lines:
/root/reference/3D-Reconstruction-main/models/gaussians/vanilla.py:359-393 (`get_gaussians`: SH colours + 0.5 / clamp or
sigmoid, sigmoid opacities, exp scales incl. the ball / 2-D variants of :112-128, normalised quats, NaN/Inf checks),
models/nodes/rigid.py:493 (validity mask on the opacities) and models/trainers/base.py:355-369 (concatenation + labels).
The replay runs in fp64 with the oracle's spherical_harmonics; tolerance: 2e-6 absolute + 1e-5 relative on outputs and
gradients (fp32 kernels).  Colour channels whose pre-clamp value lies within 1e-5 of 0 or 1 may clamp differently in
fp32 and are left out of the colour-gradient comparison (counted, budget 0.1 %)."""
import pytest
import torch

from oracle import gsplat_ref as O

pytestmark = pytest.mark.gpu


def _make_class(n, K, sh_degree, use, seed, ball=False, two_d=False, mask=False, label=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    return dict(
        means=r(n, 3) * 10.0, features_dc=r(n, 3) * 2.0, features_rest=r(n, K - 1, 3) * 0.3, opacities=r(n, 1) * 2.0,
        scales=r(n, 1 if ball else 3) * 0.7 - 2.0, quats=r(n, 4) * 1.7, sh_degree=sh_degree, sh_degree_to_use=use,
        label=label, opacity_mask=(torch.rand(n, generator=g) > 0.2) if mask else None, ball_gaussians=ball,
        gaussian_2d=two_d)


def _reference_get_gaussians(c, cam_t):
    """vanilla.py:359-386 (+ rigid.py:493 when a mask is given), tensors as given (fp64 leaves)."""
    colors = torch.cat((c["features_dc"][:, None, :], c["features_rest"]), dim=1)
    if c["sh_degree"] > 0:
        viewdirs = c["means"].detach() - cam_t
        viewdirs = viewdirs / viewdirs.norm(dim=-1, keepdim=True)
        pre = O.spherical_harmonics(c["sh_degree_to_use"], viewdirs, colors) + 0.5
        rgbs = torch.clamp(pre, 0.0, 1.0)
    else:
        pre = None
        rgbs = torch.sigmoid(colors[:, 0, :])
    opac = torch.sigmoid(c["opacities"])
    if c["opacity_mask"] is not None:
        opac = opac * c["opacity_mask"].to(opac.dtype).unsqueeze(-1)
    if c["ball_gaussians"]:
        scales = torch.exp(c["scales"]).repeat(1, 3)
    elif c["gaussian_2d"]:
        s = torch.exp(c["scales"])
        scales = torch.cat([s[..., :2], torch.zeros_like(s[..., :1])], dim=-1)
    else:
        scales = torch.exp(c["scales"])
    quats = c["quats"] / c["quats"].norm(dim=-1, keepdim=True)
    return dict(_means=c["means"], _opacities=opac, _rgbs=rgbs, _scales=scales, _quats=quats), pre


_PARAMS = ("means", "features_dc", "features_rest", "opacities", "scales", "quats")


def test_collect_matches_reference_lines(gpu_device):
    from gsplat.prepass import GaussianClass, collect_gaussians

    dev = gpu_device
    specs = [
        _make_class(5000, 16, 3, 2, 1, label=0),                 # Background, progressive degree 2 of 3
        _make_class(777, 16, 3, 3, 2, mask=True, label=1),       # RigidNodes: validity mask
        _make_class(0, 16, 3, 3, 3, label=2),                    # an empty class
        _make_class(1301, 4, 1, 1, 4, label=3),                  # SMPLNodes: K = 4
        _make_class(130, 1, 0, 0, 5, label=4),                   # sh_degree 0: sigmoid colours, no rest rows
        _make_class(640, 25, 4, 4, 6, ball=True, label=5),       # ball Gaussians, degree 4
        _make_class(333, 16, 3, 0, 7, two_d=True, label=6),      # 2-D Gaussians, degree 0 of 3 (step 0)
    ]
    c2w = torch.eye(4)
    c2w[:3, 3] = torch.tensor([1.5, -2.0, 0.7])

    # reference replay, fp64 on the host
    ref_in = [{k: (v.double().requires_grad_(True) if k in _PARAMS else v) for k, v in c.items()} for c in specs]
    ref_out, pres = zip(*[_reference_get_gaussians(c, c2w[:3, 3].double()) for c in ref_in])
    ref_cat = {k: torch.cat([o[k] for o in ref_out], dim=0) for k in ref_out[0]}
    ref_labels = torch.cat([torch.full((c["means"].shape[0],), c["label"]) for c in specs])

    gpu_in = [{k: (v.to(dev).requires_grad_(True) if k in _PARAMS else (v.to(dev) if torch.is_tensor(v) else v))
               for k, v in c.items()} for c in specs]
    gs, labels, flag = collect_gaussians([GaussianClass(**c) for c in gpu_in], c2w.to(dev))
    flag.check()  # everything finite
    assert torch.equal(labels.cpu(), ref_labels)
    for k, ref in ref_cat.items():
        got = gs[k].detach().cpu().double()
        assert got.shape == ref.shape, (k, got.shape, ref.shape)
        assert torch.allclose(got, ref.detach(), rtol=1e-5, atol=2e-6), (k, float((got - ref.detach()).abs().max()))
    assert torch.equal(gs["_means"].detach().cpu(), torch.cat([c["means"] for c in specs]))

    # gradients of a random linear functional of the five arrays
    gen = torch.Generator().manual_seed(99)
    ws = {k: torch.randn(v.shape, generator=gen) for k, v in ref_cat.items()}
    sum((ref_cat[k] * ws[k].double()).sum() for k in ws).backward()
    sum((gs[k] * ws[k].to(dev)).sum() for k in ws).backward()
    skipped = total = 0
    for ci, (cr, cg) in enumerate(zip(ref_in, gpu_in)):
        n = cr["means"].shape[0]
        near = torch.zeros(n, dtype=torch.bool)
        if pres[ci] is not None and n:
            p = pres[ci].detach()
            near = (((p - 0.0).abs() < 1e-5) | ((p - 1.0).abs() < 1e-5)).any(dim=-1)
            skipped += int(near.sum())
            total += n
        for k in _PARAMS:
            got, ref = cg[k].grad.cpu().double(), cr[k].grad
            assert got.shape == ref.shape, (ci, k)
            if k in ("features_dc", "features_rest"):
                got, ref = got[~near], ref[~near]
            assert torch.allclose(got, ref, rtol=1e-5, atol=2e-6), (ci, k, float((got - ref).abs().max()))
    assert skipped <= 1e-3 * total + 1, (skipped, total)
    # bands above the degree in use get exact zeros, and some colours did clamp (the mask matters)
    assert torch.count_nonzero(gpu_in[0]["features_rest"].grad[:, 8:, :]) == 0
    assert float((ref_cat["_rgbs"] == 0).float().mean()) > 0.01


@pytest.mark.parametrize("bad,msg", [
    (("means", float("nan")), "NaN detected in gaussian _means"),
    (("scales", 200.0), "Inf detected in gaussian _scales"),        # exp overflows
    (("features_dc", float("nan")), "NaN detected in gaussian _rgbs"),
    (("quats", 0.0), "NaN detected in gaussian _quats"),            # 0 / |0|
])
def test_finite_flag_reports_like_the_reference(gpu_device, bad, msg):
    from gsplat.prepass import GaussianClass, collect_gaussians

    dev = gpu_device
    c = _make_class(1000, 16, 3, 3, 11)
    key, val = bad
    c[key][517] = val
    c = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in c.items()}
    gs, _, flag = collect_gaussians([GaussianClass(**c)], torch.eye(4, device=dev))
    with pytest.raises(ValueError, match=msg):
        flag.check(step=7)
    flag.check()  # reported once: the word is cleared by the check that raised
    with pytest.raises(ValueError, match=msg):
        collect_gaussians([GaussianClass(**c)], torch.eye(4, device=dev), check_finite=True, step=7)


def test_prepass_feeds_rasterization(gpu_device):
    """Raw Parameters -> collect_gaussians -> rasterization (the reference's kwargs) -> backward reaches every raw
    Parameter; same image as activating with torch ops first."""
    import synth_scene
    from gsplat.cuda._wrapper import spherical_harmonics
    from gsplat.prepass import GaussianClass, collect_gaussians
    from gsplat.rendering import rasterization

    dev = gpu_device
    W, H = 160, 96
    bg = _make_class(30_000, 16, 3, 3, 21)
    bg["scales"] = bg["scales"] + 0.5
    fg = _make_class(2_000, 4, 1, 1, 22, mask=True, label=1)
    vm, Ks = (t.to(dev) for t in synth_scene.make_cameras(1, W, H))
    c2w = torch.linalg.inv(vm[0])

    def render(activate):
        cls = [{k: (v.to(dev).requires_grad_(True) if k in _PARAMS else (v.to(dev) if torch.is_tensor(v) else v))
                for k, v in c.items()} for c in (bg, fg)]
        if activate == "fused":
            gs, labels, flag = collect_gaussians([GaussianClass(**c) for c in cls], c2w)
        else:
            outs = []
            for c in cls:
                colors = torch.cat((c["features_dc"][:, None, :], c["features_rest"]), dim=1)
                viewdirs = c["means"].detach() - c2w[:3, 3]
                viewdirs = viewdirs / viewdirs.norm(dim=-1, keepdim=True)
                rgbs = torch.clamp(spherical_harmonics(c["sh_degree_to_use"], viewdirs, colors) + 0.5, 0.0, 1.0)
                opac = torch.sigmoid(c["opacities"])
                if c["opacity_mask"] is not None:
                    opac = opac * c["opacity_mask"].float().unsqueeze(-1)
                outs.append(dict(_means=c["means"], _opacities=opac, _rgbs=rgbs, _scales=torch.exp(c["scales"]),
                                 _quats=c["quats"] / c["quats"].norm(dim=-1, keepdim=True)))
            gs = {k: torch.cat([o[k] for o in outs], dim=0) for k in outs[0]}
        rc, ra, info = rasterization(means=gs["_means"], quats=gs["_quats"], scales=gs["_scales"],
                                     opacities=gs["_opacities"].squeeze(), colors=gs["_rgbs"], viewmats=vm, Ks=Ks,
                                     width=W, height=H, packed=False, absgrad=True, sparse_grad=False,
                                     rasterize_mode="classic", near_plane=0.1, far_plane=1e10, render_mode="RGB+ED")
        (rc.square().sum() + ra.sum()).backward()
        return rc.detach(), ra.detach(), [{k: c[k].grad for k in _PARAMS} for c in cls]

    rc1, ra1, g1 = render("fused")
    rc0, ra0, g0 = render("torch")
    assert torch.allclose(rc1, rc0, rtol=1e-4, atol=1e-5) and torch.allclose(ra1, ra0, rtol=1e-4, atol=1e-5)
    for a, b in zip(g1, g0):
        for k in _PARAMS:
            assert a[k] is not None and torch.isfinite(a[k]).all(), k
            rel = float((a[k].double() - b[k].double()).norm() / (b[k].double().norm() + 1e-30))
            assert rel < 1e-4, (k, rel)


def _ref_quat_to_rotmat(quat):
    """utils/geometry.py:116-152."""
    quat = quat / torch.norm(quat, dim=-1, keepdim=True)
    w, x, y, z = quat[..., 0], quat[..., 1], quat[..., 2], quat[..., 3]
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z),
                        1 - 2 * (x * x + z * z), 2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x),
                        1 - 2 * (x * x + y * y)], dim=-1).reshape(*quat.shape[:-1], 3, 3)


def _ref_quat_mult(q1, q2):
    """models/gaussians/basics.py:64-74."""
    w1, x1, y1, z1 = q1.T
    w2, x2, y2, z2 = q2.T
    return torch.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                        w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2]).T


@pytest.mark.parametrize("grouped", [True, False])
def test_rigid_transform_matches_reference_lines(gpu_device, grouped):
    """nodes/rigid.py:403-470 (transform_means / transform_quats, no test-set interpolation) in fp64 vs the kernel:
    outputs and the gradients of means, quats, instances_quats[frame], instances_trans[frame] (1e-5 relative)."""
    from gsplat.prepass import rigid_transform

    dev = gpu_device
    n, n_inst = 20_003, 37
    g = torch.Generator().manual_seed(3)
    means, quats = torch.randn(n, 3, generator=g) * 2, torch.randn(n, 4, generator=g)
    iq, it = torch.randn(n_inst, 4, generator=g) * 1.3, torch.randn(n_inst, 3, generator=g) * 10
    pid = torch.randint(0, n_inst, (n, 1), generator=g)
    if grouped:
        pid = pid.sort(dim=0).values  # points of an instance contiguous, as RigidNodes builds them
    wm_w, wq_w = torch.randn(n, 3, generator=g), torch.randn(n, 4, generator=g)

    ref = [t.double().requires_grad_(True) for t in (means, quats, iq, it)]
    quat_act = lambda x: x / x.norm(dim=-1, keepdim=True)
    rot = _ref_quat_to_rotmat(quat_act(ref[2]))[pid[..., 0]]
    wm = torch.bmm(rot, ref[0].unsqueeze(-1)).squeeze(-1) + ref[3][pid[..., 0]]
    wq = _ref_quat_mult(quat_act(ref[2][pid[..., 0]]), quat_act(ref[1]))
    ((wm * wm_w.double()).sum() + (wq * wq_w.double()).sum()).backward()

    got = [t.to(dev).requires_grad_(True) for t in (means, quats, iq, it)]
    gm, gq = rigid_transform(got[0], got[1], pid.to(dev), got[2], got[3], validate=True)
    assert torch.allclose(gm.detach().cpu().double(), wm.detach(), rtol=1e-5, atol=1e-5)
    assert torch.allclose(gq.detach().cpu().double(), wq.detach(), rtol=1e-5, atol=1e-6)
    ((gm * wm_w.to(dev)).sum() + (gq * wq_w.to(dev)).sum()).backward()
    for name, a, b in zip(("means", "quats", "inst_quats", "inst_trans"), got, ref):
        rel = float((a.grad.cpu().double() - b.grad).norm() / b.grad.norm())
        assert rel < 1e-5, (name, rel)
    # only the world means feed the loss (the other upstream gradient is absent)
    got2 = [t.to(dev).requires_grad_(True) for t in (means, quats, iq, it)]
    gm2, _ = rigid_transform(got2[0], got2[1], pid.to(dev), got2[2], got2[3])
    gm2.sum().backward()
    assert torch.allclose(got2[3].grad.cpu(), torch.bincount(pid[..., 0], minlength=n_inst).float()[:, None].expand(-1, 3))
    assert float(got2[1].grad.abs().max()) == 0.0


def test_rigid_class_chains_into_collect_and_render(gpu_device):
    """RigidNodes.get_gaussians (rigid.py:472-517) composed from the two fused ops: instance poses -> rigid_transform ->
    collect_gaussians (validity mask on the opacities) -> rasterization; the loss gradient reaches the pose rows."""
    import synth_scene
    from gsplat.prepass import GaussianClass, collect_gaussians, rigid_transform
    from gsplat.rendering import rasterization

    dev = gpu_device
    W, H, n, n_inst = 128, 80, 4000, 5
    g = torch.Generator().manual_seed(5)
    raw = _make_class(n, 16, 3, 3, 31, mask=True, label=1)
    raw["means"] = torch.randn(n, 3, generator=g) * 0.5            # object-frame points
    raw["scales"] = raw["scales"] + 0.3
    raw = {k: (v.to(dev).requires_grad_(True) if k in _PARAMS else (v.to(dev) if torch.is_tensor(v) else v))
           for k, v in raw.items()}
    pid = torch.randint(0, n_inst, (n, 1), generator=g).sort(dim=0).values.to(dev)
    inst_q = torch.randn(n_inst, 4, generator=g).to(dev).requires_grad_(True)
    inst_t = (torch.tensor([[0.0, 0.0, 8.0]]) + torch.randn(n_inst, 3, generator=g) * 2.0).to(dev).requires_grad_(True)
    vm = torch.eye(4, device=dev)[None]
    Ks = torch.tensor([[W / 2.0, 0, W / 2.0], [0, W / 2.0, H / 2.0], [0, 0, 1.0]], device=dev)[None]
    wm, wq = rigid_transform(raw["means"], raw["quats"], pid, inst_q, inst_t)
    cls = GaussianClass(**{**raw, "means": wm, "quats": wq})
    gs, labels, flag = collect_gaussians([cls], torch.eye(4, device=dev))
    rc, ra, info = rasterization(means=gs["_means"], quats=gs["_quats"], scales=gs["_scales"],
                                 opacities=gs["_opacities"].squeeze(), colors=gs["_rgbs"], viewmats=vm, Ks=Ks, width=W,
                                 height=H, packed=False, absgrad=True, near_plane=0.1, render_mode="RGB+ED")
    flag.check()
    assert float(ra.detach().sum()) > 10.0 and int(labels[0]) == 1
    (rc.square().sum() + ra.sum()).backward()
    for name, t in (("inst_quats", inst_q), ("inst_trans", inst_t), ("means", raw["means"]), ("quats", raw["quats"]),
                    ("features_rest", raw["features_rest"])):
        assert t.grad is not None and torch.isfinite(t.grad).all() and float(t.grad.abs().sum()) > 0, name


def test_rigid_transform_matches_reference_fixture(gpu_device):
    """gsplat.prepass.rigid_transform against vectors made with the REFERENCE's own helpers (utils/geometry.py
    quat_to_rotmat and safe_quaternion_multiply, tests/golden/make_golden.py::rigid_golden) on rigid.py:426-470's lines:
    world means / quaternions and the gradients of all four inputs."""
    import os

    import numpy as np

    from gsplat.prepass import rigid_transform

    dev = gpu_device
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "rigid_ref.npz"))
    f = lambda k: torch.from_numpy(z[k]).float().to(dev)
    got = [f(k).requires_grad_(True) for k in ("means", "quats", "inst_quats", "inst_trans")]
    gm, gq = rigid_transform(got[0], got[1], torch.from_numpy(z["point_ids"]).to(dev), got[2], got[3], validate=True)
    assert torch.allclose(gm.detach().cpu().double(), torch.from_numpy(z["world_means"]), rtol=1e-5, atol=1e-5)
    assert torch.allclose(gq.detach().cpu().double(), torch.from_numpy(z["world_quats"]), rtol=1e-5, atol=1e-6)
    ((gm * f("wm_w")).sum() + (gq * f("wq_w")).sum()).backward()
    for name, a in zip(("means", "quats", "inst_quats", "inst_trans"), got):
        ref = torch.from_numpy(z["grad_" + name])
        rel = float((a.grad.cpu().double() - ref).norm() / ref.norm())
        assert rel < 1e-5, (name, rel)
