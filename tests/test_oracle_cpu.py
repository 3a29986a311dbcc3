"""CPU tests of the oracle: golden vectors, analytic known-answer tests, integer-stage cross-checks.
(The reference ships no tests or vectors for this path -- SURVEY.md section 4 -- so besides the one in-tree anchor,
quat_to_rotmat, these KATs are closed-form.)"""
import math
import os

import numpy as np
import torch

import oracle
from oracle import gsplat_ref as R

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_quat_to_rotmat_matches_reference_vectors():
    # vectors produced by the reference's own utils/geometry.py:116-152 (tests/golden/make_golden.py)
    z = np.load(os.path.join(GOLD, "quat_to_rotmat_ref.npz"))
    q, ref = torch.from_numpy(z["quats"]), torch.from_numpy(z["rotmats"])
    assert torch.allclose(oracle.quat_to_rotmat(q), ref, atol=1e-14)
    from gsplat.cuda_legacy._torch_impl import quat_to_rotmat as host_q2r
    assert torch.allclose(host_q2r(q), ref, atol=1e-14)
    assert torch.allclose(host_q2r(q.float()).double(), ref, atol=1e-6)
    # known answers stated in SURVEY.md section 8(c)
    assert torch.allclose(ref[1], torch.eye(3, dtype=torch.float64))
    assert torch.allclose(ref[2], torch.tensor([[0., 0, 1], [1, 0, 0], [0, 1, 0]], dtype=torch.float64))


def test_num_sh_bases():
    from gsplat.cuda_legacy._wrapper import num_sh_bases
    assert [num_sh_bases(d) for d in range(5)] == [1, 4, 9, 16, 25]
    assert [oracle.num_sh_bases(d) for d in range(5)] == [1, 4, 9, 16, 25]


def test_sh_band0_and_orthonormality():
    C0 = 0.28209479177387814  # R/models/gaussians/basics.py:80
    c = torch.randn(10, 16, 3, dtype=torch.float64)
    d = torch.randn(10, 3, dtype=torch.float64)
    assert torch.allclose(oracle.spherical_harmonics(0, d, c), C0 * c[:, 0, :])
    # real SH are orthonormal on the sphere: Monte-Carlo Gram matrix ~ identity / (4 pi) * 4 pi
    g = torch.Generator().manual_seed(0)
    dirs = torch.randn(400_000, 3, generator=g, dtype=torch.float64)
    Bm = R._sh_bases(4, dirs)
    gram = (Bm.T @ Bm) / dirs.shape[0] * 4 * math.pi
    assert torch.allclose(gram, torch.eye(25, dtype=torch.float64), atol=2e-2)
    # golden SH fixture reproduces
    z = np.load(os.path.join(GOLD, "sh_fp64.npz"))
    for deg in range(5):
        o = oracle.spherical_harmonics(deg, torch.from_numpy(z["dirs"]), torch.from_numpy(z["coeffs"]))
        assert np.allclose(o.numpy(), z[f"out{deg}"], atol=1e-13)


def _one_cam(W, H, f):
    vm = torch.eye(4, dtype=torch.float64)[None]
    K = torch.tensor([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1.0]], dtype=torch.float64)[None]
    return vm, K


def test_single_isotropic_gaussian_closed_form():
    W, H, f, s, z, o = 64, 48, 50.0, 0.2, 4.0, 0.8
    vm, K = _one_cam(W, H, f)
    # on the optical axis, centre lands exactly on the pixel-corner (32, 24)
    means = torch.tensor([[0.0, 0.0, z]], dtype=torch.float64)
    quats = torch.tensor([[1.0, 0, 0, 0]], dtype=torch.float64)
    scales = torch.full((1, 3), s, dtype=torch.float64)
    radii, m2d, depths, conics, comps = oracle.fully_fused_projection(means, quats, scales, vm, K, W, H,
                                                                      calc_compensations=True)
    var = (f * s / z) ** 2
    assert torch.allclose(m2d[0, 0], torch.tensor([32.0, 24.0], dtype=torch.float64))
    assert abs(float(depths[0, 0]) - z) < 1e-12
    assert torch.allclose(conics[0, 0], torch.tensor([1 / (var + 0.3), 0.0, 1 / (var + 0.3)], dtype=torch.float64))
    assert int(radii[0, 0]) == math.ceil(3 * math.sqrt(var + 0.3 + math.sqrt(0.01)))  # max(0.01, b^2-det) branch
    assert abs(float(comps[0, 0]) - var / (var + 0.3)) < 1e-12
    col = torch.tensor([[0.3, 0.6, 0.9]], dtype=torch.float64)
    rc, ra, meta = oracle.rasterization(means, quats, scales, torch.tensor([o], dtype=torch.float64), col, vm, K, W, H)
    # pixel (32,24) has centre (32.5, 24.5): delta = (-0.5,-0.5)
    sigma = 0.5 * (0.25 + 0.25) / (var + 0.3)
    a = o * math.exp(-sigma)
    assert abs(float(ra[0, 24, 32, 0]) - a) < 1e-12
    assert torch.allclose(rc[0, 24, 32], col[0] * a)
    assert float(ra[0, 0, 0, 0]) == 0.0


def test_two_stacked_gaussians_compositing_order_and_background():
    W, H, f = 32, 32, 40.0
    vm, K = _one_cam(W, H, f)
    means = torch.tensor([[0.0, 0, 5.0], [0.0, 0, 3.0]], dtype=torch.float64)  # second one is in front
    quats = torch.tensor([[1.0, 0, 0, 0]] * 2, dtype=torch.float64)
    scales = torch.full((2, 3), 0.5, dtype=torch.float64)
    op = torch.tensor([0.9, 0.5], dtype=torch.float64)
    col = torch.tensor([[1.0, 0, 0], [0, 1.0, 0]], dtype=torch.float64)
    bg = torch.tensor([[0.2, 0.3, 0.4]], dtype=torch.float64)
    rc, ra, meta = oracle.rasterization(means, quats, scales, op, col, vm, K, W, H, backgrounds=bg, render_mode="RGB+ED")
    _, m2d, depths, conics, _ = oracle.fully_fused_projection(means, quats, scales, vm, K, W, H)
    px = torch.tensor([16.5, 16.5], dtype=torch.float64)

    def alpha(i):
        d = m2d[0, i] - px
        s = 0.5 * (conics[0, i, 0] * d[0] ** 2 + conics[0, i, 2] * d[1] ** 2) + conics[0, i, 1] * d[0] * d[1]
        return min(0.999, float(op[i] * torch.exp(-s)))

    a_back, a_front = alpha(0), alpha(1)
    T = (1 - a_front) * (1 - a_back)
    exp_rgb = col[1] * a_front + col[0] * a_back * (1 - a_front) + T * bg[0]
    assert torch.allclose(rc[0, 16, 16, :3], exp_rgb, atol=1e-12)
    assert abs(float(ra[0, 16, 16, 0]) - (1 - T)) < 1e-12
    exp_depth = (3.0 * a_front + 5.0 * a_back * (1 - a_front)) / (1 - T)
    assert abs(float(rc[0, 16, 16, 3]) - exp_depth) < 1e-12
    # sorted front-to-back: flatten id 1 (depth 3) precedes id 0 in every shared tile
    ids = meta["isect_ids"].numpy()
    assert np.all(np.diff(ids) >= 0)
    assert int(meta["last_ids"][0, 16, 16]) == int(meta["isect_offsets"][0, 1, 1]) + 1


def test_early_termination_and_alpha_threshold():
    W, H, f = 16, 16, 30.0
    vm, K = _one_cam(W, H, f)
    n = 12
    means = torch.stack([torch.zeros(n), torch.zeros(n), torch.linspace(2, 4, n)], -1).double()
    quats = torch.tensor([[1.0, 0, 0, 0]] * n, dtype=torch.float64)
    scales = torch.full((n, 3), 1.0, dtype=torch.float64)
    op = torch.full((n,), 0.999, dtype=torch.float64)
    col = torch.rand(n, 3, dtype=torch.float64)
    rc, ra, meta = oracle.rasterization(means, quats, scales, op, col, vm, K, W, H)
    # independent scalar restatement of Appendix A.5 (the per-pixel loop form) for a few pixels
    m2d, con, opa = meta["means2d"][0], meta["conics"][0], meta["opacities"][0]
    fids, start = meta["flatten_ids"].tolist(), int(meta["isect_offsets"][0, 0, 0])
    stopped_early = 0
    for (i, j) in [(0, 0), (7, 9), (15, 15), (3, 12)]:
        T, out, cur = 1.0, torch.zeros(3, dtype=torch.float64), 0
        for k in range(start, len(fids)):
            g = fids[k]
            dx, dy = float(m2d[g, 0]) - (j + 0.5), float(m2d[g, 1]) - (i + 0.5)
            sigma = 0.5 * (float(con[g, 0]) * dx * dx + float(con[g, 2]) * dy * dy) + float(con[g, 1]) * dx * dy
            a = min(0.999, float(opa[g]) * math.exp(-sigma))
            if sigma < 0 or a < 1.0 / 255.0:
                continue
            nT = T * (1 - a)
            if nT <= 1e-4:
                stopped_early += 1
                break
            out += col[g] * a * T
            cur, T = k, nT
        assert abs(float(ra[0, i, j, 0]) - (1 - T)) < 1e-12
        assert torch.allclose(rc[0, i, j], out, atol=1e-12)
        assert int(meta["last_ids"][0, i, j]) == cur
    assert stopped_early == 4 and int(meta["last_ids"].max()) < len(fids) - 1  # the T <= 1e-4 stop was exercised
    # transparent Gaussians (alpha < 1/255) are skipped entirely
    rc2, ra2, _ = oracle.rasterization(means, quats, scales, torch.full((n,), 1.0 / 300, dtype=torch.float64), col, vm, K, W, H)
    assert float(ra2.abs().max()) == 0.0


def test_integer_stages_against_numpy():
    z = np.load(os.path.join(GOLD, "pipeline_fp64.npz"))
    ids, fids = z["meta_isect_ids"], z["meta_flatten_ids"]
    TW, TH = math.ceil(int(z["width"]) / 16), math.ceil(int(z["height"]) / 16)
    C = z["in_viewmats"].shape[0]
    assert np.all(np.diff(ids) >= 0)
    same = np.diff(ids) == 0
    assert np.all(np.diff(fids.astype(np.int64))[same] > 0)  # stable: ties keep ascending flatten id
    tile_n_bits = int(math.floor(math.log2(TW * TH))) + 1
    lin = ((ids >> 32) >> tile_n_bits) * (TW * TH) + ((ids >> 32) & ((1 << tile_n_bits) - 1))
    off = np.searchsorted(lin, np.arange(C * TW * TH), side="left")
    assert np.array_equal(off.reshape(C, TH, TW), z["meta_isect_offsets"])
    assert int(z["meta_tiles_per_gauss"].sum()) == ids.shape[0]
    assert np.array_equal(z["meta_tiles_per_gauss"] > 0, z["meta_radii"] > 0)


def test_pipeline_fixture_reproduces_and_fp32_is_close():
    z = np.load(os.path.join(GOLD, "pipeline_fp64.npz"))
    W, H = int(z["width"]), int(z["height"])
    for dt, tol in [(torch.float64, 1e-12), (torch.float32, 2e-4)]:
        a = {k: torch.from_numpy(z[f"in_{k}"]).to(dt) for k in ["means", "quats", "scales", "opacities", "colors"]}
        rc, ra, meta = oracle.rasterization(a["means"], a["quats"], a["scales"], a["opacities"], a["colors"],
                                            torch.from_numpy(z["in_viewmats"]).to(dt), torch.from_numpy(z["in_Ks"]).to(dt),
                                            W, H, near_plane=0.1, sh_degree=3, render_mode="RGB+ED",
                                            rasterize_mode="antialiased")
        bad = (np.abs(rc.double().numpy() - z["render_colors"]) > tol * (1 + np.abs(z["render_colors"]))).mean()
        assert bad <= (0.0 if dt == torch.float64 else 5e-3), (dt, bad)
        assert np.abs(ra.double().numpy() - z["render_alphas"]).max() <= (1e-12 if dt == torch.float64 else 2e-2)


def test_projection_gradcheck_fp64():
    torch.manual_seed(0)
    N, W, H = 6, 40, 30
    means = (torch.randn(N, 3, dtype=torch.float64) * 0.5 + torch.tensor([0, 0, 4.0], dtype=torch.float64)).requires_grad_(True)
    quats = torch.randn(N, 4, dtype=torch.float64, requires_grad=True)
    scales = (torch.rand(N, 3, dtype=torch.float64) * 0.3 + 0.1).requires_grad_(True)
    vm, K = _one_cam(W, H, 30.0)

    def f(m, q, s):
        radii, m2d, d, con, comp = oracle.fully_fused_projection(m, q, s, vm, K, W, H, calc_compensations=True)
        assert bool((radii > 0).all())
        return m2d, d, con

    assert torch.autograd.gradcheck(f, (means, quats, scales), eps=1e-6, atol=1e-6)


# ------------------------------------------------------------------ the loss oracle against the reference's own file
def _loss_fixture():
    z = np.load(os.path.join(GOLD, "losses_ref.npz"))
    t = lambda k: torch.from_numpy(z[k]).double()
    return z, t


def test_loss_oracle_safe_bce_matches_reference_fixture():
    """oracle/loss_ref.py::_SafeBCE against outputs of the reference's SafeBCE (models/losses.py:33-75): forward map and
    input gradient, inputs outside [0, 1], limit 0.1, the x == y gate."""
    from oracle import loss_ref as L

    z, t = _loss_fixture()
    x = t("opacity_wild")[..., 0].requires_grad_(True)
    loss = L._SafeBCE.apply(x, t("safe_y"), float(z["limit"]))
    loss.backward(t("safe_go"))
    assert torch.allclose(loss.detach(), t("safe_loss"), rtol=1e-13, atol=1e-14)
    assert torch.allclose(x.grad, t("safe_grad_x"), rtol=1e-13, atol=1e-14)
    gate = (x.detach().clip(0, 1) == t("safe_y"))
    assert int(gate.sum()) > 0 and float(x.grad[gate].abs().max()) == 0.0


def test_loss_oracle_mask_terms_match_reference_fixture():
    """mask_loss (base.py:536-550) with binary_cross_entropy / safe_binary_cross_entropy (losses.py:77-83)."""
    from oracle import loss_ref as L

    z, t = _loss_fixture()
    for name, op_key, kind in (("bce", "opacity", "bce"), ("safe_bce", "opacity", "safe_bce"),
                               ("safe_bce_wild", "opacity_wild", "safe_bce")):
        o = t(op_key).requires_grad_(True)
        val = L.mask_loss(o, t("sky"), t("valid"), kind)
        val.backward()
        assert abs(float(val.detach()) - float(z[f"mask_{name}"])) <= 1e-13, name
        assert torch.allclose(o.grad, t(f"mask_{name}_grad"), rtol=1e-12, atol=1e-15), name


def test_loss_oracle_depth_loss_matches_reference_fixture():
    """depth_loss against the reference's DepthLoss (losses.py:91-176) in all 48 type x normalize x inverse x reduction
    settings, two percentiles, and the empty hit set."""
    from oracle import loss_ref as L
    from util import DEPTH_CASES, PERCENTILE_CASES, depth_case_key

    z, t = _loss_fixture()
    md = float(z["max_depth"])
    cases = [c + (None,) for c in DEPTH_CASES] + PERCENTILE_CASES
    for (ty, n, i, r, pc) in cases:
        p = t("depth").requires_grad_(True)
        val = L.depth_loss(p, t("lidar"), t("valid"), ty, n, i, pc, md, r)
        k = depth_case_key(ty, n, i, r, pc)
        ref = torch.from_numpy(np.asarray(z[k])).double()
        assert val.shape == ref.shape, k
        assert torch.allclose(val.detach(), ref, rtol=1e-12, atol=1e-14), k
        if r != "none":
            val.backward()
            assert torch.allclose(p.grad, t(k + "_grad"), rtol=1e-12, atol=1e-16), k
    for r in ("mean_on_hit", "mean_on_hw", "sum"):
        val = L.depth_loss(t("depth"), torch.zeros_like(t("lidar")), t("valid"), "l1", False, False, None, md, r)
        ref = float(z[f"depth_empty_{r}"])
        assert (np.isnan(ref) and bool(torch.isnan(val))) or float(val) == ref, r
