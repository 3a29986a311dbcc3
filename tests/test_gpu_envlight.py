"""gsplat.envlight.cube_texture (gs_cube_texture_fwd / _bwd): the nvdiffrast cube-map lookup of the reference's EnvLight
(/root/reference/3D-Reconstruction-main/models/modules.py:191-203).  nvdiffrast is not available (parity unpinned), so
besides the numpy restatement (oracle/cube_ref.py) the tests pin the behaviour by properties: the OpenGL face /
orientation table, exact texel centres, seamless edges (a smooth function of direction stored in the map is reproduced
across edges and corners), partition of unity, and the backward as the exact transpose of the (linear) forward."""
import numpy as np
import pytest
import torch

from oracle import cube_ref as O

pytestmark = pytest.mark.gpu


def _dirs(n, seed):
    g = torch.Generator().manual_seed(seed)
    d = torch.randn(n, 3, generator=g)
    return d * (0.2 + 3.0 * torch.rand(n, 1, generator=g))  # any length


def test_matches_numpy_restatement(gpu_device):
    from gsplat.envlight import cube_texture

    dev = gpu_device
    R = 8
    g = torch.Generator().manual_seed(0)
    tex = torch.rand(6, R, R, 3, generator=g)
    d = _dirs(3000, 1)
    # add directions hugging edges and corners
    e = torch.tensor([[1.0, 0.999, 0.3], [1.0, -0.2, 1.0001], [-1.0, 0.98, -0.99], [0.99, 1.0, 0.97], [0.3, -1.0, 0.995],
                      [1.0, 1.0, 1.0], [-1.0, 1.0, -0.999], [0.5, 0.2, -1.0]])
    d = torch.cat([d, e, -e])
    got = cube_texture(tex.to(dev), d.to(dev)).cpu().double().numpy()
    ref = O.cube_texture(tex.double().numpy(), d.double().numpy())
    assert np.abs(got - ref).max() < 2e-5, np.abs(got - ref).max()


def test_opengl_face_table_and_texel_centres(gpu_device):
    from gsplat.envlight import cube_texture

    dev = gpu_device
    R = 16
    f_id = torch.arange(6).float()[:, None, None].expand(6, R, R)
    iy = torch.arange(R).float()[None, :, None].expand(6, R, R)
    ix = torch.arange(R).float()[None, None, :].expand(6, R, R)
    tex = torch.stack([f_id, iy, ix], dim=-1).contiguous().to(dev)  # texel value = (face, row, column)
    # (direction, face, expected s, expected t) from the OpenGL table: sc / tc per major axis
    cases = [((1, 0.25, -0.5), 0, (0.5 + 1) / 2, (-0.25 + 1) / 2),   # +x: sc = -rz, tc = -ry
             ((-2, 0.5, 1.0), 1, (0.5 + 1) / 2, (-0.25 + 1) / 2),    # -x: sc = +rz, tc = -ry
             ((0.5, 1, 0.25), 2, (0.5 + 1) / 2, (0.25 + 1) / 2),     # +y: sc = +rx, tc = +rz
             ((0.5, -1, 0.25), 3, (0.5 + 1) / 2, (-0.25 + 1) / 2),   # -y: sc = +rx, tc = -rz
             ((0.5, 0.25, 1), 4, (0.5 + 1) / 2, (-0.25 + 1) / 2),    # +z: sc = +rx, tc = -ry
             ((0.5, 0.25, -1), 5, (-0.5 + 1) / 2, (-0.25 + 1) / 2)]  # -z: sc = -rx, tc = -ry
    d = torch.tensor([c[0] for c in cases], dtype=torch.float32, device=dev)
    out = cube_texture(tex, d).cpu()
    for k, (_, f, s, t) in enumerate(cases):
        assert abs(float(out[k, 0]) - f) < 1e-5, (k, out[k])
        assert abs(float(out[k, 2]) - (s * R - 0.5)) < 1e-4 and abs(float(out[k, 1]) - (t * R - 0.5)) < 1e-4, (k, out[k])
    # the direction through a texel centre returns exactly that texel
    rnd = torch.rand(6, R, R, 3, generator=torch.Generator().manual_seed(3)).to(dev)
    for f in range(6):
        ma_a, ma_s, s_a, s_s, t_a, t_s = O.FACES[f]
        jj, ii = torch.meshgrid(torch.arange(R), torch.arange(R), indexing="ij")
        sc, tc = (ii + 0.5) * 2.0 / R - 1.0, (jj + 0.5) * 2.0 / R - 1.0
        p = torch.zeros(R, R, 3)
        p[..., ma_a], p[..., s_a], p[..., t_a] = float(ma_s), s_s * sc, t_s * tc
        got = cube_texture(rnd, (p * 1.7).to(dev))
        assert torch.allclose(got, rnd[f], atol=2e-5), f


def test_seamless_and_partition_of_unity(gpu_device):
    from gsplat.envlight import cube_texture

    dev = gpu_device
    R = 64
    # store a smooth function of the unit direction at every texel centre
    def fn(u):
        return torch.stack([u[..., 0], u[..., 1] * u[..., 2], 0.5 + 0.3 * u[..., 2] - 0.2 * u[..., 0] * u[..., 1]], dim=-1)

    tex = torch.zeros(6, R, R, 3)
    for f in range(6):
        ma_a, ma_s, s_a, s_s, t_a, t_s = O.FACES[f]
        jj, ii = torch.meshgrid(torch.arange(R), torch.arange(R), indexing="ij")
        p = torch.zeros(R, R, 3)
        p[..., ma_a], p[..., s_a], p[..., t_a] = float(ma_s), s_s * ((ii + 0.5) * 2.0 / R - 1.0), t_s * ((jj + 0.5) * 2.0 / R - 1.0)
        tex[f] = fn(p / p.norm(dim=-1, keepdim=True))
    d = _dirs(300_000, 5)
    # plus directions within a texel of every edge and corner
    g = torch.Generator().manual_seed(6)
    e = torch.sign(torch.randn(100_000, 3, generator=g))
    e[:, 2] = torch.rand(100_000, generator=g) * 2 - 1
    e = e[:, torch.randperm(3, generator=g)] * (1 + 0.02 * torch.randn(100_000, 3, generator=g))
    d = torch.cat([d, e])
    got = cube_texture(tex.to(dev), d.to(dev)).cpu()
    ref = fn(d / d.norm(dim=-1, keepdim=True))
    err = (got - ref).abs().max(dim=-1).values
    assert float(err.max()) < 4e-3, float(err.max())  # a wrong fold shows up as O(0.1 .. 1) next to the edges
    const = torch.full((6, R, R, 3), 0.5, device=dev)  # EnvLight's initial map (modules.py:187-189)
    assert float((cube_texture(const, d.to(dev)) - 0.5).abs().max()) < 1e-6


def test_backward_is_the_transpose(gpu_device):
    from gsplat.envlight import EnvLight, cube_texture

    dev = gpu_device
    R = 32
    g = torch.Generator().manual_seed(9)
    tex = torch.randn(6, R, R, 3, generator=g).to(dev).requires_grad_(True)
    d = _dirs(50_000, 10).to(dev)
    v = torch.randn(50_000, 3, generator=g).to(dev)
    out = cube_texture(tex, d)
    (out * v).sum().backward()
    x = torch.randn(6, R, R, 3, generator=g).to(dev)
    lhs = float((cube_texture(x, d).double() * v.double()).sum())   # <A x, v>
    rhs = float((x.double() * tex.grad.double()).sum())              # <x, A^T v>
    assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs)), (lhs, rhs)
    # the module: same Parameter name / shape as the reference's, rotation applied, output shaped like the directions
    env = EnvLight("Sky", resolution=16, device=dev)
    assert tuple(env.state_dict()["base"].shape) == (6, 16, 16, 3)
    with torch.no_grad():
        env.base.copy_(torch.rand(6, 16, 16, 3, generator=g))
    vd = _dirs(7 * 9, 11).reshape(7, 9, 3).to(dev)
    o = env({"viewdirs": vd})
    assert o.shape == (7, 9, 3)
    assert torch.allclose(o, cube_texture(env.base, (vd.reshape(-1, 3) @ env.to_opengl.T).reshape(7, 9, 3)), atol=1e-6)
    o.sum().backward()
    assert env.base.grad is not None and float(env.base.grad.sum()) == pytest.approx(7 * 9 * 3, rel=1e-4)
