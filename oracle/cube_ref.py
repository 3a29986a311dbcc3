"""CPU restatement of the seamless bilinear cube-map lookup -- TEST INFRASTRUCTURE ONLY (imported by tests/ alone).

The reference's EnvLight (/root/reference/3D-Reconstruction-main/models/modules.py:191-203) calls
`nvdiffrast.torch.texture(base[None], l, filter_mode='linear', boundary_mode='cube')`.  nvdiffrast is not under
/root/reference and not installed here: **parity unpinned**.  This file restates the documented behaviour (OpenGL cube
map face order and orientation, bilinear filtering on texel centres, taps beyond a face edge taken from the adjacent
face) in numpy fp64, one direction at a time, with the face table written out from the OpenGL specification's
(major axis -> sc, tc) table; the GPU tests additionally pin the convention and the seamlessness by properties that do
not depend on this file."""
import numpy as np

# face -> (major axis, its sign, sc axis, sc sign, tc axis, tc sign)   [OpenGL 4.6 core, table 8.19]
FACES = [(0, +1, 2, -1, 1, -1), (0, -1, 2, +1, 1, -1), (1, +1, 0, +1, 2, +1),
         (1, -1, 0, +1, 2, -1), (2, +1, 0, +1, 1, -1), (2, -1, 0, -1, 1, -1)]


def face_of(p):
    a = np.abs(p)
    if a[0] >= a[1] and a[0] >= a[2]:
        return 0 if p[0] >= 0 else 1
    if a[1] >= a[2]:
        return 2 if p[1] >= 0 else 3
    return 4 if p[2] >= 0 else 5


def texel(f, ix, iy, R):
    """(face, iy, ix) of a tap that may lie one texel outside face f."""
    if 0 <= ix < R and 0 <= iy < R:
        return f, iy, ix
    if not 0 <= ix < R:
        iy = min(max(iy, 0), R - 1)
    ma_a, ma_s, s_a, s_s, t_a, t_s = FACES[f]
    sc, tc, ma = (ix + 0.5) * 2.0 / R - 1.0, (iy + 0.5) * 2.0 / R - 1.0, 1.0
    if abs(sc) > 1.0:
        ma, sc = ma - (abs(sc) - 1.0), np.sign(sc)
    else:
        ma, tc = ma - (abs(tc) - 1.0), np.sign(tc)
    p = np.zeros(3)
    p[ma_a], p[s_a], p[t_a] = ma_s * ma, s_s * sc, t_s * tc
    g = face_of(p)
    _, _, gs_a, gs_s, gt_a, gt_s = FACES[g]
    jx = int(np.clip(np.floor((gs_s * p[gs_a] + 1.0) * 0.5 * R), 0, R - 1))
    jy = int(np.clip(np.floor((gt_s * p[gt_a] + 1.0) * 0.5 * R), 0, R - 1))
    return g, jy, jx


def cube_texture(tex, dirs):
    """tex [6,R,R,C] float64, dirs [n,3] -> [n,C]."""
    R = tex.shape[1]
    out = np.zeros((dirs.shape[0], tex.shape[3]))
    for i, d in enumerate(dirs):
        f = face_of(d)
        ma_a, _, s_a, s_s, t_a, t_s = FACES[f]
        m = abs(d[ma_a])
        u = (s_s * d[s_a] / m + 1.0) * 0.5 * R - 0.5
        v = (t_s * d[t_a] / m + 1.0) * 0.5 * R - 0.5
        ix, iy = int(np.floor(u)), int(np.floor(v))
        ax, ay = u - ix, v - iy
        for (dx, dy, w) in ((0, 0, (1 - ax) * (1 - ay)), (1, 0, ax * (1 - ay)), (0, 1, (1 - ax) * ay), (1, 1, ax * ay)):
            g, jy, jx = texel(f, ix + dx, iy + dy, R)
            out[i] += w * tex[g, jy, jx]
    return out
