"""How much would per-8x8-quadrant splat lists shorten the rasterizer's walk?  (CPU estimate, no early termination)"""
import sys, os
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle, synth_scene
from oracle import gsplat_ref as R

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
cam = int(sys.argv[2]) if len(sys.argv) > 2 else 0
W, H = 1242, 375
g = synth_scene.make_gaussians(N, sh_degree=None)
vm, Ks = synth_scene.make_cameras(5, W, H)
vm, Ks = vm[cam:cam + 1], Ks[cam:cam + 1]
radii, m2d, depths, conics, comp = R.fully_fused_projection(g["means"], g["quats"], g["scales"], vm, Ks, W, H,
                                                            near_plane=0.1, calc_compensations=True)
vis = (radii[0] > 0).numpy()
m = m2d[0].numpy()[vis]; r = radii[0].numpy()[vis].astype(np.float32); con = conics[0].numpy()[vis]
op = (g["opacities"].numpy() * comp[0].numpy())[vis]
print("visible", vis.sum())
TW, TH = (W + 15) // 16, (H + 15) // 16
tx, ty, tr = m[:, 0] / 16, m[:, 1] / 16, r / 16
x0 = np.clip(np.floor(tx - tr), 0, TW).astype(int); x1 = np.clip(np.ceil(tx + tr), 0, TW).astype(int)
y0 = np.clip(np.floor(ty - tr), 0, TH).astype(int); y1 = np.clip(np.ceil(ty + tr), 0, TH).astype(int)
cnt = (x1 - x0) * (y1 - y0)
gid = np.repeat(np.arange(len(cnt)), cnt)
off = np.concatenate([[0], np.cumsum(cnt)])[:-1]
loc = np.arange(cnt.sum()) - np.repeat(off, cnt)
bw = np.repeat(x1 - x0, cnt)
tix = np.repeat(x0, cnt) + loc % bw
tiy = np.repeat(y0, cnt) + loc // bw
print("isects", len(gid))
LOG2E = 1.4426950408889634
A = -0.5 * LOG2E * con[gid, 0]; Bh = -0.5 * LOG2E * con[gid, 1]; Cc = -0.5 * LOG2E * con[gid, 2]
mx, my, o = m[gid, 0], m[gid, 1], op[gid]


def touch(px0, py0, ext):
    dxh = mx - px0; dxl = dxh - ext
    dyh = my - py0; dyl = dyh - ext
    cx = np.minimum(np.maximum(0, dxl), dxh); cy = np.minimum(np.maximum(0, dyl), dyh)
    ys = np.minimum(np.maximum(-Bh * cx / Cc, dyl), dyh)
    xs = np.minimum(np.maximum(-Bh * cy / A, dxl), dxh)
    pX = A * cx * cx + 2 * Bh * cx * ys + Cc * ys * ys
    pY = A * xs * xs + 2 * Bh * xs * cy + Cc * cy * cy
    pmax = np.minimum(0, np.maximum(pX, pY))
    return o * np.exp2(pmax) >= 1 / 255


bx, by = tix * 16 + 0.5, tiy * 16 + 0.5
keep = touch(bx, by, 15)
print("tile keep frac", keep.mean())
tile_id = tiy * TW + tix
ntile = TW * TH
tile_cnt = np.bincount(tile_id[keep], minlength=ntile)
for sub, name in ((8, "8x8 quadrants (4)"), (4, "4x4 blocks (16)")):
    nb = 16 // sub
    qc = np.zeros((nb * nb, ntile), np.int64)
    tot = 0
    for qy in range(nb):
        for qx in range(nb):
            kq = keep & touch(bx + sub * qx, by + sub * qy, sub - 1)
            qc[qy * nb + qx] = np.bincount(tile_id[kq], minlength=ntile)
            tot += kq.sum()
    print(f"{name}: mean blocks per kept (tile,splat) = {tot / keep.sum():.3f} of {nb*nb};  "
          f"sum_tiles max_q / sum_tiles cnt = {qc.max(0).sum() / tile_cnt.sum():.3f};  "
          f"sum_tiles mean_q / cnt = {qc.mean(0).sum() / tile_cnt.sum():.3f}")

# per-batch imbalance: tile lists in depth order, staged in chunks of B raw entries; iterations = sum over chunks of max_q
dep = depths[0].numpy()[vis][gid]
order = np.lexsort((dep, tile_id))
t_s = tile_id[order]
starts = np.searchsorted(t_s, np.arange(ntile)); pos = np.arange(len(t_s)) - starts[t_s]
masks = []
for qy in range(2):
    for qx in range(2):
        masks.append((keep & touch(bx + 8 * qx, by + 8 * qy, 7))[order])
masks = np.stack(masks)  # [4, I]
keep_s = keep[order]
for Bsz in (64, 128, 256):
    chunk = t_s.astype(np.int64) * 100000 + pos // Bsz
    u, inv = np.unique(chunk, return_inverse=True)
    per = np.stack([np.bincount(inv, weights=masks[q], minlength=len(u)) for q in range(4)])
    tile_per = np.bincount(inv, weights=keep_s, minlength=len(u))
    print(f"batch {Bsz}: iterations quad-queues / tile-queue = {per.max(0).sum() / tile_per.sum():.3f}  "
          f"(chunks {len(u)}, mean kept per chunk {tile_per.mean():.1f})")
