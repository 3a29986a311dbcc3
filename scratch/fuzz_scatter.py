"""Random shapes / launch shapes: the LDS-staged region scatter against the direct one (sorted outputs must be identical)."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "3d-reconstruction_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
from gsplat.cuda import _wrapper as W
from gsplat import _backend as B
dev = torch.device("cuda:0")
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for case in range(n_cases):
    rnd = random.Random(seed0 * 7919 + case)
    rng = np.random.default_rng(seed0 * 7919 + case)
    C = rnd.choice([1, 1, 2, 5])
    N = rnd.choice([1, 300, 5000, 40_000, 200_000])
    TW, TH = rnd.choice([(1, 1), (5, 4), (78, 24), (120, 80), (128, 128), (200, 3), (3, 200)])
    box = rnd.choice([2, 4, 9, 20])
    vis_p = rnd.choice([0.05, 0.3, 0.9])
    x0 = rng.integers(0, TW, size=(C, N)); y0 = rng.integers(0, TH, size=(C, N))
    w = rng.integers(1, box + 1, size=(C, N)); h = rng.integers(1, box + 1, size=(C, N))
    x1 = np.minimum(x0 + w, TW); y1 = np.minimum(y0 + h, TH)
    visible = rng.random((C, N)) < vis_p
    tiles = np.where(visible, (x1 - x0) * (y1 - y0), 0).astype(np.int32)
    depth = (0.5 + rng.integers(0, rnd.choice([3, 1 << 20]), size=(C, N)) * 0.37).astype(np.float32)
    rec = np.zeros((C, N, B.GS_REC), dtype=np.float32)
    rec[..., 10] = depth
    rec[..., 13] = (x0 | (y0 << 16)).astype(np.int32).view(np.float32)
    rec[..., 14] = (x1 | (y1 << 16)).astype(np.int32).view(np.float32)
    rec_d, tiles_d = torch.from_numpy(rec).to(dev), torch.from_numpy(tiles).to(dev)
    saved = (W.STAGED_SCATTER, W.SCATTER_WORKGROUPS, W.SCATTER_REGIONS)
    outs = []
    cfgs = [(False, 256, 0), (True, 256, 0), (True, rnd.choice([1, 2, 7, 64]), rnd.choice([0, 1, 2, 3, 5]))]
    try:
        for cfg in cfgs:
            W.STAGED_SCATTER, W.SCATTER_WORKGROUPS, W.SCATTER_REGIONS = cfg
            k, f, o, _, fs = W.isect_tiles_binned(rec_d, tiles_d, TW, TH, front=rnd.choice([False, True]))
            if fs is not None:
                fs.finalize()
            outs.append((k.clone(), f.clone(), o.clone()))
    finally:
        W.STAGED_SCATTER, W.SCATTER_WORKGROUPS, W.SCATTER_REGIONS = saved
    ok = all(torch.equal(a, b) for o2 in outs[1:] for a, b in zip(outs[0], o2))
    ok = ok and int(tiles.sum()) == outs[0][1].numel()
    # keys ascending, ties by id
    k0, f0 = outs[0][0], outs[0][1].long()
    if k0.numel() > 1:
        ok = ok and bool(((k0[1:] > k0[:-1]) | ((k0[1:] == k0[:-1]) & (f0[1:] > f0[:-1]))).all())
    bad += not ok
    print(("ok  " if ok else "FAIL") + f" case {case}: C={C} N={N} {TW}x{TH} box<={box} vis={vis_p} pairs={outs[0][1].numel()} cfg3={cfgs[2]}", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
