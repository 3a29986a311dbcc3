"""Experiment: the intersection stage + forward rasterizer of the headline step on camera groups over two HIP streams
(existing kernels on views of the records; no gradients) against the single call over all cameras."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-reconstruction_amd")]
import torch
import synth_scene
from gsplat import _backend as B
from gsplat.cuda import _wrapper as Wm

dev = torch.device("cuda:0")
N, C, W, H = 1_000_000, 5, 1242, 375
g = {k: v.to(dev) for k, v in synth_scene.make_gaussians(N, sh_degree=3, seed=42).items()}
vm, Ks = (t.to(dev) for t in synth_scene.make_cameras(C, W, H))
params = Wm._make_params(N, C, W, H, 16, 0.3, 0.1, 1e10, 0.0, True, B.GS_COLOR_SH, 3, 3, 16, True)
with torch.no_grad():
    splats, m2d, radii, tiles = Wm._ProjectSplats.apply(g["means"], g["quats"], g["scales"], g["opacities"], g["colors"],
                                                        vm, Ks, None, params, None)
lanes = [torch.cuda.Stream(device=dev) for _ in range(3)]
main = torch.cuda.current_stream(dev)


def single():
    with torch.no_grad():
        return Wm._RasterizeSplats.apply(splats, m2d, tiles, None, W, H, 16, 4, True, False, None, True)[:2]


def grouped(groups):
    outs = []
    with torch.no_grad():
        for i, (c0, c1) in enumerate(groups):
            ln = lanes[i % len(lanes)]
            ln.wait_stream(main)
            with torch.cuda.stream(ln):
                outs.append(Wm._RasterizeSplats.apply(splats[c0:c1], m2d[c0:c1], tiles[c0:c1], None, W, H, 16, 4, True,
                                                      False, None, True)[:2])
        for ln in lanes:
            main.wait_stream(ln)
    return outs


def bench(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n, (time.perf_counter() - t0) / n * 1e3


for spec in (0, 1):
    Wm.SPECULATE_SIZES = bool(spec)
    ref = single()
    for name, groups in (("3+2", [(0, 3), (3, 5)]), ("2+2+1", [(0, 2), (2, 4), (4, 5)]), ("1x5", [(c, c + 1) for c in range(5)])):
        outs = grouped(groups)
        torch.cuda.synchronize()
        same = torch.equal(torch.cat([o[0] for o in outs]), ref[0]) and torch.equal(torch.cat([o[1] for o in outs]), ref[1])
        print(f"speculate={spec} groups {name}: bit-identical to the single call: {same}")
    print(f"speculate={spec} single call          : gpu {bench(single)[0]:.4f} ms  (host loop {bench(single)[1]:.4f})")
    for name, groups in (("3+2", [(0, 3), (3, 5)]), ("2+2+1", [(0, 2), (2, 4), (4, 5)]), ("1x5", [(c, c + 1) for c in range(5)])):
        t = bench(lambda: grouped(groups))
        print(f"speculate={spec} groups {name:6s} over lanes: gpu {t[0]:.4f} ms  (host loop {t[1]:.4f})")
