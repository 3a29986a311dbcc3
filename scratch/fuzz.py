"""Randomised end-to-end parity sweep against the oracle (fwd + bwd), many small configurations."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "3d-reconstruction_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import oracle, synth_scene
from gsplat.rendering import rasterization
dev = torch.device("cuda:0")
if os.environ.get("FUZZ_NOCULL"):
    import gsplat.cuda._wrapper as _W
    _W.CULL_LISTS = False
base_seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
only = int(sys.argv[3]) if len(sys.argv) > 3 else None
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
worst = {}
for case in range(n_cases):
    rnd = random.Random(base_seed * 100003 + case)
    N = rnd.choice([1, 7, 500, 3000, 12000] if base_seed < 100 else [30000, 60000, 120000])
    C = rnd.choice([1, 1, 2, 3])
    W = rnd.choice([17, 64, 130, 201]); H = rnd.choice([16, 33, 75, 120])
    sh = rnd.choice([None, 0, 1, 2, 3, 4])
    mode = rnd.choice(["RGB", "RGB+D", "RGB+ED", "D", "ED"])
    rmode = rnd.choice(["classic", "antialiased"])
    packed = rnd.choice([False, False, True])
    bgs = rnd.choice([False, True]) and mode not in ("D", "ED")
    rclip = rnd.choice([0.0, 0.0, 2.5])
    extent = rnd.choice([3.0, 10.0, 25.0] if base_seed < 100 else [2.0, 4.0, 8.0])
    dense = rnd.random() < 0.3
    if only is not None and case != only:
        continue
    mode = os.environ.get("FUZZ_MODE", mode)
    g = synth_scene.make_gaussians(N, sh_degree=sh, seed=case, extent=extent)
    if dense:
        g["scales"] = g["scales"] * 3.0
        g["opacities"] = g["opacities"] * rnd.choice([1.0, 0.05])
    if rnd.random() < 0.3:
        g["opacities"][: N // 5] = 0.0
    vm, Ks = synth_scene.make_cameras(C, W, H, frame=case % 5)
    if rnd.random() < 0.3:  # off-centre principal point
        Ks[:, 0, 2] += 3.3; Ks[:, 1, 2] -= 2.1
    bg = torch.rand(C, 3) if bgs else None
    gen = torch.Generator().manual_seed(case)
    kw = dict(near_plane=0.1, sh_degree=sh, render_mode=mode, rasterize_mode=rmode, radius_clip=rclip)
    p = {k: v.to(dev).requires_grad_(True) for k, v in g.items()}
    rc, ra, meta = rasterization(p["means"], p["quats"], p["scales"], p["opacities"], p["colors"], vm.to(dev), Ks.to(dev),
                                 W, H, packed=packed, absgrad=True, backgrounds=bg.to(dev) if bgs else None, **kw)
    wc = torch.randn(rc.shape, generator=gen); wa = torch.randn(ra.shape, generator=gen)
    ((rc * wc.to(dev)).sum() + (ra * wa.to(dev)).sum()).backward()
    q = {k: v.double().requires_grad_(True) for k, v in g.items()}
    rco, rao, mo = oracle.rasterization(q["means"], q["quats"], q["scales"], q["opacities"], q["colors"], vm.double(),
                                        Ks.double(), W, H, backgrounds=bg.double() if bgs else None, **kw)
    lo = (rco * wc.double()).sum() + (rao * wa.double()).sum()
    if lo.requires_grad:
        lo.backward()
    def bad(a, b, atol=1e-5, rtol=1e-4):
        a, b = a.detach().double().cpu(), b.detach().double()
        return float(((a - b).abs() > atol + rtol * b.abs()).double().mean()) if b.numel() else 0.0
    stats = {"rc": bad(rc, rco), "ra": bad(ra, rao)}
    for k in q:
        if q[k].grad is None: continue
        ref = q[k].grad; got = p[k].grad.double().cpu()
        stats["g_" + k] = float((got - ref).norm() / (ref.norm() + 1e-12)) if float(ref.norm()) > 1e-9 else float((got - ref).norm())
    if not packed:
        stats["radii_mismatch"] = float((meta["radii"].cpu() != mo["radii"]).double().mean())
        stats["isect_n"] = abs(meta["flatten_ids"].numel() - mo["flatten_ids"].numel()) / max(1, mo["flatten_ids"].numel())
    if only is not None:
        for k in q:
            if q[k].grad is not None:
                print(k, "ref", q[k].grad.flatten()[:12].tolist(), "\n   got", p[k].grad.flatten()[:12].cpu().tolist())
        print("radii", meta["radii"].cpu().tolist(), mo["radii"].tolist())
    flag = any(v > (5e-3 if k in ("rc", "ra", "radii_mismatch", "isect_n") else 2e-3) for k, v in stats.items())
    for k, v in stats.items():
        worst[k] = max(worst.get(k, 0.0), v)
    if flag or case % 10 == 0:
        print(("FLAG " if flag else "ok   ") + f"case {case}: N={N} C={C} {W}x{H} sh={sh} {mode} {rmode} packed={packed} bg={bgs} clip={rclip} ext={extent} dense={dense} "
              + " ".join(f"{k}={v:.1e}" for k, v in stats.items()), flush=True)
    assert all(torch.isfinite(v.grad).all() for v in p.values() if v.grad is not None), case
print("worst:", " ".join(f"{k}={v:.1e}" for k, v in worst.items()))
