"""Host time per FFI entry point and per torch call class over fwd + bwd steps of one camera (launch-bound shapes)."""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-reconstruction_amd")]
import torch
import synth_scene
from gsplat import _backend as B
from gsplat.rendering import rasterization

dev = torch.device("cuda:0")
N, W, H = (int(os.environ.get("PH_N", 100000)), int(os.environ.get("PH_W", 1242)), int(os.environ.get("PH_H", 375)))
g = {k: v.to(dev).requires_grad_(True) for k, v in synth_scene.make_gaussians(N, sh_degree=None, extent=10.0).items()}
vm, Ks = (t.to(dev) for t in synth_scene.make_cameras(1, W, H))
acc = collections.defaultdict(lambda: [0, 0.0])
real = B.lib()


class Proxy:
    def __getattr__(self, name):
        fn = getattr(real, name)

        def timed(*a):
            t0 = time.perf_counter()
            r = fn(*a)
            e = acc[name]
            e[0] += 1
            e[1] += time.perf_counter() - t0
            return r

        setattr(self, name, timed)
        return timed


proxy = Proxy()
B.lib = lambda: proxy


def step():
    for v in g.values():
        v.grad = None
    rc, ra, info = rasterization(g["means"], g["quats"], g["scales"], g["opacities"], g["colors"], vm, Ks, W, H,
                                 near_plane=0.1, packed=False, absgrad=True, render_mode="RGB+ED")
    (rc.sum() + ra.sum()).backward()


for _ in range(30):
    step()
torch.cuda.synchronize()
acc.clear()
n = 300
t0 = time.perf_counter()
for _ in range(n):
    step()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"N={N} {W}x{H}: host issue {t_issue / n * 1e3:.4f} ms/step, with final sync {t_all / n * 1e3:.4f} ms/step")
tot = 0.0
for name, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"  {name:32s} {c / n:5.1f} calls/step  {t / n * 1e6:8.1f} us/step")
    tot += t
print(f"  {'FFI total':32s} {'':16s} {tot / n * 1e6:8.1f} us/step")
