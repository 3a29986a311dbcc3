"""Where the HOST time of one fwd + bwd step goes above the FFI: inclusive wall time of the Python-level functions of
the step (wrappers with perf_counter_ns, ~0.3 us each), one camera, launch-bound shape.  Companion of prof_ffi.py."""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-reconstruction_amd")]
import torch
import synth_scene
from gsplat import _backend as B
from gsplat import rendering as R
from gsplat.cuda import _wrapper as W

dev = torch.device("cuda:0")
N, Wd, H = (int(os.environ.get("PH_N", 100000)), int(os.environ.get("PH_W", 1242)), int(os.environ.get("PH_H", 375)))
g = {k: v.to(dev).requires_grad_(True) for k, v in synth_scene.make_gaussians(N, sh_degree=None, extent=10.0).items()}
vm, Ks = (t.to(dev) for t in synth_scene.make_cameras(1, Wd, H))
acc = collections.defaultdict(lambda: [0, 0])
ns = time.perf_counter_ns


def wrap(fn, name):
    def timed(*a, **k):
        t0 = ns()
        try:
            return fn(*a, **k)
        finally:
            e = acc[name]
            e[0] += 1
            e[1] += ns() - t0
    return timed


def patch(obj, attr, name=None, static=False):
    fn = getattr(obj, attr)
    w = wrap(fn, name or f"{getattr(obj, '__name__', obj)}.{attr}")
    setattr(obj, attr, staticmethod(w) if static else w)


if os.environ.get("PH_WRAP", "1") == "1":
    for cls in (W._ProjectSplats, W._RasterizeSplats):
        patch(cls, "forward", f"{cls.__name__}.forward", static=True)
        patch(cls, "backward", f"{cls.__name__}.backward", static=True)
    for a in ("isect_tiles_binned", "launch_raster_fwd", "_make_params", "_camera_centers", "take_front_sort",
              "take_raster_state", "bin_max_tiles", "_post_slot", "_spec_table"):
        if hasattr(W, a):
            patch(W, a, f"W.{a}")
    patch(W.SpecLists, "resolve", "SpecLists.resolve")
    for a in ("require_gpu", "f32c", "ptr", "stream_handle", "device_guard", "check", "side_stream"):
        patch(B, a, f"B.{a}")
    patch(torch, "empty", "torch.empty")
    patch(torch, "empty_like", "torch.empty_like")
    patch(torch, "zeros", "torch.zeros")
    patch(R, "_Meta", "R._Meta()")
rast = wrap(R.rasterization, "rasterization()")


def step():
    t0 = ns()
    for v in g.values():
        v.grad = None
    t1 = ns()
    rc, ra, info = rast(g["means"], g["quats"], g["scales"], g["opacities"], g["colors"], vm, Ks, Wd, H,
                        near_plane=0.1, packed=False, absgrad=True, render_mode="RGB+ED")
    t2 = ns()
    loss = rc.sum() + ra.sum()
    t3 = ns()
    loss.backward()
    t4 = ns()
    for k, d in (("step: zero grads", t1 - t0), ("step: loss fwd (2 sums + add)", t3 - t2), ("step: backward()", t4 - t3)):
        e = acc[k]
        e[0] += 1
        e[1] += d


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
print(f"N={N} {Wd}x{H}: host issue {t_issue / n * 1e3:.4f} ms/step, with final sync {t_all / n * 1e3:.4f} ms/step")
for name, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"  {name:36s} {c / n:6.1f} calls/step  {t / n / 1e3:8.1f} us/step (inclusive)")
