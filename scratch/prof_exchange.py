"""Host timeline of the forced one-rank exchange step at the headline shape: where does the host wait, how long do the
collective calls take on the host?  scratch/prof_exchange.py [0|1]  (1 = exchange forced)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-reconstruction_amd")]
import torch
import torch.distributed as dist
import synth_scene
from gsplat.rendering import rasterization
from gsplat.distributed import GradientExchange
from gsplat.cuda import _wrapper as W
from gsplat import _backend as B

force = len(sys.argv) > 1 and sys.argv[1] == "1"
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
if force:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
N, C, Wd, Hd = 1_000_000, 5, 1242, 375
g = synth_scene.make_gaussians(N, sh_degree=3, seed=42)
p = {k: v.to(dev).requires_grad_(True) for k, v in g.items()}
vm, Ks = (t.to(dev) for t in synth_scene.make_cameras(C, Wd, Hd))
gen = torch.Generator().manual_seed(1)
wc = torch.randn(C, Hd, Wd, 4, generator=gen).to(dev).reshape(-1)
wa = torch.randn(C, Hd, Wd, 1, generator=gen).to(dev).reshape(-1)
ex = GradientExchange(force=force)
acc = {}
def timed(obj, name, label=None):
    fn = getattr(obj, name)
    label = label or name
    def wrap(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
    setattr(obj, name, wrap)
lib = B.lib()
class LibProxy:
    def __init__(self, lib): self._lib = lib
    def __getattr__(self, n):
        f = getattr(self._lib, n)
        if n in ("gs_host_post_wait",):
            def w(*a):
                t0 = time.perf_counter(); r = f(*a); acc[n] = acc.get(n, 0.0) + time.perf_counter() - t0; return r
            return w
        return f
proxy = LibProxy(lib)
B_lib = B.lib
B.lib = lambda: proxy
timed(ex, "max_over_ranks"); timed(ex, "all_gather_messages"); timed(ex, "all_reduce")
timed(W._ProjectSplats, "forward", "project.forward"); timed(W._ProjectSplats, "backward", "project.backward")
timed(W._RasterizeSplats, "forward", "raster.forward"); timed(W._RasterizeSplats, "backward", "raster.backward")
def step():
    for v in p.values():
        v.grad = None
    t0 = time.perf_counter()
    with ex:
        rc, ra, meta = rasterization(p["means"], p["quats"], p["scales"], p["opacities"], p["colors"], vm, Ks, Wd, Hd,
                                     sh_degree=3, near_plane=0.1, packed=False, absgrad=True, render_mode="RGB+ED",
                                     rasterize_mode="antialiased")
    t1 = time.perf_counter()
    loss = torch.dot(rc.reshape(-1), wc) + torch.dot(ra.reshape(-1), wa)
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    for k, v in (("host.forward", t1 - t0), ("host.loss", t2 - t1), ("host.backward()", t3 - t2)):
        acc[k] = acc.get(k, 0.0) + v
for _ in range(10):
    step()
torch.cuda.synchronize(); acc.clear()
n = 50
t0 = time.perf_counter()
for _ in range(n):
    step()
torch.cuda.synchronize()
print(f"exchange forced = {force}: {(time.perf_counter() - t0) / n * 1e3:.4f} ms per step; host microseconds per step:")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:24s} {v / n * 1e6:8.1f}")
if force:
    dist.destroy_process_group()
