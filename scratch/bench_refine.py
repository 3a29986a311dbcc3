"""refine_gaussians vs the reference's torch lines (tests/test_gpu_refine.py::_RefClass) at 2.6 M Gaussians, K = 16."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-reconstruction_amd"), os.path.join(ROOT, "tests")]
import torch
from test_gpu_refine import _RefClass, _scene
from gsplat.densify import DensifyState, RefineConfig, refine_gaussians

dev = torch.device("cuda:0")
cfg = RefineConfig(densify_grad_thresh=0.0003, densify_size_thresh=0.05, scene_scale=1.3, cull_alpha_thresh=0.05,
                   split_screen_size=0.05, cull_scale_thresh=0.5, cull_screen_size=0.15)
p, adam, stats = _scene(dev, 2_600_000, seed=1)


def t(fn, iters=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def ref():
    r = _RefClass(p, adam, stats, cfg)  # (clones its inputs: timed separately below)
    r.refinement_after()


def clone_only():
    _RefClass(p, adam, stats, cfg)


def fused():
    return refine_gaussians(p, DensifyState(*stats), cfg, adam=adam)


print("info", fused()[2])
tc = t(clone_only)
print(f"torch lines {t(ref) - tc:.2f} ms (clone of the inputs {tc:.2f} ms excluded)   fused {t(fused):.2f} ms")
