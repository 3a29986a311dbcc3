"""Which Gaussian decides the re-run difference of v_quats in test_split_equals_fused[6000-3-200-120-3-16]?"""
import sys, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, os.path.join(ROOT, "3d-reconstruction_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p_)
import synth_scene
from test_gpu_split_backward import _grads
dev = torch.device("cuda:0")
N, C, W, H = 6000, 3, 200, 120
g = synth_scene.make_gaussians(N, sh_degree=3, extent=10.0, seed=N)
vm, Ks = synth_scene.make_cameras(C, W, H)
runs = [_grads(dev, g, vm, Ks, W, H, 3, split=(i % 2 == 1)) for i in range(6)]
ref = runs[0]["quats"].double().cpu()
for i, r in enumerate(runs[1:], 1):
    d = (r["quats"].double().cpu() - ref)
    row = int(d.abs().amax(1).argmax())
    print(f"run {i}: rel_l2 {float(d.norm() / ref.norm()):.3e} worst row {row} err {float(d.abs().max()):.3e}")
row = int(max(((r["quats"].double().cpu() - ref).abs().amax(1)) .max(0) for r in runs[1:])[1]) if False else row
mc = (vm[:, :3, :3] @ g["means"][row] + vm[:, :3, 3])
print("row", row, "depth per camera", mc[:, 2].tolist(), "scales", g["scales"][row].exp().tolist() if g["scales"][row].max() < 0 else g["scales"][row].tolist(),
      "opacity", float(g["opacities"][row]), "v_quats", ref[row].tolist(), "max|v_quats|", float(ref.abs().max()))
