"""Generates the committed golden fixtures (run in the build container, where /root/reference exists).

  quat_to_rotmat_ref.npz  inputs + outputs of the REFERENCE's own quat_to_rotmat
                          (/root/reference/3D-Reconstruction-main/utils/geometry.py:116-152, imported from where it lies;
                          the only function on this path that exists inside /root/reference) -- pins the wxyz convention
                          of gsplat.cuda_legacy._torch_impl.quat_to_rotmat and of the projection kernels.
  pipeline_fp64.npz       inputs, forward outputs and gradients of the fp64 CPU oracle (oracle/gsplat_ref.py) on a small
                          seeded scene (N=600, C=2, 80x48, SH degree 3, RGB+ED, antialiased): G1 projection, G2 integer
                          stages, G3 raster forward, G4 gradients of a fixed random linear loss.  PARITY UNPINNED w.r.t.
                          gsplat itself: these are the oracle's outputs, not gsplat's.
  sh_fp64.npz             G5: SH degree 0..4 forward + v_coeffs + v_dirs of the oracle.

    python tests/golden/make_golden.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
import synth_scene  # noqa: E402

REF_GEOMETRY = "/root/reference/3D-Reconstruction-main/utils/geometry.py"


def quat_golden():
    spec = importlib.util.spec_from_file_location("ref_geometry", REF_GEOMETRY)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g = torch.Generator().manual_seed(0)
    q = torch.randn(64, 4, generator=g, dtype=torch.float64)
    q[0] = torch.tensor([1.0, 0, 0, 0])
    q[1] = torch.tensor([2.0, 0, 0, 0])          # un-normalised identity
    q[2] = torch.tensor([0.5, 0.5, 0.5, 0.5])    # cyclic permutation
    q[3] = torch.tensor([0.0, 1.0, 0, 0])        # 180 deg about x
    R = mod.quat_to_rotmat(q)
    np.savez(os.path.join(HERE, "quat_to_rotmat_ref.npz"), quats=q.numpy(), rotmats=R.numpy())


def pipeline_golden():
    N, C, W, H = 600, 2, 80, 48
    g = synth_scene.make_gaussians(N, sh_degree=3, extent=8.0, seed=7)   # fp32 values
    vm, Ks = synth_scene.make_cameras(C, W, H)
    g["opacities"][:20] = 0.0
    g["scales"][20:30, 2] = 0.0
    go = {k: v.double().requires_grad_(True) for k, v in g.items()}
    vmo = vm.double().requires_grad_(True)
    rc, ra, meta = oracle.rasterization(go["means"], go["quats"], go["scales"], go["opacities"], go["colors"], vmo,
                                        Ks.double(), W, H, near_plane=0.1, sh_degree=3, render_mode="RGB+ED",
                                        rasterize_mode="antialiased")
    gen = torch.Generator().manual_seed(11)
    wc = torch.randn(C, H, W, 4, generator=gen, dtype=torch.float64)
    wa = torch.randn(C, H, W, 1, generator=gen, dtype=torch.float64)
    meta["means2d"].retain_grad()
    ((rc * wc).sum() + (ra * wa).sum()).backward()
    out = {f"in_{k}": v.numpy() for k, v in g.items()}
    out.update(in_viewmats=vm.numpy(), in_Ks=Ks.numpy(), width=W, height=H, wc=wc.numpy(), wa=wa.numpy())
    out.update(render_colors=rc.detach().numpy(), render_alphas=ra.detach().numpy())
    for k in ["radii", "means2d", "depths", "conics", "opacities", "tiles_per_gauss", "isect_ids", "flatten_ids",
              "isect_offsets", "last_ids"]:
        out[f"meta_{k}"] = meta[k].detach().numpy()
    for k, v in go.items():
        out[f"grad_{k}"] = v.grad.numpy()
    out["grad_viewmats"] = vmo.grad.numpy()
    out["grad_means2d"] = meta["means2d"].grad.numpy()
    np.savez_compressed(os.path.join(HERE, "pipeline_fp64.npz"), **out)


def sh_golden():
    out = {}
    g = torch.Generator().manual_seed(5)
    n = 200
    dirs = torch.randn(n, 3, generator=g).double() * 2.0
    coeffs = torch.randn(n, 25, 3, generator=g).double()
    v = torch.randn(n, 3, generator=g).double()
    out.update(dirs=dirs.numpy(), coeffs=coeffs.numpy(), v=v.numpy())
    for deg in range(5):
        d, c = dirs.clone().requires_grad_(True), coeffs.clone().requires_grad_(True)
        o = oracle.spherical_harmonics(deg, d, c)
        (o * v).sum().backward()
        out[f"out{deg}"] = o.detach().numpy()
        out[f"v_coeffs{deg}"] = c.grad.numpy()
        out[f"v_dirs{deg}"] = (d.grad if d.grad is not None else torch.zeros_like(d)).numpy()
    np.savez_compressed(os.path.join(HERE, "sh_fp64.npz"), **out)


if __name__ == "__main__":
    quat_golden()
    pipeline_golden()
    sh_golden()
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))
