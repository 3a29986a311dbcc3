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
  losses_ref.npz          inputs + outputs (values and input gradients, fp64) of the REFERENCE's own loss file
                          /root/reference/3D-Reconstruction-main/models/losses.py (numpy + torch only; imported from
                          where it lies): SafeBCE forward / backward (:33-75; limit 0.1, inputs outside [0,1], the
                          x == y gradient gate), safe_binary_cross_entropy / binary_cross_entropy "mean" (:77-83) on the
                          masks of base.py:536-550, DepthLoss (:91-176) for l1 / l2 / smooth_l1 x normalize x inverse x
                          the four reductions, two percentiles and an empty hit set, called as at base.py:553-557.
                          Pins oracle/loss_ref.py (tests/test_oracle_cpu.py) and the HIP loss kernels
                          (tests/test_gpu_losses.py).

  rigid_ref.npz           the rigid nodes' local -> world transform (models/nodes/rigid.py:426-470: R(quat_act(q_inst)) mean
                          + t_inst, quat_act(q_inst) (x) quat_act(q)) evaluated with the REFERENCE's own helpers from
                          utils/geometry.py: quat_to_rotmat (:116-152, the one rigid.py binds) and
                          safe_quaternion_multiply (:170-216, the same Hamilton product as models/gaussians/basics.py:64-74
                          `quat_mult`, whose module cannot be imported here), with gradients of a fixed linear loss.
                          Pins gsplat.prepass.rigid_transform (tests/test_gpu_prepass.py).

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
REF_LOSSES = "/root/reference/3D-Reconstruction-main/models/losses.py"

DEPTH_CASES = [(t, n, i, r) for t in ("l1", "l2", "smooth_l1") for n in (False, True) for i in (False, True)
               for r in ("mean_on_hit", "mean_on_hw", "sum", "none")]
PERCENTILE_CASES = [("l1", False, False, "mean_on_hit", 0.95), ("l2", True, False, "sum", 0.5)]


def depth_case_key(loss_type, normalize, inverse, reduction, percentile=None):
    k = f"depth_{loss_type}_n{int(normalize)}_i{int(inverse)}_{reduction}"
    return k if percentile is None else f"{k}_p{int(round(percentile * 100))}"


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


def rigid_golden():
    spec = importlib.util.spec_from_file_location("ref_geometry", REF_GEOMETRY)
    geo = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(geo)
    n, n_inst = 3001, 11
    g = torch.Generator().manual_seed(31)
    means, quats = torch.randn(n, 3, generator=g) * 2, torch.randn(n, 4, generator=g)
    iq, it = torch.randn(n_inst, 4, generator=g) * 1.3, torch.randn(n_inst, 3, generator=g) * 10
    pid = torch.randint(0, n_inst, (n, 1), generator=g).sort(dim=0).values
    wm_w, wq_w = torch.randn(n, 3, generator=g), torch.randn(n, 4, generator=g)
    ref = [t.double().requires_grad_(True) for t in (means, quats, iq, it)]
    quat_act = lambda x: x / x.norm(dim=-1, keepdim=True)          # vanilla.py:136-137
    rot_per_pts = geo.quat_to_rotmat(quat_act(ref[2]))[pid[..., 0]]  # rigid.py:426-429
    wm = torch.bmm(rot_per_pts, ref[0].unsqueeze(-1)).squeeze(-1) + ref[3][pid[..., 0]]   # :447-453
    wq = geo.safe_quaternion_multiply(quat_act(ref[2][pid[..., 0]]), quat_act(ref[1]))     # :464-470
    ((wm * wm_w.double()).sum() + (wq * wq_w.double()).sum()).backward()
    np.savez_compressed(os.path.join(HERE, "rigid_ref.npz"), means=means.numpy(), quats=quats.numpy(),
                        inst_quats=iq.numpy(), inst_trans=it.numpy(), point_ids=pid.numpy(), wm_w=wm_w.numpy(),
                        wq_w=wq_w.numpy(), world_means=wm.detach().numpy(), world_quats=wq.detach().numpy(),
                        grad_means=ref[0].grad.numpy(), grad_quats=ref[1].grad.numpy(),
                        grad_inst_quats=ref[2].grad.numpy(), grad_inst_trans=ref[3].grad.numpy())


def losses_golden():
    spec = importlib.util.spec_from_file_location("ref_losses", REF_LOSSES)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    H, W = 41, 57
    g = torch.Generator().manual_seed(23)
    opacity = torch.rand(H, W, 1, generator=g).clamp(1e-4, 1 - 1e-4)
    opacity[0, :5] = 0.0    # exact 0 / 1: the clipped ends of both BCE forms, and x == y where the target agrees
    opacity[1, :5] = 1.0
    opacity_wild = opacity.clone()      # SafeBCE only: inputs outside [0, 1] (clipped in forward, losses.py:48)
    opacity_wild[2, :9] = 1.0 + torch.rand(9, 1, generator=g) * 0.3
    opacity_wild[3, :9] = -torch.rand(9, 1, generator=g) * 0.3
    depth = torch.rand(H, W, 1, generator=g) * 90 + 0.05
    depth[4, :6] = 0.0      # below the 1e-4 validity gate (losses.py:123)
    sky = (torch.rand(H, W, generator=g) > 0.7).float()
    sky[0, :5], sky[1, :5] = 1.0, 0.0   # targets 0 / 1 under the exact opacities above
    valid = (torch.rand(H, W, generator=g) > 0.1).float()
    valid[0, :3] = 1.0
    valid[1, :3] = 1.0
    lidar = torch.where(torch.rand(H, W, generator=g) > 0.6, torch.rand(H, W, generator=g) * 100, torch.zeros(H, W))
    lidar[5, :7] = depth[5, :7, 0]      # exact hits: |d| = 0 rows of every loss type
    lidar[6, :7] = depth[6, :7, 0] + 0.5    # |d| < 1 (smooth-l1's quadratic branch) ...
    lidar[7, :7] = depth[7, :7, 0] - 3.0    # ... and > 1 whatever the random rows hold
    out = dict(opacity=opacity.numpy(), opacity_wild=opacity_wild.numpy(), depth=depth.numpy(), sky=sky.numpy(),
               valid=valid.numpy(), lidar=lidar.numpy(), limit=0.1, max_depth=80.0)
    d = lambda t: t.double()

    # SafeBCE itself, elementwise, with an upstream gradient (losses.py:41-75)
    gen_y = (torch.rand(H, W, generator=g) > 0.5).double()
    gen_y[2, :9] = (torch.arange(9) % 2).double()   # wild inputs against both targets
    gen_y[3, :9] = (torch.arange(9) % 2).double()
    go = torch.randn(H, W, generator=g).double()
    x = d(opacity_wild[..., 0]).requires_grad_(True)
    loss = ref.SafeBCE.apply(x, gen_y, 0.1)
    loss.backward(go)
    out.update(safe_y=gen_y.numpy(), safe_go=go.numpy(), safe_loss=loss.detach().numpy(), safe_grad_x=x.grad.numpy())

    # the mask loss as base.py:536-550 calls either form
    for name, op, fn in (("bce", opacity, lambda a, b: ref.binary_cross_entropy(a, b, reduction="mean")),
                         ("safe_bce", opacity, lambda a, b: ref.safe_binary_cross_entropy(a, b, limit=0.1, reduction="mean")),
                         ("safe_bce_wild", opacity_wild, lambda a, b: ref.safe_binary_cross_entropy(a, b, limit=0.1, reduction="mean"))):
        o = d(op).requires_grad_(True)
        gt_occupied = (1.0 - d(sky)) * d(valid)
        pred_occupied = o.squeeze() * d(valid)
        val = fn(pred_occupied, gt_occupied)
        val.backward()
        out[f"mask_{name}"] = val.detach().numpy()
        out[f"mask_{name}_grad"] = o.grad.numpy()

    # DepthLoss as base.py:553-557 calls it: hit mask = (lidar > 0) * valid
    def run_depth(loss_type, normalize, inverse, reduction, percentile, lid):
        fn = ref.DepthLoss(loss_type=loss_type, normalize=normalize, use_inverse_depth=inverse,
                           depth_error_percentile=percentile, upper_bound=80.0, reduction=reduction)
        p = d(depth).requires_grad_(True)
        hit = (d(lid) > 0).double() * d(valid)
        val = fn(p, d(lid), hit)
        if reduction != "none" and val.requires_grad and torch.isfinite(val):
            val.backward()
        return val.detach().numpy(), (p.grad.numpy() if p.grad is not None else np.zeros(p.shape))

    for (t, n, i, r) in DEPTH_CASES:
        v, gr = run_depth(t, n, i, r, None, lidar)
        k = depth_case_key(t, n, i, r)
        out[k] = v
        if r != "none":
            out[k + "_grad"] = gr
    for (t, n, i, r, pc) in PERCENTILE_CASES:
        v, gr = run_depth(t, n, i, r, pc, lidar)
        k = depth_case_key(t, n, i, r, pc)
        out[k], out[k + "_grad"] = v, gr
    for r in ("mean_on_hit", "mean_on_hw", "sum"):   # nothing hit: mean of an empty tensor is NaN, the sums are 0
        v, _ = run_depth("l1", False, False, r, None, torch.zeros(H, W))
        out[f"depth_empty_{r}"] = v
    np.savez_compressed(os.path.join(HERE, "losses_ref.npz"), **out)


if __name__ == "__main__":
    quat_golden()
    rigid_golden()
    losses_golden()
    pipeline_golden()
    sh_golden()
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))
