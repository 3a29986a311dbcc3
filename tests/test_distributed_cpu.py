"""world_size-2 gloo test (CPU) of the camera-sharded data-parallel path: per-rank renders of disjoint camera shards +
SUM all-reduce of the parameter gradients == the single-process C-camera result (the defined semantics of C > 1,
SURVEY.md section 8(e)).  The per-rank compute is the CPU oracle here (tests may use it); on the GPU the same helpers
wrap gsplat.rendering.rasterization over RCCL."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _by_value(d):
    """Tensors leave the worker as numpy arrays: pickled by value.  A tensor is passed as a file descriptor that the
    parent fetches from the worker when it unpickles -- too late if the worker has exited by then."""
    return {k: (v.detach().numpy() if torch.is_tensor(v) else v) for k, v in d.items()}


def _tensors(d):
    import numpy as np

    return {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in d.items()}


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "3d-reconstruction_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle
    import synth_scene
    from gsplat.distributed import (all_reduce_densification_stats, all_reduce_gradients, rasterize_sharded, shard_views)

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    N, C, W, H = 300, 4, 48, 32
    g = synth_scene.make_gaussians(N, sh_degree=None, extent=6.0, dtype=torch.float64)
    vm, Ks = synth_scene.make_cameras(C, W, H, dtype=torch.float64)
    params = {k: v.clone().requires_grad_(True) for k, v in g.items()}

    def render(viewmats, Ks):
        return oracle.rasterization(params["means"], params["quats"], params["scales"], params["opacities"],
                                    params["colors"], viewmats, Ks, W, H, render_mode="RGB+ED")

    (rc, ra, meta), views = rasterize_sharded(render, vm, Ks)
    assert list(views) == list(shard_views(C, rank, world))
    gen = torch.Generator().manual_seed(3)
    wc = torch.randn(C, H, W, 4, generator=gen, dtype=torch.float64)[views.start:views.stop]
    ((rc * wc).sum()).backward()
    all_reduce_gradients(list(params.values()))
    vis = (meta["radii"] > 0).sum(0).double()
    gsum = torch.ones(N, dtype=torch.float64) * (rank + 1)
    rmax = meta["radii"].max(0).values.double()
    all_reduce_densification_stats(gsum, vis, rmax)
    if rank == 0:
        q.put(_by_value({k: v.grad.clone() for k, v in params.items()} | {"vis": vis, "gsum": gsum, "rmax": rmax}))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_camera_sharding_matches_single_process():
    sys.path.insert(0, ROOT)
    import oracle
    import synth_scene

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = _tensors(q.get(timeout=300))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0

    N, C, W, H = 300, 4, 48, 32
    g = synth_scene.make_gaussians(N, sh_degree=None, extent=6.0, dtype=torch.float64)
    vm, Ks = synth_scene.make_cameras(C, W, H, dtype=torch.float64)
    params = {k: v.clone().requires_grad_(True) for k, v in g.items()}
    rc, ra, meta = oracle.rasterization(params["means"], params["quats"], params["scales"], params["opacities"],
                                        params["colors"], vm, Ks, W, H, render_mode="RGB+ED")
    gen = torch.Generator().manual_seed(3)
    wc = torch.randn(C, H, W, 4, generator=gen, dtype=torch.float64)
    (rc * wc).sum().backward()
    for k, v in params.items():
        assert torch.allclose(got[k], v.grad, rtol=1e-10, atol=1e-12), k
    assert torch.equal(got["vis"], (meta["radii"] > 0).sum(0).double())
    assert torch.equal(got["rmax"], meta["radii"].max(0).values.double())
    assert torch.equal(got["gsum"], torch.full((N,), 3.0, dtype=torch.float64))


# ------------------------------------------------------------------------------------- colour-gradient exchange
def _colour_worker(rank, world, port, q):
    """One rank of the "colour" exchange (gsplat.distributed.GradientExchange) on CPU: the per-rank compute is the fp64
    oracle, the message is packed / rebuilt by tests/colour_message_ref.py (the numpy restatement of what
    gs_projection_bwd_split / gs_sh_bwd_packed do on the GPU), the collectives are the product's, over gloo."""
    for p in (ROOT, os.path.join(ROOT, "3d-reconstruction_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import numpy as np

    import colour_message_ref as M
    import oracle
    import synth_scene
    from gsplat.distributed import GradientExchange, all_reduce_gradients, shard_views

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    N, C, W, H, deg = 300, 5, 48, 32, 2
    g = synth_scene.make_gaussians(N, sh_degree=3, extent=6.0, dtype=torch.float64)
    vm, Ks = synth_scene.make_cameras(C, W, H, dtype=torch.float64)
    views = shard_views(C, rank, world)  # 3 + 2 cameras: ranks with different camera counts
    sl = slice(views.start, views.stop)
    Cl = views.stop - views.start
    p = {k: v.clone().requires_grad_(True) for k, v in g.items()}
    # this rank's cameras, with the graph cut at the per-(camera, Gaussian) colours: what the geometry kernel sees
    radii, m2, depths, conics, _ = oracle.gsplat_ref.fully_fused_projection(p["means"], p["quats"], p["scales"], vm[sl],
                                                                            Ks[sl], W, H, 0.3, 0.01, 1e10, 0.0)
    campos = torch.linalg.inv(vm[sl])[:, :3, 3]
    dirs = (p["means"][None] - campos[:, None, :]).detach()
    pre = oracle.spherical_harmonics(deg, dirs, p["colors"].detach()[None].expand(Cl, -1, -1, -1), masks=radii > 0)
    pre = (pre + 0.5).requires_grad_(True)  # leaf: its gradient is the clamp-gated v_rgb of every (c, n)
    cols = torch.cat([torch.clamp_min(pre, 0.0), depths[..., None]], -1)
    TW, TH = (W + 15) // 16, (H + 15) // 16
    _, ids, fids = oracle.isect_tiles(m2, radii, depths, 16, TW, TH)
    offs = oracle.isect_offset_encode(ids, Cl, TW, TH)
    rc, ra, _ = oracle.rasterize_to_pixels(m2, conics, cols, p["opacities"][None].expand(Cl, -1), W, H, 16, offs, fids)
    gen = torch.Generator().manual_seed(3)
    wc = torch.randn(C, H, W, 4, generator=gen, dtype=torch.float64)[sl]
    (rc * wc).sum().backward()

    ex = GradientExchange(mode="colour")
    assert ex.active and ex.packs_colours
    vis = (radii > 0).numpy()
    stat = ex.max_over_ranks(torch.tensor([int(vis.sum()), Cl]))
    cap, lay_cams = int(stat[0]), int(stat[1])
    msg = torch.from_numpy(M.pack(vis, pre.grad.float().numpy(), campos.float().numpy(), lay_cams, cap))
    msgs, n_msgs, local = ex.all_gather_messages(msg)
    assert n_msgs == world and local == rank and torch.equal(msgs[rank], msg)
    geo = [p[k].grad for k in ("means", "quats", "scales", "opacities")]
    ex.all_reduce(geo).wait()
    v_sh, v_means_sh = M.rebuild([m.numpy() for m in msgs], lay_cams, N, cap, p["means"], p["colors"], deg)
    out = {"means": p["means"].grad + v_means_sh, "quats": p["quats"].grad, "scales": p["scales"].grad,
           "opacities": p["opacities"].grad, "colors": v_sh, "sent": ex.bytes_sent}

    # all_reduce_gradients: a generator argument with average=True (it is walked twice)
    probe = [torch.nn.Parameter(torch.full((3,), float(rank + 1), dtype=torch.float64)) for _ in range(2)]
    for t in probe:
        t.grad = torch.full((3,), float(rank + 1), dtype=torch.float64)
    all_reduce_gradients((t for t in probe), average=True)
    out["avg"] = probe[1].grad.clone()
    if rank == 0:
        q.put(_by_value(out))
    dist.barrier()
    dist.destroy_process_group()


def test_colour_exchange_two_ranks_matches_single_process():
    """Packed colour-gradient all-gather + local v_sh rebuild + geometry all-reduce == the single-process gradients of
    the 5-camera batch (fp32 message payload: compared at 1e-6 relative)."""
    sys.path.insert(0, ROOT)
    import oracle
    import synth_scene

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_colour_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = _tensors(q.get(timeout=300))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0

    N, C, W, H, deg = 300, 5, 48, 32, 2
    g = synth_scene.make_gaussians(N, sh_degree=3, extent=6.0, dtype=torch.float64)
    vm, Ks = synth_scene.make_cameras(C, W, H, dtype=torch.float64)
    params = {k: v.clone().requires_grad_(True) for k, v in g.items()}
    rc, ra, meta = oracle.rasterization(params["means"], params["quats"], params["scales"], params["opacities"],
                                        params["colors"], vm, Ks, W, H, sh_degree=deg, render_mode="RGB+D")
    gen = torch.Generator().manual_seed(3)
    wc = torch.randn(C, H, W, 4, generator=gen, dtype=torch.float64)
    (rc * wc).sum().backward()
    assert int((meta["radii"] > 0).sum()) > 100
    for k, v in params.items():
        ref = v.grad
        err = float((got[k] - ref).norm() / ref.norm())
        assert err <= 1e-6, (k, err)
    assert float(params["colors"].grad[:, (deg + 1) ** 2:].abs().max()) == 0.0
    assert float(got["colors"][:, (deg + 1) ** 2:].abs().max()) == 0.0
    assert torch.equal(got["avg"], torch.full((3,), 1.5, dtype=torch.float64))
    assert got["sent"] < N * 236  # less than the dense exchange even at this tiny, mostly visible scene
