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
        q.put({k: v.grad.clone() for k, v in params.items()} | {"vis": vis, "gsum": gsum, "rmax": rmax})
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
    got = q.get(timeout=300)
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
