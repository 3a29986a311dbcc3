#!/usr/bin/env python3
"""Headline benchmark: Mpix/s fwd+bwd of the gsplat `rasterization()` hot path on MI355X.

Workload (BASELINE.json configs[1], SURVEY.md section 8(d) "Config 2"): 1,000,000 Gaussians, SH degree 3 (K = 16),
5 cameras 1242x375 batched, render_mode "RGB+ED", absgrad + antialiased, packed=False; one step = SH + projection +
tile intersection + sort + offsets + rasterize forward, a fixed random linear loss over colours and alphas, and the
full backward to means / quats / scales / opacities / SH coefficients.  Synthetic scene (synth_scene.py, seed 42).

    python bench.py --gpus N --steps K --warmup W

N > 1 runs one rank per GPU over RCCL.  Either the caller launches the ranks (`python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N`: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment), or
bench.py is invoked bare and starts its N ranks itself as fresh child processes -- before anything in this process
touches the GPU -- and exits with their status.

With N ranks every rank renders its own 5-camera frame of the same Gaussians (weak scaling: cameras/frames shard,
section 8(e)) and the Gaussian-parameter gradients are exchanged inside the timed region (gsplat.distributed).
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "3d-reconstruction_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

# Multi-rank runs use three streams per process (main, the side stream of the record fill / message-size negotiation, the
# collectives' own): with HIP's default of 4 hardware queues per process, streams of PyTorch's pool land on shared queues
# and the small cross-stream hand-overs of the gradient exchange serialise behind main-stream kernels (measured, forced
# exchange on one rank: 1.29 -> 1.20 ms per step with 8 queues; the single-rank step does not change: 1.121 vs 1.122).
# Set before anything initialises HIP; an explicit setting of 8 or more is kept.
if int(os.environ.get("WORLD_SIZE", "1")) > 1 or "--force-exchange" in sys.argv:
    if int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) < 8:
        os.environ["GPU_MAX_HW_QUEUES"] = "8"

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import synth_scene  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak


def algorithmic_bytes(N, C, H, W, K, D, V, I, absgrad=True, aa=True):
    """SURVEY.md section 8(d) per-stage algorithmic HBM bytes for one step (fp32), keyed by the entry point that does
    the stage's work here.  gsplat's isect_tiles + sort + offset_encode stages are the binned pipeline's four kernels:
    `isect` (V*16 + I*12) is split over count (the V*16 box reads) and scatter (the I*12 pair writes), `sort` (I*24) is
    the per-tile sort, `offsets` (I*8 + T*4) the offsets kernel."""
    P = C * H * W
    T = C * ((H + 15) // 16) * ((W + 15) // 16)
    b = {}
    b["gs_projection_fwd"] = N * (12 + 12 * K) + V * 12 + N * 40 + C * N * 4 + V * (24 + (4 if aa else 0))
    b["gs_isect_bin_count"] = V * 16
    b["gs_isect_bin_scatter"] = I * 12
    b["gs_isect_tile_sort"] = I * 24
    b["gs_isect_bin_offsets"] = I * 8 + T * 4
    b["gs_rasterize_fwd"] = I * (28 + 4 * D) + P * (4 * D + 8)
    b["gs_rasterize_bwd"] = I * (28 + 4 * D) + P * (4 * D + 12) + V * (4 * D + 12 + 8 + 4 + (8 if absgrad else 0))
    b["gs_projection_bwd"] = V * (24 + (4 if aa else 0)) + N * 40 + C * N * 4 + N * 40 + V * 12 + N * 12 + N * 12 * K
    return b


def processed_bytes(N, C, H, W, K, D, V, Ic, Wk, absgrad=True):
    """The same per-unit figures with the units the kernels actually process in place of gsplat's I: Ic = entries of the
    culled tile lists the scatter / sort / rasterizers run on (pairs whose splat cannot reach alpha 1/255 on the tile
    are never emitted), Wk = list entries the compositor walks before every pixel of the tile has stopped (the forward
    reads that many records front to back, the backward the same ones back to front).  The sort moves 8-B (depth, id)
    pairs in and 4-B ids out: the 8-B keys are only materialised when `meta` is asked for gsplat's arrays."""
    P = C * H * W
    b = {}
    b["gs_isect_bin_scatter"] = Ic * 8 + V * 24          # pair stores + the visible-list entries and tile masks read
    b["gs_isect_tile_sort"] = Ic * (8 + 4)
    b["gs_rasterize_fwd"] = Wk * (28 + 4 * D) + P * (4 * D + 8 + 4)   # (+4: the exact final transmittance)
    b["gs_rasterize_bwd"] = Wk * (28 + 4 * D) + P * (4 * D + 12 + 4) + V * (4 * D + 12 + 8 + 4 + (8 if absgrad else 0))
    return b


def measured_traffic(kernel, N, C, H, W):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/pmc_traffic.py writes the
    file; FETCH_SIZE and WRITE_SIZE need separate passes, so they cannot be collected inside this run)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return None
    if [t.get("gaussians"), t.get("cams"), t.get("height"), t.get("width")] != [N, C, H, W]:
        return None
    k = t.get("kernels", {}).get(kernel)
    return int(k["hbm_bytes_per_launch"]) if k else None


def measured_issue(kernel):
    """The dominant kernel's secondary roofline from the committed SQ / TCC PMC passes (profiles/pmc_issue.py writes the
    file from profiles/collect_pmc.sh's runs of this same command): the tile rasterizer is not HBM-bound by
    construction, what bounds it is vector-ALU issue."""
    try:
        with open(os.path.join(ROOT, "profiles", "issue.json")) as f:
            k = json.load(f).get("kernels", {}).get(kernel)
    except (OSError, ValueError):
        return None
    if not k or "valu_issue_frac" not in k:
        return None
    # three readings of "how close to the vector-ALU issue limit" (profiles/pmc_issue.py states each formula):
    #   busy_counter      4 * SQ_ACTIVE_INST_VALU / (kernel cycles * 1024 SIMDs): a busy counter, reads > 1 when two
    #                     waves' instructions overlap in the pipe -- says "saturated", not how far from a roofline
    #   frac_at_2_cycles  SQ_INSTS_VALU * 2 cycles / (kernel cycles * 1024): the guide's v_fma_f32 rate, the hard ceiling
    #   frac_mix_weighted SQ_INSTS_VALU weighted by this part's measured per-class issue costs
    #                     (profiles/r02_valu_ubench.txt) over the kernel's static instruction mix
    return {"bound": "valu_issue", "frac": k.get("valu_issue_frac_mix", k["valu_issue_frac"]),
            "busy_counter": k["valu_active"], "frac_at_2_cycles": k.get("valu_issue_frac_2cyc"),
            "frac_mix_weighted": k.get("valu_issue_frac_mix"), "mix": k.get("valu_mix"),
            "waves_per_simd": k.get("waves_per_simd"), "insts_valu": k.get("insts_valu"),
            "kernel_cycles": k.get("kernel_cycles"), "vgprs": k.get("vgprs"),
            "lds_bank_conflict_rate": k.get("lds_bank_conflict_rate"), "atomic_GBps": k.get("atomic_GBps"),
            "atomic_frac_of_peak": k.get("atomic_frac_of_1300GBps"),
            "source": "profiles/issue.json (profiles/pmc_issue.py: rocprofv3 --pmc SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, "
                      "SQ_BUSY_CYCLES, SQ_LDS_BANK_CONFLICT, TCC_EA0_ATOMIC over this command; instruction mix and VGPRs "
                      "from the kernel's ISA, profiles/isa_mix.py)"}


def cpu_baseline(threads: int):
    """BASELINE.json configs[0]: the oracle (pure-PyTorch restatement of gsplat's algorithm) on host cores,
    100k Gaussians, SH degree 3, one 1242x375 camera, a few fwd+bwd steps (~12-20 s of CPU work)."""
    import oracle

    torch.set_num_threads(threads)
    N, W, H = 100_000, 1242, 375
    g = synth_scene.make_gaussians(N, sh_degree=3)
    vm, Ks = synth_scene.make_cameras(1, W, H)
    for v in g.values():
        v.requires_grad_(True)
    gen = torch.Generator().manual_seed(1)
    wc = torch.randn(1, H, W, 4, generator=gen)
    wa = torch.randn(1, H, W, 1, generator=gen)
    reps, t_all = 0, 0.0
    while reps < 8 and (reps < 2 or t_all < 12.0):  # a bounded sample: at least ~12 s of CPU work, at most 8 steps
        for v in g.values():
            v.grad = None
        t0 = time.perf_counter()
        rc, ra, _ = oracle.rasterization(g["means"], g["quats"], g["scales"], g["opacities"], g["colors"], vm, Ks, W, H,
                                         near_plane=0.1, sh_degree=3, render_mode="RGB+ED",
                                         rasterize_mode="antialiased")
        ((rc * wc).sum() + (ra * wa).sum()).backward()
        t_all += time.perf_counter() - t0
        reps += 1
    dt = t_all / reps
    return {
        "value": round(W * H / dt / 1e6, 5), "unit": "Mpix/s", "cores": threads, "kind": "port",
        "sample": f"configs[0]: 100k Gaussians, SH deg 3, one 1242x375 camera, {reps} fwd+bwd steps, mean {dt:.2f} s "
                  f"each (fp32, torch CPU, {threads} threads)",
    }


def run_render(args, dev, rank, world):
    """BASELINE configs[4]: render-only, packed, a --views trajectory at --width x --height sharded contiguously over
    the ranks (no collective on the data path: every rank holds all Gaussians and writes its own frames), frames
    converted to uint8 on the device and copied to pinned host memory behind the next batch's render
    (gsplat.harness.render_trajectory; the reference's loop is video_utils.py:647-706).  A step = one pass over the
    trajectory; value = all ranks' pixels / max-over-ranks time, D2H copy of the frames included."""
    import gsplat.harness as harness
    from gsplat.harness import render_trajectory

    for name in ("SH_ROWS", "SCATTER_WORKGROUPS_SHARED"):  # launch-shape sweeps
        if os.environ.get("GS_HARNESS_" + name):
            setattr(harness, name, int(os.environ["GS_HARNESS_" + name]))

    N, W, H, V = args.gaussians, args.width, args.height, args.views
    g = {k: v.to(dev) for k, v in synth_scene.make_gaussians(N, sh_degree=3, seed=42).items()}
    # (ego path x = -30 m ... +29.5 m: SURVEY.md 8(d)'s path from x = 0 leaves the +-40 m slab after frame 80, and a third
    # of the views would render empty space -- 2750 instead of the honest figure below; --ego-x0 0 restores it)
    cams = [synth_scene.make_cameras(1, W, H, frame=f % 120, origin_x=args.ego_x0) for f in range(V)]
    vm = torch.cat([c[0] for c in cams]).to(dev)
    Ks = torch.cat([c[1] for c in cams]).to(dev)
    out = [None]

    def step():
        out[0], views = render_trajectory(g["means"], g["quats"], g["scales"], g["opacities"], g["colors"], vm, Ks, W,
                                          H, sh_degree=3, views_per_call=args.views_per_call, rank=rank,
                                          world_size=world, near_plane=0.1, out=out[0], streams=args.render_streams)
        return views

    for _ in range(max(1, args.warmup)):
        views = step()
    # views that show (nearly) nothing, from the warm-up pass's frames: a sanity figure for the trajectory
    empty = int((out[0].reshape(out[0].shape[0], -1).float().mean(1) < 1.0).sum())
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.rehearse_gloo else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        print(json.dumps({
            "metric": "Mpix/s render-only (NOT the headline metric)", "value": round(V * W * H * args.steps / dt / 1e6, 1),
            "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": max(1, args.warmup),
            "ms_per_step": round(dt / args.steps * 1e3, 2), "views_per_s": round(V * args.steps / dt, 1),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"configs[4]: {N} Gaussians, SH deg 3, {V}-view {W}x{H} trajectory, packed, no "
                                   f"gradients, views sharded over {world} rank(s) ({len(views)} on rank 0), "
                                   f"{args.views_per_call} views per call over {args.render_streams} stream(s), uint8 frames copied to pinned host memory; ego "
                                   f"path from x = {args.ego_x0} m, {empty} of rank 0's views empty",
                       "gaussians": N, "views": V, "width": W, "height": H,
                       "parallelism": f"view sharding x{world}, no collective" if world > 1 else "single GPU"}}),
              flush=True)


class stdout_to_stderr:
    """While RCCL starts (NCCL_DEBUG=VERSION is set on the GPU boxes: it prints its version banner on stdout), send
    file descriptor 1 to stderr, so that this program's stdout carries the JSON line and nothing else."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def spawn_ranks(n: int) -> int:
    """Bare `python bench.py --gpus N` (no launcher): start the N ranks as fresh child processes of this, still
    GPU-free, process (nothing here has called into HIP: importing torch does not), one device per rank, rendezvous on
    127.0.0.1.  Returns the exit status for the parent: 0 only if every rank succeeded."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this driver
        if int(env.get("GPU_MAX_HW_QUEUES", "4") or 4) < 8:
            env["GPU_MAX_HW_QUEUES"] = "8"  # (see the top of this file)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    status = 0
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            rc = procs[r].poll()
            if rc is None:
                continue
            pending.discard(r)
            if rc != 0 and status == 0:
                status = rc if rc > 0 else 1
                print(f"bench.py: rank {r} exited with status {rc}; stopping the other ranks", file=sys.stderr)
                for o in pending:  # (the exact children started above: their collectives would wait forever)
                    procs[o].terminate()
        if pending:
            time.sleep(0.05)
    return status


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 20 (train, frames) / 2 (render)")
    ap.add_argument("--warmup", type=int, default=None, help="default 5 (train, frames) / 1 (render)")
    ap.add_argument("--mode", choices=["train", "frames", "render", "composite"], default="train",
                    help="train = the headline (BASELINE configs[1]; every rank one fixed 5-camera frame); frames = "
                         "configs[3] (--frames frames x --cams cameras sharded by frame over the ranks, a new frame every "
                         "step, densification statistics kept per step and all-reduced every --refine-interval steps); "
                         "render = configs[4] (render-only packed trajectory of --views views sharded over the ranks, "
                         "frames copied to pinned host memory; a step = one pass over the trajectory); composite = "
                         "configs[2] (3 M Gaussians as four concatenated groups with ~5 %% of the non-background opacities "
                         "zero, --cams 1920x1280 cameras (default 1), colours pre-evaluated by the standalone "
                         "spherical_harmonics op as the reference calls it, classic mode).  Only 'train' at its default "
                         "shape prints the headline metric")
    ap.add_argument("--gaussians", type=int, default=None, help="default 1M (train, frames) / 6M (render)")
    ap.add_argument("--cams", type=int, default=5)
    ap.add_argument("--width", type=int, default=None, help="default 1242 (train, frames) / 1920 (render)")
    ap.add_argument("--height", type=int, default=None, help="default 375 (train, frames) / 1080 (render)")
    ap.add_argument("--frames", type=int, default=120)
    ap.add_argument("--views", type=int, default=600)
    ap.add_argument("--views-per-call", type=int, default=8)
    ap.add_argument("--ego-x0", type=float, default=-30.0,
                    help="frames / render: x of the ego at frame 0 (the scene spans +-40 m, a 120-frame path is 60 m long)")
    ap.add_argument("--refine-interval", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=4)
    ap.add_argument("--render-streams", type=int, default=2,
                    help="--mode render: HIP streams that successive batches of views alternate over")
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="multi-rank dry run on fewer GPUs than ranks: gloo backend, device = LOCAL_RANK %% device_count "
                         "(exercises the N > 1 code path on a one-GPU box; the number it prints is not a result)")
    ap.add_argument("--exchange", choices=["auto", "colour", "dense"], default="auto",
                    help="N > 1: how the parameter gradients cross ranks (gsplat.distributed.GradientExchange): 'colour' "
                         "(= auto) all-gathers the packed 12-B colour gradients of the visible pairs and rebuilds v_sh on "
                         "every rank; 'dense' = one SUM all-reduce per parameter tensor")
    ap.add_argument("--sh-backward", choices=["default", "fused", "split"], default="default",
                    help="single-GPU route of the in-rasterization SH backward (gsplat.cuda._wrapper.SPLIT_SH_BACKWARD)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="N = 1 only: run the N-rank exchange path (split SH backward, packed messages, every collective) "
                         "over a one-rank RCCL group -- what a rank of the N-GPU run computes, without the wire time")
    ap.add_argument("--no-cull", action="store_true",
                    help="A/B: gsplat's full tile lists instead of the culled ones (gsplat.cuda._wrapper.CULL_LISTS)")
    ap.add_argument("--dump-stacks-after", type=float, default=0.0,
                    help="debugging: print every thread's Python stack to stderr after this many seconds (a hung run)")
    args = ap.parse_args()
    render = args.mode == "render"
    composite = args.mode == "composite"
    if composite:
        if args.cams == 5 and "--cams" not in sys.argv:
            args.cams = 1
        for name, val in (("gaussians", 2_600_000 + 250_000 + 100_000 + 7 * 6890), ("width", 1920), ("height", 1280)):
            if getattr(args, name) is None:
                setattr(args, name, val)
    for name, train_default, render_default in (("steps", 20, 2), ("warmup", 5, 1), ("gaussians", 1_000_000, 6_000_000),
                                                ("width", 1242, 1920), ("height", 375, 1080)):
        if getattr(args, name) is None:
            setattr(args, name, render_default if render else train_default)
    if args.dump_stacks_after > 0:
        import faulthandler

        faulthandler.dump_traceback_later(args.dump_stacks_after, repeat=False, file=sys.stderr)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product path has no CPU fallback)")
    if args.rehearse_gloo:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.rehearse_gloo:
            dist.init_process_group("gloo")
        else:
            with stdout_to_stderr():
                dist.init_process_group("nccl", device_id=dev)
                dist.barrier()  # (communicator creation, and with it the banner, may be deferred to the first collective)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.force_exchange and world == 1:
        import socket

        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(s.getsockname()[1]))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        with stdout_to_stderr():
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            dist.barrier()

    from gsplat import _backend as B
    from gsplat.cuda import _wrapper as Wm
    from gsplat.distributed import GradientExchange
    from gsplat.rendering import rasterization

    if args.no_cull:
        Wm.CULL_LISTS = False
    for name in ("SLICES_TOTAL", "SCATTER_REGIONS", "SCATTER_WORKGROUPS", "STAGED_SCATTER", "FRONT_FROM_CLASS", "POSTED_RESULTS", "FRONT_SORT", "SPECULATE_SIZES"):  # launch-shape sweeps (scratch/)
        if os.environ.get("GS_" + name):
            setattr(Wm, name, int(os.environ["GS_" + name]))
    if render:
        run_render(args, dev, rank, world)
        if world > 1:
            dist.destroy_process_group()
        return
    if args.sh_backward != "default":
        Wm.SPLIT_SH_BACKWARD = args.sh_backward == "split"
    exchange = GradientExchange(mode=args.exchange, force=args.force_exchange)  # (a no-op context on one rank)

    N, C, W, H, K, D = args.gaussians, args.cams, args.width, args.height, 16, 4
    headline_shape = (N, C, W, H) == (1_000_000, 5, 1242, 375) and args.mode == "train"
    g = synth_scene.make_gaussians(N, sh_degree=3, seed=42)
    if composite:
        # rows past the 2.6 M background are node Gaussians; ~5 % of them belong to instances
        # that are invalid in this frame and arrive with opacity exactly 0 (rigid.py:493)
        zero = torch.rand(N, generator=torch.Generator().manual_seed(3)) < 0.05
        zero[: min(N, 2_600_000)] = False
        g["opacities"][zero] = 0.0
    params = {k: v.to(dev).requires_grad_(True) for k, v in g.items()}
    if args.mode == "frames":
        # configs[3]: the frames of the sequence shard contiguously over the ranks; every step renders the next frame
        # of this rank's shard (all its cameras), so camera poses -- and with them visibility, list lengths and the
        # exchange's message -- change from step to step as they do in training
        from gsplat.densify import DensifyState, accumulate_stats
        from gsplat.distributed import all_reduce_densification_stats, shard_views

        mine = shard_views(args.frames, rank, world)
        assert len(mine) > 0, "more ranks than frames"
        cams = [synth_scene.make_cameras(C, W, H, frame=f, origin_x=args.ego_x0) for f in mine]
        cams = [(a.to(dev), b.to(dev)) for a, b in cams]
        dstate = DensifyState()
    else:
        # weak scaling: rank r renders frame r of the 5-camera rig (ego advances 0.5 m per frame)
        cams = [tuple(t.to(dev) for t in synth_scene.make_cameras(C, W, H, frame=rank))]
    n_steps = [0]
    cam_centre = Wm._camera_centers(cams[0][0]) if composite else None
    gen = torch.Generator().manual_seed(1 + rank)
    wc = torch.randn(C, H, W, D, generator=gen).to(dev)
    wa = torch.randn(C, H, W, 1, generator=gen).to(dev)
    wc_flat, wa_flat = wc.reshape(-1), wa.reshape(-1)
    plist = list(params.values())

    def note(msg):
        if args.rehearse_gloo:
            print(f"[rank {rank}] {msg}", file=sys.stderr, flush=True)

    def step():
        for p in plist:
            p.grad = None
        vm, Ks = cams[n_steps[0] % len(cams)]
        n_steps[0] += 1
        if composite:
            # the reference's calling pattern (vanilla.py:369-386): view directions from the first camera, the standalone
            # SH op, +0.5 / clamp, then rasterization with [N,3] colours
            dirs = (params["means"] - cam_centre[0]).detach()
            rgb = torch.clamp(Wm.spherical_harmonics(3, dirs, params["colors"]) + 0.5, 0.0, 1.0)
            with exchange:
                rc, ra, meta = rasterization(
                    params["means"], params["quats"], params["scales"], params["opacities"], rgb, vm, Ks, W, H,
                    near_plane=0.1, far_plane=1e10, packed=False, render_mode="RGB+ED", absgrad=True,
                    rasterize_mode="classic")
        else:
            with exchange:
                rc, ra, meta = rasterization(
                    params["means"], params["quats"], params["scales"], params["opacities"], params["colors"], vm, Ks,
                    W, H, near_plane=0.1, far_plane=1e10, sh_degree=3, packed=False, render_mode="RGB+ED",
                    absgrad=True, rasterize_mode="antialiased")
        # fixed random linear functional of colours and alphas, as two dot products
        loss = torch.dot(rc.reshape(-1), wc_flat) + torch.dot(ra.reshape(-1), wa_flat)
        # N > 1: the gradient exchange runs inside this backward (packed colour gradients all-gathered behind the
        # geometry kernel, geometry gradients all-reduced while the SH kernel runs); afterwards .grad is the gradient
        # of all ranks' cameras, on every rank
        loss.backward()
        note("backward + gradient exchange queued")
        if args.mode == "frames":
            # VanillaGaussians.after_train every step (vanilla.py:154-182), its three statistics summed / maxed over
            # the ranks when the refinement is due (every rank then takes the same split / duplicate / cull decisions)
            accumulate_stats(meta, dstate, absgrad=True)
            if n_steps[0] % args.refine_interval == 0:
                all_reduce_densification_stats(dstate.xys_grad_norm, dstate.vis_counts, dstate.max_2Dsize)
        return meta

    note("first step")
    for _ in range(args.warmup):
        meta = step()
        note("warm-up step done")
    if args.warmup == 0:
        meta = step()
    V = int((meta["radii"] > 0).sum().item())
    # gsplat's intersection count = the sum of the tile-box counts (what SURVEY.md 8(d)'s formulas are written in); the
    # culled lists the rasterizer runs on are shorter, and asking `meta` for flatten_ids would build the full ones
    I = int(meta["tiles_per_gauss"].sum().item())
    lists = getattr(meta, "_lists", None)
    Ic = int(lists[3].numel()) if lists is not None else I      # entries of the lists the rasterizer ran on
    work = getattr(Wm._tls, "last_tile_work", None)
    Wk = int(work.sum().item()) if work is not None else Ic     # ... and how many of them the compositor walks
    del meta, lists, work

    def barrier():
        if world > 1:
            dist.barrier()

    # untimed pre-pass: HIP-event brackets around every native entry point -> per-kernel breakdown and the dominant one
    B.profiling_start()
    for _ in range(3):
        step()
    kern_all = B.profiling_stop()
    dom = max(kern_all, key=lambda k: kern_all[k][1] * kern_all[k][0]) if kern_all else None

    # timed region: the production stream schedule; only the dominant kernel stays bracketed (two events per step on
    # its launch stream), plus one event per step boundary for the median step time of SURVEY.md section 8(d)
    B.profiling_start(only=[dom] if dom else [])
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        marks[i].record()
        step()
    marks[args.steps].record()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    kern = B.profiling_stop()
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    median_ms = step_ms[len(step_ms) // 2] if step_ms else float("nan")
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        if args.rehearse_gloo:
            t = t.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * C * H * W * args.steps / dt / 1e6
        bytes_ = algorithmic_bytes(N, C, H, W, K, D, V, I)
        proc_ = processed_bytes(N, C, H, W, K, D, V, Ic, Wk)
        kern_ms = {k: round(v[1], 4) for k, v in kern_all.items()}
        kern_calls = {k: v[0] / 3 for k, v in kern_all.items()}
        roofline = None
        if dom is not None and dom in bytes_:
            per_launch_ms = kern[dom][1]
            ach = bytes_[dom] / (per_launch_ms * 1e-3) / 1e9
            roofline = {"kernel": dom, "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": measured_traffic(dom, N, C, H, W),
                        "algorithmic_bytes_per_launch": bytes_[dom], "avg_launch_ms": round(per_launch_ms, 4),
                        # two clocks on the same kernel, both HIP events on its launch stream: `avg_launch_ms` is taken
                        # inside the timed region (production schedule: only this kernel bracketed); the pre-pass
                        # brackets EVERY entry point, which serialises the side stream's fill behind the forward and
                        # leaves more dirty lines for this kernel to wait on.  rocprofv3's average
                        # (profiles/*kernel_stats.csv) is of the pre-pass kind.  `frac` uses the first.
                        "avg_launch_ms_prepass_all_bracketed": round(kern_all[dom][1], 4),
                        "frac_prepass": round(bytes_[dom] / (kern_all[dom][1] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                        "processed_bytes_per_launch": proc_.get(dom),
                        "frac_processed": (round(proc_[dom] / (per_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
                                           if dom in proc_ else None),
                        "secondary": measured_issue(dom) if headline_shape else None}
        # the same figure for every stage that has algorithmic bytes (pre-pass launch averages; the sort chain and the
        # redo launches are folded into their stage's entry point)
        per_kernel = {}
        for k, nbytes in bytes_.items():
            # (the split SH backward's second kernel belongs to the projection-backward stage)
            ms_k = sum(v[1] * v[0] / 3 for kk, v in kern_all.items() if kk == k or kk.startswith(k + "_")
                       or (k == "gs_projection_bwd" and kk == "gs_sh_bwd_packed"))
            if ms_k > 0:
                gbs = nbytes / (ms_k * 1e-3) / 1e9
                per_kernel[k] = {"ms": round(ms_k, 4), "algorithmic_bytes": nbytes, "GB/s": round(gbs, 1),
                                 "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)}
                if k in proc_:  # reference-formula bytes above; the bytes of the lists actually processed here
                    per_kernel[k]["processed_bytes"] = proc_[k]
                    per_kernel[k]["processed_GB/s"] = round(proc_[k] / (ms_k * 1e-3) / 1e9, 1)
        total_bytes = sum(bytes_.values())
        headline = headline_shape and args.mode == "train"
        workload = (("configs[1]: " if headline else "NOT the headline: ") +
                    f"{N} Gaussians, SH deg 3, {C} cams {W}x{H} batched fwd+bwd, RGB+ED, absgrad + antialiased, "
                    "packed=False" +
                    (f"; configs[3]: {args.frames} frames sharded by frame over {world} rank(s), a new frame every step, "
                     f"densification statistics every step, all-reduced every {args.refine_interval} steps"
                     if args.mode == "frames" else "") +
                    ("; configs[2]: composite scene of four concatenated groups, ~5 % of the node opacities zero, colours "
                     "from the standalone spherical_harmonics op (+0.5, clamp) as the reference calls it, classic mode "
                     "(the workload string's 'SH deg 3 ... antialiased' describes the headline only)"
                     if composite else ""))
        out = {
            "metric": "Mpix/s fwd+bwd, 1M Gaussians @375x1242x5 cams" if headline else "Mpix/s fwd+bwd", "value": round(value, 3), "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
            "ms_per_step_median_hip_events": round(median_ms, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "gaussians": N, "cams_per_gpu": C,
                       "width": W, "height": H, "visible_pairs_V": V, "n_isects_I": I,
                       "tile_list_entries_culled": Ic, "tile_list_entries_walked": Wk,
                       "parallelism": (f"camera/frame DP x{world}, gradient exchange '{exchange.mode}' "
                                       f"({'gloo REHEARSAL, not a result' if args.rehearse_gloo else 'RCCL'}), "
                                       f"{exchange.bytes_sent // max(1, args.steps + args.warmup + 3 + (args.warmup == 0))} "
                                       "payload bytes contributed per rank and step")
                       if world > 1 else ("single GPU" + (
                           f", exchange path '{exchange.mode}' forced over a one-rank RCCL group (NOT the headline "
                           "schedule)" if args.force_exchange else "")),
                       "tile_lists": "culled (alpha < 1/255 over the whole tile left out)" if Wm.CULL_LISTS else "gsplat's",
                       "sh_backward": "split" if (Wm.SPLIT_SH_BACKWARD or (exchange.active and exchange.packs_colours))
                       else "fused"},
            "roofline": roofline,
            "step_hbm_floor": {"algorithmic_bytes_per_step": total_bytes,
                               "frac_of_hbm_peak": round(total_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
            "kernels_ms": kern_ms, "kernel_calls_per_step": kern_calls, "kernel_rooflines": per_kernel,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_threads)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
