#!/usr/bin/env python3
"""Turns two rocprofv3 PMC passes over bench.py (one with FETCH_SIZE, one with WRITE_SIZE) into profiles/traffic.json,
the per-launch HBM bytes bench.py reports as roofline.traffic.

    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_rd -o p --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_wr -o p --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline
    python profiles/pmc_traffic.py gpurun_out/pmc_rd gpurun_out/pmc_wr > profiles/traffic.json

Units and corrections (/opt/skills/guides/MI355X_MICROARCH.md, "HBM"): both counters are in KiB; on gfx950 FETCH_SIZE
tallies every fabric read request as 64 B whether it fetches 64 B or 128 B, so wide coalesced reads (128-B requests)
show exactly half their bytes -> doubled here, while a kernel whose reads are gathers of single 64-B records issues
64-B requests and its RAW counter is already exact (profiles/r02_fetch_calibration.txt: scratch/ubench/gather64.hip,
kernels of known byte count -- streaming 1 GiB reads 0.5 GiB raw; gathers of 64-B records, whole or 48 B or 16 B of
each, read 1.00 GiB raw = the lines touched).  The two rasterizer kernels are gather-dominated (the splat records they
read by id; forward 94 % of its read bytes): for them `read_bytes_per_launch` is the raw count plus the known coalesced
streams counted once more (ids, and in the backward the per-pixel inputs), and `read_bytes_upper` the doubled value.
WRITE_SIZE is exact for streaming stores and float atomics.  The x2 read correction is checked in the same pass against a
kernel of known byte count (rocBLAS dot over the loss weights: 2 x 4 x C*H*W*D bytes) and the write unit against the
torch fill of the gradient record (C*N*64 bytes); both ratios are printed into the JSON as `calibration`.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

KERNELS = {  # bench.py entry-point label -> substring(s) of the HIP kernel names it launches
    "gs_projection_fwd": ["gs::projection_fwd_kernel"],
    "gs_projection_bwd": ["gs::projection_bwd_kernel", "gs::sh_bwd_packed_kernel"],
    "gs_isect_bin_count": ["gs::isect_bin_kernel<false, true>"],
    "gs_isect_bin_scatter": ["gs::isect_region_scatter_staged_kernel<true>"],
    "gs_isect_bin_offsets": ["gs::bin_slice_prefix_kernel", "gs::bin_offsets_kernel"],
    "gs_isect_tile_sort": ["gs::tile_sort_kernel", "gs::tile_sort_front_kernel"],
    "gs_rasterize_fwd": ["gs::rasterize_fwd_kernel"],
    "gs_rasterize_bwd": ["gs::rasterize_bwd_kernel"],
}


# entry points whose reads are mostly per-id gathers of 64-B splat records: label -> coalesced bytes per step that the
# raw counter shows at half (filled in from the shape below)
GATHER_DOMINATED = ("gs_rasterize_fwd", "gs_rasterize_bwd", "gs_isect_bin_count")


def per_kernel(directory, counter):
    """{kernel name: (launches, mean counter value per launch)} from every counter_collection csv under directory."""
    tot, cnt = defaultdict(float), defaultdict(int)
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter:
                    continue
                tot[row["Kernel_Name"]] += float(row["Counter_Value"])
                cnt[row["Kernel_Name"]] += 1
    # per STEP, not per launch: an entry point may launch a kernel more than once per step (size classes of the tile
    # sort, the rasterizer's redo pass), and the steps of the run are the launches of the once-per-step projection
    steps = max([c for k, c in cnt.items() if "gs::projection_fwd_kernel" in k] or [1])
    return {k: (cnt[k], tot[k] / (steps if "gs::" in k else cnt[k])) for k in tot}


def main():
    rd_dir, wr_dir = sys.argv[1], sys.argv[2]
    N, C, H, W, D = 1_000_000, 5, 375, 1242, 4
    rd = per_kernel(rd_dir, "FETCH_SIZE")
    wr = per_kernel(wr_dir, "WRITE_SIZE")
    out = {"gaussians": N, "cams": C, "height": H, "width": W,
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py, mean per step of the entry point's launches; "
                     "FETCH_SIZE KiB x 1024 x 2 (gfx950 half-count correction), WRITE_SIZE KiB x 1024",
           "kernels": {}, "calibration": {}}
    for label, subs in KERNELS.items():
        fetch = sum(v[1] for k, v in rd.items() if any(s in k for s in subs))
        write = sum(v[1] for k, v in wr.items() if any(s in k for s in subs))
        launches = max([v[0] for k, v in rd.items() if any(s in k for s in subs)] or [0])
        read = fetch * 1024 * 2
        extra = {}
        if label in GATHER_DOMINATED:
            # coalesced streams of the kernel (128-B requests, shown at half): the backward's per-pixel inputs
            # (render_alphas, final_T, last_ids, v_colors [D], v_alphas); the id lists are a few % and left in the raw
            # (count pass: the [C,N] tile counts stream; everything else it reads is one 64-B record line per visible row,
            # touched for the box and again, a few steps later, for the conic)
            stream = {"gs_rasterize_bwd": C * H * W * 4 * (D + 4), "gs_isect_bin_count": C * N * 4}.get(label, 0)
            read = fetch * 1024 + stream / 2
            extra = {"read_bytes_upper": int(fetch * 1024 * 2),
                     "read_note": "gather-dominated: raw FETCH_SIZE (64-B requests are counted exactly) + half of the "
                                  "known coalesced per-pixel streams; read_bytes_upper = the doubled counter"}
        out["kernels"][label] = {
            "fetch_size_kib_raw": round(fetch, 1), "write_size_kib_raw": round(write, 1),
            "read_bytes_per_launch": int(read), "write_bytes_per_launch": int(write * 1024),
            "hbm_bytes_per_launch": int(read + write * 1024), "launches_seen": launches, **extra,
        }
    for k, v in rd.items():
        if "rocblas_dot_kernel" in k:  # reads two [C,H,W,D] / [C,H,W,1] fp32 vectors per call, alternately
            expect = (2 * 4 * C * H * W * D + 2 * 4 * C * H * W) / 2
            out["calibration"]["read: rocblas dot, corrected/expected"] = round(v[1] * 2048 / expect, 3)
    for k, v in wr.items():
        if "fill_zero_nt_kernel" in k:  # the gradient record's fill (nontemporal stores): C*N*64 bytes per launch
            out["calibration"]["write: gradient-record fill, counted/expected"] = round(v[1] * 1024 / (C * N * 64), 3)
        if "fillBufferAligned" in k:  # (builds before the nontemporal fill: hipMemsetAsync, + a few small fills)
            steps = max([c for kk, (c, _) in wr.items() if "gs::projection_fwd_kernel" in kk] or [1])
            out["calibration"]["write: gradient-record memset, counted/expected"] = round(
                v[1] * v[0] / steps * 1024 / (C * N * 64), 3)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
