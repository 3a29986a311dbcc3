"""Times gs_isect_bin_scatter_regions alone at the headline (or composite) shape: scratch/bench_scatter.py [composite]
env: GS_STAGED=0/1, GS_REGIONS, GS_WORKGROUPS, GS_SLICES, GSPLAT_MI355X_LIB (experimental builds: results are NOT checked)"""
import math, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "3d-reconstruction_amd"))
import synth_scene
from gsplat.cuda import _wrapper as Wm
from gsplat import _backend as B

dev = torch.device("cuda:0")
composite = len(sys.argv) > 1 and sys.argv[1] == "composite"
N, C, W, H = (3_000_000, 1, 1920, 1280) if composite else (1_000_000, 5, 1242, 375)
TW, TH = math.ceil(W / 16), math.ceil(H / 16)
T = TW * TH
g = synth_scene.make_gaussians(N, sh_degree=None, seed=42)
vm, Ks = synth_scene.make_cameras(C, W, H)
g = {k: v.to(dev) for k, v in g.items()}
params = Wm._make_params(N, C, W, H, 16, 0.3, 0.1, 1e10, 0.0, True, B.GS_COLOR_N, 3, None, 0, True)
with torch.no_grad():
    splats, means2d, radii, tpg = Wm._ProjectSplats.apply(g["means"], g["quats"], g["scales"], g["opacities"], g["colors"],
                                                          vm.to(dev), Ks.to(dev), None, params, None)
lib, st = B.lib(), B.stream_handle(dev)
slices_total = int(os.environ.get("GS_SLICES", Wm.SLICES_TOTAL))
n_slices = max(1, min(max(slices_total // C, 1), -(-N // 4096)))
regions = min(int(os.environ.get("GS_REGIONS", Wm.SCATTER_REGIONS)), TH)
group = max(1, -(-(n_slices * C * regions) // int(os.environ.get("GS_WORKGROUPS", Wm.SCATTER_WORKGROUPS))))
staged = os.environ.get("GS_STAGED", "1") == "1"
table = torch.empty(C * n_slices * T, device=dev, dtype=torch.int32)
counts = torch.empty(C * T, device=dev, dtype=torch.int32)
offsets = torch.empty(C, TH, TW, device=dev, dtype=torch.int32)
total = torch.empty(1, device=dev, dtype=torch.int64)
seg = int(lib.gs_isect_bin_slice_rows(N, n_slices))
vis_list = torch.empty(C * n_slices * seg, 2, device=dev, dtype=torch.int64)
vis_cnt = torch.empty(C * n_slices, device=dev, dtype=torch.int32)
vis_masks = torch.empty(C * n_slices * seg, device=dev, dtype=torch.int64)
B.check(lib.gs_isect_bin_count(C, N, TW, TH, B.ptr(splats), B.ptr(tpg), n_slices, B.ptr(table), B.ptr(vis_list),
                               B.ptr(vis_cnt), B.ptr(vis_masks), 16, st), "count")
B.check(lib.gs_isect_bin_offsets(C, TW, TH, n_slices, B.ptr(table), B.ptr(counts), B.ptr(offsets), B.ptr(total), st), "offsets")
I = int(total.item())
pairs = torch.empty(I, device=dev, dtype=torch.int64)
def scatter():
    B.check(lib.gs_isect_bin_scatter_regions(C, N, TW, TH, n_slices, group, regions, B.ptr(vis_list), B.ptr(vis_cnt),
                                             B.ptr(vis_masks), B.ptr(table), B.ptr(offsets),
                                             B.ptr(counts) if staged else None, B.ptr(pairs), st), "scatter")
for _ in range(3):
    scatter()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
ev[0].record()
for i in range(20):
    scatter()
    ev[i + 1].record()
torch.cuda.synchronize()
ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(20))
print(f"{os.environ.get('GSPLAT_MI355X_LIB', 'base').split('/')[-1]:24s} staged={int(staged)} slices={n_slices} regions={regions} "
      f"group={group} V={int(vis_cnt.sum())} I={I}: scatter median {ts[10] * 1e3:.1f} us  min {ts[0] * 1e3:.1f} us", flush=True)
if hasattr(lib, "gs_debug_scatter_read") or "sc4" in os.environ.get("GSPLAT_MI355X_LIB", ""):
    import ctypes, numpy as np
    raw = ctypes.CDLL(os.environ["GSPLAT_MI355X_LIB"])
    buf = np.zeros(8192 * 8, dtype=np.uint64)
    cnt = np.zeros(3, dtype=np.uint32)
    raw.gs_debug_scatter_read(ctypes.c_void_p(buf.ctypes.data), ctypes.c_void_p(cnt.ctypes.data))  # reset counters
    scatter(); torch.cuda.synchronize()
    raw.gs_debug_scatter_read(ctypes.c_void_p(buf.ctypes.data), ctypes.c_void_p(cnt.ctypes.data))
    nwg = -(-n_slices // group) * regions * C
    d = buf[: nwg * 8].reshape(nwg, 8).astype(np.int64)
    t0 = d[:, 0].min()
    start, init, w0end, end, band = (d[:, 0] - t0) / 100.0, (d[:, 1] - d[:, 0]) / 100.0, (d[:, 2] - d[:, 1]) / 100.0, (d[:, 3] - t0) / 100.0, d[:, 4]
    print(f"workgroups {nwg}; start us: min {start.min():.2f} max {start.max():.2f}; init us: mean {init.mean():.2f} max {init.max():.2f}; "
          f"wave0 walk us: mean {w0end.mean():.2f} max {w0end.max():.2f}; end us: mean {end.mean():.2f} max {end.max():.2f}")
    for b in range(regions):
        sel = band == b
        print(f"  band {b}: walk(wave 0) mean {w0end[sel].mean():.2f} max {w0end[sel].max():.2f} us; wg duration mean {(end - start)[sel].mean():.2f} max {(end - start)[sel].max():.2f}")
    print(f"  batches {cnt[2]}, small-loop wave trips {cnt[0]} ({cnt[0] / max(cnt[2], 1):.2f} per batch), big entries {cnt[1]} ({cnt[1] / max(cnt[2], 1):.2f} per batch)")
