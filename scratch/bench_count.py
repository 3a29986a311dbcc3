"""Times gs_isect_bin_count alone at the headline (or composite) shape: scratch/bench_scatter.py [composite]
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
def count():
    B.check(lib.gs_isect_bin_count(C, N, TW, TH, B.ptr(splats), B.ptr(tpg), n_slices, B.ptr(table), B.ptr(vis_list),
                                   B.ptr(vis_cnt), B.ptr(vis_masks), 16, st), "count")
for _ in range(3):
    count()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
ev[0].record()
for i in range(20):
    count()
    ev[i + 1].record()
torch.cuda.synchronize()
ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(20))
print(f"{os.environ.get('GSPLAT_MI355X_LIB', 'base').split('/')[-1]:24s} slices={n_slices} V={int(vis_cnt.sum())}: "
      f"count median {ts[10] * 1e3:.1f} us  min {ts[0] * 1e3:.1f} us", flush=True)
