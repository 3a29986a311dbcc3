"""Experimental tile orders: is the array a permutation (and camera-major)?  Runs the order kernel alone."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, os.path.join(ROOT, "3d-reconstruction_amd")):
    sys.path.insert(0, p_)
import torch
from gsplat import _backend as B
dev = torch.device("cuda:0")
T, C = 1872, 5
n = T * C
for seed in range(5):
    g = torch.Generator().manual_seed(seed)
    work = torch.randint(0, 900 if seed else 1, (n,), generator=g, dtype=torch.int32).to(dev)
    offsets = torch.zeros(n, dtype=torch.int32, device=dev)
    order = torch.full((n,), -1, dtype=torch.int32, device=dev)
    B.check(B.lib().gs_raster_tile_order(n, B.ptr(offsets), 0, B.ptr(work), B.ptr(order), None, B.stream_handle(dev)), "order")
    torch.cuda.synchronize()
    o = order.cpu()
    assert sorted(o.tolist()) == list(range(n)), f"seed {seed}: not a permutation"
    cams = (o // T)
    print("seed", seed, "permutation ok; camera-major:", bool((cams[1:] >= cams[:-1]).all()))
print("ORDER OK")
