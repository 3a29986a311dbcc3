"""Front-first tile sort (gs_isect_tile_sort_front / _redo, no gsplat counterpart): long tile lists get only their leading
entries sorted before the rasterizer runs.  Observable results must not change: the sorted prefix equals the same
positions of the full A.3 order bit for bit, meta["isect_ids"/"flatten_ids"] read through `meta` are the full sorted
arrays, images are bit-identical to the whole-list sort, and a tile whose pixels outlive the front is redone."""
import numpy as np
import pytest
import torch

import synth_scene
from test_gpu_parity import _synthetic_records

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("C,N,TW,TH,hot,levels", [
    (2, 20_000, 13, 7, 6000, 1 << 20),  # long segments, spread depths: fronts of ~768
    (2, 20_000, 13, 7, 6000, 5),        # 5 distinct depths: every front has a crowded bucket -> deferred, sorted whole
    (1, 3_000, 5, 4, 1500, 1 << 20),
    (3, 500, 78, 24, 0, 3),             # nothing longer than 1024: the short classes only
    (1, 9_000, 128, 128, 0, 1 << 20),   # 16384 tiles per camera: the largest LDS histogram the binned route takes
])
@pytest.mark.parametrize("from_class", [0, 1])
def test_front_prefix_matches_full_order(gpu_device, C, N, TW, TH, hot, levels, from_class):
    import gsplat.cuda._wrapper as W
    from gsplat.cuda._wrapper import FRONT_K_TARGET, isect_tiles_binned

    rng = np.random.default_rng(7 + C * 1000 + N)
    rec, tiles, keys, ids, offsets = _synthetic_records(rng, C, N, TW, TH, hot, levels, gpu_device)
    saved = W.FRONT_FROM_CLASS
    W.FRONT_FROM_CLASS = from_class  # the front-first sort is used from a list above 2048 (0) / 1024 (1) entries on
    try:
        k, f, o, order, fs = isect_tiles_binned(rec, tiles, TW, TH, front=True)
    finally:
        W.FRONT_FROM_CLASS = saved
    assert np.array_equal(o.cpu().numpy(), offsets)
    off = offsets.reshape(-1).astype(np.int64)
    end = np.append(off[1:], len(keys))
    n = end - off
    limit = 2048 if from_class == 0 else 1024
    if fs is None:
        # no list above the limit: the host (which reads the size classes with n_isects) sorts every list whole and
        # launches neither the front kernel nor the redo pass
        assert np.all(n <= limit)
        assert np.array_equal(k.cpu().numpy(), keys) and np.array_equal(f.cpu().numpy(), ids)
        return
    assert np.any(n > limit)
    sc = fs.sorted_counts.cpu().numpy().astype(np.int64)
    assert np.all(sc >= 0) and np.all(sc <= n)
    assert np.all(sc[n <= 1024] == n[n <= 1024])          # short lists are sorted whole
    assert np.all(sc[n > 1024] >= min(FRONT_K_TARGET, 1024))  # a front holds at least k_target entries
    kg, fg = k.cpu().numpy(), f.cpu().numpy()
    for t in range(len(off)):
        a, m = off[t], sc[t]
        assert np.array_equal(kg[a:a + m], keys[a:a + m]), t
        assert np.array_equal(fg[a:a + m], ids[a:a + m]), t
    # flag every other long tile, redo: those become whole; then finalize: everything is
    long_tiles = np.nonzero(sc < n)[0]
    flags = np.zeros(len(off), np.int32)
    flags[long_tiles[::2]] = 1
    fs.redo_flags.copy_(torch.from_numpy(np.append(flags, flags.sum()).astype(np.int32)).to(gpu_device))
    fs.redo()
    sc2 = fs.sorted_counts.cpu().numpy().astype(np.int64)
    assert np.all(sc2[flags == 1] == n[flags == 1]) and np.all(sc2[flags == 0] == sc[flags == 0])
    kg, fg = k.cpu().numpy(), f.cpu().numpy()
    for t in long_tiles[::2]:
        assert np.array_equal(kg[off[t]:end[t]], keys[off[t]:end[t]]) and np.array_equal(fg[off[t]:end[t]], ids[off[t]:end[t]])
    fs.finalize()
    assert np.array_equal(k.cpu().numpy(), keys) and np.array_equal(f.cpu().numpy(), ids)
    # the slice scatter (no visible lists) fills the segments in another order; the sorted result is the same
    W.REGION_SCATTER = False
    try:
        k2, f2, o2, _, _ = isect_tiles_binned(rec, tiles, TW, TH, front=False)
    finally:
        W.REGION_SCATTER = True
    assert torch.equal(k2, k) and torch.equal(f2, f) and torch.equal(o2, o)


def _dense_scene(n, opac_lo, opac_hi, seed=3):
    g = synth_scene.make_gaussians(n, sh_degree=None, seed=seed, extent=6.0)
    gen = torch.Generator().manual_seed(seed + 1)
    g["means"][:, 2] = torch.rand(n, generator=gen) * 3.0 + 0.2
    g["scales"] = g["scales"] * 1.5
    g["opacities"] = opac_lo + (opac_hi - opac_lo) * torch.rand(n, generator=gen)
    return g


@pytest.mark.parametrize("opac,expect_redo", [((0.3, 0.95), False), ((0.002, 0.01), True)])
def test_front_sort_renders_identically(gpu_device, opac, expect_redo):
    """End to end: front-first sort (+ redo pass for tiles whose pixels outlive the front) against sorting every list
    whole.  Forward outputs and the lazily completed isect arrays bit for bit; gradients to fp32 atomic-order noise."""
    import gsplat.cuda._wrapper as W
    from gsplat.rendering import rasterization

    dev = gpu_device
    N, Wd, Hd = 120_000, 160, 96
    g = _dense_scene(N, *opac)
    vm, Ks = synth_scene.make_cameras(2, Wd, Hd)
    gen = torch.Generator().manual_seed(11)
    wc = torch.randn(2, Hd, Wd, 4, generator=gen).to(dev)
    wa = torch.randn(2, Hd, Wd, 1, generator=gen).to(dev)

    def run(front):
        old = (W.FRONT_SORT, W.FRONT_FROM_CLASS)
        W.FRONT_SORT = front
        W.FRONT_FROM_CLASS = 1  # front-first from 1024 entries on (the culled lists of this scene stay under 2048)
        try:
            p = {k: v.to(dev).requires_grad_(True) for k, v in g.items()}
            rc, ra, meta = rasterization(p["means"], p["quats"], p["scales"], p["opacities"], p["colors"], vm.to(dev),
                                         Ks.to(dev), Wd, Hd, near_plane=0.1, packed=False, render_mode="RGB+ED",
                                         absgrad=True, rasterize_mode="antialiased")
            fs = meta._lists[5]  # the lists the rasterizer ran on (culled; meta[...] hands out gsplat's exact arrays)
            stats = None
            if fs is not None:
                off = meta._lists[1].reshape(-1).long()
                n = torch.diff(torch.cat([off, torch.tensor([fs.isect_ids.numel()], device=dev)]))
                stats = (int((fs.sorted_counts < n).sum()), int(fs.redo_flags[:-1].sum()), int((n > 1024).sum()))
            ((rc * wc).sum() + (ra * wa).sum()).backward()
            grads = {k: v.grad.clone() for k, v in p.items()}
            grads["absgrad"] = meta["means2d"].absgrad.clone()
            return rc, ra, meta, grads, stats
        finally:
            W.FRONT_SORT, W.FRONT_FROM_CLASS = old

    rc1, ra1, m1, g1, st = run(True)
    rc0, ra0, m0, g0, st0 = run(False)
    assert st0 is None and st is not None
    n_front, n_redo, n_long = st
    print(f"[front] long tiles {n_long}, front-sorted {n_front}, redone {n_redo}")
    # (a redone tile counts as whole again, so "front-sorted" is what is still partial after the redo pass)
    assert n_long > 0 and n_front + n_redo > 0, "the scene does not exercise the front sort"
    assert (n_redo > 0) == expect_redo, (n_redo, expect_redo)
    assert torch.equal(rc1, rc0) and torch.equal(ra1, ra0)
    assert torch.equal(m1["last_ids"], m0["last_ids"])
    assert torch.equal(m1["isect_offsets"], m0["isect_offsets"])
    assert torch.equal(m1["flatten_ids"], m0["flatten_ids"])  # completes the tails on access
    assert torch.equal(m1["isect_ids"], m0["isect_ids"])
    for k in g0:
        a, b = g1[k].double(), g0[k].double()
        rel = float((a - b).norm() / (b.norm() + 1e-30))
        assert rel < 1e-5, (k, rel)


def test_misclassed_tile_order_raises_the_contract_flag_and_does_not_fault(gpu_device):
    """include/gsplat_mi355x.h, precondition of gs_isect_tile_sort*: tile_order / size_classes must describe these lists.
    A 3000-entry list handed to the <= 512 class (size_classes all zero) must not be left unwritten (the rasterizer
    would gather through uninitialised ids): its ids come out unsorted but valid, the contract flag is raised, the next
    sort call returns GS_ERR_CONTRACT once, and the call after that works."""
    from gsplat import _backend as B

    dev = gpu_device
    L = B.lib()
    TW, TH, C = 4, 2, 1
    lens = np.array([40, 3000, 0, 700, 12, 513, 1, 90], np.int64)
    n = int(lens.sum())
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int32)
    rng = np.random.default_rng(5)
    depth_bits = rng.random(n, dtype=np.float32).view(np.uint32).astype(np.uint64)
    ids = rng.integers(0, 50_000, n).astype(np.uint64)
    pairs_np = (ids << np.uint64(32)) | depth_bits  # uint2 {x = depth bits, y = id} as one little-endian u64
    t = lambda a, dt: torch.from_numpy(a).to(dev).to(dt)
    offsets = t(offs, torch.int32)
    order = torch.arange(8, device=dev, dtype=torch.int32)
    good_order = t(np.argsort(-lens, kind="stable").astype(np.int32), torch.int32)
    good_classes = t(np.array([(lens > 2048).sum(), (lens > 1024).sum(), (lens > 512).sum()], np.int32), torch.int32)
    bad_classes = torch.zeros(3, device=dev, dtype=torch.int32)  # "every list has <= 512 entries"

    def sort(order_t, classes_t):
        pairs = torch.from_numpy(pairs_np.view(np.int64)).to(dev)
        keys = torch.full((n,), -1, device=dev, dtype=torch.int64)
        fl = torch.full((n,), -7, device=dev, dtype=torch.int32)  # poison: an unwritten id is out of every range
        with B.device_guard(dev):
            st = L.gs_isect_tile_sort(C, TW, TH, B.ptr(offsets), n, B.ptr(order_t), B.ptr(classes_t), B.ptr(pairs),
                                      B.ptr(keys), B.ptr(fl), B.stream_handle(dev))
        torch.cuda.synchronize()
        return st, keys.cpu().numpy(), fl.cpu().numpy()

    assert L.gs_contract_status(1) in (0, -5)  # start from a clear flag
    st, keys, fl = sort(good_order, good_classes)
    assert st == 0 and L.gs_contract_status(0) == 0
    for a, m in zip(offs, lens):  # reference order: (depth bits, id) ascending inside every list
        seg = np.lexsort((ids[a:a + m], depth_bits[a:a + m]))
        assert np.array_equal(fl[a:a + m].astype(np.uint64), ids[a:a + m][seg])

    st, keys, fl_bad = sort(order, bad_classes)
    assert st == 0                                    # asynchronous: the launch itself is fine
    assert (fl_bad >= 0).all() and (fl_bad < 50_000).all(), "a list was left unwritten"
    for a, m in zip(offs, lens):                      # every list holds exactly its own ids, sorted or not
        assert np.array_equal(np.sort(fl_bad[a:a + m].astype(np.uint64)), np.sort(ids[a:a + m]))
    assert L.gs_contract_status(0) == -5              # GS_ERR_CONTRACT, still set
    st, _, _ = sort(good_order, good_classes)
    assert st == -5                                   # reported once by the next entry point ...
    with pytest.raises(RuntimeError, match="precondition"):
        B.check(st, "gs_isect_tile_sort")
    st, _, fl2 = sort(good_order, good_classes)
    assert st == 0 and np.array_equal(fl2, fl)        # ... and cleared
