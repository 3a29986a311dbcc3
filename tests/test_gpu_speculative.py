"""Speculative list sizes (include/gsplat_mi355x.h, "device-side n_isects"; gsplat.cuda._wrapper.SPECULATE_SIZES): from the
second call of a shape on, rasterization() queues scatter, sort and the forward rasterizer on arrays sized from the last
call's intersection count, before the host has looked at this call's.  Observable results must not depend on it: same
bits when the speculation holds, and when it does not -- the count outgrows the capacity, or a tile list is longer than
the speculative sort can take -- the call is redone at the exact size with the same bits again."""
import pytest
import torch

import synth_scene
from test_gpu_front_sort import _dense_scene
from util import assert_rerun_close

pytestmark = pytest.mark.gpu
KW = dict(near_plane=0.1, packed=False, render_mode="RGB+ED", absgrad=True, rasterize_mode="antialiased")


def _step(dev, g, vm, Ks, Wd, Hd, weights, sh_degree=None):
    from gsplat.rendering import rasterization

    p = {k: v.to(dev).requires_grad_(True) for k, v in g.items()}
    rc, ra, meta = rasterization(p["means"], p["quats"], p["scales"], p["opacities"], p["colors"], vm.to(dev),
                                 Ks.to(dev), Wd, Hd, sh_degree=sh_degree, **KW)
    ((rc * weights[0]).sum() + (ra * weights[1]).sum()).backward()
    grads = {k: v.grad.clone() for k, v in p.items()}
    grads["absgrad"] = meta["means2d"].absgrad.clone()
    return rc.detach(), ra.detach(), meta, grads


def _weights(dev, C, Hd, Wd, seed=3):
    gen = torch.Generator().manual_seed(seed)
    return torch.randn(C, Hd, Wd, 4, generator=gen).to(dev), torch.randn(C, Hd, Wd, 1, generator=gen).to(dev)


class _Spy:
    """Counts how gsplat.cuda._wrapper.SpecLists.resolve() came out."""

    def __init__(self, W):
        self.W, self.results = W, []

    def __enter__(self):
        orig = self.orig = self.W.SpecLists.resolve
        spy = self

        def resolve(lists):
            ok = orig(lists)
            spy.results.append((ok, lists.n_isects, lists.capacity))
            return ok

        self.W.SpecLists.resolve = resolve
        return self

    def __exit__(self, *exc):
        self.W.SpecLists.resolve = self.orig
        return False


def _same(a, b):
    rc1, ra1, m1, g1 = a
    rc0, ra0, m0, g0 = b
    assert torch.equal(rc1, rc0) and torch.equal(ra1, ra0)
    for k in ("last_ids", "radii", "isect_offsets", "flatten_ids", "isect_ids", "tiles_per_gauss"):
        assert torch.equal(m1[k], m0[k]), k
    for k in g0:
        assert_rerun_close(f"spec.v_{k}", g1[k], g0[k])


def test_speculation_holds_and_changes_nothing(gpu_device):
    import gsplat.cuda._wrapper as W

    dev = gpu_device
    N, C, Wd, Hd = 30_000, 2, 200, 120
    g = synth_scene.make_gaussians(N, sh_degree=3, seed=5, extent=12.0)
    vm, Ks = synth_scene.make_cameras(C, Wd, Hd)
    wts = _weights(dev, C, Hd, Wd)
    saved = W.SPECULATE_SIZES
    try:
        W.SPECULATE_SIZES = False
        ref = _step(dev, g, vm, Ks, Wd, Hd, wts, 3)
        W.SPECULATE_SIZES = True
        W._spec_table().clear()
        with _Spy(W) as spy:
            first = _step(dev, g, vm, Ks, Wd, Hd, wts, 3)    # no history for this shape: the exact path
            assert spy.results == []
            second = _step(dev, g, vm, Ks, Wd, Hd, wts, 3)   # speculative, holds
            third = _step(dev, g, vm, Ks, Wd, Hd, wts, 3)
            assert [r[0] for r in spy.results] == [True, True], spy.results
            assert all(0 < n <= cap for _, n, cap in spy.results)
        for run in (first, second, third):
            _same(run, ref)
        # the internal lists are views of the capacity-sized arrays, cut to the count
        assert second[2]._lists[3].numel() == spy.results[0][1]
    finally:
        W.SPECULATE_SIZES = saved


def test_overflowing_speculation_is_redone_bit_identically(gpu_device):
    import gsplat.cuda._wrapper as W

    dev = gpu_device
    N, C, Wd, Hd = 30_000, 2, 200, 120
    g = synth_scene.make_gaussians(N, sh_degree=None, seed=6, extent=12.0)
    vm, Ks = synth_scene.make_cameras(C, Wd, Hd)
    wts = _weights(dev, C, Hd, Wd, 4)
    saved = W.SPECULATE_SIZES
    try:
        W.SPECULATE_SIZES = False
        ref = _step(dev, g, vm, Ks, Wd, Hd, wts)
        W.SPECULATE_SIZES = True
        W._spec_table().clear()
        _step(dev, g, vm, Ks, Wd, Hd, wts)
        table = W._spec_table()
        assert len(table) == 1
        for k in table:  # "the last call of this shape had 100 intersections": capacity 1149, far below this call's
            table[k] = (100, (0, 0, 0))
        with _Spy(W) as spy:
            over = _step(dev, g, vm, Ks, Wd, Hd, wts)
            assert len(spy.results) == 1 and spy.results[0][0] is False and spy.results[0][1] > spy.results[0][2]
            again = _step(dev, g, vm, Ks, Wd, Hd, wts)   # the redo refreshed the history: this one holds
            assert spy.results[1][0] is True
        _same(over, ref)
        _same(again, ref)
        # grown scene: 40 % more intersections than the last call -> overflow, redo, same bits as an exact call
        g2 = dict(g)
        g2["scales"] = g["scales"] * 1.3
        W.SPECULATE_SIZES = False
        ref2 = _step(dev, g2, vm, Ks, Wd, Hd, wts)
        W.SPECULATE_SIZES = True
        with _Spy(W) as spy:
            grown = _step(dev, g2, vm, Ks, Wd, Hd, wts)
            print(f"[spec] grown scene: {spy.results}")
        _same(grown, ref2)
    finally:
        W.SPECULATE_SIZES = saved


def test_a_list_too_long_for_the_speculative_sort_is_redone(gpu_device):
    """History says "no list above 2048 entries", the scene has lists far above 6144: the speculative whole-list sort has no
    scratch for them (contract flag), the call is redone with the front-first sort, and no later call sees the flag."""
    import gsplat.cuda._wrapper as W
    from gsplat import _backend as B

    dev = gpu_device
    N, Wd, Hd = 250_000, 96, 64
    g = _dense_scene(N, 0.3, 0.95)
    g["scales"] = g["scales"] * 2.0
    vm, Ks = synth_scene.make_cameras(1, Wd, Hd)
    wts = _weights(dev, 1, Hd, Wd, 5)
    saved = W.SPECULATE_SIZES
    try:
        W.SPECULATE_SIZES = False
        ref = _step(dev, g, vm, Ks, Wd, Hd, wts)
        off = ref[2]._lists[1].reshape(-1).long()
        lens = torch.diff(torch.cat([off, torch.tensor([ref[2]._lists[3].numel()], device=dev)]))
        assert int(lens.max()) > 6144, int(lens.max())
        W.SPECULATE_SIZES = True
        W._spec_table().clear()
        _step(dev, g, vm, Ks, Wd, Hd, wts)
        table = W._spec_table()
        for k in table:
            table[k] = (table[k][0], (0, 0, 0))
        with _Spy(W) as spy:
            out = _step(dev, g, vm, Ks, Wd, Hd, wts)
            assert len(spy.results) == 1 and spy.results[0][0] is False
            nxt = _step(dev, g, vm, Ks, Wd, Hd, wts)      # history now knows about the long lists: exact path
            assert len(spy.results) == 1
        _same(out, ref)
        _same(nxt, ref)
        torch.cuda.synchronize()
        assert B.lib().gs_contract_status(0) == 0
    finally:
        W.SPECULATE_SIZES = saved
