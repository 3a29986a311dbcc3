"""Posted results (include/gsplat_mi355x.h, "posted results"): the offsets / tile-order kernels store n_isects and the size
classes straight into a polled host slot instead of a copy + stream synchronisation.  Same lists either way; a slot
nobody posts into ends in GS_ERR_NOT_POSTED after the time limit, not in a hang."""
import ctypes
import math

import numpy as np
import pytest
import torch

from test_gpu_parity import _synthetic_records

pytestmark = pytest.mark.gpu


def test_posted_and_copied_results_agree(gpu_device):
    import gsplat.cuda._wrapper as W

    rng = np.random.default_rng(5)
    C, N, TW, TH = 2, 30_000, 40, 24
    rec, tiles, keys, ids, offsets = _synthetic_records(rng, C, N, TW, TH, 3000, 1 << 20, gpu_device)
    out = {}
    saved = W.POSTED_RESULTS
    try:
        for posted in (True, False, True):
            W.POSTED_RESULTS = posted
            k, f, o, order, fs = W.isect_tiles_binned(rec, tiles, TW, TH, front=True)
            assert fs is not None  # (a list above 2048 entries: the front-first path, with the size classes in play)
            if posted:  # the classes the offsets kernel posted are the ones the tile-order kernel found
                assert list(W._post_slot()[2]) == fs.n_long.cpu().tolist()
            fs.finalize()
            out[posted] = (k.cpu().numpy(), f.cpu().numpy(), o.cpu().numpy())
    finally:
        W.POSTED_RESULTS = saved
    for posted in (True, False):
        assert np.array_equal(out[posted][0], keys)
        assert np.array_equal(out[posted][1], ids)
        assert np.array_equal(out[posted][2], offsets)


def test_wait_on_a_slot_nobody_posts_into_gives_up(gpu_device):
    from gsplat import _backend as B

    lib = B.lib()
    p = ctypes.c_void_p()
    with B.device_guard(gpu_device):
        B.check(lib.gs_host_post_alloc(ctypes.byref(p), 64), "gs_host_post_alloc")
        try:
            n = (ctypes.c_int64 * 1).from_address(p.value)
            sc = (ctypes.c_int32 * 3).from_address(p.value + 8)
            n[0] = -1
            st = B.stream_handle(gpu_device)
            assert lib.gs_host_post_wait(p.value, None, 5, st) == -4       # GS_ERR_NOT_POSTED, after a stream sync
            n[0] = 12345
            assert lib.gs_host_post_wait(p.value, None, 5, st) == 0
            sc[0], sc[1], sc[2] = 1, 2, -1
            assert lib.gs_host_post_wait(p.value, p.value + 8, 5, st) == -4  # one size class still missing
            sc[2] = 3
            assert lib.gs_host_post_wait(p.value, p.value + 8, 5, st) == 0
            # a kernel posts: the offsets entry point on a tiny table
            C, TW, TH, S = 1, 4, 3, 2
            table = torch.ones(C * S * TW * TH, device=gpu_device, dtype=torch.int32)
            counts = torch.empty(C * TW * TH, device=gpu_device, dtype=torch.int32)
            offs = torch.empty(C * TW * TH, device=gpu_device, dtype=torch.int32)
            total = torch.empty(1, device=gpu_device, dtype=torch.int64)
            n[0] = -1
            sc[0] = sc[1] = sc[2] = -1
            B.check(lib.gs_isect_bin_offsets_post(C, TW, TH, S, B.ptr(table), B.ptr(counts), B.ptr(offs), B.ptr(total),
                                                  p.value, p.value + 8, st), "gs_isect_bin_offsets_post")
            assert lib.gs_host_post_wait(p.value, p.value + 8, 10_000, st) == 0
            assert n[0] == S * TW * TH == int(total.item())
            assert list(sc) == [0, 0, 0]  # no list above 512 entries
            assert torch.equal(offs.cpu(), torch.arange(TW * TH, dtype=torch.int32) * S)
        finally:
            torch.cuda.synchronize()
            B.check(lib.gs_host_post_free(p.value), "gs_host_post_free")
