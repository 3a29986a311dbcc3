"""gs_radix_sort_pairs_u64_u32_static -- the one entry point that allocates nothing (its digit table is a static device
array of the library; include/gsplat_mi355x.h) -- against gs_radix_sort_pairs_u64_u32_bits, the existing entry point
whose kernels it shares: bit-identical keys and values, through the C ABI, on two calls in a row, the second one
smaller (the table keeps the first call's contents)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _sort_both(B, keys_np, vals_np, begin_bit, end_bit, dev):
    lib = B.lib()
    n = keys_np.size
    outs = []
    for static in (False, True):
        ka = torch.from_numpy(keys_np.copy()).to(dev)
        va = torch.from_numpy(vals_np.copy()).to(dev)
        kb, vb = torch.empty_like(ka), torch.empty_like(va)
        in_b = ctypes.c_int32(-1)
        st = torch.cuda.current_stream(dev).cuda_stream
        if static:
            rc = lib.gs_radix_sort_pairs_u64_u32_static(n, begin_bit, end_bit, B.ptr(ka), B.ptr(va), B.ptr(kb), B.ptr(vb),
                                                        ctypes.byref(in_b), st)
        else:
            wsb = int(lib.gs_radix_sort_workspace_bytes(n))
            ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
            rc = lib.gs_radix_sort_pairs_u64_u32_bits(n, begin_bit, end_bit, B.ptr(ka), B.ptr(va), B.ptr(kb), B.ptr(vb),
                                                      B.ptr(ws), wsb, ctypes.byref(in_b), st)
        assert rc == 0, (static, rc)
        torch.cuda.synchronize()
        k, v = (kb, vb) if in_b.value else (ka, va)
        outs.append((k.cpu().numpy(), v.cpu().numpy(), in_b.value))
    return outs


def test_static_scratch_sort_matches_workspace_sort(gpu_device):
    from gsplat import _backend as B

    rng = np.random.default_rng(7)
    # two calls in a row on the same GPU, the second smaller: stale counters of the first call's larger table must not
    # leak into it; then one at the static maximum's block granularity edge
    for n, begin_bit, end_bit in [(1_000_003, 0, 46), (5_000, 0, 46), (70_001, 3, 50), (1, 0, 40)]:
        keys = rng.integers(0, (1 << 62) - 1, size=n, dtype=np.int64)
        keys[: n // 3] &= (1 << 20) - 1  # many ties: stability matters
        vals = np.arange(n, dtype=np.int32)
        (k0, v0, b0), (k1, v1, b1) = _sort_both(B, keys, vals, begin_bit, end_bit, gpu_device)
        assert b0 == b1
        assert np.array_equal(k0, k1) and np.array_equal(v0, v1), (n, begin_bit, end_bit)
        mask = ((1 << end_bit) - 1) & ~((1 << begin_bit) - 1)
        order = np.argsort((keys & mask).astype(np.uint64), kind="stable")
        assert np.array_equal(v1, vals[order]) and np.array_equal(k1, keys[order])


def test_static_scratch_sort_rejects_oversize_and_bad_args(gpu_device):
    from gsplat import _backend as B

    lib = B.lib()
    in_b = ctypes.c_int32(0)
    one = torch.zeros(8, device=gpu_device, dtype=torch.int64)
    st = torch.cuda.current_stream(gpu_device).cuda_stream
    too_big = (1 << 24) + 1  # GS_RADIX_SORT_STATIC_MAX_N + 1: rejected before anything is launched or dereferenced
    rc = lib.gs_radix_sort_pairs_u64_u32_static(too_big, 0, 46, B.ptr(one), B.ptr(one), B.ptr(one), B.ptr(one),
                                                ctypes.byref(in_b), st)
    assert rc == -3  # GS_ERR_WORKSPACE
    assert b"workspace" in lib.gs_error_string(rc).lower()
    assert lib.gs_radix_sort_pairs_u64_u32_static(8, 0, 70, B.ptr(one), B.ptr(one), B.ptr(one), B.ptr(one),
                                                  ctypes.byref(in_b), st) == -1
    assert lib.gs_radix_sort_pairs_u64_u32_static(0, 0, 46, None, None, None, None, ctypes.byref(in_b), st) == 0


def test_static_scratch_sort_on_a_side_stream_under_device_guard(gpu_device):
    """The scratch copy is chosen for the stream's device, which must be the current one (header): a non-default stream
    of the device under `device_guard` sorts like the default stream."""
    from gsplat import _backend as B

    lib = B.lib()
    rng = np.random.default_rng(11)
    n = 300_007
    keys = rng.integers(0, (1 << 45) - 1, size=n, dtype=np.int64)
    vals = np.arange(n, dtype=np.int32)
    side = torch.cuda.Stream(gpu_device)
    ka, va = torch.from_numpy(keys).to(gpu_device), torch.from_numpy(vals).to(gpu_device)
    kb, vb = torch.empty_like(ka), torch.empty_like(va)
    in_b = ctypes.c_int32(-1)
    side.wait_stream(torch.cuda.current_stream(gpu_device))
    with B.device_guard(gpu_device):
        rc = lib.gs_radix_sort_pairs_u64_u32_static(n, 0, 46, B.ptr(ka), B.ptr(va), B.ptr(kb), B.ptr(vb),
                                                    ctypes.byref(in_b), side.cuda_stream)
    assert rc == 0
    side.synchronize()
    k, v = (kb, vb) if in_b.value else (ka, va)
    order = np.argsort(keys.astype(np.uint64), kind="stable")
    assert np.array_equal(v.cpu().numpy(), vals[order]) and np.array_equal(k.cpu().numpy(), keys[order])
