"""CPU-only checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/*.h declares,
the gsplat module paths the reference imports exist, and the host wrappers validate arguments / fail loudly without
a HIP device (no CPU fallback).  No compute calls are made here."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "gsplat_mi355x.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(gs_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from gsplat import _backend as B

    lib = ctypes.CDLL(B.lib_path())
    declared = _declared_symbols()
    assert len(declared) >= 46, declared
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/gsplat_mi355x.h but not exported"
    assert set(declared) == set(B.EXPORTED_SYMBOLS), set(declared) ^ set(B.EXPORTED_SYMBOLS)
    assert B.lib().gs_version() >= 100
    assert B.lib().gs_error_string(-2).decode().startswith("gsplat_mi355x")
    # pure host-side entry points (no GPU needed)
    assert B.lib().gs_exclusive_scan_workspace_bytes(5_000_000) >= 8 * (5_000_000 // 2048)
    assert B.lib().gs_radix_sort_workspace_bytes(10_000_000) >= 512 * 4 * (10_000_000 // 4096)
    assert B.lib().gs_isect_bin_max_tiles() >= 9600  # 1920x1280 / 16^2 per camera


def test_proj_params_struct_matches_header():
    from gsplat import _backend as B

    hdr = open(os.path.join(ROOT, "include", "gsplat_mi355x.h")).read()
    body = re.search(r"typedef struct gs_proj_params \{(.*?)\} gs_proj_params;", hdr, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        typ, names = decl.split(None, 1)
        fields += [(n.strip(), typ) for n in names.split(",")]
    got = [(n, {"c_int": "int32_t", "c_float": "float"}[t.__name__]) for n, t in B.ProjParams._fields_]
    assert got == fields


def test_reference_import_paths_exist():
    # exactly the four imports of /root/reference/3D-Reconstruction-main/models/gaussians/basics.py:12-15
    from gsplat.rendering import rasterization  # noqa: F401
    from gsplat.cuda_legacy._wrapper import num_sh_bases  # noqa: F401
    from gsplat.cuda_legacy._torch_impl import quat_to_rotmat  # noqa: F401
    from gsplat.cuda._wrapper import spherical_harmonics  # noqa: F401
    import inspect

    sig = inspect.signature(rasterization)
    names = list(sig.parameters)
    assert names[:9] == ["means", "quats", "scales", "opacities", "colors", "viewmats", "Ks", "width", "height"]
    for kw, default in [("near_plane", 0.01), ("far_plane", 1e10), ("radius_clip", 0.0), ("eps2d", 0.3),
                        ("sh_degree", None), ("packed", True), ("tile_size", 16), ("backgrounds", None),
                        ("render_mode", "RGB"), ("sparse_grad", False), ("absgrad", False),
                        ("rasterize_mode", "classic"), ("channel_chunk", 32)]:
        assert sig.parameters[kw].default == default, kw


def test_stage_level_functions_have_gsplat_signatures_and_no_cpu_path():
    """gsplat 1.x's gsplat/cuda/_wrapper.py: the positional parameters and defaults of the four stage functions."""
    import inspect
    import gsplat

    want = {
        "fully_fused_projection": (["means", "covars", "quats", "scales", "viewmats", "Ks", "width", "height"],
                                   dict(eps2d=0.3, near_plane=0.01, far_plane=1e10, radius_clip=0.0, packed=False,
                                        sparse_grad=False, calc_compensations=False)),
        "isect_tiles": (["means2d", "radii", "depths", "tile_size", "tile_width", "tile_height"],
                        dict(sort=True, packed=False, n_cameras=None, camera_ids=None, gaussian_ids=None)),
        "isect_offset_encode": (["isect_ids", "n_cameras", "tile_width", "tile_height"], {}),
        "rasterize_to_pixels": (["means2d", "conics", "colors", "opacities", "image_width", "image_height", "tile_size",
                                 "isect_offsets", "flatten_ids"],
                                dict(backgrounds=None, masks=None, packed=False, absgrad=False)),
    }
    for name, (positional, defaults) in want.items():
        fn = getattr(gsplat, name)
        assert fn is getattr(gsplat.cuda._wrapper, name)
        sig = inspect.signature(fn)
        assert list(sig.parameters)[:len(positional)] == positional, name
        for kw, default in defaults.items():
            assert sig.parameters[kw].default == default, (name, kw)
    z = lambda *s: torch.zeros(*s)
    with pytest.raises(RuntimeError, match="no CPU path"):
        gsplat.fully_fused_projection(z(4, 3), None, z(4, 4), z(4, 3), z(1, 4, 4), z(1, 3, 3), 32, 32)
    with pytest.raises(RuntimeError, match="no CPU path"):
        gsplat.isect_tiles(z(1, 4, 2), z(1, 4).int(), z(1, 4), 16, 2, 2)
    with pytest.raises(RuntimeError, match="no CPU path"):
        gsplat.rasterize_to_pixels(z(1, 4, 2), z(1, 4, 3), z(1, 4, 3), z(1, 4), 32, 32, 16, z(1, 2, 2).int(), z(0).int())
    with pytest.raises(NotImplementedError):
        gsplat.fully_fused_projection(z(4, 3), None, z(4, 4), z(4, 3), z(1, 4, 4), z(1, 3, 3), 32, 32, sparse_grad=True)


def test_no_cpu_fallback_and_shape_errors():
    from gsplat.rendering import rasterization
    from gsplat.cuda._wrapper import spherical_harmonics

    N = 8
    z = lambda *s: torch.zeros(*s)
    args = dict(means=z(N, 3), quats=z(N, 4), scales=z(N, 3), opacities=z(N), colors=z(N, 3), viewmats=z(1, 4, 4),
                Ks=z(1, 3, 3), width=32, height=32, packed=False)
    with pytest.raises(RuntimeError, match="no CPU path"):
        rasterization(**args)
    with pytest.raises(RuntimeError, match="no CPU path"):
        spherical_harmonics(1, z(N, 3), z(N, 4, 3))
    bad = dict(args, opacities=z(N, 1))  # the reference squeezes [N,1] -> [N] itself (base.py:397)
    with pytest.raises(AssertionError):
        rasterization(**bad)
    with pytest.raises(AssertionError):
        rasterization(**dict(args, render_mode="RGBA"))
    with pytest.raises(AssertionError):
        spherical_harmonics(3, z(N, 3), z(N, 4, 3))  # (3+1)^2 > K
    with pytest.raises(NotImplementedError):
        rasterization(**dict(args, sparse_grad=True))


def test_product_path_does_not_import_oracle():
    pkg = os.path.join(ROOT, "3d-reconstruction_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dp, f)


def test_shard_views_partition():
    from gsplat.distributed import shard_views

    for n in [0, 1, 5, 8, 40, 600]:
        for ws in [1, 2, 3, 8]:
            got = [i for r in range(ws) for i in shard_views(n, r, ws)]
            assert got == list(range(n))
            sizes = [len(shard_views(n, r, ws)) for r in range(ws)]
            assert max(sizes) - min(sizes) <= 1


def test_camera_centers_and_saved_tensor_identity():
    from gsplat.cuda._wrapper import _camera_centers
    import synth_scene

    vm, _ = synth_scene.make_cameras(5, 64, 48, dtype=torch.float64)
    assert torch.allclose(_camera_centers(vm), torch.linalg.inv(vm)[:, :3, 3], atol=1e-12)

    # the `.absgrad` contract relies on save_for_backward handing back the SAME python object for an input
    class F(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, hook):
            ctx.save_for_backward(hook)
            return x * 2

        @staticmethod
        def backward(ctx, g):
            (hook,) = ctx.saved_tensors
            hook.absgrad = g.abs()
            return g * 2, g

    a = torch.ones(3, requires_grad=True)
    hook = a * 1.0
    hook.retain_grad()
    F.apply(a, hook).sum().backward()
    assert hasattr(hook, "absgrad") and hook.grad is not None


def test_meta_dict_completes_lazy_entries_on_read():
    """`meta` of rasterization(): front-sorted isect arrays are completed, and deferred entries built, on first read."""
    from gsplat.rendering import _Meta

    class FakeFront:
        def __init__(self):
            self.calls = 0

        def finalize(self):
            self.calls += 1

    fs = FakeFront()
    m = _Meta({"isect_ids": "ids", "flatten_ids": "flat", "radii": None, "width": 7})
    m._front_sort = fs
    built = []
    m._thunks["radii"] = lambda: built.append(1) or "radii!"
    assert m["width"] == 7 and fs.calls == 0 and not built      # unrelated keys trigger nothing
    assert "flatten_ids" in m and fs.calls == 0                   # membership does not read
    assert m["radii"] == "radii!" and built == [1] and fs.calls == 0
    assert m["radii"] == "radii!" and built == [1]                # built once
    assert m.get("flatten_ids") == "flat" and fs.calls == 1
    assert m["isect_ids"] == "ids" and fs.calls == 1              # completed once
    m2 = _Meta({"isect_ids": 1, "x": None})
    m2._front_sort = fs
    m2._thunks["x"] = lambda: 5
    assert dict(m2.items()) == {"isect_ids": 1, "x": 5} and fs.calls == 2
    assert m2.copy() == {"isect_ids": 1, "x": 5}


def test_entry_points_reject_bad_arguments_before_any_launch():
    """Argument validation happens on the host, ahead of any HIP call, so it can be exercised without a device:
    every entry point answers malformed sizes with GS_ERR_INVALID_ARG / GS_ERR_UNSUPPORTED / GS_ERR_WORKSPACE (and
    `gs_error_string` names the code) instead of launching."""
    from gsplat import _backend as B

    L = B.lib()
    INVALID, UNSUPPORTED, WORKSPACE = -1, -2, -3
    P = 0x1000  # a non-null, 16-byte aligned dummy: never dereferenced on these paths
    assert L.gs_sh_fwd(10, 16, 5, P, P, None, P, None) == INVALID                 # degree > 4
    assert L.gs_sh_fwd(10, 4, 3, P, P, None, P, None) == INVALID                  # (3+1)^2 > K
    assert L.gs_sh_bwd(-1, 16, 3, P, P, None, P, P, None, None) == INVALID
    pp = B.ProjParams(N=0, C=1, width=64, height=48, tile_size=16, tile_w=4, tile_h=3, eps2d=0.3, near_plane=0.01,
                      far_plane=1e10, radius_clip=0.0, calc_compensations=0, color_mode=3, color_dim=3, sh_degree=3,
                      sh_K=16, depth_channel=0, sh_rows=0)
    fwd = lambda: L.gs_projection_fwd(ctypes.byref(pp), *([P] * 15), None)
    assert fwd() == 0                                                             # N = 0: nothing to launch
    pp.sh_rows = 3
    assert fwd() == INVALID                                                       # sh_rows outside 0..2
    pp.sh_rows, pp.tile_w = 2, 5
    assert fwd() == INVALID                                                       # tile grid does not match the image
    assert L.gs_activate_fwd(10, 3, 3, P, P, P, P, P, None, P, 0, P, P, P, P, P, P, P, P, None) == INVALID  # K_rest < 15
    assert L.gs_activate_fwd(10, 15, 3, P, P, P, P, P, None, P, 3, P, P, P, P, P, P, P, P, None) == INVALID  # scale_mode
    assert L.gs_refine_gather_rows(10, 300, 0, P, P, None, P, P, None) == INVALID   # W > 256
    assert L.gs_refine_plan(10, 300, P, P, P, P, P, None) == INVALID                # n_split_samples > 250
    assert L.gs_l1_ssim_fwd(10, 64, P, P, None, None, P, 1 << 20, P, None) == UNSUPPORTED    # H <= 10
    assert L.gs_l1_ssim_fwd(64, 64, P, P, None, None, P, 1, P, None) == WORKSPACE
    assert L.gs_pixel_losses_fwd(0, P, None, None, None, None, 80.0, P, 1 << 20, P, None) == INVALID
    assert L.gs_pixel_losses_fwd(100, P, None, None, None, P, 80.0, P, 1 << 20, P, None) == INVALID  # lidar without depth
    assert L.gs_cube_texture_fwd(10, 0, 3, P, P, None, P, None) == INVALID          # R < 1
    assert L.gs_cube_texture_bwd(10, 8, 100, P, None, P, P, None) == INVALID        # C > 64
    assert L.gs_rigid_transform_fwd(10, 0, P, P, P, P, P, P, P, None) == INVALID    # no instances
    assert L.gs_rigid_transform_fwd(10, 3, P, P + 4, P, P, P, P, P, None) == INVALID  # misaligned quats
    too_many = L.gs_isect_bin_max_tiles() + 1
    assert L.gs_isect_bin_count(1, 10, too_many, 1, P, P, 1, P, None, None, None, 0, None) == UNSUPPORTED
    assert L.gs_isect_bin_count(1, 10, 4, 4, P, P, 1, P, None, None, P, 16, None) == INVALID  # tile masks without the lists
    assert L.gs_isect_bin_scatter_regions(1, 10, 4, 4, 1, 0, 2, P, P, None, P, P, None, P, None) == INVALID  # slices_per_group < 1
    # composite launches: rejected before their first launch
    assert L.gs_isect_bin_prepare(1, 10, 4, 4, P, P, 1, P, P, P, None, 0, P, P, P, P, P, None, None, None, 1, None) == INVALID  # no host scalar
    assert L.gs_isect_bin_prepare(1, 10, too_many, 1, P, P, 1, P, P, P, None, 0, P, P, P, P, P, None, P, None, 1, None) == UNSUPPORTED
    assert L.gs_isect_bin_prepare(1, 10, 4, 4, P, P, 1, P, P, P, None, 0, P, P, P, P, P, None, P, None, 3, None) == INVALID  # synchronize mode
    # the keys array is optional only when the host's size classes show that no list needs global scratch
    assert L.gs_isect_tile_sort_ex(1, 4, 4, P, 5, P, P, P, None, P, None, None) == INVALID
    three = (ctypes.c_int32 * 3)(1, 1, 1)
    assert L.gs_isect_tile_sort_ex(1, 4, 4, P, 5, P, P, P, None, P, ctypes.addressof(three), None) == INVALID
    assert L.gs_frames_to_uint8(-1, P, P, None) == INVALID
    assert L.gs_frames_to_uint8(8, None, P, None) == INVALID
    assert L.gs_frames_to_uint8(0, None, None, None) == 0
    # posted results: slots and waits
    assert L.gs_host_post_alloc(None, 64) == INVALID
    assert L.gs_host_post_wait(None, None, 10, None) == INVALID
    assert L.gs_host_post_wait(P, None, -1, None) == INVALID
    assert L.gs_isect_bin_offsets_post(0, 4, 4, 1, P, P, P, P, None, None, None) == INVALID
    assert L.gs_raster_tile_order_post(10, None, 0, P, P, None, P, None) == INVALID  # posted classes without the device copy
    assert b"post" in L.gs_error_string(-4).lower()
    assert L.gs_isect_bin_build(1, 10, 4, 4, 1, 0, 2, P, P, None, P, P, 5, P, P, 0, P, P, P, P, None, None) == INVALID  # slices_per_group < 1
    assert L.gs_rasterize_fwd_front(1, 64, 64, 16, 4, P, P, P, 5, None, 0, P, P, P, P, None, None, P, None, 10, None, P, P, P,
                                    None, None) == INVALID  # no sorted_counts
    assert L.gs_densify_accumulate(0, 10, P, P, 16, 1.0, 1.0, 1.0, 0, P, P, P, None) == INVALID   # C < 1
    assert L.gs_exclusive_scan_workspace_bytes(1000) > 0
    for code, word in ((INVALID, b"invalid"), (UNSUPPORTED, b"unsupported"), (WORKSPACE, b"workspace")):
        assert word in L.gs_error_string(code).lower()
    # empty inputs are fine and launch nothing
    assert L.gs_sh_fwd(0, 16, 3, None, None, None, None, None) == 0
    assert L.gs_cube_texture_fwd(0, 8, 3, None, None, None, None, None) == 0
    assert L.gs_rigid_transform_fwd(0, 0, None, None, None, None, None, None, None, None) == 0


def test_speculative_lists_protocol_on_the_host():
    """gsplat.cuda._wrapper.SpecLists (device-side n_isects): what resolve() accepts and records, with the posted count
    and the exact-size rebuild replaced by host stand-ins (no device work)."""
    from gsplat.cuda import _wrapper as W

    built = []

    def make(capacity, posted, with_ids=True):
        ids = torch.zeros(capacity, dtype=torch.int64) if with_ids else None
        fids = torch.arange(capacity, dtype=torch.int32)
        return W.SpecLists(capacity, None, ids, fids, torch.zeros(1, 2, 2, dtype=torch.int32), None,
                           lambda: posted, lambda n, classes: built.append((n, tuple(classes))) or "rebuilt", ("k", capacity))

    W._spec_table().clear()
    s = make(100, (60, [0, 0, 3]))                       # fits: the lists are cut to the posted count
    assert s.resolve() and s.n_isects == 60 and s.flatten_ids.shape == (60,) and s.isect_ids.shape == (60,)
    assert W._spec_table()[("k", 100)] == (60, (0, 0, 3))
    s = make(100, (101, [0, 0, 3]))                      # one entry too many: redone at the exact size
    assert not s.resolve() and s.redo() == "rebuilt" and built[-1] == (101, (0, 0, 3))
    assert W._spec_table()[("k", 100)] == (101, (0, 0, 3))   # ... and the next call of the shape is sized from it
    s = make(100, (100, [0, 1, 0]))                      # exactly the capacity fits
    assert s.resolve() and s.flatten_ids.shape == (100,)
    s = make(100, (0, [0, 0, 0]))                        # nothing visible: the exact path builds the empty lists
    assert not s.resolve()
    s = make(100, (50, [2, 0, 0]), with_ids=True)        # a list above 2048 entries sorts through the key array: fine
    assert s.resolve()
    s = make(100, (50, [2, 0, 0]), with_ids=False)       # ... without it the speculative sort could not have sorted it
    assert not s.resolve()
    W._spec_table().clear()
