"""Shared helpers for the parity tests (tests/ only)."""
import numpy as np
import torch

ATOL = 1e-5  # north_star: 1e-5 absolute ...
RTOL = 1e-4  # ... and 1e-4 relative (fp32)


def close_report(name, got, ref, atol=ATOL, rtol=RTOL, scale_atol=False):
    """Elementwise |got-ref| <= atol + rtol*|ref|.  With scale_atol the absolute term is taken relative to the
    tensor's peak magnitude (atol * max(1, max|ref|)): gradients in these tests are sums of thousands of fp32
    terms of size ~max|ref| (the loss is a random linear functional with unit-variance weights), so an element
    that is small by cancellation still carries rounding noise proportional to the terms, not to itself."""
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    if scale_atol and ref.numel():
        atol = atol * max(1.0, float(ref.abs().max()))
    err = (got - ref).abs()
    bad = err > (atol + rtol * ref.abs())
    frac = float(bad.double().mean()) if bad.numel() else 0.0
    rel_l2 = float((got - ref).norm() / (ref.norm() + 1e-30))
    mx = float(err.max()) if err.numel() else 0.0
    print(f"[parity] {name}: n={ref.numel()} max_abs_err={mx:.3e} rel_l2={rel_l2:.3e} out_of_tol={frac:.3e}")
    return frac, rel_l2, mx


def assert_close(name, got, ref, atol=ATOL, rtol=RTOL, max_bad_frac=0.0, max_rel_l2=None, scale_atol=False):
    frac, rel_l2, mx = close_report(name, got, ref, atol, rtol, scale_atol)
    assert frac <= max_bad_frac, f"{name}: {frac:.3e} of elements outside atol={atol} rtol={rtol} (max abs err {mx:.3e})"
    if max_rel_l2 is not None:
        assert rel_l2 <= max_rel_l2, f"{name}: rel_l2 {rel_l2:.3e} > {max_rel_l2}"


def to_dev(d, dev, dtype=torch.float32):
    return {k: (v.to(dev, dtype) if v.is_floating_point() else v.to(dev)) for k, v in d.items()}


def robust_rel_l2(got, ref, drop_frac=1e-3):
    """rel-L2 over the rows that remain after dropping the `drop_frac` rows with the largest error: a discrete decision
    that flips at an ulp (alpha < 1/255, T <= 1e-4, ceil of the radius) changes a pixel by up to ~1e-2 and with it the
    gradient of every Gaussian under that pixel; this is the error with that flip budget left out."""
    got = got.detach().double().cpu().reshape(got.shape[0], -1)
    ref = ref.detach().double().cpu().reshape(ref.shape[0], -1)
    err = (got - ref).pow(2).sum(1)
    keep = err.argsort()[: max(1, int(round(err.numel() * (1.0 - drop_frac))))]
    return float(err[keep].sum().sqrt() / (ref[keep].norm() + 1e-30))


def assert_rerun_close(name, got, ref, max_bad_frac=1e-4):
    """Two runs of the same decisions whose fp32 sums are taken in a different order (float atomics in the rasterizer
    backward).  scratch/worst_row.py on the N=6000 case: one needle-shaped Gaussian (scales 1:12, |v_quats| ~ 100 from
    cancelling terms) moves by 1e-2, 1e-4 of itself, between two orders of its atomic adds and alone decides the plain
    rel-L2 (5e-7 when the order repeats, 7e-6..1e-5 when it does not; the same single row of 6000 is outside the
    elementwise tolerance then).  So: 1e-5 with the 0.1 % worst rows left out, 1e-4 with them."""
    frac, rel_l2, mx = close_report(name, got, ref, scale_atol=True)
    assert frac <= max_bad_frac, f"{name}: {frac:.3e} of elements outside tolerance (max abs err {mx:.3e})"
    rob = robust_rel_l2(got, ref) if got.dim() >= 1 and got.shape[0] >= 1000 else rel_l2
    assert rob <= 1e-5 and rel_l2 <= 1e-4, f"{name}: rel_l2 {rel_l2:.3e} (robust {rob:.3e})"


# keys of tests/golden/losses_ref.npz (written by tests/golden/make_golden.py::losses_golden from the reference's own
# models/losses.py)
DEPTH_CASES = [(t, n, i, r) for t in ("l1", "l2", "smooth_l1") for n in (False, True) for i in (False, True)
               for r in ("mean_on_hit", "mean_on_hw", "sum", "none")]
PERCENTILE_CASES = [("l1", False, False, "mean_on_hit", 0.95), ("l2", True, False, "sum", 0.5)]


def depth_case_key(loss_type, normalize, inverse, reduction, percentile=None):
    k = f"depth_{loss_type}_n{int(normalize)}_i{int(inverse)}_{reduction}"
    return k if percentile is None else f"{k}_p{int(round(percentile * 100))}"
