"""Tolerances of the parity tests.  BASELINE.json's north_star: "within 1e-5 fp32"."""
import numpy as np

TOL = 1e-5


def rel_err(got, ref):
    """max |got - ref| relative to max(1, max |ref|): one number per tensor (hides small elements of a wide-range
    tensor — always use together with elem_excess)"""
    ref = np.asarray(ref, dtype=np.float64)
    if ref.size == 0:
        return 0.0
    return float(np.max(np.abs(np.asarray(got, dtype=np.float64) - ref))) / max(1.0, float(np.max(np.abs(ref))))


def elem_excess(got, ref, tol=TOL):
    """ELEMENTWISE bar: max over elements of |got - ref| / (tol * max(1, |ref|)); <= 1 means every element is within
    tol relative to its own magnitude (absolute below 1).  Non-finite reference elements must match in kind."""
    ref = np.asarray(ref, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64).reshape(ref.shape)
    if ref.size == 0:
        return 0.0
    fin = np.isfinite(ref)
    if not np.array_equal(np.isfinite(got), fin):
        return float("inf")
    if not fin.any():
        return 0.0
    return float(np.max(np.abs(got[fin] - ref[fin]) / (tol * np.maximum(1.0, np.abs(ref[fin])))))


def assert_close(got, ref, tol=TOL, what=""):
    r, e = rel_err(got, ref), elem_excess(got, ref, tol)
    assert r <= tol and e <= 1.0, f"{what}: max-normalised {r:.3e}, worst element at {e:.3f} x the {tol:g} bar"
