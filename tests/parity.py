"""Comparison helpers.  The parity bar (BASELINE.json north_star): bit-exact for indexing / copy,
1e-3 relative for floating point.  How "1e-3 relative" is measured, written once here:

* T = float: every element |a - b| <= rel * (|b| + rms(b)).  rms(b) keeps the bound meaningful for
  elements that are the result of cancellation; rel defaults to 1e-3, most tests pass 1e-4.
* T = bfloat: both sides are rounded to bf16 (8 significant bits, ulp = 2^-7 |b| ... 2^-8 |b|), so
  two correctly computed results whose fp32 values differ in the last bits can land on adjacent
  bf16 values.  Two modes:
    scale_aware=False (single kernels, where only the fp32 summation order differs): every
      element within `max_ulp` bf16 steps of the oracle.
    scale_aware=True (compositions: an upstream 1-ulp difference reaches a downstream dot product
      as an ABSOLUTE perturbation of about 2^-8 of a typical term, which is many ulps of an output
      that is small through cancellation): |a - b| <= max_ulp * 2^-7 * max(|b|, rms(b)).
  In both modes at most `max_frac` of the elements may differ at all and the vector-wise relative
  error ||a - b|| / ||b|| must be <= rel.
"""
import numpy as np

from oracle import mc_oracle as mo

BF16, F32 = 0, 1


def bf16_ordinal(bits: np.ndarray) -> np.ndarray:
    b = bits.astype(np.int32)
    return np.where(b & 0x8000, -(b & 0x7FFF), b & 0x7FFF)


def check(dt, got, ref, rel=1e-3, max_ulp=1, max_frac=0.02, what="", scale_aware=True):
    got = np.asarray(got).reshape(-1)
    ref = np.asarray(ref).reshape(-1)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    if dt == F32:
        a, b = got.astype(np.float64), ref.astype(np.float64)
        assert np.all(np.isfinite(a)), what
        rms = np.sqrt(np.mean(b * b)) if b.size else 0.0
        err = np.abs(a - b)
        bound = rel * (np.abs(b) + rms) + 1e-30
        worst = float(np.max(err / bound)) if b.size else 0.0
        assert worst <= 1.0, f"{what}: max err/bound = {worst:.3g} (rel {rel})"
        return dict(max_rel=float(np.max(err / (np.abs(b) + rms + 1e-30))) if b.size else 0.0)
    a, b = mo.from_bf16(got).astype(np.float64), mo.from_bf16(ref).astype(np.float64)
    assert np.all(np.isfinite(a)), what
    d = np.abs(bf16_ordinal(got) - bf16_ordinal(ref))
    frac = float(np.mean(d != 0)) if d.size else 0.0
    mx = int(d.max()) if d.size else 0
    nrm = float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))
    if scale_aware:
        rms = np.sqrt(np.mean(b * b)) if b.size else 0.0
        bound = max_ulp * 2.0 ** -7 * np.maximum(np.abs(b), rms) + 1e-30
        worst = float(np.max(np.abs(a - b) / bound)) if b.size else 0.0
        assert worst <= 1.0, f"{what}: max |a-b| / ({max_ulp} ulp at max(|b|, rms)) = {worst:.3g}"
    else:
        assert mx <= max_ulp, f"{what}: {mx} bf16 steps from the oracle (allowed {max_ulp})"
    assert frac <= max_frac, f"{what}: {frac:.4f} of elements differ (allowed {max_frac})"
    assert nrm <= rel, f"{what}: normwise rel error {nrm:.3g} > {rel}"
    return dict(max_ulp=mx, frac=frac, normwise=nrm)


def exact(got, ref, what=""):
    assert np.array_equal(np.asarray(got), np.asarray(ref)), f"{what}: not bit-identical"
