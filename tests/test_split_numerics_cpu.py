"""CPU pin of the split-precision formats' error bounds (oracle/split16_oracle.py restates csrc/split16.h in numpy)."""
import numpy as np
import pytest

from oracle import split16_oracle as S


@pytest.mark.parametrize("fmt,bound", [("f16x3", 2.0 ** -21), ("bf16x6", 2.0 ** -22), ("bf16x3", 2.0 ** -14)])
@pytest.mark.parametrize("kind", ["normal", "heavy", "tiny", "huge"])
def test_dropped_term_bound(fmt, bound, kind):
    """|split dot - exact dot| <= bound * sum |a_k b_k| (the per-product bounds of the header comment, with slack 2)."""
    rng = np.random.default_rng(7)
    for K in (16, 1152, 4608):
        a = rng.standard_normal(K).astype(np.float32)
        b = rng.standard_normal(K).astype(np.float32)
        if kind == "heavy":
            a[rng.integers(0, K, 3)] *= 1e4
            b[rng.integers(0, K, 3)] *= 1e-4
        elif kind == "tiny":
            a *= np.float32(3e-22)
        elif kind == "huge":
            a *= np.float32(7e17)
            b *= np.float32(1e-12)
        exact = float(np.dot(a.astype(np.float64), b.astype(np.float64)))
        mag = float(np.dot(np.abs(a).astype(np.float64), np.abs(b).astype(np.float64)))
        assert abs(S.dot(a, b, fmt) - exact) <= bound * mag, (fmt, kind, K)


def test_fp16_scale_keeps_planes_finite_and_floor():
    """Every |x| c stays below 2^14; elements down to 2^-16 of the maximum keep 2^-22 relative precision, smaller ones are
    represented to 2^-38 of the maximum (the absolute floor quoted in csrc/convsplit.hip)."""
    rng = np.random.default_rng(3)
    for amax in (1e-30, 3.7e-5, 1.0, 65504.0, 2.9e19):
        x = (rng.standard_normal(4096) * amax / 4).astype(np.float32)
        x[0] = amax
        x[1:64] *= np.float32(2.0 ** -30)          # elements far below the maximum
        planes, c = S.split(x, "f16x3")
        assert np.all(np.isfinite(planes[0])) and float(np.abs(x).max()) * float(c) < 2.0 ** 14
        rec = (planes[0].astype(np.float64) + planes[1].astype(np.float64)) / float(c)
        err = np.abs(rec - x.astype(np.float64))
        tol = np.maximum(2.0 ** -22 * np.abs(x.astype(np.float64)), 2.0 ** -38 * float(np.abs(x).max()))
        assert np.all(err <= tol)
    planes, c = S.split(np.zeros(8, np.float32), "f16x3")
    assert float(c) == 1.0 and not planes[0].any()


def test_bf16_three_planes_are_exact():
    """3 bf16 planes hold all 24 significand bits of an fp32 value (x0 + x1 + x2 == x)."""
    x = np.random.default_rng(0).standard_normal(10000).astype(np.float32) * np.float32(123.456)
    planes, _ = S.split(x, "bf16x6")
    assert np.array_equal((planes[0].astype(np.float64) + planes[1] + planes[2]).astype(np.float32), x)
