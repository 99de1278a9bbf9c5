"""Accuracy of the f32-only sin / cos / asin / acos / atan2 shared by the kernels and the
oracle's portable mode, against float64 references, over the ranges the renderer uses."""
import numpy as np
import pytest

SIN, COS, ACOS, ASIN, ATAN2 = range(5)


def ulp_error(got, want64):
    want32 = want64.astype(np.float32)
    spacing = np.spacing(np.abs(want32)).astype(np.float64)
    return np.abs(got.astype(np.float64) - want64) / np.maximum(spacing, 2.0 ** -149)


@pytest.mark.parametrize("fn,ref,lo,hi", [(SIN, np.sin, 0.0, 2 * np.pi), (COS, np.cos, 0.0, 2 * np.pi),
                                          (SIN, np.sin, -40.0, 40.0), (COS, np.cos, -40.0, 40.0)])
def test_sin_cos(oracle, fn, ref, lo, hi):
    x = np.linspace(lo, hi, 2_000_001).astype(np.float32)
    got = oracle.portable_math(fn, x)
    want = ref(x.astype(np.float64))
    # absolute error: near zeros of the function an ulp of the result is tiny, the
    # argument-reduction residue (~1e-8) dominates
    assert np.max(np.abs(got - want)) < 1.5e-7
    big = np.abs(want) > 0.05
    assert np.max(ulp_error(got[big], want[big])) <= 2.0


def test_asin_acos(oracle):
    x = np.linspace(-1.0, 1.0, 2_000_001).astype(np.float32)
    for fn, ref in ((ASIN, np.arcsin), (ACOS, np.arccos)):
        got = oracle.portable_math(fn, x)
        want = ref(x.astype(np.float64))
        assert np.max(np.abs(got - want)) < 4e-7
    assert np.isnan(oracle.portable_math(ASIN, np.array([1.0000001, -2.0, np.nan], np.float32))).all()
    assert oracle.portable_math(ACOS, np.array([1.0], np.float32))[0] == 0.0
    assert abs(oracle.portable_math(ACOS, np.array([-1.0], np.float32))[0] - np.float32(np.pi)) < 5e-7


def test_atan2(oracle):
    rng = np.random.default_rng(0)
    y = rng.normal(size=1_000_000).astype(np.float32)
    x = rng.normal(size=1_000_000).astype(np.float32)
    got = oracle.portable_math(ATAN2, y, x)
    want = np.arctan2(y.astype(np.float64), x.astype(np.float64))
    assert np.max(np.abs(got - want)) < 5e-7
    edge_y = np.array([0.0, 1.0, -1.0, 0.0, 0.0], np.float32)
    edge_x = np.array([1.0, 0.0, 0.0, -1.0, 0.0], np.float32)
    assert np.allclose(oracle.portable_math(ATAN2, edge_y, edge_x), [0, np.pi / 2, -np.pi / 2, np.pi, 0], atol=1e-7)


def test_cosine_lobe_quirk_is_kept(oracle):
    # cos(f32(pi/2)) is -4.37e-8 for libm and for the portable routine alike, so the reference's
    # cosine-weighted sampler (math/src/lib.rs:292-296) takes the square root of a negative
    # number when its random number is exactly 0: a NaN direction, kept as is.
    x = np.array([np.float32(np.pi) * np.float32(0.5)], np.float32)
    assert oracle.portable_math(COS, x)[0] < 0 and np.cos(x)[0] < 0
