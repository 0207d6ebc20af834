"""CPU check of the identity behind the demapper's phase detector (dabstar_amd/csrc/ofdm_core.h, phase_offset_from_diagonal):

    fmod(arg(b) [+ pi if negative], pi/2) - pi/4   (ofdm_decoder.cpp:197-202, glob_defs.h:173-182)
  = sign * (pi/4 - r),   r = atan(min(|x|,|y|) / max(|x|,|y|)),
    sign = -1 unless [(|y| > |x|) xor (x < 0) xor (y < 0)], and -1 on the axes / at the origin (either sign of zero),

which is what the device code computes from one octant arctangent and three sign bits.  This is a test of the DESIGN (the device
code itself is compared with the oracle on the GPU: tests/test_gpu_stages.py::test_demapper_phase_detector_on_the_axes and
every engine test); it documents where the two sides of the identity could part -- the axes -- and that they do not."""
import numpy as np


def _reference(y, x):
    ph = np.arctan2(y, x)
    ph = np.where(ph < 0, ph + np.pi, ph)
    return np.fmod(ph, np.pi / 2) - np.pi / 4


def _device_formula(y, x):
    ax, ay = np.abs(x), np.abs(y)
    mx = np.maximum(np.maximum(ax, ay), np.finfo(np.float32).tiny)
    t = np.minimum(ax, ay) / mx
    u = np.arctan(t) - np.pi / 4                                  # <= 0
    d = ay - ax
    neg = np.signbit(d) ^ np.signbit(x) ^ np.signbit(y)           # sign bit of the result before the axis rule
    neg = neg | (t == 0)                                          # on an axis / at the origin: forced negative
    return np.where(neg, -np.abs(u), np.abs(u))


def test_identity_on_random_points():
    rng = np.random.default_rng(1)
    y, x = rng.normal(size=200000), rng.normal(size=200000)
    assert np.max(np.abs(_reference(y, x) - _device_formula(y, x))) < 1e-12


def test_identity_on_the_axes_and_at_the_origin():
    z = np.array([0.0, -0.0])
    v = np.array([1.5, -1.5, 1e-30, -1e-30])
    ys, xs = [], []
    for a in z:                       # origin, both zero signs each
        for b in z:
            ys.append(a); xs.append(b)
    for a in z:                       # on the x axis (y = +-0) and on the y axis (x = +-0)
        for b in v:
            ys.append(a); xs.append(b)
            ys.append(b); xs.append(a)
    y, x = np.array(ys), np.array(xs)
    ref = _reference(y, x)
    # the reference lands on fmod(k pi/2, pi/2) = 0 (up to the rounding of pi in double: |.| < 1e-15) -> -pi/4 everywhere
    assert np.allclose(ref, -np.pi / 4, atol=1e-12), ref
    assert np.array_equal(_device_formula(y, x), np.full(len(y), -np.pi / 4))


def test_diagonals_are_the_zero_of_the_detector():
    a = np.array([1.0, 2.5, 1e-3])
    for sy in (1, -1):
        for sx in (1, -1):
            assert np.allclose(_device_formula(sy * a, sx * a), 0.0, atol=1e-15)
            assert np.allclose(_reference(sy * a, sx * a), 0.0, atol=1e-12)
