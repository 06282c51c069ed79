"""extract_times / resample_time_serie (csrc/time_series.cpp, host bookkeeping - no device needed): the library, the oracle's
restatement and, where it was built, the compiled reference against the golden vectors made from the reference
(tests/golden/make_labelling_golden.py); the reference wrapper's error behaviour (reference tests/python/test_rir.py:232-262)."""
import ctypes as ct
import os

import numpy as np
import pytest

from cases import resample_cases, time_axis_cases
from oracle.pyoracle import _SignalProcessingMixin

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def series_golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "time_series.npz"))


class _Product(_SignalProcessingMixin):
    """The product's C entry points through the same ctypes shims as the oracle's."""

    def __init__(self, lib):
        self.lib = lib


@pytest.fixture(scope="module")
def product(lib):
    return _Product(lib)


def _same(a, b):
    return a.shape == b.shape and a.tobytes() == b.tobytes()  # bit for bit (the sign of a zero, a NaN's payload)


@pytest.mark.parametrize("who", ["oracle", "product", "ref"])
def test_time_axes_against_the_reference_vectors(who, request, series_golden):
    impl = request.getfixturevalue(who)
    for name, vs, s in time_axis_cases():
        rc, out = impl.extract_times(vs, s)
        assert rc == 0, name
        assert _same(out, series_golden["axis_" + name]), name


@pytest.mark.parametrize("who", ["oracle", "product", "ref"])
def test_resampling_against_the_reference_vectors(who, request, series_golden):
    impl = request.getfixturevalue(who)
    for name, x, y, times, s, padd in resample_cases():
        rc, out = impl.resample_time_serie(x, y, times, s, padd)
        assert rc == 0, name
        assert _same(out, series_golden["resample_" + name]), name


@pytest.mark.parametrize("who", ["oracle", "product"])
def test_output_too_small_and_refusals(who, request):
    impl = request.getfixturevalue(who)
    vs = [np.arange(5.0), np.arange(3.0, 9.0)]
    assert impl.extract_times(vs, 0, room=3) == (-2, 9)  # signal_processing.cpp:171-175: the needed size comes back
    assert impl.resample_time_serie([0, 1], [1, 2], [0, 0.5, 1], 4, 0, room=2) == (-1, 3)  # signal_processing.cpp:187-191
    # inputs the reference never returns from (an empty run stays in its list for ever, Filters.cpp:186-205)
    for vs, s in (([[0.0, 10.0], [3.0, 4.0]], 1),  # no sample of the first vector inside [3, 4]
                  ([[np.nan, 1.0, 2.0], [1.0]], 0), ([[0.0, 1.0, np.nan], [1.0]], 0), ([[0.0, np.nan, 1.0, np.nan, 2.0], [1.0]], 0),
                  ([[], [1.0]], 0)):
        assert impl.extract_times(vs, s)[0] == -1


def test_wrappers_like_the_reference_tests():
    """reference tests/python/test_rir.py:232-262, the same calls"""
    from librir_amd import signal_processing as sp

    times1 = [0, 0.2, 1, 1.5, 2.3, 3.3, 4, 5]
    times2 = [-1, 3, 4, 4.3, 4.7]
    assert np.array_equal(sp.extract_times((times1, times2), "union"), [-1, 0, 0.2, 1, 1.5, 2.3, 3, 3.3, 4, 4.3, 4.7, 5])
    assert np.array_equal(sp.extract_times((times1, times2), "inter"), [0, 0.2, 1, 1.5, 2.3, 3, 3.3, 4, 4.3, 4.7])
    with pytest.raises(RuntimeError):
        sp.extract_times((), "inter")
    with pytest.raises(RuntimeError):
        sp.extract_times((times1, times2), "whatever")
    assert sp.extract_times((times1, range(10000)), "union").size == 10004
    with pytest.raises(RuntimeError):  # the reference does not come back from this one
        sp.extract_times(([0, 10.0], [3, 4.0]), "inter")

    x = range(10)
    y = range(10)
    times = [0, 0.2, 1, 1.5, 2.3, 3.3, 4, 5, 5.6, 9.9, 10, 12, 13]
    assert np.allclose(sp.resample_time_serie(x, y, times), [0, 0.2, 1, 1.5, 2.3, 3.3, 4, 5, 5.6, 9, 9, 9, 9], rtol=0, atol=1e-12)
    assert np.allclose(sp.resample_time_serie(x, y, times, 0), [0, 0.2, 1, 1.5, 2.3, 3.3, 4, 5, 5.6, 0, 0, 0, 0], rtol=0, atol=1e-12)
    assert np.array_equal(sp.resample_time_serie(x, y, times, None, False), [0, 0, 1, 2, 2, 3, 4, 5, 6, 9, 9, 9, 9])
    with pytest.raises(RuntimeError):
        sp.resample_time_serie([], y, times)
    with pytest.raises(RuntimeError):
        sp.resample_time_serie(x, y, [])
    with pytest.raises(RuntimeError):
        sp.resample_time_serie(list(x) + list(x), y, times)
    # a new axis more than twice as long as the series (the reference wrapper's output room ends there)
    assert sp.resample_time_serie([0, 1], [0, 10], np.linspace(0, 1, 11)).size == 11


def test_random_inputs_library_against_oracle(oracle, product):
    rng = np.random.default_rng(99)
    for it in range(1500):
        nv = int(rng.integers(1, 5))
        vs = [np.sort(rng.integers(0, 60, int(rng.integers(1, 30))) * 0.5) for _ in range(nv)]
        s = int(rng.integers(0, 2))
        if it % 4 == 0:
            for v in vs:
                if v.size >= 3 and rng.random() < 0.5:
                    v[int(rng.integers(1, v.size - 1))] = np.nan
        if it % 7 == 0:
            vs = [rng.permutation(v) if not np.isnan(v).any() else v for v in vs]
        p, o = product.extract_times(vs, s), oracle.extract_times(vs, s)
        assert p[0] == o[0]
        if p[0] == 0:
            assert _same(p[1], o[1])
        n = int(rng.integers(0, 20))
        x = np.sort(rng.integers(0, 40, n) * 0.25)
        y = rng.normal(size=n)
        t = rng.integers(-8, 48, int(rng.integers(1, 40))) * 0.25
        if it % 5:
            t = np.sort(t)
        st, padd = int(rng.choice([0, 2, 4, 6])), float(rng.normal())
        p, o = product.resample_time_serie(x, y, t, st, padd), oracle.resample_time_serie(x, y, t, st, padd)
        assert p[0] == o[0] == 0 and _same(p[1], o[1])
