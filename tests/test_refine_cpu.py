"""cylindertag_amd/csrc/ctag_refine.h on the host: the fast form of edgeRefine's per-sample normal search (32.32
fixed-point pixel walk + prefix-sum moments) returns the same BITS as the reference arithmetic (search_exact) and as the
literal reference loop (corner_detector.cpp:627-649) -- or declines; the middle form (reference coordinates + prefix sums) always does.  The oracle library hosts the probe; the oracle's own
edgeRefine never calls the product header."""
import ctypes as C
import os

import numpy as np

from ctag_testlib import GOLDEN, read_bmp_gray


def _probe(oracle, img, subpix, samples):
    img = np.ascontiguousarray(img, np.uint8)
    s = np.ascontiguousarray(samples, np.float64).reshape(-1, 4)
    n = len(s)
    exact, fast, lit, mid = np.zeros((n, 2)), np.zeros((n, 2)), np.zeros((n, 2)), np.zeros((n, 2))
    flag = np.zeros(n, np.int32)
    f = oracle.L.ctago_refine_probe
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_ssize_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    f(img.ctypes.data, img.shape[0], img.shape[1], img.strides[0], subpix, n, s.ctypes.data, exact.ctypes.data, fast.ctypes.data,
      lit.ctypes.data, flag.ctypes.data, mid.ctypes.data)
    _probe.mid = mid  # {Mn, Mcount} of search_mid where flag >= 0
    return exact, fast, lit, flag


def _same_bits(a, b):
    return a.view(np.uint64) == b.view(np.uint64)


def test_weights_are_multiples_of_2_pow_minus_39():
    """The exactness argument of the prefix-sum form: every weight (g2 - g1)^2, evaluated in float as the reference does
    (:639-645), is a multiple of 2^-39 and at most 1, for all 256 x 256 pixel pairs."""
    u = np.arange(256, dtype=np.float32) * np.float32(1.0 / 255)
    d = (u[None, :] - u[:, None]).astype(np.float32)
    w = (d * d).astype(np.float32).astype(np.float64)
    assert w.max() <= 1.0
    scaled = w * 2.0 ** 39
    assert (scaled == np.floor(scaled)).all()
    assert scaled[w > 0].min() >= 1.0


def test_fast_search_equals_reference_arithmetic(oracle):
    bmp = read_bmp_gray(os.path.join(GOLDEN, "test.bmp"))
    rng = np.random.RandomState(5)
    noise = rng.randint(0, 256, (480, 640)).astype(np.uint8)
    total_fast = 0
    for img in (bmp, noise):
        rows, cols = img.shape
        for subpix in (1, 3, 5, 8):
            n = 40000
            th = rng.uniform(0, 2 * np.pi, n)
            s = np.stack([rng.uniform(-8, cols + 8, n), rng.uniform(-8, rows + 8, n), np.cos(th), np.sin(th)], 1)
            s[:200, 2:] = [[1, 0], [0, 1], [-1, 0], [0, -1]] * 50          # axis-aligned normals
            s[:100, 0] = np.round(s[:100, 0])                               # exactly integral coordinates: the guard must fire
            s[100:200, 1] = np.round(s[100:200, 1] * 4) / 4
            exact, fast, lit, flag = _probe(oracle, img, subpix, s)
            assert _same_bits(exact, lit).all()                             # ring form == literal loop
            ok = flag == 1
            assert _same_bits(exact[ok], fast[ok]).all()
            app = flag >= 0                                                  # the middle form never declines
            assert _same_bits(exact[app], _probe.mid[app]).all() and (flag[:200][app[:200]] == 0).any()
            assert ok.sum() > 0.8 * (flag >= 0).sum() and (flag >= 0).sum() > 0.9 * n * (1 - 40.0 / min(rows, cols))
            total_fast += int(ok.sum())
    assert total_fast > 200000


def test_guard_band_and_applicability(oracle):
    bmp = read_bmp_gray(os.path.join(GOLDEN, "test.bmp"))
    rows, cols = bmp.shape
    # coordinates that land within 2^-22 px of an integer somewhere along the walk: the fast form must decline (flag 0) or agree
    base = np.array([[400.0, 300.37, 1.0, 0.0]])
    eps = np.array([0.0, 2.0 ** -30, -2.0 ** -30, 2.0 ** -24, -2.0 ** -24, 2.0 ** -23, 2.0 ** -21, 0.3])
    s = np.repeat(base, len(eps), 0)
    s[:, 0] += eps
    exact, fast, lit, flag = _probe(oracle, bmp, 5, s)
    assert list(flag[:6]) == [0] * 6 and flag[7] == 1 and _same_bits(exact, lit).all()
    assert _same_bits(exact, _probe.mid).all()
    assert _same_bits(exact[flag == 1], fast[flag == 1]).all()
    # not interior (search leaves the image) or window too large for the exactness bound: not applicable
    s = np.array([[3.0, 300.0, 1.0, 0.0], [400.0, rows - 2.0, 0.0, 1.0], [400.1, 300.2, np.cos(1.0), np.sin(1.0)]])
    _, _, _, flag = _probe(oracle, bmp, 5, s)
    assert list(flag) == [-1, -1, 1]
    _, _, _, flag = _probe(oracle, bmp, 9, s[2:])
    assert list(flag) == [-1]
