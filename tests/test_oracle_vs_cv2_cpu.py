"""Pins of the oracle's OpenCV replicas against a real OpenCV (tests/cv2_pins.py).  Skips where `import cv2` fails -- the build
image and the GPU boxes of this pool have no OpenCV, so until a host with one runs this file the oracle stays "parity unpinned"
(DESIGN.md 2).  bench.py's opencv_stage_probe runs the same comparisons on whatever host it executes on and prints the report."""
import pytest

cv2 = pytest.importorskip("cv2", reason="no OpenCV importable: the [OCV-recall] primitives stay unpinned on this host")

import cv2_pins  # noqa: E402


@pytest.fixture(scope="module")
def report(oracle, test_bmp):
    return cv2_pins.run_all(oracle, test_bmp)


def test_resize_half_equals_cv2_resize_inter_cubic(report):
    """SURVEY App. A.1 incl. the SIMD-width assumption: the oracle rounds the vector body (columns below hcols & ~7)
    half-to-even and the scalar row tail half-up, as a 128-bit-baseline build of OpenCV does.  A wider baseline (AVX2: 16-lane
    body) would show up here as mismatches confined to columns [hcols & ~15, hcols & ~7) of exact-tie pixels."""
    r = report["resize"]
    assert r["mismatching_pixels"] == 0, r


def test_ccl_label_order_equals_cv2_bbdt(report):
    """labels in block-raster order of each component's first 2x2 block, areas equal."""
    assert not report["ccl"]["mismatching_cases"], report["ccl"]


def test_fitline_l2_equals_cv2(report):
    r = report["fitline"]["l2"]
    assert r["bitwise_equal"] == r["cases"], r


def test_fitline_welsch_equals_cv2(report):
    """SURVEY App. A.6 (cv::RNG replay, restart selection).  Bitwise equality is expected where OpenCV's std::exp(float) rounds as
    ctag_math.h's exp32 does (both <= 1 ulp); the hard bar is 1e-4 on direction and point."""
    r = report["fitline"]["welsch"]
    assert r["max_abs_diff"] <= 1e-4, r
    assert r["bitwise_equal"] >= 0.9 * r["cases"], r


def test_fast_atan2_equals_cv2(report):
    assert report["fast_atan2"]["mismatches"] == 0, report["fast_atan2"]


def test_bgr2gray_equals_cv2(report):
    assert report["bgr2gray"]["mismatching_pixels"] == 0, report["bgr2gray"]
