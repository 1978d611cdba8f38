"""Primitive-level pins of the CPU oracle against a REAL OpenCV, wherever one is importable (`import cv2`).

The reference cannot be built in the build image (needs OpenCV + Ceres + Eigen + glog) and no cv2 wheel is installed there, so the
oracle's replicas of the OpenCV 4.5.3 primitives (all tagged [OCV-recall]) are "parity unpinned".  These
comparisons retire the tags one by one on any host that does have OpenCV: tests/test_oracle_vs_cv2_cpu.py asserts them (and skips
without cv2) and bench.py's opencv_stage_probe runs them on the GPU box and prints the mismatches.  Call sites in the reference:
  resize(img, Size(cols/2, rows/2), 0.5, 0.5, INTER_CUBIC)                         CylinderTag.cpp:79
  connectedComponentsWithStats(src, labels, stats, centroids, 8, 4, CCL_BBDT)      corner_detector.cpp:82
  fitLine(points, line, DIST_L2, 0, 0.01, 0.01)                                   corner_detector.cpp:136 (expand_line)
  fitLine(points, line, DIST_WELSCH, 0, 0.01, 0.01)                               corner_detector.cpp:358
  fastAtan2(y, x)                                                                  corner_detector.cpp:1028-1031
  cvtColor(frame, gray, COLOR_BGR2GRAY)                                            main.cpp:36,54
Test infrastructure only (it drives oracle/)."""
import numpy as np


def _images(test_bmp):
    """Inputs of the resize pin: the reference's own frame, crops of it at sizes whose half width is 0..7 mod 8 and at odd sizes
    (the general tap tables), and synthetic patterns built to land on the vertical pass's exact .5 rounding ties."""
    rng = np.random.RandomState(3)
    out = [("test.bmp 1920x1200", test_bmp), ("test.bmp cropped to 1920x1080", test_bmp[60:1140])]
    for dw in range(8):  # half widths 480..487: every length of the row tail of the vector body (hcols % 8)
        out.append(("crop %dx540" % (960 + 2 * dw), np.ascontiguousarray(test_bmp[100:640, 200:200 + 960 + 2 * dw])))
    for (h, w) in ((541, 963), (301, 5), (5, 301), (4, 4), (7, 9), (1081, 1921)):
        out.append(("odd %dx%d" % (w, h), np.ascontiguousarray(test_bmp[:h, :w]) if h <= test_bmp.shape[0] and w <= test_bmp.shape[1]
                    else rng.randint(0, 256, (h, w)).astype(np.uint8)))
    # ties: with taps (-192, 1216, 1216, -192) / 2048 per axis a block of constant columns whose four source rows are
    # (a, b, b, a) gives (19 b - 3 a) / 16 -- choose values whose result ends in exactly .5
    tie = np.zeros((64, 96), np.uint8)
    for y in range(64):
        for x in range(96):
            a, b = (8 * (x // 2) + y) % 256, (8 * (x // 2) + 3 * y + 8) % 256
            tie[y, x] = b if (y % 4) in (1, 2) else a
    out.append(("tie pattern 96x64", tie))
    out.append(("noise 1024x64", rng.randint(0, 256, (64, 1024)).astype(np.uint8)))
    out.append(("two-level noise 968x66", (rng.randint(0, 2, (66, 968)) * 255).astype(np.uint8)))
    return out


def pin_resize(cv2, orc, test_bmp):
    res = {"call": "cv2.resize(img, (cols//2, rows//2), 0.5, 0.5, cv2.INTER_CUBIC) vs ctago_resize_half", "cases": 0, "mismatching_cases": [],
           "pixels": 0, "mismatching_pixels": 0, "max_abs_diff": 0}
    for name, img in _images(test_bmp):
        want = cv2.resize(img, (img.shape[1] // 2, img.shape[0] // 2), None, 0.5, 0.5, cv2.INTER_CUBIC)
        got = orc.resize_half(img)
        bad = int((want != got).sum())
        res["cases"] += 1
        res["pixels"] += int(want.size)
        res["mismatching_pixels"] += bad
        if bad:
            d = np.abs(want.astype(np.int32) - got.astype(np.int32))
            res["max_abs_diff"] = max(res["max_abs_diff"], int(d.max()))
            ys, xs = np.nonzero(want != got)
            res["mismatching_cases"].append({"case": name, "pixels": bad, "first": [int(ys[0]), int(xs[0])],
                                             "columns_mod8_of_mismatches": sorted(set(int(x) % 8 for x in xs[:2000])),
                                             "in_row_tail": bool((xs >= (want.shape[1] & ~7)).all())})
    return res


def _binaries(orc, test_bmp):
    rng = np.random.RandomState(4)
    half = orc.resize_half(test_bmp)
    out = [("test.bmp thresholded", orc.threshold(half, 5))]
    out.append(("noise 40 % 301x200", ((rng.rand(200, 301) < 0.4) * 255).astype(np.uint8)))
    out.append(("noise 60 % 64x64", ((rng.rand(64, 64) < 0.6) * 255).astype(np.uint8)))
    out.append(("noise 25 % odd 77x53", ((rng.rand(53, 77) < 0.25) * 255).astype(np.uint8)))
    # label-order corner case of App. A.4: a component starting on the odd row of a 2x2 block row at smaller x precedes one that
    # starts on the even row above at larger x
    a = np.zeros((8, 12), np.uint8)
    a[1, 0] = 255
    a[0, 6] = 255
    a[5, 2:4] = 255
    a[4, 9] = 255
    out.append(("block-raster order", a))
    b = np.zeros((40, 40), np.uint8)
    b[::2, ::2] = 255  # isolated pixels on a lattice: one label per 2x2 block
    out.append(("dot lattice", b))
    c = np.zeros((33, 47), np.uint8)
    for k in range(0, 30, 3):  # diagonal staircases that merge late (equivalences resolved by the flatten pass)
        for t in range(30):
            if 0 <= k + t < 33 and t < 47:
                c[k + t if (k // 3) % 2 == 0 else 32 - (k + t) % 33, t] = 255
    out.append(("staircases", c))
    return out


def pin_ccl(cv2, orc, test_bmp):
    res = {"call": "cv2.connectedComponentsWithStatsWithAlgorithm(bin, 8, cv2.CV_32S, cv2.CCL_BBDT) vs ctago_ccl (label ORDER and areas)",
           "cases": 0, "mismatching_cases": []}
    for name, binary in _binaries(orc, test_bmp):
        n, labels, stats, _ = cv2.connectedComponentsWithStatsWithAlgorithm(binary, 8, cv2.CV_32S, cv2.CCL_BBDT)
        got_labels, got_areas = orc.ccl(binary)
        res["cases"] += 1
        ok = n == len(got_areas) and (labels == got_labels).all() and (stats[:, cv2.CC_STAT_AREA] == got_areas).all()
        if not ok:
            same_partition = n == len(got_areas) and len(set(zip(labels.ravel().tolist(), got_labels.ravel().tolist()))) == n
            res["mismatching_cases"].append({"case": name, "cv2_labels": int(n), "oracle_labels": int(len(got_areas)),
                                             "same_partition_other_order": bool(same_partition)})
    return res


def _point_sets(orc, test_bmp):
    rng = np.random.RandomState(6)
    sets = []
    for n in (2, 3, 5, 9, 10, 11, 17, 40, 135, 450):
        for k in range(6):
            t = rng.uniform(0, np.pi)
            s = np.arange(n) - n / 2.0
            x = 300 + s * np.cos(t) + rng.normal(0, 0.4, n)
            y = 200 + s * np.sin(t) + rng.normal(0, 0.4, n)
            if k >= 4 and n >= 9:  # outliers: the case DIST_WELSCH exists for
                x[::4] += rng.uniform(-6, 6, len(x[::4]))
            sets.append(np.stack([np.round(x), np.round(y)], 1).astype(np.int32))
    sets.append(np.array([[5, 5], [5, 9]], np.int32))                  # vertical
    sets.append(np.array([[5, 5], [9, 5], [13, 5]], np.int32))         # horizontal
    sets.append(np.array([[1, 1], [1, 1], [1, 1]], np.int32))          # degenerate
    return sets


def pin_fitline(cv2, orc, test_bmp):
    res = {}
    for welsch, dist_type, name in ((False, cv2.DIST_L2, "l2"), (True, cv2.DIST_WELSCH, "welsch")):
        r = {"call": "cv2.fitLine(points int32, cv2.DIST_%s, 0, 0.01, 0.01) vs ctago_fitline_%s" % (name.upper(), name), "cases": 0,
             "bitwise_equal": 0, "max_abs_diff": 0.0, "worst_case_points": None}
        for pts in _point_sets(orc, test_bmp):
            want = cv2.fitLine(pts.reshape(-1, 1, 2), dist_type, 0, 0.01, 0.01).ravel().astype(np.float32)
            got = orc.fitline(pts, welsch)
            r["cases"] += 1
            if want.tobytes() == got.tobytes():
                r["bitwise_equal"] += 1
            else:
                d = float(np.abs(want.astype(np.float64) - got.astype(np.float64)).max())
                if not np.isfinite(d):
                    d = float("inf")
                if d > r["max_abs_diff"]:
                    r["max_abs_diff"], r["worst_case_points"] = d, int(len(pts))
        res[name] = r
    return res


def pin_variants(cv2, orc, test_bmp):
    """Which of the two switchable assumptions (oracle/ctag_oracle.cpp: OracleVariants) this OpenCV matches: the place of fitLine2D's `if (err < min_err)`
    and the width of resize's vector body.  The answer names the build flag to flip (EXTRA=-DCTAG_WELSCH_MINERR_IN_LOOP / -DCTAG_RESIZE_SIMD_LANES=16)."""
    out = {"welsch_minerr": {}, "resize_simd_lanes": {}}
    sets = [p for p in _point_sets(orc, test_bmp) if len(p) >= 9]
    for variant in (0, 1):
        eq, worst = 0, 0.0
        for pts in sets:
            want = cv2.fitLine(pts.reshape(-1, 1, 2), cv2.DIST_WELSCH, 0, 0.01, 0.01).ravel().astype(np.float32)
            got = orc.fitline_welsch_variant(pts, variant)
            eq += want.tobytes() == got.tobytes()
            d = float(np.abs(want.astype(np.float64) - got.astype(np.float64)).max())
            worst = max(worst, d if np.isfinite(d) else float("inf"))
        out["welsch_minerr"]["after_the_loop (default)" if variant == 0 else "in_the_loop (-DCTAG_WELSCH_MINERR_IN_LOOP)"] = {
            "cases": len(sets), "bitwise_equal": int(eq), "max_abs_diff": worst}
    discriminating = sum(orc.fitline_welsch_variant(p, 0).tobytes() != orc.fitline_welsch_variant(p, 1).tobytes() for p in sets)
    out["welsch_minerr"]["cases_where_the_variants_differ"] = int(discriminating)
    imgs = [(n, im) for n, im in _images(test_bmp) if (im.shape[1] // 2) % 16 >= 8]
    for lanes in (8, 16):
        orc.set_variants(0, lanes)
        try:
            bad = sum(int((cv2.resize(im, (im.shape[1] // 2, im.shape[0] // 2), None, 0.5, 0.5, cv2.INTER_CUBIC) != orc.resize_half(im)).sum()) for _, im in imgs)
        finally:
            orc.set_variants(0, 8)
        out["resize_simd_lanes"]["%d%s" % (lanes, " (default)" if lanes == 8 else " (-DCTAG_RESIZE_SIMD_LANES=16)")] = {"cases": len(imgs), "mismatching_pixels": int(bad)}
    return out


def pin_fast_atan2(cv2, orc):
    rng = np.random.RandomState(8)
    y = np.concatenate([rng.uniform(-100, 100, 4000), [0, 0, 1, -1, 0, 5, -5, 1e-6]]).astype(np.float32)
    x = np.concatenate([rng.uniform(-100, 100, 4000), [1, -1, 0, 0, 0, 5, 5, -1e6]]).astype(np.float32)
    want = np.array([cv2.fastAtan2(float(a), float(b)) for a, b in zip(y, x)], np.float32)
    got = orc.math(9, y.astype(np.float64), x.astype(np.float64)).astype(np.float32)
    bad = want != got
    return {"call": "cv2.fastAtan2(y, x) vs ctm::fast_atan2_deg", "cases": int(len(y)), "mismatches": int(bad.sum()),
            "max_abs_diff_deg": float(np.abs(want - got).max())}


def pin_bgr2gray(cv2, orc, test_bmp):
    rng = np.random.RandomState(9)
    imgs = [rng.randint(0, 256, (97, 131, 3)).astype(np.uint8), np.stack([test_bmp[:300, :400], test_bmp[100:400, 50:450], test_bmp[200:500, 300:700]], 2)]
    ramp = np.zeros((256, 3 * 256, 3), np.uint8)  # every value of each channel against two levels of the others
    for c in range(3):
        ramp[:, c * 256:(c + 1) * 256, c] = np.arange(256)[None, :]
        ramp[:128, c * 256:(c + 1) * 256, (c + 1) % 3] = 255
    imgs.append(ramp)
    bad = total = 0
    for im in imgs:
        want = cv2.cvtColor(im, cv2.COLOR_BGR2GRAY)
        got = orc.bgr2gray(im)
        bad += int((want != got).sum())
        total += int(want.size)
    return {"call": "cv2.cvtColor(bgr, cv2.COLOR_BGR2GRAY) vs ctago_bgr2gray", "pixels": total, "mismatching_pixels": bad}


def run_all(orc, test_bmp):
    """Every pin; raises ImportError without cv2.  Returns a JSON-able report."""
    import cv2
    rep = {"opencv_version": cv2.__version__, "reference_pins_opencv": "4.5.3 (Release.props:11)"}
    try:
        rep["cpu_features"] = [l.strip() for l in cv2.getBuildInformation().splitlines() if "Baseline:" in l or "Dispatched code" in l][:3]
    except Exception:  # noqa: BLE001
        pass
    rep["resize"] = pin_resize(cv2, orc, test_bmp)
    rep["ccl"] = pin_ccl(cv2, orc, test_bmp)
    rep["fitline"] = pin_fitline(cv2, orc, test_bmp)
    rep["variants"] = pin_variants(cv2, orc, test_bmp)
    rep["fast_atan2"] = pin_fast_atan2(cv2, orc)
    rep["bgr2gray"] = pin_bgr2gray(cv2, orc, test_bmp)
    return rep
