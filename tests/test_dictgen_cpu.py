"""Dictionary generator (tools/dict_gen.py, restating CylinderTag_generator.m: SURVEY.md 8(f) rank 4) -- CPU tests.
The reference's own dictionary is the known answer for the predicates."""
import os
import sys

import numpy as np

from ctag_testlib import GOLDEN, ROOT, read_marker_file

sys.path.insert(0, os.path.join(ROOT, "tools"))
import dict_gen as dg  # noqa: E402


def test_reference_dictionary_satisfies_the_generator_predicates():
    """CTag_2f12c.marker was produced by the reference generator: every code is legal (generator.m:17) and every cyclic
    2-window is unique over the dictionary and its mirror image (testConflict, generator.m:247-286)."""
    state, fs = read_marker_file(os.path.join(GOLDEN, "CTag_2f12c.marker"))
    assert state.shape == (41, 12) and fs == 2
    assert all(dg.legal_code(int(c)) for c in state.ravel())
    assert dg.test_conflict(state, fs)
    # a duplicated window is detected: copy a pair of neighbours of row 0 into row 1
    bad = state.copy()
    bad[1, 3:5] = bad[0, 0:2]
    assert not dg.test_conflict(bad, fs)
    # ... and so is a window that equals the mirror image of another one
    bad = state.copy()
    bad[2, 5], bad[2, 6] = dg.invert_code(int(state[0, 1])), dg.invert_code(int(state[0, 0]))
    assert not dg.test_conflict(bad, fs)


def test_inverse_is_an_involution_and_matches_the_decoder_rule():
    """inverse(): digits inverted and reversed (generator.m:182-196); the per-code rule is the one the decoder applies to a
    reversed marker, (7 - c/8) + (7 - c%8)*8 (corner_detector.cpp match_dictionary)."""
    for fs in (2, 3):
        for v in np.random.RandomState(1).randint(0, 64 ** fs, 500):
            assert dg.inverse_value(dg.inverse_value(int(v), fs), fs) == v
    for c in range(64):
        assert dg.invert_code(c) == (7 - c // 8) + (7 - c % 8) * 8
        assert dg.legal_code(c) == dg.legal_code(dg.invert_code(c))


def test_generated_dictionaries_are_legal_unique_and_loadable(tmp_path):
    import cylindertag_amd as ca
    import testkit as tk
    for col, fs, num, seed in ((12, 2, 24, 3), (15, 2, 12, 4), (9, 3, 4, 5)):
        code = dg.Generator(col, fs, seed=seed, node_budget=40000).generate(num)
        assert code.shape == (num, col), (col, fs, code.shape)
        assert all(dg.legal_code(int(c)) for c in code.ravel())
        assert dg.test_conflict(code, fs)
        path = tmp_path / ("gen_%dc%df.marker" % (col, fs))
        dg.write_marker(str(path), code, fs)
        state, rfs = ca.load_marker_file(str(path))  # the C ABI loader (CylinderTag::load_from_file)
        assert rfs == fs and np.array_equal(state, code)
    # same seed, same dictionary
    a = dg.Generator(12, 2, seed=7).generate(6)
    b = dg.Generator(12, 2, seed=7).generate(6)
    assert np.array_equal(a, b)


def test_strip_writer_round_trip_through_the_oracle(oracle, tmp_path):
    """plot_tag / draw (generator.m:208-245): a dictionary row written as a printable strip (BMP) is read back and decoded by
    detect() to exactly that row with all 12 columns; the gap centres follow the generator's quadratic (draw :227-241)."""
    from ctag_testlib import read_bmp_gray, result_markers
    state, fs = read_marker_file(os.path.join(GOLDEN, "CTag_2f12c.marker"))
    # known answers of block_pos: cross ratio (l0 + l1)(l2 + l1) / (l1 l3) of the column's edge = the id's cross ratio
    for hid in range(8):
        L, p = 1200.0, dg.block_pos(hid, 1200.0)
        l0, l1, l2 = p - 0.1 * L, 0.2 * L, L - p - 0.1 * L
        cr = (l0 + l1) * (l2 + l1) / (l1 * L)
        assert abs(cr - dg.DECODER[hid][0]) < 1e-9 and (p > 0.5 * L) == bool(dg.DECODER[hid][1])
    for row in (0, 17, 40):
        path = str(tmp_path / ("cy%d.bmp" % (row + 1)))
        dg.write_strip_bmp(path, state[row], tag_length=400, ratio=15, margin=(150, 260))
        img = read_bmp_gray(path)
        assert img.shape == (400 + 300, 480 + 520) and img.min() == 0 and img.max() == 255
        res = oracle.detect_fast(img, state, fs)
        ms = result_markers(res)
        assert len(ms) == 1 and ms[0]["marker_id"] == row and sorted(ms[0]["pos"]) == list(range(12))
        by_pos = dict(zip(ms[0]["pos"], ms[0]["id"]))
        inv = [dg.invert_code(int(c)) for c in state[row]]
        assert all(by_pos[p] in (int(state[row][p]), inv[p]) for p in range(12))
    # the reference layout itself (no margin): tag_length x 1.5 * tag_length / ratio * columns, generator.m:212
    assert dg.render_strip(state[0]).shape == (1200, 1440)


def test_other_dictionary_shapes_15c3f_18c4f(oracle):
    """The dictionary shapes the reference's README lists beside 2f12c (15 columns / 3-feature windows, 18 / 4): fixtures written by
    tools/dict_gen.py (seed 3; `python tools/dict_gen.py 15 3 24 ... 3`) satisfy the generator predicates, and synthetic frames planted
    with them decode only to planted rows (feature_size 3 and 4 paths of markerDecoder, corner_detector.cpp:1215)."""
    import cylindertag_amd as ca
    import testkit as tk
    for name, shape, fs in (("CTag_3f15c_gen.marker", (24, 15), 3), ("CTag_4f18c_gen.marker", (24, 18), 4)):
        state, got_fs = read_marker_file(os.path.join(GOLDEN, name))
        assert state.shape == shape and got_fs == fs and all(dg.legal_code(int(c)) for c in state.ravel()) and dg.test_conflict(state, fs)
        exact = 0
        for f in range(4):
            img, truth = tk.synth_frame_host(state, 100 + f)
            res = oracle.detect_fast(img, state, fs)
            planted = sorted(int(x) for x in truth["dict_row"][:truth["n_markers"]])
            found = sorted(int(m["marker_id"]) for m in res["markers"][:res["n_markers"]])
            assert res["status"] == 0 and set(found) <= set(planted) and len(found) >= 2
            exact += planted == found
        assert exact >= 1
