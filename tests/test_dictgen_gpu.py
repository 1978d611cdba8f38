"""Round trip generator -> renderer -> detector on a dictionary that is NOT the reference's: a freshly generated 12-column
dictionary is planted into synthetic frames (ctag_synth, the reference generator's strip geometry) and must come back from
the HIP path -- decoded rows equal the planted ones and every record equals the CPU oracle's byte for byte."""
import os
import sys

import numpy as np
import pytest

import cylindertag_amd as ca
import testkit as tk
from ctag_testlib import ROOT, Oracle

sys.path.insert(0, os.path.join(ROOT, "tools"))
import dict_gen as dg  # noqa: E402

pytestmark = pytest.mark.gpu


def test_generated_dictionary_round_trip():
    code = dg.Generator(12, 2, seed=11).generate(30)
    assert code.shape == (30, 12) and dg.test_conflict(code, 2)
    det, orc = tk.Detector(code, 2, device=0), Oracle()
    exact = 0
    n = 24
    for f in range(n):
        frame, truth = tk.synth_frame_host(code, 700 + f)
        got = det.detect(frame, 5, True, 5)
        want = orc.detect_fast(frame, code, 2, 5, True, 5)
        assert got.tobytes() == want.tobytes(), f
        planted = sorted(int(x) for x in truth["dict_row"][:truth["n_markers"]])
        found = sorted(int(x) for x in got["markers"]["marker_id"][:got["n_markers"]])
        assert set(found) <= set(planted), f  # never a wrong id
        exact += planted == found
    assert exact >= int(0.85 * n)
    det.close()


def test_written_strip_decodes_through_the_cpp_reader_and_the_hip_path(tmp_path):
    """Rank 4 end to end: tools/dict_gen.py writes a dictionary row as a printable strip (plot_tag / draw, generator.m:208-245,
    plus paper margin) -> the C++ BMP reader (ctag_io) -> CylinderTag::detect on the GPU (ctag_demo) -> the written row."""
    import subprocess
    from ctag_testlib import GOLDEN, ROOT, read_marker_file
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import dict_gen as dg
    state, fs = read_marker_file(os.path.join(GOLDEN, "CTag_2f12c.marker"))
    exe = os.path.join(ROOT, "cylindertag_amd", "_build", "ctag_demo")
    for row in (3, 29):
        path = str(tmp_path / ("cy%d.bmp" % (row + 1)))
        dg.write_strip_bmp(path, state[row], tag_length=400, ratio=15, margin=(150, 260))
        out = subprocess.check_output([exe, os.path.join(GOLDEN, "CTag_2f12c.marker"), path], timeout=120).decode()
        lines = [l for l in out.splitlines() if l.startswith("id ")]
        assert out.splitlines()[0] == "markers 1" and len(lines) == 1
        toks = lines[0].split("|")[0].split()
        assert int(toks[1]) == row and int(toks[3]) == 12


@pytest.mark.parametrize("name,fs", [("CTag_3f15c_gen.marker", 3), ("CTag_4f18c_gen.marker", 4)])
def test_other_dictionary_shapes_match_the_oracle(name, fs):
    """15-column / 3-feature and 18-column / 4-feature dictionaries (README of the reference) through the HIP path: records
    byte-identical to the oracle's on planted frames, and the pose back end's 160-point model variant accepts an 18-column model."""
    from ctag_testlib import GOLDEN, read_marker_file
    state, got_fs = read_marker_file(os.path.join(GOLDEN, name))
    assert got_fs == fs
    det, orc = tk.Detector(state, fs, device=0), Oracle()
    frames = np.stack([tk.synth_frame_host(state, 100 + f)[0] for f in range(6)])
    got = det.detect_batch(frames)
    for f in range(6):
        want = orc.detect_fast(frames[f], state, fs)
        assert got[f].tobytes() == want.tobytes(), (name, f)
    assert int(got["n_markers"].sum()) >= 12
    # the same dictionary on cylinders with planted poses: the pose kernel's large point set (18 columns x 8 corners = 144 > 96)
    from pose_testlib import PoseOracle, make_camera, make_model_view, rodrigues
    K = np.array([[2600.0, 0, 960.0], [0, 2600.0, 540.0], [0, 0, 1]])
    M, corners = tk.synth3d_model(state)
    ids = np.arange(state.shape[0], dtype=np.int32)
    mv = make_model_view({"ids": ids, "size": state.shape[1], "base": np.zeros((len(ids), 3), np.float32),
                          "axis": np.zeros((len(ids), 3), np.float32), "corners": corners})
    img, truth = tk.synth3d_frame_host(state, 1, K, rows=1080, cols=1920)
    rec = det.detect(img)
    assert rec.tobytes() == orc.detect_fast(img, state, fs).tobytes()
    poses = det.estimate_pose(rec, M, ca.make_camera(K, np.zeros(5)))
    want = PoseOracle().pose_frame(rec, mv, make_camera(K, np.zeros(5)))
    assert poses.tobytes() == want.tobytes() and len(poses) >= 2
    good = 0
    for p in poses:  # a strip of 15-18 columns wraps up to 2.2 rad of its cylinder: the reference's gap rounding (:1223-1224) can misplace
        # the foreshortened end columns, which shows as a large reprojection error; the well-placed markers must give the planted pose
        if p["status"] != 0 or np.sqrt(2 * p["cost"] / p["n_points"]) > 0.5:
            continue
        k = [i for i in range(truth["n_markers"]) if truth["dict_row"][i] == p["model_index"]][0]
        R, Rt = rodrigues(p["rvec"]), truth["R"][k].reshape(3, 3)
        assert np.degrees(np.arccos(np.clip((np.trace(R.T @ Rt) - 1) / 2, -1, 1))) < 0.5
        good += 1
    assert good >= 2
    det.close()
