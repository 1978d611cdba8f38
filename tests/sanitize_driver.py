"""Child process of tests/test_oracle_sanitize_cpu.py: runs the AddressSanitizer + UBSan builds of both CPU oracles
(make -C oracle asan) over the inputs in the .npz given on the command line.  Must be started with libasan preloaded
(the Python interpreter itself is not instrumented).  Any sanitizer report aborts the process (non-zero exit)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ctag_testlib import GOLDEN, ROOT, Oracle, read_marker_file  # noqa: E402
from pose_testlib import PoseOracle, make_camera, make_model_view, read_camera_yml, read_model_file, synth_pose_results  # noqa: E402


def main():
    inputs = np.load(sys.argv[1])
    state, fs = read_marker_file(os.path.join(GOLDEN, "CTag_2f12c.marker"))
    orc = Oracle(path=os.path.join(ROOT, "oracle", "_build", "libctag_oracle_asan.so"))
    plain = Oracle()
    runs = 0
    bmp_res = None
    for name in inputs.files:
        img = inputs[name]
        params = [(5, True, 5)]
        if name == "test_bmp":
            params += [(3, True, 5), (4, True, 3), (7, False, 3)]
        for tw, subpix, dist in params:
            traced = orc.detect(img, state, fs, tw, subpix, dist)      # traced path (keeps every stage)
            fast = orc.detect_fast(img, state, fs, tw, subpix, dist)   # untraced path
            assert traced["result"].tobytes() == fast.tobytes(), name
            assert fast.tobytes() == plain.detect_fast(img, state, fs, tw, subpix, dist).tobytes(), name  # -O1 sanitized == -O2
            runs += 1
        if name == "test_bmp":
            bmp_res = orc.detect_fast(img, state, fs)
    # the thread pool of the all-cores baseline
    seq = np.stack([inputs[k] for k in inputs.files if k.startswith("seq")])
    many, used = orc.detect_many(seq, state, fs, threads=3)
    assert used == min(3, len(seq))
    for k in range(len(seq)):
        assert many[k].tobytes() == plain.detect_fast(seq[k], state, fs).tobytes()
    # primitives on degenerate inputs
    orc.fitline(np.array([[3, 3], [3, 3]], np.int32), welsch=True)
    orc.fitline(np.array([[0, 0], [5, 0], [9, 0]], np.int32), welsch=False)
    orc.ccl(np.zeros((1, 1), np.uint8))
    orc.ccl(np.full((7, 9), 255, np.uint8))
    # pose oracle: the reference's own scene + synthetic records hitting every correspondence branch
    po = PoseOracle(path=os.path.join(ROOT, "oracle", "_build", "libctag_pose_oracle_asan.so"))
    ref = PoseOracle()
    K, dist = read_camera_yml(os.path.join(GOLDEN, "cameraParams.yml"))
    model = read_model_file(os.path.join(GOLDEN, "CTag_2f12c.model"))
    mv, cam = make_model_view(model), make_camera(K, dist)
    poses = po.pose_frame(bmp_res, mv, cam)
    assert poses.tobytes() == ref.pose_frame(bmp_res, mv, cam).tobytes() and len(poses) == 5
    recs, _ = synth_pose_results(model, K, dist, 24, seed=5)
    for i, r in enumerate(recs):
        assert po.pose_frame(r, mv, cam, i).tobytes() == ref.pose_frame(r, mv, cam, i).tobytes()
    print("sanitize_driver: %d detect runs, %d pooled frames, %d pose frames: clean" % (runs, len(seq), 1 + len(recs)))


if __name__ == "__main__":
    main()
