"""CPU tests of the product's host side: the C-ABI library loads and exports what include/ctag.h declares, the
loaders behave like the reference's, the synthetic generator is deterministic, and -- without a GPU -- the
detection entry points fail loudly instead of falling back to a CPU implementation."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import cylindertag_amd as ca
import testkit as tk
from cylindertag_amd import capi
from ctag_testlib import GOLDEN, ROOT, read_marker_file


def _declared(*headers):
    hdr = "".join(open(os.path.join(ROOT, "include", f)).read() for f in headers)
    return set(re.findall(r"\b(ctag_[a-z0-9_]+)\s*\(", hdr))


def test_library_exports_every_declared_symbol():
    tk.build()
    declared = _declared("ctag.h", "ctag_pose.h", "ctag_gather.h")
    assert declared == set(capi.EXPORTS)
    L = capi.load_library()
    for s in declared:
        assert hasattr(L, s), s
    nm = subprocess.check_output(["nm", "-D", "--defined-only", ca.lib_path()]).decode()
    for s in declared:
        assert re.search(r"\bT %s\b" % s, nm), s
    assert L.ctag_version() >= 100
    # the product library exports the C ABI of its three headers and nothing else with C linkage: the test / bench scaffolding
    # (include/ctag_testkit.h) lives in libctag_testkit.so
    c_syms = set(re.findall(r"\bT (ctag_[a-z0-9_]+)\b", nm))
    assert c_syms == declared, c_syms ^ declared


def test_testkit_library_exports_every_declared_symbol():
    tk.build()
    declared = _declared("ctag_testkit.h")
    assert declared == set(tk.EXPORTS)
    T = tk.load_library()
    nm = subprocess.check_output(["nm", "-D", "--defined-only", tk.lib_path()]).decode()
    for s in declared:
        assert hasattr(T, s) and re.search(r"\bT %s\b" % s, nm), s
    assert not (declared & _declared("ctag.h", "ctag_pose.h", "ctag_gather.h"))


def test_result_record_layout_matches_header():
    assert ca.RESULT_DT.itemsize == 11616 and ca.FEATURE_DT.itemsize == 100 and ca.MARKER_DT.itemsize == 16
    assert ca.POSE_DT.itemsize == 136 and C.sizeof(capi.CameraC) == 96


def test_params_struct_layout_and_reference_defaults(oracle, dictionary, test_bmp):
    """ctag_params (include/ctag_types.h): the ctypes mirror has the C layout, ctag_params_default gives the reference's constants
    (header/corner_detector.h:90,110,122,135-137,144; corner_detector.cpp:71,88,285) and the oracle consumes the same struct."""
    src = "#include <stdio.h>\n#include <stddef.h>\n#include \"ctag_types.h\"\nint main(){printf(\"%zu %zu %zu %zu %zu\", sizeof(ctag_params), " \
          "offsetof(ctag_params, ID_cr_correspond), offsetof(ctag_params, dark_cap), offsetof(ctag_params, area_max_fraction), offsetof(ctag_params, collinear_cost));}"
    exe = os.path.join(ROOT, "cylindertag_amd", "_build", "params_layout")
    subprocess.run(["gcc", "-x", "c", "-I" + os.path.join(ROOT, "include"), "-o", exe, "-"], input=src.encode(), check=True)
    got = [int(v) for v in subprocess.check_output([exe]).split()]
    P = capi.ParamsC
    assert got == [C.sizeof(P), P.ID_cr_correspond.offset, P.dark_cap.offset, P.area_max_fraction.offset, P.collinear_cost.offset]
    p = ca.default_params()
    assert (round(p.threshold_line, 6), round(p.threshold_expand, 6), round(p.threshold_RAC, 6), p.threshold_angle, p.threshold_vertical) == (1.8, 1.2, 0.3, 5.0, 0.5)
    assert [round(v, 6) for v in p.ID_cr_correspond] == [1.47, 1.54, 1.61, 1.68]
    assert [round(v, 6) for v in p.cr_covariance_left] == [0.1, 0.035, 0.035, 0.035] and [round(v, 6) for v in p.cr_covariance_right] == [0.035, 0.035, 0.035, 0.1]
    assert (round(p.dark_cap, 6), p.area_min, p.area_max_fraction, p.collinear_cost) == (0.3, 30, 0.01, 1.05)
    state, fs = dictionary
    base = oracle.detect_fast(test_bmp, state, fs)
    oracle.set_params(p)  # the defaults, explicitly
    try:
        assert oracle.detect_fast(test_bmp, state, fs).tobytes() == base.tobytes()
        p.dark_cap = 0.2
        p.area_min = 60
        oracle.set_params(p)
        assert oracle.detect_fast(test_bmp, state, fs).tobytes() != base.tobytes()
    finally:
        oracle.set_params(None)
    assert oracle.detect_fast(test_bmp, state, fs).tobytes() == base.tobytes()


def test_marker_loader_matches_reference_format(tmp_path):
    state, fs = ca.load_marker_file(os.path.join(GOLDEN, "CTag_2f12c.marker"))
    ref, rfs = read_marker_file(os.path.join(GOLDEN, "CTag_2f12c.marker"))
    assert fs == rfs == 2 and state.shape == (41, 12) and (state == ref).all()
    bad = tmp_path / "bad.marker"
    bad.write_text("1 3 2\n1 64 3\n")  # code outside 0..63: check_dictionary (CylinderTag.cpp:56-65)
    with pytest.raises(ca.CtagError):
        ca.load_marker_file(str(bad))
    with pytest.raises(ca.CtagError):
        ca.load_marker_file(str(tmp_path / "missing.marker"))


def test_create_rejects_illegal_dictionary_before_touching_the_gpu():
    L = capi.load_library()
    h = C.c_void_p()
    bad = np.array([[1, 2, 99]], np.int32)
    assert L.ctag_create(bad.ctypes.data_as(C.POINTER(C.c_int32)), 1, 3, 2, 0, C.byref(h)) == -1  # CTAG_ERR_ARG
    assert not h.value


def test_no_cpu_fallback_without_gpu(dictionary):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    state, fs = dictionary
    with pytest.raises(ca.CtagError) as e:
        ca.Detector(state, fs)
    assert e.value.status == -2  # CTAG_ERR_HIP


def test_synthetic_generator_is_deterministic_and_planted(dictionary):
    state, fs = dictionary
    a, ta = tk.synth_frame_host(state, 7, rows=540, cols=960)
    b, tb = tk.synth_frame_host(state, 7, rows=540, cols=960)
    c, tc = tk.synth_frame_host(state, 8, rows=540, cols=960)
    assert (a == b).all() and ta.tobytes() == tb.tobytes() and not (a == c).all()
    assert ta["n_markers"] == 4 and (ta["strip_len"][:4] > 100).all()
    assert a.min() < 60 and a.max() > 200  # black quads and white paper are present


def test_strerror_messages_mirror_reference_text():
    L = capi.load_library()
    assert L.ctag_strerror(1) == b"No corner detected!" and L.ctag_strerror(2) == b"No feature detected!"
    assert [L.ctag_stage_name(i).decode() for i in range(len(ca.STAGE_NAMES))] == ca.STAGE_NAMES and L.ctag_stage_name(len(ca.STAGE_NAMES)) == b""


def test_cpp_host_layer_builds_and_keeps_reference_interface():
    so = os.path.join(ROOT, "cylindertag_amd", "_build", "libcylindertag.so")
    assert os.path.exists(so)
    nm = subprocess.check_output(["nm", "-DC", "--defined-only", so]).decode()
    assert "CylinderTag::detect(" in nm and "CylinderTag::CylinderTag(std::" in nm
    hdr = " ".join(open(os.path.join(ROOT, "cylindertag_amd", "csrc", "CylinderTag.h")).read().split())
    assert "int adaptiveThresh = 5, const bool cornerSubPix = false, int cornerSubPixDist = 3" in hdr


def test_cpp_bmp_reader_matches_python_reader(test_bmp, tmp_path):
    """Frame ingest of the host layer: the C++ BMP reader returns the same gray image as the
    test-suite's reader for the reference's 8-bit test.bmp, and applies OpenCV's fixed-point BGR2GRAY to 24-bit files."""
    so = C.CDLL(os.path.join(ROOT, "cylindertag_amd", "_build", "libcylindertag.so"))
    f = so.ctag_host_read_bmp_gray
    f.restype = C.c_int
    f.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_size_t]
    r, c = C.c_int(), C.c_int()
    buf = np.zeros(1200 * 1920, np.uint8)
    assert f(os.path.join(GOLDEN, "test.bmp").encode(), C.byref(r), C.byref(c), buf.ctypes.data, buf.size) == 0
    assert (r.value, c.value) == (1200, 1920) and (buf.reshape(1200, 1920) == test_bmp).all()
    # a 24-bit bottom-up BMP written by hand
    rng = np.random.RandomState(0)
    bgr = rng.randint(0, 256, (5, 7, 3)).astype(np.uint8)
    rowbytes = (7 * 3 + 3) // 4 * 4
    body = b"".join(bytes(bgr[y].tobytes()) + bytes(rowbytes - 21) for y in range(4, -1, -1))
    hdr = b"BM" + (54 + len(body)).to_bytes(4, "little") + bytes(4) + (54).to_bytes(4, "little")
    hdr += (40).to_bytes(4, "little") + (7).to_bytes(4, "little") + (5).to_bytes(4, "little") + (1).to_bytes(2, "little") + (24).to_bytes(2, "little")
    hdr += (0).to_bytes(4, "little") + len(body).to_bytes(4, "little") + bytes(16)
    p = tmp_path / "rgb.bmp"
    p.write_bytes(hdr + body)
    out = np.zeros(35, np.uint8)
    assert f(str(p).encode(), C.byref(r), C.byref(c), out.ctypes.data, out.size) == 0 and (r.value, c.value) == (5, 7)
    b, g, rr = bgr[..., 0].astype(np.int64), bgr[..., 1].astype(np.int64), bgr[..., 2].astype(np.int64)
    assert (out.reshape(5, 7) == ((b * 1868 + g * 9617 + rr * 4899 + 8192) >> 14)).all()
    assert f(str(tmp_path / "missing.bmp").encode(), None, None, None, 0) == -1


def test_opencv_branch_of_the_cpp_layer_passes_the_compiler():
    """COMPILE CHECK ONLY (pins nothing about OpenCV): the -DCTAG_WITH_OPENCV branch of CylinderTag.{h,cpp} -- cv::Mat in
    detect(), cv::Mat camera / pose members, cv::Mat1i dictionary: the drop-in build of INTEGRATION.md option A -- goes through
    g++'s syntax and type checks against a declaration-only stand-in for <opencv2/core.hpp> (tests/opencv_decl_stub), because
    this image has no OpenCV; without this the branch had never seen a compiler."""
    for src in (os.path.join("csrc", "CylinderTag.cpp"), os.path.join("examples", "ctag_demo.cpp")):
        cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-DCTAG_WITH_OPENCV", "-I" + os.path.join(ROOT, "tests", "opencv_decl_stub"),
               "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "cylindertag_amd", src)]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-3000:]
    # the LINK-level check (real OpenCV, demo run against the stand-alone build's output) lives in oracle/ref_build.sh and runs
    # wherever pkg-config finds opencv4; tests/test_oracle_ref_cpu.py calls the recipe
