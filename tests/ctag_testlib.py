"""Shared helpers for the test-suite: ctypes bindings of the CPU oracle (test infrastructure) and small
data readers.  Nothing here is imported by the product package."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

MAX_FEATURES, MAX_MARKERS = 100, 100

FEATURE_DT = np.dtype([("pos", "<i4"), ("id", "<i4"), ("id_left", "<i4"), ("id_right", "<i4"),
                       ("corners", "<f4", (16,)), ("center", "<f4", (2,)), ("edge_length", "<f4"),
                       ("cr_left", "<f4"), ("cr_right", "<f4")])
MARKER_DT = np.dtype([("marker_id", "<i4"), ("first_feature", "<i4"), ("n_features", "<i4"), ("n_pos", "<i4")])
RESULT_DT = np.dtype([("status", "<i4"), ("n_markers", "<i4"), ("n_features", "<i4"), ("flags", "<u4"),
                      ("markers", MARKER_DT, (MAX_MARKERS,)), ("features", FEATURE_DT, (MAX_FEATURES,))])
assert RESULT_DT.itemsize == 11616


def read_bmp_gray(path):
    """8-bit palettised (gray ramp) or 24-bit BMP -> HxW uint8 (top-down)."""
    d = open(path, "rb").read()
    assert d[:2] == b"BM"
    off = struct.unpack_from("<I", d, 10)[0]
    w, h, planes, bpp, comp = struct.unpack_from("<iiHHI", d, 18)
    assert comp == 0
    flip = h > 0
    h = abs(h)
    rowbytes = ((w * bpp + 31) // 32) * 4
    raw = np.frombuffer(d, dtype=np.uint8, count=rowbytes * h, offset=off).reshape(h, rowbytes)
    if bpp == 8:
        pal = np.frombuffer(d, dtype=np.uint8, count=1024, offset=54).reshape(256, 4)
        img = raw[:, :w]
        if not (pal[:, 0] == np.arange(256)).all():
            b, g, r = pal[:, 0].astype(np.float64), pal[:, 1].astype(np.float64), pal[:, 2].astype(np.float64)
            lut = np.clip(np.rint(0.299 * r + 0.587 * g + 0.114 * b), 0, 255).astype(np.uint8)
            img = lut[img]
    elif bpp == 24:
        px = raw[:, :w * 3].reshape(h, w, 3).astype(np.float64)
        img = np.clip(np.rint(0.114 * px[..., 0] + 0.587 * px[..., 1] + 0.299 * px[..., 2]), 0, 255).astype(np.uint8)
    else:
        raise ValueError("unsupported bpp %d" % bpp)
    if flip:
        img = img[::-1]
    return np.ascontiguousarray(img)


def read_marker_file(path):
    toks = open(path).read().split()
    n, c, fs = int(toks[0]), int(toks[1]), int(toks[2])
    state = np.array([int(t) for t in toks[3:3 + n * c]], dtype=np.int32).reshape(n, c)
    return state, fs


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


_p_u8 = C.POINTER(C.c_uint8)
_p_i32 = C.POINTER(C.c_int32)
_p_f32 = C.POINTER(C.c_float)
_p_f64 = C.POINTER(C.c_double)


def _ptr(a, t):
    return a.ctypes.data_as(t)


class Oracle:
    def __init__(self, libm=False, path=None):
        name = "libctag_oracle_libm.so" if libm else "libctag_oracle.so"
        path = path or os.path.join(ROOT, "oracle", "_build", name)  # `path`: another build of the same source (asan, native)
        if not os.path.exists(path):
            build_oracle()
        L = self.L = C.CDLL(path)
        L.ctago_detect.restype = C.c_void_p
        L.ctago_detect.argtypes = [_p_u8, C.c_int, C.c_int, C.c_ssize_t, _p_i32, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_int]
        L.ctago_free.argtypes = [C.c_void_p]
        for f in ("ctago_status", "ctago_half_rows", "ctago_half_cols", "ctago_num_labels", "ctago_num_candidates",
                  "ctago_num_quads", "ctago_num_features"):
            getattr(L, f).argtypes = [C.c_void_p]
            getattr(L, f).restype = C.c_int
        for f in ("ctago_get_half", "ctago_get_binary", "ctago_get_labels", "ctago_get_label_areas",
                  "ctago_get_candidates", "ctago_get_candidate_quads", "ctago_get_quads", "ctago_get_result",
                  "ctago_get_premarkers"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_void_p]
            getattr(L, f).restype = None
        L.ctago_get_features.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.ctago_result_bytes.restype = C.c_size_t
        assert L.ctago_result_bytes() == RESULT_DT.itemsize
        L.ctago_detect_fast.restype = C.c_int
        L.ctago_detect_fast.argtypes = [_p_u8, C.c_int, C.c_int, C.c_ssize_t, _p_i32, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.ctago_detect_many.restype = C.c_int
        L.ctago_detect_many.argtypes = [_p_u8, C.c_int, C.c_int, C.c_int, C.c_ssize_t, C.c_ssize_t, _p_i32, C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.ctago_hardware_concurrency.restype = C.c_int
        L.ctago_resize_half.argtypes = [_p_u8, C.c_int, C.c_int, C.c_ssize_t, _p_u8]
        L.ctago_bgr2gray.argtypes = [_p_u8, C.c_int, C.c_int, C.c_ssize_t, _p_u8]
        L.ctago_threshold.argtypes = [_p_u8, C.c_int, C.c_int, C.c_int, _p_u8]
        L.ctago_ccl.argtypes = [_p_u8, C.c_int, C.c_int, _p_i32, _p_i32, C.c_int]
        L.ctago_ccl.restype = C.c_int
        L.ctago_fitline_l2.argtypes = [_p_i32, C.c_int, _p_f32]
        L.ctago_fitline_welsch.argtypes = [_p_i32, C.c_int, _p_f32]
        L.ctago_math_probe.argtypes = [C.c_int, C.c_int, _p_f64, _p_f64, _p_f64]
        L.ctago_set_variants.argtypes = [C.c_int, C.c_int]
        L.ctago_set_variants.restype = None
        L.ctago_fitline_welsch_variant.argtypes = [_p_i32, C.c_int, C.c_int, _p_f32]
        L.ctago_set_params.argtypes = [C.c_void_p]
        L.ctago_set_params.restype = None

    def set_params(self, params=None):
        """The tunables of every following run: a cylindertag_amd.ParamsC (the struct ctag_create_ex takes), None = the reference's values."""
        self.L.ctago_set_params(C.byref(params) if params is not None else None)

    def set_variants(self, welsch_minerr_in_loop=0, resize_simd_lanes=8):
        """The two unverifiable assumptions about OpenCV 4.5.3 as switches (oracle/ctag_oracle.cpp: OracleVariants); process-wide, defaults (0, 8)."""
        self.L.ctago_set_variants(int(welsch_minerr_in_loop), int(resize_simd_lanes))

    def fitline_welsch_variant(self, pts, variant):
        pts = np.ascontiguousarray(pts, dtype=np.int32)
        out = np.zeros(4, np.float32)
        self.L.ctago_fitline_welsch_variant(_ptr(pts, _p_i32), pts.shape[0], int(variant), _ptr(out, _p_f32))
        return out

    # ---- full traced run -------------------------------------------------------------------------
    def detect(self, gray, state, feature_size, adaptive_thresh=5, subpix=True, subpix_dist=5):
        gray = np.ascontiguousarray(gray, dtype=np.uint8)
        state = np.ascontiguousarray(state, dtype=np.int32)
        L = self.L
        h = L.ctago_detect(_ptr(gray, _p_u8), gray.shape[0], gray.shape[1], gray.strides[0], _ptr(state, _p_i32),
                           state.shape[0], state.shape[1], feature_size, adaptive_thresh, int(subpix), subpix_dist)
        try:
            out = {"status": L.ctago_status(h)}
            hr, hc = L.ctago_half_rows(h), L.ctago_half_cols(h)
            out["half"] = np.zeros((hr, hc), np.uint8)
            out["binary"] = np.zeros((hr, hc), np.uint8)
            out["labels"] = np.zeros((hr, hc), np.int32)
            if hr * hc:
                L.ctago_get_half(h, out["half"].ctypes.data)
                L.ctago_get_binary(h, out["binary"].ctypes.data)
                L.ctago_get_labels(h, out["labels"].ctypes.data)
            nl = L.ctago_num_labels(h)
            out["areas"] = np.zeros(nl, np.int32)
            if nl:
                L.ctago_get_label_areas(h, out["areas"].ctypes.data)
            nc = L.ctago_num_candidates(h)
            out["candidates"] = np.zeros((nc, 8), np.int32)
            out["candidate_quads"] = np.zeros((nc, 8), np.float32)
            if nc:
                L.ctago_get_candidates(h, out["candidates"].ctypes.data)
                L.ctago_get_candidate_quads(h, out["candidate_quads"].ctypes.data)
            nq = L.ctago_num_quads(h)
            out["quads"] = np.zeros((nq, 8), np.float32)
            if nq:
                L.ctago_get_quads(h, out["quads"].ctypes.data)
            nf = L.ctago_num_features(h)
            out["features"] = []
            for st in range(3):
                a = np.zeros((nf, 19), np.float32)
                if nf and (st == 0 or out["status"] == 0):
                    L.ctago_get_features(h, st, a.ctypes.data)
                out["features"].append(a)
            res = np.zeros(1, RESULT_DT)
            L.ctago_get_result(h, res.ctypes.data)
            out["result"] = res[0]
            pre = np.zeros(1, RESULT_DT)
            L.ctago_get_premarkers(h, pre.ctypes.data)
            out["premarkers"] = pre[0]
            return out
        finally:
            L.ctago_free(h)

    def detect_fast(self, gray, state, feature_size, adaptive_thresh=5, subpix=True, subpix_dist=5):
        gray = np.ascontiguousarray(gray, dtype=np.uint8)
        state = np.ascontiguousarray(state, dtype=np.int32)
        res = np.zeros(1, RESULT_DT)
        self.L.ctago_detect_fast(_ptr(gray, _p_u8), gray.shape[0], gray.shape[1], gray.strides[0],
                                 _ptr(state, _p_i32), state.shape[0], state.shape[1], feature_size, adaptive_thresh,
                                 int(subpix), subpix_dist, res.ctypes.data)
        return res[0]

    def detect_many(self, frames, state, feature_size, adaptive_thresh=5, subpix=True, subpix_dist=5, threads=0):
        """n frames (n, rows, cols) through ctago_detect_many: a std::thread pool inside the oracle library (threads <= 0:
        hardware_concurrency).  Returns (records, threads_used)."""
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        state = np.ascontiguousarray(state, dtype=np.int32)
        n, rows, cols = frames.shape
        res = np.zeros(n, RESULT_DT)
        used = self.L.ctago_detect_many(_ptr(frames, _p_u8), n, rows, cols, frames.strides[1], frames.strides[0],
                                        _ptr(state, _p_i32), state.shape[0], state.shape[1], feature_size, adaptive_thresh,
                                        int(subpix), subpix_dist, threads, res.ctypes.data)
        if used < 0:
            raise RuntimeError("ctago_detect_many failed (%d)" % used)
        return res, used

    # ---- primitives ----------------------------------------------------------------------------------
    def resize_half(self, gray):
        gray = np.ascontiguousarray(gray, dtype=np.uint8)
        out = np.zeros((gray.shape[0] // 2, gray.shape[1] // 2), np.uint8)
        self.L.ctago_resize_half(_ptr(gray, _p_u8), gray.shape[0], gray.shape[1], gray.strides[0], _ptr(out, _p_u8))
        return out

    def bgr2gray(self, bgr):
        bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
        out = np.zeros(bgr.shape[:2], np.uint8)
        self.L.ctago_bgr2gray(_ptr(bgr, _p_u8), bgr.shape[0], bgr.shape[1], bgr.strides[0], _ptr(out, _p_u8))
        return out

    def threshold(self, half, tw=5):
        half = np.ascontiguousarray(half, dtype=np.uint8)
        out = np.zeros_like(half)
        self.L.ctago_threshold(_ptr(half, _p_u8), half.shape[0], half.shape[1], tw, _ptr(out, _p_u8))
        return out

    def ccl(self, binary):
        binary = np.ascontiguousarray(binary, dtype=np.uint8)
        labels = np.zeros(binary.shape, np.int32)
        areas = np.zeros(binary.size + 1, np.int32)
        n = self.L.ctago_ccl(_ptr(binary, _p_u8), binary.shape[0], binary.shape[1], _ptr(labels, _p_i32),
                             _ptr(areas, _p_i32), areas.size)
        return labels, areas[:n]

    def fitline(self, pts, welsch):
        pts = np.ascontiguousarray(pts, dtype=np.int32)
        out = np.zeros(4, np.float32)
        f = self.L.ctago_fitline_welsch if welsch else self.L.ctago_fitline_l2
        f(_ptr(pts, _p_i32), pts.shape[0], _ptr(out, _p_f32))
        return out

    def math(self, op, a, b=None):
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b if b is not None else np.zeros_like(a), dtype=np.float64)
        out = np.zeros_like(a)
        self.L.ctago_math_probe(op, a.size, _ptr(a, _p_f64), _ptr(b, _p_f64), _ptr(out, _p_f64))
        return out


def result_markers(res):
    """ctag_frame_result record -> list of dicts (one per marker) for readable assertions."""
    out = []
    for m in res["markers"][:res["n_markers"]]:
        f = res["features"][m["first_feature"]:m["first_feature"] + m["n_features"]]
        out.append({"marker_id": int(m["marker_id"]), "pos": [int(x) for x in f["pos"][:m["n_pos"]]],
                    "id": [int(x) for x in f["id"]], "id_left": [int(x) for x in f["id_left"]],
                    "id_right": [int(x) for x in f["id_right"]], "corners": f["corners"].copy(),
                    "center": f["center"].copy(), "edge_length": f["edge_length"].copy(),
                    "cr_left": f["cr_left"].copy(), "cr_right": f["cr_right"].copy()})
    return out
