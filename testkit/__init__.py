"""ctypes binding of include/ctag_testkit.h (testkit/_build/libctag_testkit.so): TEST AND BENCH SCAFFOLDING around the
product library -- parity probes, device evaluation of the shared math, the synthetic frame generators and the one-GPU
execution of the multi-rank unpack.  Nothing in cylindertag_amd/ imports this module."""
import ctypes as C
import os
import subprocess

import numpy as np

import cylindertag_amd as ca
from cylindertag_amd import capi
from cylindertag_amd.capi import CtagError, Model

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.environ.get("CTAG_TESTKIT_LIB") or os.path.join(_HERE, "_build", "libctag_testkit.so")

TRUTH3D_DT = np.dtype([("n_markers", "<i4"), ("dict_row", "<i4", (8,)), ("_pad", "<i4", (1,)), ("R", "<f8", (8, 9)), ("t", "<f8", (8, 3)),
                       ("radius", "<f8", (8,))])
TRUTH_DT = np.dtype([("n_markers", "<i4"), ("dict_row", "<i4", (8,)), ("strip_len", "<f4", (8,)),
                     ("corners", "<f4", (8, 8))])
DBG_HALF, DBG_LABELS, DBG_CANDIDATES, DBG_CAND_QUADS, DBG_FEATURES0, DBG_FEATURES1, DBG_FEATURES2, DBG_PREMARKERS, DBG_GRAY, DBG_LINES, DBG_MASK = range(1, 12)
SYNTH_SEED = 0x4354616753594E00  # "CTagSYN\0", SURVEY.md 8(d)

# every symbol include/ctag_testkit.h declares (tests check the library exports all of them)
EXPORTS = ["ctag_debug_fetch", "ctag_math_probe", "ctag_testkit_unpack_gathered", "ctag_testkit_stall_stream", "ctag_synth_frames_device", "ctag_synth_frame_host",
           "ctag_synth_layout_truth", "ctag_synth3d_frames_device", "ctag_synth3d_frame_host", "ctag_synth3d_model"]


def lib_path():
    return _LIB


def build(verbose=False):
    """Compile the product library and the test kit in-tree."""
    ca.build(verbose)
    subprocess.check_call(["make", "-C", _HERE] + ([] if verbose else ["-s"]))


_lib = None


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    capi.load_library()  # the product library first (and torch's HIP runtime before it)
    if not os.path.exists(_LIB):
        raise FileNotFoundError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`" % _LIB)
    L = C.CDLL(_LIB)
    vp, i32p = C.c_void_p, C.POINTER(C.c_int32)
    L.ctag_debug_fetch.restype = C.c_long
    L.ctag_debug_fetch.argtypes = [vp, C.c_int, C.c_int, vp, C.c_size_t]
    L.ctag_math_probe.restype = C.c_int
    L.ctag_math_probe.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp]
    L.ctag_testkit_unpack_gathered.restype = C.c_int
    L.ctag_testkit_unpack_gathered.argtypes = [vp, vp, C.c_int, C.c_int, C.c_uint64, vp]
    L.ctag_testkit_stall_stream.restype = C.c_int
    L.ctag_testkit_stall_stream.argtypes = [vp, C.c_int]
    L.ctag_synth_frames_device.restype = C.c_int
    L.ctag_synth_frames_device.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_ssize_t, C.c_ssize_t,
                                           C.c_uint64, C.c_int]
    L.ctag_synth_frame_host.restype = C.c_int
    L.ctag_synth_frame_host.argtypes = [i32p, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_ssize_t, C.c_uint64,
                                        C.c_int, vp]
    L.ctag_synth_layout_truth.restype = C.c_int
    L.ctag_synth_layout_truth.argtypes = [i32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, vp]
    L.ctag_synth3d_frames_device.restype = C.c_int
    L.ctag_synth3d_frames_device.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_ssize_t, C.c_ssize_t, C.c_uint64, C.c_int,
                                             C.c_double, C.c_double, C.c_double, C.c_double]
    L.ctag_synth3d_frame_host.restype = C.c_int
    L.ctag_synth3d_frame_host.argtypes = [i32p, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_ssize_t, C.c_uint64, C.c_int, C.c_double,
                                          C.c_double, C.c_double, C.c_double, vp]
    L.ctag_synth3d_model.restype = C.c_int
    L.ctag_synth3d_model.argtypes = [i32p, C.c_int, C.c_int, vp]
    _lib = L
    return L


def synth_frame_host(state, frame_index, rows=1080, cols=1920, seed=SYNTH_SEED, markers=4):
    """Host rendering of synthetic frame `frame_index` (same code path as the device generator)."""
    L = load_library()
    state = np.ascontiguousarray(state, dtype=np.int32)
    img = np.zeros((rows, cols), np.uint8)
    truth = np.zeros(1, TRUTH_DT)
    st = L.ctag_synth_frame_host(state.ctypes.data_as(C.POINTER(C.c_int32)), state.shape[0], state.shape[1],
                                 img.ctypes.data, frame_index, rows, cols, img.strides[0], seed, markers,
                                 truth.ctypes.data)
    if st != 0:
        raise CtagError(st)
    return img, truth[0]


def synth3d_frame_host(state, frame_index, K, rows=2160, cols=3840, seed=SYNTH_SEED, markers=4):
    """Host rendering of frame `frame_index` of the 3-D scene (cylinders with planted poses, camera matrix K) -> (image, truth)."""
    L = load_library()
    state = np.ascontiguousarray(state, dtype=np.int32)
    img = np.zeros((rows, cols), np.uint8)
    truth = np.zeros(1, TRUTH3D_DT)
    K = np.asarray(K, np.float64)
    st = L.ctag_synth3d_frame_host(state.ctypes.data_as(C.POINTER(C.c_int32)), state.shape[0], state.shape[1], img.ctypes.data, frame_index,
                                   rows, cols, img.strides[0], seed, markers, K[0, 0], K[1, 1], K[0, 2], K[1, 2], truth.ctypes.data)
    if st != 0:
        raise CtagError(st, "ctag_synth3d_frame_host")
    return img, truth[0]


def synth3d_model(state):
    """3-D corner lists of the synthetic cylinders, one model per dictionary row (marker id = row) -> Model."""
    L = load_library()
    state = np.ascontiguousarray(state, dtype=np.int32)
    corners = np.zeros((state.shape[0], state.shape[1] * 8, 3), np.float32)
    st = L.ctag_synth3d_model(state.ctypes.data_as(C.POINTER(C.c_int32)), state.shape[0], state.shape[1], corners.ctypes.data)
    if st != 0:
        raise CtagError(st, "ctag_synth3d_model")
    return Model(ids=np.arange(state.shape[0], dtype=np.int32), corners=corners, model_size=state.shape[1]), corners


def synth_truth(state, frame_index, rows=1080, cols=1920, seed=SYNTH_SEED, markers=4):
    """Planted markers (dictionary rows, strip corners) of synthetic frame `frame_index`, without rendering."""
    L = load_library()
    state = np.ascontiguousarray(state, dtype=np.int32)
    truth = np.zeros(1, TRUTH_DT)
    st = L.ctag_synth_layout_truth(state.ctypes.data_as(C.POINTER(C.c_int32)), state.shape[0], state.shape[1], frame_index, rows,
                                   cols, seed, markers, truth.ctypes.data)
    if st != 0:
        raise CtagError(st)
    return truth[0]


class Detector(ca.Detector):
    """The product's Detector plus the test kit's entry points on the same handle."""

    def __init__(self, state, feature_size, device=0, **kw):
        super().__init__(state, feature_size, device=device, **kw)
        self.T = load_library()

    def synth_frames_device(self, frames_ptr, first, n, rows, cols, row_stride, frame_stride, seed=SYNTH_SEED, markers=4):
        st = self.T.ctag_synth_frames_device(self.h, frames_ptr, first, n, rows, cols, row_stride, frame_stride, seed, markers)
        if st != 0:
            raise CtagError(st, "ctag_synth_frames_device")

    def synth3d_frames_device(self, frames_ptr, first, n, rows, cols, row_stride, frame_stride, K, seed=SYNTH_SEED, markers=4):
        K = np.asarray(K, np.float64)
        st = self.T.ctag_synth3d_frames_device(self.h, frames_ptr, first, n, rows, cols, row_stride, frame_stride, seed, markers,
                                               K[0, 0], K[1, 1], K[0, 2], K[1, 2])
        if st != 0:
            raise CtagError(st, "ctag_synth3d_frames_device")

    def stall_stream(self, milliseconds):
        """A kernel that spins for `milliseconds` on the handle's stream: a late peer, as the gather's bounded waits see one."""
        st = self.T.ctag_testkit_stall_stream(self.h, int(milliseconds))
        if st != 0:
            raise CtagError(st, "ctag_testkit_stall_stream")

    def unpack_gathered(self, gathered_ptr, n_total, world, width, out_ptr):
        """ctag_gather_end's segment table + unpack kernels for a `world`-rank job on a caller-built gathered buffer."""
        self._gcheck(self.T.ctag_testkit_unpack_gathered(self.h, gathered_ptr, n_total, world, width, out_ptr), "ctag_testkit_unpack_gathered")

    def debug(self, frame, what):
        n = self.T.ctag_debug_fetch(self.h, frame, what, None, 0)
        if n < 0:
            raise CtagError(-1, "ctag_debug_fetch(%d)" % what)
        if what in (DBG_HALF, DBG_GRAY, DBG_MASK):
            a = np.zeros(n, np.uint8)
        elif what in (DBG_LABELS, DBG_CANDIDATES, DBG_LINES):
            a = np.zeros(n, np.int32)
        elif what == DBG_PREMARKERS:
            a = np.zeros(1, ca.RESULT_DT)
        else:
            a = np.zeros(n, np.float32)
        if n:
            got = self.T.ctag_debug_fetch(self.h, frame, what, a.ctypes.data, max(n, 1))
            if got < 0:
                raise CtagError(-2, "ctag_debug_fetch(%d)" % what)
        if what in (DBG_CANDIDATES, DBG_CAND_QUADS):
            return a.reshape(-1, 8)
        if what in (DBG_FEATURES0, DBG_FEATURES1, DBG_FEATURES2):
            return a.reshape(-1, 19)
        if what == DBG_PREMARKERS:
            return a[0]
        return a

    def math(self, op, a, b=None):
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b if b is not None else np.zeros_like(a), dtype=np.float64)
        out = np.zeros_like(a)
        st = self.T.ctag_math_probe(self.h, op, a.size, a.ctypes.data, b.ctypes.data, out.ctypes.data)
        if st != 0:
            raise CtagError(st, "ctag_math_probe")
        return out
