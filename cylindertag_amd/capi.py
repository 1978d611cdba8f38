"""ctypes binding of include/ctag.h (libctag_hip.so).  Plumbing only: no arithmetic happens here."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.environ.get("CTAG_HIP_LIB") or os.path.join(_HERE, "_build", "libctag_hip.so")  # override: A/B builds (tools/)

MAX_FEATURES, MAX_MARKERS = 100, 100
FEATURE_DT = np.dtype([("pos", "<i4"), ("id", "<i4"), ("id_left", "<i4"), ("id_right", "<i4"),
                       ("corners", "<f4", (16,)), ("center", "<f4", (2,)), ("edge_length", "<f4"),
                       ("cr_left", "<f4"), ("cr_right", "<f4")])
MARKER_DT = np.dtype([("marker_id", "<i4"), ("first_feature", "<i4"), ("n_features", "<i4"), ("n_pos", "<i4")])
RESULT_DT = np.dtype([("status", "<i4"), ("n_markers", "<i4"), ("n_features", "<i4"), ("flags", "<u4"),
                      ("markers", MARKER_DT, (MAX_MARKERS,)), ("features", FEATURE_DT, (MAX_FEATURES,))])
STAGE_NAMES = ["decimate", "threshold_ccl", "seam_merge", "resolve", "candidates", "quad_pack", "quad_edges", "quad_edges_big",
               "line_sort", "welsch", "quad_final", "features", "edge_refine", "markers"]
QUAD_STAGES = ["quad_pack", "quad_edges", "quad_edges_big", "line_sort", "welsch", "quad_final"]  # a4: edgeExtraction

OPT_MAX_CHUNK, OPT_TIMING, OPT_KEEP_PREMARKERS, OPT_HOST_SUBCHUNK, OPT_GRAPH, OPT_WAVE_POINTS, OPT_FUSED_SWEEP, OPT_STREAMS, OPT_EXPAND_EXACT, OPT_BGR_DIRECT = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10

# every symbol include/ctag.h declares (tests check the library exports all of them)
EXPORTS = ["ctag_create", "ctag_create_ex", "ctag_params_default", "ctag_destroy", "ctag_load_marker_file", "ctag_free", "ctag_detect_u8", "ctag_detect_batch_u8",
           "ctag_detect_batch_device", "ctag_detect_bgr8", "ctag_detect_batch_bgr8", "ctag_detect_batch_bgr8_device", "ctag_host_alloc", "ctag_host_free", "ctag_sync", "ctag_stream", "ctag_set_option", "ctag_get_timings", "ctag_get_counters", "ctag_submit_u8", "ctag_collect",
           "ctag_stage_name", "ctag_strerror", "ctag_version"]
# ... and include/ctag_pose.h
POSE_EXPORTS = ["ctag_model_load", "ctag_model_create", "ctag_model_free", "ctag_model_get_view", "ctag_camera_load",
                "ctag_pose_batch_device", "ctag_estimate_pose", "ctag_pose_last_ms"]
# ... and include/ctag_gather.h
GATHER_EXPORTS = ["ctag_shard_range", "ctag_packed_capacity", "ctag_pack_results", "ctag_unpack_results", "ctag_comm_unique_id",
                  "ctag_comm_init", "ctag_comm_attach", "ctag_comm_destroy", "ctag_comm_native", "ctag_comm_last_error", "ctag_gather_begin",
                  "ctag_gather_end", "ctag_gather_wait", "ctag_gather", "ctag_gather_set_timeout", "ctag_gather_last_bytes"]
EXPORTS = EXPORTS + POSE_EXPORTS + GATHER_EXPORTS
COMM_ID_BYTES = 128

POSE_DT = np.dtype([("status", "<i4"), ("model_index", "<i4"), ("frame", "<i4"), ("marker", "<i4"),
                    ("n_points", "<i4"), ("iterations", "<i4"), ("rvec", "<f8", (3,)), ("tvec", "<f8", (3,)),
                    ("rvec0", "<f8", (3,)), ("tvec0", "<f8", (3,)), ("cost0", "<f8"), ("cost", "<f8")])
POSE_OK, POSE_NO_MODEL, POSE_TOO_FEW, POSE_BAD_POS, POSE_DEGENERATE = range(5)


class ParamsC(C.Structure):  # ctag_params (include/ctag_types.h): the reference's tunables
    _fields_ = [("struct_size", C.c_uint32), ("threshold_line", C.c_float), ("threshold_expand", C.c_float), ("threshold_RAC", C.c_float), ("threshold_angle", C.c_float),
                ("threshold_vertical", C.c_float), ("ID_cr_correspond", C.c_float * 4), ("cr_covariance_left", C.c_float * 4),
                ("cr_covariance_right", C.c_float * 4), ("dark_cap", C.c_float), ("area_min", C.c_int32), ("area_max_fraction", C.c_double),
                ("collinear_cost", C.c_double)]


def default_params():
    """ctag_params_default: the reference's values (header/corner_detector.h:90,110,122,135-137,144; corner_detector.cpp:71,88,285)."""
    p = ParamsC()
    load_library().ctag_params_default(C.byref(p))
    return p


COUNTER_NAMES = ["components", "candidates", "quads", "features", "markers"]
PENDING = -5  # CTAG_PENDING
ERR_ARG, ERR_HIP, ERR_LIMIT, ERR_UNSUPPORTED = -1, -2, -3, -4  # include/ctag_types.h


class CountersC(C.Structure):  # ctag_counters (include/ctag_types.h)
    _fields_ = [("frames", C.c_int64), ("sum", C.c_int64 * 5), ("max", C.c_int32 * 5), ("reruns", C.c_int32)]


class CameraC(C.Structure):  # ctag_camera
    _fields_ = [("K", C.c_float * 9), ("dist", C.c_float * 14), ("n_dist", C.c_int32)]


class ModelViewC(C.Structure):  # ctag_model_view
    _fields_ = [("n_models", C.c_int32), ("model_size", C.c_int32), ("marker_id", C.POINTER(C.c_int32)),
                ("base", C.POINTER(C.c_float)), ("axis", C.POINTER(C.c_float)), ("corners", C.POINTER(C.c_float))]


class CtagError(RuntimeError):
    def __init__(self, status, what=""):
        self.status = status
        super().__init__("%s (status %d)%s" % (_strerror(status), status, (": " + what) if what else ""))


def lib_path():
    return _LIB


def build(verbose=False):
    """Compile the HIP library in-tree (hipcc cross-compiles for gfx950 without a GPU)."""
    subprocess.check_call(["make", "-C", _HERE, "-j4"] + ([] if verbose else ["-s"]))


_lib = None


def load_library():
    """Load libctag_hip.so; raises if it has not been built.  There is no fallback implementation."""
    global _lib
    if _lib is not None:
        return _lib
    try:  # torch wheels bundle their own libamdhip64: let it load first so the process holds ONE HIP runtime
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(_LIB):
        raise FileNotFoundError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                "(the detection path has no CPU fallback)" % _LIB)
    L = C.CDLL(_LIB)
    vp, i32p, u8p = C.c_void_p, C.POINTER(C.c_int32), C.c_void_p
    L.ctag_create.restype = C.c_int
    L.ctag_create.argtypes = [i32p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.ctag_create_ex.restype = C.c_int
    L.ctag_create_ex.argtypes = [i32p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(ParamsC), C.POINTER(vp)]
    L.ctag_params_default.restype = None
    L.ctag_params_default.argtypes = [C.POINTER(ParamsC)]
    L.ctag_destroy.argtypes = [vp]
    L.ctag_destroy.restype = None
    L.ctag_load_marker_file.restype = C.c_int
    L.ctag_load_marker_file.argtypes = [C.c_char_p, C.POINTER(i32p), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                        C.POINTER(C.c_int)]
    L.ctag_free.argtypes = [vp]
    L.ctag_free.restype = None
    L.ctag_detect_u8.restype = C.c_int
    L.ctag_detect_u8.argtypes = [vp, u8p, C.c_int, C.c_int, C.c_ssize_t, C.c_int, C.c_int, C.c_int, vp]
    L.ctag_detect_batch_u8.restype = C.c_int
    L.ctag_detect_batch_u8.argtypes = [vp, u8p, C.c_int, C.c_int, C.c_int, C.c_ssize_t, C.c_ssize_t, C.c_int, C.c_int,
                                       C.c_int, vp]
    L.ctag_detect_batch_device.restype = C.c_int
    L.ctag_detect_batch_device.argtypes = L.ctag_detect_batch_u8.argtypes
    L.ctag_detect_bgr8.restype = C.c_int
    L.ctag_detect_bgr8.argtypes = L.ctag_detect_u8.argtypes
    L.ctag_detect_batch_bgr8.restype = C.c_int
    L.ctag_detect_batch_bgr8.argtypes = L.ctag_detect_batch_u8.argtypes
    L.ctag_detect_batch_bgr8_device.restype = C.c_int
    L.ctag_detect_batch_bgr8_device.argtypes = L.ctag_detect_batch_u8.argtypes
    L.ctag_host_alloc.restype = vp
    L.ctag_host_alloc.argtypes = [C.c_size_t]
    L.ctag_host_free.restype = None
    L.ctag_host_free.argtypes = [vp]
    L.ctag_sync.restype = C.c_int
    L.ctag_sync.argtypes = [vp]
    L.ctag_stream.restype = vp
    L.ctag_stream.argtypes = [vp]
    L.ctag_set_option.restype = C.c_int
    L.ctag_set_option.argtypes = [vp, C.c_int, C.c_int64]
    L.ctag_submit_u8.restype = C.c_int
    L.ctag_submit_u8.argtypes = [vp, vp, C.c_int, C.c_int, C.c_ssize_t, C.c_int, C.c_int, C.c_int]
    L.ctag_collect.restype = C.c_int
    L.ctag_collect.argtypes = [vp, vp]
    L.ctag_get_counters.restype = C.c_int
    L.ctag_get_counters.argtypes = [vp, C.POINTER(CountersC)]
    L.ctag_get_timings.restype = C.c_int
    L.ctag_get_timings.argtypes = [vp, C.POINTER(C.c_float), C.c_int]
    L.ctag_stage_name.restype = C.c_char_p
    L.ctag_stage_name.argtypes = [C.c_int]
    L.ctag_strerror.restype = C.c_char_p
    L.ctag_strerror.argtypes = [C.c_int]
    L.ctag_version.restype = C.c_int
    L.ctag_model_load.restype = C.c_int
    L.ctag_model_load.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.ctag_model_create.restype = C.c_int
    L.ctag_model_create.argtypes = [C.POINTER(ModelViewC), C.POINTER(vp)]
    L.ctag_model_free.restype = None
    L.ctag_model_free.argtypes = [vp]
    L.ctag_model_get_view.restype = C.c_int
    L.ctag_model_get_view.argtypes = [vp, C.POINTER(ModelViewC)]
    L.ctag_camera_load.restype = C.c_int
    L.ctag_camera_load.argtypes = [C.c_char_p, C.POINTER(CameraC)]
    L.ctag_pose_batch_device.restype = C.c_int
    L.ctag_pose_batch_device.argtypes = [vp, vp, C.c_int, vp, C.POINTER(CameraC), vp, vp, C.c_int]
    L.ctag_estimate_pose.restype = C.c_int
    L.ctag_estimate_pose.argtypes = [vp, vp, vp, C.POINTER(CameraC), vp]
    L.ctag_pose_last_ms.restype = C.c_float
    L.ctag_pose_last_ms.argtypes = [vp]
    u64p = C.POINTER(C.c_uint64)
    L.ctag_shard_range.restype = C.c_int
    L.ctag_shard_range.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.ctag_packed_capacity.restype = C.c_size_t
    L.ctag_packed_capacity.argtypes = [C.c_int]
    L.ctag_pack_results.restype = C.c_int
    L.ctag_pack_results.argtypes = [vp, vp, C.c_int, vp, C.c_size_t, u64p]
    L.ctag_unpack_results.restype = C.c_int
    L.ctag_unpack_results.argtypes = [vp, vp, C.c_int, vp]
    L.ctag_comm_unique_id.restype = C.c_int
    L.ctag_comm_unique_id.argtypes = [vp]
    L.ctag_comm_init.restype = C.c_int
    L.ctag_comm_init.argtypes = [vp, vp, C.c_int, C.c_int]
    L.ctag_comm_attach.restype = C.c_int
    L.ctag_comm_attach.argtypes = [vp, vp, C.c_int, C.c_int]
    L.ctag_comm_destroy.restype = C.c_int
    L.ctag_comm_destroy.argtypes = [vp]
    L.ctag_comm_native.restype = vp
    L.ctag_comm_native.argtypes = [vp]
    L.ctag_comm_last_error.restype = C.c_char_p
    L.ctag_comm_last_error.argtypes = [vp]
    L.ctag_gather_begin.restype = C.c_int
    L.ctag_gather_begin.argtypes = [vp, vp, C.c_int, C.c_int]
    L.ctag_gather_end.restype = C.c_int
    L.ctag_gather_end.argtypes = [vp, vp]
    L.ctag_gather_wait.restype = C.c_int
    L.ctag_gather_wait.argtypes = [vp]
    L.ctag_gather.restype = C.c_int
    L.ctag_gather.argtypes = [vp, vp, C.c_int, C.c_int, vp]
    L.ctag_gather_last_bytes.restype = C.c_int
    L.ctag_gather_last_bytes.argtypes = [vp, u64p, u64p]
    L.ctag_gather_set_timeout.restype = C.c_int
    L.ctag_gather_set_timeout.argtypes = [vp, C.c_int]
    _lib = L
    return L


def _strerror(status):
    try:
        return load_library().ctag_strerror(status).decode()
    except Exception:  # pragma: no cover
        return "ctag error"


def comm_unique_id():
    """ncclGetUniqueId through the C ABI: 128 bytes rank 0 hands to the other ranks."""
    buf = (C.c_ubyte * COMM_ID_BYTES)()
    st = load_library().ctag_comm_unique_id(buf)
    if st != 0:
        raise CtagError(st, "ctag_comm_unique_id (RCCL not loadable?)")
    return bytes(buf)


def shard_range(n_total, rank, world):
    lo, hi = C.c_int(), C.c_int()
    st = load_library().ctag_shard_range(n_total, rank, world, C.byref(lo), C.byref(hi))
    if st != 0:
        raise CtagError(st, "ctag_shard_range")
    return lo.value, hi.value


def packed_capacity(n):
    return int(load_library().ctag_packed_capacity(n))


def load_marker_file(path):
    """CylinderTag::load_from_file through the C ABI -> (state[int32 rows x cols], feature_size)."""
    L = load_library()
    p = C.POINTER(C.c_int32)()
    r, c, fs = C.c_int(), C.c_int(), C.c_int()
    st = L.ctag_load_marker_file(os.fsencode(path), C.byref(p), C.byref(r), C.byref(c), C.byref(fs))
    if st != 0:
        raise CtagError(st, path)
    try:
        state = np.ctypeslib.as_array(p, shape=(r.value, c.value)).copy()
    finally:
        L.ctag_free(p)
    return state, fs.value


class Model:
    """ctag_model: the reference's vector<ModelInfo> (CylinderTag::loadModel, CylinderTag.cpp:161-190)."""

    def __init__(self, path=None, ids=None, corners=None, model_size=None, base=None, axis=None):
        self.L = load_library()
        m = C.c_void_p()
        if path is not None:
            st = self.L.ctag_model_load(os.fsencode(path), C.byref(m))
        else:
            ids = np.ascontiguousarray(ids, np.int32)
            corners = np.ascontiguousarray(corners, np.float32)
            base = np.ascontiguousarray(base if base is not None else np.zeros((ids.size, 3)), np.float32)
            axis = np.ascontiguousarray(axis if axis is not None else np.zeros((ids.size, 3)), np.float32)
            fp = C.POINTER(C.c_float)
            v = ModelViewC(ids.size, int(model_size), ids.ctypes.data_as(C.POINTER(C.c_int32)), base.ctypes.data_as(fp),
                           axis.ctypes.data_as(fp), corners.ctypes.data_as(fp))
            st = self.L.ctag_model_create(C.byref(v), C.byref(m))
        if st != 0:
            raise CtagError(st, "model %s" % (path or "from arrays"))
        self.m = m

    def view(self):
        v = ModelViewC()
        self.L.ctag_model_get_view(self.m, C.byref(v))
        n, size = v.n_models, v.model_size
        return {"ids": np.ctypeslib.as_array(v.marker_id, (n,)).copy(), "size": size,
                "base": np.ctypeslib.as_array(v.base, (n, 3)).copy(), "axis": np.ctypeslib.as_array(v.axis, (n, 3)).copy(),
                "corners": np.ctypeslib.as_array(v.corners, (n, size * 8, 3)).copy()}

    def close(self):
        if getattr(self, "m", None):
            self.L.ctag_model_free(self.m)
            self.m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def load_camera(path):
    """CylinderTag::loadCamera (CylinderTag.cpp:192-196) through the C ABI -> CameraC."""
    cam = CameraC()
    st = load_library().ctag_camera_load(os.fsencode(path), C.byref(cam))
    if st != 0:
        raise CtagError(st, path)
    return cam


def make_camera(K, dist):
    cam = CameraC()
    for i, v in enumerate(np.asarray(K, np.float32).ravel()):
        cam.K[i] = float(v)
    d = np.asarray(dist, np.float32).ravel()
    for i in range(d.size):
        cam.dist[i] = float(d[i])
    cam.n_dist = int(d.size)
    return cam


class _Pinned:
    """Owner of one ctag_host_alloc() block (page-locked host memory)."""

    def __init__(self, nbytes):
        self.L = load_library()
        self.ptr = self.L.ctag_host_alloc(max(1, nbytes))
        if not self.ptr:
            raise MemoryError("ctag_host_alloc(%d)" % nbytes)
        self.buf = (C.c_ubyte * max(1, nbytes)).from_address(self.ptr)

    def __del__(self):
        if getattr(self, "ptr", None):
            self.L.ctag_host_free(self.ptr)
            self.ptr = None


def pinned_empty(shape, dtype=np.uint8):
    """numpy array in page-locked host memory (ctag_host_alloc); the memory lives as long as the array's base."""
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    owner = _Pinned(n)
    arr = np.frombuffer(owner.buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)
    arr_owner = owner  # keep alive through the ctypes buffer's _objects chain
    owner.buf._pinned_owner = arr_owner
    return arr


class Detector:
    """One ctag_handle (one GPU).  Mirrors the reference's usage: construct with the dictionary, call detect()."""

    def __init__(self, state, feature_size, device=0, params=None):
        """params: a ParamsC (default_params() edited) for ctag_create_ex; None = the reference's tunables (ctag_create)."""
        self.L = load_library()
        self.state = np.ascontiguousarray(state, dtype=np.int32)
        self.feature_size = int(feature_size)
        h = C.c_void_p()
        if params is None:
            st = self.L.ctag_create(self.state.ctypes.data_as(C.POINTER(C.c_int32)), self.state.shape[0],
                                    self.state.shape[1], self.feature_size, device, C.byref(h))
        else:
            st = self.L.ctag_create_ex(self.state.ctypes.data_as(C.POINTER(C.c_int32)), self.state.shape[0],
                                       self.state.shape[1], self.feature_size, device, C.byref(params), C.byref(h))
        if st != 0:
            raise CtagError(st, "ctag_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.ctag_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, opt, value):
        st = self.L.ctag_set_option(self.h, opt, int(value))
        if st != 0:
            raise CtagError(st, "ctag_set_option")

    # ---- host-memory entry points
    def detect(self, gray, adaptive_thresh=5, subpix=True, subpix_dist=5):
        gray = np.ascontiguousarray(gray, dtype=np.uint8)
        res = np.zeros(1, RESULT_DT)
        st = self.L.ctag_detect_u8(self.h, gray.ctypes.data, gray.shape[0], gray.shape[1], gray.strides[0],
                                   adaptive_thresh, int(subpix), subpix_dist, res.ctypes.data)
        if st < 0:
            raise CtagError(st, "ctag_detect_u8")
        return res[0]

    def submit(self, gray, adaptive_thresh=5, subpix=True, subpix_dist=5):
        """ctag_submit_u8: start one frame without waiting (at most two in flight); `gray` must stay alive until its collect()."""
        assert gray.dtype == np.uint8 and gray.ndim == 2 and gray.strides[1] == 1
        st = self.L.ctag_submit_u8(self.h, gray.ctypes.data, gray.shape[0], gray.shape[1], gray.strides[0], adaptive_thresh, int(subpix), subpix_dist)
        if st != 0:
            raise CtagError(st, "ctag_submit_u8")

    def collect(self, out=None):
        """ctag_collect: the record of the oldest submitted frame."""
        res = np.zeros(1, RESULT_DT) if out is None else out
        st = self.L.ctag_collect(self.h, res.ctypes.data)
        if st < 0:
            raise CtagError(st, "ctag_collect")
        return res[0]

    def detect_batch(self, frames, adaptive_thresh=5, subpix=True, subpix_dist=5, out=None):
        """frames: (n, rows, cols) uint8 in host memory; pass arrays from pinned_empty() to overlap upload and detection."""
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        n, rows, cols = frames.shape
        res = np.zeros(n, RESULT_DT) if out is None else out
        assert res.dtype == RESULT_DT and res.shape == (n,) and res.flags.c_contiguous
        st = self.L.ctag_detect_batch_u8(self.h, frames.ctypes.data, n, rows, cols, frames.strides[1], frames.strides[0],
                                         adaptive_thresh, int(subpix), subpix_dist, res.ctypes.data)
        if st != 0:
            raise CtagError(st, "ctag_detect_batch_u8")
        return res

    # ---- BGR frames (rows, cols, 3) uint8: cvtColor(BGR2GRAY) of main.cpp:36,52-54 happens on the device
    def detect_bgr(self, bgr, adaptive_thresh=5, subpix=True, subpix_dist=5):
        bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
        assert bgr.ndim == 3 and bgr.shape[2] == 3
        res = np.zeros(1, RESULT_DT)
        st = self.L.ctag_detect_bgr8(self.h, bgr.ctypes.data, bgr.shape[0], bgr.shape[1], bgr.strides[0], adaptive_thresh, int(subpix), subpix_dist,
                                     res.ctypes.data)
        if st < 0:
            raise CtagError(st, "ctag_detect_bgr8")
        return res[0]

    def detect_batch_bgr(self, frames, adaptive_thresh=5, subpix=True, subpix_dist=5, out=None):
        """frames: (n, rows, cols, 3) uint8 in host memory (pinned_empty() arrays overlap upload and detection)."""
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        n, rows, cols, ch = frames.shape
        assert ch == 3
        res = np.zeros(n, RESULT_DT) if out is None else out
        assert res.dtype == RESULT_DT and res.shape == (n,) and res.flags.c_contiguous
        st = self.L.ctag_detect_batch_bgr8(self.h, frames.ctypes.data, n, rows, cols, frames.strides[1], frames.strides[0], adaptive_thresh, int(subpix),
                                           subpix_dist, res.ctypes.data)
        if st != 0:
            raise CtagError(st, "ctag_detect_batch_bgr8")
        return res

    def detect_batch_bgr_device(self, frames_ptr, n, rows, cols, row_stride, frame_stride, out_ptr, adaptive_thresh=5, subpix=True, subpix_dist=5):
        st = self.L.ctag_detect_batch_bgr8_device(self.h, frames_ptr, n, rows, cols, row_stride, frame_stride, adaptive_thresh, int(subpix), subpix_dist,
                                                  out_ptr)
        if st != 0:
            raise CtagError(st, "ctag_detect_batch_bgr8_device")

    # ---- device-memory entry point (pointers are plain integers, e.g. torch.Tensor.data_ptr())
    def detect_batch_device(self, frames_ptr, n, rows, cols, row_stride, frame_stride, out_ptr, adaptive_thresh=5,
                            subpix=True, subpix_dist=5):
        st = self.L.ctag_detect_batch_device(self.h, frames_ptr, n, rows, cols, row_stride, frame_stride,
                                             adaptive_thresh, int(subpix), subpix_dist, out_ptr)
        if st != 0:
            raise CtagError(st, "ctag_detect_batch_device")

    def sync(self):
        st = self.L.ctag_sync(self.h)
        if st != 0:
            raise CtagError(st, "ctag_sync")

    def stream(self):
        return self.L.ctag_stream(self.h)

    def counters(self):
        """Per-frame counts of the last chunk: {name: (mean, max)} for components / candidates / quads / features / markers, plus
        'frames' and 'reruns' (frames completed through the any-frame workspace since the handle was created)."""
        c = CountersC()
        st = self.L.ctag_get_counters(self.h, C.byref(c))
        if st != 0:
            raise CtagError(st, "ctag_get_counters")
        out = {"frames": int(c.frames), "reruns": int(c.reruns)}
        for k, name in enumerate(COUNTER_NAMES):
            out[name] = (c.sum[k] / c.frames if c.frames else 0.0, int(c.max[k]))
        return out

    def timings(self):
        buf = (C.c_float * len(STAGE_NAMES))()
        n = self.L.ctag_get_timings(self.h, buf, len(STAGE_NAMES))
        return {STAGE_NAMES[i]: float(buf[i]) for i in range(n)}

    # ---- multi-GPU gather (include/ctag_gather.h); pointers are plain integers
    def _gcheck(self, st, what):
        if st != 0:
            raise CtagError(st, "%s: %s" % (what, self.L.ctag_comm_last_error(self.h).decode()))

    def pack_results(self, results_ptr, n, packed_ptr, capacity):
        nbytes = C.c_uint64()
        self._gcheck(self.L.ctag_pack_results(self.h, results_ptr, n, packed_ptr, capacity, C.byref(nbytes)), "ctag_pack_results")
        return int(nbytes.value)

    def unpack_results(self, packed_ptr, n, out_ptr):
        self._gcheck(self.L.ctag_unpack_results(self.h, packed_ptr, n, out_ptr), "ctag_unpack_results")

    def comm_init(self, id_bytes, rank, world):
        buf = (C.c_ubyte * COMM_ID_BYTES).from_buffer_copy(bytes(id_bytes))
        self._gcheck(self.L.ctag_comm_init(self.h, buf, rank, world), "ctag_comm_init")

    def comm_destroy(self):
        self.L.ctag_comm_destroy(self.h)

    def comm_native(self):
        return self.L.ctag_comm_native(self.h)

    def comm_attach(self, nccl_comm, rank, world):
        self._gcheck(self.L.ctag_comm_attach(self.h, nccl_comm, rank, world), "ctag_comm_attach")

    def gather_begin(self, local_ptr, n_local, n_total):
        self._gcheck(self.L.ctag_gather_begin(self.h, local_ptr, n_local, n_total), "ctag_gather_begin")

    def gather_end(self, out_ptr):
        self._gcheck(self.L.ctag_gather_end(self.h, out_ptr), "ctag_gather_end")

    def gather_wait(self):
        self._gcheck(self.L.ctag_gather_wait(self.h), "ctag_gather_wait")

    def gather(self, local_ptr, n_local, n_total, out_ptr):
        self._gcheck(self.L.ctag_gather(self.h, local_ptr, n_local, n_total, out_ptr), "ctag_gather")

    def gather_set_timeout(self, timeout_ms):
        """Deadline of the gather's host waits (ms; 0 none; < 0 the default: CTAG_GATHER_TIMEOUT_MS, else 60 s)."""
        self._gcheck(self.L.ctag_gather_set_timeout(self.h, int(timeout_ms)), "ctag_gather_set_timeout")

    def gather_last_bytes(self):
        a, b = C.c_uint64(), C.c_uint64()
        self.L.ctag_gather_last_bytes(self.h, C.byref(a), C.byref(b))
        return int(a.value), int(b.value)

    # ---- pose back end (include/ctag_pose.h)
    def estimate_pose(self, result, model, camera):
        """One frame: host ctag_frame_result record -> POSE_DT records, one per marker (CylinderTag::estimatePose
        before its erase of the model-less entries)."""
        res = np.ascontiguousarray(result).reshape(1)
        assert res.dtype == RESULT_DT
        n = int(res[0]["n_markers"]) if res[0]["status"] == 0 else 0
        out = np.zeros(max(n, 1), POSE_DT)
        st = self.L.ctag_estimate_pose(self.h, res.ctypes.data, model.m, C.byref(camera), out.ctypes.data)
        if st != 0:
            raise CtagError(st, "ctag_estimate_pose")
        return out[:n]

    def pose_batch_device(self, results_ptr, n_frames, model, camera, offsets_ptr, poses_ptr, capacity):
        st = self.L.ctag_pose_batch_device(self.h, results_ptr, n_frames, model.m, C.byref(camera), offsets_ptr, poses_ptr,
                                           capacity)
        if st != 0:
            raise CtagError(st, "ctag_pose_batch_device")

    def pose_last_ms(self):
        return float(self.L.ctag_pose_last_ms(self.h))
