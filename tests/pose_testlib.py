"""ctypes binding of the pose oracle (test infrastructure) + independent Python readers / camera model used to
check it.  Nothing here is imported by the product package."""
import ctypes as C
import os
import re

import numpy as np

from ctag_testlib import GOLDEN, RESULT_DT, ROOT, build_oracle

POSE_DT = np.dtype([("status", "<i4"), ("model_index", "<i4"), ("frame", "<i4"), ("marker", "<i4"),
                    ("n_points", "<i4"), ("iterations", "<i4"), ("rvec", "<f8", (3,)), ("tvec", "<f8", (3,)),
                    ("rvec0", "<f8", (3,)), ("tvec0", "<f8", (3,)), ("cost0", "<f8"), ("cost", "<f8")])
assert POSE_DT.itemsize == 136
MAX_POINTS = 160


class Camera(C.Structure):
    _fields_ = [("K", C.c_float * 9), ("dist", C.c_float * 14), ("n_dist", C.c_int32)]


class ModelView(C.Structure):
    _fields_ = [("n_models", C.c_int32), ("model_size", C.c_int32), ("marker_id", C.POINTER(C.c_int32)),
                ("base", C.POINTER(C.c_float)), ("axis", C.POINTER(C.c_float)), ("corners", C.POINTER(C.c_float))]


def read_model_file(path):
    """Independent reader of the .model text format (CylinderTag.cpp:161-190): ids, base, axis, corners[n][size*8][3]."""
    t = open(path).read().split()
    n, size = int(t[0]), int(t[1])
    p = 2
    ids = np.zeros(n, np.int32)
    base = np.zeros((n, 3), np.float32)
    axis = np.zeros((n, 3), np.float32)
    corners = np.zeros((n, size * 8, 3), np.float32)
    for i in range(n):
        ids[i] = int(t[p]); p += 1
        base[i] = [np.float32(x) for x in t[p:p + 3]]; p += 3
        axis[i] = [np.float32(x) for x in t[p:p + 3]]; p += 3
        for _ in range(size * 8):
            cid = int(t[p])
            corners[i, cid] = [np.float32(x) for x in t[p + 1:p + 4]]
            p += 4
    return {"ids": ids, "size": size, "base": base, "axis": axis, "corners": corners}


def read_camera_yml(path):
    """Independent reader of the two !!opencv-matrix nodes of cameraParams.yml."""
    txt = open(path).read()
    out = {}
    for name in ("cameraMatrix", "distCoeffs"):
        m = re.search(name + r":\s*!!opencv-matrix\s*rows:\s*(\d+)\s*cols:\s*(\d+)\s*dt:\s*(\w)\s*data:\s*\[([^\]]*)\]", txt)
        vals = [float(x) for x in m.group(4).replace("\n", " ").split(",")]
        out[name] = np.array(vals, np.float32).reshape(int(m.group(1)), int(m.group(2)))
    return out["cameraMatrix"], out["distCoeffs"].ravel()


def make_camera(K, dist):
    c = Camera()
    for i, v in enumerate(np.asarray(K, np.float32).ravel()):
        c.K[i] = float(v)
    d = np.asarray(dist, np.float32).ravel()
    for i in range(14):
        c.dist[i] = float(d[i]) if i < d.size else 0.0
    c.n_dist = int(d.size)
    return c


class _Held:
    pass


def make_model_view(model):
    h = _Held()
    h.ids = np.ascontiguousarray(model["ids"], np.int32)
    h.base = np.ascontiguousarray(model["base"], np.float32)
    h.axis = np.ascontiguousarray(model["axis"], np.float32)
    h.corners = np.ascontiguousarray(model["corners"], np.float32)
    v = ModelView()
    v.n_models = h.ids.size
    v.model_size = int(model["size"])
    v.marker_id = h.ids.ctypes.data_as(C.POINTER(C.c_int32))
    v.base = h.base.ctypes.data_as(C.POINTER(C.c_float))
    v.axis = h.axis.ctypes.data_as(C.POINTER(C.c_float))
    v.corners = h.corners.ctypes.data_as(C.POINTER(C.c_float))
    h.view = v
    return h


def rodrigues(r):
    r = np.asarray(r, np.float64)
    th = np.linalg.norm(r)
    if th < 1e-12:
        return np.eye(3)
    w = r / th
    Wx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    return np.cos(th) * np.eye(3) + np.sin(th) * Wx + (1 - np.cos(th)) * np.outer(w, w)


def project(K, dist, rvec, tvec, X, distort=True):
    """cv::projectPoints model (k1 k2 p1 p2 k3), float64."""
    K = np.asarray(K, np.float64)
    d = np.zeros(5)
    dd = np.asarray(dist, np.float64).ravel()
    d[:min(5, dd.size)] = dd[:5]
    P = X @ rodrigues(rvec).T + np.asarray(tvec, np.float64)
    x, y = P[:, 0] / P[:, 2], P[:, 1] / P[:, 2]
    if distort:
        r2 = x * x + y * y
        rad = 1 + d[0] * r2 + d[1] * r2 ** 2 + d[4] * r2 ** 3
        xd = x * rad + 2 * d[2] * x * y + d[3] * (r2 + 2 * x * x)
        yd = y * rad + d[2] * (r2 + 2 * y * y) + 2 * d[3] * x * y
        x, y = xd, yd
    return np.stack([K[0, 0] * x + K[0, 2], K[1, 1] * y + K[1, 2]], 1)


class PoseOracle:
    def __init__(self, path=None):
        path = path or os.path.join(ROOT, "oracle", "_build", "libctag_pose_oracle.so")
        if not os.path.exists(path):
            build_oracle()
        L = self.L = C.CDLL(path)
        pf, pd = C.POINTER(C.c_float), C.POINTER(C.c_double)
        L.ctago_undistort_points.argtypes = [C.POINTER(Camera), C.c_int, pf, C.c_int, pd]
        L.ctago_undistort_points.restype = None
        L.ctago_solve_pnp_epnp.argtypes = [C.POINTER(Camera), C.c_int, pf, pf, pd, pd]
        L.ctago_pose_ba.argtypes = [C.POINTER(Camera), C.c_int, pf, pf, pd, pd, pd, pd]
        L.ctago_build_correspondences.argtypes = [C.c_void_p, C.c_int, C.POINTER(ModelView), C.c_int, pf, pf,
                                                  C.POINTER(C.c_int)]
        L.ctago_linalg_probe.argtypes = [C.c_int, pd, pd]
        L.ctago_linalg_probe.restype = None
        L.ctago_pose_frame.argtypes = [C.c_void_p, C.POINTER(ModelView), C.POINTER(Camera), C.c_int, C.c_void_p]

    def linalg(self, op, data, n_out):
        data = np.ascontiguousarray(data, np.float64).ravel()
        out = np.zeros(n_out, np.float64)
        self.L.ctago_linalg_probe(op, data.ctypes.data_as(C.POINTER(C.c_double)), out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def undistort(self, cam, uv, with_P):
        uv = np.ascontiguousarray(uv, np.float32)
        out = np.zeros(uv.shape, np.float64)
        self.L.ctago_undistort_points(C.byref(cam), uv.shape[0], uv.ctypes.data_as(C.POINTER(C.c_float)), int(with_P),
                                      out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def epnp(self, cam, obj, img):
        obj = np.ascontiguousarray(obj, np.float32)
        img = np.ascontiguousarray(img, np.float32)
        r, t = np.zeros(3), np.zeros(3)
        st = self.L.ctago_solve_pnp_epnp(C.byref(cam), obj.shape[0], obj.ctypes.data_as(C.POINTER(C.c_float)),
                                         img.ctypes.data_as(C.POINTER(C.c_float)), r.ctypes.data_as(C.POINTER(C.c_double)),
                                         t.ctypes.data_as(C.POINTER(C.c_double)))
        return st, r, t

    def ba(self, cam, obj, img, rvec, tvec):
        obj = np.ascontiguousarray(obj, np.float32)
        img = np.ascontiguousarray(img, np.float32)
        r, t = np.array(rvec, np.float64), np.array(tvec, np.float64)
        c0, c1 = C.c_double(), C.c_double()
        it = self.L.ctago_pose_ba(C.byref(cam), obj.shape[0], obj.ctypes.data_as(C.POINTER(C.c_float)),
                                  img.ctypes.data_as(C.POINTER(C.c_float)), r.ctypes.data_as(C.POINTER(C.c_double)),
                                  t.ctypes.data_as(C.POINTER(C.c_double)), C.byref(c0), C.byref(c1))
        return it, r, t, c0.value, c1.value

    def correspondences(self, res, marker, mv, model_index):
        res = np.ascontiguousarray(res)
        obj = np.zeros((MAX_POINTS, 3), np.float32)
        img = np.zeros((MAX_POINTS, 2), np.float32)
        n = C.c_int()
        st = self.L.ctago_build_correspondences(res.ctypes.data, marker, C.byref(mv.view), model_index,
                                                obj.ctypes.data_as(C.POINTER(C.c_float)),
                                                img.ctypes.data_as(C.POINTER(C.c_float)), C.byref(n))
        return st, obj[:n.value], img[:n.value]

    def pose_frame(self, res, mv, cam, frame_index=0):
        res = np.ascontiguousarray(res)
        out = np.zeros(100, POSE_DT)
        n = self.L.ctago_pose_frame(res.ctypes.data, C.byref(mv.view), C.byref(cam), frame_index, out.ctypes.data)
        return out[:n]


def synth_pose_results(model, K, dist, n_frames, seed, noise_px=0.2, max_markers=5):
    """Detection records (RESULT_DT) whose corners are projections of the model under random poses (+ pixel noise), with
    feature id patterns that exercise every branch of the correspondence builder.  Returns (records, truth) where
    truth[f] is a list of (model_index, rvec, tvec)."""
    rng = np.random.default_rng(seed)
    res = np.zeros(n_frames, RESULT_DT)
    truth = []
    size = model["size"]
    id_patterns = [(3, 3), (3, 4), (2, 4), (1, 4), (5, -1), (0, 0), (6, 7), (7, 4)]
    for f in range(n_frames):
        nm = int(rng.integers(0, max_markers + 1))
        tf = []
        nfeat = 0
        r = res[f]
        r["status"] = 0
        for m in range(nm):
            nf = int(rng.integers(1, 8))
            if nfeat + nf > 100:
                break
            known = rng.random() < 0.85
            mi = int(rng.integers(0, model["ids"].size))
            marker_id = int(model["ids"][mi]) if known else 40  # 40: not in CTag_2f12c.model
            X = model["corners"][mi].astype(np.float64)
            c = X.mean(0)
            rv = rng.normal(0, 0.25, 3)
            dt = rng.normal(0, 1, 3) * np.array([40., 30., 60.])
            R = rodrigues(rv)
            tv = c - R @ c + dt
            p0 = int(rng.integers(0, size - nf + 1))
            pts = project(K, dist, rv, tv, X) + rng.normal(0, noise_px, (X.shape[0], 2))
            r["markers"][m] = (marker_id, nfeat, nf, nf)
            for j in range(nf):
                F = r["features"][nfeat + j]
                F["pos"] = p0 + j
                il, ir = id_patterns[int(rng.integers(0, len(id_patterns)))]
                F["id_left"], F["id_right"] = il, ir
                F["id"] = 8 * il + ir if ir >= 0 else -2
                F["corners"] = pts[(p0 + j) * 8:(p0 + j) * 8 + 8].astype(np.float32).ravel()
            nfeat += nf
            tf.append((mi if known else -1, rv, tv))
        r["n_markers"] = len(tf)
        r["n_features"] = nfeat
        truth.append(tf)
    return res, truth
