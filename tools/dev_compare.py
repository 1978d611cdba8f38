"""Developer aid: run one frame through the HIP path and the oracle and print stage-by-stage differences."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from ctag_testlib import Oracle, read_bmp_gray, read_marker_file, GOLDEN, result_markers
import cylindertag_amd as ca
import testkit as tk
from cylindertag_amd import capi

def compare(img, state, fs, det, orc, name):
    t = time.time(); o = orc.detect(img, state, fs); t_or = time.time() - t
    det.set_option(capi.OPT_KEEP_PREMARKERS, 1)
    t = time.time(); r = det.detect(img); t_g = time.time() - t
    print("== %s: oracle %.1f ms, gpu call %.1f ms, status o=%d g=%d flags g=%d" % (name, t_or*1e3, t_g*1e3, o["status"], r["status"], r["flags"]))
    half = det.debug(0, tk.DBG_HALF).reshape(o["half"].shape)
    print("half diff px:", int((half != o["half"]).sum()))
    lab = det.debug(0, tk.DBG_LABELS).reshape(o["labels"].shape)
    print("binary diff px:", int(((lab > 0) != (o["binary"] > 0)).sum()))
    # partition equality
    ol = o["labels"]; pairs = np.unique(np.stack([ol.ravel(), lab.ravel()], 1), axis=0)
    print("label partition consistent:", len(np.unique(pairs[:,0])) == len(pairs) and len(np.unique(pairs[:,1])) == len(pairs), "ncomp", len(pairs)-1)
    cand = det.debug(0, tk.DBG_CANDIDATES); oc = o["candidates"]
    print("ncand gpu", len(cand), "oracle", len(oc))
    if len(cand) == len(oc) and len(oc):
        same = (cand[:, 0:5] == oc[:, 1:6]).all()
        print("cand area/bbox/order equal:", bool(same), " has_quad equal:", bool((cand[:,5]==oc[:,6]).all()), " n_boundary equal:", bool((cand[:,6]==oc[:,7]).all()))
        bad = np.nonzero((cand[:,5]!=oc[:,6]) | (cand[:,6]!=oc[:,7]))[0]
        print("  mismatching cands:", bad[:10], cand[bad[:5]], oc[bad[:5]])
        q = det.debug(0, tk.DBG_CAND_QUADS); oq = o["candidate_quads"]
        d = np.abs(q - oq).max() if len(q) else 0
        print("quad max abs diff:", d, " bit-equal:", bool((q.view(np.uint32)==oq.view(np.uint32)).all()))
    for st, what in enumerate([tk.DBG_FEATURES0, tk.DBG_FEATURES1, tk.DBG_FEATURES2]):
        f = det.debug(0, what); of = o["features"][st]
        if f.shape == of.shape and len(f):
            print("features stage %d: n=%d max abs diff %.3g bit-equal %s" % (st, len(f), np.abs(f-of).max(), bool((f.view(np.uint32)==of.view(np.uint32)).all())))
        else:
            print("features stage %d: shape gpu %s oracle %s" % (st, f.shape, of.shape))
    pre = det.debug(0, tk.DBG_PREMARKERS)
    print("premarkers bytes equal:", pre.tobytes() == o["premarkers"].tobytes(), pre["n_markers"], o["premarkers"]["n_markers"])
    print("result bytes equal:", r.tobytes() == o["result"].tobytes())
    for m in result_markers(r): print("  gpu   ", m["marker_id"], m["pos"], m["id"])
    for m in result_markers(o["result"]): print("  oracle", m["marker_id"], m["pos"], m["id"])

if __name__ == "__main__":
    state, fs = read_marker_file(os.path.join(GOLDEN, "CTag_2f12c.marker"))
    det = tk.Detector(state, fs); orc = Oracle()
    img = read_bmp_gray(os.path.join(GOLDEN, "test.bmp"))
    compare(img, state, fs, det, orc, "test.bmp 1920x1200")
    compare(img[60:1140], state, fs, det, orc, "test.bmp crop 1920x1080")
    for f in range(3):
        s, truth = tk.synth_frame_host(state, f)
        compare(s, state, fs, det, orc, "synth %d rows=%s" % (f, list(truth["dict_row"][:truth["n_markers"]])))
