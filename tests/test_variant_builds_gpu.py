"""Builds of the library with a compile-time switch flipped, each against the oracle with the matching switch (tests/variant_worker.py in a child process:
the binding loads ONE library per process).

* `-DCTAG_WELSCH_MINERR_IN_LOOP` / `-DCTAG_RESIZE_SIMD_LANES=16`: the two assumptions about OpenCV 4.5.3 that nothing in this image can check (README, oracle/
  ctag_oracle.cpp: OracleVariants) -- whichever a host with cv2 (tests/cv2_pins.py) finds true is a flag, not a rewrite, and stays byte-identical to the oracle.
* `-DCTAG_REFINE_PLAIN_SYNC=1`: edgeRefine's LDS-only waits replaced by __syncthreads(); same records (the waits' assumptions are otherwise unchecked)."""
import json
import os
import subprocess
import sys

import pytest

from ctag_testlib import ROOT

pytestmark = pytest.mark.gpu


def _variant(name, extra):
    out = os.path.join("_var", name)
    subprocess.check_call(["make", "-s", "-j4", "-C", os.path.join(ROOT, "cylindertag_amd"), "OUT=" + out, "EXTRA=" + extra, os.path.join(out, "libctag_hip.so")])
    return os.path.join(ROOT, "cylindertag_amd", out, "libctag_hip.so")


def _run(lib, welsch, lanes):
    env = dict(os.environ)
    if lib:
        env["CTAG_HIP_LIB"] = lib
    else:
        env.pop("CTAG_HIP_LIB", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "variant_worker.py"), "--welsch", str(welsch), "--lanes", str(lanes)],
                       capture_output=True, text=True, timeout=1200, env=env)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert lines, p.stderr[-3000:]
    return p.returncode, json.loads(lines[-1])


def test_the_opencv_assumptions_are_switches_on_both_sides():
    lib = _variant("hedge", "-DCTAG_WELSCH_MINERR_IN_LOOP -DCTAG_RESIZE_SIMD_LANES=16")
    rc, rep = _run(lib, 1, 16)
    assert rc == 0 and not rep["mismatches"], rep
    assert rep["oracle_records_that_differ_from_the_default_oracle"] > 0, "the switches change nothing on these frames: the test would not notice a missing one"
    # and the switches matter: the DEFAULT build is not the variant oracle's equal (it is the default oracle's -- every other GPU test)
    rc, rep = _run(None, 1, 16)
    assert rc == 1 and rep["mismatches"], rep


def test_plain_barriers_in_edge_refine_give_the_same_records():
    lib = _variant("plainsync", "-DCTAG_REFINE_PLAIN_SYNC=1")
    rc, rep = _run(lib, 0, 8)
    assert rc == 0 and not rep["mismatches"], rep
