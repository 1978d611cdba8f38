"""ASan/UBSan run of the CPU oracles ("ASan/UBSan on the CPU restatement" -- it transliterates code full
of fixed arrays and index quirks, App. B).  `make -C oracle asan` builds both oracles with
-fsanitize=address,undefined; a child Python with libasan preloaded pushes test.bmp (several parameter sets), frames of
the test.avi substitute, synthetic marker frames, noise, blank / dark / tiny / odd-sized frames and the pose fixtures
through them.  Any report aborts the child."""
import os
import subprocess
import sys

import numpy as np
import pytest

from ctag_testlib import GOLDEN, ROOT, read_bmp_gray
from sequences import avi_substitute


def _libasan():
    p = subprocess.check_output(["g++", "-print-file-name=libasan.so"]).decode().strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_oracles_are_clean_under_asan_ubsan(tmp_path, dictionary):
    import cylindertag_amd as ca
    import testkit as tk
    lib = _libasan()
    if lib is None:
        pytest.skip("libasan.so not found next to g++")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    state, fs = dictionary
    bmp = read_bmp_gray(os.path.join(GOLDEN, "test.bmp"))
    rng = np.random.RandomState(9)
    inputs = {"test_bmp": bmp}
    for k, f in enumerate(avi_substitute(bmp, 4)):
        inputs["seq%d" % k] = f
    for f in range(8):  # the 8 synthetic goldens' inputs
        inputs["synth%d" % f] = tk.synth_frame_host(state, f)[0]
    inputs["noise"] = rng.randint(0, 256, (360, 500)).astype(np.uint8)
    inputs["dark_noise"] = rng.randint(0, 40, (300, 420)).astype(np.uint8)       # dense low-level speckle
    inputs["blank"] = np.full((200, 320), 180, np.uint8)
    inputs["all_dark"] = np.zeros((128, 160), np.uint8)
    inputs["tiny"] = rng.randint(0, 256, (4, 4)).astype(np.uint8)
    inputs["thin"] = rng.randint(0, 256, (5, 301)).astype(np.uint8)
    inputs["odd"] = np.ascontiguousarray(bmp[1:1200, 3:1914][200:745, 300:1151])    # odd rows and cols
    inputs["ragged"] = np.ascontiguousarray(bmp[3:1001, 5:1711])
    path = str(tmp_path / "inputs.npz")
    np.savez(path, **inputs)
    env = dict(os.environ, LD_PRELOAD=lib, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:verify_asan_link_order=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sanitize_driver.py"), path], env=env, capture_output=True,
                       text=True, timeout=900)
    assert p.returncode == 0, "sanitizer run failed:\n" + p.stdout[-2000:] + p.stderr[-6000:]
    assert "clean" in p.stdout
