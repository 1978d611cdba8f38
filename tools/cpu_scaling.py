#!/usr/bin/env python3
"""How the CPU oracle's frame-parallel rate scales with the thread count on this host (why bench.py's all_cores figure
is what it is): hardware threads, cgroup CPU quota, and ctago_detect_many frames/s for 1..hardware_concurrency threads."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cylindertag_amd as ca  # noqa: E402
import testkit as tk  # noqa: E402
from ctag_testlib import GOLDEN, Oracle, read_marker_file  # noqa: E402

state, fs = read_marker_file(os.path.join(GOLDEN, "CTag_2f12c.marker"))
orc = Oracle()
hw = orc.L.ctago_hardware_concurrency()
print("hardware_concurrency", hw, "affinity", len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(p):
        print(p, open(p).read().strip())
print("loadavg", open("/proc/loadavg").read().strip())
frames = np.stack([tk.synth_frame_host(state, f)[0] for f in range(64)])
t = 1
while t <= hw:
    n = max(16, min(4 * t, 1024))
    batch = frames[np.arange(n) % 64]
    t0 = time.perf_counter()
    orc.detect_many(batch, state, fs, threads=t)
    dt = time.perf_counter() - t0
    print("threads %4d  frames %5d  %.1f frames/s  (%.2f per thread)" % (t, n, n / dt, n / dt / t), flush=True)
    t *= 2
