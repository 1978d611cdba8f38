#!/usr/bin/env python3
"""Which kernels a committed profile belongs to: sha256 over the library's sources (the GPU box has no .git, so a profile
cannot carry a commit id; it carries these instead and bench.py compares them with the tree it runs from).
    all    every file under cylindertag_amd/csrc + include/
    sweep  the files the threshold+label sweep is built from (k_sweep.hip + the shared internal header)
`python tools/srcsha.py` prints both."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SWEEP_FILES = ["cylindertag_amd/csrc/k_sweep.hip", "cylindertag_amd/csrc/ctag_internal.h"]


def _sha(paths):
    h = hashlib.sha256()
    for p in sorted(paths):
        h.update(p.encode() + b"\0")
        with open(os.path.join(ROOT, p), "rb") as f:  # (whole-line comments and blank lines do not count: a reworded comment is the same kernel)
            code = b"\n".join(l for l in (x.strip() for x in f.read().split(b"\n")) if l and not l.startswith(b"//"))
            h.update(hashlib.sha256(code).digest())
    return h.hexdigest()


def sources_sha256():
    files = []
    for d in ("cylindertag_amd/csrc", "include"):
        files += [d + "/" + f for f in os.listdir(os.path.join(ROOT, d)) if f.endswith((".hip", ".h", ".cpp"))]
    return {"all": _sha(files), "sweep": _sha(SWEEP_FILES)}


def git_head():
    try:
        import subprocess
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:  # noqa: BLE001  (no .git on the GPU box)
        return None


if __name__ == "__main__":
    import json
    print(json.dumps({"sources_sha256": sources_sha256(), "git_head": git_head()}))
