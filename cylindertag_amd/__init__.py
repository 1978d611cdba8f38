"""cylindertag_amd -- MI355X-native CylinderTag detection front end.

The product is the C-ABI shared library ``cylindertag_amd/_build/libctag_hip.so`` (hand-written HIP kernels for
gfx950, sources in ``cylindertag_amd/csrc``) plus the C++ class ``CylinderTag`` that mirrors the reference
interface (``csrc/CylinderTag.h``).  This Python module is only the thin ctypes binding the test-suite and
``bench.py`` use to drive that ABI; it contains no detection logic and no CPU fallback -- if the HIP library is
missing or no GPU is usable, construction raises.
"""
from .capi import (CameraC, ParamsC, default_params, CtagError, Detector, Model, POSE_DT, load_camera, make_camera, FEATURE_DT, MARKER_DT, RESULT_DT, STAGE_NAMES, build, lib_path, load_library,
                   load_marker_file, pinned_empty)

__all__ = ["CameraC", "ParamsC", "default_params", "Model", "POSE_DT", "load_camera", "make_camera", "CtagError", "Detector", "FEATURE_DT", "MARKER_DT", "RESULT_DT", "STAGE_NAMES", "build", "lib_path",
           "load_library", "load_marker_file", "pinned_empty"]
