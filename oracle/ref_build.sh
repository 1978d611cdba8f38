#!/bin/bash
# oracle/_ref recipe: compiles the REAL reference path (/root/reference/corner_detector.cpp + CylinderTag.cpp +
# pose_estimation.cpp, where they lie, with the reference's own headers) against a REAL system OpenCV 4 (+ Ceres / Eigen /
# glog, which header/config.h:24-27 pulls into every translation unit) and links oracle/ref_driver.cpp, our dump tool.
# Outputs only into oracle/_ref/.  It never copies reference sources and never substitutes missing headers or libraries:
# where a dependency is absent the reference is UNBUILDABLE, this script says so and exits 0 without producing anything,
# and the oracle stays "parity unpinned" (DESIGN.md 2).  That is the case in this image (no OpenCV, no Ceres, no
# pkg-config); the recipe exists so that the first machine that has them can turn "partial" into "green":
#     make -C oracle ref && python -m pytest tests/test_oracle_ref_cpu.py
set -u
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${CTAG_REFERENCE_DIR:-/root/reference}"
OUT="$HERE/_ref"
say() { echo "ref_build: $*"; }
[ -f "$REF/corner_detector.cpp" ] || { say "reference checkout not found at $REF -> skipped"; exit 0; }
command -v pkg-config >/dev/null 2>&1 || { say "pkg-config absent -> reference unbuildable here (skipped)"; exit 0; }
pkg-config --exists opencv4 || { say "no system OpenCV 4 (pkg-config opencv4) -> reference unbuildable here (skipped)"; exit 0; }
CXX="${CXX:-g++}"
probe() { echo "#include <$1>" | $CXX -std=c++17 $(pkg-config --cflags opencv4) ${EIGEN_CFLAGS:-} -x c++ -fsyntax-only - >/dev/null 2>&1; }
EIGEN_CFLAGS="$(pkg-config --cflags eigen3 2>/dev/null || echo -I/usr/include/eigen3)"
for hdr in opencv2/gapi/core.hpp Eigen/Dense ceres/ceres.h glog/logging.h; do
    probe "$hdr" || { say "<$hdr> (header/config.h) not found -> reference unbuildable here (skipped)"; exit 0; }
done
mkdir -p "$OUT"
say "OpenCV $(pkg-config --modversion opencv4) (the reference pins 4.5.3, Release.props:11)"
set -e
$CXX -O2 -std=c++17 -ffp-contract=off -I"$REF/header" -I"$HERE/../include" $(pkg-config --cflags opencv4) $EIGEN_CFLAGS \
    "$HERE/ref_driver.cpp" "$REF/corner_detector.cpp" "$REF/CylinderTag.cpp" "$REF/pose_estimation.cpp" \
    -o "$OUT/ref_driver" $(pkg-config --libs opencv4) -lceres -lglog -lpthread
GOLD="$HERE/../tests/golden"
"$OUT/ref_driver" "$GOLD/CTag_2f12c.marker" "$GOLD/test.bmp" "$OUT/test_bmp"
say "wrote $OUT/test_bmp.{half,binary,components,quads,result}.bin ; now run tests/test_oracle_ref_cpu.py"
