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
# ---- link-level check of the product's own OpenCV branch (INTEGRATION.md option B): needs OpenCV only, not Ceres ----------
# cylindertag_amd/csrc/CylinderTag.{h,cpp} built with -DCTAG_WITH_OPENCV against the REAL OpenCV (cv::Mat in detect(), cv::Mat
# camera / pose members, cv::Mat1i dictionary) into oracle/_ref/, the demo linked against it, and -- where a GPU and the HIP
# library exist -- run on test.bmp: its output must equal the stand-alone build's line for line.
PROD="$HERE/../cylindertag_amd"
if [ -f "$PROD/_build/libctag_hip.so" ]; then
    mkdir -p "$OUT"
    if $CXX -O2 -std=c++17 -fPIC -shared -DCTAG_WITH_OPENCV -I"$HERE/../include" $(pkg-config --cflags opencv4) -o "$OUT/libcylindertag_ocv.so" \
            "$PROD/csrc/CylinderTag.cpp" "$PROD/csrc/ctag_io.cpp" -L"$PROD/_build" -lctag_hip $(pkg-config --libs opencv4) -Wl,-rpath,"$PROD/_build" &&
       $CXX -O2 -std=c++17 -DCTAG_WITH_OPENCV -I"$HERE/../include" $(pkg-config --cflags opencv4) -o "$OUT/ctag_demo_ocv" "$PROD/examples/ctag_demo.cpp" \
            -L"$OUT" -lcylindertag_ocv -L"$PROD/_build" -lctag_hip $(pkg-config --libs opencv4) -Wl,-rpath,"$OUT" -Wl,-rpath,"$PROD/_build"; then
        say "CTAG_WITH_OPENCV branch: libcylindertag_ocv.so and ctag_demo_ocv link against OpenCV $(pkg-config --modversion opencv4)"
        G="$HERE/../tests/golden"
        if "$PROD/_build/ctag_demo" "$G/CTag_2f12c.marker" "$G/test.bmp" 5 1 5 "$G/CTag_2f12c.model" "$G/cameraParams.yml" > "$OUT/demo_plain.txt" 2>/dev/null; then
            "$OUT/ctag_demo_ocv" "$G/CTag_2f12c.marker" "$G/test.bmp" 5 1 5 "$G/CTag_2f12c.model" "$G/cameraParams.yml" > "$OUT/demo_ocv.txt" 2>&1
            if cmp -s "$OUT/demo_plain.txt" "$OUT/demo_ocv.txt"; then say "CTAG_WITH_OPENCV branch: ctag_demo_ocv output == ctag_demo output on test.bmp (detect + estimatePose)";
            else say "CTAG_WITH_OPENCV branch: OUTPUT DIFFERS (oracle/_ref/demo_plain.txt vs demo_ocv.txt)"; exit 1; fi
        else
            say "CTAG_WITH_OPENCV branch: linked; not run (no usable GPU on this host)"
        fi
    else
        say "CTAG_WITH_OPENCV branch: DOES NOT BUILD against this OpenCV"; exit 1
    fi
else
    say "CTAG_WITH_OPENCV branch: libctag_hip.so not built yet -> link check skipped"
fi
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
