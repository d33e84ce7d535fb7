#!/bin/bash
# TEST INFRASTRUCTURE: bin/depthsensing's source against the real host half of the library + the fake device half (fake_fuser.cpp), into $1.
#   bash tests/fake_fuser/build.sh <scratch dir>   ->  <scratch dir>/depthsensing, <scratch dir>/libscanfuse.so
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$1
SRC=$ROOT/scannet_amd/csrc
mkdir -p "$OUT"
g++ -O1 -g -std=c++17 -fPIC -shared -fvisibility=hidden -I"$ROOT/include" -I"$SRC" -I"$ROOT/tools/tsan/fake_hip" \
    "$ROOT/tests/fake_fuser/fake_fuser.cpp" "$SRC/ply.cpp" "$SRC/sens.cpp" "$SRC/zlib_codec.cpp" "$SRC/jpeg.cpp" "$SRC/png.cpp" "$SRC/occipital.cpp" "$SRC/params.cpp" \
    -o "$OUT/libscanfuse.so" -lpthread
g++ -O1 -g -std=c++17 -I"$ROOT/include" "$SRC/tool_depthsensing.cpp" -o "$OUT/depthsensing" -L"$OUT" -lscanfuse -Wl,-rpath,"$OUT" -lpthread
