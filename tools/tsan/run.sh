#!/bin/bash
# ThreadSanitizer run of sf_fuse_run (decode pool + pinned ring + copy streams + reaper) against the asynchronous fake HIP runtime.
#   bash tools/tsan/run.sh [frames] [decode threads]
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/scannet_amd/_build_san
mkdir -p $OUT
SRC=$ROOT/scannet_amd/csrc
FLAGS="-std=c++17 -O1 -g -fsanitize=thread -fno-omit-frame-pointer -I$ROOT/tools/tsan/fake_hip -I$ROOT/include -I$SRC"
g++ $FLAGS -x c++ $SRC/pipeline.hip -x none $ROOT/tools/tsan/fake_hip.cpp $ROOT/tools/tsan/harness.cpp $SRC/sens.cpp $SRC/zlib_codec.cpp $SRC/jpeg.cpp $SRC/png.cpp \
    $SRC/occipital.cpp $SRC/params.cpp -o $OUT/tsan_fuse_run -lpthread
TSAN_OPTIONS="halt_on_error=1 exitcode=66 second_deadlock_stack=1" $OUT/tsan_fuse_run "${1:-300}" "${2:-8}"
echo "tsan: clean"
