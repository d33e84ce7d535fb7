#!/bin/bash
# ThreadSanitizer run of the host-side thread pools without a GPU in them (tools/tsan/host_threads.cpp): the streaming .sens writer, the image export
# pool, the mesh merge and the PLY writer.      bash tools/tsan/run_host.sh
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/scannet_amd/_build_san
mkdir -p $OUT
SRC=$ROOT/scannet_amd/csrc
g++ -std=c++17 -O1 -g -fsanitize=thread -fno-omit-frame-pointer -I$ROOT/include -I$SRC $ROOT/tools/tsan/host_threads.cpp $SRC/sens.cpp $SRC/sens_writer.cpp $SRC/sens_images.cpp \
    $SRC/ply.cpp $SRC/zlib_codec.cpp $SRC/jpeg.cpp $SRC/png.cpp $SRC/occipital.cpp $SRC/params.cpp -o $OUT/tsan_host_threads -lpthread
D=$(mktemp -d /tmp/sf_tsan_host_XXXXXX)
trap 'rm -rf "$D"' EXIT
TSAN_OPTIONS="halt_on_error=1 exitcode=66" $OUT/tsan_host_threads "$D"
