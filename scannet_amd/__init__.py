"""scannet_amd -- MI355X-native RGB-D integration hot path of the ScanNet processing pipeline.

Host-side Python mirror of the reference interfaces for this path (SensReader, the TSDF `improve` stage,
Segmentator) over the C ABI of libscanfuse.so (include/scanfuse.h).  The compute path is hand-written HIP
for gfx950; importing a compute entry point without the built library raises (no CPU fallback).
"""
# Importing this package changes nothing in the process: the HIP runtime's GPU_MAX_HW_QUEUES (hardware queues per process, default 4; sf_fuse_run
# drives up to seven streams and wants 16) is the APPLICATION's to export before its first HIP call -- bench.py, tools/e2e_bench.py and the bin/
# tools do it in their own main(); sf_fuse_run leaves a note in sf_last_error() when it ran on fewer queues than streams.
RECOMMENDED_ENV = {"GPU_MAX_HW_QUEUES": "16"}

__all__ = ["synth", "RECOMMENDED_ENV"]
