"""scannet_amd -- MI355X-native RGB-D integration hot path of the ScanNet processing pipeline.

Host-side Python mirror of the reference interfaces for this path (SensReader, the TSDF `improve` stage,
Segmentator) over the C ABI of libscanfuse.so (include/scanfuse.h).  The compute path is hand-written HIP
for gfx950; importing a compute entry point without the built library raises (no CPU fallback).
"""
import os as _os

# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and kernels of streams that share one run
# one after the other: with the fuser's two streams, two copy streams and three inflate streams an inflate batch sat in the integrate pass's
# queue (sf_fuse_run 29 k -> 20 k frames/s in its loop).  Read at the first HIP call; a value the user exported wins.  libscanfuse.so does the
# same when it is loaded (csrc/pipeline.hip), for callers that are not Python.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")

__all__ = ["synth"]
