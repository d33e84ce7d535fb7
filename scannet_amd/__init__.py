"""scannet_amd -- MI355X-native RGB-D integration hot path of the ScanNet processing pipeline.

Host-side Python mirror of the reference interfaces for this path (SensReader, the TSDF `improve` stage,
Segmentator) over the C ABI of libscanfuse.so (include/scanfuse.h).  The compute path is hand-written HIP
for gfx950; importing a compute entry point without the built library raises (no CPU fallback).
"""
__all__ = ["synth"]
