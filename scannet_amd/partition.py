"""One large scan over several GPUs (BASELINE configs[4]): slab partition of the block space along x.

Integration is independent per block given the frame, so every rank sees every depth frame (614 KB) and fuses only the
blocks whose x block-coordinate lies in its slab [planes[r], planes[r+1]) -- `Fuser.set_slab`.  The only exchange step
comes before marching cubes: a cube on the last voxel layer of a slab needs the +1 neighbours, i.e. the lowest block
layer of the slab above.  Each rank exports that one-block-thick layer, the layers are all-gathered (torch.distributed:
RCCL over xGMI with device tensors under the nccl backend, gloo on CPU tensors in the tests) and each rank imports the
layer that sits on its upper plane as GHOST blocks (read as neighbours, never fused or meshed).

Why x: the canonical mesh orders vertices by edge key and triangles by cube key, and x is the most significant field
of both -- the per-slab meshes concatenate, in slab order, into exactly the mesh one GPU would have produced; boundary
vertices (an edge shared by cubes of two slabs yields the same key and the same position on both ranks) are welded by key.
"""
import numpy as np


def slab_planes(x_lo_block, x_hi_block, world):
    """world + 1 block planes splitting [x_lo_block, x_hi_block) evenly; the outer slabs are open-ended."""
    edges = [int(round(x_lo_block + (x_hi_block - x_lo_block) * r / world)) for r in range(world + 1)]
    edges[0], edges[-1] = -(1 << 20) + 1, (1 << 20) - 1   # 21-bit block coordinates
    return edges


def planes_from_poses(poses, voxel_size, max_depth, world):
    """Slab planes from the camera trajectory: x extent of the camera centres +- the integration range."""
    t = np.asarray(poses, np.float32).reshape(-1, 4, 4)[:, 0, 3]
    t = t[np.isfinite(t)]
    block = 8.0 * voxel_size
    lo = int(np.floor((t.min() - max_depth) / block)) if len(t) else 0
    hi = int(np.ceil((t.max() + max_depth) / block)) if len(t) else 1
    return slab_planes(lo, hi, world)


def _all_gather_ragged(arr, group=None):
    """All-gather numpy arrays whose first dimension differs per rank (pad to the maximum, trim after)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    n = torch.tensor([arr.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    m = max(counts + [1])
    flat = np.zeros((m,) + arr.shape[1:], arr.dtype)
    flat[:arr.shape[0]] = arr
    t = torch.from_numpy(flat.view(np.uint8).reshape(m, -1)).to(dev)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t, group=group)   # the boundary all-gather of the north star (padded to the max count)
    res = []
    for c, o in zip(counts, outs):
        a = o.cpu().numpy().reshape(-1).view(arr.dtype).reshape((m,) + arr.shape[1:])[:c]
        res.append(a)
    return res


def exchange_boundary_layers(fuser, planes, rank, group=None, gather=None):
    """Export this rank's lowest block layer, all-gather, import the layer on this rank's upper plane as ghosts.
    `gather(coords, voxels) -> (list of coords, list of voxels)` replaces torch.distributed (single-process tests).
    Returns (blocks sent, ghost blocks received)."""
    lo, hi = planes[rank], planes[rank + 1]
    coords, vox = fuser.export_blocks_where(0, lo, lo + 1) if rank > 0 else (np.zeros((0, 3), np.int32), np.zeros((0, 512), fuser_voxel_dtype()))
    if gather is None:
        all_c = _all_gather_ragged(coords, group)
        all_v = _all_gather_ragged(vox, group)
    else:
        all_c, all_v = gather(coords, vox)
    got = 0
    if rank + 1 < len(planes) - 1:
        c, v = all_c[rank + 1], all_v[rank + 1]
        keep = c[:, 0] == hi if len(c) else np.zeros(0, bool)
        if keep.any():
            fuser.import_blocks(c[keep], v[keep], ghost=True)
            got = int(keep.sum())
    return len(coords), got


def fuser_voxel_dtype():
    from .fusion import VOXEL_DTYPE
    return VOXEL_DTYPE


def merge_slab_meshes(parts):
    """parts: per rank, in slab order, (xyz [n,3] f32, rgba [n,4] u8, tris [m,3] u32, keys [n] u64) of the rank's canonical
    mesh.  Returns the canonical mesh of the whole scan: vertices unique by key in key order, triangles concatenated."""
    keys = np.concatenate([p[3] for p in parts]) if parts else np.zeros(0, np.uint64)
    xyz = np.concatenate([p[0] for p in parts]) if parts else np.zeros((0, 3), np.float32)
    rgba = np.concatenate([p[1] for p in parts]) if parts else np.zeros((0, 4), np.uint8)
    ukeys, first = np.unique(keys, return_index=True)
    tris = [np.searchsorted(ukeys, p[3][p[2].astype(np.int64)]).astype(np.uint32) for p in parts if len(p[2])]
    tris = np.concatenate(tris) if tris else np.zeros((0, 3), np.uint32)
    return xyz[first], rgba[first], tris, ukeys
