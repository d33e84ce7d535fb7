"""One large scan over several GPUs (BASELINE configs[4]): the block space partitioned along x.

Integration is independent per block given the frame, so every rank sees every depth frame (614 KB) and fuses only the
blocks it owns.  Two ways to own blocks (include/scanfuse.h):

  * one contiguous SLAB per rank, [planes[r], planes[r+1]) -- `Fuser.set_slab`.  Fewest boundary blocks, but a camera that
    is inside one slab keeps one GPU busy and leaves the others idle: the scan is no faster than on one GPU.
  * STRIPES `thickness` block layers thick dealt round-robin -- `Fuser.set_stripes`.  Every frame's blocks spread over all
    ranks (the default here: 16 layers = 0.5 m at 4 mm voxels, a fraction of the view frustum), 1/16 of the blocks take
    part in the exchange.

The only exchange step comes before marching cubes: a cube on the last voxel layer of a slab / stripe needs the +1
neighbours, i.e. the lowest block layer of whatever lies above it.  `exchange_boundary` does it without touching host
memory: sf_fuser_export_boundary writes that layer of every owned stripe into a device buffer, the buffers are
all-gathered (torch.distributed: RCCL over xGMI under the nccl backend; gloo with CPU tensors in the tests), and
sf_fuser_import_ghosts keeps, on the device, exactly the blocks this rank needs as GHOSTS (read as neighbours, never
fused or meshed).

Why x: the canonical mesh orders vertices by edge key and triangles by cube key, and x is the most significant field
of both -- the per-rank meshes merge by key into exactly the mesh one GPU would have produced; boundary vertices (an
edge shared by cubes of two ranks yields the same key and the same position on both) are welded by key.
"""
import numpy as np

STRIPE_BLOCKS = 16


def slab_planes(x_lo_block, x_hi_block, world):
    """world + 1 block planes splitting [x_lo_block, x_hi_block) evenly; the outer slabs are open-ended."""
    edges = [int(round(x_lo_block + (x_hi_block - x_lo_block) * r / world)) for r in range(world + 1)]
    edges[0] = -(1 << 20) + 1   # 21-bit block coordinates
    for r in range(1, world + 1):
        # strictly increasing whatever the extent: an extent narrower than `world` layers used to give equal planes, i.e. an EMPTY slab -- and the
        # ring shift of exchange_boundary hands rank r's lowest layer to rank r - 1 only: with an empty slab in between, the rank that needed
        # the layer never saw it (a silent seam in the merged mesh)
        edges[r] = max(edges[r], edges[r - 1] + 1)
    edges[-1] = max(edges[-1], (1 << 20) - 1)
    return edges


def planes_from_poses(poses, voxel_size, max_depth, world):
    """Slab planes from the camera trajectory: x extent of the camera centres +- the integration range."""
    t = np.asarray(poses, np.float32).reshape(-1, 4, 4)[:, 0, 3]
    t = t[np.isfinite(t)]
    block = 8.0 * voxel_size
    lo = int(np.floor((t.min() - max_depth) / block)) if len(t) else 0
    hi = int(np.ceil((t.max() + max_depth) / block)) if len(t) else 1
    return slab_planes(lo, hi, world)


def planes_from_histogram(x_blocks, world):
    """Slab planes that give every rank the same number of blocks: `x_blocks` = x block-coordinates of the blocks a prefix of the
    scan allocated (one fuser without a partition, or the all-gathered coordinates).  Quantiles of that histogram."""
    x = np.sort(np.asarray(x_blocks, np.int64).reshape(-1))
    edges = [-(1 << 20) + 1]
    for r in range(1, world):
        edges.append(int(x[min(len(x) - 1, (len(x) * r) // world)]) if len(x) else 0)
    edges.append((1 << 20) - 1)
    for r in range(1, world + 1):   # strictly increasing, whatever the histogram looks like
        edges[r] = max(edges[r], edges[r - 1] + 1)
    return edges


def owner_of(x_block, origin, thickness, world):
    """Rank that owns block layer x under set_stripes(0, origin, thickness, world, rank) -- the host-side mirror of slab_owns."""
    return int(np.floor_divide(int(x_block) - int(origin), int(thickness)) % int(world))


def _dist_device(group=None):
    import torch
    import torch.distributed as dist
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")


def _all_gather_ragged(arr, group=None):
    """All-gather arrays whose first dimension differs per rank (pad to the maximum, trim after).  numpy in -> list of numpy out
    (staged through the backend's device); torch tensor in (already on the backend's device) -> list of tensors, no host copy."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = _dist_device(group)
    as_numpy = isinstance(arr, np.ndarray)
    t_in = torch.from_numpy(np.ascontiguousarray(arr)).to(dev) if as_numpy else arr
    n = torch.tensor([t_in.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    m = max(counts + [1])
    flat = torch.zeros((m,) + tuple(t_in.shape[1:]), dtype=t_in.dtype, device=dev)
    flat[:t_in.shape[0]] = t_in
    outs = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(outs, flat, group=group)   # the boundary all-gather of the north star (padded to the max count)
    if as_numpy:
        return [o[:c].cpu().numpy() for c, o in zip(counts, outs)]
    return [o[:c] for c, o in zip(counts, outs)]


def _ring_shift(c, v, group=None):
    """Every rank sends (c, v) to its LEFT neighbour and receives its RIGHT neighbour's: the whole exchange of the stripe / slab partition.
    The lowest layer of a stripe of rank r sits right above the highest layer of a stripe of rank r - 1 (stripes are dealt round-robin,
    slabs in rank order), so rank r's boundary blocks are wanted by rank (r - 1) mod N and by nobody else.  torch.distributed point-to-point
    (batch_isend_irecv: ncclSend / ncclRecv in one group under the nccl backend -- on a node whose GPUs are fully connected by xGMI each pair
    of neighbours has its own link, so the N transfers run at link speed side by side instead of every rank receiving everything)."""
    import torch
    import torch.distributed as dist
    world, me = dist.get_world_size(group), dist.get_rank(group)
    peer = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    left, right = peer((me - 1) % world), peer((me + 1) % world)
    dev = c.device
    n_out = torch.tensor([c.shape[0]], dtype=torch.int64, device=dev)
    n_in = torch.zeros(1, dtype=torch.int64, device=dev)
    for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, n_out, left, group), dist.P2POp(dist.irecv, n_in, right, group)]):
        req.wait()
    nr = int(n_in.item())
    rc = torch.empty((nr, 3), dtype=c.dtype, device=dev)
    rv = torch.empty((nr, 4096), dtype=v.dtype, device=dev)
    ops = []
    if c.shape[0]:
        ops += [dist.P2POp(dist.isend, c.contiguous(), left, group), dist.P2POp(dist.isend, v.contiguous(), left, group)]
    if nr:
        ops += [dist.P2POp(dist.irecv, rc, right, group), dist.P2POp(dist.irecv, rv, right, group)]
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return rc, rv


def exchange_boundary(fuser, rank=None, group=None, gather=None, mode="neighbour"):
    """Before meshing: export this rank's boundary layers, hand them to the rank that needs them, import what arrives as ghosts.
    mode "neighbour" (default): one send to the left neighbour, one receive from the right one (`_ring_shift`); "all_gather": every rank
    receives every rank's layers and keeps what it wants (the round-2 form; the north star's wording).  With a process group on the nccl
    backend everything stays in HBM (device export -> RCCL -> device import with the ownership filter in the kernel).
    `gather(coords, voxels) -> (list, list)` replaces torch.distributed in single-process tests (numpy).
    Returns (blocks sent, ghost blocks received); `exchange_boundary.last_bytes` = payload bytes this rank received."""
    if gather is not None:
        c, v = fuser.export_boundary()
        all_c, all_v = gather(c, v)
        me = rank
    else:
        import torch
        import torch.distributed as dist
        me = dist.get_rank(group) if rank is None else rank
        dev = _dist_device(group)
        if dev.type == "cuda":
            n = fuser.count_boundary()
            c = torch.empty((max(n, 1), 3), dtype=torch.int32, device=dev)
            v = torch.empty((max(n, 1), 4096), dtype=torch.uint8, device=dev)
            n = fuser.export_boundary(c, v) if n else 0
            c, v = c[:n], v[:n]
        else:   # gloo: host tensors (CPU tests; two processes sharing one GPU)
            cn, vn = fuser.export_boundary()
            c, v = torch.from_numpy(cn), torch.from_numpy(np.ascontiguousarray(vn).view(np.uint8).reshape(len(cn), 4096))
        if mode == "neighbour":
            got = 0
            if dist.get_world_size(group) > 1:
                rc, rv = _ring_shift(c, v, group)
                if dev.type == "cuda":
                    # the import kernel runs on the fuser's own stream: what RCCL queued on torch's stream must have landed first
                    torch.cuda.current_stream().synchronize()
                elif len(rc):
                    rc, rv = rc.numpy(), rv.numpy()
                got = fuser.import_ghosts(rc, rv) if len(rc) else 0
                exchange_boundary.last_bytes = int(len(rc)) * 4108
            else:
                exchange_boundary.last_bytes = 0
            return int(len(c)), got
        all_c = _all_gather_ragged(c, group)
        all_v = _all_gather_ragged(v, group)
        if dev.type == "cuda":
            torch.cuda.current_stream().synchronize()   # order RCCL's output before the import on the fuser's stream
    got = 0
    exchange_boundary.last_bytes = 0
    for r, (cc, vv) in enumerate(zip(all_c, all_v)):
        if r == me or len(cc) == 0:
            continue
        if not isinstance(cc, np.ndarray) and cc.device.type == "cpu":
            cc, vv = cc.numpy(), vv.numpy()
        exchange_boundary.last_bytes += int(len(cc)) * 4108
        got += fuser.import_ghosts(cc, vv)
    return int(len(c)), got


exchange_boundary.last_bytes = 0


def exchange_boundary_layers(fuser, planes, rank, group=None, gather=None):
    """Round-1 interface (contiguous slabs, host arrays): export the lowest block layer of this rank's slab, all-gather, import the layer
    on this rank's upper plane.  Kept for callers that hold numpy arrays; `exchange_boundary` is the device-resident form."""
    lo, hi = planes[rank], planes[rank + 1]
    coords, vox = fuser.export_blocks_where(0, lo, lo + 1) if rank > 0 else (np.zeros((0, 3), np.int32), np.zeros((0, 512), fuser_voxel_dtype()))
    if gather is None:
        all_c = _all_gather_ragged(coords, group)
        all_v = _all_gather_ragged(np.ascontiguousarray(vox).view(np.uint8).reshape(len(coords), 4096), group)
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
    """parts: per rank (xyz [n,3] f32, rgba [n,4] u8, tris [m,3] u32, keys [n] u64[, face keys [m] u64]) of the rank's canonical mesh.
    Returns the canonical mesh of the whole scan: vertices unique by key in key order; triangles concatenated in rank order (the
    one-GPU order for contiguous slabs) or, when every part carries face keys (Mesh.face_keys(): needed for stripes), stably sorted by
    cube key -- every cube belongs to one rank, so that is the one-GPU order whatever the partition."""
    keys = np.concatenate([p[3] for p in parts]) if parts else np.zeros(0, np.uint64)
    xyz = np.concatenate([p[0] for p in parts]) if parts else np.zeros((0, 3), np.float32)
    rgba = np.concatenate([p[1] for p in parts]) if parts else np.zeros((0, 4), np.uint8)
    ukeys, first = np.unique(keys, return_index=True)
    tris = [np.searchsorted(ukeys, p[3][p[2].astype(np.int64)]).astype(np.uint32) for p in parts if len(p[2])]
    tris = np.concatenate(tris) if tris else np.zeros((0, 3), np.uint32)
    if parts and all(len(p) > 4 for p in parts):
        fk = np.concatenate([p[4] for p in parts if len(p[2])]) if len(tris) else np.zeros(0, np.uint64)
        tris = tris[np.argsort(fk, kind="stable")]
    return xyz[first], rgba[first], tris, ukeys
