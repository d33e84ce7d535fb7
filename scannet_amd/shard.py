"""Scan-per-GPU sharding of independent scans (BASELINE configs[3]): one process per GPU, NO data-path
collective.  The reference serialises every GPU stage of every scan under one process-wide lock
(Server/process.py:22-23,75-76); here the scans of a rebuild are spread over the GPUs of a node.

Scheduling is a longest-first dynamic queue: the scans are ordered by decreasing cost (frame count) and every rank pops
the next index from ONE shared counter kept in the torch.distributed rendezvous store (`store.add` is an atomic
fetch-add on the TCPStore -- control plane only; no RCCL traffic, and it works unchanged with the gloo backend in the CPU
tests).  Without a process group the queue degenerates to "rank 0 takes everything".

    torchrun --nproc-per-node 8 -m scannet_amd.shard scans.txt        # one .sens path per line

runs the stage chain per scan: sf_fuse_run -> mesh on the rank's own GPU, then clean -> decimate x 2 -> segs.json on a pool of host
threads (SF_HOST_WORKERS, default: the usable CPUs divided among the local ranks) while the GPU takes the next scan -- the
sequential quadric collapse is ~20 s per scan-sized mesh, a hundred times the GPU part.
"""
import os
import sys
import time


def order_longest_first(costs):
    """Indices by decreasing cost, ties by index (deterministic on every rank)."""
    return sorted(range(len(costs)), key=lambda i: (-costs[i], i))


def static_lpt(costs, world):
    """Longest-processing-time-first static assignment (what the dynamic queue converges to when costs are exact)."""
    load = [0] * world
    out = [[] for _ in range(world)]
    for i in order_longest_first(costs):
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += costs[i]
    return out


class WorkQueue:
    """Atomic fetch-add over the rendezvous store; `key` namespaces one pass over the work list."""

    def __init__(self, n_items, key="scanfuse/queue"):
        self.n = n_items
        self.key = key
        self._local = 0
        self.store = None
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                self.store = dist.distributed_c10d._get_default_store()
        except Exception:
            self.store = None

    def pop(self):
        """Next position in the ordered work list, or None when the list is exhausted."""
        if self.store is not None:
            pos = self.store.add(self.key, 1) - 1
        else:
            pos = self._local
            self._local += 1
        return pos if pos < self.n else None


def run_sharded(items, costs, work, key="scanfuse/queue"):
    """Every rank calls this with the same `items` / `costs`; `work(item)` runs on the rank that popped it.  Returns the
    list of (item index, result) this rank produced."""
    order = order_longest_first(costs)
    q = WorkQueue(len(items), key)
    done = []
    while True:
        pos = q.pop()
        if pos is None:
            break
        i = order[pos]
        done.append((i, work(items[i])))
    return done


def fuse_scan(sens_path, device=0, params_file=None):
    """The GPU part of a scan: .sens -> fusion -> marching cubes.  Returns (mesh, info)."""
    from . import fusion, sens
    sd = sens.SensorData(sens_path)
    p = fusion.load_params(params_file) if params_file else fusion.default_params()
    p.depth_width, p.depth_height = sd.depth_width, sd.depth_height
    K = sd.intrinsic_depth
    p.fx, p.fy, p.mx, p.my, p.depth_shift = float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), sd.depth_shift
    t0 = time.perf_counter()
    with fusion.Fuser(p, device=device) as f:
        rs = f.run(sd)
        mesh = f.extract_mesh()
    return mesh, {"scan": sens_path, "frames": rs["frames_integrated"], "gpu_seconds": time.perf_counter() - t0}


def finish_scan(mesh, sens_path, clean_min_faces=7500, kthresh=0.01, seg_min_verts=20, gpu_decimate=None, gpu_clean=None):
    """The host part (Server/scan_processor.py:141-156): <id>_vh.ply, clean.mlx -> <id>_vh_clean.ply, simplify.mlx twice (each
    followed by cleanLoRes) -> <id>_vh_clean_2.ply, Segmentator -> <id>_vh_clean_2.0.010000.segs.json.  One thread, tens of seconds
    for a scan-sized mesh (the quadric collapse is sequential): run several of these side by side."""
    from . import meshclean, segmentator
    base = os.path.splitext(sens_path)[0]
    t0 = time.perf_counter()
    mesh.write_ply(base + "_vh.ply")
    cleaned, cst = meshclean.clean(mesh, meshclean.CLEAN_MLX_MERGE_DISTANCE, clean_min_faces, gpu=gpu_clean)   # gpu_clean = a device index: sf_mesh_clean_gpu, same output
    cleaned.write_ply(base + "_vh_clean.ply")
    cur = cleaned
    for _ in range(2):
        simp, _ = meshclean.simplify(cur, gpu=gpu_decimate)   # gpu_decimate = a device index: sf_mesh_simplify_gpu instead of the sequential filter
        cur, _ = meshclean.clean(simp, meshclean.CLEAN_MLX_MERGE_DISTANCE, meshclean.CLEAN_LORES_MIN_COMPONENT, gpu=gpu_clean)
    cur.write_ply(base + "_vh_clean_2.ply")
    nseg = segmentator.segment_to_json(base + "_vh_clean_2.ply", kthresh, seg_min_verts)
    return {"faces": cst["faces_out"], "faces_decimated": cur.counts()[1], "segments": nseg, "host_seconds": time.perf_counter() - t0}


def process_scan(sens_path, device=0, params_file=None, clean_min_faces=7500, kthresh=0.01, seg_min_verts=20):
    """Both parts of one scan, one after the other."""
    mesh, info = fuse_scan(sens_path, device, params_file)
    info.update(finish_scan(mesh, sens_path, clean_min_faces, kthresh, seg_min_verts))
    info["seconds"] = info["gpu_seconds"] + info["host_seconds"]
    return info


def run_pipelined(items, costs, gpu_stage, host_stage, host_workers, key="scanfuse/queue"):
    """run_sharded with the work split in two: `gpu_stage(item)` runs on this rank as items are popped, `host_stage(item, x)` (x = what
    the GPU stage returned) runs on a pool of `host_workers` threads, so the GPU takes the next scan while the meshes of the previous
    ones are cleaned, decimated and segmented (ctypes calls release the GIL).  At most 2 * host_workers results wait for a thread at
    any time (a scan-sized mesh is a few hundred MB).  Returns [(item index, host result)] in the order the items were popped."""
    from concurrent.futures import ThreadPoolExecutor
    order = order_longest_first(costs)
    q = WorkQueue(len(items), key)
    pending = []
    with ThreadPoolExecutor(max_workers=max(1, int(host_workers))) as pool:
        while True:
            pos = q.pop()
            if pos is None:
                break
            i = order[pos]
            x = gpu_stage(items[i])
            pending.append((i, pool.submit(host_stage, items[i], x)))
            while sum(1 for _, fut in pending if not fut.done()) >= 2 * max(1, int(host_workers)):
                time.sleep(0.01)
        return [(i, fut.result()) for i, fut in pending]


def main(argv=None):
    import torch
    import torch.distributed as dist
    argv = sys.argv[1:] if argv is None else argv
    # python -m scannet_amd.shard scans.txt [--host-decimate] [--host-clean]
    #   everything behind marching cubes runs on the rank's GPU by default: the cleaning filters (sf_mesh_clean_gpu: output identical to the host
    #   filters) and the quadric collapse (sf_mesh_simplify_gpu: rounds of independent collapses -- different triangles from the sequential filter,
    #   the same guarantees and tests; neither is pinned to MeshLab).  The sequential collapse is 19 of a scan's 22 s of host time and left the GPU
    #   idle 86 % of a rebuild (21.8 against 100 scans/min per GPU, profiles/r03_bench_scans_*.json): --host-decimate / --host-clean select the
    #   host filters.  (--gpu-decimate: the default since round 4, accepted for old command lines.)
    flags = {a for a in argv if a.startswith("--")}
    argv = [a for a in argv if not a.startswith("--")]
    unknown = flags - {"--gpu-decimate", "--host-clean", "--host-decimate"}
    if unknown or len(argv) != 1:
        raise SystemExit("usage: python -m scannet_amd.shard scans.txt [--host-decimate] [--host-clean]" + ("  (unknown: %s)" % " ".join(sorted(unknown)) if unknown else ""))
    scans = [ln.strip() for ln in open(argv[0]) if ln.strip()]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        dist.init_process_group(backend="nccl" if torch.cuda.is_available() else "gloo")
    costs = [os.path.getsize(s) for s in scans]  # compressed size tracks the frame count
    from . import _abi
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    workers = int(os.environ.get("SF_HOST_WORKERS", "0")) or max(1, _abi.usable_cpus() // max(1, local_world))

    def host(path, x):
        mesh, info = x
        info.update(finish_scan(mesh, path, gpu_clean=None if "--host-clean" in flags else local_rank,
                                gpu_decimate=None if "--host-decimate" in flags else local_rank))
        return info
    res = run_pipelined(scans, costs, lambda s: fuse_scan(s, device=local_rank), host, workers)
    for i, r in res:
        print("rank %d: %s" % (int(os.environ.get("RANK", "0")), r))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
