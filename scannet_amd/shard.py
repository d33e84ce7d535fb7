"""Scan-per-GPU sharding of independent scans (BASELINE configs[3]): one process per GPU, NO data-path
collective.  The reference serialises every GPU stage of every scan under one process-wide lock
(Server/process.py:22-23,75-76); here the scans of a rebuild are spread over the GPUs of a node.

Scheduling is a longest-first dynamic queue: the scans are ordered by decreasing cost (frame count) and every rank pops
the next index from ONE shared counter kept in the torch.distributed rendezvous store (`store.add` is an atomic
fetch-add on the TCPStore -- control plane only; no RCCL traffic, and it works unchanged with the gloo backend in the CPU
tests).  Without a process group the queue degenerates to "rank 0 takes everything".

    torchrun --nproc-per-node 8 -m scannet_amd.shard scans.txt        # one .sens path per line

runs the whole stage chain per scan (sf_fuse_run -> mesh -> clean -> segs.json) on the rank's own GPU.
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


def process_scan(sens_path, device=0, params_file=None, clean_min_faces=7500, kthresh=0.01, seg_min_verts=20):
    """The improve + segment stage chain for one scan on one GPU: <id>_vh.ply, <id>_vh_clean.ply, segs.json."""
    from . import fusion, meshclean, segmentator, sens
    sd = sens.SensorData(sens_path)
    p = fusion.load_params(params_file) if params_file else fusion.default_params()
    p.depth_width, p.depth_height = sd.depth_width, sd.depth_height
    K = sd.intrinsic_depth
    p.fx, p.fy, p.mx, p.my, p.depth_shift = float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), sd.depth_shift
    base = os.path.splitext(sens_path)[0]
    t0 = time.perf_counter()
    with fusion.Fuser(p, device=device) as f:
        rs = f.run(sd)
        mesh = f.extract_mesh()
    mesh.write_ply(base + "_vh.ply")
    cleaned, cst = meshclean.clean(mesh, meshclean.CLEAN_MLX_MERGE_DISTANCE, clean_min_faces)
    cleaned.write_ply(base + "_vh_clean.ply")
    nseg = segmentator.segment_to_json(base + "_vh_clean.ply", kthresh, seg_min_verts)
    return {"scan": sens_path, "frames": rs["frames_integrated"], "seconds": time.perf_counter() - t0, "faces": cst["faces_out"], "segments": nseg}


def main(argv=None):
    import torch
    import torch.distributed as dist
    argv = sys.argv[1:] if argv is None else argv
    scans = [ln.strip() for ln in open(argv[0]) if ln.strip()]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        dist.init_process_group(backend="nccl" if torch.cuda.is_available() else "gloo")
    costs = [os.path.getsize(s) for s in scans]  # compressed size tracks the frame count
    res = run_sharded(scans, costs, lambda s: process_scan(s, device=local_rank))
    for i, r in res:
        print("rank %d: %s" % (int(os.environ.get("RANK", "0")), r))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
