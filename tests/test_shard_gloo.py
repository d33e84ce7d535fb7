"""Multi-process paths on CPU (gloo, world size 2): the scan-per-GPU work queue of scannet_amd/shard.py and the host
logic of the slab partition (scannet_amd/partition.py).  No GPU: the per-scan / per-slab work is injected."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker_queue(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from scannet_amd import shard
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    costs = [5, 90, 17, 17, 300, 1, 42, 8, 8, 64, 3]
    items = ["scan%02d" % i for i in range(len(costs))]
    seen = []

    def work(item):
        seen.append(item)
        return costs[items.index(item)]

    done = shard.run_sharded(items, costs, work)
    # a second pass over the same list needs its own counter key
    done2 = shard.run_sharded(items, costs, work, key="scanfuse/queue/pass2")
    np.save(os.path.join(out_dir, "r%d.npy" % rank), np.array([i for i, _ in done] + [-1] + [i for i, _ in done2]))
    dist.barrier()
    dist.destroy_process_group()


def test_work_queue_two_ranks(tmp_path):
    port = _free_port()
    mp.spawn(_worker_queue, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    per_rank = [np.load(str(tmp_path / ("r%d.npy" % r))).tolist() for r in range(2)]
    for which in (0, 1):
        got = []
        for lst in per_rank:
            cut = lst.index(-1)
            got.append(lst[:cut] if which == 0 else lst[cut + 1:])
        allidx = sorted(got[0] + got[1])
        assert allidx == list(range(11)), "every scan exactly once"
        # within a rank the popped positions follow the longest-first order
        from scannet_amd import shard
        order = shard.order_longest_first([5, 90, 17, 17, 300, 1, 42, 8, 8, 64, 3])
        for lst in got:
            pos = [order.index(i) for i in lst]
            assert pos == sorted(pos)


def test_queue_without_process_group_and_static_lpt():
    from scannet_amd import shard
    costs = [5, 90, 17, 17, 300, 1, 42]
    assert shard.order_longest_first(costs) == [4, 1, 6, 2, 3, 0, 5]
    done = shard.run_sharded(list("abcdefg"), costs, lambda s: s.upper())
    assert [i for i, _ in done] == [4, 1, 6, 2, 3, 0, 5] and done[0][1] == "E"
    parts = shard.static_lpt(costs, 3)
    assert sorted(sum(parts, [])) == list(range(7))
    loads = [sum(costs[i] for i in p) for p in parts]
    assert max(loads) == 300 and parts[0] == [4]


def _worker_pipelined(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import threading
    import time
    import torch.distributed as dist
    from scannet_amd import shard
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    costs = [5, 90, 17, 17, 300, 1, 42, 8, 8, 64, 3]
    items = ["scan%02d" % i for i in range(len(costs))]
    gpu_thread = threading.get_ident()
    live, peak = [0], [0]
    lock = threading.Lock()

    def gpu(item):                      # always on the calling thread, one at a time, in queue order
        assert threading.get_ident() == gpu_thread
        return item + "/mesh"

    def host(item, x):                  # on the pool: several at once
        assert x == item + "/mesh" and threading.get_ident() != gpu_thread
        with lock:
            live[0] += 1
            peak[0] = max(peak[0], live[0])
        time.sleep(0.03)
        with lock:
            live[0] -= 1
        return costs[items.index(item)]

    done = shard.run_pipelined(items, costs, gpu, host, 3, key="scanfuse/queue/pipelined")
    assert all(r == costs[i] for i, r in done) and peak[0] <= 3
    np.save(os.path.join(out_dir, "p%d.npy" % rank), np.array([i for i, _ in done] + [-1, peak[0]]))
    dist.barrier()
    dist.destroy_process_group()


def test_pipelined_queue_two_ranks(tmp_path):
    """shard.run_pipelined: the GPU stage of the next scan runs while a pool of host threads finishes the previous ones; every scan
    exactly once over the two ranks, popped longest first."""
    port = _free_port()
    mp.spawn(_worker_pipelined, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    from scannet_amd import shard
    order = shard.order_longest_first([5, 90, 17, 17, 300, 1, 42, 8, 8, 64, 3])
    got, peaks = [], []
    for r in range(2):
        lst = np.load(str(tmp_path / ("p%d.npy" % r))).tolist()
        cut = lst.index(-1)
        got.append(lst[:cut])
        peaks.append(lst[cut + 1])
        pos = [order.index(i) for i in lst[:cut]]
        assert pos == sorted(pos)
    assert sorted(got[0] + got[1]) == list(range(11))
    assert max(peaks) >= 2, "host stages overlapped"


def _worker_gather(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from scannet_amd import partition
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    rng = np.random.default_rng(100 + rank)
    n = [0, 37][rank]                      # rank 0 (lowest slab) sends nothing
    coords = rng.integers(-50, 50, (n, 3)).astype(np.int32)
    vox = rng.integers(0, 256, (n, 4096), dtype=np.uint8)
    all_c = partition._all_gather_ragged(coords)
    all_v = partition._all_gather_ragged(vox)
    assert [len(c) for c in all_c] == [0, 37] and all_v[1].shape == (37, 4096)
    np.save(os.path.join(out_dir, "g%d.npy" % rank), np.concatenate([all_c[1].ravel(), all_v[1].ravel().astype(np.int32)]))

    class FakeFuser:                        # the host logic of the exchange without a GPU
        def __init__(self):
            self.imported = None

        def export_blocks_where(self, axis, lo, hi):
            assert axis == 0 and hi == lo + 1
            c = coords.copy()
            c[:, 0] = lo
            return c, vox

        def import_blocks(self, c, v, ghost=True):
            self.imported = (c.copy(), v.copy(), ghost)

    f = FakeFuser()
    planes = [-1000, 7, 1000]
    sent, got = partition.exchange_boundary_layers(f, planes, rank)
    if rank == 0:
        assert sent == 0 and got == 37 and f.imported[2] is True and (f.imported[0][:, 0] == 7).all()
    else:
        assert sent == 37 and got == 0 and f.imported is None
    dist.barrier()
    dist.destroy_process_group()


def test_boundary_all_gather_two_ranks(tmp_path):
    port = _free_port()
    mp.spawn(_worker_gather, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(str(tmp_path / "g0.npy")), np.load(str(tmp_path / "g1.npy"))
    assert np.array_equal(a, b) and len(a) == 37 * 3 + 37 * 4096


def _worker_exchange(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from scannet_amd import partition
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    origin, thick = -3, 4

    class FakeFuser:                        # the exchange logic of partition.exchange_boundary without a GPU
        """Owns the stripes of `rank`; holds one block per x layer in [-20, 20) it owns, each voxel byte = the layer's x + 64."""
        def __init__(self):
            self.ghosts = {}

        def _layers(self):
            return [x for x in range(-20, 20) if partition.owner_of(x, origin, thick, world) == rank]

        def export_boundary(self):
            xs = [x for x in self._layers() if partition.owner_of(x - 1, origin, thick, world) != rank]
            c = np.array([[x, 1, 2] for x in xs], np.int32).reshape(-1, 3)
            v = np.stack([np.full(4096, x + 64, np.uint8) for x in xs]) if xs else np.zeros((0, 4096), np.uint8)
            return c, v

        def count_boundary(self):
            return len(self.export_boundary()[0])

        def import_ghosts(self, c, v):
            n = 0
            for (x, y, z), tile in zip(np.asarray(c), np.asarray(v)):
                if partition.owner_of(x, origin, thick, world) != rank and partition.owner_of(x - 1, origin, thick, world) == rank:
                    assert (tile == x + 64).all()
                    self.ghosts[int(x)] = True
                    n += 1
            return n

    for mode in ("neighbour", "all_gather"):   # the ring shift to the one rank that needs the layers, and the all-gather of round 2
        f = FakeFuser()
        sent, got = partition.exchange_boundary(f, mode=mode)
        mine = f._layers()
        want = sorted(x + 1 for x in mine if x + 1 < 20 and partition.owner_of(x + 1, origin, thick, world) != rank)
        assert sent == f.count_boundary() and sorted(f.ghosts) == want and got == len(want), (mode, rank, sorted(f.ghosts), want)
        if mode == "neighbour":   # exactly what this rank keeps arrived, nothing else
            assert partition.exchange_boundary.last_bytes == 4108 * len(want)
    np.save(os.path.join(out_dir, "x%d.npy" % rank), np.array([sent, got]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_stripe_boundary_exchange_two_ranks(tmp_path, world):
    """partition.exchange_boundary at world size 2 and 3: the ring shift (every boundary layer goes to the left neighbour and nowhere else) and
    the all-gather + ownership filter give every rank exactly the layers above its own."""
    port = _free_port()
    mp.spawn(_worker_exchange, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    a, b = np.load(str(tmp_path / "x0.npy")), np.load(str(tmp_path / "x1.npy"))
    assert a[0] > 0 and b[0] > 0 and a[1] > 0 and b[1] > 0


def test_stripe_ownership_and_balanced_planes():
    from scannet_amd import partition
    # floor division for negative layers; every layer has exactly one owner; stripes are `thickness` layers thick
    owners = [partition.owner_of(x, -3, 4, 3) for x in range(-15, 15)]
    assert owners[12:16] == [0, 0, 0, 0] and owners[16:20] == [1, 1, 1, 1] and owners[8:12] == [2, 2, 2, 2] and owners[0:4] == [0, 0, 0, 0]
    # slab planes from the block histogram of a prefix: equal counts per slab whatever the geometry (a long corridor with one busy room)
    x = np.concatenate([np.arange(0, 1000), np.full(3000, 1500), np.arange(2000, 2200)])
    planes = partition.planes_from_histogram(x, 4)
    counts = [int(((x >= planes[r]) & (x < planes[r + 1])).sum()) for r in range(4)]
    assert sum(counts) == len(x) and planes[0] < -(1 << 19) and planes[-1] > (1 << 19) and all(planes[r] < planes[r + 1] for r in range(4))
    assert max(counts) <= 3000 + 1050   # the 3000 blocks of one layer cannot be split; the rest is even


def test_slab_planes_and_mesh_merge():
    from scannet_amd import partition
    p = partition.slab_planes(-40, 40, 4)
    assert p[1:-1] == [-20, 0, 20] and p[0] < -(1 << 19) and p[-1] > (1 << 19)
    poses = np.tile(np.eye(4, dtype=np.float32), (5, 1, 1))
    poses[:, 0, 3] = [0.0, 1.0, 2.0, 3.0, -np.inf]
    q = partition.planes_from_poses(poses, 0.004, 4.0, 2)
    assert q[1] == int(round((np.floor(-4.0 / 0.032) + np.ceil(7.0 / 0.032)) / 2))
    # two slab meshes sharing two boundary vertices (keys 20, 30)
    a = (np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0]], np.float32), np.full((3, 4), 1, np.uint8), np.array([[0, 1, 2]], np.uint32), np.array([10, 20, 30], np.uint64))
    b = (np.array([[1, 0, 0], [1, 1, 0], [2, 0, 0]], np.float32), np.full((3, 4), 2, np.uint8), np.array([[0, 2, 1]], np.uint32), np.array([20, 30, 40], np.uint64))
    xyz, rgba, tris, keys = partition.merge_slab_meshes([a, b])
    assert keys.tolist() == [10, 20, 30, 40] and tris.tolist() == [[0, 1, 2], [1, 3, 2]]
    assert xyz.tolist() == [[0, 0, 0], [1, 0, 0], [1, 1, 0], [2, 0, 0]] and rgba[:, 0].tolist() == [1, 1, 1, 2]


def test_slab_planes_never_leave_an_empty_slab():
    """an extent narrower than `world` layers gave equal planes (an empty slab); the ring shift of exchange_boundary sends a rank's
    lowest layer to rank - 1 only, so the rank below an empty slab never received the layer it needed.  Planes are strictly increasing now, and
    with them the neighbour rule delivers every layer that is wanted (simulated here layer by layer for world 3 over a 2-layer extent)."""
    from scannet_amd import partition
    for lo, hi, world in ((0, 2, 3), (5, 5, 4), (-3, -1, 8), (0, 100, 3)):
        p = partition.slab_planes(lo, hi, world)
        assert len(p) == world + 1 and all(p[r] < p[r + 1] for r in range(world)), p
        assert p[0] < -(1 << 19) and p[-1] >= (1 << 20) - 1
        owner = lambda c: next(r for r in range(world) if p[r] <= c < p[r + 1])   # noqa: E731
        layers = range(lo - 2, hi + 3)   # layers that hold blocks
        for c in layers:
            # layer c is a boundary layer of its owner iff the layer below belongs to somebody else; that somebody must be owner - 1,
            # the only rank the ring shift sends to
            if owner(c) != owner(c - 1):
                assert owner(c - 1) == owner(c) - 1, (p, c)
