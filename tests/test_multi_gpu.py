"""SURVEY 8e on a node with at least two GPUs: one process per GPU over RCCL (torch.distributed backend "nccl").  Every test here skips on a one-GPU
box -- there the same control flow runs with both ranks on GPU 0 over gloo (tests/test_gpu_pipeline.py::test_bench_with_two_ranks_sharing_one_gpu,
::test_real_fusers_in_separate_processes_exchange_and_merge), which cannot exercise RCCL's device-to-device send / recv or a per-rank device."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from scannet_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gpus():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


needs_two = pytest.mark.skipif(_gpus() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")


def _bench(extra, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    import tempfile
    detail = os.path.join(tempfile.mkdtemp(prefix="sf_bench_"), "detail.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + extra, capture_output=True, text=True, cwd=ROOT, timeout=timeout,
                       env=dict(env, SF_BENCH_DETAIL=detail))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 4096, r.stdout[-1500:]          # rank 0 prints ONE compact line; the full measurement goes to the detail file
    assert json.loads(lines[0])["n_gpus"] == 2
    return json.load(open(detail))


@needs_two
def test_bench_partition_on_two_gpus_over_rccl():
    """configs[4] at N = 2: `python bench.py --gpus 2 --config partition` (bare: it re-executes itself under torch.distributed.run, one rank per GPU,
    nccl) -- stripes dealt to two fusers on two devices, the boundary layers ring-shifted device to device by RCCL send / recv, and the merged mesh of
    a prefix equal to ONE fuser's."""
    j = _bench(["--config", "partition", "--scan-frames", "256"])
    assert j["n_gpus"] == 2 and j["rccl_ranks"] == 2 and j["process_group"] == "nccl" and j["scaling"] == "strong"
    ex = j["exchange"]
    assert ex["mode"] == "neighbour" and ex["boundary_blocks_sent_total"] == ex["ghost_blocks_received_total"] > 0
    assert ex["payload_bytes_received_total"] == 4108 * ex["ghost_blocks_received_total"]
    pc = j["prefix_check"]
    assert pc["sha256_equal"] and pc["faces"] > 10000 and pc["boundary_blocks_sent"] == pc["ghost_blocks_received"] > 0
    j = _bench(["--config", "partition", "--scan-frames", "256", "--exchange", "all_gather"])
    assert j["rccl_ranks"] == 2 and j["exchange"]["mode"] == "all_gather" and j["prefix_check"]["sha256_equal"]


@needs_two
def test_bench_stream_and_scans_on_two_gpus():
    """configs[1] and configs[3] at N = 2: an independent scan per GPU, no data-path collective; RCCL carries the barriers and the max over ranks."""
    j = _bench(["--steps", "64", "--warmup", "5", "--repeats", "3", "--no-pmc"])
    assert j["n_gpus"] == 2 and j["rccl_ranks"] == 2 and j["process_group"] == "nccl" and j["scaling"] == "weak"
    assert len(j["per_rank_frames_per_s"]) == 2 and min(j["per_rank_frames_per_s"]) > 0 and j["value"] > max(j["per_rank_frames_per_s"])
    j = _bench(["--config", "scans", "--steps", "4", "--host-stage", "gpu"])
    assert j["n_gpus"] == 2 and j["rccl_ranks"] == 2 and j["value"] > 0 and j["unit"] == "scans/min"


def _room_frames(n, W, H, total, stride=9):
    return [(synth.render_room_depth(synth.trajectory_pose(i * stride, total), W, H, noise_frame=i), synth.trajectory_pose(i * stride, total)) for i in range(n)]


def _rccl_worker(rank, world, port, out_dir, mode, thickness):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from scannet_amd import fusion, partition
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        W, H = 320, 240
        fx, fy, mx, my = synth.intrinsics(W, H)
        gp = fusion.default_params(depth_width=W, depth_height=H, fx=fx, fy=fy, mx=mx, my=my, voxel_size=0.01, num_sdf_blocks=1 << 17)
        with fusion.Fuser(gp, device=rank) as f:
            f.set_stripes(0, -7, thickness, world, rank)
            for d, pose in _room_frames(24, W, H, 1200, stride=40):
                f.integrate(d, pose)
            oc, ov = f.export_blocks()
            sent, got = partition.exchange_boundary(f, mode=mode)          # device export -> RCCL -> device import
            m = f.extract_mesh()
            xyz, rgba, tris, keys = m.arrays(keys=True)
            np.savez(os.path.join(out_dir, "part%d.npz" % rank), xyz=xyz, rgba=rgba, tris=tris, keys=keys, fk=m.face_keys(), oc=oc, ov=ov.view(np.uint8),
                     sent=sent, got=got, bytes_in=partition.exchange_boundary.last_bytes)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@needs_two
@pytest.mark.parametrize("mode,thickness", [("neighbour", 4), ("neighbour", 16), ("all_gather", 16)])
def test_exchange_boundary_over_rccl_between_two_gpus(tmp_path, mode, thickness):
    """partition.exchange_boundary with a real process group of two ranks on two devices (nccl): everything stays in HBM -- sf_fuser_export_boundary
    into a device tensor, batch_isend_irecv (ncclSend / ncclRecv over the pair's xGMI link) or the ragged all-gather, sf_fuser_import_ghosts from the
    device tensor -- and the two ranks' meshes merge by key into the mesh ONE fuser extracts, byte for byte."""
    import socket
    import torch.multiprocessing as mp
    from scannet_amd import fusion, partition
    world = 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_rccl_worker, args=(world, port, str(tmp_path), mode, thickness), nprocs=world, join=True)
    W, H = 320, 240
    fx, fy, mx, my = synth.intrinsics(W, H)
    gp = fusion.default_params(depth_width=W, depth_height=H, fx=fx, fy=fy, mx=mx, my=my, voxel_size=0.01, num_sdf_blocks=1 << 17)
    with fusion.Fuser(gp) as whole:
        for d, pose in _room_frames(24, W, H, 1200, stride=40):
            whole.integrate(d, pose)
        ref = whole.extract_mesh().arrays(keys=True)
        wc, wv = whole.export_blocks()
    parts = [np.load(str(tmp_path / ("part%d.npz" % r))) for r in range(world)]
    allc = np.concatenate([p["oc"] for p in parts]); allv = np.concatenate([p["ov"] for p in parts])
    order = np.lexsort((allc[:, 2], allc[:, 1], allc[:, 0]))
    assert np.array_equal(allc[order], wc) and np.array_equal(allv[order].reshape(len(wc), -1), wv.view(np.uint8).reshape(len(wc), -1))
    assert all(int(p["sent"]) > 0 and int(p["got"]) > 0 for p in parts)
    if mode == "neighbour":
        assert all(int(p["bytes_in"]) == 4108 * int(p["got"]) for p in parts)
        assert sum(int(p["got"]) for p in parts) == sum(int(p["sent"]) for p in parts)
    xyz, rgba, tris, keys = partition.merge_slab_meshes([(p["xyz"], p["rgba"], p["tris"], p["keys"], p["fk"]) for p in parts])
    assert np.array_equal(keys, ref[3]) and np.array_equal(xyz.view(np.uint32), ref[0].view(np.uint32))
    assert np.array_equal(rgba, ref[1]) and np.array_equal(tris, ref[2])
