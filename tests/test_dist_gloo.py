"""world_size = 2 on CPU (gloo): the N > 1 path of bench.py / shard.py.  The per-rank evaluation is done by the
oracle here (the HIP path needs a GPU); what is under test is the sharding, the count re-weighting and the
all-reduce plumbing: the reduced blocks must equal a single-process evaluation of the whole problem."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import binding as B
    import mba_vo_amd  # noqa: F401
    from mba_vo_amd import shard
    import scenes
    # (1) one joint problem, keypoints sharded over ranks
    sc = scenes.Scene(S=4, F=2, k=4, P=8, K=120, seed=5)
    lo, hi = shard.keypoint_range_of_rank(sc.K, rank, world)
    full_xy, full_z = sc.kp_xy, sc.kp_z
    sc.kp_xy, sc.kp_z, sc.K = np.ascontiguousarray(full_xy[lo:hi]), np.ascontiguousarray(full_z[lo:hi]), hi - lo
    p, keep = sc.oracle_problem(B)
    r = B.evaluate(p)
    local = torch.from_numpy(r["frame_blocks"].copy())
    joint = shard.combine_keypoint_shards(local, sc.K * sc.F * sc.P)
    # (2) independent pairs, round-robin, packed blocks summed with one all-reduce
    pairs = shard.pairs_of_rank(6, rank, world)
    acc = torch.zeros(6, r["frame_blocks"].shape[1], dtype=torch.float64)
    for b in pairs:
        s2 = scenes.Scene(S=2, F=1, k=4, P=8, K=40, seed=100 + b)
        p2, keep2 = s2.oracle_problem(B)
        acc[b] = torch.from_numpy(B.evaluate(p2)["frame_blocks"][0].copy())
    shard.allreduce_blocks(acc)
    if rank == 0:
        np.save(os.path.join(out_dir, "joint.npy"), joint.numpy())
        np.save(os.path.join(out_dir, "pairs.npy"), acc.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo(orc, mbavo, tmp_path):
    import torch.multiprocessing as mp
    import scenes
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    joint = np.load(tmp_path / "joint.npy")
    pairs = np.load(tmp_path / "pairs.npy")
    sc = scenes.Scene(S=4, F=2, k=4, P=8, K=120, seed=5)
    p, keep = sc.oracle_problem(orc)
    ref = orc.evaluate(p)["frame_blocks"]
    assert np.abs(joint - ref).max() <= 1e-12 * np.abs(ref).max()
    for b in range(6):
        s2 = scenes.Scene(S=2, F=1, k=4, P=8, K=40, seed=100 + b)
        p2, keep2 = s2.oracle_problem(orc)
        assert np.array_equal(pairs[b], orc.evaluate(p2)["frame_blocks"][0])


def test_shard_partitions_cover_everything(mbavo):
    from mba_vo_amd import shard
    for world in (1, 2, 4, 8):
        seen = sorted(sum((shard.pairs_of_rank(512, r, world) for r in range(world)), []))
        assert seen == list(range(512))
        ranges = [shard.keypoint_range_of_rank(307200, r, world) for r in range(world)]
        assert ranges[0][0] == 0 and ranges[-1][1] == 307200
        assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
