"""world_size = 2 on CPU (gloo): the N > 1 path of bench.py / shard.py without a GPU.  What runs here is the PRODUCT's
sharding (mbavo_shard_keypoints / mbavo_shard_frames through the C ABI: pure host pointer arithmetic, fed with host
addresses), the product's host merge (mbavo_merge_host) and the all-reduce plumbing; the per-rank evaluation itself is
done by the oracle (the HIP engine needs a GPU -- tests/test_gpu_dist.py is the same test on the engine + RCCL).
The reduced objects must equal a single-process evaluation of the whole problem to 1e-12."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _addr(a):
    return a.ctypes.data


def host_problem(M, sc, cur_ptrs):
    """capi.Problem of a Scene whose 'device' pointers are HOST addresses (only ever used for pointer arithmetic)."""
    p = M.capi.Problem()
    p.S, p.F, p.K, p.P, p.N, p.H, p.W = sc.S, sc.F, sc.K, sc.P, sc.N, sc.H, sc.W
    p.d_ref_img, p.d_ref_dIxy, p.d_cur_imgs = _addr(sc.ref), _addr(sc.grad), _addr(cur_ptrs)
    p.d_kp_xy, p.kp_stride, p.d_kp_z, p.d_pattern = _addr(sc.kp_xy), 2, _addr(sc.kp_z), _addr(sc.pattern)
    p.d_outlier, p.num_bad = (_addr(sc.outlier) if sc.outlier is not None else None), sc.num_bad
    p.d_cap_time, p.d_exp_time, p.t0, p.dt = _addr(sc.cap), _addr(sc.exp), sc.t0, sc.dt
    p.d_knots_t, p.d_knots_R = _addr(sc.knots_t), _addr(sc.knots_R)
    p.h_start_idx = sc.start_idx.ctypes.data_as(C.POINTER(C.c_int))
    p.huber_a = sc.huber
    return p


def oracle_shard_blocks(B, sc, shard, first, mode):
    """Frame blocks of the shard described by the product's `shard` struct, evaluated by the oracle on the matching
    slices of the scene and re-scaled to the WHOLE problem's residual count (what the engine does with num_residuals)."""
    import copy
    s = copy.copy(sc)
    if mode == "keypoints":
        lo, hi = first, first + shard.K
        s.kp_xy, s.kp_z, s.K = np.ascontiguousarray(sc.kp_xy[lo:hi]), np.ascontiguousarray(sc.kp_z[lo:hi]), hi - lo
        if sc.outlier is not None:
            s.outlier = np.ascontiguousarray(sc.outlier[lo:hi])
            s.num_bad = 0
    else:
        lo, hi = first, first + shard.F
        s.cur, s.cap, s.exp, s.F = sc.cur[lo:hi], np.ascontiguousarray(sc.cap[lo:hi]), np.ascontiguousarray(sc.exp[lo:hi]), hi - lo
        s.start_idx = np.ascontiguousarray(sc.start_idx[lo:hi])
    if s.K == 0 or s.F == 0:
        return np.zeros((s.F, sc.E))
    p, keep = s.oracle_problem(B)
    fb = B.evaluate(p)["frame_blocks"]
    local = (s.K - s.num_bad) * s.F * s.P
    return fb * (local / float(shard.num_residuals))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import binding as B
    import mba_vo_amd as M
    from mba_vo_amd import shard
    import scenes
    lib = M.load()
    # (1) one joint problem, keypoints sharded: the packed frame blocks of the ranks add up
    sc = scenes.Scene(S=4, F=2, k=4, P=8, K=121, seed=5, outlier_frac=0.1)
    curp = np.array([_addr(c) for c in sc.cur], np.int64)
    whole = (M.capi.Problem * 1)(host_problem(M, sc, curp))
    sh, first = shard.shard_array(lib, whole, rank, world, "keypoints")
    lo, hi = shard.keypoint_range_of_rank(sc.K, rank, world)
    assert (first[0], sh[0].K) == (lo, hi - lo) and sh[0].num_residuals == (sc.K - sc.num_bad) * sc.F * sc.P
    assert sh[0].d_kp_xy == whole[0].d_kp_xy + lo * 16 and sh[0].d_kp_z == whole[0].d_kp_z + lo * 8
    assert sh[0].d_outlier == whole[0].d_outlier + lo and sh[0].F == sc.F
    blocks = torch.from_numpy(oracle_shard_blocks(B, sc, sh[0], int(first[0]), "keypoints").copy())
    shard.allreduce_blocks(blocks)
    # (2) one joint problem, frames sharded: every rank merges its frames into the 6N x 6N system, the systems add up
    sc2 = scenes.Scene(S=2, F=3, k=4, P=8, K=40, seed=11)
    curp2 = np.array([_addr(c) for c in sc2.cur], np.int64)
    whole2 = (M.capi.Problem * 1)(host_problem(M, sc2, curp2))
    sh2, first2 = shard.shard_array(lib, whole2, rank, world, "frames")
    f0, f1 = shard.frame_range_of_rank(sc2.F, rank, world)
    assert (first2[0], sh2[0].F) == (f0, f1 - f0) and sh2[0].num_residuals == sc2.K * sc2.F * sc2.P and sh2[0].K == sc2.K
    assert sh2[0].d_cur_imgs == whole2[0].d_cur_imgs + 8 * f0 and sh2[0].d_cap_time == whole2[0].d_cap_time + 8 * f0
    fb2 = np.ascontiguousarray(oracle_shard_blocks(B, sc2, sh2[0], f0, "frames"))
    n = 6 * sc2.N
    cost, H, g = np.zeros(1), np.zeros(n * n), np.zeros(n)
    assert lib.mbavo_merge_host(sh2[0].F, sc2.k, M.capi.dp(fb2), sh2[0].h_start_idx, sc2.N, M.capi.dp(cost), M.capi.dp(H),
                                M.capi.dp(g)) == 0
    system = torch.from_numpy(np.concatenate([cost, g, H]))
    shard.allreduce_blocks(system)
    # (3) independent pairs, pair b -> rank b % world, in the product's rank-major layout (shard.pair_layout):
    # every rank fills ITS contiguous slice of a zero send buffer, one all-reduce leaves every pair's blocks everywhere.
    # Pairs have different frame counts here (1 or 2), so the slices have different lengths.
    frames = [1 + (b % 3 == 1) for b in range(7)]
    row_base, row_of_pair = shard.pair_layout(frames, world)
    send = torch.zeros(row_base[-1], sc.E, dtype=torch.float64)
    row = row_base[rank]
    for b in shard.pairs_of_rank(7, rank, world):
        s3 = scenes.Scene(S=2, F=frames[b], k=4, P=8, K=40, seed=100 + b)
        p3, keep3 = s3.oracle_problem(B)
        assert row == row_of_pair[b]
        send[row:row + frames[b]] = torch.from_numpy(B.evaluate(p3)["frame_blocks"].copy())
        row += frames[b]
    assert row == row_base[rank + 1]
    acc = send.clone()  # (gloo's all-reduce is in place; the product's RCCL call is out of place, send stays untouched)
    shard.allreduce_blocks(acc)
    assert torch.equal(acc[row_base[rank]:row_base[rank + 1]], send[row_base[rank]:row_base[rank + 1]])  # x + 0 is exact
    # (3b) the same pairs through the DEFAULT collective of pair mode (round 4): equal slices (pair_layout(..., equal_slices=True):
    # the fullest rank's rows everywhere, unused tail rows zero), ONE in-place all-gather -- no sum, half the bytes on the wire
    eq_base, eq_row = shard.pair_layout(frames, world, equal_slices=True)
    width = eq_base[1] - eq_base[0]
    mine_eq = torch.zeros(width, sc.E, dtype=torch.float64)
    for b in shard.pairs_of_rank(7, rank, world):
        mine_eq[eq_row[b] - eq_base[rank]:eq_row[b] - eq_base[rank] + frames[b]] = send[row_of_pair[b]:row_of_pair[b] + frames[b]]
    gathered_eq = torch.zeros(world * width, sc.E, dtype=torch.float64)
    dist.all_gather_into_tensor(gathered_eq, mine_eq)
    for b in range(7):
        assert torch.equal(gathered_eq[eq_row[b]:eq_row[b] + frames[b]], acc[row_of_pair[b]:row_of_pair[b] + frames[b]])
    # ---- pair-sharded batched LM: the record all-gather (shard.lm_record_layout; the LM itself needs the GPU) -- every
    # rank fills its pairs' records, equal slices are gathered, and pair b's record sits at row_of_pair[b] on every rank
    rows, rec, lm_row = shard.lm_record_layout(7, world, 4)
    mine = torch.zeros(rows, rec, dtype=torch.float64)
    for j, b in enumerate(shard.pairs_of_rank(7, rank, world)):
        assert lm_row[b] == rank * rows + j
        mine[j] = torch.arange(rec, dtype=torch.float64) + 1000.0 * (b + 1)
    gathered = torch.zeros(world * rows, rec, dtype=torch.float64)
    dist.all_gather_into_tensor(gathered, mine)
    for b in range(7):
        assert torch.equal(gathered[lm_row[b]], torch.arange(rec, dtype=torch.float64) + 1000.0 * (b + 1))
    if rank == 0:
        np.save(os.path.join(out_dir, "joint.npy"), blocks.numpy())
        np.save(os.path.join(out_dir, "system.npy"), system.numpy())
        np.save(os.path.join(out_dir, "pairs.npy"), acc.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo(orc, mbavo, tmp_path):
    import torch.multiprocessing as mp
    import scenes
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    joint, system, pairs = np.load(tmp_path / "joint.npy"), np.load(tmp_path / "system.npy"), np.load(tmp_path / "pairs.npy")
    sc = scenes.Scene(S=4, F=2, k=4, P=8, K=121, seed=5, outlier_frac=0.1)
    p, keep = sc.oracle_problem(orc)
    ref = orc.evaluate(p)["frame_blocks"]
    assert np.abs(joint - ref).max() <= 1e-12 * np.abs(ref).max()
    sc2 = scenes.Scene(S=2, F=3, k=4, P=8, K=40, seed=11)
    p2, keep2 = sc2.oracle_problem(orc)
    r2 = orc.evaluate(p2)
    n = 6 * sc2.N
    ref_sys = np.concatenate([[r2["cost"]], r2["g"], r2["H"].T.ravel()])
    assert system.shape == ref_sys.shape == (mbavo.load().mbavo_system_len(sc2.N),)
    assert np.abs(system - ref_sys).max() <= 1e-12 * np.abs(ref_sys).max()
    from mba_vo_amd import shard
    frames = [1 + (b % 3 == 1) for b in range(7)]
    row_base, row_of_pair = shard.pair_layout(frames, 2)
    assert pairs.shape[0] == sum(frames) == row_base[-1]
    for b in range(7):
        s3 = scenes.Scene(S=2, F=frames[b], k=4, P=8, K=40, seed=100 + b)
        p3, keep3 = s3.oracle_problem(orc)
        assert np.array_equal(pairs[row_of_pair[b]:row_of_pair[b] + frames[b]], orc.evaluate(p3)["frame_blocks"])  # bit-exact


def test_pair_layout_is_a_rank_major_partition(mbavo):
    from mba_vo_amd import shard
    rng = np.random.default_rng(0)
    for world in (1, 2, 3, 4, 8):
        for B in (1, 5, 64, 512):
            frames = [int(f) for f in rng.integers(1, 4, B)]
            row_base, row_of_pair = shard.pair_layout(frames, world)
            assert row_base[0] == 0 and row_base[-1] == sum(frames) and len(row_base) == world + 1
            rows = np.full(sum(frames), -1)
            for b in range(B):
                r = b % world
                assert row_base[r] <= row_of_pair[b] and row_of_pair[b] + frames[b] <= row_base[r + 1]  # inside its rank's slice
                assert (rows[row_of_pair[b]:row_of_pair[b] + frames[b]] == -1).all()                   # disjoint
                rows[row_of_pair[b]:row_of_pair[b] + frames[b]] = b
            assert (rows >= 0).all()
            for r in range(world):  # ascending pair order inside a slice
                mine = shard.pairs_of_rank(B, r, world)
                assert [row_of_pair[b] for b in mine] == sorted(row_of_pair[b] for b in mine)
            # equal slices (the all-gather's layout): the same order inside a slice, every slice as wide as the fullest rank's
            eq_base, eq_row = shard.pair_layout(frames, world, equal_slices=True)
            width = max(sum(frames[b] for b in shard.pairs_of_rank(B, r, world)) for r in range(world))
            assert eq_base == [r * width for r in range(world + 1)]
            for b in range(B):
                assert eq_row[b] - eq_base[b % world] == row_of_pair[b] - row_base[b % world]


def test_shard_partitions_cover_everything(mbavo):
    from mba_vo_amd import shard
    lib = mbavo.load()
    for world in (1, 2, 3, 4, 8):
        seen = sorted(sum((shard.pairs_of_rank(512, r, world) for r in range(world)), []))
        assert seen == list(range(512))
        ranges = [shard.keypoint_range_of_rank(307200, r, world) for r in range(world)]
        assert ranges[0][0] == 0 and ranges[-1][1] == 307200
        assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
        # the C ABI shards agree with the index helpers, cover the problem, and carry the whole residual count
        whole = mbavo.capi.Problem()
        whole.S, whole.F, whole.K, whole.P, whole.N, whole.kp_stride, whole.num_bad = 8, 5, 1001, 8, 6, 3, 7
        whole.d_kp_xy, whole.d_kp_z, whole.d_outlier, whole.d_cur_imgs = 1 << 20, 2 << 20, 3 << 20, 4 << 20
        whole.d_cap_time, whole.d_exp_time = 5 << 20, 6 << 20
        start = (C.c_int * 5)(0, 0, 1, 1, 2)
        whole.h_start_idx = C.cast(start, C.POINTER(C.c_int))
        k_next = f_next = 0
        for r in range(world):
            s, first = mbavo.capi.Problem(), C.c_int(-1)
            assert lib.mbavo_shard_keypoints(C.byref(whole), r, world, C.byref(s), C.byref(first)) == 0
            assert first.value == k_next == shard.keypoint_range_of_rank(1001, r, world)[0]
            assert s.d_kp_xy == whole.d_kp_xy + 24 * first.value and s.d_kp_z == whole.d_kp_z + 8 * first.value
            assert s.d_outlier == whole.d_outlier + first.value and s.F == 5
            assert s.num_residuals == (1001 - 7) * 5 * 8
            k_next += s.K
            assert lib.mbavo_shard_frames(C.byref(whole), r, world, C.byref(s), C.byref(first)) == 0
            assert first.value == f_next == shard.frame_range_of_rank(5, r, world)[0]
            assert s.d_cur_imgs == whole.d_cur_imgs + 8 * first.value and s.d_exp_time == whole.d_exp_time + 8 * first.value
            assert s.K == 1001 and s.num_residuals == (1001 - 7) * 5 * 8
            if s.F > 0:
                assert s.h_start_idx[0] == start[first.value]
            f_next += s.F
        assert k_next == 1001 and f_next == 5
    bad = mbavo.capi.Problem()
    assert lib.mbavo_shard_frames(C.byref(whole), 2, 2, C.byref(bad), None) == -1  # rank out of range
    assert lib.mbavo_system_len(4) == 601 and lib.mbavo_system_len(6) == 1333


def test_lm_record_layout_is_an_equal_slice_partition(mbavo):
    from mba_vo_amd import shard
    for world in (1, 2, 3, 4, 8):
        for B in (1, 7, 64, 512):
            rows, rec, row_of = shard.lm_record_layout(B, world, 16)
            assert rec == 7 * 16 + shard.LM_RECORD_SCALARS and rows * world >= B and (rows - 1) * world < B
            assert len(set(row_of)) == B and max(row_of) < rows * world
            for r in range(world):
                mine = shard.pairs_of_rank(B, r, world)
                assert [row_of[b] for b in mine] == list(range(r * rows, r * rows + len(mine)))  # contiguous from the slice's start
