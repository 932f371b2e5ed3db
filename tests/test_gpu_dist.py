"""-m gpu: the multi-GPU path on the HIP engine + RCCL: shards of ONE joint problem (mbavo_shard_frames /
mbavo_shard_keypoints) evaluated by the fused engine on every rank, merged on the device (mbavo_merge_device) and summed
with mbavo_allreduce_blocks on the context's own RCCL communicator (mbavo_comm_init), against (a) a single-GPU
evaluation of the whole problem (1e-12: only the summation order differs), (b) the product's host merge (bit-exact) and
(c) the oracle (1e-9).  The reference's reduction point: merge_hessian_gradient_cost.cpp:39-86.

world = 1 runs on the one-GPU box (every code path, a 1-rank communicator); world = 2 needs two GPUs and is skipped
otherwise -- and runs as TWO PROCESSES SHARING THE ONE GPU (round 4): every line of ranks > 0 executes -- shards, slices,
layouts, device merge, the batched LM per rank -- with shard.HostStagedCollective (gloo on a pinned host copy) standing
in for the RCCL collectives, which cannot form a communicator over duplicate devices; only ncclAllReduce / ncclAllGather
themselves stay unexercised at N > 1.  The same body runs under torchrun:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 tests/test_gpu_dist.py
"""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def rank_body(rank, world, local_rank, port, out_path, comm="rccl"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(port))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    if comm == "rccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
    else:  # several ranks on one GPU: gloo is the process group AND stands in for the product's collectives
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import mba_vo_amd as M
    from mba_vo_amd import shard, workloads as wl
    from oracle import binding as B
    B.build()
    ctx = M.capi.Context(local_rank, stream=torch.cuda.current_stream().cuda_stream)
    assert ctx.lib.mbavo_comm_ranks(ctx.handle) == 0
    assert ctx.lib.mbavo_allreduce_blocks(ctx.handle, None, 8, 1) == -1  # no communicator yet: MBAVO_E_ARG
    if comm == "rccl":
        assert shard.comm_init(ctx, rank, world, shard.torch_bcast(dev)) == world
        coll = shard.RcclCollective(ctx)
    elif comm == "p2p":  # the product's one-shot collectives over peer-mapped regions (no RCCL): ranks sharing the GPU can use them
        assert ctx.lib.mbavo_allreduce_blocks_p2p(ctx.handle, 8, 1) == -1  # not created yet: MBAVO_E_ARG
        coll = shard.P2PCollective(ctx, rank, world, max_doubles=4096)      # (small: the regions must GROW during the test)
    else:
        coll = shard.HostStagedCollective(ctx, rank, world)
    res = {"world": world, "rccl_ranks": ctx.lib.mbavo_comm_ranks(ctx.handle) if comm == "rccl" else world, "collective": coll.name}

    def rel(a, b):
        return float((a - b).abs().max() / b.abs().max())

    # ---- frames mode: max(world, 2) + 1 blurred frames on one spline, 2 pyramid levels, dense
    F = max(world, 2) + 1
    probs = wl.pyramid_pair(120, 160, 2, S=8, k=4, N=4, mode="dense", seed=7, frames=F)
    dw = wl.DeviceWorkload(probs, device=dev)
    se = shard.ShardedEvaluation(ctx, dw.array, 4, rank, world, "frames", dev, collective=coll)
    for _ in range(3):  # repeated steps reuse the cached merge descriptors
        se.step(True)
    torch.cuda.synchronize()
    got, ref = se.reduced.clone(), se.reference()
    res["frames_vs_single_gpu"] = rel(got, ref)
    # (b) the device merge of the whole problem against the product's host merge of the same blocks: identical bits
    fb = se._ref_fb.cpu().numpy().reshape(-1, se.E)
    off, row, exact = 0, 0, True
    sys_ref = ref.cpu().numpy()
    worst_oracle = 0.0
    for p in probs:
        n = 6 * p.N
        cost, H, g = np.zeros(1), np.zeros(n * n), np.zeros(n)
        blk = np.ascontiguousarray(fb[row:row + p.F])
        assert ctx.lib.mbavo_merge_host(p.F, p.k, M.capi.dp(blk), M.capi.ip(p.start_idx), p.N, M.capi.dp(cost), M.capi.dp(H),
                                        M.capi.dp(g)) == 0
        host = np.concatenate([cost, g, H])
        exact = exact and np.array_equal(host, sys_ref[off:off + host.size])
        # (c) the oracle on the whole F-frame problem
        op, keep = B.make_problem(p.S, p.F, p.K, p.P, p.k, p.N, p.H, p.W, p.ref, p.grad, p.cur, p.kp_xy, p.kp_z, p.pattern,
                                  p.intr, p.cap, p.exp, p.t0, p.dt, p.knots_t, p.knots_R, p.start_idx, p.huber)
        ro = B.evaluate(op)
        orc_sys = np.concatenate([[ro["cost"]], ro["g"], ro["H"].T.ravel()])
        mine = got.cpu().numpy()[off:off + host.size]
        worst_oracle = max(worst_oracle, float(np.abs(mine - orc_sys).max() / np.abs(orc_sys).max()))
        off += host.size
        row += p.F
    res["device_merge_equals_host_merge"] = bool(exact)
    res["frames_vs_oracle"] = worst_oracle

    # ---- frame_blocks mode (bench.py's default for a single pair): the same frame shards, but the packed blocks themselves are
    # summed from every rank's slice of a zero send buffer -- no merge kernel; rows in rank-major order
    se_fb = shard.ShardedEvaluation(ctx, dw.array, 4, rank, world, "frame_blocks", dev, collective=coll)
    for _ in range(2):
        se_fb.step(True)
    torch.cuda.synchronize()
    got_fb, ref_fb = se_fb.reduced.clone(), se_fb.reference()
    res["frame_blocks_vs_single_gpu"] = rel(got_fb, ref_fb)
    lo_fb, hi_fb = se_fb.row_base[rank] * se_fb.E, se_fb.row_base[rank + 1] * se_fb.E
    res["frame_blocks_foreign_slices_zero"] = bool(float(se_fb.send[:lo_fb].abs().sum() + se_fb.send[hi_fb:].abs().sum()) == 0.0)
    # the consumer's merge of the reduced blocks == the 'frames' mode's reduced systems
    blocks_pm = torch.empty_like(got_fb).view(se_fb.nbf_whole, se_fb.E)
    blocks_pm[se_fb._perm] = got_fb.view(se_fb.nbf_whole, se_fb.E)  # back to problem-major rows
    sys_fb = torch.zeros_like(got)
    assert ctx.lib.mbavo_merge_device(ctx.handle, len(probs), dw.array, 4, blocks_pm.data_ptr(), sys_fb.data_ptr()) == 0
    torch.cuda.synchronize()
    res["frame_blocks_merged_vs_frames_mode"] = rel(sys_fb, got)

    # ---- keypoints mode: 5 semi-dense pairs, every pair's keypoints sharded, packed blocks summed
    pb = wl.pair_batch(5, H=240, W=320, S=8, k=4, N=4, mode="semidense", seed=3)
    dw2 = wl.DeviceWorkload(pb, device=dev)
    se2 = shard.ShardedEvaluation(ctx, dw2.array, 4, rank, world, "keypoints", dev, collective=coll)
    se2.step(True)
    torch.cuda.synchronize()
    got2, ref2 = se2.reduced.clone(), se2.reference()
    res["keypoints_vs_single_gpu"] = rel(got2, ref2)
    # ---- pairs mode: 7 rendered pairs, pair b on rank b % world, slices of one zero send buffer, ONE
    # out-of-place all-reduce; three steps in a row (the send buffer's foreign slices must still be zero afterwards)
    rb = wl.RenderedPairBatch(ctx, 7, H=240, W=320, S=8, k=4, device=dev, seed=5)
    # (the default collective of pair mode: ONE in-place all-gather of equal slices -- 7 pairs do not divide by 2)
    se3g = shard.ShardedEvaluation(ctx, rb.array, 4, rank, world, "pairs", dev, collective=coll)
    for _ in range(3):
        se3g.step(True)
    torch.cuda.synchronize()
    res["pairs_allgather_vs_single_gpu"] = rel(se3g.reduced.clone(), se3g.reference())
    res["pairs_allgather_rows"] = int(se3g.rows)
    se3 = shard.ShardedEvaluation(ctx, rb.array, 4, rank, world, "pairs", dev, collective=coll, pair_collective="allreduce")
    for _ in range(3):
        se3.step(True)
    torch.cuda.synchronize()
    got3, ref3 = se3.reduced.clone(), se3.reference()
    res["pairs_vs_single_gpu"] = rel(got3, ref3)
    res["pairs_gather_equals_reduce"] = bool(all(torch.equal(se3g.blocks_of_pair(b), se3.blocks_of_pair(b)) for b in range(7)))
    lo, hi = se3.row_base[rank] * se3.E, se3.row_base[rank + 1] * se3.E
    res["pairs_own_slice_exact"] = bool(torch.equal(got3[lo:hi], se3.send[lo:hi]))
    res["pairs_foreign_slices_zero"] = bool(float(se3.send[:lo].abs().sum() + se3.send[hi:].abs().sum()) == 0.0)
    op3, keep3 = B.make_problem(*_oracle_args(rb.host_problem(3)))  # (keep3: the arrays the problem struct points into)
    o3 = B.evaluate(op3)["frame_blocks"][0]
    g3 = se3.blocks_of_pair(3)[0].cpu().numpy()
    res["pairs_vs_oracle"] = float(np.abs(g3 - o3).max() / np.abs(o3).max())
    # ---- whole alignments sharded (shard.ShardedLmBatch): the batched LM on this rank's pairs of the same 7, ONE all-gather
    # of the records; against the whole batch aligned by this rank alone (same counts, knots to 1e-9)
    o = M.capi.LmBatchOpts()
    o.spline_deg_k, o.max_num_iterations, o.max_consecutive_nonmonotonic_steps, o.solver_type, o.sync_every = 4, 6, 5, 0, 0
    o.min_step_quality, o.min_abs_cost_decrease, o.max_chi_square_error = 0.5, 0.0, 3.0
    init = [(h["kt"], h["kR"]) for h in rb._host]
    sl = shard.ShardedLmBatch(ctx, rb.array, 4, rank, world, dev, o, init, collective=coll)
    assert sl.run() == 0
    alone = shard.ShardedLmBatch(ctx, rb.array, 4, 0, 1, dev, o, init)
    assert alone.run() == 0
    worst, same = 0.0, True
    for b in range(7):
        a, r = sl.record(b, 4), alone.record(b, 4)
        same = same and (a["iterations"], a["accepted"], a["rejected"], a["invalid"]) == (r["iterations"], r["accepted"], r["rejected"], r["invalid"])
        worst = max(worst, float(np.abs(a["knots_t"] - r["knots_t"]).max()), float(np.abs(a["knots_R"] - r["knots_R"]).max()))
    res["lm_sharded_same_counts"] = bool(same and sum(alone.record(b, 4)["accepted"] for b in range(7)) > 0)
    res["lm_sharded_knots_vs_single_gpu"] = worst
    res["nonzero"] = bool(float(ref.abs().max()) > 0 and float(ref2.abs().max()) > 0 and float(ref3.abs().max()) > 0)
    # every rank holds the same reduced object
    chk = got2.clone() if comm == "rccl" else got2.cpu()
    dist.all_reduce(chk, op=dist.ReduceOp.MAX)
    res["ranks_agree"] = bool(torch.equal(chk, got2 if comm == "rccl" else got2.cpu()))
    gt = se3g.reduced.clone() if comm == "rccl" else se3g.reduced.cpu()
    gmax = gt.clone()
    dist.all_reduce(gmax, op=dist.ReduceOp.MAX)
    res["ranks_agree"] = res["ranks_agree"] and bool(torch.equal(gmax, gt))
    if comm == "p2p":
        # the collectives alone, against torch: odd counts, unaligned slices, many steps in a row (two parities of slots), every
        # rank's result identical bits; the all-reduce adds in rank order
        g = torch.Generator(device="cpu").manual_seed(1234)
        every = [torch.randn(40001, dtype=torch.float64, generator=g) for _ in range(world)]
        want_sum = every[0].clone()
        for r in range(1, world):
            want_sum = want_sum + every[r]
        ok_sum, ok_gather = True, True
        for step in range(60):
            # (40 001 doubles = 20 workgroups per rank: the in-place sum must walk the elements in the send phase's own partition --
            # a workgroup that summed an element another workgroup had not sent yet was round 5's one race, seen once in three runs;
            # tools/p2p_stress.py runs random sequences up to 300 001 doubles)
            n = (5001, 1, 777, 4096, 40001, 20000)[step % 6]
            v = (every[rank][:n] * (step + 1)).to(dev)
            coll.allreduce(v, v, n)
            w = want_sum[:n].clone() if world == 1 else None
            acc = every[0][:n] * (step + 1)
            for r in range(1, world):
                acc = acc + every[r][:n] * (step + 1)
            torch.cuda.synchronize()
            ok_sum = ok_sum and bool(torch.equal(v.cpu(), acc))
            base = torch.zeros(world * n + 1, dtype=torch.float64, device=dev)[1:]  # 8-byte aligned only
            base[rank * n:(rank + 1) * n] = (every[rank][:n] + step).to(dev)
            coll.allgather(base, n)
            torch.cuda.synchronize()
            ok_gather = ok_gather and bool(torch.equal(base.cpu(), torch.cat([every[r][:n] + step for r in range(world)])))
        res["p2p_allreduce_rank_order_exact"], res["p2p_allgather_exact"] = ok_sum, ok_gather
        res["p2p_capacity_grew"] = bool(coll.cap > 4096)
        coll.close()
        assert ctx.lib.mbavo_p2p_ranks(ctx.handle) == 0
    dist.barrier()
    assert ctx.lib.mbavo_comm_destroy(ctx.handle) == 0 and ctx.lib.mbavo_comm_ranks(ctx.handle) == 0
    ctx.close()
    dist.destroy_process_group()
    if rank == 0 and out_path:
        json.dump(res, open(out_path, "w"))
    return res


def _oracle_args(p):
    return (p.S, p.F, p.K, p.P, p.k, p.N, p.H, p.W, p.ref, p.grad, p.cur, p.kp_xy, p.kp_z, p.pattern, p.intr, p.cap, p.exp,
            p.t0, p.dt, p.knots_t, p.knots_R, p.start_idx, p.huber)


def check(res, world):
    if "p2p_allgather_exact" in res:
        assert res["p2p_allreduce_rank_order_exact"] and res["p2p_allgather_exact"] and res["p2p_capacity_grew"]
    assert res["world"] == world and res["rccl_ranks"] == world
    assert res["nonzero"] and res["ranks_agree"] and res["device_merge_equals_host_merge"]
    assert res["frames_vs_single_gpu"] <= 1e-12 and res["keypoints_vs_single_gpu"] <= 1e-12
    assert res["frame_blocks_vs_single_gpu"] <= 1e-12 and res["frame_blocks_foreign_slices_zero"]
    assert res["frame_blocks_merged_vs_frames_mode"] <= 1e-12
    assert res["frames_vs_oracle"] <= 1e-9 and res["pairs_vs_oracle"] <= 1e-9
    assert res["pairs_vs_single_gpu"] <= 1e-12 and res["pairs_own_slice_exact"] and res["pairs_foreign_slices_zero"]
    assert res["pairs_allgather_vs_single_gpu"] <= 1e-12 and res["pairs_gather_equals_reduce"]
    assert res["pairs_allgather_rows"] == world * -(-7 // world)
    assert res["lm_sharded_same_counts"] and res["lm_sharded_knots_vs_single_gpu"] <= 1e-9


def _spawned(rank, world, port, out_path, comm="rccl"):
    rank_body(rank, world, rank if comm == "rccl" else 0, port, out_path, comm)


def _run(world, tmp_path, comm="rccl"):
    import torch.multiprocessing as mp
    out = str(tmp_path / "res.json")
    port = 29600 + (os.getpid() % 2000)
    mp.spawn(_spawned, args=(world, port, out, comm), nprocs=world, join=True)
    res = json.load(open(out))
    check(res, world)
    return res


def test_sharded_evaluation_one_rank(mbavo, tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _run(1, tmp_path)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_evaluation_ranks_sharing_one_gpu(mbavo, tmp_path, world):
    """Two (three) processes, each with its own HIP context on GPU 0: the whole sharded path of ranks > 0 on the HIP engine
    (frames / frame_blocks / keypoints / pairs by all-gather and by all-reduce, ShardedLmBatch), gloo on a host copy in
    the place of the RCCL collective (module docstring).  Same assertions as with RCCL."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    res = _run(world, tmp_path, comm="gloo")
    assert res["collective"].startswith("gloo")


@pytest.mark.parametrize("world", [1, 2, 3])
def test_sharded_evaluation_p2p_collectives_ranks_sharing_one_gpu(mbavo, tmp_path, world):
    """The same body with the PRODUCT's one-shot collectives (csrc/p2p_comm.hip: peer-mapped receive regions over hipIpc, one
    kernel per collective, no RCCL) between two and three processes that share GPU 0 -- every sharding mode, the batched LM's
    record gather, and the collectives alone against torch (bit-exact rank-order sums, odd counts, 8-byte-aligned slices,
    40 steps in a row across both slot parities, regions growing on demand).  VERDICT r04 next-round 4."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    res = _run(world, tmp_path, comm="p2p")
    assert res["collective"].startswith("p2p")


def _stress(args, tmp_path, timeout=600):
    import subprocess
    import sys
    out = tmp_path / "stress"
    out.mkdir()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "p2p_stress.py")] + args + ["--out-dir", str(out)], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    lines = {f: open(str(out / f)).read().strip() for f in sorted(os.listdir(str(out)))}
    return p.returncode, lines, p.stdout.decode()[-3000:]


@pytest.mark.parametrize("world", [4, 8])
def test_p2p_collectives_under_skew(mbavo, tmp_path, world):
    """tools/p2p_stress.py with the rank counts a node will use (4 and 8 processes sharing GPU 0): 400 / 200 mixed all-reduces and
    all-gathers of 8 B .. 2.4 MB (the path's 2.6 KB .. 1.33 MB inside) with one rank delayed by 1 ms every 100 steps and rank 0
    jittering -- the two-parity slot logic under skew: a rank may run ahead of a slow peer by at most one collective.  Every result
    bit for bit against torch on the host, on every rank (10 000 steps: profiles/r06_fuzz.txt)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    steps = 400 if world <= 4 else 200
    rc, lines, tail = _stress([str(world), str(steps), "--skew"], tmp_path)
    assert rc == 0 and len(lines) == world, tail
    for r in range(world):
        assert lines["rank%d.txt" % r].startswith("rank %d of %d: %d steps with skew, 0 bad" % (r, world, steps)), lines


def test_p2p_lost_peer_times_out(mbavo, tmp_path):
    """One of three ranks leaves without tear-down after 20 collectives (a crashed peer).  The others' next collective gives up
    within the configured wait (mbavo_p2p_set_timeout 0.5 s instead of the default 20 s): mbavo_p2p_status says MBAVO_E_TIMEOUT
    and what the collective would have written is NaN -- an unreduced buffer cannot pass for a result (VERDICT r05
    next-round 6b)."""
    import re
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    rc, lines, tail = _stress(["3", "40", "--kill-rank", "1", "--kill-at", "20", "--timeout", "0.5"], tmp_path, timeout=300)
    assert rc == 0 and sorted(lines) == ["rank0.txt", "rank2.txt"], (rc, lines, tail)
    for f, line in lines.items():
        m = re.search(r"status (-?\d+) after ([0-9.]+) s, output NaN: (\w+)", line)
        assert m and int(m.group(1)) == -4 and m.group(3) == "True" and 0.4 <= float(m.group(2)) < 5.0, line


def test_sharded_evaluation_two_ranks(mbavo, tmp_path):
    import torch
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    _run(2, tmp_path)


if __name__ == "__main__":  # under torchrun
    w, r, lr = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    out = rank_body(r, w, lr, int(os.environ.get("MASTER_PORT", "29650")), None)
    if r == 0:
        check(out, w)
        print("test_gpu_dist under torchrun: OK", json.dumps(out))


@pytest.mark.parametrize("world", [2, 4, 8])
def test_frame_shards_of_every_rank_add_up_on_one_gpu(mbavo, world):
    """bench.py --gpus N on configs[1]: N blurred frames of one joint problem, frame r on rank r (mbavo_shard_frames),
    every rank's blocks scattered into the 6N x 6N system on the device (mbavo_merge_device), the systems summed.  Here
    the N ranks' shards are evaluated one after the other on THIS GPU by the HIP engine (a shard has one frame: the
    fused kernel takes the pose prologue) and summed on the host in rank order, against the whole problem evaluated at
    once: 1e-12 (only the summation order differs).  No communicator involved: what this pins is the sharding of
    ranks > 0, which a one-GPU box cannot reach through RCCL."""
    import torch
    from mba_vo_amd import shard, workloads as wl
    dev = "cuda:0"
    ctx = mbavo.capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    try:
        probs = wl.pyramid_pair(240, 320, 3, S=8, k=4, N=4, mode="dense", seed=11, frames=world)
        dw = wl.DeviceWorkload(probs, device=dev)
        total, ref, total_fb, ref_fb = None, None, None, None
        for r in range(world):
            se = shard.ShardedEvaluation(ctx, dw.array, 4, r, world, "frames", dev)
            assert sum(se.shards[b].F for b in range(se.B)) == len(probs)  # one frame of every pyramid level
            se.step(True, reduce=False)
            torch.cuda.synchronize()
            part = se.reduced.clone()
            total = part if total is None else total + part
            # the default of bench.py: the packed blocks in the rank's slice of the zero send buffer, no merge kernel
            sb = shard.ShardedEvaluation(ctx, dw.array, 4, r, world, "frame_blocks", dev)
            sb.step(True, reduce=False)
            torch.cuda.synchronize()
            total_fb = sb.send.clone() if total_fb is None else total_fb + sb.send
            if r == world - 1:
                ref, ref_fb = se.reference(), sb.reference()
        assert float(ref.abs().max()) > 0 and float(ref_fb.abs().max()) > 0
        assert float((total - ref).abs().max() / ref.abs().max()) <= 1e-12
        assert float((total_fb - ref_fb).abs().max() / ref_fb.abs().max()) <= 1e-12
    finally:
        ctx.close()
