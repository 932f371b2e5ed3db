"""Stress of the one-shot p2p collectives between processes that share the visible GPU(s) (csrc/p2p_comm.hip; the reduction
point of the path, merge_hessian_gradient_cost.cpp:39-86, across ranks):

    python tools/p2p_stress.py [world] [steps] [--skew] [--kill-rank R --kill-at S --timeout T]

Every rank runs the same pseudo-random sequence of in-place all-reduces and all-gathers (sizes 1 .. 300 000 doubles: 8 B .. 2.4 MB,
the path's 2.6 KB .. 1.33 MB messages inside; aligned and 8-byte-aligned buffers) and checks every result bit for bit against torch
on the host; prints the first mismatches and one summary line per rank.
  --skew     one rank (step // 100 % world) sleeps 1 ms before its collective every 100 steps, and rank 0 every 7th step for a
             random 0 .. 300 us: the parity-slot logic (a rank may run ahead of a slow peer by at most one collective) under skew;
  --kill-rank R --kill-at S: rank R leaves WITHOUT tear-down after S collectives (os._exit: a crashed peer); the others must come
             back from collective S + 1 within the timeout (--timeout, seconds; mbavo_p2p_set_timeout) with MBAVO_E_TIMEOUT in
             mbavo_p2p_status and a NaN-filled output -- each prints "rank r: peer lost -> status -4 after X s, output NaN: True"."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SIZES = [1, 2, 3, 325, 511, 2600, 4097, 20000, 65537, 166400, 300001]  # (325 doubles = one packed block, 166400 = 512 of them)


def body(rank, world, port, steps, skew, kill_rank, kill_at, timeout, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("OMP_NUM_THREADS", "1")       # (the sandboxed hosts meter CPU time: 8 ranks x a thread pool each would thrash)
    import numpy as np
    import torch
    import torch.distributed as dist
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import mba_vo_amd as M
    from mba_vo_amd import shard
    ctx = M.capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    # (kill mode: the regions at their final size from the start -- growing them is a rendezvous of ALL ranks)
    coll = shard.P2PCollective(ctx, rank, world, max_doubles=(1 << 12) if kill_rank < 0 else 300001, timeout=timeout)
    rng = np.random.default_rng(7)                      # the same sequence on every rank
    lag = np.random.default_rng(1000 + rank)            # (the delays are a rank's own)
    g = torch.Generator().manual_seed(99)
    # every rank's vectors live on the GPU: the expected results are formed there too, in the collective's own order (rank order,
    # fp64 adds are the same IEEE operations on either side), so the check costs the host nothing
    pool = [torch.randn(300001, dtype=torch.float64, generator=g).to("cuda:0") for _ in range(world)]
    bad, step = 0, -1
    t0 = time.time()
    for step in range(steps):
        n = int(rng.choice(SIZES))
        mode = int(rng.integers(0, 2))
        off = int(rng.integers(0, 2))                   # 8-byte-only alignment half of the time
        scale = float(step + 1)
        if kill_rank == rank and step == kill_at:
            torch.cuda.synchronize()
            print("rank %d: leaving without tear-down after %d collectives" % (rank, step), flush=True)
            os._exit(0)
        if skew and ((step % 100 == 0 and (step // 100) % world == rank) or (rank == 0 and step % 7 == 3)):
            time.sleep(1e-3 if step % 100 == 0 else float(lag.uniform(0.0, 3e-4)))
        if mode == 0:
            buf = torch.zeros(n + off, dtype=torch.float64, device="cuda:0")[off:]
            buf.copy_(pool[rank][:n] * scale)
            coll.allreduce(buf, buf, n)
            want = pool[0][:n] * scale
            for r in range(1, world):
                want = want + pool[r][:n] * scale
        else:
            buf = torch.zeros(world * n + off, dtype=torch.float64, device="cuda:0")[off:]
            buf[rank * n:(rank + 1) * n] = pool[rank][:n] + scale
            coll.allgather(buf, n)
            want = torch.cat([pool[r][:n] + scale for r in range(world)])
        t_c = time.time()
        torch.cuda.synchronize()
        st = coll.status()
        got = buf
        if kill_rank >= 0 and step >= kill_at:
            # the peer is gone: this collective must have given up within the timeout, flagged and poisoned
            if mode == 0:
                poisoned = bool(torch.isnan(got).all())
            else:
                mine = got[rank * n:(rank + 1) * n]
                others = torch.cat([got[r * n:(r + 1) * n] for r in range(world) if r != rank])
                poisoned = bool(torch.isnan(others).all()) and bool(torch.equal(mine, want[rank * n:(rank + 1) * n]))
            line = "rank %d: peer lost -> status %d after %.2f s, output NaN: %s" % (rank, st, time.time() - t_c, poisoned)
            print(line, flush=True)
            if out_dir:
                open(os.path.join(out_dir, "rank%d.txt" % rank), "w").write(line + "\n")
            os._exit(0 if (st == -4 and poisoned) else 5)  # (no tear-down: its barriers would wait for the lost rank)
        if st != 0 or not torch.equal(got, want):
            got, want = got.cpu(), want.cpu()
            w = torch.nonzero(got != want).flatten()
            print("rank %d step %d mode %s n %d off %d status %d: %d wrong, first at %s (got %r want %r)" %
                  (rank, step, "allreduce" if mode == 0 else "allgather", n, off, st, w.numel(), w[:4].tolist(),
                   got[w[:2]].tolist(), want[w[:2]].tolist()), flush=True)
            bad += 1
            if bad > 3:
                break
    line = "rank %d of %d: %d steps%s, %d bad, %.1f s" % (rank, world, step + 1, " with skew" if skew else "", bad, time.time() - t0)
    print(line, flush=True)
    if out_dir:
        open(os.path.join(out_dir, "rank%d.txt" % rank), "w").write(line + "\n")
    coll.close()
    ctx.close()
    dist.destroy_process_group()
    if bad:
        sys.exit(4)


def main(argv):
    import argparse
    import torch.multiprocessing as mp
    ap = argparse.ArgumentParser()
    ap.add_argument("world", nargs="?", type=int, default=3)
    ap.add_argument("steps", nargs="?", type=int, default=300)
    ap.add_argument("--skew", action="store_true")
    ap.add_argument("--kill-rank", type=int, default=-1)
    ap.add_argument("--kill-at", type=int, default=50)
    ap.add_argument("--timeout", type=float, default=None)
    ap.add_argument("--out-dir", default=None, help="every rank also writes its summary line to <out-dir>/rank<r>.txt")
    a = ap.parse_args(argv)
    ctx = mp.spawn(body, args=(a.world, 29800 + os.getpid() % 100, a.steps, a.skew, a.kill_rank, a.kill_at, a.timeout, a.out_dir),
                   nprocs=a.world, join=False)
    # (join=False + our own wait: in the kill mode the ranks leave by os._exit at different times, which mp.spawn's join treats as a failure)
    deadline = time.time() + 3600
    codes = None
    while time.time() < deadline:
        codes = [p.exitcode for p in ctx.processes]
        if all(c is not None for c in codes):
            break
        time.sleep(0.05)
    for p in ctx.processes:
        if p.exitcode is None:
            p.terminate()
    return 0 if codes and all(c == 0 for c in codes) else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
