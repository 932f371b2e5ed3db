"""Stress of the one-shot p2p collectives between processes that share the visible GPU(s): python tools/p2p_stress.py [world] [steps]
Every rank runs the same pseudo-random sequence of in-place all-reduces and all-gathers (sizes 1 .. 300 000 doubles, aligned and
8-byte-aligned buffers) and checks every result bit for bit against torch on the host; prints the first mismatch."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def body(rank, world, port, steps):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import numpy as np
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import mba_vo_amd as M
    from mba_vo_amd import shard
    ctx = M.capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    coll = shard.P2PCollective(ctx, rank, world, max_doubles=1 << 12)
    rng = np.random.default_rng(7)                      # the same sequence on every rank
    g = torch.Generator().manual_seed(99)
    pool = [torch.randn(300001, dtype=torch.float64, generator=g) for _ in range(world)]
    bad = 0
    t0 = time.time()
    for step in range(steps):
        n = int(rng.choice([1, 2, 3, 511, 2600, 4097, 20000, 65537, 166400, 300001]))
        mode = int(rng.integers(0, 2))
        off = int(rng.integers(0, 2))                   # 8-byte-only alignment half of the time
        scale = float(step + 1)
        if mode == 0:
            buf = torch.zeros(n + off, dtype=torch.float64, device="cuda:0")[off:]
            buf.copy_(pool[rank][:n] * scale)
            coll.allreduce(buf, buf, n)
            want = pool[0][:n] * scale
            for r in range(1, world):
                want = want + pool[r][:n] * scale
        else:
            buf = torch.zeros(world * n + off, dtype=torch.float64, device="cuda:0")[off:]
            buf[rank * n:(rank + 1) * n] = (pool[rank][:n] + scale).to("cuda:0")
            coll.allgather(buf, n)
            want = torch.cat([pool[r][:n] + scale for r in range(world)])
        torch.cuda.synchronize()
        st = ctx.lib.mbavo_p2p_status(ctx.handle)
        got = buf.cpu()
        if st != 0 or not torch.equal(got, want):
            w = torch.nonzero(got != want).flatten()
            print("rank %d step %d mode %s n %d off %d status %d: %d wrong, first at %s (got %r want %r)" %
                  (rank, step, "allreduce" if mode == 0 else "allgather", n, off, st, w.numel(), w[:4].tolist(),
                   got[w[:2]].tolist(), want[w[:2]].tolist()), flush=True)
            bad += 1
            if bad > 3:
                break
    print("rank %d: %d steps, %d bad, %.1f s" % (rank, step + 1, bad, time.time() - t0), flush=True)
    coll.close()
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    mp.spawn(body, args=(world, 29800 + os.getpid() % 100, steps), nprocs=world, join=True)
