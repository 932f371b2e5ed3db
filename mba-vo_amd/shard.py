"""Multi-GPU sharding of the tracking path: one process per GPU, plumbing over the C ABI (include/mbavo.h, "multi-GPU").

The path shards by construction: every pixel-sample is independent and the only coupling is the sum into the normal
equations (the reference's reduction point is merge_hessian_gradient_cost.cpp:39-86).
  * ONE joint problem, keypoints sharded (`mbavo_shard_keypoints`): contiguous keypoint bands per rank; every rank's
    packed frame blocks are partial sums scaled by the WHOLE problem's residual count (`num_residuals`), so the blocks
    of all ranks add up to the whole problem's blocks: one all-reduce of the B*F*E packed doubles.
  * ONE joint problem, frames sharded (`mbavo_shard_frames`): rank r owns a contiguous frame range; every rank scatters
    its frames' blocks into the 6N x 6N system on the device (`mbavo_merge_device`) and the partial systems
    [cost | g | H] are summed: one all-reduce of 1 + 6N + 36N^2 doubles per problem.
Both end in `mbavo_allreduce_blocks` on the context's own RCCL communicator (`mbavo_comm_init`), enqueued on the stream
of the evaluation.  torch.distributed is used for the rendezvous only (broadcast of the 128-byte communicator id).
The pure index helpers below are also what the CPU (gloo) tests exercise.
"""
import ctypes as C

import numpy as np

from . import capi


def pairs_of_rank(num_pairs, rank, world):
    """Indices of the independent pairs owned by `rank` (round-robin: pair b -> rank b % world)."""
    return list(range(rank, num_pairs, world))


def keypoint_range_of_rank(K, rank, world):
    """Contiguous keypoint range [lo, hi) of a joint problem for `rank` (== mbavo_shard_keypoints)."""
    return (K * rank) // world, (K * (rank + 1)) // world


def frame_range_of_rank(F, rank, world):
    """Contiguous frame range [lo, hi) of a joint problem for `rank` (== mbavo_shard_frames)."""
    return (F * rank) // world, (F * (rank + 1)) // world


def shard_array(lib, whole, rank, world, mode):
    """Problem array of the whole workload -> (array of this rank's shards, one per problem; first index per problem).
    mode 'keypoints' | 'frames'.  Frame shards may be empty (F == 0) when world > F."""
    B = len(whole)
    out = (capi.Problem * B)()
    first = np.zeros(B, np.int32)
    fn = lib.mbavo_shard_keypoints if mode == "keypoints" else lib.mbavo_shard_frames
    for b in range(B):
        f = C.c_int(0)
        capi.check(fn(C.byref(whole[b]), rank, world, C.byref(out[b]), C.byref(f)), "mbavo_shard_" + mode)
        first[b] = f.value
    return out, first


def comm_init(ctx, rank, world, bcast):
    """Create the context's RCCL communicator.  `bcast(bytes_or_None) -> bytes` hands rank 0's 128-byte id to every rank
    (torch.distributed broadcast in bench.py / the tests; identity for world == 1)."""
    ident = C.create_string_buffer(128)
    if rank == 0:
        capi.check(ctx.lib.mbavo_comm_unique_id(ident), "mbavo_comm_unique_id")
    raw = bcast(ident.raw if rank == 0 else None)
    capi.check(ctx.lib.mbavo_comm_init(ctx.handle, raw, rank, world), "mbavo_comm_init")
    n = ctx.lib.mbavo_comm_ranks(ctx.handle)
    if n != world:
        raise RuntimeError("RCCL communicator has %d ranks, expected %d" % (n, world))
    return n


def torch_bcast(device):
    """bcast callable for comm_init on an initialised torch.distributed process group."""
    import torch
    import torch.distributed as dist

    def bcast(raw):
        t = torch.zeros(128, dtype=torch.uint8, device=device)
        if raw is not None:
            t.copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
        dist.broadcast(t, src=0)
        return bytes(t.cpu().numpy().tobytes())
    return bcast


class ShardedEvaluation:
    """This rank's share of one GN-iteration evaluation of a workload (a list of joint problems resident on this GPU as
    `whole`, a capi.Problem array) and the reduction of the normal equations over the ranks.

        step()      : evaluate the shard (+ device merge in 'frames' mode) + ONE all-reduce; asynchronous
        reduced     : device tensor holding the reduced object after step(): packed frame blocks of the whole workload
                      ('keypoints') or the merged [cost | g | H] systems ('frames')
        reference() : the same object computed by THIS rank alone from the whole workload (for the N = 1 equality check)
    """

    def __init__(self, ctx, whole, k, rank, world, mode, device):
        import torch
        assert mode in ("keypoints", "frames")
        self.ctx, self.whole, self.k, self.rank, self.world, self.mode = ctx, whole, k, rank, world, mode
        self.B = len(whole)
        lib = ctx.lib
        self.E = lib.mbavo_packed_len(k)
        self.shards, self.first = shard_array(lib, whole, rank, world, mode)
        live = [b for b in range(self.B) if self.shards[b].F > 0]
        self.live = (capi.Problem * max(len(live), 1))(*[self.shards[b] for b in live])
        self.n_live = len(live)
        self.nbf = sum(self.shards[b].F for b in range(self.B))
        self.nbf_whole = sum(whole[b].F for b in range(self.B))
        self.sys_len = sum(lib.mbavo_system_len(whole[b].N) for b in range(self.B))
        z = lambda n: torch.zeros(max(n, 1), dtype=torch.float64, device=device)
        self.frame_blocks, self.valid = z(self.nbf * self.E), z(self.nbf)
        self.systems = z(self.sys_len) if mode == "frames" else None
        self.reduced = self.systems if mode == "frames" else self.frame_blocks
        self.count = self.sys_len if mode == "frames" else self.nbf * self.E
        self._ref_fb, self._ref_sys = z(self.nbf_whole * self.E), z(self.sys_len)

    def evaluate_local(self, with_hessian=True):
        lib, ctx = self.ctx.lib, self.ctx
        if self.n_live:
            capi.check(lib.mbavo_eval_batch(ctx.handle, self.n_live, self.live, self.k, 1 if with_hessian else 0,
                                            self.frame_blocks.data_ptr(), None, self.valid.data_ptr()), "mbavo_eval_batch")
        if self.mode == "frames":
            capi.check(lib.mbavo_merge_device(ctx.handle, self.B, self.shards, self.k, self.frame_blocks.data_ptr(),
                                              self.systems.data_ptr()), "mbavo_merge_device")

    def step(self, with_hessian=True, reduce=True):
        self.evaluate_local(with_hessian)
        if reduce:
            capi.check(self.ctx.lib.mbavo_allreduce_blocks(self.ctx.handle, None, self.reduced.data_ptr(), self.count),
                       "mbavo_allreduce_blocks")

    def reference(self):
        """The reduced object of the WHOLE workload evaluated by this rank alone (synchronous; returns a clone)."""
        import torch
        lib, ctx = self.ctx.lib, self.ctx
        capi.check(lib.mbavo_eval_batch(ctx.handle, self.B, self.whole, self.k, 1, self._ref_fb.data_ptr(), None, None),
                   "mbavo_eval_batch")
        if self.mode == "frames":
            capi.check(lib.mbavo_merge_device(ctx.handle, self.B, self.whole, self.k, self._ref_fb.data_ptr(),
                                              self._ref_sys.data_ptr()), "mbavo_merge_device")
        torch.cuda.synchronize()
        return (self._ref_sys if self.mode == "frames" else self._ref_fb).clone()


def allreduce_blocks(blocks, group=None):
    """In-place sum over ranks of a tensor of packed blocks through torch.distributed (CPU / gloo tests)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(blocks, op=dist.ReduceOp.SUM, group=group)
    return blocks
