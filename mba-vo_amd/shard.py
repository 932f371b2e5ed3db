"""Multi-GPU sharding of the tracking path: one process per GPU, plumbing over the C ABI (include/mbavo.h, "multi-GPU").

The path shards by construction: every pixel-sample is independent and the only coupling is the sum into the normal
equations (the reference's reduction point is merge_hessian_gradient_cost.cpp:39-86).
  * ONE joint problem, keypoints sharded (`mbavo_shard_keypoints`): contiguous keypoint bands per rank; every rank's
    packed frame blocks are partial sums scaled by the WHOLE problem's residual count (`num_residuals`), so the blocks
    of all ranks add up to the whole problem's blocks: one all-reduce of the B*F*E packed doubles.
  * B INDEPENDENT pairs, pairs sharded (`pairs_of_rank`): pair b on rank b % world, whole; every rank
    evaluates its pairs into ITS slice of the result buffer (`pair_layout`: equal slices, rank-major) and ONE in-place
    all-gather (`mbavo_allgather_blocks`, the default since round 4) leaves every pair's blocks on every rank -- the
    pairs share nothing, so nothing has to be ADDED: (N-1)/N of the buffer crosses the links once.  `pair_collective=
    "allreduce"` keeps the wording of BASELINE.json's north_star ("a final RCCL all-reduce of the normal equations"): the
    slices sit in a zero send buffer and ONE out-of-place all-reduce (`mbavo_allreduce_blocks_to`) sums x + 0 + ... + 0 --
    exact and bit-identical, at twice the bytes on the wire.
  * ONE joint problem, frames sharded, packed blocks summed ('frame_blocks', bench.py's default for a single pair): as
    'frames' below, but what travels are the packed blocks: with the same number of frame rows on every rank (frame r on rank r)
    the rank's evaluation writes straight into its slice of the result buffer and ONE in-place all-gather of equal slices
    moves them (round 5; `pair_collective="allreduce"` or ragged shares: a zero send buffer and an out-of-place all-reduce);
    the scatter into the 6N x 6N system is the consumer's (no merge kernel in the step).
  * ONE joint problem, frames sharded (`mbavo_shard_frames`): rank r owns a contiguous frame range; every rank scatters
    its frames' blocks into the 6N x 6N system on the device (`mbavo_merge_device`) and the partial systems
    [cost | g | H] are summed: one all-reduce of 1 + 6N + 36N^2 doubles per problem.
  * B INDEPENDENT pairs, whole ALIGNMENTS sharded (`ShardedLmBatch`): pair b on rank b % world, the device-side LM loop
    (`mbavo_lm_batch`) runs on every rank over its own pairs with no collective inside, and ONE all-gather of fixed-size
    records (final knots + result scalars, `lm_record_layout`) at the end leaves every pair's aligned spline on every rank.
The evaluations end in ONE `mbavo_allreduce_blocks[_to]` on the context's own RCCL communicator (`mbavo_comm_init`), enqueued on the stream
of the evaluation.  torch.distributed is used for the rendezvous only (broadcast of the 128-byte communicator id).
The pure index helpers below are also what the CPU (gloo) tests exercise.
"""
import ctypes as C

import numpy as np

from . import capi


def pairs_of_rank(num_pairs, rank, world):
    """Indices of the independent pairs owned by `rank` (round-robin: pair b -> rank b % world)."""
    return list(range(rank, num_pairs, world))


def pair_layout(frames_per_pair, world, equal_slices=False):
    """Rank-major layout of the packed frame blocks of B independent pairs sharded pair -> rank b % world.
    Returns (row_base, row_of_pair): rank r's pairs occupy rows [row_base[r], ...) of the result buffer, in ascending pair
    order; pair b's first frame block is row row_of_pair[b].  Every rank writes a CONTIGUOUS slice, so its evaluation
    writes straight into the buffer of the collective (no scatter kernel).  equal_slices (what ncclAllGather wants): every
    rank's slice has the rows of the fullest rank, the unused tail rows of the others stay zero."""
    B = len(frames_per_pair)
    per_rank = [sum(int(frames_per_pair[b]) for b in pairs_of_rank(B, r, world)) for r in range(world)]
    width = max(per_rank) if per_rank else 0
    row_base = [0]
    row_of_pair = [0] * B
    for r in range(world):
        row = row_base[-1]
        for b in pairs_of_rank(B, r, world):
            row_of_pair[b] = row
            row += int(frames_per_pair[b])
        row_base.append(row_base[-1] + width if equal_slices else row)
    return row_base, row_of_pair


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


def ctx_stream(ctx):
    """torch stream context of the stream the library works on (mbavo_set_stream); the null stream if none was set."""
    import torch
    if not ctx.stream:
        return torch.cuda.stream(torch.cuda.default_stream())
    return torch.cuda.stream(torch.cuda.ExternalStream(ctx.stream))


class RcclCollective:
    """The product's collectives: mbavo_allreduce_blocks[_to] / mbavo_allgather_blocks on the context's own RCCL
    communicator (mbavo_comm_init), enqueued on the stream of the evaluation.  Tensors are device tensors of doubles."""
    name = "rccl"

    def __init__(self, ctx):
        self.ctx = ctx

    def allreduce(self, send, recv, count):
        lib, h = self.ctx.lib, self.ctx.handle
        if send.data_ptr() == recv.data_ptr():
            capi.check(lib.mbavo_allreduce_blocks(h, None, recv.data_ptr(), count), "mbavo_allreduce_blocks")
        else:
            capi.check(lib.mbavo_allreduce_blocks_to(h, None, send.data_ptr(), recv.data_ptr(), count), "mbavo_allreduce_blocks_to")

    def allgather(self, buf, count_per_rank):
        capi.check(self.ctx.lib.mbavo_allgather_blocks(self.ctx.handle, None, buf.data_ptr(), count_per_rank), "mbavo_allgather_blocks")


class P2PCollective:
    """The product's one-shot collectives over peer-mapped receive regions (csrc/p2p_comm.hip: mbavo_allgather_blocks_p2p /
    mbavo_allreduce_blocks_p2p): ONE kernel per collective on the evaluation's stream, no RCCL -- every rank stores its slice
    into every peer's region, raises a flag, waits for the peers' flags, gathers / adds in rank order.  Sized for this path's
    latency-bound messages (19 KB ... 1.3 MB).  The ranks may share a GPU (the one-GPU tests) or sit on the GPUs of one node.
    torch.distributed (any backend) carries the 64-byte IPC handles once per (re)allocation and the barrier before a region is
    released; the regions grow on demand -- every rank sees the same sequence of counts, so they grow together."""
    name = "p2p(one-shot, peer-mapped regions)"

    def __init__(self, ctx, rank, world, group=None, max_doubles=1 << 15, timeout=None):
        """timeout: seconds a collective waits for a peer before it gives up (None: the library's 20 s)"""
        self.ctx, self.rank, self.world, self.group, self.cap, self.timeout = ctx, rank, world, group, 0, timeout
        self._ensure(max_doubles)

    def _ensure(self, doubles):
        import torch
        import torch.distributed as dist
        if doubles <= self.cap:
            return
        lib, h = self.ctx.lib, self.ctx.handle
        if self.cap:
            self._teardown()
        cap = max(int(doubles), 2 * self.cap)
        self.cap = 0
        mine = C.create_string_buffer(64)
        # Every decision below is taken on values ALL ranks hold (the gathered return codes), so a failure on one rank -- an
        # allocation, an IPC export or import the platform refuses -- raises on every rank at the same point and leaves nobody
        # waiting in a collective of torch.distributed or of this class.
        rc = lib.mbavo_p2p_create(h, self.rank, self.world, cap, mine)
        every = [(rc, mine.raw)]
        if self.world > 1:
            every = [None] * self.world
            dist.all_gather_object(every, (rc, mine.raw), group=self.group)
        if any(r != 0 for r, _ in every):
            if rc == 0:
                lib.mbavo_p2p_destroy(h)
            raise RuntimeError("mbavo_p2p_create failed on rank(s) %s (codes %s)" % ([i for i, (r, _) in enumerate(every) if r != 0], [r for r, _ in every]))
        rc = lib.mbavo_p2p_connect(h, b"".join(raw for _, raw in every))
        codes = [rc]
        if self.world > 1:
            codes = [None] * self.world
            dist.all_gather_object(codes, rc, group=self.group)
        if any(r != 0 for r in codes):
            lib.mbavo_p2p_disconnect(h)
            if self.world > 1:
                dist.barrier(group=self.group)  # nobody maps anybody any more
            lib.mbavo_p2p_destroy(h)
            raise RuntimeError("mbavo_p2p_connect failed on rank(s) %s (codes %s)" % ([i for i, r in enumerate(codes) if r != 0], codes))
        assert lib.mbavo_p2p_ranks(h) == self.world
        if self.timeout is not None:
            capi.check(lib.mbavo_p2p_set_timeout(h, float(self.timeout)), "mbavo_p2p_set_timeout")
        self.cap = cap

    def status(self):
        """0, or MBAVO_E_TIMEOUT once a collective gave up on a peer (its output was filled with NaN); synchronises the stream.
        The host-side consumer of a reduced buffer calls this before it trusts the numbers."""
        return int(self.ctx.lib.mbavo_p2p_status(self.ctx.handle))

    def allreduce(self, send, recv, count):
        self._ensure(count)
        if send.data_ptr() != recv.data_ptr():
            with ctx_stream(self.ctx):
                recv[:count].copy_(send[:count], non_blocking=True)
        capi.check(self.ctx.lib.mbavo_allreduce_blocks_p2p(self.ctx.handle, recv.data_ptr(), count), "mbavo_allreduce_blocks_p2p")

    def allgather(self, buf, count_per_rank):
        self._ensure(count_per_rank)
        capi.check(self.ctx.lib.mbavo_allgather_blocks_p2p(self.ctx.handle, buf.data_ptr(), count_per_rank), "mbavo_allgather_blocks_p2p")

    def _teardown(self):
        """Two-phase, all ranks together: collectives done -> barrier -> everybody unmaps its peers -> barrier -> everybody frees its
        own region (a region freed while a peer still maps it makes the next hipIpcGetMemHandle fail)."""
        import torch
        import torch.distributed as dist
        lib, h = self.ctx.lib, self.ctx.handle
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier(group=self.group)  # nobody is still reading or writing the regions
        capi.check(lib.mbavo_p2p_disconnect(h), "mbavo_p2p_disconnect")
        if self.world > 1:
            dist.barrier(group=self.group)  # nobody maps anybody any more
        capi.check(lib.mbavo_p2p_destroy(h), "mbavo_p2p_destroy")

    def close(self):
        """All ranks together, after their last collective has completed."""
        import torch
        if self.cap:
            torch.cuda.synchronize()
            st = self.ctx.lib.mbavo_p2p_status(self.ctx.handle)
            self._teardown()
            self.cap = 0
            capi.check(st, "mbavo_p2p_status")


class HostStagedCollective:
    """STAND-IN for RcclCollective where RCCL cannot form the communicator: several ranks on ONE GPU (RCCL refuses duplicate
    devices) -- the two-process -m gpu test and `bench.py --comm gloo`, which execute every line of the N > 1 path on a
    one-GPU box except ncclAllReduce / ncclAllGather themselves.  Same call sites, same device buffers, same layouts: the
    buffer is copied to pinned host memory on the context's stream, the stream is drained, torch.distributed (gloo) runs
    the collective on the host copy, and the result is copied back on the same stream.  Test / measurement plumbing, not
    a product path: nothing selects it unless asked to."""
    name = "gloo(host-staged stand-in for RCCL)"

    def __init__(self, ctx, rank, world, group=None):
        self.ctx, self.rank, self.world, self.group, self._stage = ctx, rank, world, group, {}

    def _host(self, n):
        import torch
        t = self._stage.get(n)
        if t is None:
            t = self._stage[n] = torch.zeros(abs(n), dtype=torch.float64).pin_memory()
        return t

    def allreduce(self, send, recv, count):
        import torch
        import torch.distributed as dist
        h = self._host(count)
        with ctx_stream(self.ctx):
            h.copy_(send[:count], non_blocking=True)
            torch.cuda.current_stream().synchronize()
            if self.world > 1:
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            recv[:count].copy_(h, non_blocking=True)

    def allgather(self, buf, count_per_rank):
        import torch
        import torch.distributed as dist
        n = count_per_rank
        mine, every = self._host(n), self._host(-(n * self.world))  # (two staging buffers even when their sizes coincide)
        with ctx_stream(self.ctx):
            mine.copy_(buf[self.rank * n:(self.rank + 1) * n], non_blocking=True)
            torch.cuda.current_stream().synchronize()
            if self.world > 1:
                dist.all_gather_into_tensor(every, mine, group=self.group)
                buf[:n * self.world].copy_(every, non_blocking=True)


class ShardedEvaluation:
    """This rank's share of one GN-iteration evaluation of a workload (a list of joint problems resident on this GPU as
    `whole`, a capi.Problem array) and the reduction of the normal equations over the ranks.

        step()      : evaluate the shard (+ device merge in 'frames' mode) + ONE all-reduce; asynchronous
        reduced     : device tensor holding the reduced object after step(): packed frame blocks of the whole workload
                      ('keypoints'; 'pairs': the same rows in the rank-major order of pair_layout, see blocks_of_pair) or
                      the merged [cost | g | H] systems ('frames')
        reference() : the same object computed by THIS rank alone from the whole workload (for the N = 1 equality check)
    """

    def __init__(self, ctx, whole, k, rank, world, mode, device, collective=None, pair_collective="allgather",
                 frames_per_pair=None):
        """collective: RcclCollective(ctx) unless given (HostStagedCollective: several ranks on one GPU).
        pair_collective ('pairs' mode): 'allgather' (default) or 'allreduce' (module docstring).
        frames_per_pair ('pairs' mode): F of every pair when `whole` only holds this rank's pairs (weak scaling: every rank
        renders its own pairs only; reference() is then not available)."""
        import torch
        assert mode in ("keypoints", "frames", "frame_blocks", "pairs") and pair_collective in ("allgather", "allreduce")
        self.ctx, self.whole, self.k, self.rank, self.world, self.mode = ctx, whole, k, rank, world, mode
        self.coll = collective if collective is not None else RcclCollective(ctx)
        self.pair_collective = pair_collective if mode in ("pairs", "frame_blocks") else None
        self.fb_allgather = False
        self.B = len(whole)
        lib = ctx.lib
        self.E = lib.mbavo_packed_len(k)
        if mode == "pairs":
            self._init_pairs(device, frames_per_pair)
            return
        if mode == "frame_blocks":
            self._init_frame_blocks(device)
            return
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

    def _init_frame_blocks(self, device):
        """Frames sharded as in 'frames', but what is summed over the ranks are the PACKED FRAME BLOCKS themselves: rank r's
        frames occupy a contiguous slice (rank-major rows) of a zero send buffer its evaluation writes straight into, ONE
        out-of-place all-reduce leaves every frame's block on every rank, and the 6N x 6N scatter
        (merge_hessian_gradient_cost.cpp:39-86) is the consumer's, as on one GPU -- no merge kernel in the step (one launch,
        ~3 us, less than 'frames'; F*E instead of 1 + 6N + 36N^2 doubles per problem on the wire, latency-bound either way)."""
        import torch
        lib, whole, rank, world = self.ctx.lib, self.whole, self.rank, self.world
        self.shards, self.first = shard_array(lib, whole, rank, world, "frames")
        live = [b for b in range(self.B) if self.shards[b].F > 0]
        self.live = (capi.Problem * max(len(live), 1))(*[self.shards[b] for b in live])
        self.n_live = len(live)
        self.nbf = sum(self.shards[b].F for b in range(self.B))
        self.nbf_whole = sum(whole[b].F for b in range(self.B))
        bf_base = np.cumsum([0] + [whole[b].F for b in range(self.B)])
        perm, self.row_base = [], [0]
        for r in range(world):
            for b in range(self.B):
                f0, f1 = frame_range_of_rank(whole[b].F, r, world)
                perm += [int(bf_base[b]) + f for f in range(f0, f1)]
            self.row_base.append(len(perm))
        assert sorted(perm) == list(range(self.nbf_whole)) and self.row_base[rank + 1] - self.row_base[rank] == self.nbf
        self.sys_len = 0
        z = lambda n: torch.zeros(max(n, 1), dtype=torch.float64, device=device)
        # every rank holds the same number of frame rows (bench.py's default: frame r on rank r) and the caller did not ask for the
        # all-reduce: ONE in-place all-gather of equal slices instead -- the rank's evaluation writes straight into its slice of the
        # result buffer, half the bytes on the wire, no send buffer (and for the p2p collective no copy in front of it)
        self.fb_allgather = world > 1 and self.pair_collective == "allgather" and self.nbf > 0 and \
            all(self.row_base[r + 1] - self.row_base[r] == self.nbf for r in range(world))
        if self.fb_allgather:
            self.reduced = z(self.nbf_whole * self.E)
            self.send = self.reduced
            self.frame_blocks = self.reduced[self.row_base[rank] * self.E:self.row_base[rank + 1] * self.E]
        else:
            self.send = z(self.nbf_whole * self.E)
            self.frame_blocks = self.send[self.row_base[rank] * self.E:self.row_base[rank + 1] * self.E] if self.nbf else z(1)
            self.reduced = z(self.nbf_whole * self.E) if world > 1 else self.send  # (one rank: its slice is the whole buffer)
        self.valid = z(self.nbf)
        self.systems = None
        self.count = self.nbf_whole * self.E
        self._ref_fb, self._ref_sys = z(self.nbf_whole * self.E), None
        self._perm = torch.from_numpy(np.array(perm, np.int64)).to(device)
        self.row_of_frame = {int(p): i for i, p in enumerate(perm)}  # problem-major frame slot -> row of `reduced`

    def _init_pairs(self, device, frames_per_pair=None):
        import torch
        lib, whole, rank, world = self.ctx.lib, self.whole, self.rank, self.world
        mine = pairs_of_rank(self.B, rank, world)
        self.partial_whole = frames_per_pair is not None  # only this rank's pairs are populated in `whole`
        frames = [int(f) for f in frames_per_pair] if self.partial_whole else [int(whole[b].F) for b in range(self.B)]
        gather = self.pair_collective == "allgather"
        self.frames = frames
        self.row_base, self.row_of_pair = pair_layout(frames, world, equal_slices=gather)
        self.shards = (capi.Problem * self.B)()   # this rank's pairs whole, the others empty (F == 0)
        for b in range(self.B):
            C.memmove(C.byref(self.shards[b]), C.byref(whole[b]), C.sizeof(capi.Problem))
            if b % world != rank:
                self.shards[b].F = 0
        self.first = np.zeros(self.B, np.int32)
        self.live = (capi.Problem * max(len(mine), 1))(*[whole[b] for b in mine])
        self.n_live = len(mine)
        self.nbf = sum(frames[b] for b in mine)
        self.nbf_whole = sum(frames)
        self.rows = self.row_base[-1]             # rows of the result buffer (>= nbf_whole with equal slices)
        assert self.row_base[rank + 1] - self.row_base[rank] >= self.nbf and self.rows >= self.nbf_whole
        self.sys_len = 0
        z = lambda n: torch.zeros(max(n, 1), dtype=torch.float64, device=device)
        lo = self.row_base[rank] * self.E
        if gather:
            # in place: every rank evaluates straight into its slice of the result buffer; the all-gather fills the others
            self.slice_rows = self.row_base[1] - self.row_base[0]
            self.reduced = z(self.rows * self.E)
            self.send = self.reduced
            self.count = self.rows * self.E       # doubles every rank holds afterwards ((N-1)/N of them cross the links)
        else:
            self.send = z(self.rows * self.E)     # zero outside this rank's slice, for good: nothing else writes there
            self.reduced = z(self.rows * self.E) if world > 1 else self.send  # one rank: the send buffer
            self.count = self.rows * self.E
        self.frame_blocks = self.send[lo:lo + max(self.nbf, 0) * self.E] if self.nbf else z(1)
        self.valid = z(self.nbf)
        self.systems = None
        self._ref_fb, self._ref_sys = (None if self.partial_whole else z(self.nbf_whole * self.E)), None
        # rows of the whole workload in problem order -> rows of the result buffer
        first_row = 0
        dst = np.zeros(self.nbf_whole, np.int64)
        for b in range(self.B):
            for f in range(frames[b]):
                dst[first_row + f] = self.row_of_pair[b] + f
            first_row += frames[b]
        self._dst_rows = torch.from_numpy(dst).to(device)

    def evaluate_local(self, with_hessian=True):
        lib, ctx = self.ctx.lib, self.ctx
        if self.mode == "frames" and with_hessian and self.n_live == self.B:
            # every problem has frames on this rank: evaluation and merge in one call (the finalize step stores the systems itself
            # where the shard is one frame on N == k knots -- bench.py's joint workload --, the merge kernel runs behind it otherwise)
            capi.check(lib.mbavo_eval_batch_merged(ctx.handle, self.B, self.live, self.k, self.frame_blocks.data_ptr(), self.systems.data_ptr(),
                                                   None, self.valid.data_ptr()), "mbavo_eval_batch_merged")
            return
        if self.n_live:
            capi.check(lib.mbavo_eval_batch(ctx.handle, self.n_live, self.live, self.k, 1 if with_hessian else 0,
                                            self.frame_blocks.data_ptr(), None, self.valid.data_ptr()), "mbavo_eval_batch")
        if self.mode == "frames":
            capi.check(lib.mbavo_merge_device(ctx.handle, self.B, self.shards, self.k, self.frame_blocks.data_ptr(),
                                              self.systems.data_ptr()), "mbavo_merge_device")

    def reduce(self):
        self._reduce()

    def step(self, with_hessian=True, reduce=True):
        self.evaluate_local(with_hessian)
        if reduce:
            self._reduce()

    def _reduce(self):
        if self.mode == "pairs" and self.pair_collective == "allgather":
            self.coll.allgather(self.reduced, self.slice_rows * self.E)
        elif self.mode == "frame_blocks" and self.fb_allgather:
            self.coll.allgather(self.reduced, self.nbf * self.E)
        elif self.mode in ("pairs", "frame_blocks"):
            self.coll.allreduce(self.send, self.reduced, self.count)
        else:
            self.coll.allreduce(self.reduced, self.reduced, self.count)

    def reference(self):
        """The reduced object of the WHOLE workload evaluated by this rank alone (synchronous; returns a clone)."""
        import torch
        lib, ctx = self.ctx.lib, self.ctx
        if self._ref_fb is None:
            raise RuntimeError("reference() needs the whole workload on this rank (frames_per_pair was given: weak scaling)")
        capi.check(lib.mbavo_eval_batch(ctx.handle, self.B, self.whole, self.k, 1, self._ref_fb.data_ptr(), None, None),
                   "mbavo_eval_batch")
        if self.mode == "frames":
            capi.check(lib.mbavo_merge_device(ctx.handle, self.B, self.whole, self.k, self._ref_fb.data_ptr(),
                                              self._ref_sys.data_ptr()), "mbavo_merge_device")
        torch.cuda.synchronize()
        if self.mode == "pairs":  # the same rows where the result buffer holds them (padding rows of equal slices stay zero)
            out = torch.zeros(self.rows, self.E, dtype=torch.float64, device=self._ref_fb.device)
            out[self._dst_rows] = self._ref_fb.view(self.nbf_whole, self.E)
            return out.reshape(-1)
        if self.mode == "frame_blocks":  # the same rows in the rank-major order of the reduced buffer
            return self._ref_fb.view(self.nbf_whole, self.E)[self._perm].reshape(-1).clone()
        return (self._ref_sys if self.mode == "frames" else self._ref_fb).clone()

    def blocks_of_pair(self, b):
        """'pairs' mode: the reduced packed frame blocks [F_b, E] of pair b (a view of `reduced`)."""
        r0 = self.row_of_pair[b]
        return self.reduced.view(self.rows, self.E)[r0:r0 + self.frames[b]]


LM_RECORD_SCALARS = 9  # initial cost, final cost, radius, iterations, accepted, rejected, invalid, outliers, control knots N


def lm_record_layout(num_pairs, world, max_N):
    """Layout of the all-gathered records of a pair-sharded batched LM: (rows_per_rank, record_len, row_of_pair).
    Rank r's pairs (b = r, r + world, ...) fill rows [r * rows_per_rank, ...) in ascending pair order -- equal slices, as
    ncclAllGather wants them; a record is [knots_t 3 max_N | knots_R 4 max_N | LM_RECORD_SCALARS] doubles."""
    rows = -(-num_pairs // world)
    return rows, 7 * max_N + LM_RECORD_SCALARS, [(b % world) * rows + b // world for b in range(num_pairs)]


class ShardedLmBatch:
    """B independent keyframe-pair alignments over the ranks, pair b on rank b % world ("independent
    keyframe-pair alignments shard naturally"): every rank runs the device-side LM loop (mbavo_lm_batch; what
    BlurAwareDirectTracker::optimizePyramidLevel does per pair, blur_aware_direct_tracker.cpp:590-924) over ITS pairs --
    there is no collective inside the loop, the pairs share nothing -- with the control knots living in this object's record
    buffer, and ONE in-place all-gather (mbavo_allgather_blocks) of the records at the end.

        whole      : capi.Problem array of all B pairs; only this rank's entries have to be populated
        init_knots : per pair (knots_t [N, 3], knots_R [N, 4]) numpy arrays (None for pairs of other ranks)
        run()      : reset the knots, mbavo_lm_batch on this rank's pairs, records of every rank gathered; returns the rc
        record(b)  : dict of pair b's result after run() (any rank)
    """

    def __init__(self, ctx, whole, k, rank, world, device, opts, init_knots, collective=None):
        import torch
        self.ctx, self.k, self.rank, self.world, self.opts = ctx, k, rank, world, opts
        self.coll = collective if collective is not None else RcclCollective(ctx)
        self.B = len(whole)
        self.mine = pairs_of_rank(self.B, rank, world)
        own = set(self.mine)
        self.N = [int(whole[b].N) if b in own else 0 for b in range(self.B)]  # (other ranks' N arrives with their records)
        self.max_N = 16  # the reference's max_num_ctrl_knots: the same record length on every rank without an exchange
        self.rows, self.rec, self.row_of_pair = lm_record_layout(self.B, world, self.max_N)
        self.records = torch.zeros(world * self.rows * self.rec, dtype=torch.float64, device=device)
        self.live = (capi.Problem * max(len(self.mine), 1))()
        init = np.zeros((self.rows, self.rec))
        base = self.records.data_ptr()
        for j, b in enumerate(self.mine):
            C.memmove(C.byref(self.live[j]), C.byref(whole[b]), C.sizeof(capi.Problem))
            row = self.row_of_pair[b]
            assert row == rank * self.rows + j
            kt, kR = init_knots[b]
            n = self.N[b]
            init[j, :3 * n] = np.asarray(kt, np.float64).ravel()
            init[j, 3 * self.max_N:3 * self.max_N + 4 * n] = np.asarray(kR, np.float64).ravel()
            self.live[j].d_knots_t = base + 8 * (row * self.rec)
            self.live[j].d_knots_R = base + 8 * (row * self.rec + 3 * self.max_N)
        self._init = torch.from_numpy(init.ravel()).pin_memory()
        self._scal = torch.zeros(self.rows, LM_RECORD_SCALARS, dtype=torch.float64).pin_memory()
        self.res = (capi.LmBatchResult * max(len(self.mine), 1))()
        lo = rank * self.rows * self.rec
        self._slice = self.records[lo:lo + self.rows * self.rec]
        self._scal_dst = self._slice.view(self.rows, self.rec)[:, 7 * self.max_N:]

    def run(self, gather=True):
        lib, ctx = self.ctx.lib, self.ctx
        with ctx_stream(self.ctx):  # (the LM loop and the all-gather run on the context's stream: so do the copies)
            self._slice.copy_(self._init, non_blocking=True)  # initial knots (the LM updates them in place), scalars zero
        rc = 0
        if self.mine:
            rc = lib.mbavo_lm_batch(ctx.handle, len(self.mine), self.live, C.byref(self.opts), self.res, None, 0)  # (returns synchronised)
            sc = self._scal.numpy()
            for j in range(len(self.mine)):
                r = self.res[j]
                sc[j] = (r.initial_cost, r.final_cost, r.radius, r.iterations, r.accepted, r.rejected, r.invalid, r.num_outliers,
                         self.N[self.mine[j]])
            with ctx_stream(self.ctx):
                self._scal_dst.copy_(self._scal, non_blocking=True)
        if gather and self.world > 1:
            self.coll.allgather(self.records, self.rows * self.rec)
        return rc

    def record(self, b, N=None):
        """Pair b's gathered record (synchronises): knots_t [N, 3], knots_R [N, 4] and the result scalars.  The number of
        control knots travels in the record (the pairs of other ranks are not populated in `whole`)."""
        import torch
        torch.cuda.synchronize()
        row = self.records.view(-1, self.rec)[self.row_of_pair[b]].cpu().numpy()
        s = row[7 * self.max_N:]
        n = N if N is not None else int(s[8])
        return {"knots_t": row[:3 * n].reshape(n, 3).copy(), "knots_R": row[3 * self.max_N:3 * self.max_N + 4 * n].reshape(n, 4).copy(),
                "initial_cost": float(s[0]), "final_cost": float(s[1]), "radius": float(s[2]), "iterations": int(s[3]),
                "accepted": int(s[4]), "rejected": int(s[5]), "invalid": int(s[6]), "num_outliers": int(s[7])}


def allreduce_blocks(blocks, group=None):
    """In-place sum over ranks of a tensor of packed blocks through torch.distributed (CPU / gloo tests)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(blocks, op=dist.ReduceOp.SUM, group=group)
    return blocks
