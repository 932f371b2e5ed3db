// p2p_comm.hip -- a collective that fits the message: the reduction point of the path,
// merge_hessian_gradient_cost (ba_tracker/merge_hessian_gradient_cost.cpp:39-86), exchanges 19 KB (one joint system) to 1.3 MB
// (512 packed blocks) per step -- latency-bound messages for which a ring collective pays one hop per rank.  Here every rank
// maps every peer's RECEIVE REGION (hipIpcGetMemHandle / hipIpcOpenMemHandle, handles exchanged by the caller like the RCCL
// communicator id), and a collective is ONE kernel on the evaluation's stream:
//
//   send     every workgroup stores its share of this rank's slice straight into slot [parity][rank] of every peer's region
//            (16-byte stores over xGMI, or through the local fabric when ranks share a GPU), makes them visible at system scope
//            and takes a ticket; the workgroup that completes a peer's tickets raises this rank's FLAG in that peer's region to
//            the step's sequence number (system-scope release);
//   wait     every workgroup polls its OWN region's flags (local memory, system-scope acquire loads) until all peers have
//            raised theirs for this step -- bounded: a peer that never arrives ends the kernel with a status, not a hang;
//   gather   all-gather: the slots are copied into the caller's buffer at their rank offsets; all-reduce: the N slots are
//            summed in RANK ORDER (own contribution in its place) -- fixed order, so every rank holds the same bits, and the
//            same bits as ncclAllReduce on ranks whose vectors are x + 0 + ... + 0 (pair sharding).
//
// Two parities of slots: a rank may run ahead of a slow peer by at most one collective (its step s + 1 flag is what lets the
// peer finish s + 1, and it is raised only after step s was gathered, stream order), so step s + 2 never overwrites a slot a
// peer is still reading.  The regions are allocated uncached / fine-grained so that a peer's stores are visible to the owner
// without an L2 invalidate of the pyramid.  No RCCL involved; RCCL stays the default until a multi-GPU box confirms the gain.
#include "../../include/mbavo.h"
#include "engine.h"

#include <cstdio>
#include <cstring>

#define P2P_TRY(expr)                                                                          \
    do                                                                                         \
    {                                                                                          \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
        {                                                                                      \
            fprintf(stderr, "mbavo: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return (int)e_;                                                                    \
        }                                                                                      \
    } while (0)

namespace mbavo
{
    typedef double d2_t __attribute__((ext_vector_type(2)));
    static constexpr int kP2PMaxWorld = 16;
    static constexpr int kP2PBlocks = 32;          // workgroups of a collective (co-resident: the wait phase needs no other block)
    static constexpr int kP2PThreads = 256;
    static constexpr long long kP2PTicksPerSecond = 100000000ll; // s_memrealtime: 100 MHz
    static constexpr long long kP2PSpinLimit = 20 * kP2PTicksPerSecond; // default wait for a peer: 20 s (mbavo_p2p_set_timeout)

    struct P2PState
    {
        int rank = 0, world = 0;
        size_t slot_bytes = 0, flags_off = 0, total = 0;
        char *local = nullptr;                 // this rank's region: [2][world] slots, then world flags (8 B each, 64 B apart)
        char *peer[kP2PMaxWorld] = {};         // every rank's region as mapped here (peer[rank] == local)
        bool opened[kP2PMaxWorld] = {};
        unsigned long long seq = 0;            // collectives enqueued so far
        long long spin_limit = kP2PSpinLimit;  // ticks a collective waits for a peer's flag before it gives up
        int *tickets = nullptr;                // device: world counters + 1 status word
        hipIpcMemHandle_t handle;
        bool connected = false;
    };

    struct P2PArgs
    {
        char *peer[kP2PMaxWorld];
        int rank, world;
        unsigned long long seq, slot_bytes, flags_off;
        long long spin_limit;
        int *tickets; // [world] tickets, [world] = status (1: a peer did not arrive in time)
    };

    __device__ __forceinline__ unsigned long long *p2p_flag(char *region, unsigned long long flags_off, int src)
    {
        return reinterpret_cast<unsigned long long *>(region + flags_off + (size_t)src * 64);
    }

    // MODE 0: all-gather in place (buf holds world slices of `count` doubles, this rank's at its rank offset);
    // MODE 1: all-reduce in place over `count` doubles.
    template <int MODE>
    __global__ __launch_bounds__(kP2PThreads) void k_p2p_collective(P2PArgs a, double *__restrict__ buf, long long count)
    {
        const int nb = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
        const int parity = (int)(a.seq & 1ull);
        const double *mine = MODE == 0 ? buf + (size_t)a.rank * (size_t)count : buf;
        // ---- send: (peer, chunk) work items dealt over the workgroups; 16-byte stores when the slice allows it
        const long long pairs = count >> 1; // double2 elements
        for (int p = 0; p < a.world; ++p)
        {
            if (p == a.rank) continue;
            double *dst = reinterpret_cast<double *>(a.peer[p] + ((size_t)parity * a.world + a.rank) * a.slot_bytes);
            if ((reinterpret_cast<size_t>(mine) & 15) == 0)
            {
                const d2_t *s2 = reinterpret_cast<const d2_t *>(mine);
                d2_t *d2 = reinterpret_cast<d2_t *>(dst);
                for (long long i = (long long)b * kP2PThreads + tid; i < pairs; i += (long long)nb * kP2PThreads)
                    __builtin_nontemporal_store(s2[i], d2 + i);
                if ((count & 1) && b == 0 && tid == 0) dst[count - 1] = mine[count - 1];
            }
            else
                for (long long i = (long long)b * kP2PThreads + tid; i < count; i += (long long)nb * kP2PThreads) dst[i] = mine[i];
        }
        // every wave's stores have left the CU, then ONE thread per workgroup publishes them system-wide and takes the tickets
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0)
        {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, ""); // system scope
            for (int p = 0; p < a.world; ++p)
            {
                if (p == a.rank) continue;
                const int t = __hip_atomic_fetch_add(a.tickets + p, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                if (t == nb - 1)
                { // every workgroup's share has reached peer p: raise this rank's flag there
                    __hip_atomic_store(a.tickets + p, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(p2p_flag(a.peer[p], a.flags_off, a.rank), a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
        // ---- wait: lane p of wave 0 watches peer p's flag in OUR region
        __shared__ int s_ok;
        if (tid == 0) s_ok = 1;
        __syncthreads();
        if (tid < a.world && tid != a.rank)
        {
            unsigned long long *f = p2p_flag(a.peer[a.rank], a.flags_off, tid);
            const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
            while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < a.seq)
            {
                __builtin_amdgcn_s_sleep(2);
                if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > a.spin_limit)
                {
                    s_ok = 0;
                    __hip_atomic_store(a.tickets + a.world, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        __syncthreads();
        if (!s_ok)
        { // A peer never arrived.  The caller's buffer must not pass for a result (the call itself returned long ago): what this
          // workgroup would have written -- the peers' slices of an all-gather, its share of an all-reduce, in the send phase's own
          // partition (see the gather below) -- becomes NaN; mbavo_p2p_status reports MBAVO_E_TIMEOUT.
            const double nan = __builtin_nan("");
            if (MODE == 0)
            {
                for (int p = 0; p < a.world; ++p)
                    if (p != a.rank)
                        for (long long i = (long long)b * kP2PThreads + tid; i < count; i += (long long)nb * kP2PThreads) buf[(size_t)p * (size_t)count + i] = nan;
            }
            else if ((reinterpret_cast<size_t>(mine) & 15) == 0)
            {
                for (long long i = (long long)b * kP2PThreads + tid; i < pairs; i += (long long)nb * kP2PThreads) { buf[2 * i] = nan; buf[2 * i + 1] = nan; }
                if ((count & 1) && b == 0 && tid == 0) buf[count - 1] = nan;
            }
            else
                for (long long i = (long long)b * kP2PThreads + tid; i < count; i += (long long)nb * kP2PThreads) buf[i] = nan;
            return;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, ""); // the peers' slot stores are visible to every thread of this workgroup
        // ---- gather
        const char *base = a.peer[a.rank] + (size_t)parity * a.world * a.slot_bytes;
        if (MODE == 0)
        {
            for (int p = 0; p < a.world; ++p)
            {
                if (p == a.rank) continue;
                const double *src = reinterpret_cast<const double *>(base + (size_t)p * a.slot_bytes);
                double *dst = buf + (size_t)p * (size_t)count;
                for (long long i = (long long)b * kP2PThreads + tid; i < count; i += (long long)nb * kP2PThreads)
                    dst[i] = __builtin_nontemporal_load(src + i);
            }
        }
        else
        {
            // In place: an element may only be overwritten by the thread that SENT it (another workgroup of this kernel may still be in
            // its send phase: nothing orders the workgroups of one rank against each other) -- so the sums walk the elements in the
            // send phase's own partition: 16-byte pairs where the sends were pairs, the odd tail by the thread that sent it.
            auto reduce_at = [&](long long i) {
                double acc = 0.0;
                for (int p = 0; p < a.world; ++p) // rank order, own contribution in its place: the same bits on every rank
                    acc += p == a.rank ? buf[i] : __builtin_nontemporal_load(reinterpret_cast<const double *>(base + (size_t)p * a.slot_bytes) + i);
                buf[i] = acc;
            };
            if ((reinterpret_cast<size_t>(mine) & 15) == 0)
            {
                for (long long i = (long long)b * kP2PThreads + tid; i < pairs; i += (long long)nb * kP2PThreads) { reduce_at(2 * i); reduce_at(2 * i + 1); }
                if ((count & 1) && b == 0 && tid == 0) reduce_at(count - 1);
            }
            else
                for (long long i = (long long)b * kP2PThreads + tid; i < count; i += (long long)nb * kP2PThreads) reduce_at(i);
        }
    }

    int Engine::p2p_create(int rank, int world, long long max_doubles_per_slot, unsigned char *handle_out)
    {
        if (p2p_ || world < 1 || world > kP2PMaxWorld || rank < 0 || rank >= world || max_doubles_per_slot < 1 || !handle_out) return MBAVO_E_ARG;
        hipError_t e = hipSetDevice(device_);
        if (e != hipSuccess) return (int)e;
        P2PState *s = new P2PState;
        s->rank = rank; s->world = world;
        s->slot_bytes = (((size_t)max_doubles_per_slot * sizeof(double)) + 255) & ~(size_t)255;
        s->flags_off = 2 * (size_t)world * s->slot_bytes;
        s->total = s->flags_off + (size_t)world * 64;
        void *p = nullptr;
        // a peer's stores must be visible to the owner's loads without invalidating its L2: uncached, else fine-grained
        if (hipExtMallocWithFlags(&p, s->total, hipDeviceMallocUncached) != hipSuccess)
        {
            (void)hipGetLastError();
            if (hipExtMallocWithFlags(&p, s->total, hipDeviceMallocFinegrained) != hipSuccess)
            {
                (void)hipGetLastError();
                delete s;
                return (int)hipErrorOutOfMemory;
            }
        }
        s->local = (char *)p;
        e = hipMemset(p, 0, s->total);
        if (e == hipSuccess) e = hipMalloc((void **)&s->tickets, sizeof(int) * (world + 1));
        if (e == hipSuccess) e = hipMemset(s->tickets, 0, sizeof(int) * (world + 1));
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e == hipSuccess) e = hipIpcGetMemHandle(&s->handle, p);
        if (e != hipSuccess)
        {
            fprintf(stderr, "mbavo: p2p_create (rank %d of %d, %zu bytes): %s\n", rank, world, s->total, hipGetErrorString(e));
            (void)hipFree(p); (void)hipFree(s->tickets);
            delete s;
            return (int)e;
        }
        static_assert(sizeof(hipIpcMemHandle_t) <= MBAVO_P2P_HANDLE_BYTES, "handle size");
        memset(handle_out, 0, MBAVO_P2P_HANDLE_BYTES);
        memcpy(handle_out, &s->handle, sizeof(hipIpcMemHandle_t));
        p2p_ = s;
        return 0;
    }

    int Engine::p2p_connect(const unsigned char *all_handles)
    {
        P2PState *s = p2p_;
        if (!s || s->connected || !all_handles) return MBAVO_E_ARG;
        hipError_t e = hipSetDevice(device_);
        if (e != hipSuccess) return (int)e;
        for (int p = 0; p < s->world; ++p)
        {
            if (p == s->rank) { s->peer[p] = s->local; continue; }
            hipIpcMemHandle_t h;
            memcpy(&h, all_handles + (size_t)p * MBAVO_P2P_HANDLE_BYTES, sizeof(h));
            void *m = nullptr;
            e = hipIpcOpenMemHandle(&m, h, hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess)
            {
                fprintf(stderr, "mbavo: p2p_connect: hipIpcOpenMemHandle(rank %d): %s\n", p, hipGetErrorString(e));
                for (int q = 0; q < p; ++q) // the mappings made so far are released: a retry starts from nothing
                    if (s->opened[q]) { (void)hipIpcCloseMemHandle(s->peer[q]); s->opened[q] = false; s->peer[q] = nullptr; }
                return (int)e;
            }
            s->peer[p] = (char *)m;
            s->opened[p] = true;
        }
        s->connected = true;
        return 0;
    }

    int Engine::p2p_ranks() const { return p2p_ && p2p_->connected ? p2p_->world : 0; }

    int Engine::p2p_collective(int mode, double *d_buf, long long count)
    {
        P2PState *s = p2p_;
        if (!s || !s->connected || !d_buf || count < 0 || (mode != 0 && mode != 1)) return MBAVO_E_ARG;
        if ((size_t)count * sizeof(double) > s->slot_bytes) return MBAVO_E_ARG; // the slots were sized at p2p_create
        if (count == 0 || s->world == 1) return 0; // one rank: its slice is the buffer, its vector the sum
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess || cur != device_) P2P_TRY(hipSetDevice(device_));
        P2PArgs a;
        memset(&a, 0, sizeof(a));
        for (int p = 0; p < s->world; ++p) a.peer[p] = s->peer[p];
        a.rank = s->rank; a.world = s->world; a.seq = ++s->seq; a.slot_bytes = s->slot_bytes; a.flags_off = s->flags_off;
        a.tickets = s->tickets;
        a.spin_limit = s->spin_limit;
        // small messages: fewer workgroups (every one of them polls and fences); 19 KB is one workgroup's work
        long long want = (count * (long long)sizeof(double) + 16383) / 16384;
        const int nb = (int)(want < 1 ? 1 : (want > kP2PBlocks ? kP2PBlocks : want));
        if (mode == 0)
            hipLaunchKernelGGL((k_p2p_collective<0>), dim3(nb), dim3(kP2PThreads), 0, stream_, a, d_buf, count);
        else
            hipLaunchKernelGGL((k_p2p_collective<1>), dim3(nb), dim3(kP2PThreads), 0, stream_, a, d_buf, count);
        return (int)hipGetLastError();
    }

    int Engine::p2p_set_timeout(double seconds)
    { // how long a collective waits for a peer before it gives up (default 20 s); applies to the collectives enqueued from here on
        P2PState *s = p2p_;
        if (!s || !(seconds > 0.0) || seconds > 3600.0) return MBAVO_E_ARG;
        s->spin_limit = (long long)(seconds * (double)kP2PTicksPerSecond);
        return 0;
    }

    int Engine::p2p_status()
    { // 0, or MBAVO_E_TIMEOUT when a collective gave up on a peer (synchronises the stream).  STICKY: after a timeout the ranks'
      // sequence numbers no longer agree and the regions must be torn down (disconnect + destroy) and created again
        P2PState *s = p2p_;
        if (!s) return MBAVO_E_ARG;
        int st = 0;
        P2P_TRY(hipStreamSynchronize(stream_));
        P2P_TRY(hipMemcpy(&st, s->tickets + s->world, sizeof(int), hipMemcpyDeviceToHost));
        return st ? MBAVO_E_TIMEOUT : 0;
    }

    int Engine::p2p_disconnect()
    { // unmap the peers' regions (collectives are refused from here on); the own region stays until p2p_destroy
        P2PState *s = p2p_;
        if (!s) return 0;
        (void)hipStreamSynchronize(stream_);
        for (int p = 0; p < s->world; ++p)
            if (s->opened[p]) { (void)hipIpcCloseMemHandle(s->peer[p]); s->opened[p] = false; s->peer[p] = nullptr; }
        s->connected = false;
        return 0;
    }

    int Engine::p2p_destroy()
    {
        P2PState *s = p2p_;
        if (!s) return 0;
        (void)hipStreamSynchronize(stream_);
        for (int p = 0; p < s->world; ++p)
            if (s->opened[p]) (void)hipIpcCloseMemHandle(s->peer[p]);
        (void)hipFree(s->local);
        (void)hipFree(s->tickets);
        delete s;
        p2p_ = nullptr;
        return 0;
    }
} // namespace mbavo
