// multi_gpu.hip -- the multi-GPU side of the path: one process per GPU, every rank evaluates a shard
// (keypoint band or frame range) of a problem with the fused engine, scatters its packed frame blocks into the
// 6N x 6N normal equations ON THE DEVICE and the ranks' partial systems are summed with one RCCL all-reduce over xGMI.
// The reference's reduction point is merge_hessian_gradient_cost (ba_tracker/merge_hessian_gradient_cost.cpp:39-86,
// called at spline_update_step.cpp:232-239); this file is its device twin plus the sharding and the communicator.
// RCCL is bound at run time (dlopen) so that the library loads, and everything single-GPU works, on hosts without it.
#include "../../include/mbavo.h"
#include "engine.h"
#include "pixel_math.h"

#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <vector>

namespace mbavo
{
    // ------------------------------------------------------------------ RCCL, resolved at run time
    namespace
    {
        struct Rccl
        {
            typedef struct { char internal[MBAVO_COMM_ID_BYTES]; } UniqueId; // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES 128)
            int (*GetUniqueId)(UniqueId *) = nullptr;
            int (*CommInitRank)(void **, int, UniqueId, int) = nullptr;
            int (*CommCount)(const void *, int *) = nullptr;
            int (*CommDestroy)(void *) = nullptr;
            int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
            int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
            const char *(*GetErrorString)(int) = nullptr;
            bool ok = false;
        };

        const Rccl &rccl()
        {
            static Rccl r;
            static bool tried = false;
            if (tried) return r;
            tried = true;
            // the SONAME first: binds to the RCCL instance the process already uses (e.g. the one PyTorch ships), so a
            // caller-owned ncclComm_t and the library's own communicator live in the same RCCL
            void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (!h)
            {
                fprintf(stderr, "mbavo: cannot load librccl.so: %s\n", dlerror());
                return r;
            }
            r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
            r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
            r.CommCount = (decltype(r.CommCount))dlsym(h, "ncclCommCount");
            r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
            r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
            r.AllGather = (decltype(r.AllGather))dlsym(h, "ncclAllGather");
            r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
            r.ok = r.GetUniqueId && r.CommInitRank && r.CommCount && r.CommDestroy && r.AllReduce;
            if (!r.ok) fprintf(stderr, "mbavo: librccl.so lacks an expected symbol\n");
            return r;
        }

        int rccl_rc(int rc, const char *what)
        {
            if (rc == 0) return 0;
            const Rccl &r = rccl();
            fprintf(stderr, "mbavo: %s failed: %s (ncclResult_t %d)\n", what, r.GetErrorString ? r.GetErrorString(rc) : "?", rc);
            return -2000 - rc;
        }
    } // namespace

    int comm_unique_id(unsigned char *id)
    {
        const Rccl &r = rccl();
        if (!r.ok) return MBAVO_E_NODEVICE;
        Rccl::UniqueId u;
        const int rc = r.GetUniqueId(&u);
        if (rc) return rccl_rc(rc, "ncclGetUniqueId");
        memcpy(id, u.internal, MBAVO_COMM_ID_BYTES);
        return 0;
    }

    int Engine::comm_init(const unsigned char *id, int rank, int world)
    {
        const Rccl &r = rccl();
        if (!r.ok) return MBAVO_E_NODEVICE;
        if (comm_) return MBAVO_E_ARG; // one communicator per context
        hipError_t e = hipSetDevice(device_);
        if (e != hipSuccess) return (int)e;
        Rccl::UniqueId u;
        memcpy(u.internal, id, MBAVO_COMM_ID_BYTES);
        void *c = nullptr;
        const int rc = r.CommInitRank(&c, world, u, rank);
        if (rc) return rccl_rc(rc, "ncclCommInitRank");
        comm_ = c;
        return 0;
    }

    int Engine::comm_ranks() const
    {
        if (!comm_) return 0;
        int n = 0;
        return rccl().CommCount(comm_, &n) == 0 ? n : 0;
    }

    int Engine::comm_destroy()
    {
        if (!comm_) return 0;
        (void)hipStreamSynchronize(stream_); // no collective of ours may still be in flight
        const int rc = rccl().CommDestroy(comm_);
        comm_ = nullptr;
        return rccl_rc(rc, "ncclCommDestroy");
    }

    int Engine::allreduce(void *comm, const double *send, double *recv, long long count)
    {
        const Rccl &r = rccl();
        if (!r.ok) return MBAVO_E_NODEVICE;
        void *c = comm ? comm : comm_;
        if (!c || !send || !recv || count < 0) return MBAVO_E_ARG;
        if (count == 0) return 0;
        int nranks = 0;
        if (r.CommCount(c, &nranks) == 0 && nranks == 1)
        { // a communicator of one rank has nothing to add: no collective kernel (~4.5 us in the stream), at most a copy
            if (send == recv) return 0;
            return (int)hipMemcpyAsync(recv, send, (size_t)count * sizeof(double), hipMemcpyDeviceToDevice, stream_);
        }
        // ncclDouble = 8, ncclSum = 0 (rccl.h); on the stream the evaluation and the merge were enqueued on
        return rccl_rc(r.AllReduce(send, recv, (size_t)count, 8, 0, c, stream_), "ncclAllReduce");
    }

    int Engine::allgather(void *comm, double *d, long long count_per_rank)
    {
        const Rccl &r = rccl();
        if (!r.ok || !r.AllGather) return MBAVO_E_NODEVICE;
        void *c = comm ? comm : comm_;
        if (!c || !d || count_per_rank < 0) return MBAVO_E_ARG;
        if (count_per_rank == 0) return 0;
        int nranks = 0;
        if (r.CommCount(c, &nranks) == 0 && nranks == 1) return 0; // one rank, in place: its slice IS the buffer (no collective kernel)
        int rank = 0;
        { // in place: this rank's slice sits at its rank offset of the receive buffer
            static int (*UserRank)(const void *, int *) = nullptr;
            if (!UserRank)
            {
                void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
                if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
                UserRank = h ? (decltype(UserRank))dlsym(h, "ncclCommUserRank") : nullptr;
            }
            if (!UserRank) return MBAVO_E_NODEVICE;
            const int rc = UserRank(c, &rank);
            if (rc) return rccl_rc(rc, "ncclCommUserRank");
        }
        return rccl_rc(r.AllGather(d + (size_t)rank * (size_t)count_per_rank, d, (size_t)count_per_rank, 8, c, stream_), "ncclAllGather");
    }

    // ------------------------------------------------------------------ shards (host pointer arithmetic only)
    static long long residuals_of(const mbavo_problem &p)
    {
        return p.num_residuals > 0 ? p.num_residuals : (long long)(p.K - p.num_bad) * p.F * p.P;
    }

    int shard_keypoints(const mbavo_problem *whole, int rank, int world, mbavo_problem *out, int *first)
    {
        if (!whole || !out || world < 1 || rank < 0 || rank >= world || whole->K < 0) return MBAVO_E_ARG;
        const long long K = whole->K;
        const int lo = (int)(K * rank / world), hi = (int)(K * (rank + 1) / world);
        mbavo_problem s = *whole;
        s.K = hi - lo;
        s.d_kp_xy = whole->d_kp_xy ? whole->d_kp_xy + (size_t)lo * whole->kp_stride : nullptr;
        s.d_kp_z = whole->d_kp_z ? whole->d_kp_z + lo : nullptr;
        s.d_outlier = whole->d_outlier ? whole->d_outlier + lo : nullptr;
        s.num_residuals = residuals_of(*whole); // outliers of the whole problem included: flagged keypoints stay flagged
        s.num_bad = 0;                          // (unused once num_residuals is set)
        *out = s;
        if (first) *first = lo;
        return 0;
    }

    int shard_frames(const mbavo_problem *whole, int rank, int world, mbavo_problem *out, int *first)
    {
        if (!whole || !out || world < 1 || rank < 0 || rank >= world || whole->F < 0) return MBAVO_E_ARG;
        const long long F = whole->F;
        const int lo = (int)(F * rank / world), hi = (int)(F * (rank + 1) / world);
        mbavo_problem s = *whole;
        s.F = hi - lo;
        s.d_cur_imgs = whole->d_cur_imgs ? whole->d_cur_imgs + lo : nullptr;
        s.d_cap_time = whole->d_cap_time ? whole->d_cap_time + lo : nullptr;
        s.d_exp_time = whole->d_exp_time ? whole->d_exp_time + lo : nullptr;
        s.h_start_idx = whole->h_start_idx ? whole->h_start_idx + lo : nullptr;
        s.num_residuals = residuals_of(*whole);
        *out = s;
        if (first) *first = lo;
        return 0;
    }

    // ------------------------------------------------------------------ merge on the device
    struct MergeDesc
    {
        int F, N, bf_base, start_base;
        long long sys_base;
    };

    // One block per (problem, chunk of 256 system entries); entry 0 = cost, 1 .. 6N = g, then H column-major.  Gather
    // form: every output entry walks the problem's frames in ascending order and adds the packed entry that lands on
    // it, if any (merge_hessian_gradient_cost.cpp:52-62 local -> global index map, inverted) -- no atomics.
    template <int KD>
    __global__ __launch_bounds__(256) void k_merge(const MergeDesc *__restrict__ descs, const int *__restrict__ start_idx,
                                                   const double *__restrict__ fb, double *__restrict__ systems)
    {
        constexpr int M3 = 3 * KD, ND = 6 * KD + 1, E = ND * (ND + 1) / 2;
        const MergeDesc d = descs[blockIdx.y];
        const int n = 6 * d.N, len = 1 + n + n * n;
        const int o = blockIdx.x * 256 + threadIdx.x;
        if (o >= len) return;
        // global unknown index -> (knot, offset inside the local row [t-knots (3k) | w-knots (3k)] for start index 0)
        auto local_of = [&](int G, int st) -> int {
            const bool rot = G >= 3 * d.N;
            const int gi = rot ? G - 3 * d.N : G;
            const int j = gi - 3 * st;
            if (j < 0 || j >= M3) return -1;
            return rot ? M3 + j : j;
        };
        double acc = 0.0;
        for (int f = 0; f < d.F; ++f)
        {
            const double *blk = fb + (size_t)(d.bf_base + f) * E;
            const int st = start_idx[d.start_base + f];
            if (o == 0) acc += blk[0];
            else if (o <= n)
            {
                const int j = local_of(o - 1, st);
                if (j >= 0) acc += blk[1 + j];
            }
            else
            {
                const int h = o - 1 - n, C = h / n, R = h - C * n;
                const int r = local_of(R, st), c = local_of(C, st);
                if (r >= 0 && c >= 0)
                {
                    const int i = r < c ? r : c, j = r < c ? c : r;
                    // packed (i, j), i <= j, of the (6k+1) x (6k+1) triangle whose row / column 0 is the residual
                    const int ii = i + 1, jj = j + 1;
                    acc += blk[ii * ND - ii * (ii - 1) / 2 + (jj - ii)];
                }
            }
        }
        systems[d.sys_base + o] = acc;
    }

    int Engine::merge_device(int B, const mbavo_problem *probs, int kdeg, const double *d_fb, double *d_systems)
    {
        if (B < 1 || !probs || !d_fb || !d_systems || (kdeg != 2 && kdeg != 4)) return MBAVO_E_ARG;
        std::vector<MergeDesc> descs((size_t)B);
        std::vector<int> start;
        int bf = 0, max_len = 0;
        long long sys = 0;
        for (int b = 0; b < B; ++b)
        {
            const mbavo_problem &p = probs[b];
            if (p.F < 0 || p.N < kdeg || (p.F > 0 && !p.h_start_idx)) return MBAVO_E_ARG;
            MergeDesc &d = descs[b];
            d.F = p.F; d.N = p.N; d.bf_base = bf; d.start_base = (int)start.size(); d.sys_base = sys;
            for (int f = 0; f < p.F; ++f)
            {
                if (p.h_start_idx[f] < 0 || p.h_start_idx[f] + kdeg > p.N) return MBAVO_E_RANGE;
                start.push_back(p.h_start_idx[f]);
            }
            const int n = 6 * p.N, len = 1 + n + n * n;
            max_len = len > max_len ? len : max_len;
            bf += p.F;
            sys += len;
        }
        hipError_t e = hipSuccess;
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess || cur != device_) e = hipSetDevice(device_); // ~5 us per call on this runtime
        if (e != hipSuccess) return (int)e;
        // descriptors and start indices are re-uploaded only when they change (pageable source: the runtime stages the
        // copy before returning, so the vectors may go away)
        const size_t db = descs.size() * sizeof(MergeDesc), sb = (start.size() + 1) * sizeof(int);
        const bool same = kdeg == merge_kdeg_ && merge_descs_.size() == db && merge_start_ == start &&
                          memcmp(merge_descs_.data(), descs.data(), db) == 0;
        void *d_desc = named_scratch(7, db), *d_start = named_scratch(11, sb);
        if (!d_desc || !d_start) return (int)hipErrorOutOfMemory;
        if (!same || d_desc != merge_dev_[0] || d_start != merge_dev_[1])
        {
            if ((e = hipMemcpyAsync(d_desc, descs.data(), db, hipMemcpyHostToDevice, stream_)) != hipSuccess) return (int)e;
            if (!start.empty() && (e = hipMemcpyAsync(d_start, start.data(), start.size() * sizeof(int), hipMemcpyHostToDevice, stream_)) != hipSuccess) return (int)e;
            merge_descs_.assign((const char *)descs.data(), (const char *)descs.data() + db);
            merge_start_ = start;
            merge_kdeg_ = kdeg;
            merge_dev_[0] = d_desc; merge_dev_[1] = d_start;
        }
        const dim3 grid((max_len + 255) / 256, B);
        if (kdeg == 4)
            hipLaunchKernelGGL((k_merge<4>), grid, dim3(256), 0, stream_, (const MergeDesc *)d_desc, (const int *)d_start, d_fb, d_systems);
        else
            hipLaunchKernelGGL((k_merge<2>), grid, dim3(256), 0, stream_, (const MergeDesc *)d_desc, (const int *)d_start, d_fb, d_systems);
        return (int)hipGetLastError();
    }
} // namespace mbavo
