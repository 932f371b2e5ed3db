// engine.hip -- fused batch evaluation kernels for gfx950 (MI355X) and their host driver.
//
// Work mapping:
//   * one lane = one pixel (keypoint x pattern entry); the S blur samples run
//     sequentially in registers, so the reference's S-wide shared-memory
//     reductions and its per-sample Jacobian scratch do not exist here;
//   * the per-sample pose table is read with wave-uniform addresses (scalar loads); where every
//     tile has a CU to itself the workgroup computes its frame's entries itself (k_fused<.., POSE>:
//     no pose launch);
//   * each wave parks the weighted rows [r | J] of its own 64 pixels in a private LDS
//     slab ([pixel][entry]) and feeds them back to the matrix core as BOTH operands of
//     v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 blocks per instruction): the row is
//     cut into groups of four entries and every unordered pair of groups is one block slot
//     (OuterAcc below; k = 4: 7 instructions per four pixels -- the 28 pairs of 7 groups as four
//     closed trails, one per block of the instruction -- and 14 accumulator VGPRs).
//     On gfx950 the f64 MFMA issues at the FP64 VALU rate and SHARES that pipe
//     (tools/micro/mfma_valu_overlap.hip), so it is not a free second engine: what it buys
//     over per-lane VALU accumulators is registers and instruction count (no cross-wave
//     barrier, ~95 fewer VGPRs).  (The padded 16x16x4 tiles of round 1: history, profiles/r02_kfused_experiments.txt.)
//   * accumulators live across the whole tile loop and are reduced once per
//     workgroup in a fixed order -- no atomics.
#include "engine.h"
#include <hip/hip_ext.h>
#include "pixel_math.h"
#include "se3_math.h"
#include "timing.h"

#if defined(MBAVO_FUSED_STAMPS) // timing experiment (tools/fused_stamps.py): where a workgroup of k_fused spends its time
namespace mbavo
{
    __device__ unsigned long long g_fused_stamps[2048 * 8];
    __device__ unsigned long long g_wave_stamps[1024 * 16 * 4]; // [block][wave][loop start, loop end, HW_ID, rounds]
}
#define MBAVO_WSTAMP(i, v) do { if (lane == 0 && blockIdx.x < 1024) g_wave_stamps[(blockIdx.x * 16 + wave) * 4 + (i)] = (v); } while (0)
#if defined(MBAVO_POSE_STAMPS) // (tools/pose_stamps.py) the pose prologue's steps instead of the kernel's phases: 1 descriptors read, 2 stage A done, 3 stage B done, 4 visible + scalar cache invalidated, 5 ready
#define MBAVO_PSTAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 2048) g_fused_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define MBAVO_FSTAMP(i) do { if ((i) == 0 && threadIdx.x == 0 && blockIdx.x < 2048) g_fused_stamps[blockIdx.x * 8] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MBAVO_PSTAMP(i) do { } while (0)
#define MBAVO_FSTAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 2048) g_fused_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#endif
#else
#define MBAVO_PSTAMP(i) do { } while (0)
#define MBAVO_FSTAMP(i) do { } while (0)
#define MBAVO_WSTAMP(i, v) do { } while (0)
#endif
#include "pose_entries.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>

namespace mbavo
{
#define HIP_TRY(expr)                                                                       \
    do                                                                                      \
    {                                                                                       \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
        {                                                                                   \
            fprintf(stderr, "mbavo: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), \
                    __FILE__, __LINE__);                                                    \
            return (int)e_;                                                                 \
        }                                                                                   \
    } while (0)

#ifndef MBAVO_WAVES_PER_GROUP
#define MBAVO_WAVES_PER_GROUP 12
#endif
    // Waves per workgroup of the fused kernel, per instantiation: one workgroup is resident per CU, so this is the
    // occupancy.  k = 4 with Jacobians: 168 VGPRs (the budget of 3 waves per SIMD = 12 per CU is 170; 16 waves spill
    // vectors) and 44 scalar spills into VGPR lanes, none of them inside the sample-pair loop; k = 2 (124 VGPRs) and
    // the cost-only kernels (63; 117 with the pose prologue) take the 16 waves a workgroup can have.  Figures: profiles/r02_kernel_resources.txt
    // (tools/kernel_resources.py, from the code-object metadata).
#ifndef MBAVO_WAVES_K2
#define MBAVO_WAVES_K2 16
#endif
    template <int KD, bool WITH_J>
    constexpr int waves_of() { return WITH_J ? (KD == 4 ? MBAVO_WAVES_PER_GROUP : MBAVO_WAVES_K2) : 16; }

    template <int KD>
    struct Pack
    {
        static constexpr int ND = 6 * KD + 1;
        static constexpr int E = ND * (ND + 1) / 2;
        static constexpr int PSTRIDE = E + 2; // partial: [nvalid | g,H sums (1..E-1) | cost | spare]
    };

    // ------------------------------------------------------------------ pose table (device code: pose_entries.h)
    // grid = ceil(entries / kPoseSPB), block = KD waves
    template <int KD, bool WITH_J>
    __global__ __launch_bounds__(64 * KD) void k_pose_table(const ProblemDesc *__restrict__ descs, const int *__restrict__ entry_prob,
                                                            int total_entries, PoseEntry<KD> *__restrict__ table,
                                                            int *__restrict__ status)
    {
        constexpr int NCOL = WITH_J ? 3 : 1, NSEG = KD - 1;
        __shared__ SplineSeg segs[kPoseSPB][NSEG];
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
        const int e0 = blockIdx.x * kPoseSPB;
        // stage A: wave g evaluates segment g of sample `lane`
        if (wave < NSEG && lane < kPoseSPB && e0 + lane < total_entries)
        {
            const int gid = e0 + lane;
            const ProblemDesc &d = descs[entry_prob[gid]]; // one load instead of a search over the problems' pose_base
            const int local = gid - d.pose_base;
            const int f = local / d.S;
            int idx;
            double u;
            bool oob;
            pose_sample_segment<KD>(d, f, local - f * d.S, idx, u, oob);
            spline_segment_eval<WITH_J>(d.knots_R + 4 * (idx + wave), d.knots_R + 4 * (idx + wave + 1), seg_weight<KD>(u, wave),
                                        segs[lane][wave]);
        }
        __syncthreads();
        // stage B: wave = knot, lane = (sample, column); cost-only: wave 0, one lane per sample
        const int sl = lane / NCOL, col = lane - sl * NCOL, gid = e0 + sl;
        if (sl >= kPoseSPB || gid >= total_entries || (!WITH_J && wave != 0)) return;
        const ProblemDesc &d = descs[entry_prob[gid]];
        const int local = gid - d.pose_base;
        const int f = local / d.S;
        int idx;
        double u;
        bool oob;
        pose_sample_segment<KD>(d, f, local - f * d.S, idx, u, oob);
        if (oob && wave == 0 && col == 0) atomicAdd(status, 1);
        pose_stage_b<KD, WITH_J>(d.knots_t, d.knots_R, idx, u, wave, col, segs[sl], table[gid]);
    }

    // ------------------------------------------------------------------ fused kernel
    typedef double f64x4 __attribute__((ext_vector_type(4)));

    // Per-wave outer product rows^T * rows on v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 blocks per instruction,
    // ~17 cycles).  The row of ND = 6k + 1 entries is cut into G groups of four (25 -> 7 groups, 13 -> 4; the padding
    // needs no zeros: entry (i, j) of a block depends only on A's row i and B's column j, so whatever a padded lane
    // reads -- the next row's first entries -- only reaches padded outputs, which are never gathered).
    // Lane = 16 * pixel + 4 * d + e holds W[m] = row[4 * ((m + d) % G) + e]; instruction (delta, h) takes A = W[4h],
    // B = W[(4h + delta) % G], so its block d is group (4h + d) % G times group (4h + d + delta) % G.  delta = 0 .. G/2
    // and h = 0 .. ceil(G/4) - 1 cover every unordered pair of groups (a few block slots repeat a pair and are
    // ignored): k = 4: 8 instructions (~140 cycles) per four pixels against 3 x 66 for the padded 16x16x4 tiles
    // (round 1's scheme), for 7 LDS reads per step instead of 2; 16 accumulator VGPRs instead
    // of 24.  The MFMA shares the FP64 pipe with the VALU, so instructions saved here are kernel time saved.
    // (Layout probed in tools/micro/mfma4_probe.hip and mfma4_outer_probe.hip; the cbsz / abid broadcast controls
    // are ignored by this instruction, which rules out a 7-instruction scheme with A broadcast from block 0.)
    // G = 7 (k = 4: 25 entries): SEVEN instructions instead of eight.  The 28 unordered pairs of 7 groups (self-pairs
    // included) are the edges of K7 with a loop at every vertex -- every degree is 8, even -- and that graph splits into
    // four CLOSED TRAILS of length 7 (found by exhaustive search, tools/mfma_trails.py).  Block d walks trail d: the lane
    // loads W[m] = group kTrail7[d][m] (seven LDS reads, as before; only the lane's read offsets differ) and instruction
    // m multiplies W[m] by W[m + 1 mod 7] -- the same two registers for every lane, a different pair of groups in every
    // block, every pair exactly once in the 28 block slots.  One MFMA in eight and two accumulator VGPRs saved.
    constexpr int kTrail7[4][7] = {{0, 0, 1, 1, 2, 2, 3}, {0, 2, 4, 0, 5, 1, 6}, {1, 3, 3, 5, 2, 6, 4}, {3, 4, 4, 5, 5, 6, 6}};
    constexpr unsigned trail7_word(int d) // the seven groups of trail d, three bits each
    {
        unsigned w = 0;
        for (int m = 0; m < 7; ++m) w |= (unsigned)kTrail7[d][m] << (3 * m);
        return w;
    }
    // where the product of groups I <= J sits: bits 0-2 instruction, 3-4 block, 5 = the block holds J x I (transposed)
    constexpr unsigned trail7_pair(int I, int J)
    {
        for (int d = 0; d < 4; ++d)
            for (int m = 0; m < 7; ++m)
            {
                const int a = kTrail7[d][m], b = kTrail7[d][(m + 1) % 7];
                if (a == I && b == J) return (unsigned)m | (unsigned)d << 3;
                if (a == J && b == I) return (unsigned)m | (unsigned)d << 3 | 32u;
            }
        return 0xffu; // (never: the trails cover every pair, checked below)
    }
    constexpr unsigned long long trail7_row(int I) // codes of (I, J), J = 0 .. 6, six bits each
    {
        unsigned long long w = 0;
        for (int J = I; J < 7; ++J) w |= (unsigned long long)(trail7_pair(I, J) & 63u) << (6 * J);
        return w;
    }
    constexpr bool trail7_covers()
    {
        unsigned seen = 0;
        for (int I = 0; I < 7; ++I)
            for (int J = I; J < 7; ++J)
            {
                const unsigned c = trail7_pair(I, J);
                if (c == 0xffu) return false;
                seen |= 1u << ((c & 7u) + 7u * ((c >> 3) & 3u));
            }
        return seen == (1u << 28) - 1u; // 28 pairs in 28 different slots
    }
    static_assert(trail7_covers(), "the four trails must cover every pair of groups exactly once");

    template <int ND>
    struct OuterAcc
    {
        static constexpr int G = (ND + 3) / 4, ND_DELTA = G / 2 + 1, NH = (G + 3) / 4;
        static constexpr bool TRAIL7 = G == 7;
        static constexpr int NI = TRAIL7 ? 7 : ND_DELTA * NH;
        static constexpr int STRIDE = ND;
        static constexpr int ROWS = 64;
        static constexpr int SLAB = ROWS * ND > 768 ? ROWS * ND : 768;
        static_assert(NI * 64 <= SLAB, "parked accumulators must fit the slab");
        double acc[NI];
        __device__ __forceinline__ void init(int)
        {
#pragma unroll
            for (int i = 0; i < NI; ++i) acc[i] = 0.0;
        }
        __device__ __forceinline__ void accumulate(const double *slab, int lane, int nsteps = ROWS / 4)
        {
            // the lane's G read offsets are recomputed per call (a dozen integer instructions per 64 pixels): hoisted
            // out of the caller's pixel loop they would occupy VGPRs through the sample loop, which has none to spare
            asm volatile("" : "+v"(lane));
            const int kq = lane >> 4, d = (lane >> 2) & 3, e = lane & 3;
            const double *base[G];
            if constexpr (TRAIL7)
            {
                constexpr unsigned t0 = trail7_word(0), t1 = trail7_word(1), t2 = trail7_word(2), t3 = trail7_word(3);
                const unsigned tw = d == 0 ? t0 : d == 1 ? t1 : d == 2 ? t2 : t3;
#pragma unroll
                for (int m = 0; m < G; ++m) base[m] = slab + kq * ND + 4 * (int)((tw >> (3 * m)) & 7u) + e;
            }
            else
            {
#pragma unroll
                for (int m = 0; m < G; ++m) base[m] = slab + kq * ND + 4 * ((m + d) % G) + e;
            }
            if constexpr (TRAIL7)
            {
                if (nsteps == ROWS / 4)
                { // full slab: the reads of the next two steps are in flight while the current two are multiplied
                    double Wa[2 * G], Wb[2 * G];
#pragma unroll
                    for (int m = 0; m < G; ++m) { Wa[m] = base[m][0]; Wa[G + m] = base[m][4 * ND]; }
#pragma unroll
                    for (int step = 0; step < ROWS / 4; step += 4)
                    {
#pragma unroll
                        for (int m = 0; m < G; ++m) { Wb[m] = base[m][4 * (step + 2) * ND]; Wb[G + m] = base[m][4 * (step + 3) * ND]; }
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int m = 0; m < 7; ++m)
                                acc[m] = __builtin_amdgcn_mfma_f64_4x4x4f64(Wa[h * G + m], Wa[h * G + (m + 1) % 7], acc[m], 0, 0, 0);
                        if (step + 4 < ROWS / 4)
                        {
#pragma unroll
                            for (int m = 0; m < G; ++m) { Wa[m] = base[m][4 * (step + 4) * ND]; Wa[G + m] = base[m][4 * (step + 5) * ND]; }
                        }
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int m = 0; m < 7; ++m)
                                acc[m] = __builtin_amdgcn_mfma_f64_4x4x4f64(Wb[h * G + m], Wb[h * G + (m + 1) % 7], acc[m], 0, 0, 0);
                    }
                    return;
                }
            }
#pragma unroll 4
            for (int step = 0; step < nsteps; ++step)
            {
                double W[G];
#pragma unroll
                for (int m = 0; m < G; ++m) W[m] = base[m][4 * step * ND];
                if constexpr (TRAIL7)
                {
#pragma unroll
                    for (int m = 0; m < 7; ++m) acc[m] = __builtin_amdgcn_mfma_f64_4x4x4f64(W[m], W[(m + 1) % 7], acc[m], 0, 0, 0);
                }
                else
                {
#pragma unroll
                    for (int dl = 0; dl < ND_DELTA; ++dl)
#pragma unroll
                        for (int h = 0; h < NH; ++h)
                            acc[dl * NH + h] = __builtin_amdgcn_mfma_f64_4x4x4f64(W[(4 * h) % G], W[(4 * h + dl) % G], acc[dl * NH + h], 0, 0, 0);
                }
            }
        }
        __device__ __forceinline__ void store(double *dst, int lane) const
        {
#pragma unroll
            for (int i = 0; i < NI; ++i) dst[i * 64 + lane] = acc[i];
        }
        // sum of element (i, j), i <= j, over the waves' parked accumulators
        static __device__ __forceinline__ double gather(const double *rows, int i, int j, int nwaves)
        {
            const int I = i >> 2, J = j >> 2, dd = J - I; // 0 .. G - 1
            int off;
            if constexpr (TRAIL7)
            {
                constexpr unsigned long long r0 = trail7_row(0), r1 = trail7_row(1), r2 = trail7_row(2), r3 = trail7_row(3),
                                             r4 = trail7_row(4), r5 = trail7_row(5), r6 = trail7_row(6);
                const unsigned long long rw = I == 0 ? r0 : I == 1 ? r1 : I == 2 ? r2 : I == 3 ? r3 : I == 4 ? r4 : I == 5 ? r5 : r6;
                const unsigned c = (unsigned)(rw >> (6 * J)) & 63u;
                const bool flip = (c & 32u) != 0; // the block holds group J x group I: transposed
                const int ri = flip ? j & 3 : i & 3, cj = flip ? i & 3 : j & 3;
                off = (int)(c & 7u) * 64 + 16 * ri + 4 * (int)((c >> 3) & 3u) + cj;
            }
            else
            {
                int dl, g, ri, cj;
                if (dd <= G / 2) { dl = dd; g = I; ri = i & 3; cj = j & 3; }
                else { dl = G - dd; g = J; ri = j & 3; cj = i & 3; } // the block holds group J x group I: transposed
                const int h = g >> 2, d = g & 3;
                off = (dl * NH + h) * 64 + 16 * ri + 4 * d + cj;
            }
            double s = 0.0;
            for (int wv = 0; wv < nwaves; ++wv) s += rows[wv * SLAB + off];
            return s;
        }
    };

    // Device-scope, cache-bypassing accesses (global_load / global_store ... sc1): a store is written through to memory, a load
    // is served from there -- what one workgroup hands to a workgroup on another CU / XCD without a cache-wide fence on either side
    // (MI355X: per-XCD L2s, a CU's L1 is never refreshed by other CUs' stores).  Ordered by s_waitcnt vmcnt(0) before the flag.
    __device__ __forceinline__ double ld_fresh(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ __forceinline__ void st_fresh(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    // (tile partials: plain stores behind the agent-scope release of ticket_finalize, sc1 loads by the summing workgroup.  The fence-free
    // form -- sc0 sc1 stores and loads on both sides, MI355X_MICROARCH.md "valid forms" -- is correct once the per-patch costs bound for
    // pinned host memory are stored the same way, and exactly as fast: profiles/r05_kfused_experiments.txt 2.)
    __device__ __forceinline__ double ld_part(const double *p) { return ld_fresh(p); }

    // 1 / ((K - bad) F P) of a problem.  Device-side LM and the host-driven loop keep it outside the descriptor (inv_ptr:
    // device memory -- for the host-driven loop a word of the CPU-writable push block, see Engine::push_block): read FRESH
    // by one lane per wave (a scalar load could hit a stale scalar-cache line inside the persistent kernel) and broadcast.
    // (Not from pinned HOST memory: a system-scope load per wave from there measured +35 us per evaluation.)
    template <bool FRESH>
    __device__ __forceinline__ double residual_scale(const ProblemDesc &d, int lane)
    {
        if (d.inv_ptr == nullptr) return d.inv_num_residuals;
        if constexpr (!FRESH) return *d.inv_ptr; // one evaluation per launch: the caches start empty
        double v = 0.0;
        if (lane == 0) v = __hip_atomic_load(d.inv_ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return __shfl(v, 0, 64);
    }

    // outlier flag of a keypoint.  FRESH (persistent kernel: the host rewrites the flags between commands while the caches
    // stay warm): the aligned word that holds the byte, through a device-scope atomic load that bypasses the caches.
    template <bool FRESH>
    __device__ __forceinline__ unsigned outlier_flag(const unsigned char *flags, int kp)
    {
        if constexpr (!FRESH) return flags[kp];
        const unsigned long long a = (unsigned long long)(flags + kp);
        const unsigned w = __hip_atomic_load((const unsigned *)(a & ~3ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return (w >> (8 * (unsigned)(a & 3ull))) & 0xffu;
    }

    // pixel index -> patch index; P is wave-uniform, so the branches are scalar and the common patch sizes (1, and
    // the 8-pixel pattern of the reference's tests) skip the ~14-instruction integer division
    __device__ __forceinline__ int patch_of(int g, int P) { return P == 1 ? g : (P == 8 ? g >> 3 : g / P); }

    __device__ __forceinline__ double wave_sum(double v)
    {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
        return v;
    }

    // A wave-uniform 64-bit value moved into SGPRs.  The compiler keeps the result of a uniform VECTOR operation (a pointer
    // fetched through a descriptor, an fp64 quotient -- there is no scalar fp64 unit) in a VGPR pair for the whole kernel;
    // the k = 4 kernel has none to spare (168 of 170), a spilled SGPR costs one v_readlane per use instead.
    __device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v)
    {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return ((unsigned long long)hi << 32) | lo;
    }
    __device__ __forceinline__ double uniform_f64(double v) { return __builtin_bit_cast(double, uniform_u64(__builtin_bit_cast(unsigned long long, v))); }

    // Workgroups are handed to the 8 XCDs round-robin (workgroup b runs on XCD b % 8) and every XCD has its own L2:
    // give each XCD a CONTIGUOUS range of tiles (= a band of the keyframe and of the current image), so that the
    // rows neighbouring tiles share are fetched into one L2 instead of eight.  Bijection for any tile count.
    __device__ __forceinline__ int xcd_tile_of_block(int b, int ntiles)
    {
        constexpr int kXcds = 8;
        const int q = ntiles / kXcds, r = ntiles % kXcds;
        const int x = b % kXcds, j = b / kXcds;
        return x * q + (x < r ? x : r) + j;
    }

    // One round of the tile in SAMPLE-PARALLEL form (see k_fused_sp below), with S = 2^logs a run-time value: used
    // by k_fused for the pixels of a tile beyond its last full round when all their (pixel, sample) pairs fit the
    // workgroup.  On configs[1] a tile is 1 600 pixels = 25 chunks of 64 on 12 waves: the 25th chunk used to be one
    // more chunk for one wave, i.e. 7 chunks on its SIMD against 6 on the others (+5 us measured against a tile of
    // exactly two rounds).
    // LOGS_CT >= 0: S is the compile-time 2^LOGS_CT (the exchange loops over the samples unroll); -1: run-time `logs_rt`.
    template <int KD, bool WITH_J, int HALF_GRAD, int NWAVES, int LOGS_CT = -1, int MS = 1>
    __device__ __forceinline__ void sp_round_rt(const ProblemDesc &d, const TileDesc &tile, const Camera &cam,
                                                const PoseEntry<KD> *__restrict__ ftab, const PoseEntry<KD> &mid,
                                                const unsigned char *__restrict__ I_cur, int logs_rt, int base, int npx,
                                                long long pix0, int lane, int wave, double *slab,
                                                OuterAcc<6 * KD + 1> &acc, double *__restrict__ rho_out, int &nvalid,
                                                double inv, double *__restrict__ patch_cost,
                                                double *__restrict__ patch_blocks_strided, int frame, double &cost_local)
    {
        constexpr int ND = 6 * KD + 1, RS = OuterAcc<ND>::STRIDE, E = ND * (ND + 1) / 2;
        const int logs = LOGS_CT >= 0 ? LOGS_CT : logs_rt;
        const int SS = 1 << logs, PXW = 64 >> logs, P = d.P;
        const int pw = lane >> logs, sidx = lane & (SS - 1), lane0 = lane & ~(SS - 1);
        const unsigned long long gmask = (SS == 64 ? ~0ull : ((1ull << SS) - 1ull)) << lane0;
        // MS = 2: a pixel takes S / 2 lanes and every lane two consecutive blur samples, issued and retired as a pair like the
        // lane-per-pixel loop's.  The per-pixel part of the lane (patch centre, ray, Huber, exchange: 600 of a lane's 751
        // instructions at one sample) is then paid by half as many lanes -- the remainder round's waves halve.
        const double fS = (double)(float)(SS * MS); // A8
        const int g = base + wave * PXW + pw;
        const bool in = g < npx;
        double res = 0.0, w = 0.0, rho = 0.0, cur = 0.0;
        double vals[MS] = {};
        bool ok_l = false, flagged = false;
        double Jc[WITH_J ? 6 * KD : 1] = {};
        // A lane's pose entry is its SAMPLE's: 59 doubles per lane that no longer arrive as scalar operands.  Read from
        // global memory they come in four or five dependent pieces (the registers cannot hold an entry), each an L2
        // round trip: 1.7 us of the round's 5.3 us.  So the wave first copies the frame's S entries into its own slab
        // (free until the samples are done and the rows are exchanged) and every lane reads its entry from LDS.
        constexpr int EW = (int)(sizeof(PoseEntry<KD>) / sizeof(double));
        const bool staged = WITH_J && SS * MS * EW <= OuterAcc<ND>::SLAB; // the cost-only kernels have no slabs (and need 15 of the 59 doubles)
        if (staged)
        {
            const MBAVO_GLOBAL double *src = (const MBAVO_GLOBAL double *)ftab;
            for (int z = lane; z < SS * MS * EW; z += 64) slab[z] = src[z];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (in)
        {
            const int kpl = patch_of(g, P), pp = g - kpl * P;
            const int kp = tile.kp_begin + kpl;
            flagged = d.outlier != nullptr && d.outlier[kp] == 1;
            const double kx = d.kp_xy[(size_t)kp * d.kp_stride], ky = d.kp_xy[(size_t)kp * d.kp_stride + 1];
            const double kz = d.kp_z[kp];
            double pcx, pcy;
            patch_centre_rt(mid.rt, mid.q, kx, ky, kz, cam, pcx, pcy);
            const int px = (int)(pcx + d.pattern[2 * pp]); // truncation, A3 (pixel_row)
            const int py = (int)(pcy + d.pattern[2 * pp + 1]);
            if (!(px < 0 || px > cam.W - 1 || py < 0 || py > cam.H - 1))
            {
                cur = (double)((const MBAVO_GLOBAL unsigned char *)I_cur)[py * cam.W + px];
                double ray[3];
                unit_ray(cam, (double)px, (double)py, ray);
                const double iz = reciprocal(kz + 1e-8);
                auto lane_samples = [&](const PoseEntry<KD> *pe) {
                    if constexpr (MS == 1)
                    {
                        SampleInFlight f;
                        sample_issue<KD, WITH_J, HALF_GRAD>(pe[0], ray, kz, iz, cam, d.ref_img, d.ref_dIxy, f);
                        ok_l = f.taps.ok;
                        sample_retire<KD, WITH_J, false, HALF_GRAD>(pe[0], f, ray, kz, iz, cam, vals[0], Jc);
                    }
                    else
                    {
                        static_assert(MS == 1 || MS == 2, "one or two samples per lane");
                        SampleInFlight fa, fb;
                        sample_issue<KD, WITH_J, HALF_GRAD>(pe[0], ray, kz, iz, cam, d.ref_img, d.ref_dIxy, fa);
                        sample_issue<KD, WITH_J, HALF_GRAD>(pe[1], ray, kz, iz, cam, d.ref_img, d.ref_dIxy, fb);
                        ok_l = fa.taps.ok && fb.taps.ok;
                        sample_retire<KD, WITH_J, false, HALF_GRAD>(pe[0], fa, ray, kz, iz, cam, vals[0], Jc);
                        sample_retire<KD, WITH_J, false, HALF_GRAD>(pe[1], fb, ray, kz, iz, cam, vals[MS - 1], Jc);
                    }
                };
                if (staged)
                    lane_samples(((const PoseEntry<KD> *)slab) + MS * sidx); // LDS
                else
                    lane_samples(ftab + MS * sidx);
            }
        }
        if (staged)
        { // every lane is done with the staged entries before the rows overwrite them
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        const bool valid = in && (__ballot(ok_l) & gmask) == gmask;
        double isum = 0.0;
        for (int j = 0; j < SS; ++j)
#pragma unroll
            for (int m = 0; m < MS; ++m) isum += __shfl(vals[m], lane0 + j, 64); // sample order, as the sequential loop
        if (valid) res = quotient(isum, fS) - cur;
        huber_weight(res, d.huber_a, w, rho);
        if (in && sidx == 0)
        {
            if (P == 1)
            { // one-pixel patches: the patch cost is this pixel's (see k_fused)
                const double c = rho * inv;
                const long long patch = (long long)frame * d.K + tile.kp_begin + g;
                if (patch_cost) patch_cost[d.patch_base + patch] = c;
                if (patch_blocks_strided) patch_blocks_strided[patch * E] = c;
                if (!flagged) cost_local += c;
            }
            else
                rho_out[pix0 + g] = rho;
            nvalid += valid ? 1 : 0;
        }
        const bool keep = valid && !flagged;
        if (WITH_J)
        {
            double *mine = slab + lane * RS;
#pragma unroll
            for (int i = 0; i < 6 * KD; ++i) mine[i] = Jc[i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            constexpr int NOUT = (6 * KD + 3) / 4; // S >= 4 lanes share the 6k entries of a pixel
            // a dropped pixel (invalid, out of the image, outlier) parks a ZERO row by a select, not by a zero weight: its
            // samples' contributions may be non-finite (a NaN / inf keypoint depth passes the detector's `!(z < 1e-2)`
            // test as in the reference; an out-of-bounds warp) and 0 * NaN would poison the frame's H and g
            const double inv = 1.0 / fS;
            double outv[NOUT];
#pragma unroll
            for (int t = 0; t < NOUT; ++t)
            {
                const int i = sidx + t * SS;
                double a = 0.0;
                if (i < 6 * KD)
                    for (int j = 0; j < SS; ++j) a += slab[(lane0 + j) * RS + i];
                outv[t] = keep ? w * (a * inv) : 0.0;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            double *row = slab + pw * RS;
            if (sidx == 0) row[0] = keep ? w * res : 0.0;
#pragma unroll
            for (int t = 0; t < NOUT; ++t)
            {
                const int i = sidx + t * SS;
                if (i < 6 * KD) row[1 + i] = outv[t];
            }
            if (PXW < 4)
                for (int z = lane; z < (4 - PXW) * RS; z += 64) slab[PXW * RS + z] = 0.0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            acc.accumulate(slab, lane, (PXW < 4 ? 4 : PXW) / 4);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }

    __device__ __forceinline__ void set_prio(int p) // s_setprio takes an immediate; p is wave-uniform
    {
        switch (p)
        {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
        }
    }
#ifndef MBAVO_SP_REM_MOD
#define MBAVO_SP_REM_MOD 256 // (kThreads: the remainder beyond the last FULL round only, the scheme before)
#endif
    // POSE: no pose kernel ahead of this one -- every workgroup computes ITS frame's S table entries itself (the two stages of
    // k_pose_table, segments in the not-yet-used row slabs) into its own S entries of `table_w`, which the host passes as
    // `table` too: the sample loop needs the entries behind wave-uniform SCALAR loads from read-only memory (through LDS it
    // was 1.5x slower), so they go out through the L2 and come back through the scalar cache.
    template <int KD, bool WITH_J, int HALF_GRAD, bool POSE = false> // HALF_GRAD: 0 float pairs, 1 IEEE half pairs, 2 packed keyframe words
    __global__ __launch_bounds__((waves_of<KD, WITH_J>() * 64)) void k_fused(const ProblemDesc *__restrict__ descs,
                                                        const TileDesc *__restrict__ tiles,
                                                        const PoseEntry<KD> *__restrict__ table,
                                                        double *__restrict__ rho_out,
                                                        double *__restrict__ patch_cost,
                                                        double *__restrict__ patch_blocks_strided,
                                                        double *__restrict__ partials,
                                                        PoseEntry<KD> *table_w, int *status, int table_stride)
    {

        constexpr int ND = Pack<KD>::ND, E = Pack<KD>::E, PS = Pack<KD>::PSTRIDE;
        constexpr int kWavesPerGroup = waves_of<KD, WITH_J>(), kThreads = kWavesPerGroup * 64;
        constexpr int SLAB = OuterAcc<ND>::SLAB;     // doubles per wave: rows, and the parked accumulators at the end
        constexpr int RS = OuterAcc<ND>::STRIDE;     // row stride (>= ND, zero padded)
        extern __shared__ __attribute__((aligned(16))) double lds[];
        double *rows = lds;                                               // [8 waves][64 pixels][ND] (WITH_J only)
        double *red = lds + (WITH_J ? kWavesPerGroup * SLAB : 0);         // [2][8]

        const int lane = threadIdx.x & 63;
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        MBAVO_FSTAMP(0);
#if defined(MBAVO_FUSED_STAMPS)
        if (threadIdx.x == 0 && blockIdx.x < 2048)
        {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            g_fused_stamps[blockIdx.x * 8 + 7] = ((unsigned long long)xcc << 32) | hw;
        }
#endif
        const int tile_id = xcd_tile_of_block((int)blockIdx.x, (int)gridDim.x);
        const TileDesc tile = tiles[tile_id];
        const ProblemDesc &d = descs[tile.prob];
        if (d.active != nullptr && (*d.active & (WITH_J ? 2 : 1)) == 0) return; // device-side LM: problem sits this pass out
        const int S = d.S, K = d.K, P = d.P, frame = tile.frame;
        Camera cam;
        cam.fx = d.fx; cam.fy = d.fy; cam.cx = d.cx; cam.cy = d.cy; cam.H = d.H; cam.W = d.W;
        // The intrinsics live in VGPRs: the sample loop streams the pose entry through ~100 SGPRs, and with four more
        // doubles there the allocator spilled scalars into VGPR lanes inside the loop (16 v_readlane per sample pair);
        // u = fx * x + cx also has two scalar operands otherwise (one extra move each).
        // (k = 4 with Jacobians only: the other instantiations have SGPRs to spare and, at 4 waves per SIMD, no VGPRs.)
        if constexpr (KD == 4 && WITH_J) asm volatile("" : "+v"(cam.fx), "+v"(cam.fy), "+v"(cam.cx), "+v"(cam.cy));
        // The frame's S table entries are read with wave-uniform addresses -> scalar loads.  (Staging the table
        // in LDS and reading it as a broadcast was measured 1.5x SLOWER on the fused kernel: one ds_read per FMA
        // operand instead of an SGPR operand.)
        long long tab_off = d.pose_base + frame * S;
        if constexpr (POSE)
        {
            tab_off = (long long)blockIdx.x * table_stride; // (the batch's largest S: problems of a batch may differ)
            // (cost-only kernels have no row slabs: the segments get their own LDS behind the wave sums, see the launch)
            SplineSeg *segs = (SplineSeg *)(WITH_J ? rows : red + 2 * kWavesPerGroup);
            MBAVO_PSTAMP(1);
            frame_pose_entries<KD, WITH_J, true>(d, d.knots_t, d.knots_R, frame, table_w + tab_off, segs, wave, lane, status,
                                                 tile.kp_begin == 0);
            MBAVO_PSTAMP(3);
            // the entries are in this XCD's L2 (vector stores write through) once every wave's stores are performed; the
            // scalar cache has never seen these lines in this launch.  The offset is made opaque AFTER the barrier: the
            // reads below are loads from `table` (read-only, no alias as far as the compiler knows) and would otherwise be
            // free to move above it.
            // (vmcnt(0) explicitly: a workgroup-scope release only orders accesses through the CU's vector L1, which the
            // stores write through; the READERS go through the scalar cache to the L2, so every storing wave waits for its
            // stores to be acknowledged by the L2 before it arrives at the barrier)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            __builtin_amdgcn_s_dcache_inv();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the invalidation is complete before any table load issues
            asm volatile("" : "+s"(tab_off) : : "memory");
            MBAVO_PSTAMP(4);
        }
        const PoseEntry<KD> *__restrict__ ftab = table + tab_off;
        {
            // Warm the scalar cache: the frame's S entries were written by the pose kernel (cold here, often in another
            // XCD's L2), and the sample loop reads them in ~11 dependent scalar-load groups per sample pair -- each a
            // miss on the first pass.  One 8-byte read per 64-byte line, spread over the waves, all in flight together.
            const double *tb = (const double *)ftab;
            const int nlines = (S * (int)sizeof(PoseEntry<KD>) + 63) / 64;
            double warm = 0.0;
            for (int l = wave; l < nlines; l += kWavesPerGroup) warm += tb[l * 8];
            if (warm == 1.2345678e301) red[0] = warm; // never true: keeps the loads alive
        }
        const PoseEntry<KD> &mid = ftab[S / 2]; // patch centres use sample S/2 (compute_local_patches_xy.cu:26)
        const unsigned char *__restrict__ I_cur = (const unsigned char *)uniform_u64((unsigned long long)d.cur_imgs[frame]);
        const long long pix0 = d.pixel_base + ((long long)frame * K + tile.kp_begin) * P;
        const int npx = tile.kp_count * P;
        // (double)(float)S (A8), the reciprocal quotient() forms of it and 1 / it: the same bits pixel_row computes per call
        const double fS_u = uniform_f64((double)(float)S), rS_u = uniform_f64(quotient_recip((double)(float)S)),
                     inv_S_u = uniform_f64(1.0 / (double)(float)S);
        const double inv_fx = uniform_f64(1.0 / d.fx), inv_fy = uniform_f64(1.0 / d.fy); // patch_centre_fast

        OuterAcc<ND> acc;
        acc.init(lane);
        double *slab = rows + wave * SLAB;
        int nvalid = 0;

        // S = 2^sp_logs in 4 .. 64: the last round may go sample-parallel (sp_round_rt)
        int sp_logs = 0;
        while ((1 << sp_logs) < S) ++sp_logs;
        const bool sp_ok = (1 << sp_logs) == S && sp_logs >= 2 && sp_logs <= 6;
        // The pixels beyond the last full round go FIRST and sample-parallel: spread over the waves of all four SIMDs
        // and overlapped with the other waves' first round, instead of one more chunk for one wave (and so for one
        // SIMD: 7 chunks against 6 on the others) at the end.
        // One-pixel patches (dense mode): the patch cost is the pixel's own, taken where rho is computed; the
        // per-pixel rho scratch (8 B per pixel written and read back at the end of the tile) is not touched.
        const double inv = uniform_f64(residual_scale<false>(d, lane));
        double cost_local = 0.0;
        int main_end = npx;
        MBAVO_FSTAMP(1);
        MBAVO_PSTAMP(5);
        // The SIMD arbiter serves the OLDEST ready wave first: of the three waves a SIMD holds, the oldest ran ahead
        // through all its rounds and the youngest finished last, alone, at a third of the SIMD's rate (s_memrealtime
        // stamps per wave, tools/fused_stamps.py: 24.4 / 29.9 / 33.7 us on configs[1]).  The user priority outranks
        // age: every wave starts at 3 and steps down over its last three rounds (3, 2, 1; the parking + MFMA phase of a
        // round one lower), so that a wave that is behind overtakes and the three finish together.  A/B, fused kernel:
        // configs[1] 34.6 -> 33.4 us, 512 pairs 106.1 -> 101.5, 1080p S=16 212.4 -> 196.0 (profiles/r02_kfused_experiments.txt).
        __builtin_amdgcn_s_setprio(3);
        if (sp_ok)
        {
            // S >= 8: two samples per lane (S / 2 lanes per pixel), see sp_round_rt
            const int ms = sp_logs >= 3 ? 2 : 1;
            const int lane_logs = sp_logs - (ms == 2 ? 1 : 0);
            // The lane-per-pixel rounds take a multiple of 256 pixels (the same number of 64-pixel chunks on each of the four
            // SIMDs; the last of these rounds may be a partial one), the pixels beyond go sample-parallel.
            // Worth it while the remainder's lanes, spread over the four SIMDs, cost a SIMD fewer instructions than the one
            // more 64-pixel chunk it would otherwise get (per lane ~600 per-pixel + 151 per sample, profiles/r02_pmc_sq.json)
            const int rem = npx % MBAVO_SP_REM_MOD;
            const long long sp_lanes = (long long)rem << lane_logs;
            if (rem > 0 && sp_lanes <= kThreads && sp_lanes * (600 + 151 * ms) < 256ll * (600 + 151 * S))
            {
                main_end = npx - rem;
                // dealt out from the LAST wave down: when the last lane-per-pixel round is a partial one, the waves without
                // a chunk in it take the remainder (64 pairs: the other way round cost 1.4 us, waves 0-4 then had both)
                const int sp_wave = kWavesPerGroup - 1 - wave;
                if (main_end + sp_wave * (64 >> lane_logs) < npx)
                {
#define MBAVO_SP_ROUND(L, M)                                                                                           \
    sp_round_rt<KD, WITH_J, HALF_GRAD, kWavesPerGroup, L, M>(d, tile, cam, ftab, mid, I_cur, lane_logs, main_end, npx, pix0, \
                                                             lane, sp_wave, slab, acc, rho_out, nvalid, inv, patch_cost,     \
                                                             patch_blocks_strided, frame, cost_local)
                    // S = 4, 8, 16 as compile-time cases (the exchange loops over the samples unroll: -1 us of the
                    // remainder round's 4.4 us on configs[1]); other powers of two take the run-time form
                    if (ms == 2)
                        switch (sp_logs)
                        {
                        case 3: MBAVO_SP_ROUND(2, 2); break;
                        case 4: MBAVO_SP_ROUND(3, 2); break;
                        default: MBAVO_SP_ROUND(-1, 2); break;
                        }
                    else
                        switch (sp_logs)
                        {
                        case 2: MBAVO_SP_ROUND(2, 1); break;
                        default: MBAVO_SP_ROUND(-1, 1); break;
                        }
#undef MBAVO_SP_ROUND
                }
            }
        }
#if defined(MBAVO_EXP_NO_ROUNDS) // timing experiment: launch + prologue + end-of-tile work only
        main_end = 0;
#endif
        MBAVO_FSTAMP(2);
        MBAVO_WSTAMP(0, __builtin_amdgcn_s_memrealtime());
#if defined(MBAVO_FUSED_STAMPS)
        {
            unsigned hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            MBAVO_WSTAMP(2, hw);
        }
#endif
        for (int base = 0; base < main_end; base += kThreads)
        {
            const int g = base + (int)threadIdx.x;
            double res = 0.0, w = 0.0, rho = 0.0;
            bool keep = false;
            double Jrow[WITH_J ? 6 * KD : 1]; // the SUM over the samples, defined only where keep (see pixel_row)
            if (g < main_end)
            {
                const int kpl = patch_of(g, P), pp = g - kpl * P;
                const int kp = tile.kp_begin + kpl;
                const bool flagged = d.outlier != nullptr && d.outlier[kp] == 1;
                const double kx = d.kp_xy[(size_t)kp * d.kp_stride], ky = d.kp_xy[(size_t)kp * d.kp_stride + 1];
                const double kz = d.kp_z[kp];
                double pcx, pcy;
                // The pixel is (int)(centre + pattern offset): the cheap centre is good wherever that sum is not within
                // 1e-5 of an integer; if ANY lane of the wave is (zero motion: all of them), the wave takes the reference's
                // operation order -- a uniform branch, never taken on moving cameras (2e-5 of the pixels per axis).
                patch_centre_fast(mid.rt, mid.R, kx, ky, kz, cam, inv_fx, inv_fy, pcx, pcy);
                const bool unsure = !(patch_centre_sure(pcx + d.pattern[2 * pp]) && patch_centre_sure(pcy + d.pattern[2 * pp + 1]));
                if (__builtin_amdgcn_ballot_w64(unsure) != 0) patch_centre_rt(mid.rt, mid.q, kx, ky, kz, cam, pcx, pcy);
                // Wave priority by remaining work (see below the round loop's head): the sample loop of a wave that is
                // behind goes first, the row parking + MFMA phase one step lower.
                const int rem_rounds = (main_end - base + kThreads - 1) / kThreads; // 1 in the last round
                const int prio = rem_rounds > 3 ? 3 : rem_rounds;
                set_prio(prio);
                const bool valid = pixel_row<KD, WITH_J, HALF_GRAD>(ftab, S, cam, d.ref_img, d.ref_dIxy, I_cur, pcx, pcy, kz,
                                                         d.pattern[2 * pp], d.pattern[2 * pp + 1], res, Jrow, fS_u, rS_u);
                set_prio(prio - 1);
                huber_weight(res, d.huber_a, w, rho);
                if (P == 1)
                {
                    const double c = rho * inv;
                    const long long patch = (long long)frame * K + kp;
                    if (patch_cost) patch_cost[d.patch_base + patch] = c;
                    if (patch_blocks_strided) patch_blocks_strided[patch * E] = c;
                    if (!flagged) cost_local += c;
                }
                else
                    rho_out[pix0 + g] = rho;
                nvalid += valid ? 1 : 0;
                keep = valid && !flagged;
            }
            if (WITH_J)
            {
                // rows of this wave's 64 pixels -> its LDS slab; the same wave reads back what it wrote: LDS executes a
                // wave's operations in order, only the compiler has to be kept from reordering across these points.
                // Inactive / invalid / outlier pixels park ZERO rows by a branch, not by a zero weight: their Jrow is
                // undefined (an out-of-bounds sample may have left NaNs in it).
                static_assert(OuterAcc<ND>::ROWS == 64, "one row per lane");
                double *mine = slab + lane * RS;
                if (keep)
                {
                    const double wj = w * inv_S_u; // Huber weight times the 1/S of the mean over the samples
                    mine[0] = w * res;
#pragma unroll
                    for (int i = 0; i < 6 * KD; ++i) mine[1 + i] = wj * Jrow[i];
                }
                else
                {
#pragma unroll
                    for (int i = 0; i < ND; ++i) mine[i] = 0.0;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#if !defined(MBAVO_EXP_NO_MFMA) // timing experiment switch
                acc.accumulate(slab, lane);
#endif
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }

        MBAVO_WSTAMP(1, __builtin_amdgcn_s_memrealtime());
        MBAVO_FSTAMP(3); // (wave 0's own rounds are done; the barrier below waits for the slowest wave)
        // per-patch cost = slot 0 of the reference's patch block (:232-238), and the
        // tile's share of the frame cost (outlier patches skipped, :265-272)
        if (P != 1)
        { // the patches' pixels were handled by other waves: their rho values are read back (one-pixel patches took
          // their cost in the round loop)
            __syncthreads();
            for (int kpl = threadIdx.x; kpl < tile.kp_count; kpl += kThreads)
            {
                const double *r = rho_out + pix0 + (long long)kpl * P;
                const double c = patch_rho_sum(r, P) * inv; // reduction.h order
                const int kp = tile.kp_begin + kpl;
                const long long patch = (long long)frame * K + kp;
                if (patch_cost) patch_cost[d.patch_base + patch] = c;
                if (patch_blocks_strided) patch_blocks_strided[patch * E] = c;
                if (!(d.outlier != nullptr && d.outlier[kp] == 1)) cost_local += c;
            }
        }
        const double wc = wave_sum(cost_local);
        const double wv = wave_sum((double)nvalid);
        if (lane == 0) { red[wave] = wc; red[kWavesPerGroup + wave] = wv; }
        // every wave parks its accumulators in its slab; entry e = (i, j) of the packed block is then the sum
        // over waves (and pixel groups) in a fixed order.  One barrier for the scratch and the slabs.
        if (WITH_J) acc.store(slab, lane);
        __syncthreads();
        MBAVO_FSTAMP(4);
        double *out = partials + (size_t)tile_id * PS;
        if (threadIdx.x == 0)
        {
            double c = 0.0, v = 0.0;
            for (int i = 0; i < kWavesPerGroup; ++i) { c += red[i]; v += red[kWavesPerGroup + i]; }
            out[0] = v;
            out[E] = c;
        }
        if (WITH_J)
        {
            for (int e = 1 + threadIdx.x; e < E; e += kThreads)
            {
                int i, j;
                tri_decode(e, ND, i, j);
                out[e] = OuterAcc<ND>::gather(rows, i, j, kWavesPerGroup);
            }
        }
#if defined(MBAVO_FUSED_STAMPS)
        __syncthreads();
        MBAVO_FSTAMP(5);
#endif
    }

    // ------------------------------------------------------------------ single-launch pieces
    // An evaluation used to be three dependent launches (pose table -> fused -> finalize): for small problems that is
    // pure launch latency (20 us for ~2 us of arithmetic).  With these two pieces the fused kernel does all of it:
    //  * prologue: the workgroup computes ITS frame's S pose entries itself (the pose kernel's arithmetic, same
    //    lane layout: one wave per knot, one lane per (sample, Jacobian column)) straight into LDS;
    //  * epilogue: the workgroup that retires the LAST tile of a (problem, frame) slot -- a ticket counter, device-scope
    //    fences -- sums the slot's tile partials in a fixed order and writes the frame block.  No float atomics, still
    //    bit-reproducible run to run.
#if defined(MBAVO_PERSIST_STAMPS)
    __device__ __forceinline__ unsigned long long *stamp_area()
    {
        __shared__ unsigned long long st[24];
        return st;
    }
#define MBAVO_STAMP(i) do { if (threadIdx.x == 0) stamp_area()[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MBAVO_STAMP(i) do { } while (0)
#endif
    // merge_hessian_gradient_cost (merge_hessian_gradient_cost.cpp:39-86) folded into the finalize step for lists of problems
    // with ONE frame and N == k control knots (start index 0: the local -> global index map of :52-62 is the identity): the
    // thread that holds entry e of the frame block also stores it into the problem's system [cost | g (6k) | H (6k x 6k,
    // column-major, both triangles)] -- the reference's unit ends there (spline_update_step.cpp:232-239), no merge launch.
    template <int KD>
    __device__ __forceinline__ void store_system_entry(double *__restrict__ sys, int e, double v)
    {
        constexpr int M6 = 6 * KD, ND = M6 + 1, E = Pack<KD>::E;
        if (e == E) sys[0] = v; // (the partial slot of the cost)
        else if (e <= M6) sys[e] = v;
        else
        {
            int r, c;
            tri_decode(e - ND, M6, r, c);
            sys[1 + M6 + c * M6 + r] = v;
            if (r != c) sys[1 + M6 + r * M6 + c] = v;
        }
    }

    struct OneArgs
    {
        double *systems;                         // merged outputs (store_system_entry), or null
        const int *bf_tile_begin;                // [nBF + 1]
        int *tickets;                            // [nBF], zero between launches (the last workgroup of a slot resets it)
        double *frame_blocks, *valid_out;        // outputs of the finalize step
        int *status;                             // out-of-range counter (see k_pose_table)
        int *slots_done;                         // one counter: slots finished in this launch
        unsigned long long *host_flag;           // pinned host word, or null: set to `seq` when every slot is done
        unsigned long long seq;
        int nbf;
        unsigned long long t_seen;               // (timing experiment MBAVO_PERSIST_STAMPS)
    };

    // After the workgroup wrote partials[tile_id]: take a ticket of the tile's (problem, frame) slot; the workgroup
    // that draws the last one sums the slot's partials -- LANES tile-lanes per entry, lane l adds tiles l, l + LANES,
    // ... in order, then the lanes are added in order (for slots of up to four tiles and LANES = 2 that is
    // (t0 + t2) + (t1 + t3), the order of k_finalize_flat) -- and writes the frame block.  `scratch`: LANES * EPAD
    // doubles of LDS nobody else is using any more.
#ifndef MBAVO_FIN_INFLIGHT
#define MBAVO_FIN_INFLIGHT 16
#endif
    template <int KD, bool WITH_J, int NTHREADS>
    __device__ __forceinline__ bool ticket_finalize(const ProblemDesc &d, int bf, const double *__restrict__ partials,
                                                    const OneArgs &oa, double *scratch, double inv)
    {
        constexpr int E = Pack<KD>::E, PS = Pack<KD>::PSTRIDE;
        // (workgroups of fewer threads than partial slots -- four waves, k = 4 -- take the slots in passes, one tile-lane)
        constexpr int EFULL = KD == 4 ? 384 : 128, EPAD = EFULL <= NTHREADS ? EFULL : NTHREADS, LANES = NTHREADS / EPAD;
        constexpr int NPASS = (EFULL + EPAD - 1) / EPAD;
        static_assert(E + 1 <= EFULL && LANES >= 1 && (NPASS == 1 || LANES == 1), "one thread per partial slot and tile-lane");
        __shared__ int s_last;
        // Release: every wave's partial stores are acknowledged before the barrier (vmcnt(0): a workgroup-scope release emits no
        // wait outside tgsplit mode), then ONE thread makes them visible device-wide (agent-scope release: L2 write-back) and
        // takes the ticket; the last workgroup acquires.  A device-scope fence by all 768 threads costs 9 us here (measured).
        // Round 3 tried the hand-over WITHOUT the two cache-wide fences (partials stored and loaded with sc1 accesses, vmcnt(0),
        // ticket): 27 parity tests failed and trackFrame was no longer reproducible run to run -- and it was not faster
        // (0.384 vs 0.376 ms per frame; profiles/r03_kfused_experiments.txt 4.).
        // A slot of ONE tile (the coarse pyramid levels of a semi-dense pair) has nobody to hand over to: this workgroup's own
        // partial is in its XCD's L2 once its stores are acknowledged (the L1 is write-through) and the sc1 loads below read it
        // there -- no agent-scope release, no ticket, no acquire (round 5; ~2.5 us of such an evaluation).
        const int n_tiles = oa.bf_tile_begin[bf + 1] - oa.bf_tile_begin[bf];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (n_tiles > 1)
        {
            if (threadIdx.x == 0)
            {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                const int last = atomicAdd(&oa.tickets[bf], 1) == n_tiles - 1 ? 1 : 0;
                if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // the other workgroups' partials (other XCDs' L2s) are read from memory
                s_last = last;
            }
            __syncthreads();
            if (!s_last) return false;
        }
        MBAVO_STAMP(2);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const int t0 = oa.bf_tile_begin[bf], t1 = oa.bf_tile_begin[bf + 1];
        const bool to_host = oa.host_flag != nullptr;
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass)
        {
        const int e = pass * EPAD + threadIdx.x % EPAD, l = threadIdx.x / EPAD;
        const bool mine = l < LANES && e <= E && (WITH_J || e == 0 || e == E);
        double acc = 0.0;
        if (mine)
        {
            const double *pp = partials;
#if MBAVO_FIN_INFLIGHT > 4
            // every partial of this tile-lane in flight at once (predicated loads), added in tile order: each load is a miss to
            // another XCD's L2 or to memory, and a lane of a semi-dense level has 3 .. 16 of them -- as batches of four plus a
            // serial remainder they were up to six latencies in a row
            for (int t = t0 + l; t < t1; t += MBAVO_FIN_INFLIGHT * LANES)
            {
                double v[MBAVO_FIN_INFLIGHT];
#pragma unroll
                for (int u = 0; u < MBAVO_FIN_INFLIGHT; ++u) v[u] = t + u * LANES < t1 ? ld_part(pp + (size_t)(t + u * LANES) * PS + e) : 0.0;
#pragma unroll
                for (int u = 0; u < MBAVO_FIN_INFLIGHT; ++u)
                    if (t + u * LANES < t1) acc += v[u];
            }
#else
            int t = t0 + l;
            for (; t + 3 * LANES < t1; t += 4 * LANES)
            { // four loads in flight, added in tile order
                const double a = ld_part(pp + (size_t)t * PS + e), b = ld_part(pp + (size_t)(t + LANES) * PS + e);
                const double c = ld_part(pp + (size_t)(t + 2 * LANES) * PS + e), dd = ld_part(pp + (size_t)(t + 3 * LANES) * PS + e);
                acc += a; acc += b; acc += c; acc += dd;
            }
            for (; t < t1; t += LANES) acc += ld_part(pp + (size_t)t * PS + e);
#endif
        }
        if (LANES > 1)
        {
            if (mine) scratch[l * EPAD + e] = acc;
            __syncthreads();
            if (mine && l == 0)
                for (int j = 1; j < LANES; ++j) acc += scratch[j * EPAD + e];
        }
        if (mine && l == 0)
        {
            if (e == 0) { if (oa.valid_out) oa.valid_out[bf] = acc; }
            else if (e == E) oa.frame_blocks[(size_t)bf * E] = acc; // cost: patch costs are already scaled
            else oa.frame_blocks[(size_t)bf * E + e] = acc * inv; // (the caller's copy: the scale may live in host memory)
            if (e != 0 && oa.systems != nullptr) // (one frame per problem: slot bf IS problem bf)
                store_system_entry<KD>(oa.systems + (size_t)bf * (1 + 6 * KD + 36 * KD * KD), e, e == E ? acc : acc * inv);
        }
        }
        // the frame block (pinned host memory) must land before the completion word does: every wave's stores are performed
        // at workgroup scope before the barrier, ONE thread then fences at system scope (a system-scope fence by all 768
        // threads was 3 of the 4.8 us this epilogue took inside the persistent kernel)
        if (to_host)
        {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wave's frame-block stores are acknowledged
        }
        __syncthreads();
        if (threadIdx.x == 0)
        {
            if (n_tiles > 1) __hip_atomic_store(&oa.tickets[bf], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // ready for the next evaluation
            if (to_host) __threadfence_system(); // THIS slot's frame block is on its way to the host before the slot counts as done
            // (one slot -- one frame, the tracker's case -- needs no count: a device-memory round trip less before the word)
            if (to_host && (oa.nbf == 1 || atomicAdd(oa.slots_done, 1) == oa.nbf - 1))
            {
                if (oa.nbf > 1) __hip_atomic_store(oa.slots_done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if defined(MBAVO_PERSIST_STAMPS) // the summing workgroup's own times beside the completion word (persistent_wait prints their means)
                oa.host_flag[2] = oa.t_seen; oa.host_flag[3] = stamp_area()[0]; oa.host_flag[4] = stamp_area()[1];
                oa.host_flag[5] = stamp_area()[2]; oa.host_flag[1] = __builtin_amdgcn_s_memrealtime();
                for (int i = 4; i < 8; ++i) oa.host_flag[5 + i] = stamp_area()[i]; // (tile sub-stamps: words 9 .. 12)
                for (int i = 0; i < 8; ++i) oa.host_flag[16 + i] = stamp_area()[16 + i]; // (the waves' arrivals at the tile's last barrier)
                __threadfence_system();
#endif
                __hip_atomic_store(oa.host_flag, oa.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); // (ordered by the fence above)
            }
        }
        return true;
    }

    // ------------------------------------------------------------------ fused kernel, sample-parallel variant
    // For SMALL problems (semi-dense keypoints: a few thousand pixels in all) the lane-per-pixel kernel above leaves
    // the machine empty and each lone wave walks its S samples one after the other at dependent-issue speed (one
    // instruction per ~6 cycles): 20 us for a few hundred pixels.  Here a pixel takes S = 2^LOGS adjacent lanes, one
    // per blur sample, so the sample loop is ONE iteration; the per-sample Jacobian contributions meet through the
    // wave's LDS slab and are summed in sample order (the order of the lane-per-pixel kernel), the intensities by
    // shuffles in sample order (bit-identical residuals and Huber costs), and the 64 / S finished rows of a wave go
    // through the same MFMA outer product.  The pose entry of a lane is per lane now (vector loads, L1-resident).
#ifndef MBAVO_SP_WAVES
#define MBAVO_SP_WAVES 8 // (12 until round 3: trackFrame 0.422 -> 0.382 ms per frame with 8 -- two waves per SIMD walk their dependent chains faster than three --, 6: 0.381, 4: 0.390; the semi-dense 4-level evaluation 18.2 us with 12 and with 8, 18.7 with 6, 26.0 with 4)
#endif
    constexpr int kSpWaves = MBAVO_SP_WAVES;
    // LDS of k_fused_sp: the slabs, the cost / valid-count scratch, and -- when it fits the 160 KB -- the frame's S pose
    // entries, staged once per workgroup (a lane's entry is its SAMPLE's; read from global memory it arrives in four
    // or five dependent pieces, each an L2 round trip).  ONE (single launch): the entries are COMPUTED into that area
    // by the workgroup itself, cost-only kernels included.
    template <int KD, bool WITH_J, int LOGS, bool ONE>
    struct SpLds
    {
        static constexpr size_t kBase = ((WITH_J ? (size_t)kSpWaves * OuterAcc<6 * KD + 1>::SLAB : 0) + 4 * kSpWaves) * sizeof(double);
        static constexpr size_t kEntries = ((size_t)1 << LOGS) * sizeof(PoseEntry<KD>);
        static constexpr bool kFits = kBase + kEntries <= 160 * 1024;
        static constexpr bool kStage = (WITH_J || ONE) && kFits;
        // the ticket epilogue needs LANES * EPAD doubles of scratch: the slabs when there are any, else its own area
        // ... and the pose prologue S x (KD - 1) SplineSeg (<= 21 samples per pass): both live in the slabs when there are any
        static constexpr size_t kSegs = (size_t)(((1 << LOGS) < kPoseSPB ? (1 << LOGS) : kPoseSPB) * (KD - 1)) * sizeof(SplineSeg);
        static constexpr size_t kEpilogue = ONE && !WITH_J ? (kSegs > (size_t)kSpWaves * 64 * sizeof(double) ? kSegs : (size_t)kSpWaves * 64 * sizeof(double)) : 0;
        static constexpr size_t kBytes = kBase + (kStage ? kEntries : 0) + kEpilogue;
        // the persistent kernel keeps the slabs of an H / g evaluation for a re-summation (sp_resum_body): its ticket epilogue
        // takes an area of its own there too
        static constexpr size_t kPersistBytes = kBytes + (ONE && WITH_J ? (size_t)kSpWaves * 64 * sizeof(double) : 0);
    };
    // can the single-launch form of the sample-parallel kernel run this (k, S)?  (k = 4, S = 32: the entries do not fit)
    static bool sp_one_fits(int kdeg, int logs)
    {
        if (kdeg == 2) return logs >= 2 && logs <= 5;
        return logs >= 2 && logs <= 4;
    }

    template <int KD, bool WITH_J, int LOGS, bool ONE, bool PERSIST>
    __device__ __forceinline__ bool sp_tile_finish(double *lds, const ProblemDesc &d, int tile_id, int frame, double *__restrict__ partials,
                                                   const OneArgs &oa, double inv);

    // the body of k_fused_sp (one tile of one evaluation); also run, evaluation after evaluation, by the persistent kernel
    template <int KD, bool WITH_J, int HALF_GRAD, int LOGS, bool ONE, bool PERSIST = false>
    __device__ __forceinline__ bool sp_tile_body(double *lds, const ProblemDesc *__restrict__ descs,
                                                 const TileDesc *__restrict__ tiles,
                                                 const PoseEntry<KD> *__restrict__ table,
                                                 double *__restrict__ rho_out,
                                                 double *__restrict__ patch_cost,
                                                 double *__restrict__ patch_blocks_strided,
                                                 double *__restrict__ partials, const OneArgs &oa,
                                                 const double *knots_t_fresh = nullptr, const double *knots_R_fresh = nullptr)
    {
        constexpr int ND = Pack<KD>::ND, E = Pack<KD>::E;
        constexpr int kWavesPerGroup = kSpWaves, kThreads = kWavesPerGroup * 64;
        constexpr int SS = 1 << LOGS, PXW = 64 >> LOGS, PXG = kWavesPerGroup * PXW; // lanes per pixel, pixels per wave / per round
        constexpr int SLAB = OuterAcc<ND>::SLAB;     // doubles per wave: rows, and the parked accumulators at the end
        constexpr int RS = OuterAcc<ND>::STRIDE;     // row stride (>= ND, zero padded)
        double *rows = lds;                                               // [12 waves][64 pixels][ND] (WITH_J only)
        double *red = lds + (WITH_J ? kWavesPerGroup * SLAB : 0);         // [4][waves]: cost, valid pixels; (persistent kernel) the wave's unscaled patch cost, its valid pixels

        const int lane = threadIdx.x & 63;
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int tile_id = xcd_tile_of_block((int)blockIdx.x, (int)gridDim.x);
        const TileDesc tile = tiles[tile_id];
        const ProblemDesc &d = descs[tile.prob];
        if (d.active != nullptr && (*d.active & (WITH_J ? 2 : 1)) == 0) return false; // device-side LM: problem sits this pass out
        const int S = d.S, K = d.K, P = d.P, frame = tile.frame;
        Camera cam;
        cam.fx = d.fx; cam.fy = d.fy; cam.cx = d.cx; cam.cy = d.cy; cam.H = d.H; cam.W = d.W;
        const unsigned char *__restrict__ I_cur = d.cur_imgs[frame];
        const long long pix0 = d.pixel_base + ((long long)frame * K + tile.kp_begin) * P;
        const int npx = tile.kp_count * P;
        constexpr bool STAGE = SpLds<KD, WITH_J, LOGS, ONE>::kStage;
        static_assert(!ONE || STAGE, "the single-launch kernel keeps its pose entries in LDS");
        double *stage = red + 4 * kSpWaves;
        const PoseEntry<KD> *__restrict__ ftab = table + d.pose_base + frame * S; // S == SS (checked by the host)
        if constexpr (ONE)
        {
            SplineSeg *segs = (SplineSeg *)(WITH_J ? rows : stage + SS * (int)(sizeof(PoseEntry<KD>) / sizeof(double)));
            frame_pose_entries<KD, WITH_J>(d, PERSIST ? knots_t_fresh : d.knots_t, PERSIST ? knots_R_fresh : d.knots_R, frame,
                                           (PoseEntry<KD> *)stage, segs, wave, lane, oa.status, tile.kp_begin == 0);
            __syncthreads();
            MBAVO_STAMP(0);
        }
        else if constexpr (STAGE)
        {
            constexpr int EW = (int)(sizeof(PoseEntry<KD>) / sizeof(double));
            const MBAVO_GLOBAL double *src = (const MBAVO_GLOBAL double *)ftab;
            for (int z = threadIdx.x; z < SS * EW; z += kSpWaves * 64) stage[z] = src[z];
            __syncthreads();
        }
        // patch centres use sample S/2 (compute_local_patches_xy.cu:26)
        double mid_rt[3], mid_q[4];
        if constexpr (STAGE)
        {
            const PoseEntry<KD> &m = ((const PoseEntry<KD> *)stage)[SS / 2];
            for (int i = 0; i < 3; ++i) mid_rt[i] = m.rt[i];
            for (int i = 0; i < 4; ++i) mid_q[i] = m.q[i];
        }
        else
        {
            const PoseEntry<KD> &m = ftab[S / 2];
            for (int i = 0; i < 3; ++i) mid_rt[i] = m.rt[i];
            for (int i = 0; i < 4; ++i) mid_q[i] = m.q[i];
        }

        OuterAcc<ND> acc;
        acc.init(lane);
        double *slab = rows + wave * SLAB;
        int nvalid = 0;
        double cost_local = 0.0, x_keep = 0.0;
        const double inv = residual_scale<PERSIST>(d, lane);
        // Power-of-two patches that fit the pixels of one wave round (the 8-pixel pattern at S <= 8): the patch cost is
        // reduced across the wave in the order of the reference's reduce() -- no rho scratch, no second pass.
        const bool wave_patches = (P & (P - 1)) == 0 && P <= PXW;

        const int pw = lane >> LOGS, sidx = lane & (SS - 1), lane0 = lane & ~(SS - 1);
        const unsigned long long gmask = (SS == 64 ? ~0ull : ((1ull << SS) - 1ull)) << lane0;
        const double fS = (double)(float)SS; // A8
        for (int base = 0; base < npx; base += PXG)
        {
            const int g = base + wave * PXW + pw;
            const bool in = g < npx;
            double res = 0.0, w = 0.0, rho = 0.0, cur = 0.0, val = 0.0;
            bool ok_l = false, flagged = false;
            double Jc[WITH_J ? 6 * KD : 1] = {};
            int kp = 0;
            if (in)
            {
                const int kpl = patch_of(g, P), pp = g - kpl * P;
                kp = tile.kp_begin + kpl;
                flagged = d.outlier != nullptr && outlier_flag<PERSIST>(d.outlier, kp) == 1;
                const double kx = d.kp_xy[(size_t)kp * d.kp_stride], ky = d.kp_xy[(size_t)kp * d.kp_stride + 1];
                const double kz = d.kp_z[kp];
                double pcx, pcy;
                patch_centre_rt(mid_rt, mid_q, kx, ky, kz, cam, pcx, pcy);
                const int px = (int)(pcx + d.pattern[2 * pp]); // truncation, A3 (pixel_row)
                const int py = (int)(pcy + d.pattern[2 * pp + 1]);
                if (!(px < 0 || px > cam.W - 1 || py < 0 || py > cam.H - 1))
                {
                    cur = (double)((const MBAVO_GLOBAL unsigned char *)I_cur)[py * cam.W + px];
                    double ray[3];
                    unit_ray(cam, (double)px, (double)py, ray);
                    const double iz = reciprocal(kz + 1e-8);
                    auto one_sample = [&](const PoseEntry<KD> &pe) {
                        SampleInFlight f;
                        sample_issue<KD, WITH_J, HALF_GRAD>(pe, ray, kz, iz, cam, d.ref_img, d.ref_dIxy, f);
                        ok_l = f.taps.ok;
                        sample_retire<KD, WITH_J, false, HALF_GRAD>(pe, f, ray, kz, iz, cam, val, Jc);
                    };
                    if constexpr (STAGE)
                        one_sample(((const PoseEntry<KD> *)stage)[sidx]); // LDS
                    else
                        one_sample(ftab[sidx]);
                }
            }
            MBAVO_STAMP(4); // (thread 0's sample done: keypoint, centre, taps, Jacobian terms)
            // the pixel is valid iff all its samples are (A9); intensities summed in sample order
            const bool valid = in && (__ballot(ok_l) & gmask) == gmask;
            double isum = 0.0;
#pragma unroll
            for (int j = 0; j < SS; ++j) isum += __shfl(val, lane0 + j, 64);
            if (valid) res = quotient(isum, fS) - cur;
            huber_weight(res, d.huber_a, w, rho);
            MBAVO_STAMP(5);
            if (wave_patches)
            { // b[i] += b[i + s], s = P/2 .. 1 (reduction.h:43-54): pixel pw + s sits SS * s lanes up
                double x = rho;
                for (int st = P >> 1; st >= 1; st >>= 1) x = x + __shfl(x, lane + st * SS, 64);
                if (in && sidx == 0)
                {
                    if ((pw & (P - 1)) == 0)
                    {
                        const double c = x * inv;
                        const long long patch = (long long)frame * K + kp;
                        if (patch_cost) patch_cost[d.patch_base + patch] = c;
                        if (patch_blocks_strided) patch_blocks_strided[patch * E] = c;
                        if (!flagged) cost_local += c;
                        if constexpr (PERSIST && WITH_J) x_keep = x; // (sp_resum_body: one patch per wave, lane 0's)
                    }
                    nvalid += valid ? 1 : 0;
                }
            }
            else if (in && sidx == 0)
            {
                rho_out[pix0 + g] = rho;
                nvalid += valid ? 1 : 0;
            }
            const bool keep = valid && !flagged;
            if (WITH_J)
            {
                // 1. every lane parks its sample's contribution; 2. lane (pixel, j) sums entries j, j + S, ... over the
                // samples in order; 3. the pixel's weighted row replaces row `pw` of the slab; 4. MFMA over the rows
                double *mine = slab + lane * RS;
#pragma unroll
                for (int i = 0; i < 6 * KD; ++i) mine[i] = Jc[i];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                constexpr int NOUT = (6 * KD + SS - 1) / SS;
                // dropped pixels park ZERO rows by a select (their contributions may be non-finite, see sp_round_rt)
                const double invS = 1.0 / fS;
                double outv[NOUT];
#pragma unroll
                for (int t = 0; t < NOUT; ++t)
                {
                    const int i = sidx + t * SS;
                    double a = 0.0;
                    if (i < 6 * KD)
                    {
#pragma unroll
                        for (int j = 0; j < SS; ++j) a += slab[(lane0 + j) * RS + i];
                    }
                    outv[t] = keep ? w * (a * invS) : 0.0;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                double *row = slab + pw * RS;
                if (sidx == 0) row[0] = keep ? w * res : 0.0;
#pragma unroll
                for (int t = 0; t < NOUT; ++t)
                {
                    const int i = sidx + t * SS;
                    if (i < 6 * KD) row[1 + i] = outv[t];
                }
                if (PXW < 4) // pad to the 4 rows of one MFMA step
                    for (int z = lane; z < (4 - PXW) * RS; z += 64) slab[PXW * RS + z] = 0.0;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                acc.accumulate(slab, lane, (PXW < 4 ? 4 : PXW) / 4);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            MBAVO_STAMP(6);
        }

        // per-patch cost = slot 0 of the reference's patch block (:232-238), and the
        // tile's share of the frame cost (outlier patches skipped, :265-272)
        if (!wave_patches)
        { // the patches' pixels were handled by other waves: their rho values are read back
            __syncthreads();
            for (int kpl = threadIdx.x; kpl < tile.kp_count; kpl += kThreads)
            {
                const double *r = rho_out + pix0 + (long long)kpl * P;
                const double c = patch_rho_sum(r, P) * inv; // reduction.h order
                const int kp = tile.kp_begin + kpl;
                const long long patch = (long long)frame * K + kp;
                if (patch_cost) patch_cost[d.patch_base + patch] = c;
                if (patch_blocks_strided) patch_blocks_strided[patch * E] = c;
                if (!(d.outlier != nullptr && d.outlier[kp] == 1)) cost_local += c;
            }
        }
        const double wc = wave_sum(cost_local);
        const double wv = wave_sum((double)nvalid);
        if (lane == 0)
        {
            red[wave] = wc; red[kWavesPerGroup + wave] = wv;
            if constexpr (PERSIST && WITH_J) { red[2 * kWavesPerGroup + wave] = x_keep; red[3 * kWavesPerGroup + wave] = wv; }
        }
        // every wave parks its accumulators in its slab; entry e = (i, j) of the packed block is then the sum
        // over waves (and pixel groups) in a fixed order.  One barrier for the scratch and the slabs.
        if (WITH_J) acc.store(slab, lane);
#if defined(MBAVO_PERSIST_STAMPS)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (lane == 0) stamp_area()[16 + wave] = __builtin_amdgcn_s_memrealtime(); // (each wave's arrival at the tile's last barrier)
#endif
        return sp_tile_finish<KD, WITH_J, LOGS, ONE, PERSIST>(lds, d, tile_id, frame, partials, oa, inv);
    }

    // the end of a tile (after the waves parked their sums in `red` and their accumulators in the slabs): the tile's partial
    // block, and in the single-launch forms the ticket of its slot
    template <int KD, bool WITH_J, int LOGS, bool ONE, bool PERSIST>
    __device__ __forceinline__ bool sp_tile_finish(double *lds, const ProblemDesc &d, int tile_id, int frame, double *__restrict__ partials,
                                                   const OneArgs &oa, double inv)
    {
        constexpr int ND = Pack<KD>::ND, E = Pack<KD>::E, PS = Pack<KD>::PSTRIDE;
        constexpr int kWavesPerGroup = kSpWaves, kThreads = kWavesPerGroup * 64, SS = 1 << LOGS;
        double *rows = lds, *red = lds + (WITH_J ? kWavesPerGroup * OuterAcc<ND>::SLAB : 0), *stage = red + 4 * kSpWaves;
        __syncthreads();
        MBAVO_STAMP(7);
        double *out = partials + (size_t)tile_id * PS;
        auto put = [&](int e, double v) { out[e] = v; };
        if (threadIdx.x == 0)
        {
            double c = 0.0, v = 0.0;
            for (int i = 0; i < kWavesPerGroup; ++i) { c += red[i]; v += red[kWavesPerGroup + i]; }
            put(0, v);
            put(E, c);
        }
        if (WITH_J)
        {
            for (int e = 1 + threadIdx.x; e < E; e += kThreads)
            {
                int i, j;
                tri_decode(e, ND, i, j);
                put(e, OuterAcc<ND>::gather(rows, i, j, kWavesPerGroup));
            }
        }
        if constexpr (ONE)
        {
            // scratch for the tile-lane sums: the slabs (every thread is past its gather after the barrier inside), or the
            // kernel's epilogue area when there are none -- or when they are kept (persistent kernel: sp_resum_body)
            double *scratch = WITH_J && !PERSIST ? rows : stage + SS * (int)(sizeof(PoseEntry<KD>) / sizeof(double));
            MBAVO_STAMP(1);
            return ticket_finalize<KD, WITH_J, kThreads>(d, d.bf_base + frame, partials, oa, scratch, inv);
        }
        return false;
    }

    // RE-SUMMATION of the H / g evaluation this workgroup ran LAST, under outlier flags and a residual scale the host changed since
    // (round 5; persistent kernel, command mode 3).  An accepted LM step is followed by an H / g evaluation at the very knots of
    // the candidate just evaluated (blur_aware_direct_tracker.cpp:896-903); what differs is detectOutliers' doing (:639-699): a few
    // more patches flagged, hence another 1 / ((K - bad) F P).  Neither enters a pixel's row: a flagged patch's rows are parked as
    // zeros, the scale is applied to the finished sums.  Where a wave's accumulators hold exactly ONE patch (P pixels == the pixels
    // of a wave, tiles of one round -- Engine::persistent_resum_ok) they are still in its slab: the newly flagged patches' slabs are
    // zeroed (what their waves would have accumulated: 0 + 0 * 0), the waves' cost shares rebuilt from the kept unscaled patch
    // costs, and the tile ends as it did before -- same partial sums in the same order, bit for bit the evaluation the reference
    // runs, without pose entries, taps or Jacobians (~6 of its ~13.5 us).
    template <int KD, int LOGS>
    __device__ __forceinline__ bool sp_resum_body(double *lds, const ProblemDesc *__restrict__ descs, const TileDesc *__restrict__ tiles,
                                                  double *__restrict__ patch_cost, double *__restrict__ partials, const OneArgs &oa)
    {
        constexpr int ND = Pack<KD>::ND;
        constexpr int kWavesPerGroup = kSpWaves, PXW = 64 >> LOGS, SLAB = OuterAcc<ND>::SLAB, NI = OuterAcc<ND>::NI;
        double *rows = lds, *red = lds + kWavesPerGroup * SLAB;
        const int lane = threadIdx.x & 63;
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int tile_id = xcd_tile_of_block((int)blockIdx.x, (int)gridDim.x);
        const TileDesc tile = tiles[tile_id];
        const ProblemDesc &d = descs[tile.prob];
        MBAVO_STAMP(0);
        const double inv = residual_scale<true>(d, lane);
        const bool has = wave < tile.kp_count; // (P == PXW: wave w holds patch kp_begin + w)
        const int kp = tile.kp_begin + wave;
        const bool flagged = has && d.outlier != nullptr && outlier_flag<true>(d.outlier, kp) == 1;
        if (flagged)
        {
            double *slab = rows + wave * SLAB;
#pragma unroll
            for (int i = 0; i < NI; ++i) slab[i * 64 + lane] = 0.0;
        }
        if (lane == 0)
        {
            const double c = red[2 * kWavesPerGroup + wave] * inv;
            if (has && patch_cost) patch_cost[d.patch_base + (long long)tile.frame * d.K + kp] = c;
            red[wave] = has && !flagged ? c : 0.0;
            red[kWavesPerGroup + wave] = red[3 * kWavesPerGroup + wave];
        }
        (void)PXW;
        return sp_tile_finish<KD, true, LOGS, true, true>(lds, d, tile_id, tile.frame, partials, oa, inv);
    }

    template <int KD, bool WITH_J, int HALF_GRAD, int LOGS, bool ONE>
    __global__ __launch_bounds__((kSpWaves * 64)) void k_fused_sp(const ProblemDesc *__restrict__ descs,
                                                        const TileDesc *__restrict__ tiles,
                                                        const PoseEntry<KD> *__restrict__ table,
                                                        double *__restrict__ rho_out,
                                                        double *__restrict__ patch_cost,
                                                        double *__restrict__ patch_blocks_strided,
                                                        double *__restrict__ partials, const OneArgs oa)
    {
        extern __shared__ __attribute__((aligned(16))) double lds[];
        (void)sp_tile_body<KD, WITH_J, HALF_GRAD, LOGS, ONE>(lds, descs, tiles, table, rho_out, patch_cost, patch_blocks_strided, partials, oa);
    }

    // ------------------------------------------------------------------ persistent evaluation (host-driven LM loop)
    // One launch per pyramid level instead of one per evaluation: the workgroups of the level's tiles stay resident and
    // take COMMANDS from a pinned host block -- the host writes the next knots (pinned), the outlier flags (pinned), the
    // residual scale (pinned) and then {mode, seq}; thread 0 of every workgroup polls seq with system-scope loads, the
    // workgroup runs the single-launch body (pose prologue, tile, ticket finalize into pinned host memory) and the last
    // slot's workgroup publishes seq in the completion word the host spins on.  What a launch costs per evaluation (~4 us
    // of host enqueue + ~6 us until the kernel starts) is paid once per level.  A workgroup gives up after ~2 s without a
    // command (s_memrealtime, 100 MHz), so a dead host cannot hang the device.
#ifndef MBAVO_PERSIST_SLEEP
#define MBAVO_PERSIST_SLEEP 8
#endif
    // ONE 8-byte word carries the whole command (round 3: sequence number, mode and generation used to be three words, i.e. two
    // more dependent device-memory round trips before a workgroup could start): the host stores it AFTER the inputs are in place.
    //   bits 24..63 sequence number (40 bits)   bits 20..23 second problem (below)   bits 8..19 generation (12 bits): which launch the
    //   command is for (workgroups of an earlier launch that have not seen their exit command yet leave when they meet a command of
    //   a later generation; joint mode starts one launch per tracked frame, so the field wraps every 4096 frames -- ADVICE r05: the
    //   second problem's bits come out of the sequence field, not out of this one)
    //   bits 4..7 problem of the kernel's list the command is for (the pyramid level when one kernel serves all levels)
    //   bits 0..3 mode: 0 = exit, 1 = cost-only evaluation, 2 = H/g evaluation, 3 = the last H/g evaluation summed again under the
    //   flags and the scale as they are now (sp_resum_body)
    struct PersistCmd
    {
        unsigned long long word;
    };
    // (round 5) `prob2` < 15: a SECOND problem of the kernel's list evaluated by the same command, with H / g, at the knots of the
    // second knot area -- the next pyramid level's first evaluation riding along with this level's candidate (tracker.cpp); its
    // completion word is the second line of the pinned flag area.  15: none.
    static inline unsigned long long persist_word(unsigned long long seq, int gen, int mode, int prob = 0, int prob2 = 15)
    {
        return (seq << 24) | ((unsigned long long)(prob2 & 0xf) << 20) | ((unsigned long long)(gen & 0xfff) << 8) | ((unsigned long long)(prob & 0xf) << 4) |
               (unsigned long long)(mode & 0xf);
    }
    template <int KD, int LOGS>
    __global__ __launch_bounds__((kSpWaves * 64)) void k_sp_persist(const ProblemDesc *__restrict__ descs, const TileDesc *__restrict__ tiles,
                                                                  double *__restrict__ rho_out, double *__restrict__ patch_cost,
                                                                  double *__restrict__ partials, OneArgs oa,
                                                                  const PersistCmd *__restrict__ cmd, unsigned long long last_seq, int gen)
    {
        extern __shared__ __attribute__((aligned(16))) double lds[];
        __shared__ unsigned long long s_seq;
        __shared__ int s_mode, s_second;
        __shared__ double knots_lds[7 * 16]; // this command's control knots [t (3N) | R (4N)], N <= 16
        // One kernel may serve SEVERAL problems (round 3: the pyramid levels of a tracked frame, one launch per frame instead of
        // one per level): a command names its problem, the workgroups of the other problems' tiles skip it.
        const int my_prob = tiles[xcd_tile_of_block((int)blockIdx.x, (int)gridDim.x)].prob;
        for (;;)
        {
            if (threadIdx.x == 0)
            {
                unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                unsigned long long q;
                int m = 0, pr = 0, second = 0;
                for (;;)
                { // relaxed: an acquire here would invalidate the caches on every poll
                    const unsigned long long w = __hip_atomic_load(&cmd->word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    q = w >> 24;
                    if (q != last_seq)
                    {
                        m = (int)(w & 0xf);
                        pr = (int)((w >> 4) & 0xf);
                        if ((int)((w >> 8) & 0xfff) != (gen & 0xfff)) m = 0;
                        second = m != 0 && (int)((w >> 20) & 0xf) == my_prob && pr != my_prob; // the command's second problem: H / g
                        if (second) m = 2;
                        if (m == 0 || pr == my_prob || second) break;
                        last_seq = q; // another problem's evaluation: not for this workgroup
                        t0 = __builtin_amdgcn_s_memrealtime(); // ... but the host is alive: the give-up timer starts over
                    }
                    if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) { m = 0; break; } // ~2 s: give up
                    __builtin_amdgcn_s_sleep(MBAVO_PERSIST_SLEEP);
                }
                s_seq = q;
                s_mode = m;
                s_second = second;
            }
            __syncthreads();
            const int mode = s_mode;
            last_seq = s_seq;
            if (mode == 0) return;
            // nothing read from memory may be carried over from the previous command, and nothing should be: values
            // hoisted out of this loop stay live across both bodies and spill
            asm volatile("" ::: "memory");
            // The inputs the host rewrote (it wrote them BEFORE the command word, sfence in between): the knots go into
            // LDS through cache-bypassing loads, the residual scale and the outlier flags are read the same way where they
            // are used.  No acquire fence: the images, keypoints and descriptors stay in the caches across commands.
            const ProblemDesc &d0 = descs[my_prob];
            double *kn = knots_lds;
            // (the second problem of a command reads the SECOND knot area, 7 N doubles behind the first, and answers on the second
            // line of the flag area)
            const int second = s_second, koff = second ? 7 * d0.N : 0;
            for (int i = threadIdx.x; i < 7 * d0.N; i += kSpWaves * 64)
                kn[i] = __hip_atomic_load((i < 3 * d0.N ? d0.knots_t + i : d0.knots_R + (i - 3 * d0.N)) + koff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            OneArgs oa_cmd = oa;
            OneArgs &oa = oa_cmd; // (shadows the kernel argument for the body below)
            oa.nbf = d0.F; // the slots of THIS problem's evaluation
            oa.seq = last_seq;
            if (second) { oa.host_flag += 8; oa.slots_done += 1; }
#if defined(MBAVO_PERSIST_STAMPS)
            oa.t_seen = __builtin_amdgcn_s_memrealtime();
#endif
            if (mode == 3)
                (void)sp_resum_body<KD, LOGS>(lds, descs, tiles, patch_cost, partials, oa);
            else if (mode == 2)
                (void)sp_tile_body<KD, true, false, LOGS, true, true>(lds, descs, tiles, nullptr, rho_out, patch_cost, nullptr, partials, oa, kn, kn + 3 * d0.N);
            else
                (void)sp_tile_body<KD, false, false, LOGS, true, true>(lds, descs, tiles, nullptr, rho_out, patch_cost, nullptr, partials, oa, kn, kn + 3 * d0.N);
            __syncthreads(); // LDS (and s_seq / s_mode) are reused by the next command
        }
    }


    // ------------------------------------------------------------------ finalize
    // Sum of the tile partials of one (problem, frame) in a fixed order: 16 tile-lanes each add
    // every 16th tile, then a fixed tree over the tile-lanes.  grid = (nBF, ceil((E+1)/16)),
    // block = 16 entries x 16 tile-lanes.  Slot E of a partial is the tile's cost share, slot 0 its
    // valid-pixel count.
    template <int KD, bool WITH_J>
    __global__ __launch_bounds__(256) void k_finalize(const ProblemDesc *__restrict__ descs,
                                                      const int *__restrict__ bf_prob,
                                                      const int *__restrict__ bf_tile_begin,
                                                      const double *__restrict__ partials,
                                                      double *__restrict__ frame_blocks,
                                                      double *__restrict__ valid_out, double *__restrict__ systems)
    {
        constexpr int E = Pack<KD>::E, PS = Pack<KD>::PSTRIDE;
        __shared__ double sm[16][17];
        const int bf = blockIdx.x;
        const ProblemDesc &pd = descs[bf_prob[bf]];
        if (pd.active != nullptr && (*pd.active & (WITH_J ? 2 : 1)) == 0) return; // outputs keep their previous values
        const int el = threadIdx.x & 15, tl = threadIdx.x >> 4;
        const int e = blockIdx.y * 16 + el; // partial slot 0..E
        const int t0 = bf_tile_begin[bf], t1 = bf_tile_begin[bf + 1];
        double s = 0.0;
        if (e <= E && (WITH_J || e == 0 || e == E))
        { // same order as a plain loop; four loads in flight (the partials come from other XCDs: every load misses L2;
          // sixteen in flight measured the same, 41.9 vs 42.0 us per dense step: the launch itself is what is left)
            int t = t0 + tl;
            for (; t + 48 < t1; t += 64)
            {
                const double a = partials[(size_t)t * PS + e], b = partials[(size_t)(t + 16) * PS + e];
                const double c = partials[(size_t)(t + 32) * PS + e], d = partials[(size_t)(t + 48) * PS + e];
                s += a; s += b; s += c; s += d;
            }
            for (; t < t1; t += 16) s += partials[(size_t)t * PS + e];
        }
        sm[tl][el] = s;
        __syncthreads();
        for (int h = 8; h >= 1; h >>= 1)
        {
            if (tl < h) sm[tl][el] += sm[tl + h][el];
            __syncthreads();
        }
        if (tl == 0 && e <= E)
        {
            const double v = sm[0][el];
            if (e == 0) { if (valid_out) valid_out[bf] = v; }
            else if (e == E) frame_blocks[(size_t)bf * E] = v; // cost: patch costs are already scaled
            else if (WITH_J) frame_blocks[(size_t)bf * E + e] = v * (pd.inv_ptr != nullptr ? *pd.inv_ptr : pd.inv_num_residuals);
            if (WITH_J && e != 0 && systems != nullptr)
                store_system_entry<KD>(systems + (size_t)bf_prob[bf] * (1 + 6 * KD + 36 * KD * KD), e,
                                       e == E ? v : v * (pd.inv_ptr != nullptr ? *pd.inv_ptr : pd.inv_num_residuals));
        }
    }

    // The same sums for batches of MANY (problem, frame) slots with at most four tiles each (hundreds of semi-dense
    // pairs): one block per slot, one thread per packed entry, the tiles added in the order the tree above gives them
    // ((t0 + t2) + (t1 + t3), absent tiles = +0) -- 512 blocks instead of 512 x 21 (8.6 -> 4.5 us on 512 pairs).
    template <int KD, bool WITH_J>
    __global__ __launch_bounds__(384) void k_finalize_flat(const ProblemDesc *__restrict__ descs,
                                                           const int *__restrict__ bf_prob,
                                                           const int *__restrict__ bf_tile_begin,
                                                           const double *__restrict__ partials,
                                                           double *__restrict__ frame_blocks,
                                                           double *__restrict__ valid_out, double *__restrict__ systems)
    {
        constexpr int E = Pack<KD>::E, PS = Pack<KD>::PSTRIDE;
        const int bf = blockIdx.x, e = threadIdx.x; // partial slot 0..E
        if (e > E || !(WITH_J || e == 0 || e == E)) return;
        const ProblemDesc &pd = descs[bf_prob[bf]];
        if (pd.active != nullptr && (*pd.active & (WITH_J ? 2 : 1)) == 0) return; // outputs keep their previous values
        const int t0 = bf_tile_begin[bf], n = bf_tile_begin[bf + 1] - t0;
        double t[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < n) t[i] = partials[(size_t)(t0 + i) * PS + e];
        const double v = (t[0] + t[2]) + (t[1] + t[3]);
        if (e == 0) { if (valid_out) valid_out[bf] = v; }
        else if (e == E) frame_blocks[(size_t)bf * E] = v; // cost: patch costs are already scaled
        else frame_blocks[(size_t)bf * E + e] = v * (pd.inv_ptr != nullptr ? *pd.inv_ptr : pd.inv_num_residuals);
        if (WITH_J && e != 0 && systems != nullptr)
            store_system_entry<KD>(systems + (size_t)bf_prob[bf] * (1 + 6 * KD + 36 * KD * KD), e,
                                   e == E ? v : v * (pd.inv_ptr != nullptr ? *pd.inv_ptr : pd.inv_num_residuals));
    }

    // ------------------------------------------------------------------ host driver
    Engine::Engine(int device) : device_(device)
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
            num_cus_ = prop.multiProcessorCount;
    }

    Engine *Engine::companion()
    {
        if (!companion_) companion_ = new Engine(device_);
        companion_->set_stream(stream_);
        companion_->opts_ = opts_; // (plain copy: set_options would drop its layout)
        return companion_;
    }

    int Engine::prepare(int B, const mbavo_problem *probs, int kdeg, const int *d_active, const double *d_inv)
    {
        if (B < 1 || !probs || (kdeg != 2 && kdeg != 4) || persist_mask_) return MBAVO_E_ARG;
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess || cur != device_) HIP_TRY(hipSetDevice(device_));
        const int rc = rebuild_layout(B, probs, kdeg, d_active, d_inv);
        if (rc == 0) { same_list_probs_ = probs; same_list_gen_ = layout_gen_; }
        return rc;
    }

    Engine::~Engine()
    {
        delete companion_;
        (void)persistent_end_all();
        (void)comm_destroy();
        (void)p2p_destroy();
        void *bufs[] = {d_layout_, d_poses_, d_rho_, d_partials_,
                        d_status_, d_tickets_};
        if (h_flag_) (void)hipHostFree(h_flag_);
        if (d_push_) (void)hipFree(d_push_);
        for (void *p : bufs)
            if (p) (void)hipFree(p);
        for (ParkedLayout &s : parked_)
            if (s.d_layout) (void)hipFree(s.d_layout);
        if (h_fb_) (void)hipHostFree(h_fb_);
        for (void *q : pinned_)
            if (q) (void)hipHostFree(q);
        for (void *p : slots_)
            if (p) (void)hipFree(p);
        for (hipEvent_t e : prof_ev_) (void)hipEventDestroy(e);
    }

    int Engine::ensure(void **ptr, size_t *cap, size_t bytes)
    {
        if (bytes <= *cap && *ptr) return 0;
        if (*ptr) HIP_TRY(hipFree(*ptr));
        *ptr = nullptr;
        size_t want = bytes + bytes / 4 + 256;
        HIP_TRY(hipMalloc(ptr, want));
        *cap = want;
        return 0;
    }

    void *Engine::named_scratch(int slot, size_t bytes)
    {
        if (slot < 0 || slot >= kSlots) return nullptr;
        if (ensure(&slots_[slot], &slot_cap_[slot], bytes ? bytes : 1) != 0) return nullptr;
        return slots_[slot];
    }

    double *Engine::host_frame_blocks(size_t n)
    {
        if (n * sizeof(double) > cap_hfb_ || !h_fb_)
        {
            if (h_fb_) (void)hipHostFree(h_fb_);
            h_fb_ = nullptr;
            cap_hfb_ = n * sizeof(double) + 4096;
            if (hipHostMalloc(&h_fb_, cap_hfb_, hipHostMallocDefault) != hipSuccess) { h_fb_ = nullptr; cap_hfb_ = 0; }
        }
        return (double *)h_fb_;
    }

    void *Engine::pinned_scratch(int slot, size_t bytes)
    {
        if (slot < 0 || slot >= kPinnedSlots) return nullptr;
        if (bytes > pinned_cap_[slot] || !pinned_[slot])
        {
            if (pinned_[slot]) (void)hipHostFree(pinned_[slot]);
            pinned_[slot] = nullptr;
            pinned_cap_[slot] = bytes + 4096;
            if (hipHostMalloc(&pinned_[slot], pinned_cap_[slot], hipHostMallocDefault) != hipSuccess) { pinned_[slot] = nullptr; pinned_cap_[slot] = 0; }
        }
        return pinned_[slot];
    }

    // The engine's scheduling choices: mbavo_engine_opts (set_options) under the environment's override layer (options.h).
    EngineTuning Engine::tuning() const
    {
        const EnvOverrides e = read_env_overrides();
        EngineTuning t;
        t.sample_parallel = e.sp != kEnvUnset ? (e.sp == 0 ? 0 : (e.sp == 1 ? 1 : -1)) : (opts_.sample_parallel == 0 ? -1 : (opts_.sample_parallel > 0 ? 1 : 0));
        t.single_launch = opt_flag(opts_.single_launch, e.one, true);
        t.fused_pose = opt_flag(opts_.fused_pose, e.fused_pose, true);
        t.fused_pose_max_samples = opt_number(opts_.fused_pose_max_samples, e.fused_pose_max_s, 8);
        t.persistent = opt_flag(opts_.persistent, e.persist, true);
        t.prelaunch = opt_flag(opts_.prelaunch, e.prelaunch, true);
        t.tiles_per_cu = opt_number(opts_.tiles_per_cu, e.tiles_per_cu, 1);
        t.min_tile_pixels = opt_number(opts_.min_tile_pixels, e.min_tile_px, 256);
        t.sp_max_slot_tiles = opt_number(opts_.sp_max_slot_tiles, e.sp_max_slot_tiles, 64);
        return t;
    }

    void Engine::swap_layout(ParkedLayout &s)
    {
        h_descs_.swap(s.descs); h_tiles_.swap(s.tiles); h_bf_tile_begin_.swap(s.bf_tile_begin); h_bf_prob_.swap(s.bf_prob);
        h_entry_prob_.swap(s.entry_prob);
        std::swap(cached_kdeg_, s.kdeg); std::swap(total_bf_, s.total_bf); std::swap(total_entries_, s.total_entries);
        std::swap(sp_logs_, s.sp_logs); std::swap(total_pixels_, s.total_pixels); std::swap(total_patches_, s.total_patches);
        std::swap(layout_uploaded_, s.uploaded); std::swap(flat_finalize_, s.flat_finalize); std::swap(empty_slots_, s.empty_slots);
        std::swap(d_layout_, s.d_layout); std::swap(cap_layout_, s.cap_layout);
        std::swap(d_descs_, s.d_descs); std::swap(d_tiles_, s.d_tiles); std::swap(d_bf_tile_begin_, s.d_bf_tile_begin);
        std::swap(d_bf_prob_, s.d_bf_prob); std::swap(d_entry_prob_, s.d_entry_prob);
        std::swap(layout_key_, s.key);
        ++layout_gen_;
    }

    int Engine::rebuild_layout(int B, const mbavo_problem *probs, int kdeg, const int *d_active, const double *d_inv, bool cached_only)
    {
        std::vector<ProblemDesc> &descs = scratch_descs_; // member: no allocation per call in the LM loop
        descs.resize((size_t)B);
        long long pixels = 0, patches = 0;
        int entries = 0, bf = 0;
        for (int b = 0; b < B; ++b)
        {
            const mbavo_problem &p = probs[b];
            if (p.S < 1 || p.F < 1 || p.K < 0 || p.P < 1 || p.N < kdeg || !p.d_ref_img || !p.d_cur_imgs ||
                !p.d_kp_xy || !p.d_kp_z || !p.d_pattern || !p.d_cap_time || !p.d_exp_time || !p.d_knots_t ||
                !p.d_knots_R || (p.kp_stride != 2 && p.kp_stride != 3) || p.H < 2 || p.W < 2 ||
                (long long)p.H * p.W > (1ll << 29)) // the tap loads use 32-bit byte offsets (8 B per gradient pixel)
                return MBAVO_E_ARG;
            ProblemDesc &d = descs[b];
            memset(&d, 0, sizeof(d));
            d.ref_img = p.d_ref_img; d.ref_dIxy = p.d_ref_dIxy; d.cur_imgs = p.d_cur_imgs;
            d.kp_xy = p.d_kp_xy; d.kp_z = p.d_kp_z; d.pattern = p.d_pattern; d.outlier = p.d_outlier;
            d.cap = p.d_cap_time; d.exp_t = p.d_exp_time; d.knots_t = p.d_knots_t; d.knots_R = p.d_knots_R;
            d.fx = p.intrinsics[0]; d.fy = p.intrinsics[1]; d.cx = p.intrinsics[2]; d.cy = p.intrinsics[3];
            d.t0 = p.t0; d.dt = p.dt; d.huber_a = p.huber_a;
            // 1/((K - num_bad)*F*P), counting out-of-bounds pixels (spline_update_step.cpp:116-117)
            // (a shard of a larger problem carries the whole problem's count, so that the shards' blocks add up)
            const long long num_residuals = p.num_residuals > 0 ? p.num_residuals : (long long)(p.K - p.num_bad) * p.F * p.P;
            d.inv_num_residuals = num_residuals > 0 ? 1.0 / (double)num_residuals : 0.0; // empty problem: all-zero blocks
            d.S = p.S; d.F = p.F; d.K = p.K; d.P = p.P; d.N = p.N; d.H = p.H; d.W = p.W; d.kp_stride = p.kp_stride;
            d.grad_fp16 = p.grad_fp16 == 2 ? 2 : (p.grad_fp16 ? 1 : 0);
            d.active = d_active ? d_active + b : nullptr;
            d.inv_ptr = d_inv ? d_inv + b : nullptr;
            d.pose_base = entries; d.bf_base = bf; d.pixel_base = pixels; d.patch_base = patches;
            entries += p.F * p.S;
            bf += p.F;
            pixels += (long long)p.F * p.K * p.P;
            patches += (long long)p.F * p.K;
        }
        const EngineTuning tune = tuning();
        LayoutKey key;
        key.tile_target = tile_target_; key.tiles_per_cu = tune.tiles_per_cu; key.sample_parallel = tune.sample_parallel;
        key.min_tile_pixels = tune.min_tile_pixels; key.sp_max_slot_tiles = tune.sp_max_slot_tiles;
        const bool same = kdeg == cached_kdeg_ && key == layout_key_ && descs.size() == h_descs_.size() &&
                          memcmp(descs.data(), h_descs_.data(), descs.size() * sizeof(ProblemDesc)) == 0;
        if (same && layout_uploaded_) return 0;
        if (cached_only && B > 8) return 2;
        if (B <= 8)
        { // a parked layout of the same problem list?  Otherwise the active one is parked (oldest slot) and the new one
          // is built over that slot's buffers (stream order protects an arena a queued kernel still reads)
            for (ParkedLayout &s : parked_)
                if (s.uploaded && s.kdeg == kdeg && s.key == key && s.descs.size() == descs.size() &&
                    memcmp(descs.data(), s.descs.data(), descs.size() * sizeof(ProblemDesc)) == 0)
                {
                    swap_layout(s);
                    return 0;
                }
            if (cached_only) return 2; // (the caller must not build / upload now: a persistent kernel holds the stream)
            if (layout_uploaded_ && h_descs_.size() <= 8)
            {
                swap_layout(parked_[parked_victim_]);
                parked_victim_ = (parked_victim_ + 1) % kParkedLayouts;
                layout_uploaded_ = false;
            }
        }

        // tiles: contiguous keypoint ranges of one (problem, frame).  One workgroup is resident per CU (LDS), so
        // the tile count must not exceed CUs x rounds or a nearly empty extra round doubles the time: take the
        // smallest tile size (in pixels) whose tile count fits, found by bisection.
        const int tiles_per_cu = tune.tiles_per_cu;
        const long long target_tiles = tile_target_ > 0 ? tile_target_ : (long long)num_cus_ * (tiles_per_cu > 0 ? tiles_per_cu : 1);
        auto count_tiles = [&](long long ppt) {
            long long n = 0;
            for (int b = 0; b < B; ++b)
            {
                long long kpt = ppt / descs[b].P;
                if (kpt < 1) kpt = 1;
                n += (long long)descs[b].F * ((descs[b].K + kpt - 1) / kpt);
            }
            return n;
        };
        // small problems take the sample-parallel kernel (k_fused_sp): same S = 4 .. 32 everywhere, fp32 gradients, and
        // no more pixels than two of its rounds on every CU
        int sp_logs = 0;
        {
            const int S0 = descs[0].S;
            int lg = 0;
            while ((1 << lg) < S0) ++lg;
            bool ok = (1 << lg) == S0 && lg >= 2 && lg <= 5;
            for (int b = 0; b < B && ok; ++b) ok = descs[b].S == S0 && descs[b].grad_fp16 == 0;
            const long long round_px = (long long)kSpWaves * (64 >> lg);
            const int force = tune.sample_parallel;
            if (ok && force != 0 && (force == 1 || pixels <= 2 * round_px * num_cus_)) sp_logs = lg;
        }
        auto tile_px = [&](int logs) {
            long long lo = logs ? (long long)kSpWaves * (64 >> logs) : tune.min_tile_pixels, hi = pixels > lo ? pixels : lo;
            if (count_tiles(lo) > target_tiles)
            {
                while (lo < hi)
                {
                    const long long mid = (lo + hi) / 2;
                    if (count_tiles(mid) <= target_tiles) hi = mid; else lo = mid + 1;
                }
            }
            return lo;
        };
        long long px_per_tile = tile_px(sp_logs);
        if (sp_logs && tune.sample_parallel != 1)
        { // The single-launch form ends with ONE workgroup summing its slot's partials (ticket_finalize): fine for the handful of
          // tiles of a semi-dense level, slow for a slot of hundreds (a dense 160 x 120 level alone: 253 tiles, 38.1 us against 20.5
          // for the lane-per-pixel kernel + finalize kernel; tools/level_bench.py).  Such lists take the lane-per-pixel kernel.
            long long worst = 0;
            for (int b = 0; b < B; ++b)
            {
                long long kpt = px_per_tile / descs[b].P;
                if (kpt < 1) kpt = 1;
                worst = std::max(worst, (descs[b].K + kpt - 1) / kpt);
            }
            // (the summing workgroup has threads / padded-entries tile-lanes: 4 for k = 2, 1 for k = 4; 64 partials per lane at most)
            const int lanes = kdeg == 2 ? kSpWaves * 64 / 128 : (kSpWaves * 64 >= 384 ? kSpWaves * 64 / 384 : 1);
            if (worst > (long long)tune.sp_max_slot_tiles * lanes)
            {
                sp_logs = 0;
                px_per_tile = tile_px(0);
            }
        }
        sp_logs_ = sp_logs;
        std::vector<TileDesc> tiles;
        std::vector<int> bf_tile_begin, bf_prob;
        for (int b = 0; b < B; ++b)
        {
            const ProblemDesc &d = descs[b];
            long long kpt = px_per_tile / d.P;
            if (kpt < 1) kpt = 1;
            for (int f = 0; f < d.F; ++f)
            {
                bf_tile_begin.push_back((int)tiles.size());
                bf_prob.push_back(b);
                for (long long k0 = 0; k0 < d.K; k0 += kpt)
                {
                    TileDesc t;
                    t.prob = b; t.frame = f; t.kp_begin = (int)k0;
                    t.kp_count = (int)((d.K - k0) < kpt ? (d.K - k0) : kpt);
                    tiles.push_back(t);
                }
            }
        }
        bf_tile_begin.push_back((int)tiles.size());
        // many slots of at most four tiles each: the one-block-per-slot finalize kernel
        int max_tiles_per_bf = 0;
        for (size_t i = 0; i + 1 < bf_tile_begin.size(); ++i)
            max_tiles_per_bf = std::max(max_tiles_per_bf, bf_tile_begin[i + 1] - bf_tile_begin[i]);
        flat_finalize_ = bf_prob.size() >= 64 && max_tiles_per_bf <= 4;
        empty_slots_ = false;
        for (size_t i = 0; i + 1 < bf_tile_begin.size(); ++i) empty_slots_ = empty_slots_ || bf_tile_begin[i + 1] == bf_tile_begin[i];

        h_descs_ = descs;
        layout_key_ = key;
        ++layout_gen_;
        h_tiles_.swap(tiles);
        h_bf_tile_begin_.swap(bf_tile_begin);
        h_bf_prob_.swap(bf_prob);
        h_entry_prob_.resize(entries > 0 ? entries : 1);
        for (int b = 0; b < B; ++b)
            for (int e = 0; e < h_descs_[b].F * h_descs_[b].S; ++e) h_entry_prob_[h_descs_[b].pose_base + e] = b;
        cached_kdeg_ = kdeg;
        total_bf_ = bf; total_entries_ = entries; total_pixels_ = pixels; total_patches_ = patches;

        int layout_max_S = 1;
        for (const ProblemDesc &pd : h_descs_) layout_max_S = pd.S > layout_max_S ? pd.S : layout_max_S;
        // (one table of max S entries per tile when the fused kernel computes its own: see fused_pose in evaluate)
        const size_t pose_entries = std::max((size_t)entries, h_tiles_.size() <= (size_t)num_cus_ ? h_tiles_.size() * (size_t)layout_max_S : (size_t)0);
        const size_t pose_bytes = pose_entries * (kdeg == 2 ? sizeof(PoseEntry<2>) : sizeof(PoseEntry<4>));
        const size_t pstride = kdeg == 2 ? Pack<2>::PSTRIDE : Pack<4>::PSTRIDE;
        int rc;
        // descriptors, tiles and index tables live in ONE device arena filled by ONE copy: a pageable H2D copy is staged
        // synchronously by the runtime (~2.5 us each), and the layout changes with every pyramid level and every outlier
        // update of the LM loop -- five copies were half of that loop's enqueue time (host phase timers)
        auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
        const size_t o_descs = 0, o_tiles = align(h_descs_.size() * sizeof(ProblemDesc));
        const size_t o_bf = o_tiles + align((h_tiles_.size() + 1) * sizeof(TileDesc));
        const size_t o_bfp = o_bf + align(h_bf_tile_begin_.size() * sizeof(int));
        const size_t o_ep = o_bfp + align((h_bf_prob_.size() + 1) * sizeof(int));
        const size_t layout_bytes = o_ep + align(h_entry_prob_.size() * sizeof(int));
        if ((rc = ensure(&d_layout_, &cap_layout_, layout_bytes))) return rc;
        d_descs_ = (char *)d_layout_ + o_descs; d_tiles_ = (char *)d_layout_ + o_tiles; d_bf_tile_begin_ = (char *)d_layout_ + o_bf;
        d_bf_prob_ = (char *)d_layout_ + o_bfp; d_entry_prob_ = (char *)d_layout_ + o_ep;
        h_layout_.resize(layout_bytes);
        memcpy(h_layout_.data() + o_descs, h_descs_.data(), h_descs_.size() * sizeof(ProblemDesc));
        if (!h_tiles_.empty()) memcpy(h_layout_.data() + o_tiles, h_tiles_.data(), h_tiles_.size() * sizeof(TileDesc));
        memcpy(h_layout_.data() + o_bf, h_bf_tile_begin_.data(), h_bf_tile_begin_.size() * sizeof(int));
        if (!h_bf_prob_.empty()) memcpy(h_layout_.data() + o_bfp, h_bf_prob_.data(), h_bf_prob_.size() * sizeof(int));
        memcpy(h_layout_.data() + o_ep, h_entry_prob_.data(), h_entry_prob_.size() * sizeof(int));
        if ((rc = ensure(&d_poses_, &cap_poses_, pose_bytes + 64))) return rc; // + one cache line: the scalar-cache warm-up reads whole lines
        if ((rc = ensure(&d_rho_, &cap_rho_, (size_t)(pixels + 1) * sizeof(double)))) return rc;
        if ((rc = ensure(&d_partials_, &cap_partials_, (h_tiles_.size() + 1) * pstride * sizeof(double)))) return rc;
        {   // ticket counters of the single-launch kernels: one per (problem, frame) slot + the slots-done counter;
            // zero between launches by construction, so only (re)allocation clears them
            const size_t before = cap_tickets_;
            if ((rc = ensure(&d_tickets_, &cap_tickets_, (size_t)(bf + 2) * sizeof(int)))) return rc; // (+ the second problem's slots-done counter)
            if (cap_tickets_ != before) HIP_TRY(hipMemsetAsync(d_tickets_, 0, cap_tickets_, stream_));
        }
        if (!d_status_)
        { // out-of-range counter: only ever incremented by the pose kernel; fetch_status() reports the delta
            HIP_TRY(hipMalloc(&d_status_, sizeof(int)));
            HIP_TRY(hipMemset(d_status_, 0, sizeof(int)));
        }
        // a pageable copy is staged synchronously by the runtime, so the host buffer may change afterwards
        HIP_TRY(hipMemcpyAsync(d_layout_, h_layout_.data(), layout_bytes, hipMemcpyHostToDevice, stream_));
        layout_uploaded_ = true;
        return 0;
    }

    // Launch with the kernel's own begin / end timestamps attached to an event pair (hipExtLaunchKernelGGL) when this
    // launch is one of the timed ones: the duration is the kernel's, as rocprofv3 reports it, and no barrier packet is
    // added to the queue (an event RECORDED around the launch brackets the dispatch too, +3-4 us, and costs a launch gap)
#define MBAVO_LAUNCH_TIMED(kernel, grid, block, lds_bytes, ...)                                              \
    do                                                                                                      \
    {                                                                                                       \
        hipEvent_t ev0_, ev1_;                                                                              \
        if (eng->prof_events(&ev0_, &ev1_))                                                                 \
            hipExtLaunchKernelGGL(kernel, grid, block, lds_bytes, st, ev0_, ev1_, 0, __VA_ARGS__);          \
        else                                                                                                \
            hipLaunchKernelGGL(kernel, grid, block, lds_bytes, st, __VA_ARGS__);                            \
    } while (0)

    template <int KD, bool WITH_J>
    static int launch_all(Engine *eng, hipStream_t st, int max_S, int grad_mode, int sp_logs, bool one, bool flat_finalize, const ProblemDesc *descs, const int *entry_prob, int entries, const TileDesc *tiles, int ntiles,
                          const int *bf_prob, const int *bf_tile_begin, int nbf, void *poses, double *rho,
                          double *patch_cost, double *patch_blocks_strided, double *partials, int *status,
                          double *frame_blocks, double *valid, const OneArgs &oa, bool fused_pose_ok, bool external_poses)
    {
        PoseEntry<KD> *table = (PoseEntry<KD> *)poses;
        // one workgroup per CU at most, lane-per-pixel kernel: the pose entries are the fused kernel's prologue (k = 2 too since
        // round 4: through the two stages, see frame_pose_entries)
        const bool fused_pose = fused_pose_ok && sp_logs == 0 && !one && ntiles > 0;
        if (one)
        { // single launch: pose entries in the prologue, finalize by the last workgroup of every slot
#define MBAVO_SP_ONE(LG)                                                                                                       \
    do                                                                                                                         \
    {                                                                                                                          \
        if constexpr (SpLds<KD, WITH_J, LG, true>::kFits)                                                                      \
        {                                                                                                                      \
            const size_t lds_sp = SpLds<KD, WITH_J, LG, true>::kBytes;                                                         \
            HIP_TRY(eng->ensure_lds((const void *)k_fused_sp<KD, WITH_J, false, LG, true>, lds_sp));                           \
            MBAVO_LAUNCH_TIMED((k_fused_sp<KD, WITH_J, false, LG, true>), dim3(ntiles), dim3(kSpWaves * 64), lds_sp, descs, tiles, \
                               table, rho, patch_cost, patch_blocks_strided, partials, oa);                                    \
        }                                                                                                                      \
    } while (0)
            switch (sp_logs)
            {
            case 2: MBAVO_SP_ONE(2); break;
            case 3: MBAVO_SP_ONE(3); break;
            case 4: MBAVO_SP_ONE(4); break;
            default: MBAVO_SP_ONE(5); break;
            }
#undef MBAVO_SP_ONE
            HIP_TRY(hipGetLastError());
            return 0;
        }
        if (!fused_pose && !external_poses)
            hipLaunchKernelGGL((k_pose_table<KD, WITH_J>), dim3((entries + kPoseSPB - 1) / kPoseSPB), dim3(64 * KD), 0, st, descs, entry_prob,
                               entries, table, status);
        if (ntiles > 0)
        {
            constexpr int kWavesPerGroup = waves_of<KD, WITH_J>(), kThreads = kWavesPerGroup * 64;
            const size_t lds_plain = (WITH_J ? (size_t)kWavesPerGroup * OuterAcc<Pack<KD>::ND>::SLAB : 0) * sizeof(double) +
                               2 * kWavesPerGroup * sizeof(double)
                ;
            const size_t lds = lds_plain;
            (void)max_S;
            // the large-LDS attribute is per device and per kernel: remembered per engine (= per device)
            // gradient format of the instantiation: cost-only passes of the packed format tap the u8 image like format 0
            // (and are format 0's instantiation); MBAVO_GRAD_CASE runs its statement with G = the compile-time format
            const int gm = (!WITH_J && grad_mode == 2) ? 0 : grad_mode;
#define MBAVO_GRAD_CASE(...)                                                       \
    do                                                                             \
    {                                                                              \
        if (gm == 1) { constexpr int G = 1; __VA_ARGS__; }                         \
        else if (gm == 2) { constexpr int G = WITH_J ? 2 : 0; __VA_ARGS__; }       \
        else { constexpr int G = 0; __VA_ARGS__; }                                 \
    } while (0)
            MBAVO_GRAD_CASE(HIP_TRY(eng->ensure_lds((const void *)k_fused<KD, WITH_J, G>, lds)));
            if (sp_logs > 0)
            {
#define MBAVO_SP_LAUNCH(LG)                                                                                                    \
    do                                                                                                                         \
    {                                                                                                                          \
        const size_t lds_sp = SpLds<KD, WITH_J, LG, false>::kBytes;                                                             \
        HIP_TRY(eng->ensure_lds((const void *)k_fused_sp<KD, WITH_J, false, LG, false>, lds_sp));                               \
        MBAVO_LAUNCH_TIMED((k_fused_sp<KD, WITH_J, false, LG, false>), dim3(ntiles), dim3(kSpWaves * 64), lds_sp, descs, tiles, table, rho, \
                           patch_cost, patch_blocks_strided, partials, oa);                                                    \
    } while (0)
                switch (sp_logs)
                {
                case 2: MBAVO_SP_LAUNCH(2); break;
                case 3: MBAVO_SP_LAUNCH(3); break;
                case 4: MBAVO_SP_LAUNCH(4); break;
                default: MBAVO_SP_LAUNCH(5); break;
                }
#undef MBAVO_SP_LAUNCH
            }
            else if (fused_pose)
            {
                // cost-only: the pose prologue's segments need LDS of their own (there are no slabs to borrow)
                const size_t lds = lds_plain + (WITH_J ? 0 : (size_t)kPoseSPB * (KD - 1) * sizeof(SplineSeg));
                MBAVO_GRAD_CASE(HIP_TRY(eng->ensure_lds((const void *)k_fused<KD, WITH_J, G, true>, lds));
                                MBAVO_LAUNCH_TIMED((k_fused<KD, WITH_J, G, true>), dim3(ntiles), dim3(kThreads), lds, descs, tiles, table, rho,
                                                   patch_cost, patch_blocks_strided, partials, table, status, max_S));
            }
            else
                MBAVO_GRAD_CASE(MBAVO_LAUNCH_TIMED((k_fused<KD, WITH_J, G>), dim3(ntiles), dim3(kThreads), lds, descs, tiles, table, rho, patch_cost,
                                                   patch_blocks_strided, partials, table, status, max_S));
#undef MBAVO_GRAD_CASE
        }
        static_assert(Pack<KD>::E + 1 <= 384, "one thread per partial slot");
        if (eng->take_deferral(flat_finalize)) {} // the caller's kernels sum the tile partials (engine.h: set_defer_finalize)
        else if (flat_finalize)
            hipLaunchKernelGGL((k_finalize_flat<KD, WITH_J>), dim3(nbf), dim3(KD == 2 ? 128 : 384), 0, st, descs, bf_prob, bf_tile_begin,
                               partials, frame_blocks, valid, oa.systems);
        else
            hipLaunchKernelGGL((k_finalize<KD, WITH_J>), dim3(nbf, (Pack<KD>::E + 1 + 15) / 16), dim3(256), 0, st, descs, bf_prob,
                               bf_tile_begin, partials, frame_blocks, valid, oa.systems);
        HIP_TRY(hipGetLastError());
        return 0;
    }

    int Engine::evaluate(int B, const mbavo_problem *probs, int kdeg, bool with_hessian, double *d_frame_blocks,
                         double *d_patch_cost, double *d_valid, double *d_patch_blocks_strided, const int *d_active,
                         const double *d_inv, bool signal_host, bool same_list)
    {
        if (B < 1 || !probs || !d_frame_blocks || (kdeg != 2 && kdeg != 4)) return MBAVO_E_ARG;
        if (d_patch_blocks_strided && B != 1) return MBAVO_E_ARG;
        if (persist_mask_) return MBAVO_E_ARG; // the stream is held by persistent kernels: persistent_end_all() first
        int rc;
        {
            PhaseScope ps_layout(PhaseTimers::kLevel);
            // hipSetDevice costs ~5 us per call on this runtime (measured with the host phase timers: it was two thirds of
            // the enqueue time of an evaluation); hipGetDevice is a thread-local read
            int cur = -1;
            if (hipGetDevice(&cur) != hipSuccess || cur != device_) HIP_TRY(hipSetDevice(device_));
            // same_list: the caller vouches for the list; the engine still checks that it is the list pointer it built this very
            // layout from and that nothing has replaced the layout since
            const bool reuse = same_list && layout_uploaded_ && (int)h_descs_.size() == B && cached_kdeg_ == kdeg && same_list_probs_ == probs &&
                               same_list_gen_ == layout_gen_;
            rc = reuse ? 0 : rebuild_layout(B, probs, kdeg, d_active, d_inv);
            if (rc == 0) { same_list_probs_ = probs; same_list_gen_ = layout_gen_; }
        }
        if (rc) return rc;
        const ProblemDesc *descs = (const ProblemDesc *)d_descs_;
        const TileDesc *tiles = (const TileDesc *)d_tiles_;
        const int ntiles = (int)h_tiles_.size();
        int max_S = 1;
        for (const ProblemDesc &pd : h_descs_) max_S = pd.S > max_S ? pd.S : max_S;
        // the gradient storage format selects the kernel instantiation, so it must be the same for the whole batch
        const int half_grad = h_descs_[0].grad_fp16; // 0 float pairs, 1 IEEE half pairs, 2 packed keyframe words
        for (const ProblemDesc &pd : h_descs_)
            if (pd.grad_fp16 != half_grad) return MBAVO_E_ARG;
        // small problems: ONE launch (pose entries in the fused kernel's prologue, finalize by the last workgroup of a slot)
        // (profiles/r02_single_launch_ab.txt: with the two-stage pose prologue it wins for every spline degree and mode;
        // before it, the k = 4 H/g prologue was a 2 600-instruction chain with vector spills and lost to three launches).
        // MBAVO_ONE=0 forces three launches.
        // A slot without tiles (K == 0: a pyramid level with no surviving keypoint, a keypoint shard of K < world) has no
        // workgroup to finalize it: such lists take the finalize KERNEL, which writes the all-zero block and valid count.
        const EngineTuning tune = tuning();
        const bool one = sp_logs_ > 0 && sp_one_fits(kdeg, sp_logs_) && !empty_slots_ && tune.single_launch;
        OneArgs oa;
        memset(&oa, 0, sizeof(oa));
        // merged systems asked for (set_merge_target): by the finalize step itself when the merge is a plain unpack
        double *const merge_to = with_hessian ? merge_target_ : nullptr;
        merge_target_ = nullptr;
        bool merge_fused = merge_to != nullptr && !defer_finalize_;
        for (const ProblemDesc &pd : h_descs_)
            if (pd.F != 1 || pd.N != kdeg) merge_fused = false;
        merge_fused_last_ = merge_fused;
        oa.systems = merge_fused ? merge_to : nullptr;
        flag_pending_ = false;
        deferred_last_ = false; // (launch_all sets it when it leaves the finalize to the caller)
        if (one)
        {
            oa.bf_tile_begin = (const int *)d_bf_tile_begin_;
            oa.tickets = (int *)d_tickets_;
            oa.slots_done = (int *)d_tickets_ + total_bf_;
            oa.frame_blocks = d_frame_blocks; oa.valid_out = d_valid; oa.status = (int *)d_status_;
            oa.nbf = total_bf_;
            if (signal_host && !d_active && ntiles > 0)
            { // completion word in pinned host memory: the caller spins on it instead of synchronising the stream
                if (!h_flag_ && hipHostMalloc((void **)&h_flag_, 256, hipHostMallocDefault) != hipSuccess) h_flag_ = nullptr;
                if (h_flag_)
                {
                    oa.host_flag = (unsigned long long *)h_flag_;
                    oa.seq = ++flag_seq_;
                    flag_pending_ = true;
                }
            }
        }
        // Pose entries in the fused kernel's prologue (no pose launch) when every workgroup has a CU to itself -- with several
        // tiles per CU the prologue would be paid once per tile -- and the per-workgroup tables fit the pose buffer
        // (rebuild_layout sizes it for that).  MBAVO_FUSED_POSE=0 keeps the pose kernel.
        // Measured (profiles/r02_kfused_experiments.txt 16.): S = 8: the prologue costs a workgroup 4.8 us against the pose
        // kernel's 5.7; S = 16: 7.4 us, slower than the kernel.
        const bool fused_pose_ok = !external_poses_ && ntiles <= num_cus_ && max_S <= tune.fused_pose_max_samples && max_S <= kPoseSPB &&
                                   tune.fused_pose;
#define MBAVO_LAUNCH(KD, WJ)                                                                                      \
    launch_all<KD, WJ>(this, stream_, max_S, half_grad, sp_logs_, one, flat_finalize_, descs, (const int *)d_entry_prob_, total_entries_, tiles, ntiles, (const int *)d_bf_prob_,                 \
                       (const int *)d_bf_tile_begin_, total_bf_, external_poses_ && external_table_ ? external_table_ : d_poses_, (double *)d_rho_, d_patch_cost,        \
                       d_patch_blocks_strided, (double *)d_partials_, (int *)d_status_, d_frame_blocks, d_valid, oa, fused_pose_ok, external_poses_)
        if (kdeg == 4) rc = with_hessian ? MBAVO_LAUNCH(4, true) : MBAVO_LAUNCH(4, false);
        else rc = with_hessian ? MBAVO_LAUNCH(2, true) : MBAVO_LAUNCH(2, false);
#undef MBAVO_LAUNCH
        last_kernel_id_[0] = kdeg; last_kernel_id_[1] = with_hessian; last_kernel_id_[2] = half_grad; last_kernel_id_[3] = sp_logs_;
        last_kernel_id_[4] = one; last_kernel_id_[5] = fused_pose_ok && sp_logs_ == 0 && !one && ntiles > 0;
        if (rc == 0 && merge_to != nullptr && !merge_fused)
        { // several frames per problem or more knots than the spline degree: the gather kernel (multi_gpu.hip) behind the finalize
            if (deferred_last_) return MBAVO_E_ARG; // (no frame blocks were written)
            rc = merge_device(B, probs, kdeg, d_frame_blocks, merge_to);
        }
        return rc;
    }

    // Host side of the push block: write-combining stores through the PCIe BAR, ordered by a store fence.  Only x86-64 is
    // known to behave as the persistent path needs (sfence drains the write-combining buffers in order); elsewhere the
    // push block is not offered and the trackers take one launch per evaluation.
#if defined(__x86_64__)
    static inline void host_store_fence() { __builtin_ia32_sfence(); }
    static inline void host_spin_pause() { __builtin_ia32_pause(); }
    static constexpr bool kHostCanPush = true;
#else
    static inline void host_store_fence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
    static inline void host_spin_pause() {}
    static constexpr bool kHostCanPush = false;
#endif

    void *Engine::push_block(int slot, size_t bytes)
    {
        if (slot < 0 || slot >= kPushSlots || !kHostCanPush) return nullptr;
        if (!tuning().persistent) return nullptr; // opt-out before anything touches device memory from the CPU
        if (push_probe_ == 0)
        { // CPU stores into device memory need the whole VRAM behind the PCIe BAR (large / resizable BAR); without it the
          // allocation below still succeeds and the first store faults
            int cur0 = -1, large = 0;
            if (hipGetDevice(&cur0) != hipSuccess || cur0 != device_) (void)hipSetDevice(device_);
            if (hipDeviceGetAttribute(&large, hipDeviceAttributeIsLargeBar, device_) != hipSuccess) { large = 0; (void)hipGetLastError(); }
            push_probe_ = large ? 1 : -1;
        }
        if (push_probe_ < 0) return nullptr;
        const size_t stride = (bytes + 4095) & ~(size_t)4095;
        if (d_push_ && stride <= push_stride_) return (char *)d_push_ + (size_t)slot * push_stride_;
        if (persist_mask_) return nullptr; // cannot grow under a running kernel
        if (d_push_) { (void)hipFree(d_push_); d_push_ = nullptr; cap_push_ = 0; push_stride_ = 0; }
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess || cur != device_) (void)hipSetDevice(device_);
        const size_t want = stride * kPushSlots;
        if (hipExtMallocWithFlags(&d_push_, want, hipDeviceMallocFinegrained) != hipSuccess) { d_push_ = nullptr; (void)hipGetLastError(); return nullptr; }
        if (hipMemset(d_push_, 0, want) != hipSuccess) { (void)hipFree(d_push_); d_push_ = nullptr; return nullptr; }
        (void)hipDeviceSynchronize();
        cap_push_ = want;
        push_stride_ = stride;
        return (char *)d_push_ + (size_t)slot * push_stride_;
    }
    int Engine::persistent_begin(int slot, const mbavo_problem &p, int kdeg, double *h_frame_blocks, double *h_patch_cost,
                                 const double *h_inv, bool cached_only)
    {
        return persistent_begin(slot, 1, &p, kdeg, h_frame_blocks, h_patch_cost, h_inv, cached_only);
    }

    int Engine::persistent_begin(int slot, int B, const mbavo_problem *probs, int kdeg, double *h_frame_blocks, double *h_patch_cost,
                                 const double *h_inv, bool cached_only)
    {
        if (B < 1 || B > 15 || !probs) return MBAVO_E_ARG;
        const mbavo_problem &p = probs[0];
        if (slot < 0 || slot >= kPushSlots || persistent_active(slot) || !h_frame_blocks || !h_inv || (kdeg != 2 && kdeg != 4)) return MBAVO_E_ARG;
        const EngineTuning tune = tuning();
        if (!tune.persistent || !tune.single_launch) return 1;
        if (!d_push_ || push_stride_ == 0) return 1; // no CPU-writable device memory: the caller takes the per-evaluation launches
        if (cached_only && !tune.prelaunch) return 1;
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess || cur != device_) HIP_TRY(hipSetDevice(device_));
        int rc = rebuild_layout(B, probs, kdeg, nullptr, h_inv, cached_only);
        if (rc == 2) return 1;
        if (rc) return rc;
        const int ntiles = (int)h_tiles_.size();
        if (sp_logs_ <= 0 || !sp_one_fits(kdeg, sp_logs_) || ntiles < 1 || ntiles > num_cus_ || h_descs_[0].grad_fp16 || p.N > 16 || empty_slots_) return 1;
        for (int b = 1; b < B; ++b) // (one knot buffer for all the problems of a persistent kernel)
            if (probs[b].N != p.N || probs[b].d_knots_t != p.d_knots_t || probs[b].d_knots_R != p.d_knots_R) return 1;
        if (!h_flag_ && hipHostMalloc((void **)&h_flag_, 256, hipHostMallocDefault) != hipSuccess) { h_flag_ = nullptr; return (int)hipErrorOutOfMemory; }
        volatile PersistCmd *cmd = (volatile PersistCmd *)((char *)d_push_ + (size_t)slot * push_stride_);
        cmd->word = persist_word(flag_seq_, ++persist_gen_, 0);
        host_store_fence();
        OneArgs oa;
        memset(&oa, 0, sizeof(oa));
        oa.bf_tile_begin = (const int *)d_bf_tile_begin_;
        oa.tickets = (int *)d_tickets_;
        oa.slots_done = (int *)d_tickets_ + total_bf_;
        oa.frame_blocks = h_frame_blocks; oa.valid_out = nullptr; oa.status = (int *)d_status_;
        oa.nbf = total_bf_;
        oa.host_flag = (unsigned long long *)h_flag_;
        hipStream_t st = stream_;
#define MBAVO_PERSIST_LAUNCH(KD, LG)                                                                                              \
    do                                                                                                                            \
    {                                                                                                                             \
        if constexpr (SpLds<KD, true, LG, true>::kFits)                                                                           \
        {                                                                                                                         \
            const size_t lds_sp = SpLds<KD, true, LG, true>::kPersistBytes > SpLds<KD, false, LG, true>::kBytes                   \
                                      ? SpLds<KD, true, LG, true>::kPersistBytes : SpLds<KD, false, LG, true>::kBytes;             \
            HIP_TRY(ensure_lds((const void *)k_sp_persist<KD, LG>, lds_sp));                                                      \
            hipLaunchKernelGGL((k_sp_persist<KD, LG>), dim3(ntiles), dim3(kSpWaves * 64), lds_sp, st, (const ProblemDesc *)d_descs_, \
                               (const TileDesc *)d_tiles_, (double *)d_rho_, h_patch_cost, (double *)d_partials_, oa,             \
                               (const PersistCmd *)cmd, flag_seq_, persist_gen_);                                                  \
        }                                                                                                                         \
    } while (0)
#define MBAVO_PERSIST_K(KD)                                     \
    switch (sp_logs_)                                           \
    {                                                           \
    case 2: MBAVO_PERSIST_LAUNCH(KD, 2); break;                  \
    case 3: MBAVO_PERSIST_LAUNCH(KD, 3); break;                  \
    case 4: MBAVO_PERSIST_LAUNCH(KD, 4); break;                  \
    default: MBAVO_PERSIST_LAUNCH(KD, 5); break;                 \
    }
        if (kdeg == 4) { MBAVO_PERSIST_K(4) } else { MBAVO_PERSIST_K(2) }
#undef MBAVO_PERSIST_K
#undef MBAVO_PERSIST_LAUNCH
        HIP_TRY(hipGetLastError());
        persist_mask_ |= 1u << slot;
        persist_gen_of_[slot] = persist_gen_;
        // problems whose waves hold ONE patch each: as many pixels per patch as a wave has, every tile a single round
        persist_resum_[slot] = 0;
        for (int b = 0; b < B; ++b)
        {
            bool ok = h_descs_[b].P == (64 >> sp_logs_) && h_descs_[b].outlier != nullptr;
            for (const TileDesc &t : h_tiles_) ok = ok && (t.prob != b || t.kp_count <= kSpWaves);
            if (ok) persist_resum_[slot] |= 1u << b;
        }
        last_kernel_id_[0] = kdeg; last_kernel_id_[1] = 1; last_kernel_id_[2] = 0; last_kernel_id_[3] = sp_logs_; last_kernel_id_[4] = 1;
        return 0;
    }

    int Engine::persistent_post(int slot, bool with_hessian, int prob, int prob2)
    {
        if (!persistent_active(slot) || prob < 0 || prob > 14 || prob2 > 14 || prob2 == prob) return MBAVO_E_ARG;
        volatile PersistCmd *cmd = (volatile PersistCmd *)((char *)d_push_ + (size_t)slot * push_stride_);
        const unsigned long long seq = ++flag_seq_;
        host_store_fence(); // the inputs (knots, flags, scale: the push block, write-combining) are out ...
        cmd->word = persist_word(seq, persist_gen_of_[slot], with_hessian ? 2 : 1, prob, prob2 < 0 ? 15 : prob2);
        host_store_fence(); // ... before the command word, which leaves the write-combining buffer now
        pending_seq_ = seq;
        pending_mode_ = with_hessian ? 2 : 1;
        return 0;
    }
    int Engine::persistent_post_resum(int slot, int prob)
    {
        if (!persistent_resum_ok(slot, prob)) return MBAVO_E_ARG;
        volatile PersistCmd *cmd = (volatile PersistCmd *)((char *)d_push_ + (size_t)slot * push_stride_);
        const unsigned long long seq = ++flag_seq_;
        host_store_fence(); // (the flags and the scale are out before the command word, as in persistent_post)
        cmd->word = persist_word(seq, persist_gen_of_[slot], 3, prob);
        host_store_fence();
        pending_seq_ = seq;
        pending_mode_ = 3;
        return 0;
    }
    int Engine::persistent_wait()
    {
        if (!persist_mask_) return MBAVO_E_ARG;
        const unsigned long long seq = pending_seq_;
        const auto t0 = std::chrono::steady_clock::now();
        volatile unsigned long long *f = (volatile unsigned long long *)h_flag_;
        for (long spins = 1;; ++spins)
        {
            if (*f == seq)
            {
                __atomic_thread_fence(__ATOMIC_ACQUIRE);
#if defined(MBAVO_PERSIST_STAMPS)
                {
                    // (per command mode; the summing workgroup's own stamps; a re-summation has no prologue: its stamp 0 is its start)
                    static double sum[4] = {}, host[4] = {}, ph[4][3] = {}; static long cnt[4] = {};
                    const int m = pending_mode_ & 3;
                    sum[m] += (double)(f[1] - f[2]) * 0.01; host[m] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6;
                    ph[m][0] += (double)(f[3] - f[2]) * 0.01; ph[m][1] += (double)(f[4] - f[3]) * 0.01; ph[m][2] += (double)(f[5] - f[4]) * 0.01;
                    static double sub[4][4] = {};
                    if (m == 2) { sub[m][0] += (double)(long long)(f[9] - f[3]) * 0.01; sub[m][1] += (double)(long long)(f[10] - f[9]) * 0.01; sub[m][2] += (double)(long long)(f[11] - f[10]) * 0.01; sub[m][3] += (double)(long long)(f[12] - f[11]) * 0.01; }
                    static double arr[8] = {};
                    if (m == 2) for (int i = 0; i < 8; ++i) arr[i] += (double)(long long)(f[16 + i] - f[3]) * 0.01;
                    if (m == 2 && (cnt[m] + 1) % 50 == 0)
                        fprintf(stderr, "persist: waves at the last barrier, us after the prologue: %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f\n", arr[0] / (cnt[m] + 1), arr[1] / (cnt[m] + 1),
                                arr[2] / (cnt[m] + 1), arr[3] / (cnt[m] + 1), arr[4] / (cnt[m] + 1), arr[5] / (cnt[m] + 1), arr[6] / (cnt[m] + 1), arr[7] / (cnt[m] + 1));
                    if (m == 2 && (cnt[m] + 1) % 50 == 0)
                        fprintf(stderr, "persist: tile of mode 2: sample done %.2f | intensity sum + weight %.2f | rows parked, summed, outer product %.2f | accumulators parked + barrier %.2f us\n",
                                sub[m][0] / (cnt[m] + 1), sub[m][1] / (cnt[m] + 1), sub[m][2] / (cnt[m] + 1), sub[m][3] / (cnt[m] + 1));
                    if (++cnt[m] % 50 == 0)
                        fprintf(stderr, "persist: mode %d kernel-side %.2f us (pose prologue %.2f, tile %.2f, partial + ticket wait %.2f, final %.2f), host round trip %.2f us (mean of %ld)\n",
                                m, sum[m] / cnt[m], ph[m][0] / cnt[m], ph[m][1] / cnt[m], ph[m][2] / cnt[m], (sum[m] - ph[m][0] - ph[m][1] - ph[m][2]) / cnt[m], host[m] / cnt[m], cnt[m]);
                }
#endif
                return 0;
            }
            host_spin_pause();
            if ((spins & 0xffff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(3)) break;
        }
        fprintf(stderr, "mbavo: persistent evaluation timed out\n");
        persist_mask_ = 0; // the workgroups give up by themselves (~2 s each kernel)
        (void)hipStreamSynchronize(stream_);
        (void)hipMemsetAsync(d_tickets_, 0, cap_tickets_, stream_);
        return (int)hipErrorLaunchTimeOut;
    }

    // the SECOND problem of the command posted with sequence number `seq`: has it completed (no waiting)? / wait for it
    bool Engine::persistent_second_done(unsigned long long seq) const
    {
        return h_flag_ && ((volatile unsigned long long *)h_flag_)[8] == seq;
    }
    int Engine::persistent_wait_second(unsigned long long seq)
    {
        if (!persist_mask_ || !h_flag_) return MBAVO_E_ARG;
        const auto t0 = std::chrono::steady_clock::now();
        volatile unsigned long long *f = (volatile unsigned long long *)h_flag_ + 8;
        for (long spins = 1;; ++spins)
        {
            if (*f == seq) { __atomic_thread_fence(__ATOMIC_ACQUIRE); return 0; }
            host_spin_pause();
            if ((spins & 0xffff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(3)) break;
        }
        fprintf(stderr, "mbavo: persistent evaluation (second problem) timed out\n");
        persist_mask_ = 0;
        (void)hipStreamSynchronize(stream_);
        (void)hipMemsetAsync(d_tickets_, 0, cap_tickets_, stream_);
        return (int)hipErrorLaunchTimeOut;
    }

    int Engine::persistent_end(int slot)
    {
        if (!persistent_active(slot)) return 0;
        volatile PersistCmd *cmd = (volatile PersistCmd *)((char *)d_push_ + (size_t)slot * push_stride_);
        host_store_fence();
        cmd->word = persist_word(++flag_seq_, persist_gen_of_[slot], 0);
        host_store_fence();
        persist_mask_ &= ~(1u << slot);
        return 0;
    }
    int Engine::persistent_end_all()
    {
        for (int sl = 0; sl < kPushSlots; ++sl) (void)persistent_end(sl);
        return 0;
    }
    int Engine::wait_evaluation()
    {
        if (flag_pending_ && h_flag_)
        {
            volatile unsigned long long *f = (volatile unsigned long long *)h_flag_;
            for (long spins = 0; spins < 20000000L; ++spins)
            {
                if (*f == flag_seq_) { flag_pending_ = false; return 0; }
                host_spin_pause();
            }
        }
        flag_pending_ = false;
        return (int)hipStreamSynchronize(stream_);
    }


    // ------------------------------------------------------------------ resident LM loop: host side
    // is `p` a problem the single-launch sample-parallel kernel takes (rebuild_layout's rule for a list of one)?
    const char *Engine::last_kernel()
    {
        const int *k = last_kernel_id_;
        if (k[0] == 0) last_kernel_[0] = 0;
        else if (k[3] > 0)
            snprintf(last_kernel_, sizeof(last_kernel_), "k_fused_sp<%d,%s,false,%d,%s>", k[0], k[1] ? "true" : "false", k[3], k[4] ? "true" : "false");
        else
            snprintf(last_kernel_, sizeof(last_kernel_), "k_fused<%d,%s,%s,%s>", k[0], k[1] ? "true" : "false", k[2] == 2 && k[1] ? "packed" : k[2] == 1 ? "true" : "false", k[5] ? "true" : "false");
        return last_kernel_;
    }

    void Engine::profile_enable(int every)
    {
        prof_every_ = every > 0 ? every : 0;
        prof_used_ = 0;
        prof_seen_ = 0;
    }

    hipError_t Engine::ensure_lds(const void *kernel, size_t bytes)
    {
        size_t &have = lds_attr_[kernel];
        if (bytes <= have) return hipSuccess;
        const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e == hipSuccess) have = bytes;
        return e;
    }

    // Event pair for the fused kernel of every prof_every_-th launch (see MBAVO_LAUNCH_TIMED); false = not this one.
    bool Engine::prof_events(hipEvent_t *e0, hipEvent_t *e1)
    {
        if (prof_every_ == 0) return false;
        if ((prof_seen_++ % prof_every_) != 0) return false;
        if (prof_used_ * 2 + 1 >= (int)prof_ev_.size())
        {
            if (prof_ev_.size() >= 2 * 8192) return false; // cap; later launches are not timed
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return false;
            prof_ev_.push_back(a);
            prof_ev_.push_back(b);
        }
        *e0 = prof_ev_[prof_used_ * 2];
        *e1 = prof_ev_[prof_used_ * 2 + 1];
        ++prof_used_;
        return true;
    }

    int Engine::profile_read(double *ms_sum, int *launches)
    {
        double sum = 0.0;
        int n = 0;
        const int pairs = prof_used_ < (int)prof_ev_.size() / 2 ? prof_used_ : (int)prof_ev_.size() / 2;
        for (int i = 0; i < pairs; ++i)
        {
            float ms = 0.f;
            hipError_t e = hipEventSynchronize(prof_ev_[2 * i + 1]);
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, prof_ev_[2 * i], prof_ev_[2 * i + 1]);
            if (e != hipSuccess) return (int)e;
            sum += ms;
            ++n;
        }
        if (ms_sum) *ms_sum = sum;
        if (launches) *launches = n;
        return 0;
    }

    int Engine::fetch_status_enqueue(int *h_pinned)
    {
        *h_pinned = status_seen_;
        if (!d_status_) return 0;
        return (int)hipMemcpyAsync(h_pinned, d_status_, sizeof(int), hipMemcpyDeviceToHost, stream_);
    }

    int Engine::fetch_status_take(const int *h_pinned)
    {
        const int delta = *h_pinned - status_seen_;
        status_seen_ = *h_pinned;
        return delta;
    }

    int Engine::fetch_status()
    {
        int s = 0;
        if (d_status_ && hipMemcpy(&s, d_status_, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
        const int delta = s - status_seen_;
        status_seen_ = s;
        return delta;
    }
} // namespace mbavo

#if defined(MBAVO_FUSED_STAMPS) // experiment builds only (tools/fused_stamps.py)
extern "C" int mbavo_debug_fused_stamps(unsigned long long *out, int n)
{
    if (n > 2048 * 8) n = 2048 * 8;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mbavo::g_fused_stamps), sizeof(unsigned long long) * n);
}
extern "C" int mbavo_debug_wave_stamps(unsigned long long *out, int n)
{
    if (n > 1024 * 16 * 4) n = 1024 * 16 * 4;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mbavo::g_wave_stamps), sizeof(unsigned long long) * n);
}
#endif
