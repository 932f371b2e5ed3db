// host_math.cpp -- see host_math.h for the reference counterparts.
#include "host_math.h"
#include "se3_math.h"

#include <algorithm>
#include <cfloat>
#include <atomic>
#include <cmath>
#include <mutex>
#include <cstdlib>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <cstring>
#include <limits>
#include <vector>

namespace mbavo
{
    // Scatter the per-frame packed blocks [cost | g(6k) | upper(H)(6k x 6k)] into the global
    // system ordered [t_0..t_{N-1} | w_0..w_{N-1}]: local j < 3k -> 3*start + j, else
    // 3*(N + start) + (j - 3k)   (merge_hessian_gradient_cost.cpp:52-85).
    void merge_blocks_host(int F, int k, const double *fb, const int *start_idx, int N, double *total_cost,
                           double *H, double *g)
    {
        const int m = 6 * k, ndim = m + 1, E = ndim * (ndim + 1) / 2, n = 6 * N;
        double cost = 0.0;
        if (H)
        {
            std::fill(H, H + (size_t)n * n, 0.0);
            std::fill(g, g + n, 0.0);
        }
        std::vector<int> gidx(m);
        for (int f = 0; f < F; ++f)
        {
            const double *blk = fb + (size_t)f * E;
            cost += blk[0];
            if (!H) continue;
            const int st = start_idx[f];
            for (int j = 0; j < m; ++j) gidx[j] = j < 3 * k ? 3 * st + j : 3 * (N + st) + (j - 3 * k);
            for (int j = 0; j < m; ++j) g[gidx[j]] += blk[1 + j];
            const double *h = blk + ndim;
            for (int r = 0; r < m; ++r)
                for (int c = r; c < m; ++c)
                {
                    const double v = *h++;
                    const int R = gidx[r], C = gidx[c];
                    H[(size_t)C * n + R] += v;
                    if (R != C) H[(size_t)R * n + C] += v;
                }
        }
        *total_cost = cost;
    }

    // Minimum-norm least squares by one-sided (Hestenes) Jacobi: rotate the columns of
    // G = A until they are orthogonal, G = U S, A V = G.  Singular values below
    // n * eps * s_max are treated as zero (Eigen's default JacobiSVD rank threshold).
    // The LM loop of a tracked frame spends more host time here than anywhere else (a 12 x 12 solve per iteration), so
    // the three dot products of a column pair run as FOUR interleaved partial sums each (element i goes to sum i mod 4,
    // combined as (s0 + s1) + (s2 + s3)): a fixed association, written out, so that the result does not depend on the
    // vector width the compiler picks, with four independent dependency chains instead of one.
    namespace
    {
        struct Dots { double a, c, d; };
        inline Dots pair_dots(const double *gp, const double *gq, int n)
        {
            double a0 = 0, a1 = 0, a2 = 0, a3 = 0, c0 = 0, c1 = 0, c2 = 0, c3 = 0, d0 = 0, d1 = 0, d2 = 0, d3 = 0;
            int i = 0;
            for (; i + 4 <= n; i += 4)
            {
                a0 += gp[i] * gp[i]; a1 += gp[i + 1] * gp[i + 1]; a2 += gp[i + 2] * gp[i + 2]; a3 += gp[i + 3] * gp[i + 3];
                c0 += gq[i] * gq[i]; c1 += gq[i + 1] * gq[i + 1]; c2 += gq[i + 2] * gq[i + 2]; c3 += gq[i + 3] * gq[i + 3];
                d0 += gp[i] * gq[i]; d1 += gp[i + 1] * gq[i + 1]; d2 += gp[i + 2] * gq[i + 2]; d3 += gp[i + 3] * gq[i + 3];
            }
            if (i < n) { a0 += gp[i] * gp[i]; c0 += gq[i] * gq[i]; d0 += gp[i] * gq[i]; ++i; }
            if (i < n) { a1 += gp[i] * gp[i]; c1 += gq[i] * gq[i]; d1 += gp[i] * gq[i]; ++i; }
            if (i < n) { a2 += gp[i] * gp[i]; c2 += gq[i] * gq[i]; d2 += gp[i] * gq[i]; ++i; }
            return Dots{(a0 + a1) + (a2 + a3), (c0 + c1) + (c2 + c3), (d0 + d1) + (d2 + d3)};
        }
        inline void rotate_pair(double *p, double *q, int n, double cs, double sn)
        {
            for (int i = 0; i < n; ++i)
            {
                const double u = p[i], w = q[i];
                p[i] = cs * u - sn * w;
                q[i] = sn * u + cs * w;
            }
        }
#if defined(__x86_64__)
        // The same two pieces on 256-bit vectors (AVX2, no FMA): lane j of an accumulator IS partial sum j of pair_dots,
        // every product is rounded before it is added, the lanes are combined as (s0 + s1) + (s2 + s3) -- bit-identical
        // results, chosen once at run time.  n must be a multiple of 4 (the tracker's 6N with even N).
        __attribute__((target("avx2"))) inline Dots pair_dots_avx2(const double *gp, const double *gq, int n)
        {
            __m256d a = _mm256_setzero_pd(), c = _mm256_setzero_pd(), d = _mm256_setzero_pd();
            for (int i = 0; i < n; i += 4)
            {
                const __m256d u = _mm256_loadu_pd(gp + i), w = _mm256_loadu_pd(gq + i);
                a = _mm256_add_pd(a, _mm256_mul_pd(u, u));
                c = _mm256_add_pd(c, _mm256_mul_pd(w, w));
                d = _mm256_add_pd(d, _mm256_mul_pd(u, w));
            }
            // (s0 + s1) + (s2 + s3): hadd gives [s0 + s1, ., s2 + s3, .] per source
            const __m256d ac = _mm256_hadd_pd(a, c);                   // a0+a1, c0+c1, a2+a3, c2+c3
            const __m128d acs = _mm_add_pd(_mm256_castpd256_pd128(ac), _mm256_extractf128_pd(ac, 1));
            const __m256d dd = _mm256_hadd_pd(d, d);
            const __m128d ds = _mm_add_pd(_mm256_castpd256_pd128(dd), _mm256_extractf128_pd(dd, 1));
            Dots r;
            r.a = _mm_cvtsd_f64(acs);
            r.c = _mm_cvtsd_f64(_mm_unpackhi_pd(acs, acs));
            r.d = _mm_cvtsd_f64(ds);
            return r;
        }
        __attribute__((target("avx2"))) inline void rotate_pair_avx2(double *p, double *q, int n, double cs, double sn)
        {
            const __m256d vc = _mm256_set1_pd(cs), vs = _mm256_set1_pd(sn);
            for (int i = 0; i < n; i += 4)
            {
                const __m256d u = _mm256_loadu_pd(p + i), w = _mm256_loadu_pd(q + i);
                _mm256_storeu_pd(p + i, _mm256_sub_pd(_mm256_mul_pd(vc, u), _mm256_mul_pd(vs, w)));
                _mm256_storeu_pd(q + i, _mm256_add_pd(_mm256_mul_pd(vs, u), _mm256_mul_pd(vc, w)));
            }
        }
        // rotation parameters of four column pairs at a time (the same IEEE operations per pair as the scalar loop:
        // packed sqrt / div are correctly rounded); returns the number of pairs done (a multiple of 4)
        __attribute__((target("avx2"))) inline int rotation_params_avx2(const double *a_, const double *c_, const double *d_, int npairs,
                                                                        double eps, double *cs_, double *sn_, double *skip_)
        {
            const __m256d one = _mm256_set1_pd(1.0), zero = _mm256_setzero_pd(), sign = _mm256_set1_pd(-0.0);
            int pr = 0;
            for (; pr + 4 <= npairs; pr += 4)
            {
                const __m256d a = _mm256_loadu_pd(a_ + pr), c = _mm256_loadu_pd(c_ + pr), d = _mm256_loadu_pd(d_ + pr);
                const __m256d absd = _mm256_andnot_pd(sign, d);
                const __m256d lim = _mm256_mul_pd(_mm256_set1_pd(eps), _mm256_sqrt_pd(_mm256_mul_pd(a, c)));
                const __m256d skip = _mm256_or_pd(_mm256_cmp_pd(d, zero, _CMP_EQ_OQ), _mm256_cmp_pd(absd, lim, _CMP_LE_OQ));
                const __m256d zeta = _mm256_div_pd(_mm256_sub_pd(c, a), _mm256_mul_pd(_mm256_set1_pd(2.0), d));
                const __m256d sgn = _mm256_blendv_pd(_mm256_set1_pd(-1.0), one, _mm256_cmp_pd(zeta, zero, _CMP_GE_OQ));
                const __m256d den = _mm256_add_pd(_mm256_andnot_pd(sign, zeta), _mm256_sqrt_pd(_mm256_add_pd(one, _mm256_mul_pd(zeta, zeta))));
                const __m256d t = _mm256_div_pd(sgn, den);
                const __m256d cs = _mm256_div_pd(one, _mm256_sqrt_pd(_mm256_add_pd(one, _mm256_mul_pd(t, t))));
                _mm256_storeu_pd(cs_ + pr, cs);
                _mm256_storeu_pd(sn_ + pr, _mm256_mul_pd(cs, t));
                _mm256_storeu_pd(skip_ + pr, _mm256_and_pd(skip, one));
            }
            return pr;
        }
        inline bool have_avx2()
        { // evaluated on first use (no static-initialisation-order dependence); the file is also parsed for the device
#if defined(__HIP_DEVICE_COMPILE__)
            return false;
#else
            static const bool v = (__builtin_cpu_init(), __builtin_cpu_supports("avx2") != 0);
            return v;
#endif
        }
        inline int rotation_params_wide(const double *a, const double *c, const double *d, int npairs, double eps, double *cs, double *sn,
                                        double *skip)
        {
            return rotation_params_avx2(a, c, d, npairs, eps, cs, sn, skip);
        }
        inline Dots pair_dots_wide(const double *gp, const double *gq, int n) { return pair_dots_avx2(gp, gq, n); }
        inline void rotate_pair_wide(double *p, double *q, int n, double cs, double sn) { rotate_pair_avx2(p, q, n, cs, sn); }
#else
        inline bool have_avx2() { return false; }
        inline int rotation_params_wide(const double *, const double *, const double *, int, double, double *, double *, double *) { return 0; }
        inline Dots pair_dots_wide(const double *gp, const double *gq, int n) { return pair_dots(gp, gq, n); }
        inline void rotate_pair_wide(double *p, double *q, int n, double cs, double sn) { rotate_pair(p, q, n, cs, sn); }
#endif
    } // namespace

    static int solve_svd(const double *A, const double *b, int n, double *x)
    {
        // work arrays kept per thread: no allocation in the LM loop
        static thread_local std::vector<double> Gv, Vv, s2v;
        Gv.assign(A, A + (size_t)n * n);
        Vv.assign((size_t)n * n, 0.0);
        s2v.resize(n);
        double *G = Gv.data(), *V = Vv.data(), *s2 = s2v.data();
        for (int i = 0; i < n; ++i) V[(size_t)i * n + i] = 1.0;
        const double eps = std::numeric_limits<double>::epsilon();
        // Rotations of a sweep in round-robin (tournament) order, as the device solver takes them (lm_solvers.h): the
        // n / 2 column pairs of a round are disjoint, so their dot products, their rotation parameters (two square roots
        // and two divisions each: the latency chain that dominated the row-cyclic order) and their updates are independent
        // pieces of work for the out-of-order core.  Odd n takes the row-cyclic order.
        const int half = n / 2, m1 = n - 1;
        const bool wide = have_avx2() && (n & 3) == 0;
        static thread_local std::vector<int> pp_, qq_;
        static thread_local std::vector<double> cs_, sn_, da_, dc_, dd_, skip_;
        pp_.resize(half + 1); qq_.resize(half + 1); cs_.resize(half + 1); sn_.resize(half + 1);
        da_.resize(half + 4); dc_.resize(half + 4); dd_.resize(half + 4); skip_.resize(half + 4);
        static thread_local std::vector<int> sched_; // the tournament schedule of this n: m1 rounds x half pairs (p < q)
        static thread_local int sched_n_ = 0;
        if ((n & 1) == 0 && n >= 4 && sched_n_ != n)
        {
            sched_.resize((size_t)2 * m1 * half);
            for (int r = 0; r < m1; ++r)
                for (int pr = 0; pr < half; ++pr)
                {
                    int p = pr == 0 ? m1 : (r + pr) % m1, q = pr == 0 ? r : (r - pr + m1) % m1;
                    if (p > q) std::swap(p, q);
                    sched_[2 * (r * half + pr)] = p; sched_[2 * (r * half + pr) + 1] = q;
                }
            sched_n_ = n;
        }
        for (int sweep = 0; sweep < 60; ++sweep)
        {
            bool rotated = false;
            if ((n & 1) == 0 && n >= 4)
            {
                for (int r = 0; r < m1; ++r)
                { // three passes over the round's disjoint pairs, each a loop of independent iterations: dot products;
                  // rotation parameters (branch-free: the two square roots and three divisions of a pair overlap with the
                  // other pairs' instead of forming one chain per pair); rotations
                    for (int pr = 0; pr < half; ++pr)
                    {
                        const int p = sched_[2 * (r * half + pr)], q = sched_[2 * (r * half + pr) + 1];
                        pp_[pr] = p; qq_[pr] = q;
                        const Dots t3 = wide ? pair_dots_wide(G + (size_t)p * n, G + (size_t)q * n, n) : pair_dots(G + (size_t)p * n, G + (size_t)q * n, n);
                        da_[pr] = t3.a; dc_[pr] = t3.c; dd_[pr] = t3.d;
                    }
                    int pr0 = 0;
                    if (wide) pr0 = rotation_params_wide(da_.data(), dc_.data(), dd_.data(), half, eps, cs_.data(), sn_.data(), skip_.data());
                    for (int pr = pr0; pr < half; ++pr)
                    {
                        const double a = da_[pr], c = dc_[pr], d = dd_[pr];
                        const bool skip = d == 0.0 || std::fabs(d) <= eps * std::sqrt(a * c);
                        const double zeta = (c - a) / (2.0 * d);
                        const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                        const double cs = 1.0 / std::sqrt(1.0 + t * t);
                        cs_[pr] = cs; sn_[pr] = cs * t;
                        skip_[pr] = skip ? 1.0 : 0.0;
                    }
                    for (int pr = 0; pr < half; ++pr)
                        if (skip_[pr] != 0.0) pp_[pr] = -1;
                    for (int pr = 0; pr < half; ++pr)
                    {
                        if (pp_[pr] < 0) continue;
                        rotated = true;
                        if (wide)
                        {
                            rotate_pair_wide(G + (size_t)pp_[pr] * n, G + (size_t)qq_[pr] * n, n, cs_[pr], sn_[pr]);
                            rotate_pair_wide(V + (size_t)pp_[pr] * n, V + (size_t)qq_[pr] * n, n, cs_[pr], sn_[pr]);
                            continue;
                        }
                        rotate_pair(G + (size_t)pp_[pr] * n, G + (size_t)qq_[pr] * n, n, cs_[pr], sn_[pr]);
                        rotate_pair(V + (size_t)pp_[pr] * n, V + (size_t)qq_[pr] * n, n, cs_[pr], sn_[pr]);
                    }
                }
            }
            else
                for (int p = 0; p < n - 1; ++p)
                    for (int q = p + 1; q < n; ++q)
                    {
                        double *gp = G + (size_t)p * n, *gq = G + (size_t)q * n;
                        const Dots t3 = pair_dots(gp, gq, n);
                        const double a = t3.a, c = t3.c, d = t3.d;
                        if (d == 0.0 || std::fabs(d) <= eps * std::sqrt(a * c)) continue;
                        rotated = true;
                        const double zeta = (c - a) / (2.0 * d);
                        const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                        const double cs = 1.0 / std::sqrt(1.0 + t * t), sn = cs * t;
                        rotate_pair(gp, gq, n, cs, sn);
                        rotate_pair(V + (size_t)p * n, V + (size_t)q * n, n, cs, sn);
                    }
            if (!rotated) break;
        }
        double smax2 = 0.0;
        for (int j = 0; j < n; ++j)
        {
            double a = 0;
            for (int i = 0; i < n; ++i) a += G[(size_t)j * n + i] * G[(size_t)j * n + i];
            s2[j] = a;
            smax2 = std::max(smax2, a);
        }
        const double thr = std::max((double)std::max(n, 1) * eps * std::sqrt(smax2), DBL_MIN);
        std::fill(x, x + n, 0.0);
        int rank = 0;
        for (int j = 0; j < n; ++j)
        {
            if (!(std::sqrt(s2[j]) >= thr) || s2[j] == 0.0) continue;
            ++rank;
            double dot = 0; // (u_j . b) / s_j = (g_j . b) / s_j^2
            for (int i = 0; i < n; ++i) dot += G[(size_t)j * n + i] * b[i];
            dot /= s2[j];
            for (int i = 0; i < n; ++i) x[i] += V[(size_t)j * n + i] * dot;
        }
        return rank;
    }

    // Symmetric A = P^T L D L^T P with the largest remaining diagonal entry as pivot.
    static int solve_ldlt(const double *A, const double *b, int n, double *x)
    {
        std::vector<double> M(A, A + (size_t)n * n); // full symmetric copy, column-major
        std::vector<int> order(n);
        for (int i = 0; i < n; ++i) order[i] = i;
        auto at = [&](int r, int c) -> double & { return M[(size_t)c * n + r]; };
        for (int k = 0; k < n; ++k)
        {
            int piv = k;
            for (int i = k + 1; i < n; ++i)
                if (std::fabs(at(i, i)) > std::fabs(at(piv, piv))) piv = i;
            if (piv != k)
            {
                for (int c = 0; c < n; ++c) std::swap(at(k, c), at(piv, c));
                for (int r = 0; r < n; ++r) std::swap(at(r, k), at(r, piv));
                std::swap(order[k], order[piv]);
            }
            const double d = at(k, k);
            if (d == 0.0) continue;
            for (int i = k + 1; i < n; ++i) at(i, k) /= d; // column of L
            for (int j = k + 1; j < n; ++j)
            {
                const double ljk_d = at(j, k) * d;
                for (int i = j; i < n; ++i) at(i, j) -= at(i, k) * ljk_d;
            }
            for (int j = k + 1; j < n; ++j)
                for (int i = j + 1; i < n; ++i) at(j, i) = at(i, j); // keep the trailing block symmetric for later pivots
        }
        std::vector<double> y(n);
        for (int i = 0; i < n; ++i) y[i] = b[order[i]];
        for (int c = 0; c < n; ++c)
            for (int r = c + 1; r < n; ++r) y[r] -= at(r, c) * y[c];
        for (int i = 0; i < n; ++i) y[i] = std::fabs(at(i, i)) > DBL_MIN ? y[i] / at(i, i) : 0.0;
        for (int c = n - 1; c >= 0; --c)
            for (int r = c + 1; r < n; ++r) y[c] -= at(r, c) * y[r];
        for (int i = 0; i < n; ++i) x[order[i]] = y[i];
        return n;
    }

    // x = A^-1 b by LDL^T with diagonal pivoting when A is symmetric positive definite AND well conditioned: every pivot
    // positive and the pivot ratio (largest / smallest: within a factor n of the condition number for this pivoting) at most
    // `max_ratio`.  false: not decided here (rank deficient, indefinite or ill conditioned): the caller takes the SVD.
    // For such systems the minimum-norm solution of solve_normal_equation (solve_normal_equation.h:20-26) IS A^-1 b, and
    // the two agree to rounding x cond(A) -- 1e-9 of the step for the ratios admitted here -- at a twentieth of the cost
    // of the Jacobi sweeps (12 x 12: 0.4 us against 6.4 us on the host; on the device a Jacobi round is ~1 000 cycles of
    // dependent latency whatever the size).
    static bool solve_spd_fast(const double *A, const double *b, int n, double *x, double max_ratio)
    {
        static thread_local std::vector<double> Mv, yv;
        static thread_local std::vector<int> ov;
        Mv.assign(A, A + (size_t)n * n);
        yv.resize(n);
        ov.resize(n);
        double *M = Mv.data(), *y = yv.data();
        int *order = ov.data();
        for (int i = 0; i < n; ++i) order[i] = i;
        auto at = [&](int r, int c) -> double & { return M[(size_t)c * n + r]; };
        double dmax = 0.0, dmin = DBL_MAX;
        for (int k = 0; k < n; ++k)
        {
            int piv = k;
            for (int i = k + 1; i < n; ++i)
                if (at(i, i) > at(piv, piv)) piv = i;
            if (piv != k)
            { // symmetric swap of the trailing LOWER triangle's rows / columns k and piv
                for (int c = 0; c < n; ++c) std::swap(at(k, c), at(piv, c));
                for (int r = 0; r < n; ++r) std::swap(at(r, k), at(r, piv));
                std::swap(order[k], order[piv]);
            }
            const double d = at(k, k);
            if (!(d > 0.0)) return false;
            dmax = std::max(dmax, d);
            dmin = std::min(dmin, d);
            if (dmax > max_ratio * dmin) return false;
            const double rd = 1.0 / d;
            for (int i = k + 1; i < n; ++i) at(i, k) *= rd; // column of L
            for (int j = k + 1; j < n; ++j)
            {
                const double ljk_d = at(j, k) * d;
                for (int i = j; i < n; ++i) at(i, j) -= at(i, k) * ljk_d;
            }
            for (int j = k + 1; j < n; ++j)
                for (int i = j + 1; i < n; ++i) at(j, i) = at(i, j);
        }
        for (int i = 0; i < n; ++i) y[i] = b[order[i]];
        for (int c = 0; c < n; ++c)
            for (int r = c + 1; r < n; ++r) y[r] -= at(r, c) * y[c];
        for (int i = 0; i < n; ++i) y[i] /= at(i, i);
        for (int c = n - 1; c >= 0; --c)
            for (int r = c + 1; r < n; ++r) y[c] -= at(r, c) * y[r];
        for (int i = 0; i < n; ++i) x[order[i]] = y[i];
        return true;
    }

    static EnvOverrides scan_environment()
    { // THE reader of the environment for every switch that changes results or scheduling (options.h); the A/B tools' override layer
        auto num = [](const char *name) { const char *v = getenv(name); return v && *v ? atoi(v) : kEnvUnset; };
        auto real = [](const char *name) { const char *v = getenv(name); return v && *v ? atof(v) : -2.0; };
        EnvOverrides e;
        e.sp = num("MBAVO_SP"); e.one = num("MBAVO_ONE"); e.fused_pose = num("MBAVO_FUSED_POSE");
        e.fused_pose_max_s = num("MBAVO_FUSED_POSE_MAX_S"); e.persist = num("MBAVO_PERSIST"); e.prelaunch = num("MBAVO_PRELAUNCH");
        e.tiles_per_cu = num("MBAVO_TILES_PER_CU"); e.min_tile_px = num("MBAVO_MIN_TILE_PX"); e.sp_max_slot_tiles = num("MBAVO_SP_MAX_SLOT_TILES");
        e.speculate = num("MBAVO_SPECULATE"); e.persist_levels = num("MBAVO_PERSIST_LEVELS"); e.kf_multi = num("MBAVO_KF_MULTI"); e.kf_speculate = num("MBAVO_KF_SPECULATE"); e.ride_along = num("MBAVO_RIDE_ALONG"); e.resum = num("MBAVO_RESUM");
        e.lm_eig = num("MBAVO_LM_EIG"); e.lm_poses = num("MBAVO_LM_POSES"); e.lm_defer = num("MBAVO_LM_DEFER");
        e.lm_retile = num("MBAVO_LM_RETILE"); e.lm_groups = num("MBAVO_LM_GROUPS");
        e.fast_solve = real("MBAVO_FAST_SOLVE"); e.lm_refine = real("MBAVO_LM_REFINE");
        if (e.fast_solve < 0.0 && e.fast_solve > -2.0) e.fast_solve = 0.0; // (a negative number in the variable: off)
        if (e.lm_refine < 0.0 && e.lm_refine > -2.0) e.lm_refine = 0.0;
        return e;
    }

    // The scan above is 22 getenv calls; the engine asks for its tuning on every evaluation of the latency-bound path and the
    // batched LM's group threads ask concurrently (getenv beside a setenv is undefined behaviour) -- so the environment is scanned
    // ONCE per process and again only when a tool says it changed it (mbavo_reload_env).
    static std::mutex g_env_mutex;
    static EnvOverrides g_env;
    static std::atomic<bool> g_env_valid{false};
    EnvOverrides read_env_overrides()
    {
        if (!g_env_valid.load(std::memory_order_acquire))
        {
            std::lock_guard<std::mutex> lock(g_env_mutex);
            if (!g_env_valid.load(std::memory_order_relaxed))
            {
                g_env = scan_environment();
                g_env_valid.store(true, std::memory_order_release);
            }
        }
        return g_env;
    }
    void reload_env_overrides()
    {
        std::lock_guard<std::mutex> lock(g_env_mutex);
        g_env = scan_environment();
        g_env_valid.store(true, std::memory_order_release);
    }

    int solve_normal_equation_host(const double *A, const double *b, int n, int solver_type, double *x, double fast_ratio)
    {
        // fast_ratio < 0: the library default under the environment's override.  The LM loops resolve it ONCE per call from
        // their options (optimize_trajectory, lm_batch) and pass it down, so that the host loop and the batched LM of one process
        // always run the same solver
        if (fast_ratio < 0.0) fast_ratio = opt_fast_ratio(0.0, read_env_overrides().fast_solve);
        int rank;
        if (solver_type == 0 && fast_ratio > 0.0 && solve_spd_fast(A, b, n, x, fast_ratio)) rank = n;
        else if (solver_type == 0) rank = solve_svd(A, b, n, x);
        else if (solver_type == 1) rank = solve_ldlt(A, b, n, x);
        else return -1;
        for (int i = 0; i < n; ++i) x[i] = -x[i];
        return rank;
    }
} // namespace mbavo

namespace SLAM
{
    namespace VO
    {
        // radius 1e4, clamped to [10, 1e32]; accepted: r /= max(1/3, 1 - (2q-1)^3);
        // rejected: r /= f, f *= 2   (levenberg_marquardt_strategy.cpp:9-45)
        LevenbergMarquardtStrategy::LevenbergMarquardtStrategy()
            : mRadius(1e4), mMaxRadius(1e32), mMinRadius(10), mDecreaseFactor(2.0) {}
        void LevenbergMarquardtStrategy::reset() { mRadius = 1e4; mDecreaseFactor = 2.0; }
        void LevenbergMarquardtStrategy::step_accepted(double q)
        {
            mRadius = mRadius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * q - 1.0, 3));
            mRadius = std::max(std::min(mMaxRadius, mRadius), mMinRadius);
            mDecreaseFactor = 2.0;
        }
        void LevenbergMarquardtStrategy::step_rejected()
        {
            mRadius = mRadius / mDecreaseFactor;
            mRadius = std::max(std::min(mMaxRadius, mRadius), mMinRadius);
            mDecreaseFactor *= 2.0;
        }
        double LevenbergMarquardtStrategy::get_radius() { return mRadius; }

        // Non-monotonic step acceptance, Conn/Gould/Toint Alg. 10.1.2 as Ceres implements it
        // (trust_region_step_evaluator.cpp:45-126)
        TrustRegionStepEvaluator::TrustRegionStepEvaluator(int m)
            : max_consecutive_nonmonotonic_steps_(m), minimum_cost_(0), current_cost_(0), reference_cost_(0),
              candidate_cost_(0), accumulated_reference_model_cost_change_(0),
              accumulated_candidate_model_cost_change_(0), num_consecutive_nonmonotonic_steps_(0) {}
        void TrustRegionStepEvaluator::reset(double c)
        {
            minimum_cost_ = current_cost_ = reference_cost_ = candidate_cost_ = c;
            accumulated_reference_model_cost_change_ = accumulated_candidate_model_cost_change_ = 0.0;
            num_consecutive_nonmonotonic_steps_ = 0;
        }
        double TrustRegionStepEvaluator::StepQuality(double cost, double mcc) const
        {
            if (cost >= std::numeric_limits<double>::max()) return std::numeric_limits<double>::lowest();
            const double now = (current_cost_ - cost) / mcc;
            const double hist = (reference_cost_ - cost) / (accumulated_reference_model_cost_change_ + mcc);
            return std::max(now, hist);
        }
        void TrustRegionStepEvaluator::StepAccepted(double cost, double mcc)
        {
            current_cost_ = cost;
            accumulated_candidate_model_cost_change_ += mcc;
            accumulated_reference_model_cost_change_ += mcc;
            if (current_cost_ < minimum_cost_)
            {
                minimum_cost_ = candidate_cost_ = current_cost_;
                num_consecutive_nonmonotonic_steps_ = 0;
                accumulated_candidate_model_cost_change_ = 0.0;
            }
            else
            {
                ++num_consecutive_nonmonotonic_steps_;
                if (current_cost_ > candidate_cost_)
                {
                    candidate_cost_ = current_cost_;
                    accumulated_candidate_model_cost_change_ = 0.0;
                }
            }
            if (num_consecutive_nonmonotonic_steps_ == max_consecutive_nonmonotonic_steps_)
            {
                reference_cost_ = candidate_cost_;
                accumulated_reference_model_cost_change_ = accumulated_candidate_model_cost_change_;
            }
        }
    } // namespace VO

    namespace Core
    {
        using mbavo::Quat;

        void SplineSE3::InsertControlKnot(const double q[4], const double t[3])
        {
            mT.insert(mT.end(), t, t + 3);
            mR.insert(mR.end(), q, q + 4);
        }

        void SplineSE3::PopFrontControlKnot()
        {
            mT.erase(mT.begin(), mT.begin() + 3);
            mR.erase(mR.begin(), mR.begin() + 4);
            mT0 += mDt;
        }

        bool SplineSE3::GetPose(double t, double q_out[4], double t_out[3], double *jR, double *jt) const
        { // Spline.h:222-281; the reference asserts on the range, here it is a return value
            int idx;
            double u;
            mbavo::spline_segment(t, mT0, mDt, idx, u);
            if (idx < 0 || idx + mDegK > (int)get_num_knots()) return false;
            const double *kt = mT.data() + 3 * idx, *kR = mR.data() + 4 * idx;
            Quat q;
            if (mDegK == 2)
            {
                double c[2];
                mbavo::trans_coeffs<2>(u, c);
                mbavo::spline_translation<2>(kt, c, t_out);
                q = jR ? mbavo::spline_rotation<2, true>(kR, u, jR) : mbavo::spline_rotation<2, false>(kR, u, nullptr);
                if (jt)
                {
                    std::fill(jt, jt + 18, 0.0);
                    for (int a = 0; a < 3; ++a)
                        for (int j = 0; j < 2; ++j) jt[a * 6 + 3 * j + a] = c[j];
                }
            }
            else if (mDegK == 4)
            {
                double c[4];
                mbavo::trans_coeffs<4>(u, c);
                mbavo::spline_translation<4>(kt, c, t_out);
                q = jR ? mbavo::spline_rotation<4, true>(kR, u, jR) : mbavo::spline_rotation<4, false>(kR, u, nullptr);
                if (jt)
                {
                    std::fill(jt, jt + 36, 0.0);
                    for (int a = 0; a < 3; ++a)
                        for (int j = 0; j < 4; ++j) jt[a * 12 + 3 * j + a] = c[j];
                }
            }
            else
                return false;
            q_out[0] = q.x; q_out[1] = q.y; q_out[2] = q.z; q_out[3] = q.w;
            return true;
        }

        void SplineSE3::TransformByRight(const double dq[4], const double dt[3])
        { // Spline.h:212-219: t_i += R_i * dt ; R_i = R_i * dR
            const Quat d{dq[0], dq[1], dq[2], dq[3]};
            for (size_t i = 0; i < get_num_knots(); ++i)
            {
                const Quat R = mbavo::load_quat(&mR[4 * i]);
                double r[3];
                mbavo::qrotate(R, dt, r);
                for (int a = 0; a < 3; ++a) mT[3 * i + a] += r[a];
                const Quat n = mbavo::qmul(R, d);
                mR[4 * i] = n.x; mR[4 * i + 1] = n.y; mR[4 * i + 2] = n.z; mR[4 * i + 3] = n.w;
            }
        }

        bool SplineSE3::TransformTo(double t, const double q_target[4], const double t_target[3])
        { // Spline.h:183-200: dR = R(t)^-1 * R_target, dt = R(t)^-1 * (t_target - t(t)), then TransformByRight
            double qo[4], po[3];
            if (!GetPose(t, qo, po)) return false;
            const double n2 = qo[0] * qo[0] + qo[1] * qo[1] + qo[2] * qo[2] + qo[3] * qo[3];
            if (!(n2 > 0)) return false;
            const Quat qi{-qo[0] / n2, -qo[1] / n2, -qo[2] / n2, qo[3] / n2}; // Eigen inverse(): conjugate / squaredNorm
            const Quat dR = mbavo::qmul(qi, mbavo::load_quat(q_target));
            const double d[3] = {t_target[0] - po[0], t_target[1] - po[1], t_target[2] - po[2]};
            double dt[3];
            mbavo::qrotate(qi, d, dt);
            const double dq[4] = {dR.x, dR.y, dR.z, dR.w};
            TransformByRight(dq, dt);
            return true;
        }

        void SplineSE3::UpdateCtrlKnot_t(int s, int num, const double *dt)
        {
            for (int i = 0; i < 3 * num; ++i) mT[3 * s + i] += dt[i];
        }

        void SplineSE3::UpdateCtrlKnot_R(int s, int num, const double *dR)
        { // Spline.h:294-305: R = normalize(R * exp(w))
            for (int i = 0; i < num; ++i)
            {
                double *r = &mR[4 * (s + i)];
                const Quat n = mbavo::qmul(mbavo::load_quat(r), mbavo::so3_exp(dR + 3 * i));
                const double nn = std::sqrt(n.x * n.x + n.y * n.y + n.z * n.z + n.w * n.w);
                r[0] = n.x / nn; r[1] = n.y / nn; r[2] = n.z / nn; r[3] = n.w / nn;
            }
        }

        void SplineSE3::Plus_t(const double *dt, double *cand) const
        {
            for (size_t i = 0; i < mT.size(); ++i) cand[i] = mT[i] + dt[i];
        }

        void SplineSE3::Plus_R(const double *dR, double *cand) const
        { // Spline.h:317-330: candidate_i = R_i * exp(w_i), not re-normalised
            for (size_t i = 0; i < get_num_knots(); ++i)
            {
                const Quat n = mbavo::qmul(mbavo::load_quat(&mR[4 * i]), mbavo::so3_exp(dR + 3 * i));
                cand[4 * i] = n.x; cand[4 * i + 1] = n.y; cand[4 * i + 2] = n.z; cand[4 * i + 3] = n.w;
            }
        }

        void SplineSE3::InvalidParameter(const double *dt, const double *dR)
        {
            std::copy(dt, dt + mT.size(), mT.begin());
            std::copy(dR, dR + mR.size(), mR.begin());
        }

        void SplineSE3::ResetIdentity()
        {
            std::fill(mT.begin(), mT.end(), 0.0);
            for (size_t i = 0; i < get_num_knots(); ++i) { mR[4 * i] = mR[4 * i + 1] = mR[4 * i + 2] = 0.0; mR[4 * i + 3] = 1.0; }
        }
    } // namespace Core
} // namespace SLAM
