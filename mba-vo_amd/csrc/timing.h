// timing.h -- optional host-side phase timers of the tracking loop (MBAVO_TIMING=1 prints the totals to stderr at exit).
// Development aid for the latency work on the single-pair path; costs two clock reads per phase when enabled.
#ifndef MBAVO_TIMING_H
#define MBAVO_TIMING_H
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace mbavo
{
    struct PhaseTimers
    {
        enum { kUpload, kKeyframe, kEnqueue, kWait, kMerge, kSolve, kOutliers, kLevel, kTrack, kFlagsChanged, kFlagsSame, kCount };
        // (atomics: mbavo_lm_batch's groups time their evaluations from two host threads; nanoseconds so that the sum is an integer add)
        std::atomic<long long> nsec[kCount] = {};
        std::atomic<long> calls[kCount] = {};
        bool on;
        PhaseTimers() { const char *v = getenv("MBAVO_TIMING"); on = v && *v && *v != '0'; }
        ~PhaseTimers() { report(); }
        void report()
        {
            if (!on) return;
            static const char *names[kCount] = {"upload+pyramid", "keyframe processing", "evaluate enqueue", "evaluate wait", "host merge",
                                                "host solve+step", "outlier detection", "level setup", "trackFrame (all)", "accepted: flags changed", "accepted: flags same"};
            for (int i = 0; i < kCount; ++i)
                if (calls[i].load())
                    fprintf(stderr, "mbavo timing: %-20s %8ld calls  %9.3f ms total  %7.2f us each\n", names[i], calls[i].load(), nsec[i].load() * 1e-6,
                            nsec[i].load() * 1e-3 / calls[i].load());
            for (int i = 0; i < kCount; ++i) { nsec[i] = 0; calls[i] = 0; }
        }
        static PhaseTimers &get() { static PhaseTimers t; return t; }
    };
    struct PhaseScope
    {
        int id;
        std::chrono::steady_clock::time_point t0;
        bool on;
        explicit PhaseScope(int i) : id(i), on(PhaseTimers::get().on) { if (on) t0 = std::chrono::steady_clock::now(); }
        ~PhaseScope() { stop(); }
        void stop() // ends the phase early; the destructor then adds nothing
        {
            if (!on) return;
            on = false;
            PhaseTimers &t = PhaseTimers::get();
            t.nsec[id].fetch_add(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(), std::memory_order_relaxed);
            t.calls[id].fetch_add(1, std::memory_order_relaxed);
        }
    };
    // the joint persistent kernel's ride-along evaluations, process-wide (mbavo_ride_along_stats): commands that carried one, level
    // starts that found theirs (one dependent evaluation saved), level starts that had to wait a wasted one out
    struct RideAlongStats
    {
        std::atomic<long long> posts{0}, hits{0}, waits{0};
        static RideAlongStats &get() { static RideAlongStats s; return s; }
    };
} // namespace mbavo
#endif
