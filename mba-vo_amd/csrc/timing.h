// timing.h -- optional host-side phase timers of the tracking loop (MBAVO_TIMING=1 prints the totals to stderr at exit).
// Development aid for the latency work on the single-pair path; costs two clock reads per phase when enabled.
#ifndef MBAVO_TIMING_H
#define MBAVO_TIMING_H
#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace mbavo
{
    struct PhaseTimers
    {
        enum { kUpload, kKeyframe, kEnqueue, kWait, kMerge, kSolve, kOutliers, kLevel, kTrack, kFlagsChanged, kFlagsSame, kCount };
        double sec[kCount] = {};
        long calls[kCount] = {};
        bool on;
        PhaseTimers() { const char *v = getenv("MBAVO_TIMING"); on = v && *v && *v != '0'; }
        ~PhaseTimers() { report(); }
        void report()
        {
            if (!on) return;
            static const char *names[kCount] = {"upload+pyramid", "keyframe processing", "evaluate enqueue", "evaluate wait", "host merge",
                                                "host solve+step", "outlier detection", "level setup", "trackFrame (all)", "accepted: flags changed", "accepted: flags same"};
            for (int i = 0; i < kCount; ++i)
                if (calls[i]) fprintf(stderr, "mbavo timing: %-20s %8ld calls  %9.3f ms total  %7.2f us each\n", names[i], calls[i], sec[i] * 1e3, sec[i] * 1e6 / calls[i]);
            for (int i = 0; i < kCount; ++i) { sec[i] = 0; calls[i] = 0; }
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
            t.sec[id] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            ++t.calls[id];
        }
    };
} // namespace mbavo
#endif
