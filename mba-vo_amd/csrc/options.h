// options.h -- every switch that changes results or scheduling, as DATA.
// The reference configures its tracker through one plain struct (blur_aware_direct_tracker.h:15-67); a library whose solver
// depended on the process environment would not be a drop-in.  The switches live in the option structs of include/mbavo.h
// (mbavo_engine_opts on the context, the tails of mbavo_track_opts / mbavo_vo_options / mbavo_lm_batch_opts); a zeroed struct is
// the default everywhere: tri-state ints 0 = default, 1 = on, -1 = off; numbers 0 = default.
// The environment variables of the A/B tools still OVERRIDE an option, and read_env_overrides() below is the ONE function that
// reads them (once per process; a tool that flips one inside a process calls mbavo_reload_env afterwards).  Pure diagnostics -- MBAVO_TIMING,
// MBAVO_LM_STAMPS, MBAVO_LM_STATS -- change no result and no schedule and stay environment-only where they are used.
#ifndef MBAVO_OPTIONS_H
#define MBAVO_OPTIONS_H

#include <climits>

namespace mbavo
{
    constexpr int kEnvUnset = INT_MIN;

    struct EnvOverrides
    { // kEnvUnset / a negative ratio sentinel (-2) = the variable is not set
        int sp, one, fused_pose, fused_pose_max_s, persist, prelaunch, tiles_per_cu, min_tile_px, sp_max_slot_tiles; // engine
        int speculate, persist_levels, kf_multi, kf_speculate, ride_along, resum;                                                                   // host LM loop, front end
        int lm_eig, lm_poses, lm_defer, lm_retile, lm_groups;                                                       // batched LM
        double fast_solve, lm_refine; // the variable's number; -2: unset
    };
    EnvOverrides read_env_overrides(); // host_math.cpp: the process's environment, scanned once and cached
    void reload_env_overrides();       // ... scanned again (mbavo_reload_env: the A/B tools call it after they change a variable)

    // tri-state option (0 default / 1 on / -1 off) under an environment override (any integer: non-zero = on)
    inline bool opt_flag(int option, int env, bool dflt) { return env != kEnvUnset ? env != 0 : (option == 0 ? dflt : option > 0); }
    // numeric option (0 = default)
    inline int opt_number(int option, int env, int dflt) { return env != kEnvUnset ? env : (option > 0 ? option : dflt); }
    // Pivot ratio up to which LDL^T stands in for the reference's solvers (host_math.cpp: solve_spd_fast; lm_solvers.h):
    // option 0 = 1e8, < 0 = never, > 1 = that ratio.  MBAVO_FAST_SOLVE: 0 never, 1 default, > 1 the ratio.
    inline double opt_fast_ratio(double option, double env)
    {
        if (env > -2.0) return env > 1.0 ? env : (env == 1.0 ? 1e8 : 0.0);
        return option < 0.0 ? 0.0 : (option > 1.0 ? option : 1e8);
    }
    // Ratio up to which the REFINED stand-in is admitted (double-double residuals): option 0 = 1e13, < 0 = never, > 1 that ratio;
    // nothing without the plain stand-in.  MBAVO_LM_REFINE: 0 never, 1 default, > 1 the ratio.
    inline double opt_refined_ratio(double option, double env, double fast_ratio)
    {
        if (fast_ratio <= 0.0) return 0.0;
        if (env > -2.0) return env > 1.0 ? env : (env == 1.0 ? 1e13 : 0.0);
        return option < 0.0 ? 0.0 : (option > 1.0 ? option : 1e13);
    }

    // The evaluation engine's scheduling choices, resolved (mbavo_engine_opts + environment): Engine::tuning()
    struct EngineTuning
    {
        int sample_parallel = -1;       // -1 auto (by size), 0 never, 1 always where the list allows it   [MBAVO_SP]
        bool single_launch = true;      // small lists: pose prologue + ticket epilogue in ONE launch       [MBAVO_ONE]
        bool fused_pose = true;         // pose entries as the fused kernel's prologue                      [MBAVO_FUSED_POSE]
        int fused_pose_max_samples = 8; //                                                                  [MBAVO_FUSED_POSE_MAX_S]
        bool persistent = true;         // persistent evaluation kernels for the host-driven LM loop        [MBAVO_PERSIST]
        bool prelaunch = true;          // next level's persistent kernel enqueued behind the running one   [MBAVO_PRELAUNCH]
        int tiles_per_cu = 1;           //                                                                  [MBAVO_TILES_PER_CU]
        int min_tile_pixels = 256;      //                                                                  [MBAVO_MIN_TILE_PX]
        int sp_max_slot_tiles = 64;     //                                                                  [MBAVO_SP_MAX_SLOT_TILES]
    };
} // namespace mbavo

#endif
