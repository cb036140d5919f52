// Run-time switches of the library, read from the environment ONCE per object: a context (cugo_ctx_create), a
// solver (cugo_chol_create) and an optimiser (Engine) each take a snapshot when they are created; nothing is read
// per launch or per LM trial.  The defaults are the measured optimum on MI355X (README.md lists the variables).
#pragma once
#include <cstdlib>

namespace cugo_host
{

struct Options
{
    // ---- Schur complement (ba_kernels.hip: launch_schur) ----
    int hsc_mfma = 1;      // CUGO_HSC_MFMA: 1 both H-side kernels on the matrix cores, 2 only the off-diagonal one, 0 vector lanes
    bool hsc_xcd = true;   // CUGO_HSC_XCD=0: Hsc blocks in dispatch order instead of contiguous ranges per XCD
    bool hsc_rows = false, hsc_strip = false, schur_plan = false; // CUGO_HSC_ROWS / CUGO_HSC_STRIP / CUGO_SCHUR_PLAN
    bool fuse_t = true;    // CUGO_FUSE_T=0: T = Hpl invHll always by the Schur edge kernel
    bool pose_schur = true; // CUGO_POSE_SCHUR=0: Hsc's diagonal blocks always by k_build_poses + k_hsc_diag*, never by k_pose_schur
    bool float32 = false;  // CUGO_FLOAT32=1
    // ---- LM loop ----
    bool speculate = true, trial_event = true, trial_poll = true; // CUGO_SPECULATE / CUGO_TRIAL_EVENT / CUGO_TRIAL_POLL = 0
    bool trial_from_build = true; // CUGO_TRIAL_FROM_BUILD=0: every trial ends with an error pass of its own
    bool profile = false;  // CUGO_PROFILE
    // ---- initialize() / structure ----
    bool init_timing = false;       // CUGO_INIT_TIMING
    bool structure_reuse = true;    // CUGO_NO_STRUCTURE_REUSE
    bool flatten_reuse = true;      // CUGO_NO_FLATTEN_REUSE
    bool async_structure = true;    // CUGO_ASYNC_STRUCTURE=0
    bool host_structure = false;    // CUGO_HOST_STRUCTURE
    bool upload_thread = true;      // CUGO_UPLOAD_THREAD=0
    // ---- sparse LL^T (chol_solver.cpp; the ordering / supernode knobs are CholOptions::from_env) ----
    int own_subtrees = -1;          // CUGO_OWN_SUBTREES: 1 force, 0 forbid, -1 by work (own_min_gflop)
    double own_min_gflop = 3.0;     // CUGO_OWN_MIN_GFLOP
    bool lookahead = false;         // CUGO_LOOKAHEAD
    bool ea_lds = true, panel16 = true, asm_fronts = true; // CUGO_EA_LDS / CUGO_PANEL16 / CUGO_ASM_FRONTS = 0
    // ---- multi-GPU exchange (engine.cpp) ----
    bool reduce_scatter = true;     // CUGO_REDUCE_SCATTER=0: all-reduce of [Hsc | bsc] also with rank-owned subtrees

    static Options from_env()
    {
        Options o;
        auto flag = [](const char* name) { return std::getenv(name) != nullptr; };
        auto off = [](const char* name) {
            const char* e = std::getenv(name);
            return e && e[0] == '0';
        };
        auto on = [](const char* name) {
            const char* e = std::getenv(name);
            return e && e[0] == '1';
        };
        if (const char* e = std::getenv("CUGO_HSC_MFMA"))
            o.hsc_mfma = e[0] == '0' ? 0 : e[0] == '2' ? 2 : 1;
        o.hsc_xcd = !off("CUGO_HSC_XCD");
        o.schur_plan = flag("CUGO_SCHUR_PLAN");
        o.hsc_rows = on("CUGO_HSC_ROWS"), o.hsc_strip = on("CUGO_HSC_STRIP");
        o.fuse_t = !off("CUGO_FUSE_T");
        o.pose_schur = !off("CUGO_POSE_SCHUR");
        o.float32 = on("CUGO_FLOAT32");
        o.speculate = !off("CUGO_SPECULATE"), o.trial_event = !off("CUGO_TRIAL_EVENT"), o.trial_poll = !off("CUGO_TRIAL_POLL");
        o.trial_from_build = !off("CUGO_TRIAL_FROM_BUILD");
        o.profile = flag("CUGO_PROFILE");
        o.init_timing = flag("CUGO_INIT_TIMING");
        o.structure_reuse = !flag("CUGO_NO_STRUCTURE_REUSE");
        o.flatten_reuse = !flag("CUGO_NO_FLATTEN_REUSE");
        o.async_structure = !off("CUGO_ASYNC_STRUCTURE");
        o.host_structure = flag("CUGO_HOST_STRUCTURE");
        o.upload_thread = !off("CUGO_UPLOAD_THREAD");
        if (const char* e = std::getenv("CUGO_OWN_SUBTREES"))
            o.own_subtrees = e[0] == '1' ? 1 : e[0] == '0' ? 0 : -1;
        if (const char* e = std::getenv("CUGO_OWN_MIN_GFLOP"))
            o.own_min_gflop = std::atof(e);
        if (const char* e = std::getenv("CUGO_LOOKAHEAD"))
            o.lookahead = std::atoi(e) != 0;
        o.ea_lds = !off("CUGO_EA_LDS"), o.panel16 = !off("CUGO_PANEL16"), o.asm_fronts = !off("CUGO_ASM_FRONTS");
        o.reduce_scatter = !off("CUGO_REDUCE_SCATTER");
        return o;
    }
};

} // namespace cugo_host
