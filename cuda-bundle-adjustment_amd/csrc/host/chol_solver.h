#pragma once
#include "../kernels/kernels.h"
#include "chol_symbolic.h"
#include "hip_util.h"
#include "options.h"

#include <array>
#include <functional>
#include <memory>

// definition of the opaque cugo_chol of include/cugo_hip.h
struct cugo_chol
{
    cugo_ctx* ctx = nullptr;
    bool analyzed = false;
    cugo_host::Options opt = cugo_host::Options::from_env(); // the environment switches, read once when the solver is created
    cugo_host::CholPlan plan;
    std::vector<int32_t> trans32; // blk_trans widened for cugo_chol_plan_array
    std::vector<int32_t> asm_info; // (cugo_chol_plan_array)
    cugo_k::CholPlanDev dev{};
    size_t lds_factor = 0, lds_backward = 0;
    // CUGO_LOOKAHEAD=1 at analyze(): the bulk of a level's update matrix is computed while the next level
    // factors (k_up_potrf_la / k_up_lead).  Measured slower than the level-synchronous default on MI355X
    // (kitti_00 shape: 17.5 vs 14.7 ms per step, profiles/r02_lookahead_timeline.txt): the lead workgroup
    // that the next potrf waits for costs as much as a whole single-round tile launch.  Kept under test.
    bool lookahead = false;
    bool asm_fronts = true; // one assembly launch that writes every entry (k_assemble_fronts) instead of clear + scatter

    // the plan's index arrays, packed (chol_solver.cpp: upload)
    std::vector<int32_t> pack32;
    std::vector<int64_t> pack64;
    std::array<size_t, 30> po{}; // offsets of the arrays inside the packs (pack)
    cugo_host::DevBuf<int32_t> d_pack32;
    cugo_host::DevBuf<int64_t> d_pack64;
    const int32_t* d_wl_ptr = nullptr; // work-item triples inside d_pack32
    cugo_host::DevBuf<double> d_fronts, d_xnew, d_junk, d_winv, d_l21;

    // rank-owned elimination subtrees of a landmark-sharded run (CholPlan::owner): this rank factors its own
    // subtrees and the replicated top; update blocks that enter the top and the solution of the other ranks'
    // subtrees arrive by `bcast(device pointer, doubles, root rank)` on the solver's stream
    int rank = 0, world = 1;
    std::function<void(double*, size_t, int)> bcast;
    std::function<void(bool)> bcast_group; // brackets the broadcasts that may be fused into one operation
    ~cugo_chol();
#ifdef CUGO_DEBUG_HOOKS // ---- diagnosis, only in libcugo_hip_hooks.so (make HOOKS=1) ----
    // diagnosis (CUGO_DEBUG_HASH): 64 checksum slots of the factorisation being queued, or null —
    // 11 fronts after the assembly, 12 / 13 / 14 W, L21, fronts after the forward pass, 15 x after the backward pass,
    // 16 + st: W after the potrf launch of stage st, 40 + st: the fronts after the tile launches of stage st
    unsigned long long* dbg_hash = nullptr;
    int dbg_calls = 0; // factor_solve calls so far (CUGO_DEBUG_SKIP counts them)
    // diagnosis (CUGO_DEBUG_KEEP=1): after every one of the first 16 factor_solve calls the fronts, W, L21, x (permuted)
    // and x are copied (same stream) into a slot of their own; cugo_debug_dump() writes the slots of the solver that
    // ran last to files — the autopsy of a run that deviated from its twin (tools/autopsy.py)
    std::vector<std::unique_ptr<cugo_host::DevBuf<double>>> keep;
    cugo_host::DevBuf<double> dbg_scratch; // (CUGO_DEBUG_STALE: the other version of the line under test)
    void dump_kept(const char* dir);
    void dump_slot(int call, const char* path);
#endif
    bool own_subtrees() const { return bcast && plan.owned; }

    void analyze(int n, const int32_t* rowptr, const int32_t* colind);
    void analyze_host(int n, const int32_t* rowptr, const int32_t* colind); // without upload()
    void pack(); // host half of upload() (analyze_host calls it)
    void upload(hipStream_t s);
    void factor_solve(const double* d_Hsc, double lambda, const double* d_bsc, double* d_x,
                      int32_t* d_fail);
};

#ifdef CUGO_DEBUG_HOOKS
// diagnosis: writes the CUGO_DEBUG_KEEP slots of the solver that ran last to dir/call<k>.bin; returns their number
int cugo_debug_dump_last_solver(const char* dir);
cugo_chol* cugo_debug_solver(int which); // 0: the solver that ran last with CUGO_DEBUG_KEEP, 1: the pinned one
void cugo_debug_pin_reference_solver();
#endif
