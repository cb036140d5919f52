// cugo_chol_*: sparse block LL^T solver object of the C ABI (include/cugo_hip.h).
// ref: HscSparseLinearSolver (src/cuda_linear_solver.cpp:27-57) + CuSparseCholeskySolver
// (src/cholesky.hpp:170-309): initialize(pattern) once, solve(A, b, x)->bool per LM trial.
#include "chol_solver.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

using namespace cugo_host;

void cugo_chol::upload(hipStream_t s)
{
    const CholPlan& P = plan;
    d_ncb.upload(P.ncb, s), d_nb.upload(P.nb, s), d_off.upload(P.off, s), d_col0.upload(P.col0, s);
    d_woff.upload(P.woff, s), d_l21off.upload(P.l21off, s), d_ldf.upload(P.ldf, s);
    d_alias_of.upload(P.alias_of, s), d_bw_np.upload(P.bw_np, s);
    d_rows_ptr.upload(P.rows_ptr, s), d_rows.upload(P.rows, s);
    d_child_ptr.upload(P.child_ptr, s), d_child.upload(P.child, s);
    d_rel_ptr.upload(P.rel_ptr, s), d_rel.upload(P.rel, s);
    d_task_ptr.upload(P.task_ptr, s), d_task_fronts.upload(P.task_fronts, s);
    d_blk_front.upload(P.blk_front, s), d_blk_row.upload(P.blk_row, s);
    d_blk_col.upload(P.blk_col, s), d_blk_trans.upload(P.blk_trans, s);
    d_perm.upload(P.perm, s), d_col_front.upload(P.col_front, s);
    d_wl.upload(P.wl, s);
    d_fronts.resize((size_t)P.front_doubles + 16);
    d_xnew.resize((size_t)6 * P.n + 16);
    d_junk.resize(64 * 1024);
    d_winv.resize((size_t)P.winv_doubles + 16);
    d_l21.resize((size_t)P.l21_doubles + 16);
    CUGO_HIP(hipStreamSynchronize(s)); // host vectors may be reused after return

    cugo_k::CholPlanDev& D = dev;
    D.n_fronts = P.n_super;
    D.ncb = d_ncb.data(), D.nb = d_nb.data(), D.off = d_off.data(), D.col0 = d_col0.data();
    D.rows_ptr = d_rows_ptr.data(), D.rows = d_rows.data();
    D.child_ptr = d_child_ptr.data(), D.child = d_child.data();
    D.rel_ptr = d_rel_ptr.data(), D.rel = d_rel.data();
    D.n_stages = P.n_stages;
    D.task_ptr = d_task_ptr.data(), D.task_fronts = d_task_fronts.data();
    D.n_hsc_blocks = (int)P.blk_front.size();
    D.blk_front = d_blk_front.data(), D.blk_row = d_blk_row.data();
    D.blk_col = d_blk_col.data(), D.blk_trans = d_blk_trans.data();
    D.n = P.n, D.perm = d_perm.data(), D.col_front = d_col_front.data();
    D.junk = d_junk.data();
    D.woff = d_woff.data(), D.winv = d_winv.data(), D.nc_max = P.nc_max;
    D.l21off = d_l21off.data(), D.l21 = d_l21.data();
    D.ldf = d_ldf.data(), D.alias_of = d_alias_of.data(), D.bw_np = d_bw_np.data();
    lds_factor = cugo_k::chol_lds_factor_bytes(P.nc_max);
    lds_backward = cugo_k::chol_lds_backward_bytes(P.nc_max, P.ld_max);
}

void cugo_chol::analyze(int n, const int32_t* rowptr, const int32_t* colind)
{
    chol_analyze(n, rowptr, colind, CholOptions::from_env(), plan);
    trans32.assign(plan.blk_trans.begin(), plan.blk_trans.end());
    if (plan.nc_max > cugo_k::chol_max_pivot_cols() ||
        cugo_k::chol_lds_factor_bytes(plan.nc_max) > 160 * 1024 ||
        cugo_k::chol_lds_backward_bytes(plan.nc_max, plan.ld_max) > 160 * 1024)
        throw std::runtime_error("cugo: a front exceeds the LDS budget (nc=" +
                                 std::to_string(plan.nc_max) + ", ld=" + std::to_string(plan.ld_max) + ")");
    if (ctx)
        upload(ctx->stream);
    analyzed = true;
}

void cugo_chol::factor_solve(const double* d_Hsc, double lambda, const double* d_bsc, double* d_x,
                             int32_t* d_fail)
{
    hipStream_t s = ctx->stream;
    static const bool dbg = std::getenv("CUGO_DEBUG_STAMPS") != nullptr;
    static long long* d_stamps = nullptr;
    if (dbg && !d_stamps)
    {
        CUGO_HIP(hipMalloc(reinterpret_cast<void**>(&d_stamps), 64 * sizeof(long long)));
        CUGO_HIP(hipMemset(d_stamps, 0, 64 * sizeof(long long)));
        cugo_k::set_debug_stamps(d_stamps);
    }
    cugo_k::launch_chol_assemble(s, dev, d_fronts.data(), (size_t)plan.front_doubles, d_Hsc, lambda,
                                 d_bsc, d_fail, false);
    for (int st = 0; st < plan.n_stages; st++)
    {
        const int t0 = plan.stage_task_ptr[st], t1 = plan.stage_task_ptr[st + 1];
        if (plan.has_subtree_stage && st == 0)
            cugo_k::launch_chol_subtree_stage(s, dev, d_fronts.data(), t0, t1 - t0, lds_factor, d_fail);
        else
            cugo_k::launch_chol_upper_stage(
                s, dev, d_fronts.data(), t0, t1 - t0, d_wl.data(), plan.ea_ptr[st],
                plan.ea_ptr[st + 1] - plan.ea_ptr[st], plan.eab_ptr[st],
                plan.eab_ptr[st + 1] - plan.eab_ptr[st], plan.syrk_ptr[st],
                plan.syrk_ptr[st + 1] - plan.syrk_ptr[st], lds_factor, d_fail);
    }
    for (int st = plan.n_stages - 1; st >= 0; st--)
    {
        const int t0 = plan.stage_task_ptr[st], t1 = plan.stage_task_ptr[st + 1];
        // the level's fronts, plus the ahead-of-time mat-vecs of their children as extra workgroups
        cugo_k::launch_chol_backward_stage(s, dev, d_fronts.data(), t0, t1 - t0, lds_backward, d_xnew.data(),
                                           d_x, d_wl.data() + 3L * plan.bwg_ptr[st],
                                           plan.bwg_ptr[st + 1] - plan.bwg_ptr[st]);
        const int st_top = st;
        if (dbg && st_top == plan.n_stages - 1)
        { // keep the top stage's backward stamps (kernel 3) in slots 48.. before stage 0 overwrites them
            CUGO_HIP(hipStreamSynchronize(s));
            CUGO_HIP(hipMemcpy(d_stamps + 48, d_stamps + 24, 8 * sizeof(long long), hipMemcpyDeviceToDevice));
        }
    }
    CUGO_HIP(hipGetLastError());
    if (dbg)
    {
        long long h[64];
        CUGO_HIP(hipStreamSynchronize(s));
        CUGO_HIP(hipMemcpy(h, d_stamps, sizeof h, hipMemcpyDeviceToHost));
        static int calls = 0;
        if (++calls == 5)
            for (int k = 0; k < 7; k++)
            {
                std::printf("stamps kernel %d (nc=%lld):", k, h[k * 8 + 6]);
                for (int i = 1; i < 8; i++)
                    if (i != 6 && h[k * 8 + i])
                        std::printf(" [%d] +%lld", i, h[k * 8 + i] - h[k * 8]);
                std::printf("  (cycles since kernel start; last launch of the kernel)\n");
            }
    }
}
