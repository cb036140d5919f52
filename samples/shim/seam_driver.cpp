// seam_driver.cpp — EXECUTES the reference's kernel seam on the shim: the free functions of
// `namespace cugo::gpu` (ref: src/cuda/cuda_block_solver.h:55-256) and the Hsc linear solver are
// called in exactly the order BlockSolver and the LM loop call them
// (ref: src/block_solver.cpp:250-421 computeErrors / buildSystem / maxDiagonal / setLambda / solve /
// update / computeScale / restoreDiagonal / push / pop; src/cuda_graph_optimisation.cpp:60-147),
// with the reference's container types (shim_types.h stand-ins) as arguments:
//
//   iteration 0 : computeErrors, buildSystem, lambda = tau * maxDiagonal, ONE trial (accepted by the
//                 caller's data: a well-posed BA step), estimates swapped
//   iteration 1 : computeErrors, buildSystem, a trial that is FORCED down the reject path whatever its
//                 rho (lambda *= nu, nu *= 2, restoreDiagonal x2, pop), then the retry at the new lambda
//
// tests/test_boundary.py builds this file (it includes the shim translation unit: the solver class
// lives there) into a shared library, hands it a flattened problem through the C ABI's own device
// allocations and compares every number it returns with the CPU oracle.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "shim_cugo_hip.cpp"

namespace
{
using namespace cugo;

struct Dev
{
    std::vector<void*> owned;
    template <typename T>
    T* alloc(size_t n)
    {
        void* p = nullptr;
        shim::ok(cugo_malloc(&p, std::max<size_t>(n, 2) * sizeof(T)), "cugo_malloc");
        shim::ok(cugo_memset(shim::state().ctx, p, 0, std::max<size_t>(n, 2) * sizeof(T)), "cugo_memset");
        owned.push_back(p);
        return static_cast<T*>(p);
    }
    ~Dev()
    {
        for (void* p : owned)
            cugo_free(p);
    }
};
} // namespace

// out[0..] = F0, lambda0, ok1, Fhat1, scale1, F1, okA, FhatA, scaleA, lambdaB, okB, FhatB, scaleB
// h_xp1 / h_xl1: the step of the accepted trial; h_xpB / h_xlB: the step of the retry.
// d_pose / d_lm: [Pall][7], [Lall][3] current estimates (device, through cugo_malloc); overwritten.
extern "C" int cugo_seam_two_iterations(const cugo_edges* ev, const cugo_hsc_struct* hs, const int* h_rowptr,
                                        const int* h_colind, double* d_pose, double* d_lm, double tau, double* out,
                                        double* h_xp1, double* h_xl1, double* h_xpB, double* h_xlB)
{
    using namespace cugo;
    try
    {
        if (!shim::state().ctx)
            shim::open(-1);
        shim::bind_edges(*ev);
        shim::bind_hsc(*hs);
        cugo_ctx* ctx = shim::state().ctx;
        const int E = ev->n_edges, P = ev->n_poses_free, L = ev->n_landmarks_free;
        const int Pall = ev->n_poses_total, Lall = ev->n_landmarks_total, B = hs->n_blocks;
        Dev dev;
        // the buffers BlockSolver owns (ref: block_solver.h:175-247)
        GpuPxPBlockVec Hpp{dev.alloc<double>(36 * (size_t)P), P}, HppBackupDummy{};
        GpuPx1BlockVec bp{dev.alloc<double>(6 * (size_t)P), P}, bsc{dev.alloc<double>(6 * (size_t)P), P},
            xp{nullptr, P}, HppBackup{dev.alloc<double>(6 * (size_t)P), P};
        GpuLxLBlockVec Hll{dev.alloc<double>(9 * (size_t)L), L}, invHll{dev.alloc<double>(9 * (size_t)L), L};
        GpuLx1BlockVec bl{dev.alloc<double>(3 * (size_t)L), L}, xl{nullptr, L}, HllBackup{dev.alloc<double>(3 * (size_t)L), L};
        GpuPxLBlockVec Hpl_invHll{dev.alloc<double>(18 * (size_t)E), E};
        GpuHplBlockMat Hpl;
        Hpl.ptr = dev.alloc<double>(18 * (size_t)E), Hpl.nnz_ = E;
        GpuHscBlockMat Hsc;
        Hsc.ptr = dev.alloc<double>(36 * (size_t)B), Hsc.nnz_ = B, Hsc.brows_ = Hsc.bcols_ = P;
        (void)HppBackupDummy;
        // d_x_ = [xp | xl], d_b_ = [bp | bl] are single buffers in the reference (block_solver.cpp:99-117)
        double* d_x = dev.alloc<double>(6 * (size_t)P + 3 * (size_t)L);
        xp.ptr = d_x, xl.ptr = d_x + 6 * (size_t)P;
        GpuVec1d xvec{d_x, (size_t)(6 * P + 3 * L)}, bvec{bp.ptr, (size_t)(6 * P)}, HscCSR{}, chi{dev.alloc<double>(4), 1};
        GpuVec1i BSR2CSR{}, edge2Hpl{}, outliers{};
        GpuVec3i mulIds{};
        GpuVec1b flags{};
        GpuVec2i edge2PL{};
        GpuVec5d cameras{};
        GpuVec1d omegas{};
        GpuVecxd<2> meas2{}, err2{};
        GpuVecxd<3> meas3{}, err3{};
        GpuVec3d Xcs{};
        // estimates: current and backup (push / pop), ref: d_solution_ / d_solutionBackup_
        double* pose_buf[2] = {d_pose, dev.alloc<double>(7 * (size_t)Pall)};
        double* lm_buf[2] = {d_lm, dev.alloc<double>(3 * (size_t)Lall)};
        shim::ok(cugo_memcpy_d2d(ctx, pose_buf[1], pose_buf[0], sizeof(double) * 7 * (size_t)Pall), "cugo_memcpy_d2d");
        shim::ok(cugo_memcpy_d2d(ctx, lm_buf[1], lm_buf[0], sizeof(double) * 3 * (size_t)Lall), "cugo_memcpy_d2d");
        int cur = 0;
        auto poses = [&](int k) { return GpuVecSe3d{reinterpret_cast<Se3Pod*>(pose_buf[k]), (size_t)Pall}; };
        auto lms = [&](int k) { return GpuVec3d{reinterpret_cast<Vec3dPod*>(lm_buf[k]), (size_t)Lall}; };
        const CudaDeviceInfo di{};
        RobustKernel rk{};
        gpu::createRkFunction(RobustKernelType::None, GpuVec1d{}, 1.0, di);

        HscSparseLinearSolver solver;
        HschurSparseBlockMatrix pat;
        pat.brows_ = P, pat.outer = h_rowptr, pat.inner = h_colind;
        solver.initialize(pat, di); // ref: BlockSolver::buildStructure, block_solver.cpp:226-234

        // ref: BlockSolver::computeErrors — one call per edge set (mono, stereo), chi2 summed on the host
        auto computeErrors = [&](int k) {
            shim::bind_estimates(pose_buf[k], lm_buf[k], pose_buf[1 - k], lm_buf[1 - k], xl.ptr);
            double f = gpu::computeActiveErrors_<2>(poses(k), lms(k), meas2, omegas, edge2PL, cameras, rk, outliers, err2, Xcs,
                                                    nullptr, chi.ptr, di);
            f += gpu::computeActiveErrors_<3>(poses(k), lms(k), meas3, omegas, edge2PL, cameras, rk, outliers, err3, Xcs,
                                              nullptr, chi.ptr, di);
            return f;
        };
        // ref: BlockSolver::buildSystem
        auto buildSystem = [&](int k) {
            shim::bind_estimates(pose_buf[k], lm_buf[k], pose_buf[1 - k], lm_buf[1 - k], xl.ptr);
            gpu::constructQuadraticForm_<2>(Xcs, poses(k), err2, omegas, edge2PL, edge2Hpl, flags, cameras, rk, outliers,
                                            Hpp, bp, Hll, bl, Hpl, di);
            gpu::constructQuadraticForm_<3>(Xcs, poses(k), err3, omegas, edge2PL, edge2Hpl, flags, cameras, rk, outliers,
                                            Hpp, bp, Hll, bl, Hpl, di);
        };
        // one LM trial from the estimates in buffer k (ref: cuda_graph_optimisation.cpp:77-92):
        // push, setLambda, solve, update, computeErrors at the updated estimates, computeScale
        auto trial = [&](int k, double lambda, double& Fhat, double& scale) {
            // push(): the estimates are kept in buffer k, the update lands in the other buffer
            shim::bind_estimates(pose_buf[k], lm_buf[k], pose_buf[1 - k], lm_buf[1 - k], xl.ptr);
            gpu::addLambda(Hpp, lambda, HppBackup, di);
            gpu::addLambda(Hll, lambda, HllBackup, di);
            gpu::computeBschure(bp, Hpl, Hll, bl, bsc, invHll, Hpl_invHll, di);
            gpu::computeHschure(Hpp, Hpl_invHll, Hpl, mulIds, Hsc, di);
            gpu::convertHschureBSRToCSR(Hsc, BSR2CSR, HscCSR, di);
            const bool ok = solver.solve(Hsc.values(), bsc.values(), xp.values());
            if (!ok)
                return false; // ref: block_solver.cpp:374-378
            gpu::schurComplementPost(invHll, bl, Hpl, xp, xl, di);
            auto pk = poses(k);
            auto lk = lms(k);
            gpu::updatePoses(xp, pk, di);
            gpu::updateLandmarks(xl, lk, di);
            Fhat = computeErrors(1 - k); // the updated estimates
            gpu::computeScale(xvec, bvec, chi.ptr, lambda, di);
            shim::ok(cugo_memcpy_d2h(ctx, &scale, chi.ptr, sizeof scale), "cugo_memcpy_d2h");
            scale += 1e-3;
            return true;
        };
        auto download_step = [&](double* hxp, double* hxl) {
            shim::ok(cugo_memcpy_d2h(ctx, hxp, xp.ptr, sizeof(double) * 6 * (size_t)P), "cugo_memcpy_d2h");
            shim::ok(cugo_memcpy_d2h(ctx, hxl, xl.ptr, sizeof(double) * 3 * (size_t)L), "cugo_memcpy_d2h");
        };

        // ---- iteration 0
        double F = computeErrors(cur);
        buildSystem(cur);
        // ref: BlockSolver::maxDiagonal — Hpp first, then Hll (two statements: the order matters to the shim,
        // whose second call returns the joint maximum)
        const double maxP = gpu::maxDiagonal(Hpp, nullptr, nullptr, di);
        const double maxL = gpu::maxDiagonal(Hll, nullptr, nullptr, di);
        double lambda = tau * std::max(maxP, maxL);
        double nu = 2;
        out[0] = F, out[1] = lambda;
        double Fhat = 0, scale = 0;
        const bool ok1 = trial(cur, lambda, Fhat, scale);
        out[2] = ok1, out[3] = Fhat, out[4] = scale;
        download_step(h_xp1, h_xl1);
        const double rho = ok1 ? (F - Fhat) / scale : -1;
        if (!(rho > 0))
            throw std::runtime_error("seam driver: the first trial was not accepted (rho <= 0)");
        { // accepted (ref: cuda_graph_optimisation.cpp:94-101): the updated estimates are the estimates
            double alpha = 1.0 - std::pow(2 * rho - 1, 3);
            alpha = std::min(alpha, 2.0 / 3.0);
            lambda *= std::max(1.0 / 3.0, alpha);
            nu = 2;
            F = Fhat;
            cur = 1 - cur;
        }
        // ---- iteration 1: a trial forced down the reject path, then the retry
        const double F1 = computeErrors(cur);
        buildSystem(cur);
        out[5] = F1;
        double FhatA = 0, scaleA = 0;
        const bool okA = trial(cur, lambda, FhatA, scaleA);
        out[6] = okA, out[7] = FhatA, out[8] = scaleA;
        // rejected (ref: cuda_graph_optimisation.cpp:102-112): lambda *= nu, nu *= 2, restoreDiagonal, pop —
        // pop(): the estimates of buffer `cur` were never touched, the trial's are simply dropped
        lambda *= nu;
        nu *= 2;
        gpu::restoreDiagonal(Hpp, HppBackup, di);
        gpu::restoreDiagonal(Hll, HllBackup, di);
        out[9] = lambda;
        double FhatB = 0, scaleB = 0;
        const bool okB = trial(cur, lambda, FhatB, scaleB);
        out[10] = okB, out[11] = FhatB, out[12] = scaleB;
        download_step(h_xpB, h_xlB);
        // hand the estimates of the last trial back in the caller's buffers
        if (1 - cur != 0)
        {
            shim::ok(cugo_memcpy_d2d(ctx, pose_buf[0], pose_buf[1], sizeof(double) * 7 * (size_t)Pall), "cugo_memcpy_d2d");
            shim::ok(cugo_memcpy_d2d(ctx, lm_buf[0], lm_buf[1], sizeof(double) * 3 * (size_t)Lall), "cugo_memcpy_d2d");
        }
        gpu::waitForKernelCompletion();
        return 0;
    }
    catch (const std::exception& e)
    {
        std::fprintf(stderr, "seam driver: %s\n", e.what());
        return -1;
    }
}
